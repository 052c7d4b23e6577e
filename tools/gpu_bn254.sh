#!/bin/bash
# One short box visit for the BN254 build (liblasso_*_bn254.so): whole proofs against the BN254 oracle next to the curve25519 golden tests (both
# library pairs in one process), every entry point against the BN254 mock (tests/test_gpu_kernels.py re-run with LASSO_TEST_CURVE=bn254, as a
# child process of the last test), and bench lines at BASELINE.json's configs[1] (AND C=4 2^20, G=BN254) and at the metric's shape (AND C=1 2^24).
OUT=gpurun_out/bn254
mkdir -p $OUT
timeout 60 python -m pytest tests/test_golden.py tests/test_gpu_bn254.py -m gpu -x -q -s > $OUT/pytest_bn254.log 2>&1; echo "rc=$?" | tee -a $OUT/pytest_bn254.log
grep -a "bn254\]\|passed\|failed\|Error\|error" $OUT/pytest_bn254.log | tail -8
timeout 20 python bench.py --curve bn254 --c 4 --log-s 20 --steps 5 --warmup 2 --no-cpu-baseline --concurrent 0 > $OUT/bench_bn254_config1.json 2> $OUT/bench_bn254_config1.err; echo "rc=$?"
timeout 25 python bench.py --curve bn254 --steps 3 --warmup 1 --no-cpu-baseline --concurrent 0 > $OUT/bench_bn254_2p24.json 2> $OUT/bench_bn254_2p24.err; echo "rc=$?"
cut -c1-330 $OUT/bench_bn254_config1.json; echo; cut -c1-330 $OUT/bench_bn254_2p24.json; echo
exit 0
