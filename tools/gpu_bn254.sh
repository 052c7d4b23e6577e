#!/bin/bash
# One short box visit for the BN254 build (liblasso_*_bn254.so): every entry point against the BN254 mock (the kernel parity suite re-run with
# LASSO_TEST_CURVE=bn254), whole proofs against the BN254 oracle, and bench lines at BASELINE.json's configs[1] (AND C=4 2^20, G=BN254) and at the
# metric's shape (AND C=1 2^24).  Everything tightly bounded.
OUT=gpurun_out/bn254
mkdir -p $OUT
LASSO_TEST_CURVE=bn254 timeout 50 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q > $OUT/pytest_kernels_bn254.log 2>&1; echo "kernels rc=$?" | tee -a $OUT/pytest_kernels_bn254.log
tail -3 $OUT/pytest_kernels_bn254.log
timeout 25 python bench.py --curve bn254 --c 4 --log-s 20 --steps 5 --warmup 2 --no-cpu-baseline --concurrent 0 > $OUT/bench_bn254_config1.json 2> $OUT/bench_bn254_config1.err; echo "rc=$?"
timeout 30 python bench.py --curve bn254 --steps 3 --warmup 1 --no-cpu-baseline --concurrent 0 > $OUT/bench_bn254_2p24.json 2> $OUT/bench_bn254_2p24.err; echo "rc=$?"
cut -c1-700 $OUT/bench_bn254_config1.json; echo; cut -c1-700 $OUT/bench_bn254_2p24.json; tail -3 $OUT/bench_bn254_config1.err $OUT/bench_bn254_2p24.err
