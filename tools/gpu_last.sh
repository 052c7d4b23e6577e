#!/bin/bash
# Short box visit: full GPU parity suite on the default path, then the opt-in mid-round kernel (golden parity + A/B bench), all tightly bounded.
OUT=gpurun_out/last
mkdir -p $OUT
timeout 140 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -3 $OUT/pytest_gpu.log
LASSO_CUBIC_MID=1 timeout 40 python -m pytest tests/test_golden.py -m gpu -x -q > $OUT/pytest_mid.log 2>&1; echo "mid golden rc=$?" | tee -a $OUT/pytest_mid.log
tail -3 $OUT/pytest_mid.log
LASSO_CUBIC_MID=1 timeout 30 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-prof --concurrent 0 > $OUT/bench_mid1.json 2> $OUT/bench_mid1.err; echo "rc=$?"
timeout 30 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-prof --concurrent 0 > $OUT/bench_mid0.json 2> $OUT/bench_mid0.err; echo "rc=$?"
cat $OUT/bench_mid1.json $OUT/bench_mid0.json
