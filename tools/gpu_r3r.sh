#!/bin/bash
# Round 3, visit r: resident tail turn, step 1 (one product per term lane, operands of the bind fetched while the challenge travels): parity + bench
OUT=gpurun_out/r3r; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q > $OUT/pytest_kernels.log 2>&1; grep -E "passed|failed|error" $OUT/pytest_kernels.log | tail -2
LASSO_TEST_CURVE=bn254 timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "tail or cubic" > $OUT/pytest_kernels_bn254.log 2>&1; grep -E "passed|failed|error" $OUT/pytest_kernels_bn254.log | tail -2
timeout 60 tools/tail_bench | tee $OUT/tail_bench.txt
for i in 1 2 3; do python bench.py --steps 20 --warmup 3 --no-cpu-baseline --concurrent 0 --no-slab-leg --no-prof 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('and ms',d['ms_per_step'])"; done | tee $OUT/results.txt
timeout 600 python bench.py --no-slab-leg --concurrent 0 > $OUT/bench_default.json 2> $OUT/bench_default.err; python -c "
import json;d=json.load(open('$OUT/bench_default.json'));print('ms',d['ms_per_step'],'parity',d['parity_checked'].get('equal'),d['parity_checked'].get('commitment_equal'))"
exit 0
