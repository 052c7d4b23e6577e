#!/bin/bash
# Round 3, visit 4e: both factor tables of a large eq table in one launch: parity, bench
OUT=gpurun_out/r4e; mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x > $OUT/pytest_kernels.log 2>&1; grep -E "passed|failed|error" $OUT/pytest_kernels.log | tail -2; grep -E "^FAILED|Error" $OUT/pytest_kernels.log | head -5
LASSO_TEST_CURVE=bn254 timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "eq" > $OUT/pytest_kernels_bn254.log 2>&1; grep -E "passed|failed|error" $OUT/pytest_kernels_bn254.log | tail -2
timeout 300 python -m pytest tests/test_gpu_prover.py -m gpu -q -x -k "concurrent or bit_exact_vs_oracle or at_scale" > $OUT/pytest_proofs.log 2>&1; grep -E "passed|failed|error" $OUT/pytest_proofs.log | tail -2
for i in 1 2 3 4 5 6; do timeout 120 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --concurrent 0 --no-slab-leg --no-prof 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('and %.3f' % d['ms_per_step'])"; done | tee $OUT/results.txt
timeout 200 python bench.py --no-slab-leg --concurrent 0 > $OUT/bench_default.json 2> $OUT/bench_default.err; python -c "
import json;d=json.load(open('$OUT/bench_default.json'));print('ms',d['ms_per_step'],'parity',d['parity_checked'].get('equal'),d['parity_checked'].get('commitment_equal'))"
exit 0
