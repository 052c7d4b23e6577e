#!/bin/bash
# Round 3, visit 4b: 8-entry pointer tables / no unused EqInline in the cubic launches' kernel arguments: parity + A/B vs the previous build is not possible in one tree, so: absolute numbers, 10 runs
OUT=gpurun_out/r4b; mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x > $OUT/pytest_kernels.log 2>&1; grep -E "passed|failed|error" $OUT/pytest_kernels.log | tail -2; grep -E "^FAILED|Error" $OUT/pytest_kernels.log | head -5
LASSO_TEST_CURVE=bn254 timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "cubic or tail" > $OUT/pytest_kernels_bn254.log 2>&1; grep -E "passed|failed|error" $OUT/pytest_kernels_bn254.log | tail -2
timeout 300 python -m pytest tests/test_gpu_prover.py -m gpu -q -x -k "concurrent or bit_exact_vs_oracle" > $OUT/pytest_proofs.log 2>&1; grep -E "passed|failed|error" $OUT/pytest_proofs.log | tail -2
timeout 40 tools/tail_bench 8192 | tee $OUT/tail_bench.txt
for i in 1 2 3 4 5 6 7 8 9 10; do timeout 120 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --concurrent 0 --no-slab-leg --no-prof 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('and %.3f' % d['ms_per_step'])"; done | tee $OUT/results.txt
python -c "
v=sorted(float(l.split()[1]) for l in open('$OUT/results.txt')); print('mean %.3f median %.3f min %.3f' % (sum(v)/len(v), v[len(v)//2], v[0]))"
exit 0
