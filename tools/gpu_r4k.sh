#!/bin/bash
# Round 3, visit 4k: cubic tail of 1024 pairs (first turn on register-resident arrays): parity, A/B
OUT=gpurun_out/r4k; mkdir -p $OUT
timeout 200 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "tail or abort" > $OUT/pytest_tail.log 2>&1; grep -E "passed|failed|error" $OUT/pytest_tail.log | tail -2; grep -E "^FAILED|Error" $OUT/pytest_tail.log | head -3
LASSO_TEST_CURVE=bn254 timeout 200 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "tail" > $OUT/pytest_tail_bn254.log 2>&1; grep -E "passed|failed|error" $OUT/pytest_tail_bn254.log | tail -2
timeout 200 python -m pytest tests/test_gpu_prover.py -m gpu -q -x -k "bit_exact_vs_oracle or concurrent" > $OUT/pytest_proofs.log 2>&1; grep -E "passed|failed|error" $OUT/pytest_proofs.log | tail -2
for i in 1 2 3 4 5; do for h in 1 0; do LASSO_TAIL_HEAD=$h timeout 100 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --concurrent 0 --no-slab-leg --no-prof 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('head=$h %.3f' % d['ms_per_step'])"; done; done | tee $OUT/results.txt
python - <<'PY'
import collections
d=collections.defaultdict(list)
for l in open('gpurun_out/r4k/results.txt'):
    a=l.split(); d[a[0]].append(float(a[1]))
for k,v in d.items():
    v.sort(); print(k,'mean %.3f median %.3f min %.3f'%(sum(v)/len(v), v[len(v)//2], v[0]))
PY
timeout 150 python bench.py --steps 5 --warmup 2 --concurrent 0 --no-slab-leg 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('parity',d['parity_checked']['equal'],d['parity_checked']['commitment_equal'])"
exit 0
