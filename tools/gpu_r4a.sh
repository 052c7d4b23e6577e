#!/bin/bash
# Round 3, visit 4a: does the size of the by-value pointer tables (2 x 1088 bytes of kernel arguments per cubic launch) cost anything?  A/B against a build with 16-entry tables
OUT=gpurun_out/r4a; mkdir -p $OUT
run() { local label=$1; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 120 python bench.py "$@" --no-cpu-baseline --concurrent 0 --no-slab-leg --no-prof > $OUT/b.json 2> $OUT/b.err
  python -c "
import json;d=json.load(open('$OUT/b.json'));print('$label %.3f' % d['ms_per_step'])" | tee -a $OUT/results.txt; }
P=$PWD/ab/ptrs16
for i in 1 2 3 4 5 6 7 8; do run "tables136" X=1 -- --steps 20 --warmup 3; run "tables16" LASSO_PROVER_LIB=$P/liblasso_prover.so LASSO_DEVICE_LIB=$P/liblasso_hip.so -- --steps 20 --warmup 3; done
python - <<'PY'
import collections
d=collections.defaultdict(list)
for l in open('gpurun_out/r4a/results.txt'):
    a=l.split(); d[a[0]].append(float(a[1]))
for k,v in d.items():
    v.sort(); print(k,'mean %.3f median %.3f min %.3f'%(sum(v)/len(v), v[len(v)//2], v[0]))
PY
exit 0
