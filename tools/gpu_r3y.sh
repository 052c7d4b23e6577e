#!/bin/bash
# Round 3, visit y: tighter A/B of the direct block-sum hand-off (10 alternating pairs, 20 steps each)
OUT=gpurun_out/r3y; mkdir -p $OUT
run() { local label=$1; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 120 python bench.py "$@" --no-cpu-baseline --concurrent 0 --no-slab-leg --no-prof > $OUT/b.json 2> $OUT/b.err
  python -c "
import json;d=json.load(open('$OUT/b.json'));print('$label %.3f' % d['ms_per_step'])" | tee -a $OUT/results.txt; }
for i in 1 2 3 4 5 6 7 8 9 10; do run "and direct_nx=16" LASSO_DIRECT_NX=16 -- --steps 20 --warmup 3; run "and direct_nx=0" LASSO_DIRECT_NX=0 -- --steps 20 --warmup 3; done
python - <<'PY'
import collections
d=collections.defaultdict(list)
for l in open('gpurun_out/r3y/results.txt'):
    a=l.split(); d[a[1]].append(float(a[2]))
for k,v in d.items():
    v.sort(); print(k,'mean %.3f median %.3f min %.3f'%(sum(v)/len(v), v[len(v)//2], v[0]))
PY
exit 0
