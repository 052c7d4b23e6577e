#!/bin/bash
# Round 3, visit 4i: what do the live HIP-event brackets around the large launches cost in the timed region?
OUT=gpurun_out/r4i; mkdir -p $OUT
for i in 1 2 3 4 5 6; do
  for mode in "" "--no-prof"; do
    timeout 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --concurrent 0 --no-slab-leg $mode 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('brackets=%s %.3f' % ('off' if '$mode' else 'on', d['ms_per_step']))"
  done
done | tee $OUT/results.txt
python - <<'PY'
import collections
d=collections.defaultdict(list)
for l in open('gpurun_out/r4i/results.txt'):
    a=l.split(); d[a[0]].append(float(a[1]))
for k,v in d.items():
    v.sort(); print(k,'mean %.3f median %.3f min %.3f'%(sum(v)/len(v), v[len(v)//2], v[0]))
PY
exit 0
