#!/bin/bash
# Round 3, visit u: which switch breaks the concurrent-proofs leg?
OUT=gpurun_out/r3u; mkdir -p $OUT
try() { local label=$1; shift
  env "$@" timeout 90 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-slab-leg --no-prof > $OUT/b.json 2> $OUT/b_$label.err; echo "$label rc=$?"
  python -c "
import json;d=json.load(open('$OUT/b.json'));print('$label', d['ms_per_step'], d.get('concurrent_proofs'))" 2>&1 | tail -1 | cut -c1-300 | tee -a $OUT/results.txt; }
try default X=1
try devmail0 LASSO_DEVICE_MAILBOX=0
try devmail0_tagged0 LASSO_DEVICE_MAILBOX=0 LASSO_TAGGED_RESULTS=0
try default_again X=1
exit 0
