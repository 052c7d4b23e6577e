#!/bin/bash
# Round 3, visit n: kernel-argument preload into SGPRs (-mllvm -amdgpu-kernarg-preload-count=16) A/B against the default build, alternating runs
OUT=gpurun_out/r3n; mkdir -p $OUT
run() { # label, env assignments..., -- bench args
  local label=$1; shift
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py "$@" --no-cpu-baseline --concurrent 0 --no-slab-leg --no-prof > $OUT/b.json 2> $OUT/b.err
  python -c "
import json;d=json.load(open('$OUT/b.json'));print('$label %.3f parity=%s' % (d['ms_per_step'], d.get('parity_checked')))" | tee -a $OUT/results.txt
}
P=$PWD/ab/preload
for i in 1 2 3 4; do
  run "default" X=1 -- --steps 20 --warmup 3
  run "preload16" LASSO_PROVER_LIB=$P/liblasso_prover.so LASSO_DEVICE_LIB=$P/liblasso_hip.so -- --steps 20 --warmup 3
done
tail -3 $OUT/b.err
exit 0
