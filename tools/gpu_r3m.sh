#!/bin/bash
# Round 3, visit m: HIP runtime knobs for the launch path (where kernel arguments live, how they are copied), alternating runs; host-category trace on HEAD
OUT=gpurun_out/r3m; mkdir -p $OUT
run() { # label, env assignments..., -- bench args
  local label=$1; shift
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py "$@" --no-cpu-baseline --concurrent 0 --no-slab-leg --no-prof > $OUT/b.json 2> $OUT/b.err
  python -c "
import json;d=json.load(open('$OUT/b.json'));print('$label %.3f' % d['ms_per_step'])" | tee -a $OUT/results.txt
}
for i in 1 2 3; do
  run "default" X=1 -- --steps 20 --warmup 3
  run "dev_kernarg=1" HIP_FORCE_DEV_KERNARG=1 -- --steps 20 --warmup 3
  run "dev_kernarg=0" HIP_FORCE_DEV_KERNARG=0 -- --steps 20 --warmup 3
  run "fgs_kernarg=0" ROC_USE_FGS_KERNARG=0 -- --steps 20 --warmup 3
  run "fgs_kernarg=1" ROC_USE_FGS_KERNARG=1 -- --steps 20 --warmup 3
  run "kernarg_copy_opt=0" DEBUG_HIP_KERNARG_COPY_OPT=0 -- --steps 20 --warmup 3
  run "kernarg_copy_opt=1" DEBUG_HIP_KERNARG_COPY_OPT=1 -- --steps 20 --warmup 3
  run "hdp_flush_wa=0" DEBUG_CLR_KERNARG_HDP_FLUSH_WA=0 -- --steps 20 --warmup 3
done
LASSO_TRACE=2 timeout 100 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --concurrent 0 --no-slab-leg --no-prof > /dev/null 2> $OUT/trace2.txt
grep "\[host\]" $OUT/trace2.txt | tail -12; grep "\[trace\]" $OUT/trace2.txt | tail -20
exit 0
