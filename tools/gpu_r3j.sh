#!/bin/bash
# Round 3, visit j: tighter A/B of the u32 sumcheck rounds (alternating, 20 steps), plus two cheap sweeps on HEAD (opening-MSM workgroups, cubic grid width)
OUT=gpurun_out/r3j; mkdir -p $OUT
run() { # label, env assignments..., -- bench args
  local label=$1; shift
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py "$@" --no-cpu-baseline --concurrent 0 --no-slab-leg --no-prof > $OUT/b.json 2> $OUT/b.err
  python -c "
import json;d=json.load(open('$OUT/b.json'));print('$label %.3f' % d['ms_per_step'])" | tee -a $OUT/results.txt
}
for i in 1 2 3 4 5 6; do run "and u32=1" LASSO_SUMCHECK_U32=1 -- --steps 20 --warmup 3; run "and u32=0" LASSO_SUMCHECK_U32=0 -- --steps 20 --warmup 3; done
for i in 1 2 3; do run "xor_c8 u32=1" LASSO_SUMCHECK_U32=1 -- --kind xor --c 8 --steps 5 --warmup 1; run "xor_c8 u32=0" LASSO_SUMCHECK_U32=0 -- --kind xor --c 8 --steps 5 --warmup 1; done
for i in 1 2; do run "range_c4_2p26 u32=1" LASSO_SUMCHECK_U32=1 -- --kind range --c 4 --log-s 26 --steps 3 --warmup 1; run "range_c4_2p26 u32=0" LASSO_SUMCHECK_U32=0 -- --kind range --c 4 --log-s 26 --steps 3 --warmup 1; done
for w in 192 224 256 320 384; do run "and msm_wgs=$w" LASSO_MSM_DIRECT_WGS=$w LASSO_SUMCHECK_U32=0 -- --steps 10 --warmup 2; done
for nx in 256 384 512 768 1024; do run "and cubic_nx=$nx" LASSO_CUBIC_NX=$nx LASSO_SUMCHECK_U32=0 -- --steps 10 --warmup 2; done
exit 0
