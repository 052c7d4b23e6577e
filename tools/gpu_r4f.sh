#!/bin/bash
# Round 3, visit 4f: software-pipelined loads in the fused cubic round (252 registers, two waves per SIMD) against the plain loop (156, three waves)
OUT=gpurun_out/r4f; mkdir -p $OUT
LASSO_FUSED_PIPELINE=1 timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "cubic" > $OUT/pytest_kernels.log 2>&1; grep -E "passed|failed|error" $OUT/pytest_kernels.log | tail -2
run() { local label=$1; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 200 python bench.py "$@" --no-cpu-baseline --concurrent 0 --no-slab-leg > $OUT/b.json 2> $OUT/b.err
  python -c "
import json;d=json.load(open('$OUT/b.json'));k=[x for x in d['kernels_one_profiled_step'] if x['kernel'].startswith('sumcheck_cubic')][0];print('$label %.3f ms; cubic family %.3f ms; large launches %.1f us avg, roofline frac %.3f' % (d['ms_per_step'], k['ms'], d['roofline'].get('avg_launch_us', 0), d['roofline']['frac']))" | tee -a $OUT/results.txt; }
for i in 1 2 3; do run "and pipe=1" LASSO_FUSED_PIPELINE=1 -- --steps 20 --warmup 3; run "and pipe=0" LASSO_FUSED_PIPELINE=0 -- --steps 20 --warmup 3; done
for i in 1 2; do run "xor_c8 pipe=1" LASSO_FUSED_PIPELINE=1 -- --kind xor --c 8 --steps 5 --warmup 1; run "xor_c8 pipe=0" LASSO_FUSED_PIPELINE=0 -- --kind xor --c 8 --steps 5 --warmup 1; done
exit 0
