#!/bin/bash
# Round 3, visit s: the resident tail one round ahead of the host (k_cubic_tail_ahead): phases, parity, A/B
OUT=gpurun_out/r3s; mkdir -p $OUT
(timeout 30 tools/tail_phase_bench 512 1; timeout 30 tools/tail_phase_bench 64 1; timeout 30 tools/tail_phase_bench 64 0) 2>&1 | tee $OUT/tail_phase.txt
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x > $OUT/pytest_kernels.log 2>&1; grep -E "passed|failed|error" $OUT/pytest_kernels.log | tail -2
LASSO_TEST_CURVE=bn254 timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "tail or cubic" > $OUT/pytest_kernels_bn254.log 2>&1; grep -E "passed|failed|error" $OUT/pytest_kernels_bn254.log | tail -2
for a in 1 0; do echo "LASSO_TAIL_AHEAD=$a"; LASSO_TAIL_AHEAD=$a timeout 60 tools/tail_bench; done 2>&1 | tee $OUT/tail_bench.txt
run() { local label=$1; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py "$@" --no-cpu-baseline --concurrent 0 --no-slab-leg --no-prof > $OUT/b.json 2> $OUT/b.err
  python -c "
import json;d=json.load(open('$OUT/b.json'));print('$label %.3f' % d['ms_per_step'])" | tee -a $OUT/results.txt; }
for i in 1 2 3; do run "and ahead=1" LASSO_TAIL_AHEAD=1 -- --steps 20 --warmup 3; run "and ahead=0" LASSO_TAIL_AHEAD=0 -- --steps 20 --warmup 3; done
timeout 600 python bench.py --no-slab-leg --concurrent 0 > $OUT/bench_default.json 2> $OUT/bench_default.err; python -c "
import json;d=json.load(open('$OUT/bench_default.json'));print('ms',d['ms_per_step'],'parity',d['parity_checked'].get('equal'),d['parity_checked'].get('commitment_equal'))"
exit 0
