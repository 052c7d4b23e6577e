#!/bin/bash
# Round 3, visit t: the tails' mailbox in device memory (host writes through the BAR): parity + A/B
OUT=gpurun_out/r3t; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x > $OUT/pytest_kernels.log 2>&1; grep -E "passed|failed|error" $OUT/pytest_kernels.log | tail -2
LASSO_TEST_CURVE=bn254 timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "tail or cubic or abort" > $OUT/pytest_kernels_bn254.log 2>&1; grep -E "passed|failed|error" $OUT/pytest_kernels_bn254.log | tail -2
for a in 1 0; do echo "LASSO_DEVICE_MAILBOX=$a"; LASSO_DEVICE_MAILBOX=$a timeout 60 tools/tail_bench; done 2>&1 | tee $OUT/tail_bench.txt
run() { local label=$1; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py "$@" --no-cpu-baseline --concurrent 0 --no-slab-leg --no-prof > $OUT/b.json 2> $OUT/b.err
  python -c "
import json;d=json.load(open('$OUT/b.json'));print('$label %.3f' % d['ms_per_step'])" | tee -a $OUT/results.txt; }
for i in 1 2 3 4; do run "and devmail=1" LASSO_DEVICE_MAILBOX=1 -- --steps 20 --warmup 3; run "and devmail=0" LASSO_DEVICE_MAILBOX=0 -- --steps 20 --warmup 3; done
timeout 600 python bench.py --no-slab-leg > $OUT/bench_default.json 2> $OUT/bench_default.err; python -c "
import json;d=json.load(open('$OUT/bench_default.json'));print('ms',d['ms_per_step'],'parity',d['parity_checked'].get('equal'),d['parity_checked'].get('commitment_equal'),'concurrent',d['concurrent_proofs'].get('value'))"
exit 0
