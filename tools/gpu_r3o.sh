#!/bin/bash
# Round 3, visit o: self-validating result chunks (no ticket, no flag) for the sumcheck hand-offs: parity, per-hand-off latency, A/B at the headline
OUT=gpurun_out/r3o; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q 2>&1 | tail -3 | tee $OUT/pytest_kernels.log
timeout 900 python -m pytest tests/test_gpu_prover.py -m gpu -x -q -k "not full_size and not AT_SIZE" 2>&1 | tail -3 | tee $OUT/pytest_proofs.log
for t in 1 0; do echo "LASSO_TAGGED_RESULTS=$t"; LASSO_TAGGED_RESULTS=$t timeout 60 tools/tail_bench; done 2>&1 | tee $OUT/tail_bench.txt
run() { # label, env assignments..., -- bench args
  local label=$1; shift
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py "$@" --no-cpu-baseline --concurrent 0 --no-slab-leg --no-prof > $OUT/b.json 2> $OUT/b.err
  python -c "
import json;d=json.load(open('$OUT/b.json'));print('$label %.3f' % d['ms_per_step'])" | tee -a $OUT/results.txt
}
for i in 1 2 3 4; do
  run "and tagged=1" LASSO_TAGGED_RESULTS=1 -- --steps 20 --warmup 3
  run "and tagged=0" LASSO_TAGGED_RESULTS=0 -- --steps 20 --warmup 3
done
for i in 1 2; do
  run "xor_c8 tagged=1" LASSO_TAGGED_RESULTS=1 -- --kind xor --c 8 --steps 5 --warmup 1
  run "xor_c8 tagged=0" LASSO_TAGGED_RESULTS=0 -- --kind xor --c 8 --steps 5 --warmup 1
done
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --concurrent 0 --no-slab-leg 2> $OUT/bench_parity.err | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('parity_checked',d.get('parity_checked'),'ms',d['ms_per_step'])"
exit 0
