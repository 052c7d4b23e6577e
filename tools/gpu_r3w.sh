#!/bin/bash
# Round 3, visit w: wide start of the resident cubic tail (G workgroups per circuit): kernel parity, proofs, A/B
OUT=gpurun_out/r3w; mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "tail or abort" > $OUT/pytest_tail.log 2>&1; grep -E "passed|failed|error" $OUT/pytest_tail.log | tail -2; grep -E "^FAILED|Error" $OUT/pytest_tail.log | head -5
LASSO_TEST_CURVE=bn254 timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "tail or abort" > $OUT/pytest_tail_bn254.log 2>&1; grep -E "passed|failed|error" $OUT/pytest_tail_bn254.log | tail -2
timeout 300 python -m pytest tests/test_gpu_prover.py -m gpu -q -x -k "concurrent or bit_exact_vs_oracle" > $OUT/pytest_proofs.log 2>&1; grep -E "passed|failed|error" $OUT/pytest_proofs.log | tail -2
run() { local label=$1; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 120 python bench.py "$@" --no-cpu-baseline --concurrent 0 --no-slab-leg --no-prof > $OUT/b.json 2> $OUT/b.err
  python -c "
import json;d=json.load(open('$OUT/b.json'));print('$label %.3f' % d['ms_per_step'])" | tee -a $OUT/results.txt; }
for i in 1 2 3; do for w in 8 1 4 2; do run "and wide=$w" LASSO_TAIL_WIDE=$w -- --steps 20 --warmup 3; done; done
for w in 8 1; do run "xor_c8 wide=$w" LASSO_TAIL_WIDE=$w -- --kind xor --c 8 --steps 5 --warmup 1; done
for w in 8 1; do run "bn254_c4_2p20 wide=$w" LASSO_TAIL_WIDE=$w -- --curve bn254 --c 4 --log-s 20 --steps 10 --warmup 2; done
timeout 200 python bench.py --no-slab-leg > $OUT/bench_default.json 2> $OUT/bench_default.err; python -c "
import json;d=json.load(open('$OUT/bench_default.json'));print('ms',d['ms_per_step'],'parity',d['parity_checked'].get('equal'),d['parity_checked'].get('commitment_equal'),'concurrent',d['concurrent_proofs'].get('value'),d['concurrent_proofs'].get('error'))"
exit 0
