#!/bin/bash
# Round 3, visit x: small launches publish every workgroup's block sums (host adds them): parity + A/B over the threshold
OUT=gpurun_out/r3x; mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x > $OUT/pytest_kernels.log 2>&1; grep -E "passed|failed|error" $OUT/pytest_kernels.log | tail -2; grep -E "^FAILED|Error" $OUT/pytest_kernels.log | head -5
LASSO_TEST_CURVE=bn254 timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "cubic" > $OUT/pytest_kernels_bn254.log 2>&1; grep -E "passed|failed|error" $OUT/pytest_kernels_bn254.log | tail -2
timeout 300 python -m pytest tests/test_gpu_prover.py -m gpu -q -x -k "concurrent or bit_exact_vs_oracle" > $OUT/pytest_proofs.log 2>&1; grep -E "passed|failed|error" $OUT/pytest_proofs.log | tail -2
for d in 16 0; do echo "LASSO_DIRECT_NX=$d"; LASSO_DIRECT_NX=$d timeout 40 tools/tail_bench 8192; done 2>&1 | grep "one launch" | tee $OUT/tail_bench.txt
run() { local label=$1; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 120 python bench.py "$@" --no-cpu-baseline --concurrent 0 --no-slab-leg --no-prof > $OUT/b.json 2> $OUT/b.err
  python -c "
import json;d=json.load(open('$OUT/b.json'));print('$label %.3f' % d['ms_per_step'])" | tee -a $OUT/results.txt; }
for i in 1 2 3; do for w in 16 0 64 4; do run "and direct_nx=$w" LASSO_DIRECT_NX=$w -- --steps 20 --warmup 3; done; done
for w in 16 0; do run "xor_c8 direct_nx=$w" LASSO_DIRECT_NX=$w -- --kind xor --c 8 --steps 5 --warmup 1; done
exit 0
