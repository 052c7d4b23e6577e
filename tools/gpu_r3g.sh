#!/bin/bash
# Round 3, visit g: eq tables built inside the first round's launch — parity (entry points, whole proofs, slab, both curves), then A/B timing
OUT=gpurun_out/r3g; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu > $OUT/pytest_kernels.log 2>&1; echo "kernels rc=$?"; grep -E "passed|failed" $OUT/pytest_kernels.log | tail -1
LASSO_TEST_CURVE=bn254 timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "tail or cubic or inline" > $OUT/pytest_kernels_bn254.log 2>&1; echo "kernels bn254 rc=$?"; grep -E "passed|failed" $OUT/pytest_kernels_bn254.log | tail -1
timeout 1500 python -m pytest tests/test_gpu_prover.py tests/test_golden.py tests/test_gpu_bn254.py -x -q -m gpu -k "not at_baseline_size and not full_size and not verifies_at_scale" > $OUT/pytest_proofs.log 2>&1; echo "proofs rc=$?"; grep -E "passed|failed" $OUT/pytest_proofs.log | tail -1
run() { # name, env...
  local name=$1; shift
  env "$@" python bench.py --steps 10 --warmup 2 --no-cpu-baseline --concurrent 0 --no-slab-leg > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  python -c "
import json;d=json.load(open('$OUT/bench_$name.json'));print('$name ms_per_step %.3f' % d['ms_per_step'], [(k['kernel'][:12],k['launches'],k['ms']) for k in d['kernels_one_profiled_step'] if k['kernel'][:5] in ('sumch','eq_ev')])"
}
run head LASSO_X=1
run noinline LASSO_EQ_INLINE=0
run head2 LASSO_X=1
run noinline2 LASSO_EQ_INLINE=0
python bench.py --kind xor --c 8 --log-s 24 --steps 3 --warmup 1 --no-cpu-baseline --concurrent 0 --no-slab-leg > $OUT/bench_xor_c8.json 2> $OUT/bench_xor_c8.err
python -c "
import json;d=json.load(open('$OUT/bench_xor_c8.json'));print('xor c8 ms_per_step %.3f' % d['ms_per_step'])"
python bench.py --curve bn254 --c 4 --log-s 20 --steps 10 --warmup 2 --no-cpu-baseline --concurrent 0 --no-slab-leg > $OUT/bench_bn254_config1.json 2> $OUT/bench_bn254_config1.err
python -c "
import json;d=json.load(open('$OUT/bench_bn254_config1.json'));print('bn254 config1 ms_per_step %.3f' % d['ms_per_step'])"
exit 0
