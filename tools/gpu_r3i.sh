#!/bin/bash
# Round 3, visit i: the primary sumcheck's first round and first bind from the u32 values — parity, then A/B (LASSO_SUMCHECK_U32=0/1) at the metric, configs[2], configs[3]
OUT=gpurun_out/r3i; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu > $OUT/pytest_kernels.log 2>&1; echo "kernels rc=$?"; grep -E "passed|failed" $OUT/pytest_kernels.log | tail -1
LASSO_TEST_CURVE=bn254 timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "linear or lt" > $OUT/pytest_kernels_bn254.log 2>&1; echo "kernels bn254 rc=$?"; grep -E "passed|failed" $OUT/pytest_kernels_bn254.log | tail -1
timeout 1500 python -m pytest tests/test_gpu_prover.py tests/test_golden.py tests/test_gpu_bn254.py -x -q -m gpu -k "not at_baseline_size and not full_size and not verifies_at_scale" > $OUT/pytest_proofs.log 2>&1; echo "proofs rc=$?"; grep -E "passed|failed" $OUT/pytest_proofs.log | tail -1
run() { # name, args..., then env via LASSO_SUMCHECK_U32
  local name=$1; local u=$2; shift; shift
  LASSO_SUMCHECK_U32=$u python bench.py "$@" --no-cpu-baseline --concurrent 0 --no-slab-leg > $OUT/bench_${name}_u32_$u.json 2> $OUT/bench_${name}_u32_$u.err
  python -c "
import json;d=json.load(open('$OUT/bench_${name}_u32_$u.json'));print('$name u32=$u ms_per_step %.3f' % d['ms_per_step'], [(k['kernel'][:12],k['launches'],k['ms']) for k in d['kernels_one_profiled_step'] if k['kernel'][:5] in ('bind_','sumch')])"
}
for u in 1 0 1 0; do run and $u --steps 10 --warmup 2; done
for u in 1 0; do run xor_c8 $u --kind xor --c 8 --steps 3 --warmup 1; done
for u in 1 0; do run range_c4_2p26 $u --kind range --c 4 --log-s 26 --steps 3 --warmup 1; done
exit 0
