#!/bin/bash
# Round 3, visit k: LT round 0 in integer arithmetic — parity (entry point, LT proofs on both curves), LT C=16 timing at 2^22 / 2^24 with and without it
OUT=gpurun_out/r3k; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "lt or combine or linear" > $OUT/pytest_kernels.log 2>&1; echo "kernels rc=$?"; grep -E "passed|failed" $OUT/pytest_kernels.log | tail -1
LASSO_TEST_CURVE=bn254 timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "lt or combine" > $OUT/pytest_kernels_bn254.log 2>&1; echo "kernels bn254 rc=$?"; grep -E "passed|failed" $OUT/pytest_kernels_bn254.log | tail -1
timeout 1500 python -m pytest tests/test_gpu_prover.py tests/test_golden.py tests/test_gpu_bn254.py -x -q -m gpu -k "not at_baseline_size and not full_size and not verifies_at_scale and not slab" > $OUT/pytest_proofs.log 2>&1; echo "proofs rc=$?"; grep -E "passed|failed" $OUT/pytest_proofs.log | tail -1
timeout 600 python -m pytest tests/test_gpu_prover.py -x -q -m gpu -k "full_size and lt" > $OUT/pytest_lt_full.log 2>&1; echo "lt full size rc=$?"; grep -E "passed|failed" $OUT/pytest_lt_full.log | tail -1
for u in 1 0; do for LS in 22 24; do
  LASSO_SUMCHECK_U32=$u python bench.py --kind lt --c 16 --log-s $LS --steps 3 --warmup 1 --no-cpu-baseline --concurrent 0 --no-slab-leg > $OUT/bench_lt_c16_2p${LS}_u32_$u.json 2> $OUT/bench_lt_c16_2p${LS}_u32_$u.err
  python -c "
import json;d=json.load(open('$OUT/bench_lt_c16_2p${LS}_u32_$u.json'));print('lt c16 2^$LS u32=$u ms_per_step %.2f' % d['ms_per_step'], [(k['kernel'][:10],k['launches'],k['ms']) for k in d['kernels_one_profiled_step'] if k['kernel'][:5] in ('sumch','misc')])"
done; done
exit 0
