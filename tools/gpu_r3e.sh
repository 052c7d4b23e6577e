#!/bin/bash
# Round 3, visit e: point-split LT round + prescale, sharded openings (slab kernels on the device), counters spread over 64 slots; spans of configs[3] for the Amdahl table
OUT=gpurun_out/r3e; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "combine or claim or lt or msm or hyrax or bullet" > $OUT/pytest_kernels.log 2>&1; echo "kernels rc=$?"; tail -2 $OUT/pytest_kernels.log
LASSO_TEST_CURVE=bn254 timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "combine or claim or lt or bullet or msm" > $OUT/pytest_kernels_bn254.log 2>&1; echo "kernels bn254 rc=$?"; tail -2 $OUT/pytest_kernels_bn254.log
timeout 1200 python -m pytest tests/test_gpu_prover.py tests/test_golden.py tests/test_gpu_bn254.py -x -q -m gpu -k "not at_baseline_size and not full_size and not verifies_at_scale" > $OUT/pytest_proofs.log 2>&1; echo "proofs rc=$?"; tail -2 $OUT/pytest_proofs.log
python bench.py --kind lt --c 16 --log-s 24 --steps 3 --warmup 1 --no-cpu-baseline --concurrent 0 --no-slab-leg > $OUT/bench_lt_c16_2p24.json 2> $OUT/bench_lt_c16_2p24.err; echo "bench lt 2^24 rc=$?"
python -c "
import json;d=json.load(open('$OUT/bench_lt_c16_2p24.json'));print('lt c16 2^24 ms_per_step', d['ms_per_step'])
for k in d['kernels_one_profiled_step']: print('   ',k['kernel'],k['launches'],k['ms'],k['avg_launch_us'], k.get('alg_GBps'))"
for F in 0 1 0 1; do
  LASSO_MSM_FUSED=$F python bench.py --steps 10 --warmup 2 --no-cpu-baseline --concurrent 0 --no-slab-leg > $OUT/bench_fused$F.json 2> $OUT/bench_fused$F.err
  python -c "
import json;d=json.load(open('$OUT/bench_fused$F.json'));print('fused=$F ms_per_step', d['ms_per_step'], 'opening', [ (k['launches'],k['ms'],k['avg_launch_us']) for k in d['kernels_one_profiled_step'] if k['kernel'].startswith('msm_o')], 'frac', d['roofline_msm']['opening']['frac'], d['roofline_msm']['commit']['frac'])"
done
LASSO_TRACE=1 python bench.py --kind range --c 4 --log-s 26 --steps 1 --warmup 1 --no-cpu-baseline --concurrent 0 --no-slab-leg --no-prof > $OUT/bench_range_c4_2p26.json 2> $OUT/trace_spans_range_c4_2p26.txt; echo "range trace rc=$?"
python -c "import json;d=json.load(open('$OUT/bench_range_c4_2p26.json'));print('range c4 2^26 ms_per_step', d['ms_per_step'])"
LASSO_TRACE=1 python bench.py --kind xor --c 8 --log-s 24 --steps 1 --warmup 1 --no-cpu-baseline --concurrent 0 --no-slab-leg --no-prof > $OUT/bench_xor_c8_2p24.json 2> $OUT/trace_spans_xor_c8_2p24.txt; echo "xor trace rc=$?"
python -c "import json;d=json.load(open('$OUT/bench_xor_c8_2p24.json'));print('xor c8 2^24 ms_per_step', d['ms_per_step'])"
exit 0
