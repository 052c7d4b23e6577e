#!/bin/bash
# Round 3, visit c: fused bullet round with the extra workgroups dispatched first and <= 256 workgroups per launch — parity subset, then A/B timing
OUT=gpurun_out/r3c; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "bullet or msm" > $OUT/pytest_kernels.log 2>&1; echo "kernels rc=$?"; tail -2 $OUT/pytest_kernels.log
timeout 600 python -m pytest tests/test_golden.py tests/test_gpu_bn254.py -x -q -m gpu -k "not full_size" > $OUT/pytest_proofs.log 2>&1; echo "proofs rc=$?"; tail -2 $OUT/pytest_proofs.log
for F in 0 1 0 1; do
  LASSO_MSM_FUSED=$F python bench.py --steps 10 --warmup 2 --no-cpu-baseline --concurrent 0 --no-slab-leg > $OUT/bench_fused$F.json 2> $OUT/bench_fused$F.err; echo "bench fused=$F rc=$?"
  python -c "
import json;d=json.load(open('$OUT/bench_fused$F.json'));print('fused=$F ms_per_step', d['ms_per_step'])
for k in d['kernels_one_profiled_step']:
    if k['kernel'].startswith('msm_o') or k['kernel']=='misc': print('   ',k['kernel'],k['launches'],k['ms'],k['avg_launch_us'])"
done
for F in 0 1; do
  LASSO_MSM_FUSED=$F python bench.py --curve bn254 --c 4 --log-s 20 --steps 10 --warmup 2 --no-cpu-baseline --concurrent 0 --no-slab-leg > $OUT/bench_bn254_fused$F.json 2> $OUT/bench_bn254_fused$F.err; echo "bench bn254 fused=$F rc=$?"
  python -c "import json;d=json.load(open('$OUT/bench_bn254_fused$F.json'));print('bn254 fused=$F ms_per_step', d['ms_per_step'])"
done
exit 0
