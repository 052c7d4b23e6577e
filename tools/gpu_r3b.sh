#!/bin/bash
# Round 3, visit b: the fused bullet-round kernel — parity (entry points + whole proofs), then A/B timing (LASSO_MSM_FUSED=0/1) on the headline and BN254 configs[1]
OUT=gpurun_out/r3b; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "bullet or msm or hyrax" > $OUT/pytest_kernels.log 2>&1; echo "kernels rc=$?"; tail -3 $OUT/pytest_kernels.log
timeout 600 python -m pytest tests/test_gpu_prover.py tests/test_golden.py tests/test_gpu_bn254.py -x -q -m gpu -k "not at_baseline_size and not full_size and not slab" > $OUT/pytest_proofs.log 2>&1; echo "proofs rc=$?"; tail -3 $OUT/pytest_proofs.log
for F in 0 1; do
  LASSO_MSM_FUSED=$F python bench.py --steps 10 --warmup 2 --no-cpu-baseline --concurrent 0 --no-slab-leg > $OUT/bench_fused$F.json 2> $OUT/bench_fused$F.err; echo "bench fused=$F rc=$?"
  python -c "import json;d=json.load(open('$OUT/bench_fused$F.json'));print('fused=$F ms_per_step', d['ms_per_step'])"
  LASSO_MSM_FUSED=$F python bench.py --curve bn254 --c 4 --log-s 20 --steps 10 --warmup 2 --no-cpu-baseline --concurrent 0 --no-slab-leg > $OUT/bench_bn254_fused$F.json 2> $OUT/bench_bn254_fused$F.err; echo "bench bn254 fused=$F rc=$?"
  python -c "import json;d=json.load(open('$OUT/bench_bn254_fused$F.json'));print('bn254 fused=$F ms_per_step', d['ms_per_step'])"
done
LASSO_TRACE=1 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --concurrent 0 --no-slab-leg --no-prof > /dev/null 2> $OUT/trace_spans.txt
exit 0
