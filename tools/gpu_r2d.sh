#!/bin/bash
# Round 2, visit 4: the byte-table commitment kernel (k_msm_rows8) — parity first, then A/B timing against the bucket kernel, then kernel stats
OUT=gpurun_out/r2d
mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_prover.py tests/test_golden.py -m gpu -x -q -p no:cacheprovider -k "hyrax or bit_exact_vs_oracle or golden or slab_commitment or at_baseline_size" > $OUT/pytest_subset.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_subset.log
BARGS="--steps 5 --warmup 2 --no-cpu-baseline --concurrent 0 --no-slab-leg"
for v in 1 0 1 0; do LASSO_MSM_ROWS8=$v timeout 100 python bench.py $BARGS > $OUT/bench_rows8_$v.json 2> $OUT/bench_rows8_$v.err; python -c "
import json; d=json.loads(open('$OUT/bench_rows8_$v.json').read().strip().splitlines()[-1]); m=[k for k in d['kernels_one_profiled_step'] if k['kernel'].startswith('msm_commit')][0]; print('ROWS8=$v ms_per_step', round(d['ms_per_step'],3), 'commit msm', m['ms'], 'commit_s', d['config']['commit_s'], d['config']['commit_warm_s'], 'msm roof', d['roofline_msm']['commit']['frac'])"; done
LASSO_TRACE=1 timeout 100 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --concurrent 0 --no-slab-leg --no-prof > /dev/null 2> $OUT/trace.txt; grep "Subtables.commit\|SparsePoly.prove" $OUT/trace.txt | tail -2
timeout 100 python bench.py --curve bn254 $BARGS > $OUT/bench_bn254_rows8.json 2>/dev/null; LASSO_MSM_ROWS8=0 timeout 100 python bench.py --curve bn254 $BARGS > $OUT/bench_bn254_rows8_off.json 2>/dev/null; python -c "
import json
for f in ('bench_bn254_rows8','bench_bn254_rows8_off'):
    d=json.loads(open('$OUT/'+f+'.json').read().strip().splitlines()[-1]); print(f, round(d['ms_per_step'],3))"
ls $OUT
exit 0
