#!/bin/bash
# visit 5: factored k_eq_small — parity (eq tables at every ell, whole proofs, slab variants), then timing
OUT=gpurun_out/r2e
mkdir -p $OUT
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_prover.py tests/test_golden.py -m gpu -x -q -p no:cacheprovider -k "eq_evals or bit_exact_vs_oracle or golden or slab or at_baseline_size or full_size or cubic_batched" > $OUT/pytest_subset.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_subset.log | cut -c1-200
BARGS="--steps 5 --warmup 2 --no-cpu-baseline --concurrent 0 --no-slab-leg"
for i in 1 2; do timeout 100 python bench.py $BARGS > $OUT/bench_$i.json 2> $OUT/bench_$i.err; python -c "
import json; d=json.loads(open('$OUT/bench_$i.json').read().strip().splitlines()[-1]); e=[k for k in d['kernels_one_profiled_step'] if k['kernel']=='eq_evals'][0]; print('ms_per_step', round(d['ms_per_step'],3), 'eq_evals', e)"; done
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_e -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --concurrent 0 --no-slab-leg > /dev/null 2> $GRAFT_REPO_ROOT/$OUT/rocprof.err); f=$(find /tmp/prof_e -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $OUT/kernel_stats.csv; grep -E "k_eq_small|k_eq_outer|k_msm_rows8|k_msm_buckets" $OUT/kernel_stats.csv | cut -c1-160
exit 0
