#!/bin/bash
# fused fingerprint + first tree layer: parity, then timing at the headline and at configs[2]
OUT=gpurun_out/r2k
mkdir -p $OUT
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_prover.py tests/test_golden.py -m gpu -x -q -p no:cacheprovider -k "fingerprint or gp_build or bit_exact_vs_oracle or golden or slab or at_baseline_size or full_size" > $OUT/pytest_subset.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $OUT/pytest_subset.log | tail -1
BARGS="--steps 5 --warmup 2 --no-cpu-baseline --concurrent 0 --no-slab-leg"
for i in 1 2; do timeout 100 python bench.py $BARGS > $OUT/bench_$i.json 2>/dev/null; python -c "
import json; d=json.loads(open('$OUT/bench_$i.json').read().strip().splitlines()[-1]); ks={k['kernel']:(k['launches'],k['ms']) for k in d['kernels_one_profiled_step']}; print('ms_per_step', round(d['ms_per_step'],3), 'fingerprint', ks.get('fingerprint'), 'gp_build', ks.get('gp_build'))"; done
timeout 200 python bench.py --kind xor --c 8 --steps 2 --warmup 1 --no-cpu-baseline --concurrent 0 --no-slab-leg > $OUT/bench_xor_c8.json 2>/dev/null; python -c "
import json; d=json.loads(open('$OUT/bench_xor_c8.json').read().strip().splitlines()[-1]); ks={k['kernel']:(k['launches'],k['ms']) for k in d['kernels_one_profiled_step']}; print('xor c8 ms', round(d['ms_per_step'],2), 'fingerprint', ks.get('fingerprint'), 'gp_build', ks.get('gp_build'))"
exit 0
