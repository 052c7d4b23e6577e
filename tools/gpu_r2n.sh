#!/bin/bash
mkdir -p gpurun_out/r2n
OUT=gpurun_out/r2n
timeout 500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_prover.py tests/test_gpu_bn254.py tests/test_golden.py -m gpu -x -q -p no:cacheprovider -k "combine or lt or golden or bit_exact or full_size" > $OUT/pytest_subset.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $OUT/pytest_subset.log | tail -1
for ls in 20 22 24; do timeout 300 python bench.py --kind lt --c 16 --log-s $ls --steps 2 --warmup 1 --no-cpu-baseline --concurrent 0 --no-slab-leg > $OUT/bench_lt_c16_2p$ls.json 2>/dev/null; python -c "
import json; d=json.loads(open('$OUT/bench_lt_c16_2p$ls.json').read().strip().splitlines()[-1]); ks={k['kernel']:(k['launches'],k['ms'],k['alg_GBps']) for k in d['kernels_one_profiled_step']}; print('lt c16 2^$ls ms', round(d['ms_per_step'],2), 'combine', ks.get('sumcheck_combine'))"; done
timeout 100 python bench.py --kind lt --c 4 --log-s 22 --steps 2 --warmup 1 --no-cpu-baseline --concurrent 0 --no-slab-leg --no-prof > $OUT/bench_lt_c4.json 2>/dev/null; python -c "
import json; d=json.loads(open('$OUT/bench_lt_c4.json').read().strip().splitlines()[-1]); print('lt c4 2^22 ms', round(d['ms_per_step'],2))"
exit 0
