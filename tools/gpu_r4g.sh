#!/bin/bash
# Round 3, visit 4g: BN254 workgroup tree with one addition per lane at the two widest levels: parity, phases, configs[1]
OUT=gpurun_out/r4g; mkdir -p $OUT
LASSO_TEST_CURVE=bn254 timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x > $OUT/pytest_kernels_bn254.log 2>&1; grep -E "passed|failed|error" $OUT/pytest_kernels_bn254.log | tail -2; grep -E "^FAILED|Error" $OUT/pytest_kernels_bn254.log | head -5
timeout 600 python -m pytest tests/test_gpu_bn254.py -m gpu -q -x > $OUT/pytest_bn254.log 2>&1; grep -E "passed|failed|error" $OUT/pytest_bn254.log | tail -2
timeout 60 tools/msm_phase_bench_bn254 4096 2>&1 | grep "k_msm_direct" | tee $OUT/msm_phase_bn254.txt
for i in 1 2 3 4; do timeout 120 python bench.py --curve bn254 --c 4 --log-s 20 --steps 10 --warmup 2 --no-cpu-baseline --concurrent 0 --no-slab-leg --no-prof 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('bn254 and c4 2^20 %.3f' % d['ms_per_step'])"; done | tee $OUT/results.txt
exit 0
