#!/bin/bash
# Round 3, visit p: full parity on the tagged-result build (both curves), default bench line
OUT=gpurun_out/r3p; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q > $OUT/pytest_kernels.log 2>&1; grep -E "passed|failed|error" $OUT/pytest_kernels.log | tail -2
LASSO_TEST_CURVE=bn254 timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q > $OUT/pytest_kernels_bn254.log 2>&1; grep -E "passed|failed|error" $OUT/pytest_kernels_bn254.log | tail -2
timeout 1500 python -m pytest tests -m gpu -q --deselect tests/test_gpu_kernels.py > $OUT/pytest_rest.log 2>&1; grep -E "passed|failed|error" $OUT/pytest_rest.log | tail -2
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; python -c "
import json;d=json.load(open('$OUT/bench_default.json'));print('ms',d['ms_per_step'],'parity',d['parity_checked'].get('equal'),d['parity_checked'].get('commitment_equal'),'slab',d['slab_mode'].get('parity'),d['slab_mode'].get('ms_per_proof'),'roof',d['roofline']['frac'])"
exit 0
