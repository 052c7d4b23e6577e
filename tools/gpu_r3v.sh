#!/bin/bash
# Round 3, visit v: HEAD after the tail experiments were backed out: concurrency test, kernel tests, default bench
OUT=gpurun_out/r3v; mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_prover.py -m gpu -q -x -k "concurrent or bit_exact_vs_oracle" > $OUT/pytest_proofs.log 2>&1; grep -E "passed|failed|error" $OUT/pytest_proofs.log | tail -2
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x > $OUT/pytest_kernels.log 2>&1; grep -E "passed|failed|error" $OUT/pytest_kernels.log | tail -2
timeout 200 python bench.py --no-slab-leg > $OUT/bench_default.json 2> $OUT/bench_default.err; python -c "
import json;d=json.load(open('$OUT/bench_default.json'));print('ms',d['ms_per_step'],'parity',d['parity_checked'].get('equal'),d['parity_checked'].get('commitment_equal'),'concurrent',d['concurrent_proofs'])"
exit 0
