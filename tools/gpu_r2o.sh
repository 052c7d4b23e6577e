#!/bin/bash
# A/B: register budget of the fused cubic round (3 waves/SIMD default, 4 with a 100-byte spill, 2 with room to spare)
OUT=gpurun_out/r2o
mkdir -p $OUT
BARGS="--steps 5 --warmup 2 --no-cpu-baseline --concurrent 0 --no-slab-leg"
cp lasso_amd/liblasso_hip.so /tmp/orig.so
run() { timeout 100 python bench.py $BARGS > $OUT/bench_$1.json 2>/dev/null; python -c "
import json; d=json.loads(open('$OUT/bench_$1.json').read().strip().splitlines()[-1]); r=d['roofline']; print('$1 ms', round(d['ms_per_step'],3), 'cubic large GB/s', r['achieved'], 'us', r['avg_launch_us'])"; }
run default_a
cp lasso_amd/alt_occ4_liblasso_hip.bin lasso_amd/liblasso_hip.so; run occ4_a
cp lasso_amd/alt_occ2_liblasso_hip.bin lasso_amd/liblasso_hip.so; run occ2_a
cp /tmp/orig.so lasso_amd/liblasso_hip.so; run default_b
cp lasso_amd/alt_occ4_liblasso_hip.bin lasso_amd/liblasso_hip.so; run occ4_b
cp /tmp/orig.so lasso_amd/liblasso_hip.so
exit 0
