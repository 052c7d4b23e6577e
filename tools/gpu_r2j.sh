#!/bin/bash
OUT=gpurun_out/r2j
mkdir -p $OUT
export TMPDIR=/tmp
for nx in 0 32 128 256 512; do LASSO_CUBIC_NX=$nx timeout 200 python bench.py --kind xor --c 8 --steps 2 --warmup 1 --no-cpu-baseline --concurrent 0 --no-slab-leg > $OUT/bench_xor_c8_nx$nx.json 2>/dev/null; python -c "
import json
d=json.loads(open('$OUT/bench_xor_c8_nx$nx.json').read().strip().splitlines()[-1]); r=d['roofline']; b=d['roofline_bind_top']
print('NX=$nx ms', round(d['ms_per_step'],2), 'cubic large', r['achieved'], r['avg_launch_us'], 'bind', b['achieved'], b['avg_launch_us'])"; done
for nx in 128 256; do LASSO_CUBIC_NX=$nx timeout 100 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --concurrent 0 --no-slab-leg > $OUT/bench_and_c1_nx$nx.json 2>/dev/null; python -c "
import json
d=json.loads(open('$OUT/bench_and_c1_nx$nx.json').read().strip().splitlines()[-1]); r=d['roofline']
print('AND C=1 NX=$nx ms', round(d['ms_per_step'],2), 'cubic large', r['achieved'], r['avg_launch_us'])"; done
exit 0
