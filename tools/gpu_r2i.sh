#!/bin/bash
OUT=gpurun_out/r2i
mkdir -p $OUT
export TMPDIR=/tmp
timeout 200 python bench.py --kind xor --c 8 --steps 2 --warmup 1 --no-cpu-baseline --concurrent 0 --no-slab-leg > $OUT/bench_xor_c8.json 2>/dev/null
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r2i/bench_xor_c8.json").read().strip().splitlines()[-1])
print("xor c8 2^24 ms", d["ms_per_step"]); 
for k in d["kernels_one_profiled_step"]: print(k["kernel"], k["launches"], k["ms"], k["alg_GBps"])
print("roofline", {x: d["roofline"].get(x) for x in ("kernel","achieved","frac","launches","avg_launch_us")}); print("bind", {x: d["roofline_bind_top"].get(x) for x in ("achieved","frac","launches")})
print(d["large_launches_timed"])
PY
LASSO_TRACE=1 timeout 100 python bench.py --kind xor --c 8 --steps 1 --warmup 1 --no-cpu-baseline --concurrent 0 --no-slab-leg --no-prof > /dev/null 2> $OUT/trace_xor.txt; grep trace $OUT/trace_xor.txt | tail -21
exit 0
