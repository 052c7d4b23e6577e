#!/bin/bash
mkdir -p gpurun_out/r2m
timeout 300 python bench.py --kind lt --c 16 --log-s 22 --steps 2 --warmup 1 --no-cpu-baseline --concurrent 0 --no-slab-leg > gpurun_out/r2m/bench_lt_c16_2p22.json 2>/dev/null
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r2m/bench_lt_c16_2p22.json").read().strip().splitlines()[-1])
print("lt c16 2^22 ms", d["ms_per_step"])
for k in d["kernels_one_profiled_step"]: print(k["kernel"], k["launches"], k["ms"], k["alg_GBps"])
PY
LASSO_TRACE=1 timeout 100 python bench.py --kind lt --c 16 --log-s 22 --steps 1 --warmup 1 --no-cpu-baseline --concurrent 0 --no-slab-leg --no-prof > /dev/null 2> gpurun_out/r2m/trace.txt; grep trace gpurun_out/r2m/trace.txt | tail -21 | cut -c1-120
exit 0
