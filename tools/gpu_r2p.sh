#!/bin/bash
mkdir -p gpurun_out/r2p
timeout 300 python bench.py --kind range --c 4 --log-s 26 --steps 2 --warmup 1 --no-cpu-baseline --concurrent 0 --no-slab-leg > gpurun_out/r2p/bench_range_c4_2p26.json 2>/dev/null
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r2p/bench_range_c4_2p26.json").read().strip().splitlines()[-1])
print("range c4 2^26 ms", d["ms_per_step"], "densify_s", d["config"]["densify_s"], "commit_s", d["config"]["commit_s"], d["config"]["commit_warm_s"])
for k in d["kernels_one_profiled_step"]: print(k["kernel"], k["launches"], k["ms"], k["alg_GBps"])
print({x: d["roofline"].get(x) for x in ("kernel","achieved","frac","launches")}, {x: d["roofline_bind_top"].get(x) for x in ("achieved","frac","launches")})
PY
timeout 200 python bench.py --kind and --c 4 --log-s 20 --steps 3 --warmup 1 --no-cpu-baseline --concurrent 0 --no-slab-leg > gpurun_out/r2p/bench_and_c4_2p20.json 2>/dev/null
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r2p/bench_and_c4_2p20.json").read().strip().splitlines()[-1])
print("and c4 2^20 (curve25519) ms", d["ms_per_step"])
for k in d["kernels_one_profiled_step"]: print(k["kernel"], k["launches"], k["ms"])
PY
exit 0
