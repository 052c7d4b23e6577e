#!/bin/bash
# Round 3, closing visit: the whole -m gpu suite and smoke() on HEAD, the default bench line as the driver runs it
OUT=gpurun_out/r4h; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $OUT/pytest_gpu.log | tail -1
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4h/bench.json").read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], "value", d["value"]); print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], "parity", d["parity_checked"]["equal"], d["parity_checked"]["commitment_equal"])
print("roofline", {k: d["roofline"].get(k) for k in ("achieved", "frac", "achieved_traffic", "frac_traffic")})
print("msm", d["roofline_msm"]["commit"]["frac"], d["roofline_msm"]["opening"]["frac"]); print("slab", d["slab_mode"].get("ms_per_proof"), d["slab_mode"].get("parity"), "conc", d["concurrent_proofs"].get("value")); print(d["lib_sha"])
PY
timeout 200 python bench.py --curve bn254 --c 4 --log-s 20 --steps 10 --warmup 2 --concurrent 0 --no-slab-leg > $OUT/bench_bn254_config1.json 2> $OUT/bench_bn254.err; python -c "
import json;d=json.loads(open('$OUT/bench_bn254_config1.json').read().strip().splitlines()[-1]);print('bn254 configs[1] ms', d['ms_per_step'], 'parity', d.get('parity_checked'))" | cut -c1-300
exit 0
