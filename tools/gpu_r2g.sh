#!/bin/bash
# Round 2, closing visit on HEAD: the whole -m gpu suite, smoke(), the default bench line (as the driver runs it), rocprofv3 kernel stats, sizes sweep
OUT=gpurun_out/r2g
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $OUT/pytest_gpu.log | tail -1
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
timeout 500 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r2g/bench.json").read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], "value", d["value"]); print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], "parity", d["parity_checked"]["equal"], d["parity_checked"]["commitment_equal"])
print("roofline", {k: d["roofline"].get(k) for k in ("achieved", "frac", "achieved_traffic", "frac_traffic")}, "bind", {k: d["roofline_bind_top"].get(k) for k in ("achieved", "frac", "frac_traffic")})
print("msm", d["roofline_msm"]["commit"]["frac"], d["roofline_msm"]["commit"]["executed"], d["roofline_msm"]["opening"]["frac"]); print("slab", d["slab_mode"]["ms_per_proof"], "conc", d["concurrent_proofs"]["value"])
PY
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_g -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --concurrent 0 --no-slab-leg > $R/$OUT/bench_under_rocprof.json 2> $R/$OUT/rocprof.err); f=$(find /tmp/prof_g -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $OUT/bench_2p24_kernel_stats.csv; head -6 $OUT/bench_2p24_kernel_stats.csv | cut -c1-150
for ls in 20 22 26 28; do timeout 200 python bench.py --log-s $ls --steps 3 --warmup 1 --no-cpu-baseline --concurrent 0 --no-slab-leg --no-prof > $OUT/bench_2p$ls.json 2>/dev/null; python -c "
import json; d=json.loads(open('$OUT/bench_2p$ls.json').read().strip().splitlines()[-1]); print('2^$ls', round(d['ms_per_step'],2), 'ms', '%.3e' % d['value'])"; done
timeout 200 python bench.py --kind xor --c 8 --steps 2 --warmup 1 --no-cpu-baseline --concurrent 0 --no-slab-leg --no-prof > $OUT/bench_xor_c8_2p24.json 2>/dev/null; python -c "
import json; d=json.loads(open('$OUT/bench_xor_c8_2p24.json').read().strip().splitlines()[-1]); print('xor c8 2^24', round(d['ms_per_step'],2), 'ms')"
timeout 100 python bench.py --curve bn254 --kind and --c 4 --log-s 20 --steps 3 --warmup 1 --no-cpu-baseline --concurrent 0 --no-slab-leg --no-prof > $OUT/bench_bn254_config1.json 2>/dev/null; python -c "
import json; d=json.loads(open('$OUT/bench_bn254_config1.json').read().strip().splitlines()[-1]); print('bn254 config1', round(d['ms_per_step'],2), 'ms')"
exit 0
