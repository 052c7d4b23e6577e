#!/bin/bash
# Round 3, validation visit on HEAD: the whole -m gpu suite, smoke(), the default bench line as the driver runs it, rocprofv3 kernel stats, PMC passes for the metric, configs[2] and configs[3]
OUT=gpurun_out/r3z; mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $OUT/pytest_gpu.log | tail -1
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3z/bench.json").read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], "value", d["value"]); print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], "parity", d["parity_checked"]["equal"], d["parity_checked"]["commitment_equal"])
print("roofline", {k: d["roofline"].get(k) for k in ("achieved", "frac", "achieved_traffic", "frac_traffic")}, "bind", {k: d["roofline_bind_top"].get(k) for k in ("achieved", "frac", "frac_traffic")})
print("msm", d["roofline_msm"]["commit"]["frac"], d["roofline_msm"]["opening"]["frac"]); print("slab", d["slab_mode"].get("ms_per_proof"), d["slab_mode"].get("parity"), "conc", d["concurrent_proofs"]["value"]); print(d["lib_sha"])
PY
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_h -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --concurrent 0 --no-slab-leg > $R/$OUT/bench_under_rocprof.json 2> $R/$OUT/rocprof.err); f=$(find /tmp/prof_h -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $OUT/bench_2p24_kernel_stats.csv; head -8 $OUT/bench_2p24_kernel_stats.csv | cut -c1-150
pmc() { # key, bench args...
  local key=$1; shift
  mkdir -p $OUT/pmc/$key
  for CTR in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout 500 rocprofv3 --pmc $CTR --kernel-trace --output-format csv -d /tmp/pmc_${key}_$CTR -o bench -- python $R/bench.py "$@" --steps 1 --warmup 1 --no-cpu-baseline --concurrent 0 --no-slab-leg > $R/$OUT/pmc/$key/bench_under_pmc_$CTR.json 2> $R/$OUT/pmc/$key/rocprof_$CTR.err); echo "pmc $key $CTR rc=$?"
    f=$(find /tmp/pmc_${key}_$CTR -name '*counter_collection.csv' | head -1); [ -n "$f" ] && cp $f $OUT/pmc/$key/bench_${CTR}_counter_collection.csv
  done
  python tools/pmc_summary.py $OUT/pmc/$key/bench_FETCH_SIZE_counter_collection.csv $OUT/pmc/$key/bench_WRITE_SIZE_counter_collection.csv $OUT/pmc/$key/bench_under_pmc_FETCH_SIZE.json $OUT/pmc/$key/bench_traffic.json "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command on MI355X, round 3 (profiles/r03_pmc/$key/, tools/gpu_r3z.sh)" > $OUT/pmc/$key/pmc_summary.log 2>&1
  python -c "
import json; d=json.load(open('$OUT/pmc/$key/bench_traffic.json'))
print('$key', d.get('_workload'))
for k,v in d.items():
    if k != '_workload': print('  ', k, v['bytes_per_launch'], v['alg_bytes_per_launch'], v['traffic_over_algorithmic'], v['launches'])"
  # the per-dispatch CSVs are large: keep only the last proof's rows of the streaming kernels
  for CTR in FETCH_SIZE WRITE_SIZE; do python - $OUT/pmc/$key/bench_${CTR}_counter_collection.csv <<'PY'
import csv, sys
p = sys.argv[1]
rows = list(csv.DictReader(open(p)))
if rows:
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    idx = [i for i, r in enumerate(rows) if "k_gather_u32" in r["Kernel_Name"]]
    rows = rows[idx[-1]:] if idx else rows
    keep = ["Dispatch_Id", "Kernel_Name", "Grid_Size", "Workgroup_Size", "Counter_Name", "Counter_Value"]
    keep = [k for k in keep if k in rows[0]]
    with open(p, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=keep); w.writeheader()
        for r in rows: w.writerow({k: (r[k][:70] if k == "Kernel_Name" else r[k]) for k in keep})
PY
  done
}
pmc and_c1_m16_2p24_curve25519
pmc xor_c8_m16_2p24_curve25519 --kind xor --c 8
pmc range_c4_m16_2p26_curve25519 --kind range --c 4 --log-s 26
timeout 300 python bench.py --kind lt --c 16 --log-s 24 --steps 3 --warmup 1 --no-cpu-baseline --concurrent 0 --no-slab-leg > $OUT/bench_lt_c16_2p24.json 2> $OUT/bench_lt_c16_2p24.err; python -c "
import json; d=json.loads(open('$OUT/bench_lt_c16_2p24.json').read().strip().splitlines()[-1]); print('lt c16 2^24', round(d['ms_per_step'],2), 'ms', [(k['kernel'][:10], k['ms']) for k in d['kernels_one_profiled_step']])"
# the N > 1 path end to end on this 1-GPU box: bench.py starts its own two ranks (gloo: they share the device), one proof per rank + ONE proof over both (slab leg, sharded openings; RCCL declines consistently)
timeout 600 python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --concurrent 0 --slab-steps 1 > $OUT/bench_2ranks_one_gpu_self_launched.json 2> $OUT/bench_2ranks_one_gpu_self_launched.err; echo "2 ranks rc=$?"
python -c "
import json; d=json.loads(open('$OUT/bench_2ranks_one_gpu_self_launched.json').read().strip().splitlines()[-1]); print('2 ranks on one GPU: n_gpus', d['n_gpus'], 'ms_per_step', round(d['ms_per_step'],2), 'distinct', d['config']['distinct_proofs'], 'slab', {k: d['slab_mode'].get(k) for k in ('n_gpus','ms_per_proof','parity','rccl_ranks','error')})"
for ls in 20 22 26 28; do timeout 200 python bench.py --log-s $ls --steps 3 --warmup 1 --no-cpu-baseline --concurrent 0 --no-slab-leg --no-prof > $OUT/bench_2p$ls.json 2>/dev/null; python -c "
import json; d=json.loads(open('$OUT/bench_2p$ls.json').read().strip().splitlines()[-1]); print('2^$ls', round(d['ms_per_step'],2), 'ms', '%.3e' % d['value'])"; done
exit 0
