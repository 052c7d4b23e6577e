#!/bin/bash
# ONE parameterised script for everything that runs on the MI355X box (replaces the 62 one-off tools/gpu_r*.sh of rounds 1-3; tools/README.md maps every file under profiles/ to
# the recipe and arguments that produced it).  Usage on the box (through gpurun):   bash tools/gpu.sh <recipe> [args...]   — several recipes: separate with `--`.
# Every recipe writes under gpurun_out/<tag>/ (scratch, merged back); what is to be judged is copied into profiles/ by hand.  Every command runs under its own `timeout`.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; export TMPDIR=/tmp
say() { echo "=== $*"; }

r_tests() {        # tests [pytest -k expression] [tag]: the -m gpu suite (or a selection), summary line + failures
  local k="${1:-}" tag="${2:-tests}"; mkdir -p gpurun_out/$tag
  if [ -n "$k" ]; then timeout 1500 python -m pytest tests -m gpu -q -x -k "$k" -s > gpurun_out/$tag/pytest.log 2>&1; else timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/$tag/pytest.log 2>&1; fi
  grep -E "passed|failed|error" gpurun_out/$tag/pytest.log | tail -2; grep -E "^FAILED|^ERROR|Error" gpurun_out/$tag/pytest.log | head -5; grep -E "^\[(slab|oracle|verify)\]" gpurun_out/$tag/pytest.log
}
r_tests_bn254() {  # tests_bn254 [k] [tag]: the entry-point cases on the BN254 library pair
  local k="${1:-}" tag="${2:-tests_bn254}"; mkdir -p gpurun_out/$tag
  LASSO_TEST_CURVE=bn254 timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_bn254.py -m gpu -q -x ${k:+-k "$k"} > gpurun_out/$tag/pytest.log 2>&1; grep -E "passed|failed|error" gpurun_out/$tag/pytest.log | tail -2
}
r_bench() {        # bench <tag> [bench.py args...]: one bench.py run, JSON line kept, headline printed
  local tag="$1"; shift; mkdir -p gpurun_out/$tag
  timeout 900 python bench.py "$@" > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err
  python - "$tag" <<'PY'
import json, sys
tag = sys.argv[1]
try:
    d = json.loads(open(f"gpurun_out/{tag}/bench.json").read().strip().splitlines()[-1])
except Exception as e:
    print("no bench line:", e); print(open(f"gpurun_out/{tag}/bench.err").read()[-1500:]); sys.exit(0)
rf = d.get("roofline") or {}; rm = d.get("roofline_msm") or {}
print(f"{tag}: {d.get('ms_per_step', 0):.3f} ms/step  value {d.get('value', 0):.4g}  roofline frac {rf.get('frac')} traffic {rf.get('frac_traffic')}  parity {(d.get('parity_checked') or {}).get('equal')}")
for k in ("commit", "opening"):
    if rm.get(k): print(f"  msm {k}: frac {rm[k].get('frac')} achieved {rm[k].get('achieved')} launches {rm[k].get('launches')} avg {rm[k].get('avg_launch_us')} us")
for k in d.get("kernels_one_profiled_step", []): print(f"  {k['kernel']:38s} {k['launches']:4d} launches {k['ms']:8.3f} ms  {k.get('alg_GBps')} GB/s")
cp = d.get("concurrent_proofs")
if cp: print("  concurrent:", [(x["streams"], round(x["value"] / 1e9, 3), x["proofs_differing_from_sequential"]) for x in cp.get("sweep", [])], cp.get("error"))
bs = d.get("bind_top_sweep")
if bs: print("  bind_top_sweep:", [(x["log_n"], x["polys"], x.get("alg_GBps"), x.get("frac"), x.get("error")) for x in bs["rows"]])
sm = d.get("slab_mode")
if sm: print("  slab:", {k: sm.get(k) for k in ("ms_per_proof", "parity", "peak_bytes_per_rank", "model_bytes_per_rank", "error", "skipped")}, "capacity:", sm.get("capacity_mode"))
PY
}
r_prof() {         # prof <tag> [bench.py args...]: rocprofv3 --kernel-trace --stats of a short bench run; csv files kept
  local tag="$1"; shift; mkdir -p gpurun_out/$tag
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o bench -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --concurrent 0 --no-slab-leg --no-bind-sweep "$@" > $ROOT/gpurun_out/$tag/bench_under_rocprof.json 2> $ROOT/gpurun_out/$tag/rocprof.err)
  find /tmp/prof_$tag -name '*stats.csv' -exec cp {} gpurun_out/$tag/ \; ; ls gpurun_out/$tag | head
}
r_pmc() {          # pmc <tag> [bench.py args...]: the two separate PMC passes (FETCH_SIZE, WRITE_SIZE; only --kernel-trace beside --pmc) + tools/pmc_summary.py
  local tag="$1"; shift; mkdir -p gpurun_out/$tag
  for CTR in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout 900 rocprofv3 --pmc $CTR --kernel-trace --output-format csv -d /tmp/pmc_${tag}_$CTR -o bench -- python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --concurrent 0 --no-slab-leg --no-bind-sweep "$@" > $ROOT/gpurun_out/$tag/bench_$CTR.json 2> $ROOT/gpurun_out/$tag/rocprof_$CTR.err)
    f=$(find /tmp/pmc_${tag}_$CTR -name '*counter_collection.csv' | head -1); [ -n "$f" ] && cp $f gpurun_out/$tag/bench_${CTR}_counter_collection.csv
  done
  python tools/pmc_summary.py gpurun_out/$tag/bench_FETCH_SIZE_counter_collection.csv gpurun_out/$tag/bench_WRITE_SIZE_counter_collection.csv gpurun_out/$tag/bench_FETCH_SIZE.json gpurun_out/$tag/bench_traffic.json "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of bench.py $* (tools/gpu.sh pmc)" | tail -30
}
r_pmc_bind() {     # pmc_bind <tag>: PMC passes + kernel stats of the kernel-level bind_top sweep (bench.py --only-bind-sweep)
  local tag="$1"; mkdir -p gpurun_out/$tag
  for CTR in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout 600 rocprofv3 --pmc $CTR --kernel-trace --output-format csv -d /tmp/pmcb_${tag}_$CTR -o bench -- python $ROOT/bench.py --only-bind-sweep > $ROOT/gpurun_out/$tag/sweep_$CTR.json 2> $ROOT/gpurun_out/$tag/rocprof_$CTR.err)
    f=$(find /tmp/pmcb_${tag}_$CTR -name '*counter_collection.csv' | head -1); [ -n "$f" ] && cp $f gpurun_out/$tag/sweep_${CTR}_counter_collection.csv
  done
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/statb_$tag -o bench -- python $ROOT/bench.py --only-bind-sweep > $ROOT/gpurun_out/$tag/sweep_stats_run.json 2> $ROOT/gpurun_out/$tag/rocprof_stats.err)
  find /tmp/statb_$tag -name '*kernel_stats.csv' -exec cp {} gpurun_out/$tag/ \;
  python tools/pmc_bind_summary.py gpurun_out/$tag | tail -20
}
r_memtable() {     # memtable <tag> [tools/slab_mem_table.py args...]
  local tag="$1"; shift; mkdir -p gpurun_out/$tag
  timeout 1200 python tools/slab_mem_table.py --out gpurun_out/$tag/slab_peak_bytes.json "$@" > gpurun_out/$tag/memtable.log 2> gpurun_out/$tag/memtable.err; grep '"world"' gpurun_out/$tag/memtable.log | cut -c1-400; tail -3 gpurun_out/$tag/memtable.err
}
r_benv() {         # benv <tag> "<VAR=val VAR=val ...>" [bench.py args...]: r_bench under environment switches (A/B runs); the quick form of the bench (no CPU baseline / sweeps)
  local tag="$1" envs="$2"; shift 2; mkdir -p gpurun_out/$tag
  env $envs timeout 600 python bench.py --no-cpu-baseline --concurrent 0 --no-slab-leg --no-bind-sweep "$@" > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err
  python - "$tag" "$envs" <<'PY'
import json, sys
tag, envs = sys.argv[1], sys.argv[2]
try:
    d = json.loads(open(f"gpurun_out/{tag}/bench.json").read().strip().splitlines()[-1])
    print(f"{tag} [{envs}]: {d.get('ms_per_step', 0):.3f} ms/step  parity {(d.get('parity_checked') or {}).get('equal')}  roofline {((d.get('roofline') or {}).get('frac'))}")
except Exception as e:
    print(f"{tag} [{envs}]: no bench line: {e}"); print(open(f"gpurun_out/{tag}/bench.err").read()[-1200:])
PY
}
r_ab() {           # ab <tag> <reps> "<env A>" "<env B>" ...: alternating runs of the quick bench under each environment (one process per run: the switches are read once); median / min / all
  local tag="$1" reps="$2"; shift 2; mkdir -p gpurun_out/$tag; : > gpurun_out/$tag/ab.txt
  for rep in $(seq 1 $reps); do
    for envs in "$@"; do
      ms=$(env $envs timeout 300 python bench.py --no-cpu-baseline --concurrent 0 --no-slab-leg --no-bind-sweep --steps ${AB_STEPS:-30} --warmup 5 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])" 2>/dev/null)
      echo "$envs|$ms" >> gpurun_out/$tag/ab.txt
    done
  done
  python - gpurun_out/$tag/ab.txt <<'PY'
import sys, collections, statistics
runs = collections.OrderedDict()
for l in open(sys.argv[1]):
    k, v = l.rstrip("\n").split("|")
    if v: runs.setdefault(k, []).append(float(v))
for k, v in runs.items():
    print(f"{k:60s} median {statistics.median(v):7.3f}  min {min(v):7.3f}  " + " ".join(f"{x:.2f}" for x in v))
PY
}
r_trace() {        # trace <tag> [bench.py args...]: per-launch kernel trace of ONE proof (timeline csv: start, duration, gap before) + LASSO_TRACE=1 spans + LASSO_TRACE=2 host buckets
  local tag="$1"; shift; mkdir -p gpurun_out/$tag
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_$tag -o bench -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --concurrent 0 --no-slab-leg --no-bind-sweep --no-prof "$@" > $ROOT/gpurun_out/$tag/bench_under_trace.json 2> $ROOT/gpurun_out/$tag/trace.err)
  f=$(find /tmp/kt_$tag -name '*kernel_trace.csv' | head -1)
  python - "$f" > gpurun_out/$tag/kernel_trace_one_proof.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1]))); rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_gather_u32")]
rows = rows[idx[-1]:] if idx else rows
t0 = int(rows[0]["Start_Timestamp"]); prev = t0
print("start_us,dur_us,gap_us,grid,wg,kernel")
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%.2f,%.2f,%.2f,%s,%s,%s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")), r["Kernel_Name"].split("(")[0].replace(",", ";")[:60]))
    prev = max(prev, e)
PY
  wc -l gpurun_out/$tag/kernel_trace_one_proof.csv; tail -1 gpurun_out/$tag/kernel_trace_one_proof.csv
  LASSO_TRACE=1 timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --concurrent 0 --no-slab-leg --no-bind-sweep --no-prof "$@" > /dev/null 2> gpurun_out/$tag/trace_spans.txt; grep "\[trace\]" gpurun_out/$tag/trace_spans.txt | tail -22
  LASSO_TRACE=2 timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --concurrent 0 --no-slab-leg --no-bind-sweep --no-prof "$@" > /dev/null 2> gpurun_out/$tag/host_buckets.txt; grep "\[host\]" gpurun_out/$tag/host_buckets.txt | tail -14
}
r_numa() {         # numa <tag> [reps]: where the GPU hangs (NUMA node of its PCIe function) and what the quick bench takes with the process confined to each node's CPUs (taskset) and unconfined
  local tag="$1" reps="${2:-2}"; mkdir -p gpurun_out/$tag; local out=gpurun_out/$tag/numa.txt; : > $out
  { lscpu | grep -iE "^CPU\(s\)|socket|thread|numa|model name"; for f in /sys/class/drm/card*/device/numa_node; do echo "$f: $(cat $f 2>/dev/null)"; done; } >> $out 2>&1
  local nodes; nodes=$(ls -d /sys/devices/system/node/node* 2>/dev/null | sed 's/.*node//' | sort -n)
  for rep in $(seq 1 $reps); do
    for n in $nodes free; do
      local pre=""; [ "$n" != free ] && pre="taskset -c $(cat /sys/devices/system/node/node$n/cpulist)"
      ms=$($pre timeout 300 python bench.py --no-cpu-baseline --concurrent 0 --no-slab-leg --no-bind-sweep --steps 30 --warmup 5 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])" 2>/dev/null)
      echo "node $n: $ms ms per proof" >> $out
    done
  done
  cat $out
}
r_micro() {        # micro <tag> [n ...]: round 6's microbenchmarks — phase-stamped timeline of one k_bullet_msm launch per curve (tools/bullet_phase_bench*) and the mixed addition's ceiling (tools/madd_bench*)
  local tag="$1"; shift; mkdir -p gpurun_out/$tag
  for n in "${@:-4096}"; do
    timeout 120 tools/bullet_phase_bench $n > gpurun_out/$tag/bullet_phase_curve25519_n$n.txt 2>&1; timeout 120 tools/bullet_phase_bench_bn254 $n > gpurun_out/$tag/bullet_phase_bn254_n$n.txt 2>&1
  done
  timeout 200 tools/madd_bench > gpurun_out/$tag/madd_bench_curve25519.txt 2>&1; timeout 200 tools/madd_bench_bn254 > gpurun_out/$tag/madd_bench_bn254.txt 2>&1
  grep -h "####\|MADD_CEILING\|k_msm_rows8w" gpurun_out/$tag/*.txt | cut -c1-260
}
r_scale() {        # scale <tag> [bench.py args...]: FIRST CONTACT WITH AN 8-GPU NODE.  bench.py --gpus N for N = 1, 2, 4, 8 (those the node has) in both modes —
                   # independent proofs (weak scaling: the driver's SCALE line) and ONE proof sharded over the N GPUs (--shard-proof, slab mode, configs[3] by default) — then one table:
                   # N, mode, lookups/s, ms per step, x vs N = 1, ranks that joined the RCCL communicator, the exchange used, per-rank peak bytes, parity.  Every leg under its own timeout.
  local tag="$1"; shift; mkdir -p gpurun_out/$tag
  local ngpu; ngpu=$(python -c "import torch; print(torch.cuda.device_count())" 2>/dev/null || echo 1)
  say "visible GPUs: $ngpu"
  for N in 1 2 4 8; do
    [ "$N" -gt "$ngpu" ] && { echo "N=$N: skipped (the node shows $ngpu GPUs)"; continue; }
    local port=$((29500 + N))
    if [ "$N" -eq 1 ]; then
      timeout 900 python bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --concurrent 0 --no-slab-leg --no-bind-sweep "$@" > gpurun_out/$tag/indep_$N.json 2> gpurun_out/$tag/indep_$N.err
      timeout 900 python bench.py --gpus 1 --steps 3 --warmup 1 --kind range --c 4 --log-s 26 --no-cpu-baseline --concurrent 0 --no-slab-leg --no-bind-sweep > gpurun_out/$tag/slab_$N.json 2> gpurun_out/$tag/slab_$N.err
    else
      HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $port bench.py --gpus $N --steps 10 --warmup 3 --no-cpu-baseline --concurrent 0 --no-slab-leg --no-bind-sweep "$@" > gpurun_out/$tag/indep_$N.json 2> gpurun_out/$tag/indep_$N.err
      HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((port + 20)) bench.py --gpus $N --shard-proof --steps 3 --warmup 1 --kind range --c 4 --log-s 26 --no-cpu-baseline --concurrent 0 --no-slab-leg --no-bind-sweep > gpurun_out/$tag/slab_$N.json 2> gpurun_out/$tag/slab_$N.err
    fi
  done
  python - "$tag" <<'PY'
import json, os, sys
tag = sys.argv[1]
def line(path):
    try:
        return json.loads([l for l in open(path).read().strip().splitlines() if l.startswith("{")][-1])
    except Exception as e:
        return None
print(f"{'N':>2} {'mode':<11} {'lookups/s':>12} {'ms/step':>9} {'x vs N=1':>9} {'rccl':>5}  {'peak GB/rank':>12}  parity  exchange")
base = {}
for mode in ("indep", "slab"):
    for N in (1, 2, 4, 8):
        d = line(f"gpurun_out/{tag}/{mode}_{N}.json")
        if d is None:
            err = f"gpurun_out/{tag}/{mode}_{N}.err"
            if os.path.exists(err): print(f"{N:>2} {mode:<11} no bench line — {open(err).read()[-300:].strip()!r}")
            continue
        mg = d.get("multi_gpu") or {}
        if N == 1: base[mode] = d["value"]
        peaks = mg.get("peak_bytes_per_rank") or []
        par = (d.get("parity_checked") or {}).get("equal", (d.get("slab_mode") or {}).get("parity"))
        print(f"{N:>2} {mode:<11} {d['value']:12.4g} {d['ms_per_step']:9.2f} {d['value'] / base.get(mode, d['value']):9.2f} {mg.get('rccl_ranks', 0):>5}  {max(peaks) / 1e9 if peaks else 0:12.2f}  {str(par):<6}  {str(mg.get('exchange'))[:110]}")
PY
}
r_sh() { "$@"; }   # sh <command...>: anything else, verbatim

while [ $# -gt 0 ]; do
  recipe="$1"; shift; args=()
  while [ $# -gt 0 ] && [ "$1" != "--" ]; do args+=("$1"); shift; done
  [ $# -gt 0 ] && shift
  say "$recipe ${args[*]:-}"
  "r_$recipe" "${args[@]}"
done
exit 0
