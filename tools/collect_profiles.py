#!/usr/bin/env python3
"""Copy what is to be judged from gpurun_out/ (scratch) into profiles/ (tracked), named per round: the evidence pass of `tools/gpu.sh pmc / pmc_bind / prof / trace / bench`.
Usage: python tools/collect_profiles.py r05 [prefix of the gpurun_out tags, default r5_]
  <prefix>pmc_headline, <prefix>pmc_xor      -> profiles/<round>_pmc/<workload key>/ (bench_traffic.json, the two bench lines, counter rows of the LAST proof only)
  <prefix>pmc_bind                           -> profiles/<round>_pmc/bind_top_sweep/
  <prefix>prof[_bn254|_xor]                  -> profiles/<round>_rocprofv3/bench_*_kernel_stats.csv
  <prefix>trace                              -> profiles/<round>_kernel_trace_one_proof_2p24.csv, <round>_trace_spans_2p24.txt, <round>_host_buckets_2p24.txt
  <prefix><name>/bench.json                  -> profiles/<round>_bench_<name>.json"""
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1]; pre = sys.argv[2] if len(sys.argv) > 2 else "r5_"
G = os.path.join(ROOT, "gpurun_out"); P = os.path.join(ROOT, "profiles")


def last_proof_rows(src, dst):
    rows = list(csv.DictReader(open(src)))
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_gather")]
    rows = rows[idx[-1]:] if idx else rows
    keep = ["Dispatch_Id", "Kernel_Name", "Grid_Size", "Workgroup_Size", "Counter_Name", "Counter_Value"]
    keep = [k for k in keep if k in rows[0]]
    with open(dst, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=keep); w.writeheader()
        for r in rows:
            r = {k: r[k] for k in keep}; r["Kernel_Name"] = r["Kernel_Name"].split("(")[0]
            w.writerow(r)


for tag in sorted(os.listdir(G)):
    d = os.path.join(G, tag)
    if not tag.startswith(pre) or not os.path.isdir(d):
        continue
    name = tag[len(pre):]
    if name.startswith("pmc_bind"):
        out = os.path.join(P, f"{rnd}_pmc", "bind_top_sweep"); os.makedirs(out, exist_ok=True)
        for f in ("bench_traffic.json", "sweep_FETCH_SIZE.json", "bench_kernel_stats.csv"):
            if os.path.exists(os.path.join(d, f)):
                shutil.copy(os.path.join(d, f), os.path.join(out, {"sweep_FETCH_SIZE.json": "sweep_under_FETCH_SIZE_pass.json", "bench_kernel_stats.csv": "sweep_kernel_stats.csv"}.get(f, f)))
        os.makedirs(os.path.join(P, f"{rnd}_rocprofv3"), exist_ok=True)
        if os.path.exists(os.path.join(d, "bench_kernel_stats.csv")):
            shutil.copy(os.path.join(d, "bench_kernel_stats.csv"), os.path.join(P, f"{rnd}_rocprofv3", "bind_top_sweep_kernel_stats.csv"))
    elif name.startswith("pmc"):
        t = json.load(open(os.path.join(d, "bench_traffic.json")))
        out = os.path.join(P, f"{rnd}_pmc", t["_workload"]); os.makedirs(out, exist_ok=True)
        shutil.copy(os.path.join(d, "bench_traffic.json"), out)
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            shutil.copy(os.path.join(d, f"bench_{ctr}.json"), os.path.join(out, f"bench_under_pmc_{ctr}.json"))
            last_proof_rows(os.path.join(d, f"bench_{ctr}_counter_collection.csv"), os.path.join(out, f"bench_{ctr}_counter_collection_last_proof.csv"))
    elif name.startswith("prof"):
        out = os.path.join(P, f"{rnd}_rocprofv3"); os.makedirs(out, exist_ok=True)
        suffix = {"prof": "2p24", "prof_bn254": "bn254_config1", "prof_xor": "xor_c8_2p24"}.get(name, name)
        shutil.copy(os.path.join(d, "bench_kernel_stats.csv"), os.path.join(out, f"bench_{suffix}_kernel_stats.csv"))
    elif name.startswith("trace"):
        shutil.copy(os.path.join(d, "kernel_trace_one_proof.csv"), os.path.join(P, f"{rnd}_kernel_trace_one_proof_2p24.csv"))
        for src, dst in (("trace_spans.txt", f"{rnd}_trace_spans_2p24.txt"), ("host_buckets.txt", f"{rnd}_host_buckets_2p24.txt")):
            lines = [l for l in open(os.path.join(d, src)) if l.startswith(("[trace]", "[host]"))]
            open(os.path.join(P, dst), "w").writelines(lines[-40:])
    elif os.path.exists(os.path.join(d, "bench.json")):
        txt = open(os.path.join(d, "bench.json")).read().strip().splitlines()
        if txt:
            json.dump(json.loads(txt[-1]), open(os.path.join(P, f"{rnd}_bench_{name}.json"), "w"), indent=1)
print("collected into profiles/:", sorted(f for f in os.listdir(P) if f.startswith(rnd)))
