#!/usr/bin/env python3
"""HBM traffic of the kernel-level bind_top sweep from the two rocprofv3 PMC passes of `bench.py --only-bind-sweep` (tools/gpu.sh pmc_bind): per (n, polys) row the average
over that row's k_bind_top dispatches of 2 x FETCH_SIZE (gfx950 counts a wide coalesced stream at half its bytes, MI355X_MICROARCH.md) + WRITE_SIZE, both in KB, against the
algorithmic 48 n p bytes.  The sweep launches rows in a fixed order with a fixed number of launches per row (warmup + iterations), which is how dispatches map to rows.
Writes <dir>/bench_traffic.json (bench.py attaches it to `bind_top_sweep.traffic` when it is committed under profiles/r0N_pmc/bind_top_sweep/, and labels it stale when the device sources have changed since)."""
import csv
import json
import os
import sys


def load(path):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            name = r["Kernel_Name"]; name = name[5:] if name.startswith("void ") else name
            if name.startswith("k_bind_top"):
                rows.append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    rows.sort()
    return [v for _, v in rows]


def main(d):
    whole = json.loads(open(os.path.join(d, "sweep_FETCH_SIZE.json")).read().strip().splitlines()[-1])
    line = whole["bind_top_sweep"]
    per_row = line["iterations"] + line["warmup"]
    fetch, write = load(os.path.join(d, "sweep_FETCH_SIZE_counter_collection.csv")), load(os.path.join(d, "sweep_WRITE_SIZE_counter_collection.csv"))
    rows = [x for x in line["rows"] if "launches" in x]
    out = {"_curve": "curve25519", "_device_sources_sha256": whole.get("lib_sha", {}).get("device_sources_sha256"), "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of `bench.py --only-bind-sweep` (tools/gpu.sh pmc_bind, tools/pmc_bind_summary.py)", "rows": []}
    ok = len(fetch) == per_row * len(rows) == len(write)
    out["dispatches_match_the_sweep"] = ok
    for i, x in enumerate(rows if ok else []):
        f = fetch[i * per_row + line["warmup"]:(i + 1) * per_row]; w = write[i * per_row + line["warmup"]:(i + 1) * per_row]
        rb, wb = 2.0 * 1024.0 * sum(f) / len(f), 1024.0 * sum(w) / len(w)
        out["rows"].append({"log_n": x["log_n"], "polys": x["polys"], "read_bytes_per_launch": round(rb), "write_bytes_per_launch": round(wb), "bytes_per_launch": round(rb + wb),
                            "alg_bytes_per_launch": x["alg_bytes_per_launch"], "traffic_over_algorithmic": round((rb + wb) / x["alg_bytes_per_launch"], 3)})
    with open(os.path.join(d, "bench_traffic.json"), "w") as fo:
        json.dump(out, fo, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1])
