#!/usr/bin/env python3
"""Summarise the rocprofv3 PMC passes of bench.py (tools/gpu_pmc.sh, tools/gpu_r2c.sh) into profiles/r0N_pmc/bench_traffic.json.

Per kernel family: HBM bytes per launch = 2 x FETCH_SIZE (gfx950 counts a wide coalesced stream at half its bytes:
MI355X_MICROARCH.md §HBM, re-confirmed by profiles/r01_pmc/README.md's copy kernel) + WRITE_SIZE (1:1), both reported in KB,
averaged over the launches of the HBM-bound regime only: the `top` largest dispatches of each kernel in the LAST proof of the run,
which are the same launches bench.py brackets with HIP events (algorithmic bytes >= 256 MiB)."""
import csv
import json
import sys
from collections import defaultdict

FAMILIES = {   # bench.py family name -> kernel-name prefixes
    "bind_top(+fused linear round)": ("k_dot_eqw_fused", "k_bind_top"),
    "sumcheck_cubic_round(+fused bind)": ("k_cubic_eqw_fused", "k_cubic_eqw_lb", "k_cubic_round_lb"),
    "sumcheck_combine": ("k_dot_eqw_lb", "k_combine_round_linear", "k_combine_claim"),
    "multi_dot": ("k_multi_dot",),
    "matvec_left": ("k_matvec_left",),
    "fingerprint": ("k_fingerprint_ops",),          # k_fingerprint_ops_l1 since round 2 (leaves + first product layer)
    "gp_build": ("k_gp_layer",),
}
# families whose bench.py "launch" (one ProfScope bracket) is a SEQUENCE of dispatches — a product tree is one k_gp_layer dispatch per layer: the traffic per
# launch is the sum over all of the last proof's dispatches divided by the launches per proof, not the mean of the largest dispatches
SUMMED = {"gp_build": ("k_gp_layer", "k_gp_tail")}


def load(path):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            name = r["Kernel_Name"]
            rows.append((int(r["Dispatch_Id"]), name[5:] if name.startswith("void ") else name, float(r["Counter_Value"])))   # templated kernels print as "void k<..>(..)"
    rows.sort()
    return rows


def last_proof(rows):
    starts = [i for i, r in enumerate(rows) if r[1].startswith("k_gather")]
    return rows[starts[-1]:] if starts else rows


def main(fetch_csv, write_csv, bench_json, out_json, source):
    """bench_json: the JSON line bench.py printed under the FETCH_SIZE pass; its `large_launches_timed` gives, per family, how many launches
    per proof fall in the bracketed class — the PMC average is taken over the same number of largest dispatches."""
    fetch, write = last_proof(load(fetch_csv)), last_proof(load(write_csv))
    with open(bench_json) as f:
        line = json.loads(f.read().strip().splitlines()[-1])
    large = line.get("large_launches_timed", {})
    out = {"_workload": line.get("config", {}).get("workload_key", "and_c1_m16_2p24_curve25519"),      # bench.py applies a traffic file only to the workload it was taken on
           # ... and calls it measured only for the device sources it was taken on (bench.py roof(): anything else is labelled STALE)
           "_device_sources_sha256": line.get("lib_sha", {}).get("device_sources_sha256")}
    for fam, prefixes in FAMILIES.items():
        top = large.get(fam, {}).get("per_step", 0)
        if not top:
            continue
        f = sorted((v for _, n, v in fetch if n.startswith(prefixes)), reverse=True)[:top]
        w = sorted((v for _, n, v in write if n.startswith(prefixes)), reverse=True)[:top]
        if not f or not w:
            continue
        fb = 2.0 * 1024.0 * sum(f) / len(f)
        wb = 1024.0 * sum(w) / len(w)
        out[fam] = {"bytes_per_launch": round(fb + wb), "read_bytes_per_launch": round(fb), "write_bytes_per_launch": round(wb), "launches": len(f),
                    "alg_bytes_per_launch": large[fam]["alg_bytes_per_launch"], "traffic_over_algorithmic": round((fb + wb) / large[fam]["alg_bytes_per_launch"], 3), "source": source}
    for fam, prefixes in SUMMED.items():
        top = large.get(fam, {}).get("per_step", 0)
        if not top:
            continue
        fb = 2.0 * 1024.0 * sum(v for _, n, v in fetch if n.startswith(prefixes)) / top
        wb = 1024.0 * sum(v for _, n, v in write if n.startswith(prefixes)) / top
        out[fam] = {"bytes_per_launch": round(fb + wb), "read_bytes_per_launch": round(fb), "write_bytes_per_launch": round(wb), "launches": top,
                    "alg_bytes_per_launch": large[fam]["alg_bytes_per_launch"], "traffic_over_algorithmic": round((fb + wb) / large[fam]["alg_bytes_per_launch"], 3), "source": source,
                    "note": "sum over all dispatches of the family in the last proof / bracketed launches per proof (includes the small trees)"}
    with open(out_json, "w") as fo:
        json.dump(out, fo, indent=1)
    print(json.dumps(out, indent=1))
    return out


if __name__ == "__main__":
    main(*sys.argv[1:6])
