#!/usr/bin/env python3
"""Per-rank device memory of ONE proof in slab mode, measured: BASELINE.json configs[3] (RangeCheck, C=4, 2^26 lookups) over P = 1, 2, 4, 8 ranks — P contexts of
the one MI355X, each holding 1/P of every polynomial (tests/cpp/slab_threads.cpp over lasso_host_set_comm_shm) — in pooled and in capacity mode (lasso_host_set_capacity).
Per run: the largest per-rank high-water mark of lasso_mem_stats, the most the prover itself had in use, ms per proof, byte parity with the oracle's digests, and bench.py's
model (slab_bytes_per_rank) beside it.  Prints one JSON object; `--out FILE` also writes it (profiles/r04_slab_peak_bytes.json)."""
import argparse
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kind", default="range"); ap.add_argument("--c", type=int, default=4); ap.add_argument("--log-s", type=int, default=26)
    ap.add_argument("--log-m", type=int, default=16); ap.add_argument("--log-r", type=int, default=40)
    ap.add_argument("--worlds", default="1,2,4,8"); ap.add_argument("--modes", default="pooled,capacity"); ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    import importlib.util
    from lasso_amd import HostProver, _abi
    import test_gpu_prover as T
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    lib = T._build_slab_hip()
    hp = HostProver()
    s = 1 << a.log_s
    alpha = 2 * a.c if a.kind == "lt" else a.c
    idx = hp.gen_indices(s, 1 << a.log_m, a.c); r = hp.gen_random_point(a.log_s)
    hp.close()
    S = _abi.Strategy(_abi.KINDS[a.kind], a.c, a.log_m, a.log_r if a.kind == "range" else 0)
    gold = bench.golden_digest(a.kind, a.c, a.log_m, S.log_r, a.log_s)
    rows = []
    for w in [int(x) for x in a.worlds.split(",")]:
        for mode in a.modes.split(","):
            try:
                comm, proof, info = T.run_slab_threads(lib, w, S, alpha, idx, r, shm_name=f"/lasso_memtab_{os.getpid()}_{w}_{mode}", capacity=(mode == "capacity"), steps=a.steps)
            except AssertionError as e:
                rows.append({"world": w, "mode": mode, "error": str(e)[:300]}); continue
            row = {"world": w, "mode": mode, "peak_bytes_per_rank": max(info["peak_bytes_per_rank"]), "prover_peak_bytes_per_rank": max(info["prover_peak_bytes_per_rank"]),
                   "sum_over_ranks_GiB": round(sum(info["peak_bytes_per_rank"]) / 2**30, 2), "ms_per_proof": round(info["ms_per_proof"], 2),
                   "model_bytes_per_rank": int(bench.slab_bytes_per_rank(a.kind, a.c, a.log_s, w, a.log_m, mode == "capacity")), "proof_sha256": hashlib.sha256(proof).hexdigest()}
            row["peak_GiB_per_rank"] = round(row["peak_bytes_per_rank"] / 2**30, 2); row["model_over_measured"] = round(row["model_bytes_per_rank"] / row["peak_bytes_per_rank"], 3)
            if gold:
                row["parity"] = row["proof_sha256"] == gold["proof_sha256"] and hashlib.sha256(comm).hexdigest() == gold["commitment_sha256"]
            rows.append(row)
            print(json.dumps(row), flush=True)
    out = {"workload": f"{a.kind.upper()} C={a.c} M=2^{a.log_m} s=2^{a.log_s}, ONE proof over P contexts of one MI355X (slab mode, lasso_host_set_comm_shm)", "rows": rows,
           "extrapolation_C16_2p28_world8": {f"{k}{'_capacity' if cap else ''}": {"bytes_per_rank": int(bench.slab_bytes_per_rank(k, 16, 28, 8, 16, cap)), "units_of_s_over_P": bench.slab_units(k, 16, cap)}
                                             for k in ("and", "spark", "lt") for cap in (False, True)}}
    print(json.dumps(out))
    if a.out:
        with open(a.out, "w") as f:
            json.dump(out, f, indent=1); f.write("\n")


if __name__ == "__main__":
    main()
