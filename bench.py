#!/usr/bin/env python3
"""bench.py — prover lookups/s for SparsePolynomialEvaluationProof on MI355X (BASELINE.json metric).

A "step" = one SparsePolynomialEvaluationProof::prove over one batch of s synthetic lookups whose densified representation
is already resident in HBM (the reference's `SparsePoly.prove` span; generator construction, densify, commit and verify are
outside the timed region exactly as in the reference's published logs).  Default workload = the configuration the metric
is quoted on: AND table, C=1, M=2^16, s=2^24 (src/benches/bench.rs `halo2_comparison_benchmarks`, src/benches/*.log).

N > 1 (torch.distributed / RCCL, one process per GPU): each rank proves an independent batch of s lookups — the path
partitions by proof, there is no data-path collective — so value = N*s*K / max-over-ranks time and scaling is "weak".

One JSON line on rank 0.  Extra objects: roofline (dominant streaming kernel, HIP events on the library's stream),
kernels (every kernel family: launches, ms, algorithmic GB/s), cpu_baseline (oracle port timed on host cores, rank 0, N=1).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec peak
# the only published number for this exact metric — 2^24 AND lookups, C=1, `SparsePoly.prove` span, rayon on an Apple M1 16 GB:
# 35.3 s (src/benches/m1_16gb_parallel_benches.log:439) = 2^24 / 35.3 lookups/s.  Other hardware: a reference point, not a like-for-like comparison.
PUBLISHED_LOOKUPS_PER_S = (1 << 24) / 35.3


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=3)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--log-s", type=int, default=24, help="log2 of lookups per proof (default 24 = the headline metric)")
    p.add_argument("--c", type=int, default=1)
    p.add_argument("--kind", default="and", choices=["and", "or", "xor", "lt", "range"])
    p.add_argument("--log-m", type=int, default=16)
    p.add_argument("--log-r", type=int, default=40)
    p.add_argument("--cpu-log-s", type=int, default=22, help="log2 lookups of the bounded CPU-baseline sample (2^22: ~20 s of one host core)")
    p.add_argument("--curve", default="curve25519", choices=["curve25519", "bn254"], help="the group G: curve25519 (the reference harness's, the headline metric) or BN254 G1 "
                                                                                          "(BASELINE.json configs[1]; the liblasso_*_bn254.so pair)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-prof", action="store_true")
    p.add_argument("--concurrent", type=int, default=4, help="extra leg at N=1: this many independent proofs proved concurrently on the one GPU (own context, stream and host "
                                                              "thread each); reported beside the headline, never as `value`.  0 = skip")
    p.add_argument("--backend", default=None, choices=["nccl", "gloo"], help="torch.distributed backend for N > 1 (default: nccl = RCCL when a GPU is visible).  gloo lets the N > 1 "
                                                                              "code path be exercised on a box with fewer GPUs than ranks (ranks share devices)")
    p.add_argument("--shard-proof", action="store_true", help="N > 1: ONE proof of s lookups sharded over the N GPUs by low index bits (slab mode, strong scaling) "
                                                                "instead of one independent proof per GPU (the default, weak scaling)")
    return p.parse_args()


def cpu_baseline(kind_id, c, log_m, log_r, log_s, curve="curve25519"):
    """Oracle ("port") prover, serial, on this box's host cores; a bounded sample of the same workload shape."""
    import subprocess
    so = "liblasso_oracle_bn254.so" if curve == "bn254" else "liblasso_oracle.so"
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), so])
    orc = C.CDLL(os.path.join(ROOT, "oracle", so))
    td, tc, tp = C.c_double(), C.c_double(), C.c_double()
    rc = orc.orc_bench(kind_id, C.c_size_t(c), C.c_size_t(1 << log_m), C.c_size_t(log_r), C.c_size_t(1 << log_s), C.byref(td), C.byref(tc), C.byref(tp), 0)
    if rc != 0:
        return None
    return {"value": (1 << log_s) / tp.value, "unit": "lookups/s", "cores": 1, "kind": "port",
            "sample": f"oracle (serial C++ restatement) prove, {kind_id=} C={c} M=2^{log_m} s=2^{log_s}: {tp.value:.2f}s (densify {td.value:.3f}s, commit {tc.value:.2f}s)"}


def pmc_traffic():
    """HBM bytes per launch from the committed rocprofv3 PMC passes of this same command (profiles/r01_pmc/bench_traffic.json, written by
    tools/pmc_summary.py: FETCH_SIZE x2 per MI355X_MICROARCH.md's gfx950 correction + WRITE_SIZE, large launches only).  {} when absent."""
    path = os.path.join(ROOT, "profiles", "r01_pmc", "bench_traffic.json")
    try:
        with open(path) as f:
            return json.load(f)
    except Exception:
        return {}


def concurrent_leg(HostProver, _abi, streams, steps, S, c, log_m, log_s, curve="curve25519"):
    """T independent proofs at a time on one GPU.  One proof is latency-bound by its ~470 sequential transcript rounds (the device idles ~25% of the time
    at 2^24 lookups); independent proofs on their own streams fill the gaps.  Serving-style throughput, reported separately from the one-proof-at-a-time metric."""
    import threading
    s = 1 << log_s
    alpha = 2 * c if S.kind == _abi.KINDS["lt"] else c
    workers = []
    for t in range(streams):
        hp = HostProver(curve=curve)
        idx = (hp.gen_indices(s, 1 << log_m, c) + t) % (1 << log_m)
        r = hp.gen_random_point(log_s)
        gens = hp.gens(c, s, alpha, log_m); dense = hp.densify(idx, log_m)
        hp.prove(dense, gens, S, r)      # warm-up
        workers.append((hp, dense, gens, r))
    bar = threading.Barrier(streams + 1)

    def run(w):
        hp, dense, gens, r = w
        bar.wait()
        for _ in range(steps):
            hp.prove(dense, gens, S, r)
        bar.wait()
    ths = [threading.Thread(target=run, args=(w,)) for w in workers]
    for th in ths:
        th.start()
    bar.wait(); t0 = time.perf_counter(); bar.wait(); el = time.perf_counter() - t0
    for th in ths:
        th.join()
    for hp, dense, gens, r in workers:
        hp.free(dense, gens); hp.close()
    return {"streams": streams, "proofs": streams * steps, "value": streams * steps * s / el, "unit": "lookups/s", "ms_per_round_of_proofs": el / steps * 1e3,
            "note": "independent proofs proved concurrently on one GPU (one context, stream and host thread each); not the headline metric"}


def main():
    a = parse()
    from lasso_amd import HostProver, _abi
    from lasso_amd.parallel import Group, shard_indices
    grp = Group(backend=a.backend)     # torch.distributed (nccl = RCCL) only when WORLD_SIZE > 1
    rank, world = grp.rank, grp.world
    hp = HostProver(device=grp.device_index, curve=a.curve)
    slab = a.shard_proof and world > 1
    if slab:
        hp.set_comm(grp)             # slab mode: every polynomial split by low index bits, RCCL all_gather of per-round sums / row commitments
    lib = hp.lib
    dev_lib = C.CDLL(os.path.join(ROOT, "lasso_amd", "liblasso_hip_bn254.so" if a.curve == "bn254" else "liblasso_hip.so"))
    _abi.declare(dev_lib)
    ctx = hp.ctx()

    kind_id = _abi.KINDS[a.kind]
    c, log_m, s = a.c, a.log_m, 1 << a.log_s
    alpha = 2 * c if a.kind == "lt" else c
    S = _abi.Strategy(kind_id, c, log_m, a.log_r if a.kind == "range" else 0)

    t0 = time.time()
    idx = shard_indices(hp, s, 1 << log_m, c, 0 if slab else rank)   # benches/bench.rs:13-21 (rank 0 exactly; other ranks: their own batch, or the same lookups in slab mode)
    r = hp.gen_random_point(a.log_s)                        # benches/bench.rs:27-34
    gens = hp.gens(c, s, alpha, log_m)                      # SparsePolyCommitmentGens::new(b"gens_sparse_poly", C, S, C, log_m)
    t_setup = time.time() - t0
    t0 = time.time(); dense = hp.densify(idx, log_m); dev_lib.lasso_sync(ctx); t_densify = time.time() - t0
    t0 = time.time(); comm = hp.commit(dense, gens); t_commit = time.time() - t0
    t0 = time.time(); hp.commit(dense, gens); t_commit_warm = time.time() - t0      # the first call also pays first-use costs (kernel load, table faults)

    def barrier():
        dev_lib.lasso_sync(ctx)
        grp.barrier()

    proof = None
    for _ in range(a.warmup):
        proof = hp.prove(dense, gens, S, r)
    # Roofline numbers are measured live, inside the timed region, with HIP events on the library's stream — but only around the launches
    # in the HBM-bound regime (>= 256 MiB of algorithmic bytes, past the Infinity Cache: ~20 per proof), so the brackets cost nothing.
    # Bracketing all ~950 launches of a proof adds ~8% wall time; that full per-family table comes from one extra, untimed, profiled step.
    STREAM = [_abi.K_BIND, _abi.K_CUBIC, _abi.K_COMBINE, _abi.K_EQ, _abi.K_GP, _abi.K_FINGERPRINT, _abi.K_DOT, _abi.K_MATVEC]
    LARGE_ONLY = 0x40000000
    if not a.no_prof:
        dev_lib.lasso_prof_reset(ctx); dev_lib.lasso_prof_enable(ctx, sum(1 << k for k in STREAM) | LARGE_ONLY)
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        proof = hp.prove(dense, gens, S, r)
    dev_lib.lasso_sync(ctx)
    elapsed = time.perf_counter() - t0
    barrier()

    def family(kid, large):
        n = C.c_uint64(); ms = C.c_double(); b = C.c_double()
        (dev_lib.lasso_prof_get_large if large else dev_lib.lasso_prof_get)(ctx, kid, C.byref(n), C.byref(ms), C.byref(b))
        if not n.value:
            return None
        return {"kernel": _abi.KERNEL_NAMES[kid], "launches": n.value, "ms": round(ms.value, 3), "alg_GB": round(b.value / 1e9, 3),
                "alg_GBps": round(b.value / (ms.value * 1e-3) / 1e9, 1) if ms.value > 0 else None, "avg_launch_us": round(ms.value * 1e3 / n.value, 2)}
    kernels, timed_large = [], {}
    if not a.no_prof:
        dev_lib.lasso_prof_enable(ctx, 0)
        timed_large = {k: family(k, True) for k in STREAM}
        dev_lib.lasso_prof_reset(ctx); dev_lib.lasso_prof_enable(ctx, (1 << _abi.K_COUNT) - 1)
        hp.prove(dense, gens, S, r)                      # extra untimed step, every launch of every family bracketed
        dev_lib.lasso_prof_enable(ctx, 0)
        kernels = [f for f in (family(k, False) for k in range(_abi.K_COUNT)) if f]
    elapsed = grp.max_over_ranks(elapsed)
    digests = grp.gather_digests(proof)

    if rank == 0:
        ms_per_step = elapsed / a.steps * 1e3
        value = (1 if slab else world) * s * a.steps / elapsed
        out = {"metric": "prover lookups/sec for SparsePolynomialEvaluationProof, 2^24 AND lookups" if (a.kind, a.log_s, c) == ("and", 24, 1) else f"prover lookups/sec for SparsePolynomialEvaluationProof, 2^{a.log_s} {a.kind.upper()} lookups",
               "value": value, "unit": "lookups/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_per_step,
               "higher_is_better": True, "scaling": "strong" if slab else "weak",
               "vs_baseline": (value / PUBLISHED_LOOKUPS_PER_S) if (a.kind, a.log_s, c, a.curve) == ("and", 24, 1, "curve25519") else None, "dtype": "u256 (Montgomery Fr / BN254 Fq integers)" if a.curve == "bn254" else "u256 (Montgomery Fr / ed25519 Fq integers)", "data": "synthetic",
               "config": {"workload": f"{a.kind.upper()} subtable, C={c}, M=2^{log_m}, s=2^{a.log_s} lookups per proof, G={'BN254 G1 (ark_bn254)' if a.curve == 'bn254' else 'curve25519 (ark_curve25519)'}, harness inputs of src/benches/bench.rs; "
                                      f"timed = SparsePolynomialEvaluationProof::prove with the densified representation resident in HBM",
                          "vs_baseline_reference": "2^24 AND lookups, C=1, SparsePoly.prove 35.3 s with rayon on an Apple M1 16 GB (reference's src/benches/m1_16gb_parallel_benches.log:439; BASELINE.md §1)",
                          "per_rank": ("one proof sharded over the ranks by low index bits (slab mode)" if slab else "one independent proof per rank") if world > 1 else "single proof",
                          "proof_bytes": len(proof), "distinct_proofs": len(set(digests)), "densify_s": round(t_densify, 3), "commit_s": round(t_commit, 3), "commit_warm_s": round(t_commit_warm, 3), "gens_setup_s": round(t_setup, 3),
                          "whole_bench_lookups_per_s": s / (t_densify + t_commit + ms_per_step / 1e3)}}
        if kernels:
            out["kernels_one_profiled_step"] = kernels
            out["large_launches_timed"] = {v["kernel"]: {"per_step": v["launches"] // a.steps, "alg_bytes_per_launch": round(v["alg_GB"] * 1e9 / v["launches"])} for v in timed_large.values() if v}
            traffic = pmc_traffic()
            def roof(kid):
                """HBM roofline of one kernel family from the launches bracketed inside the timed region (the HBM-bound regime)."""
                k = timed_large.get(kid)
                if not k:
                    return None
                ach = k["alg_GB"] / (k["ms"] * 1e-3)
                allk = next((x for x in kernels if x["kernel"] == k["kernel"]), None)
                tr = traffic.get(k["kernel"])
                return {"kernel": k["kernel"], "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                        "traffic": tr["bytes_per_launch"] if tr else None, "alg_bytes_per_launch": round(k["alg_GB"] * 1e9 / k["launches"]),
                        "launches": k["launches"], "avg_launch_us": k["avg_launch_us"],
                        "scope": "launches with >= 256 MiB algorithmic bytes, HIP events inside the timed region",
                        "traffic_source": tr["source"] if tr else None,
                        "all_launches_one_profiled_step": {"achieved": allk["alg_GBps"], "launches": allk["launches"], "avg_launch_us": allk["avg_launch_us"]} if allk else None}
            stream = [k for k in kernels if k["kernel"] in [_abi.KERNEL_NAMES[i] for i in STREAM]]
            dom = max(stream, key=lambda k: k["ms"]) if stream else None                      # dominant streaming kernel family by total time
            if dom:
                out["roofline"] = roof(_abi.KERNEL_NAMES.index(dom["kernel"]))
            out["roofline_bind_top"] = roof(_abi.K_BIND)                                        # the kernel BASELINE.json's north_star names (bound_poly_var)
        if world == 1 and a.concurrent > 1 and a.concurrent * s * alpha * 450 < 150e9:   # ~400 bytes of HBM per lookup and memory per resident proof
            out["concurrent_proofs"] = concurrent_leg(HostProver, _abi, a.concurrent, max(2, a.steps), S, c, log_m, a.log_s, a.curve)
        if world == 1 and not a.no_cpu_baseline:
            cb = cpu_baseline(kind_id, c, log_m, S.log_r, min(a.cpu_log_s, a.log_s), a.curve)
            if cb:
                out["cpu_baseline"] = cb
        print(json.dumps(out))
    hp.free(dense, gens)
    hp.close()
    grp.close()


if __name__ == "__main__":
    main()
