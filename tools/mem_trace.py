#!/usr/bin/env python3
"""One proof on ONE context with the device bytes it needs: per prover span under LASSO_TRACE=3 (bytes live at entry, the span's own high-water mark, bytes live at exit —
lasso_mem_stats of the main context), per proof otherwise (time, peak, verifier's answer) as one JSON line.
Usage (GPU box): [LASSO_TRACE=3] [LASSO_CAPACITY=1] python tools/mem_trace.py [kind c log_s log_m] [--verify] [--steps K]"""
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lasso_amd import HostProver, _abi  # noqa: E402

pos = [a for a in sys.argv[1:] if not a.startswith("--")]
kind, c, log_s, log_m = (pos + ["range", "4", "26", "16"][len(pos):])[:4]
steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 2
c, log_s, log_m = int(c), int(log_s), int(log_m)
alpha = 2 * c if kind == "lt" else c
hp = HostProver(device=0)
S = _abi.Strategy(_abi.KINDS[kind], c, log_m, 40 if kind == "range" else 0)
s = 1 << log_s
idx = hp.gen_indices(s, 1 << log_m, c); r = hp.gen_random_point(log_s)
gens = hp.gens(c, s, alpha, log_m)
print("after gens:", hp.mem_stats(reset=True), file=sys.stderr)
t0 = time.perf_counter(); dense = hp.densify(idx, log_m); t_densify = time.perf_counter() - t0; del idx
info = hp.dense_info(dense)
print("dense:", info, file=sys.stderr)
t0 = time.perf_counter(); comm = hp.commit(dense, gens); t_commit = time.perf_counter() - t0
times = []
for i in range(steps):
    print(f"--- proof {i}", file=sys.stderr)
    t0 = time.perf_counter(); proof = hp.prove(dense, gens, S, r); times.append(time.perf_counter() - t0)
st = hp.mem_stats()
print("end:", st, file=sys.stderr)
out = {"workload": f"{kind.upper()} C={c} M=2^{log_m} s=2^{log_s}, one context of one MI355X", "capacity_mode": os.environ.get("LASSO_CAPACITY") == "1", "compact_dim_read": info["compact"],
       "densify_s": round(t_densify, 3), "commit_s": round(t_commit, 3), "prove_s": [round(t, 4) for t in times], "lookups_per_s": round(s / min(times)),
       "peak_bytes": st["peak_bytes"], "prover_peak_bytes": st["prover_peak_bytes"], "proof_bytes": len(proof), "proof_sha256": hashlib.sha256(proof).hexdigest()}
if "--verify" in sys.argv:
    t0 = time.perf_counter(); out["verifier_accepts"] = bool(hp.verify(gens, S, s, r, proof, comm)); out["verify_s"] = round(time.perf_counter() - t0, 3)
    bad = bytearray(proof); bad[len(bad) // 2] ^= 1
    try:
        out["verifier_rejects_flipped_bit"] = not hp.verify(gens, S, s, r, bytes(bad), comm)
    except Exception:
        out["verifier_rejects_flipped_bit"] = True
print(json.dumps(out))
