#!/usr/bin/env python3
"""Where the HOST spends its share of a proof: LASSO_TRACE=2 time buckets (prover.hpp HostClock) of the headline instance, third proof of three.
Usage (GPU box): LASSO_TRACE=2 python tools/host_buckets.py [kind c log_s log_m] [--curve bn254]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lasso_amd import HostProver, _abi  # noqa: E402

pos = [a for a in sys.argv[1:] if not a.startswith("--") and a not in ("bn254", "curve25519")]
kind, c, log_s, log_m = (pos + ["and", "1", "24", "16"][len(pos):])[:4]
c, log_s, log_m = int(c), int(log_s), int(log_m)
curve = "bn254" if "bn254" in sys.argv else "curve25519"
alpha = 2 * c if kind == "lt" else c
hp = HostProver(device=0, curve=curve)
S = _abi.Strategy(_abi.KINDS[kind], c, log_m, 40 if kind == "range" else 0)
s = 1 << log_s
idx = hp.gen_indices(s, 1 << log_m, c); r = hp.gen_random_point(log_s)
gens = hp.gens(c, s, alpha, log_m); dense = hp.densify(idx, log_m); del idx
for i in range(3):
    print(f"--- proof {i}", file=sys.stderr)
    t0 = time.perf_counter(); hp.prove(dense, gens, S, r); print(f"[host] whole proof {1e3 * (time.perf_counter() - t0):.3f} ms", file=sys.stderr)
