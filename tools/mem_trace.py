#!/usr/bin/env python3
"""Where a proof's device bytes are: LASSO_TRACE=3 prints, per prover span, the bytes live at entry, the span's own high-water mark and the bytes live at exit
(lasso_mem_stats of the main context).  Usage (GPU box): LASSO_TRACE=3 [LASSO_CAPACITY=1] python tools/mem_trace.py [kind c log_s log_m]   — one proof on one context."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lasso_amd import HostProver, _abi  # noqa: E402

kind, c, log_s, log_m = (sys.argv[1:] + ["range", "4", "26", "16"][len(sys.argv) - 1:])[:4]
c, log_s, log_m = int(c), int(log_s), int(log_m)
alpha = 2 * c if kind == "lt" else c
hp = HostProver(device=0)
S = _abi.Strategy(_abi.KINDS[kind], c, log_m, 40 if kind == "range" else 0)
s = 1 << log_s
idx = hp.gen_indices(s, 1 << log_m, c); r = hp.gen_random_point(log_s)
gens = hp.gens(c, s, alpha, log_m)
print("after gens:", hp.mem_stats(reset=True), file=sys.stderr)
dense = hp.densify(idx, log_m); del idx
print("dense:", hp.dense_info(dense), file=sys.stderr)
comm = hp.commit(dense, gens)
for i in range(2):
    print(f"--- proof {i}", file=sys.stderr)
    proof = hp.prove(dense, gens, S, r)
print("end:", hp.mem_stats(), file=sys.stderr)
