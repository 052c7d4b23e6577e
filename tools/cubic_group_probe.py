"""Does the cubic round lose bandwidth because 16 circuits stream at once?  One lasso_sumcheck_cubic_eqw2 round over k circuits in one launch against the same
work as k/G launches of G circuits each (GPU box; python tools/cubic_group_probe.py)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lasso_amd.device import Device
d = Device()
log_n, k = 23, 16
n = 1 << log_n
rng = np.random.default_rng(1)
base = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64); base[:, 3] &= (1 << 59) - 1
A = [d.upload(np.roll(base, i + 1, axis=0)) for i in range(k)]; B = [d.upload(np.roll(base, 100 + i, axis=0)) for i in range(k)]
E = d.upload(base[: n // 2])
def run(groups, fused_r=None):
    t0 = time.perf_counter(); outs = []
    for g in groups:
        outs.append(d.sumcheck_cubic_eqw2([A[c] for c in g], [B[c] for c in g], E, n, fused_r))
    return (time.perf_counter() - t0) * 1e3, np.concatenate(outs)
for G in (16, 8, 4, 2):
    groups = [list(range(c, c + G)) for c in range(0, k, G)]
    run(groups)
    ts = [run(groups)[0] for _ in range(5)]
    alg = 32.0 * (2 * k + 1) * n
    print(f"round 0 (read only): {k} circuits as {len(groups)} launches of {G}: {min(ts):.3f} ms  ({alg / min(ts) / 1e6:.0f} GB/s algorithmic)")
ref = run([list(range(k))])[1]
assert np.array_equal(ref, run([list(range(c, c + 4)) for c in range(0, k, 4)])[1])
print("grouped results identical")
