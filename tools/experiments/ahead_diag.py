import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lasso_amd import HostProver, _abi
hp = HostProver(device=0)
S = _abi.Strategy(_abi.KINDS["and"], 1, 16, 0)
s = 1 << 10
idx = hp.gen_indices(s, 1 << 16, 1); r = hp.gen_random_point(10)
gens = hp.gens(1, s, 1, 16); dense = hp.densify(idx, 16)
t0 = time.perf_counter()
try:
    p = hp.prove(dense, gens, S, r); print("proof ok", len(p), "in", round(time.perf_counter() - t0, 3), "s")
except Exception as e:
    print("FAILED after", round(time.perf_counter() - t0, 3), "s:", e)
