"""CPU: the host prover's buffer policy, observed through lasso_mem_stats of the mock device (oracle/mock_hip.cpp counts lasso_alloc'd bytes and calls exactly as the device
library does for hipMalloc):
  * repeated proofs reach a steady state that makes NO device allocation, pooled and in capacity mode (round 4: a small pool miss that evicted a parked gigabyte started a
    miss / evict cycle of hipFree / hipMalloc that repeated every proof, 650 ms at configs[3] over two ranks);
  * capacity mode's high-water mark is well below the pooled one and does not grow from proof to proof (the second proof's buffers once sat on top of the first one's);
  * the proofs are the same bytes in every mode."""
import ctypes as C
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.join(%(root)r, "tests"))
from lasso_amd import _abi
from proverutil import HostProver, build_mock_prover
lib = C.CDLL(build_mock_prover()); lib.mock_alloc_calls.restype = C.c_uint64; lib.mock_alloc_calls.argtypes = [C.c_void_p]
kind, c, log_m, lookups = %(kind)r, %(c)d, 8, 1 << 11
S = _abi.Strategy(_abi.KINDS[kind], c, log_m, 0)
out = {}
for mode in ("pooled", "capacity"):
    hp = HostProver(lib)
    if mode == "capacity": hp.set_capacity(True)
    idx = np.ascontiguousarray(np.random.default_rng(5).integers(0, 1 << log_m, size=(lookups, c), dtype=np.uint64))
    r = np.ascontiguousarray(hp.gen_random_point(11), dtype=np.uint64)
    gens = hp.gens(c, lookups, 2 * c if kind == "lt" else c, log_m); dense = hp.densify(idx, log_m)
    hp.commit(dense, gens)
    proofs, peaks, calls = [], [], []
    for i in range(4):
        hp.mem_stats(reset=True); n0 = lib.mock_alloc_calls(hp.ctx())
        proofs.append(hp.prove(dense, gens, S, r))
        peaks.append(hp.mem_stats()["peak_bytes"]); calls.append(lib.mock_alloc_calls(hp.ctx()) - n0)
    out[mode] = (proofs, peaks, calls)
    assert all(p == proofs[0] for p in proofs), mode
    assert calls[2] == 0 and calls[3] == 0, (mode, calls)          # steady state: every buffer comes from the pool
    assert peaks[3] == peaks[2] == peaks[1], (mode, peaks)          # ... and the high-water mark does not creep
assert out["pooled"][0][0] == out["capacity"][0][0]
ratio = out["capacity"][1][3] / out["pooled"][1][3]
assert ratio < 0.75, ratio                                          # leafless trees + compact dim / read + no eq / chi tables while the trees stand
print("OK", out["pooled"][1][3], out["capacity"][1][3], round(ratio, 3), out["pooled"][2], out["capacity"][2])
"""


@pytest.mark.parametrize("kind,c", [("and", 2), ("lt", 2), ("range", 3), ("range", 4), ("xor", 1), ("spark", 2), ("spark", 4)])
def test_steady_state_allocates_nothing_and_capacity_mode_holds_less(oracle, kind, c):
    env = dict(os.environ, LASSO_LEAFLESS_MIN="64", LASSO_CUBIC_TAIL="0", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), OMP_NUM_THREADS="2")
    env.pop("LASSO_CAPACITY", None)
    res = subprocess.run([sys.executable, "-c", SCRIPT % {"root": ROOT, "kind": kind, "c": c}], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and "OK" in res.stdout, res.stdout[-1500:] + res.stderr[-3000:]


LEAK_SCRIPT = r"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.join(%(root)r, "tests"))
from lasso_amd import _abi
from proverutil import HostProver, build_mock_prover
lib = C.CDLL(build_mock_prover())
hp = HostProver(lib)
for kind, c in (("and", 2), ("lt", 1), ("spark", 2)):
    S = _abi.Strategy(_abi.KINDS[kind], c, 8, 0)
    idx = np.ascontiguousarray(np.random.default_rng(9).integers(0, 256, size=(1 << 10, c), dtype=np.uint64))
    r = np.ascontiguousarray(hp.gen_random_point(10), dtype=np.uint64)
    for cap in (False, True):
        hp.set_capacity(cap)
        gens = hp.gens(c, 1 << 10, 2 * c if kind == "lt" else c, 8); dense = hp.densify(idx, 8)
        comm = hp.commit(dense, gens); proof = hp.prove(dense, gens, S, r)
        assert hp.verify(gens, S, 1 << 10, r, proof, comm)
        hp.free(dense, gens)
        hp.set_capacity(True)          # trims the recycling pool
        live = hp.mem_stats()["live_bytes"]
        assert live == 0, (kind, c, cap, live)      # every buffer the representation, the prover and the verifier took went back
print("OK")
"""


def test_nothing_is_left_allocated_after_a_proof(oracle):
    """densify + commit + prove + verify + free, pooled and capacity: once the pool is trimmed the mock device holds no byte of this host"""
    env = dict(os.environ, LASSO_LEAFLESS_MIN="64", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), OMP_NUM_THREADS="2")
    env.pop("LASSO_CAPACITY", None)
    res = subprocess.run([sys.executable, "-c", LEAK_SCRIPT % {"root": ROOT}], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and "OK" in res.stdout, res.stdout[-1500:] + res.stderr[-3000:]
