"""CPU: capacity mode (lasso_host_set_capacity / LASSO_CAPACITY=1) — the read / write product trees kept WITHOUT their leaf layers, the bottom layer's two streaming
sumcheck rounds recomputing the fingerprints chunk by chunk (Prover::leaf_round) — produces the commitment and proof bytes of the ordinary prover and of the oracle.
Host prover over the oracle's mock of the device ABI; each configuration in its own process (the switches are read once per process):
  * LASSO_LEAFLESS_MIN=64 LASSO_CUBIC_TAIL=0: the chunked rounds at toy sizes (no resident tail, so rounds 0 and 1 of the bottom layer stream);
  * LASSO_LEAFLESS_MIN=64 alone: the bottom layer is short enough for the resident tail, so the leaves are materialised after all (the fallback of cubic_rounds);
  * one proof over P = 2, 4 ranks (tests/cpp/slab_threads.cpp) in capacity mode;
  * LASSO_CAPACITY_COMPACT=0: leafless trees over the field-element dim / read (round 4's first form of the mode).
In every other case the representation is COMPACT (dim / read held as 4-byte integers, lifted one polynomial at a time where field elements are needed): asserted through
lasso_host_dense_info, whose byte count is checked against the layout."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import ctypes as C, hashlib, os, sys
import numpy as np
sys.path.insert(0, os.path.join(%(root)r, "tests"))
from lasso_amd import _abi
from proverutil import HostProver, OracleSession, build_mock_prover
kind, c, log_m, log_r, lookups, world = %(kind)r, %(c)d, %(log_m)d, %(log_r)d, %(lookups)d, %(world)d
s = 1 << (lookups - 1).bit_length(); alpha = 2 * c if kind == "lt" else c
lib = C.CDLL(build_mock_prover())
hp = HostProver(lib)
idx = np.ascontiguousarray(np.random.default_rng(lookups * 31 + c).integers(0, 1 << log_m, size=(lookups, c), dtype=np.uint64))
r = np.ascontiguousarray(hp.gen_random_point(s.bit_length() - 1), dtype=np.uint64)
S = _abi.Strategy(_abi.KINDS[kind], c, log_m, log_r)
if world == 1:
    hp.set_capacity(True)
    gens = hp.gens(c, s, alpha, log_m); dense = hp.densify(idx, log_m)
    info = hp.dense_info(dense)
    want_compact = os.environ.get("LASSO_CAPACITY_COMPACT", "1") != "0"
    assert info["compact"] == want_compact, info     # dim / read as 4-byte integers: no field-element copy of the merged operations polynomial
    n_l = 1 << (2 * c * s - 1).bit_length(); n_m = (1 << (c - 1).bit_length()) << log_m
    assert info["device_bytes"] == (32 * n_m + 4 * 2 * c * s if want_compact else 32 * (n_l + n_m) + 4 * c * s), info
    comm = hp.commit(dense, gens); proof = hp.prove(dense, gens, S, r)
    assert proof == hp.prove(dense, gens, S, r)
    hp.set_capacity(False)     # a compact representation stays provable after the mode is switched off (its trees stay leafless)
    assert proof == hp.prove(dense, gens, S, r)
else:
    import test_slab_sharding_cpu as T
    slab = T.build_slab_lib()
    cb = (C.c_uint8 * (1 << 20))(); pb = (C.c_uint8 * (1 << 22))(); cl = C.c_size_t(); pl = C.c_size_t(); nc = C.c_size_t(); nb = C.c_size_t(); err = C.create_string_buffer(512)
    rc = slab.slab_prove_threads_ex(world, C.byref(S), C.c_size_t(alpha), idx.ctypes.data_as(C.c_void_p), C.c_size_t(lookups), r.ctypes.data_as(C.c_void_p), C.c_size_t(r.shape[0]),
                                    None, 1, 2, cb, C.c_size_t(len(cb)), C.byref(cl), pb, C.c_size_t(len(pb)), C.byref(pl), C.byref(nc), C.byref(nb), None, None, None, err, C.c_size_t(512))
    assert rc == 0, err.value.decode()
    comm, proof = bytes(cb[: cl.value]), bytes(pb[: pl.value])
oracle = C.CDLL(os.path.join(%(root)r, "oracle", "liblasso_oracle.so"))
oracle.orc_last_error.restype = C.c_char_p
for f in ("orc_session_new",): getattr(oracle, f).restype = C.c_void_p
orc = OracleSession(oracle, _abi.KINDS[kind], c, log_m, log_r, idx, r)
assert comm == orc.commit(), "commitment differs from the oracle's"
assert proof == orc.prove(), "proof differs from the oracle's"
print("OK", hashlib.sha256(proof).hexdigest())
"""

CASES = [("and", 1, 8, 0, 1 << 9, 1), ("xor", 2, 6, 0, 300, 1), ("lt", 2, 6, 0, 1 << 8, 1), ("range", 3, 8, 40, 1 << 8, 1), ("and", 2, 8, 0, 1 << 10, 2), ("range", 2, 8, 12, 1 << 10, 4),
         ("spark", 2, 6, 0, 1 << 8, 1)]


@pytest.mark.parametrize("compact", ["compact dim/read", "field-element dim/read"])
@pytest.mark.parametrize("tails", ["streaming bottom layer (chunked leaf rounds)", "resident tail (leaves materialised after all)"])
@pytest.mark.parametrize("kind,c,log_m,log_r,lookups,world", CASES)
def test_capacity_mode_is_byte_identical(oracle, kind, c, log_m, log_r, lookups, world, tails, compact):
    if compact.startswith("field") and not (tails.startswith("streaming") and (kind, c) in (("and", 1), ("lt", 2), ("and", 2))):
        pytest.skip("the old form is covered on three cases")
    env = dict(os.environ, LASSO_CAPACITY="1", LASSO_LEAFLESS_MIN="64", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), OMP_NUM_THREADS="2")
    if tails.startswith("streaming"):
        env["LASSO_CUBIC_TAIL"] = "0"
    if compact == "field-element dim/read":
        env["LASSO_CAPACITY_COMPACT"] = "0"
    res = subprocess.run([sys.executable, "-c", SCRIPT % {"root": ROOT, "kind": kind, "c": c, "log_m": log_m, "log_r": log_r, "lookups": lookups, "world": world}],
                         env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and "OK" in res.stdout, res.stdout[-1500:] + res.stderr[-3000:]
