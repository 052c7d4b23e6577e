"""ctypes access to the host prover (include/lasso_prover.h), either the product library (liblasso_prover.so, HIP backend)
or the test-only build of the same host sources against the oracle's mock of the device ABI."""
import ctypes as C
import os
import subprocess

import numpy as np

from lasso_amd import _abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_mock_prover(curve="curve25519"):
    """The host prover sources linked against the oracle's mock of the device ABI; curve = "bn254" builds both with -DLASSO_BN254 / -DORC_BN254."""
    out_dir = os.path.join(ROOT, "tests", "_build")
    os.makedirs(out_dir, exist_ok=True)
    bn = curve == "bn254"
    so = os.path.join(out_dir, "liblasso_prover_mock_bn254.so" if bn else "liblasso_prover_mock.so")
    srcs = [os.path.join(ROOT, "lasso_amd", "host", f) for f in ("prover_capi.cpp", "prover.hpp", "field_host.hpp", "hashes.hpp", "modinv.hpp")]
    srcs += [os.path.join(ROOT, "lasso_amd", "csrc", f) for f in ("fr.cuh", "fq.cuh", "bn254_fr.cuh", "bn254_fq.cuh")]
    srcs += [os.path.join(ROOT, "oracle", f) for f in ("mock_hip.cpp", "lasso_oracle.hpp", "ff.hpp", "ed25519.hpp", "bn254.hpp", "hashes.hpp")]
    srcs += [os.path.join(ROOT, "include", f) for f in ("lasso_hip.h", "lasso_prover.h")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        flags = ["-DLASSO_BN254", "-DORC_BN254"] if bn else []
        tmp = f"{so}.{os.getpid()}.tmp"    # several test processes (pytest -n) may find the library stale at once: each links its own file, the rename is atomic
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wno-unknown-pragmas", "-fno-gnu-unique", "-Wl,-Bsymbolic", *flags, "-o", tmp,   # see oracle/Makefile
                               os.path.join(ROOT, "lasso_amd", "host", "prover_capi.cpp"), os.path.join(ROOT, "oracle", "mock_hip.cpp")])
        os.replace(tmp, so)
    return so


from lasso_amd.prover import HostProver, declare_prover  # noqa: E402,F401


class OracleSession:
    """The CPU oracle's statement of the same instance (oracle/oracle_capi.cpp)."""

    def __init__(self, oracle, kind, c, log_m, log_r, indices, r):
        self.o = oracle
        indices = np.ascontiguousarray(indices, dtype=np.uint64)
        r = np.ascontiguousarray(r, dtype=np.uint64)
        oracle.orc_session_new.argtypes = [C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
        self.s = oracle.orc_session_new(kind, c, 1 << log_m, log_r, indices.ctypes.data_as(C.c_void_p), indices.shape[0], r.ctypes.data_as(C.c_void_p))
        if not self.s:
            raise RuntimeError(oracle.orc_last_error().decode())

    def _bytes(self, fn, cap=1 << 22):
        buf = (C.c_uint8 * cap)(); n = C.c_size_t()
        rc = fn(C.c_void_p(self.s), buf, C.c_size_t(cap), C.byref(n))
        if rc != 0:
            raise RuntimeError(self.o.orc_last_error().decode())
        return bytes(buf[: n.value])

    def commit(self):
        return self._bytes(self.o.orc_session_commit)

    def prove(self):
        return self._bytes(self.o.orc_session_prove)

    def verify(self, proof, commitment=None):
        if commitment is None:
            return self.o.orc_session_verify(C.c_void_p(self.s), proof, C.c_size_t(len(proof)))
        return self.o.orc_session_verify_with_commitment(C.c_void_p(self.s), proof, C.c_size_t(len(proof)), commitment, C.c_size_t(len(commitment)))

    def close(self):
        if self.s:
            self.o.orc_session_free(C.c_void_p(self.s)); self.s = None


def oracle_harness_proof(oracle, kind, c, log_m, log_r, log_s, threads=0, verify=False):
    """The oracle's densify -> commit -> prove on the reference harness's inputs (benches/bench.rs:13-34: gen_indices / gen_random_point from a fresh
    test_rng) on `threads` OpenMP threads (0 = one per physical core; bytes do not depend on it, tests/test_oracle_parallel.py).
    Returns (commitment bytes, proof bytes, {"densify_s", "commit_s", "prove_s", "threads"})."""
    oracle.orc_set_threads(threads)      # 0 = one thread per physical core
    td, tc, tp = C.c_double(), C.c_double(), C.c_double()
    cap = 1 << 23
    pb = (C.c_uint8 * cap)(); cb = (C.c_uint8 * cap)(); pl = C.c_size_t(); cl = C.c_size_t()
    rc = oracle.orc_bench_bytes(kind, C.c_size_t(c), C.c_size_t(1 << log_m), C.c_size_t(log_r), C.c_size_t(1 << log_s), C.byref(td), C.byref(tc), C.byref(tp),
                                1 if verify else 0, pb, C.c_size_t(cap), C.byref(pl), cb, C.c_size_t(cap), C.byref(cl))
    if rc != 0:
        raise RuntimeError(oracle.orc_last_error().decode())
    return bytes(cb[: cl.value]), bytes(pb[: pl.value]), {"densify_s": td.value, "commit_s": tc.value, "prove_s": tp.value, "threads": oracle.orc_max_threads()}


def cubic_batched_case(host, oracle, k, ell, special, seed):
    """prove_cubic_batched with a scripted eq point: the host prover's eq-weighted two-sum rounds (device ABI) against the oracle's literal
    three-polynomial loop (sumcheck.rs:27-135).  `special` maps round -> 0 or 1: rand_t = 0 disables the claim-derived evaluation (three-sum
    fallback), rand_t = 1 makes the eq table's prefix vanish (explicit tables)."""
    from gpuutil import rand_fr
    from fieldref import L as FR_P, limbs, to_mont
    rng = np.random.default_rng(seed)
    n = 1 << ell
    A = np.stack([rand_fr(rng, n) for _ in range(k)]); B = np.stack([rand_fr(rng, n) for _ in range(k)])
    rand = rand_fr(rng, max(ell, 1), edge=False)[:ell]
    for t, v in special.items():
        if t < ell:
            rand[t] = np.array(limbs(to_mont(v, FR_P)), dtype=np.uint64)
    coeffs = rand_fr(rng, k, edge=False)
    oracle.orc_cubic_batched.argtypes = [C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    cap = 1 << 16; buf = (C.c_uint8 * cap)(); ln = C.c_size_t()
    vp = lambda a: np.ascontiguousarray(a).ctypes.data_as(C.c_void_p)
    Ac, Bc, rc_, cc = (np.ascontiguousarray(x) for x in (A, B, rand, coeffs))
    rc = oracle.orc_cubic_batched(k, ell, vp(Ac), vp(Bc), vp(rc_), vp(cc), b"test", buf, cap, C.byref(ln))
    assert rc == 0, oracle.orc_last_error().decode()
    want = bytes(buf[: ln.value])
    from fieldref import to_mont as tm
    claim_int = int.from_bytes(want[:32], "little")
    claim = np.array(limbs(tm(claim_int, FR_P)), dtype=np.uint64)
    S = _abi.Strategy(_abi.KINDS["and"], 1, 4, 0)
    gens = host.gens(1, 4, 1, 4)
    dense = host.densify(np.zeros((4, 1), dtype=np.uint64), 4)
    try:
        got = host.debug_cubic_batched(dense, gens, S, A, B, rand, coeffs, claim)
    finally:
        host.free(dense, gens)
    assert got == want[32:]
