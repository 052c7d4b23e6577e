"""ctypes access to the host prover (include/lasso_prover.h), either the product library (liblasso_prover.so, HIP backend)
or the test-only build of the same host sources against the oracle's mock of the device ABI."""
import ctypes as C
import os
import subprocess

import numpy as np

from lasso_amd import _abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_mock_prover():
    out_dir = os.path.join(ROOT, "tests", "_build")
    os.makedirs(out_dir, exist_ok=True)
    so = os.path.join(out_dir, "liblasso_prover_mock.so")
    srcs = [os.path.join(ROOT, "lasso_amd", "host", f) for f in ("prover_capi.cpp", "prover.hpp", "field_host.hpp", "hashes.hpp")]
    srcs += [os.path.join(ROOT, "lasso_amd", "csrc", f) for f in ("fr.cuh", "fq.cuh")]
    srcs += [os.path.join(ROOT, "oracle", f) for f in ("mock_hip.cpp", "lasso_oracle.hpp", "ff.hpp", "ed25519.hpp", "hashes.hpp")]
    srcs += [os.path.join(ROOT, "include", f) for f in ("lasso_hip.h", "lasso_prover.h")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wno-unknown-pragmas", "-o", so,
                               os.path.join(ROOT, "lasso_amd", "host", "prover_capi.cpp"), os.path.join(ROOT, "oracle", "mock_hip.cpp")])
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
