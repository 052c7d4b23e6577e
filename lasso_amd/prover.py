"""Python view of the host prover (include/lasso_prover.h): the reference's three calls — DensifiedRepresentation::from_lookup_indices,
DensifiedRepresentation::commit, SparsePolynomialEvaluationProof::prove (src/benches/bench.rs:54-66) — over liblasso_prover.so.
No CPU fallback: the default library is the HIP-backed one and loading fails loudly if it is missing."""
import ctypes as C
import os

import numpy as np

from . import _abi
from .device import LassoError

HERE = os.path.dirname(os.path.abspath(__file__))


def load_prover_library(path=None, curve="curve25519"):
    """curve = "bn254": the BN254 pair (liblasso_prover_bn254.so over liblasso_hip_bn254.so); both pairs can live in one process."""
    # LASSO_PROVER_LIB: an explicit path to another build of the SAME C ABI (the CPU tests of bench.py's multi-rank plumbing point it at the host sources linked
    # against the test mock of the device ABI).  Never set by the package itself; when unset the HIP-backed library is the only candidate and its absence is fatal.
    path = path or os.environ.get("LASSO_PROVER_LIB") or os.path.join(HERE, "liblasso_prover_bn254.so" if curve == "bn254" else "liblasso_prover.so")
    if not os.path.exists(path):
        raise LassoError(f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` (no CPU fallback exists)")
    return C.CDLL(path)


ALLGATHER_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)   # (user, send, recv, bytes per rank) -> 0
# include/lasso_prover.h lasso_transcript_vtbl: merlin::Transcript as append_message / challenge_bytes callbacks (labels are pointer + length)
APPEND_FN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_uint8), C.c_size_t, C.POINTER(C.c_uint8), C.c_size_t)
CHALLENGE_FN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_uint8), C.c_size_t, C.POINTER(C.c_uint8), C.c_size_t)


class TranscriptVtbl(C.Structure):
    _fields_ = [("append_message", APPEND_FN), ("challenge_bytes", CHALLENGE_FN)]


def declare_prover(lib):
    vp, sz, i32 = C.c_void_p, C.c_size_t, C.c_int32
    lib.lasso_host_last_error.restype = C.c_char_p
    lib.lasso_host_create.argtypes = [i32, C.POINTER(vp)]
    lib.lasso_host_destroy.argtypes = [vp]
    lib.lasso_host_ctx.argtypes = [vp]; lib.lasso_host_ctx.restype = vp
    u64p = C.POINTER(C.c_uint64)
    lib.lasso_host_mem_stats.argtypes = [vp, u64p, u64p, u64p, i32]
    lib.lasso_host_set_capacity.argtypes = [vp, i32]
    lib.lasso_host_set_throughput_mode.argtypes = [vp, i32]
    lib.lasso_host_set_comm.argtypes = [vp, i32, i32, ALLGATHER_FN, vp]
    lib.lasso_host_set_comm_shm.argtypes = [vp, i32, i32, C.c_char_p]
    lib.lasso_host_gens_new.argtypes = [vp, C.c_char_p, sz, sz, sz, sz, C.POINTER(vp)]
    lib.lasso_host_gens_from_points.argtypes = [vp, sz, sz, sz, sz, vp, sz, vp, sz, vp, sz, C.POINTER(vp)]
    lib.lasso_host_gens_prepare.argtypes = [vp]
    lib.lasso_host_gens_points.argtypes = [vp, i32, vp, sz, C.POINTER(sz)]
    lib.lasso_host_gens_free.argtypes = [vp]
    lib.lasso_host_densify.argtypes = [vp, vp, sz, sz, sz, C.POINTER(vp)]
    lib.lasso_host_dense_free.argtypes = [vp]
    lib.lasso_host_dense_info.argtypes = [vp, u64p, C.POINTER(C.c_int32)]
    lib.lasso_host_commit.argtypes = [vp, vp, vp, sz, C.POINTER(sz)]
    lib.lasso_host_prove.argtypes = [vp, vp, vp, C.POINTER(_abi.Strategy), vp, sz, C.c_char_p, C.c_char_p, vp, sz, C.POINTER(sz)]
    lib.lasso_host_verify.argtypes = [vp, vp, C.POINTER(_abi.Strategy), sz, vp, sz, C.c_char_p, C.c_char_p, sz, C.c_char_p, sz, C.POINTER(i32)]
    lib.lasso_host_debug_cubic_batched.argtypes = [vp, vp, vp, C.POINTER(_abi.Strategy), sz, sz, vp, vp, vp, vp, vp, C.c_char_p, vp, sz, C.POINTER(sz)]
    vt = C.POINTER(TranscriptVtbl)
    lib.lasso_host_prove_cb.argtypes = [vp, vp, vp, C.POINTER(_abi.Strategy), vp, sz, vt, vp, vt, vp, vp, sz, C.POINTER(sz)]
    lib.lasso_host_verify_cb.argtypes = [vp, vp, C.POINTER(_abi.Strategy), sz, vp, sz, vt, vp, C.c_char_p, sz, C.c_char_p, sz, C.POINTER(i32)]
    lib.lasso_host_merlin_new.argtypes = [C.c_char_p]; lib.lasso_host_merlin_new.restype = vp
    lib.lasso_host_random_tape_new.argtypes = [C.c_char_p]; lib.lasso_host_random_tape_new.restype = vp
    lib.lasso_host_merlin_free.argtypes = [vp]
    lib.lasso_host_merlin_vtbl.argtypes = []; lib.lasso_host_merlin_vtbl.restype = vt
    lib.lasso_host_gen_indices.argtypes = [sz, sz, vp]
    lib.lasso_host_gen_random_point.argtypes = [sz, vp]
    return lib


class HostProver:
    """One lasso_host (device context + host prover). `lib` defaults to the product library; tests may inject the mock-backed build."""

    def __init__(self, lib=None, device=0, curve="curve25519"):
        self.lib = declare_prover(lib or load_prover_library(curve=curve))
        h = C.c_void_p()
        if self.lib.lasso_host_create(device, C.byref(h)) != 0:
            raise LassoError("lasso_host_create: " + self.lib.lasso_host_last_error().decode())
        self.h = h

    def set_comm(self, group):
        """Slab mode (one proof sharded over the ranks of `group`, lasso_amd.parallel.Group): must precede gens()/densify()."""
        self._allgather = ALLGATHER_FN(group.allgather_callback())       # keep the callback object alive
        self._chk(self.lib.lasso_host_set_comm(self.h, group.rank, group.world, self._allgather, None))

    def set_comm_shm(self, rank, world, name):
        """Slab mode over the library's own shared-memory exchange (one node): `name` = "/something", the same on every rank."""
        self._chk(self.lib.lasso_host_set_comm_shm(self.h, rank, world, name.encode()))

    def _chk(self, rc):
        if rc != 0:
            raise LassoError(f"lasso_host error {rc}: " + self.lib.lasso_host_last_error().decode())

    def gen_indices(self, s, m, c):
        one = np.empty(s, dtype=np.uint64)
        self.lib.lasso_host_gen_indices(s, m, one.ctypes.data_as(C.c_void_p))
        return np.repeat(one[:, None], c, axis=1).copy()     # `[x; C]`: one draw per lookup, replicated (benches/bench.rs:17)

    def gen_random_point(self, bits):
        r = np.empty((max(bits, 1), 4), dtype=np.uint64)
        self.lib.lasso_host_gen_random_point(bits, r.ctypes.data_as(C.c_void_p))
        return r[:bits]

    def gens(self, c, s, num_memories, log_m, label=b"gens_sparse_poly"):
        g = C.c_void_p()
        self._chk(self.lib.lasso_host_gens_new(self.h, label, c, s, num_memories, log_m, C.byref(g)))
        return g

    def gens_prepare(self, gens):
        """build every device table a proof over `gens` reads now instead of inside the first commit / prove (lasso_host_gens_prepare)"""
        self._chk(self.lib.lasso_host_gens_prepare(gens))

    def gens_from_points(self, c, s, num_memories, log_m, l_variate, log_m_variate, derefs):
        """The caller's generators (surge.rs:119-125 `gens: &SparsePolyCommitmentGens<G>`): three arrays of shape (n + 2, 8) uint64 — affine (x, y) Montgomery limbs in the
        order G[0..n), gens_1.G[0], h (lasso_host_gens_from_points)."""
        sets = [np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 8) for a in (l_variate, log_m_variate, derefs)]
        g = C.c_void_p()
        args = []
        for a in sets:
            args += [a.ctypes.data_as(C.c_void_p), a.shape[0]]
        self._chk(self.lib.lasso_host_gens_from_points(self.h, c, s, num_memories, log_m, *args, C.byref(g)))
        return g

    def gens_points(self, gens, which):
        """the points of generator set `which` (0 l-variate, 1 log_m-variate, 2 derefs) as an (n + 2, 8) uint64 array (lasso_host_gens_points)"""
        n = C.c_size_t()
        self.lib.lasso_host_gens_points(gens, which, None, 0, C.byref(n))
        out = np.empty((n.value, 8), dtype=np.uint64)
        self._chk(self.lib.lasso_host_gens_points(gens, which, out.ctypes.data_as(C.c_void_p), n.value, C.byref(n)))
        return out

    def densify(self, indices, log_m):
        indices = np.ascontiguousarray(indices, dtype=np.uint64)
        d = C.c_void_p()
        self._chk(self.lib.lasso_host_densify(self.h, indices.ctypes.data_as(C.c_void_p), indices.shape[0], indices.shape[1], log_m, C.byref(d)))
        return d

    def dense_info(self, dense):
        """device bytes a densified representation holds, and whether dim / read are in capacity mode's compact form"""
        b, c = C.c_uint64(), C.c_int32()
        self._chk(self.lib.lasso_host_dense_info(dense, C.byref(b), C.byref(c)))
        return {"device_bytes": b.value, "compact": bool(c.value)}

    _cap = 1 << 22      # output buffer of the byte-returning calls: a proof is 0.15 MB (C = 1) to 1.3 MB (C = 16); grows, and stays grown, if a call reports more

    def _bytes_call(self, fn, *args):
        while True:
            cap = self._cap
            buf = (C.c_uint8 * cap)()
            n = C.c_size_t()
            rc = fn(*args, buf, cap, C.byref(n))
            if rc == -2 and n.value > cap:      # the call ran to the end before it knew (a proof is proved to be measured): never pay that twice for one shape
                self._cap = 2 * n.value
                continue
            self._chk(rc)
            return bytes(buf[: n.value])

    def commit(self, dense, gens):
        return self._bytes_call(self.lib.lasso_host_commit, dense, gens)

    def prove(self, dense, gens, strategy, r, transcript=b"example", tape=b"proof"):
        r = np.ascontiguousarray(r, dtype=np.uint64).reshape(-1, 4)
        return self._bytes_call(self.lib.lasso_host_prove, self.h, dense, gens, C.byref(strategy), r.ctypes.data_as(C.c_void_p), r.shape[0], transcript, tape)

    def prove_with(self, dense, gens, strategy, r, transcript, tape):
        """SparsePolynomialEvaluationProof::prove against LIVE transcripts (surge.rs:119-125 `&mut Transcript`, `&mut RandomTape`): `transcript` / `tape` are
        (vtbl pointer, user pointer) pairs — lasso_host_prove_cb; see Transcript below for the library's own Merlin behind that interface"""
        r = np.ascontiguousarray(r, dtype=np.uint64).reshape(-1, 4)
        return self._bytes_call(lambda *a: self.lib.lasso_host_prove_cb(self.h, dense, gens, C.byref(strategy), r.ctypes.data_as(C.c_void_p), r.shape[0], transcript[0], transcript[1], tape[0], tape[1], *a))

    def verify_with(self, gens, strategy, s, r, proof, commitment, transcript):
        r = np.ascontiguousarray(r, dtype=np.uint64).reshape(-1, 4)
        ok = C.c_int32(-1)
        self._chk(self.lib.lasso_host_verify_cb(self.h, gens, C.byref(strategy), s, r.ctypes.data_as(C.c_void_p), r.shape[0], transcript[0], transcript[1], proof, len(proof), commitment, len(commitment), C.byref(ok)))
        return ok.value == 1

    def verify(self, gens, strategy, s, r, proof, commitment, transcript=b"example"):
        """SparsePolynomialEvaluationProof::verify (surge.rs:214-271) over the wire bytes: True = Ok(()), False = Err(ProofVerifyError); raises LassoError on bytes
        that do not deserialize or on shapes the reference would assert on."""
        r = np.ascontiguousarray(r, dtype=np.uint64).reshape(-1, 4)
        ok = C.c_int32(-1)
        self._chk(self.lib.lasso_host_verify(self.h, gens, C.byref(strategy), s, r.ctypes.data_as(C.c_void_p), r.shape[0], transcript, proof, len(proof), commitment, len(commitment), C.byref(ok)))
        return ok.value == 1

    def debug_cubic_batched(self, dense, gens, strategy, A, B, rand, coeffs, claim, transcript=b"test"):
        """test support: prove_cubic_batched on caller arrays with a scripted eq point (lasso_host_debug_cubic_batched)"""
        A = np.ascontiguousarray(A, dtype=np.uint64); B = np.ascontiguousarray(B, dtype=np.uint64)     # (k, 2^ell, 4)
        k, n = A.shape[0], A.shape[1]; ell = n.bit_length() - 1
        rand = np.ascontiguousarray(rand, dtype=np.uint64).reshape(-1, 4); coeffs = np.ascontiguousarray(coeffs, dtype=np.uint64).reshape(-1, 4)
        claim = np.ascontiguousarray(claim, dtype=np.uint64).reshape(4)
        assert rand.shape[0] == ell and coeffs.shape[0] == k and B.shape == A.shape
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        return self._bytes_call(self.lib.lasso_host_debug_cubic_batched, self.h, dense, gens, C.byref(strategy), k, ell, vp(A), vp(B), vp(rand), vp(coeffs), vp(claim), transcript)

    def free(self, dense=None, gens=None):
        if dense:
            self.lib.lasso_host_dense_free(dense)
        if gens:
            self.lib.lasso_host_gens_free(gens)

    def close(self):
        if self.h:
            self.lib.lasso_host_destroy(self.h)
            self.h = None

    def ctx(self):
        return self.lib.lasso_host_ctx(self.h)

    def mem_stats(self, reset=False):
        """device bytes held by this host now / at most, and the most the prover itself had in use (lasso_host_mem_stats)"""
        live, peak, used = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self._chk(self.lib.lasso_host_mem_stats(self.h, C.byref(live), C.byref(peak), C.byref(used), 1 if reset else 0))
        return {"live_bytes": live.value, "peak_bytes": peak.value, "prover_peak_bytes": used.value}

    def set_throughput_mode(self, on=True):
        """this host is one of several proving concurrently on the GPU: no kernel of it waits on the device for its host thread (lasso_host_set_throughput_mode)"""
        self._chk(self.lib.lasso_host_set_throughput_mode(self.h, 1 if on else 0))

    def set_capacity(self, on=True):
        """capacity mode (lasso_host_set_capacity): large buffers go back to the driver on release; the per-rank high-water mark is the live peak"""
        self._chk(self.lib.lasso_host_set_capacity(self.h, 1 if on else 0))


class Transcript:
    """The library's own Merlin transcript as an object that lives across calls (lasso_host_merlin_new / lasso_host_random_tape_new): `pair()` is what
    HostProver.prove_with / verify_with take.  tape=True builds RandomTape::new(label) (utils/random.rs:15-31) instead of Transcript::new(label)."""

    def __init__(self, lib, label, tape=False):
        self.lib = lib
        self.m = (lib.lasso_host_random_tape_new if tape else lib.lasso_host_merlin_new)(label)
        if not self.m:
            raise LassoError("lasso_host_merlin_new failed")
        self.vt = lib.lasso_host_merlin_vtbl()

    def append_message(self, label, msg):
        lb = (C.c_uint8 * len(label)).from_buffer_copy(label); mb = (C.c_uint8 * max(len(msg), 1)).from_buffer_copy(msg or b"\0")
        self.vt.contents.append_message(self.m, lb, len(label), mb, len(msg))

    def challenge_bytes(self, label, n):
        lb = (C.c_uint8 * len(label)).from_buffer_copy(label); out = (C.c_uint8 * n)()
        self.vt.contents.challenge_bytes(self.m, lb, len(label), out, n)
        return bytes(out)

    def pair(self):
        return (self.vt, C.c_void_p(self.m))

    def close(self):
        if self.m:
            self.lib.lasso_host_merlin_free(self.m); self.m = None
