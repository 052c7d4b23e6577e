"""Thin ctypes wrapper over liblasso_hip.so (include/lasso_hip.h).  numpy arrays carry field elements as (n, 4) uint64
Montgomery limbs — the same bytes as ark-ff's Fp256 — and device buffers are raw device pointers."""
import ctypes as C
import os

import numpy as np

from . import _abi

HERE = os.path.dirname(os.path.abspath(__file__))


class LassoError(RuntimeError):
    pass


def load_device_library(path=None, curve="curve25519"):
    """curve = "bn254" loads the BN254 build of the same kernels (liblasso_hip_bn254.so: same ABI over ark-bn254's Fr / G1)."""
    path = path or os.path.join(HERE, "liblasso_hip_bn254.so" if curve == "bn254" else "liblasso_hip.so")
    if not os.path.exists(path):
        raise LassoError(f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` (no CPU fallback exists)")
    lib = C.CDLL(path)
    _abi.declare(lib)
    return lib


def _vp(a):
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data_as(C.c_void_p)
    return C.c_void_p(a)


class Device:
    """One lasso_ctx.  `lib` may be injected (tests inject the oracle's mock ABI); the default is the HIP library."""

    def __init__(self, device=0, lib=None, curve="curve25519"):
        self.lib = lib or load_device_library(curve=curve)
        ctx = C.c_void_p()
        rc = self.lib.lasso_ctx_create(device, C.byref(ctx))
        if rc != 0:
            raise LassoError(f"lasso_ctx_create failed ({rc}): {self.lib.lasso_last_error(None).decode()}")
        self.ctx = ctx

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.lasso_ctx_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise LassoError(f"lasso error {rc}: {self.lib.lasso_last_error(self.ctx).decode()}")

    # ---- memory
    def alloc(self, nbytes):
        p = C.c_void_p()
        self._chk(self.lib.lasso_alloc(self.ctx, nbytes, C.byref(p)))
        return p.value

    def free(self, ptr):
        self._chk(self.lib.lasso_free(self.ctx, C.c_void_p(ptr)))

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        p = self.alloc(arr.nbytes)
        self._chk(self.lib.lasso_upload(self.ctx, C.c_void_p(p), _vp(arr), arr.nbytes))
        return p

    def download(self, ptr, shape, dtype=np.uint64):
        out = np.empty(shape, dtype=dtype)
        self._chk(self.lib.lasso_download(self.ctx, _vp(out), C.c_void_p(ptr), out.nbytes))
        return out

    def sync(self):
        self._chk(self.lib.lasso_sync(self.ctx))

    def _ptrs(self, ptrs):
        return (C.c_void_p * len(ptrs))(*ptrs)

    # ---- kernels (argument meaning: include/lasso_hip.h)
    def fr_from_u32(self, d_src, n, d_dst):
        self._chk(self.lib.lasso_fr_from_u32(self.ctx, C.c_void_p(d_src), n, C.c_void_p(d_dst)))

    def fr_to_u32(self, d_src, n, d_dst):
        """the integers behind n field elements; raises if one does not fit 32 bits; returns the largest"""
        mx = C.c_uint32(0)
        self._chk(self.lib.lasso_fr_to_u32(self.ctx, C.c_void_p(d_src), n, C.c_void_p(d_dst), C.byref(mx)))
        return mx.value

    def gather(self, d_table, d_idx, n, d_out):
        self._chk(self.lib.lasso_gather(self.ctx, C.c_void_p(d_table), C.c_void_p(d_idx), n, C.c_void_p(d_out)))

    def eq_evals(self, r, d_out):
        r = np.ascontiguousarray(r, dtype=np.uint64).reshape(-1, 4)
        self._chk(self.lib.lasso_eq_evals(self.ctx, _vp(r), r.shape[0], C.c_void_p(d_out)))

    def bind_top(self, ptrs, n, r):
        r = np.ascontiguousarray(r, dtype=np.uint64)
        self._chk(self.lib.lasso_bind_top(self.ctx, self._ptrs(ptrs), len(ptrs), n, _vp(r)))

    def sumcheck_cubic_round(self, a_ptrs, b_ptrs, d_c, n):
        out = np.empty((len(a_ptrs) * 3, 4), dtype=np.uint64)
        self._chk(self.lib.lasso_sumcheck_cubic_round(self.ctx, self._ptrs(a_ptrs), self._ptrs(b_ptrs), len(a_ptrs), C.c_void_p(d_c), n, _vp(out)))
        return out

    def sumcheck_linear_eqw_round(self, ptrs, d_e, n):
        out = np.empty((3 * len(ptrs), 4), dtype=np.uint64)
        self._chk(self.lib.lasso_sumcheck_linear_eqw_round(self.ctx, self._ptrs(ptrs), len(ptrs), C.c_void_p(d_e), n, _vp(out)))
        return out.reshape(len(ptrs), 3, 4)[:, :2].copy()

    def sumcheck_linear_eqw_round_fused(self, ptrs, d_e, n, r):
        r = np.ascontiguousarray(r, dtype=np.uint64)
        out = np.empty((3 * len(ptrs), 4), dtype=np.uint64)
        self._chk(self.lib.lasso_sumcheck_linear_eqw_round_fused(self.ctx, self._ptrs(ptrs), len(ptrs), C.c_void_p(d_e), n, _vp(r), _vp(out)))
        return out.reshape(len(ptrs), 3, 4)[:, :2].copy()

    def sumcheck_linear_eqw_round_fused_from(self, src_ptrs, dst_ptrs, d_e, n, r):
        r = np.ascontiguousarray(r, dtype=np.uint64)
        out = np.empty((3 * len(dst_ptrs), 4), dtype=np.uint64)
        self._chk(self.lib.lasso_sumcheck_linear_eqw_round_fused_from(self.ctx, self._ptrs(src_ptrs), self._ptrs(dst_ptrs), len(dst_ptrs), C.c_void_p(d_e), n, _vp(r), _vp(out)))
        return out.reshape(len(dst_ptrs), 3, 4)[:, :2].copy()

    def sumcheck_cubic_eqw_round(self, a_ptrs, b_ptrs, d_e, n):
        out = np.empty((3 * len(a_ptrs), 4), dtype=np.uint64)
        self._chk(self.lib.lasso_sumcheck_cubic_eqw_round(self.ctx, self._ptrs(a_ptrs), self._ptrs(b_ptrs), len(a_ptrs), C.c_void_p(d_e), n, _vp(out)))
        return out

    def sumcheck_cubic_eqw_round_fused(self, a_ptrs, b_ptrs, d_e, n, r):
        r = np.ascontiguousarray(r, dtype=np.uint64)
        out = np.empty((3 * len(a_ptrs), 4), dtype=np.uint64)
        self._chk(self.lib.lasso_sumcheck_cubic_eqw_round_fused(self.ctx, self._ptrs(a_ptrs), self._ptrs(b_ptrs), len(a_ptrs), C.c_void_p(d_e), n, _vp(r), _vp(out)))
        return out

    def sumcheck_cubic_eqw2(self, a_ptrs, b_ptrs, d_e, n, r=None):
        """two-sum eq-weighted round (lasso_sumcheck_cubic_eqw2_begin + lasso_result_wait): rows (q(0), q_inf) per circuit"""
        rp = None if r is None else _vp(np.ascontiguousarray(r, dtype=np.uint64))
        out = np.empty((2 * len(a_ptrs), 4), dtype=np.uint64)
        self._chk(self.lib.lasso_sumcheck_cubic_eqw2_begin(self.ctx, self._ptrs(a_ptrs), self._ptrs(b_ptrs), len(a_ptrs), C.c_void_p(d_e), n, rp))
        self._chk(self.lib.lasso_result_wait(self.ctx, _vp(out), 2 * len(a_ptrs)))
        return out

    def sumcheck_cubic_eqw2_eq(self, a_ptrs, b_ptrs, d_e_out, n, point, scale=None):
        """first round of a layer with the eq table (scale * eq(point), n/2 entries) built inside the launch and left in d_e_out (lasso_sumcheck_cubic_eqw2_begin_eq)"""
        point = np.ascontiguousarray(point, dtype=np.uint64).reshape(-1, 4)
        sp = None if scale is None else _vp(np.ascontiguousarray(scale, dtype=np.uint64))
        out = np.empty((2 * len(a_ptrs), 4), dtype=np.uint64)
        self._chk(self.lib.lasso_sumcheck_cubic_eqw2_begin_eq(self.ctx, self._ptrs(a_ptrs), self._ptrs(b_ptrs), len(a_ptrs), C.c_void_p(d_e_out), n, _vp(point), point.shape[0], sp))
        self._chk(self.lib.lasso_result_wait(self.ctx, _vp(out), 2 * len(a_ptrs)))
        return out

    def sumcheck_cubic_tail_eq(self, a_ptrs, b_ptrs, n, point, scale, challenges):
        """resident tail from a layer's first round on, eq table derived in the kernel (lasso_sumcheck_cubic_tail_begin_eq)"""
        k = len(a_ptrs)
        point = np.ascontiguousarray(point, dtype=np.uint64).reshape(-1, 4)
        sp = None if scale is None else _vp(np.ascontiguousarray(scale, dtype=np.uint64))
        outs = []
        self._chk(self.lib.lasso_sumcheck_cubic_tail_begin_eq(self.ctx, self._ptrs(a_ptrs), self._ptrs(b_ptrs), k, n, _vp(point) if point.shape[0] else None, point.shape[0], sp))
        out = np.empty((2 * k, 4), dtype=np.uint64); self._chk(self.lib.lasso_result_wait(self.ctx, _vp(out), 2 * k)); outs.append(out)
        for ch in challenges:
            ch = np.ascontiguousarray(ch, dtype=np.uint64)
            self._chk(self.lib.lasso_sumcheck_cubic_tail_next(self.ctx, _vp(ch)))
            out = np.empty((2 * k, 4), dtype=np.uint64); self._chk(self.lib.lasso_result_wait(self.ctx, _vp(out), 2 * k)); outs.append(out)
        return outs

    def eq_evals_scaled(self, r, scale, d_out):
        r = np.ascontiguousarray(r, dtype=np.uint64).reshape(-1, 4)
        sp = None if scale is None else _vp(np.ascontiguousarray(scale, dtype=np.uint64))
        self._chk(self.lib.lasso_eq_evals_scaled(self.ctx, _vp(r) if r.shape[0] else None, r.shape[0], sp, C.c_void_p(d_out)))

    def sumcheck_cubic_tail(self, a_ptrs, b_ptrs, d_e, n, r, challenges):
        """resident tail kernel (lasso_sumcheck_cubic_tail_begin / _next): returns the list of per-round (2k, 4) results, the last entry being the heads"""
        k = len(a_ptrs)
        rp = None if r is None else _vp(np.ascontiguousarray(r, dtype=np.uint64))
        outs = []
        self._chk(self.lib.lasso_sumcheck_cubic_tail_begin(self.ctx, self._ptrs(a_ptrs), self._ptrs(b_ptrs), k, C.c_void_p(d_e), n, rp))
        out = np.empty((2 * k, 4), dtype=np.uint64); self._chk(self.lib.lasso_result_wait(self.ctx, _vp(out), 2 * k)); outs.append(out)
        for ch in challenges:
            ch = np.ascontiguousarray(ch, dtype=np.uint64)
            self._chk(self.lib.lasso_sumcheck_cubic_tail_next(self.ctx, _vp(ch)))
            out = np.empty((2 * k, 4), dtype=np.uint64); self._chk(self.lib.lasso_result_wait(self.ctx, _vp(out), 2 * k)); outs.append(out)
        return outs

    def sumcheck_linear_tail(self, ptrs, d_e, n, r, challenges):
        """resident tail of the primary sumcheck (lasso_sumcheck_linear_tail_begin + lasso_sumcheck_cubic_tail_next): per-round (alpha, 2, 4) sums, then the (alpha, 4) heads"""
        k = len(ptrs)
        rp = None if r is None else _vp(np.ascontiguousarray(r, dtype=np.uint64))
        outs = []
        self._chk(self.lib.lasso_sumcheck_linear_tail_begin(self.ctx, self._ptrs(ptrs), k, C.c_void_p(d_e), n, rp))
        out = np.empty((2 * k, 4), dtype=np.uint64); self._chk(self.lib.lasso_result_wait(self.ctx, _vp(out), 2 * k)); outs.append(out.reshape(k, 2, 4))
        for t, ch in enumerate(challenges):
            ch = np.ascontiguousarray(ch, dtype=np.uint64)
            self._chk(self.lib.lasso_sumcheck_cubic_tail_next(self.ctx, _vp(ch)))
            last = t == len(challenges) - 1
            out = np.empty(((1 if last else 2) * k, 4), dtype=np.uint64); self._chk(self.lib.lasso_result_wait(self.ctx, _vp(out), out.shape[0]))
            outs.append(out if last else out.reshape(k, 2, 4))
        return outs

    def sumcheck_combine_round(self, strategy, ptrs, d_eq, n, degree):
        out = np.empty((degree + 1, 4), dtype=np.uint64)
        self._chk(self.lib.lasso_sumcheck_combine_round(self.ctx, C.byref(strategy), self._ptrs(ptrs), C.c_void_p(d_eq), n, degree, _vp(out)))
        return out

    def lt_prescale(self, strategy, ptrs, n, src=None):
        """LT_m <- 32^-(C-1-m) LT_m (the form lasso_sumcheck_combine_round_lt_scaled takes); src: read from there instead (all 2C polynomials land in ptrs)"""
        self._chk(self.lib.lasso_lt_prescale(self.ctx, C.byref(strategy), None if src is None else self._ptrs(src), self._ptrs(ptrs), n))

    def sumcheck_combine_round_lt_scaled(self, strategy, ptrs, d_eq, n, degree):
        out = np.empty((degree + 1, 4), dtype=np.uint64)
        self._chk(self.lib.lasso_sumcheck_combine_round_lt_scaled(self.ctx, C.byref(strategy), self._ptrs(ptrs), C.c_void_p(d_eq), n, degree, _vp(out)))
        return out

    def sumcheck_linear_eqw_round_u32(self, u32_ptrs, d_e, n):
        out = np.empty((3 * len(u32_ptrs), 4), dtype=np.uint64)
        self._chk(self.lib.lasso_sumcheck_linear_eqw_round_u32(self.ctx, self._ptrs(u32_ptrs), len(u32_ptrs), C.c_void_p(d_e), n, _vp(out)))
        return out.reshape(len(u32_ptrs), 3, 4)[:, :2].copy()

    def sumcheck_linear_eqw_round_fused_from_u32(self, u32_ptrs, ptrs, d_e, n, r):
        r = np.ascontiguousarray(r, dtype=np.uint64)
        out = np.empty((3 * len(ptrs), 4), dtype=np.uint64)
        self._chk(self.lib.lasso_sumcheck_linear_eqw_round_fused_from_u32(self.ctx, self._ptrs(u32_ptrs), self._ptrs(ptrs), len(ptrs), C.c_void_p(d_e), n, _vp(r), _vp(out)))
        return out.reshape(len(ptrs), 3, 4)[:, :2].copy()

    def sumcheck_combine_round_lt_u32(self, strategy, u32_ptrs, d_eq, n, degree):
        out = np.empty((degree + 1, 4), dtype=np.uint64)
        self._chk(self.lib.lasso_sumcheck_combine_round_lt_u32(self.ctx, C.byref(strategy), self._ptrs(u32_ptrs), C.c_void_p(d_eq), n, degree, _vp(out)))
        return out

    def combine_claim(self, strategy, ptrs, d_eq, n):
        out = np.empty((1, 4), dtype=np.uint64)
        self._chk(self.lib.lasso_combine_claim(self.ctx, C.byref(strategy), self._ptrs(ptrs), C.c_void_p(d_eq), n, _vp(out)))
        return out

    def multi_dot(self, ptrs, d_w, n):
        out = np.empty((len(ptrs), 4), dtype=np.uint64)
        self._chk(self.lib.lasso_multi_dot(self.ctx, self._ptrs(ptrs), len(ptrs), C.c_void_p(d_w), n, _vp(out)))
        return out

    def read_heads(self, ptrs):
        out = np.empty((len(ptrs), 4), dtype=np.uint64)
        self._chk(self.lib.lasso_read_heads(self.ctx, self._ptrs(ptrs), len(ptrs), _vp(out)))
        return out

    def gp_build(self, d_tree, n):
        self._chk(self.lib.lasso_gp_build(self.ctx, C.c_void_p(d_tree), n))

    def fingerprint_ops(self, d_table, d_dim, d_read, s, gamma, tau, d_ro, d_wo):
        g = np.ascontiguousarray(gamma, dtype=np.uint64); t = np.ascontiguousarray(tau, dtype=np.uint64)
        self._chk(self.lib.lasso_fingerprint_ops(self.ctx, C.c_void_p(d_table), C.c_void_p(d_dim), C.c_void_p(d_read), s, _vp(g), _vp(t), C.c_void_p(d_ro), C.c_void_p(d_wo)))

    def fingerprint_ops_gp(self, d_table, d_dim, d_read, s, gamma, tau, d_tree_r, d_tree_w):
        g = np.ascontiguousarray(gamma, dtype=np.uint64); t = np.ascontiguousarray(tau, dtype=np.uint64)
        self._chk(self.lib.lasso_fingerprint_ops_gp(self.ctx, C.c_void_p(d_table), C.c_void_p(d_dim), C.c_void_p(d_read), s, _vp(g), _vp(t), C.c_void_p(d_tree_r), C.c_void_p(d_tree_w)))

    def fingerprint_ops_gp_upper(self, d_table, d_dim, d_read, s, gamma, tau, d_upper_r, d_upper_w, read_u32=False):
        """capacity mode: both trees without their leaf layers (s - 2 elements each); read_u32: d_read holds 32-bit integers"""
        g = np.ascontiguousarray(gamma, dtype=np.uint64); t = np.ascontiguousarray(tau, dtype=np.uint64)
        fn = self.lib.lasso_fingerprint_ops_gp_upper_u32 if read_u32 else self.lib.lasso_fingerprint_ops_gp_upper
        self._chk(fn(self.ctx, C.c_void_p(d_table), C.c_void_p(d_dim), C.c_void_p(d_read), s, _vp(g), _vp(t), C.c_void_p(d_upper_r), C.c_void_p(d_upper_w)))

    def fingerprint_ops_strips(self, d_table, d_dim, d_read, s, gamma, tau, nstrips, i0, cs, d_out_r, d_out_w, read_u32=False):
        """capacity mode: the leaves of one strip set of the bottom layer (2 * nstrips * cs elements per circuit)"""
        g = np.ascontiguousarray(gamma, dtype=np.uint64); t = np.ascontiguousarray(tau, dtype=np.uint64)
        fn = self.lib.lasso_fingerprint_ops_strips_u32 if read_u32 else self.lib.lasso_fingerprint_ops_strips
        self._chk(fn(self.ctx, C.c_void_p(d_table), C.c_void_p(d_dim), C.c_void_p(d_read), s, _vp(g), _vp(t), nstrips, i0, cs, C.c_void_p(d_out_r), C.c_void_p(d_out_w)))

    def fingerprint_mem(self, d_table, d_final, m, gamma, tau, d_io, d_fo):
        g = np.ascontiguousarray(gamma, dtype=np.uint64); t = np.ascontiguousarray(tau, dtype=np.uint64)
        self._chk(self.lib.lasso_fingerprint_mem(self.ctx, C.c_void_p(d_table), C.c_void_p(d_final), m, _vp(g), _vp(t), C.c_void_p(d_io), C.c_void_p(d_fo)))

    def matvec_left(self, d_z, L, l_size, r_size):
        L = np.ascontiguousarray(L, dtype=np.uint64)
        out = np.empty((r_size, 4), dtype=np.uint64)
        self._chk(self.lib.lasso_matvec_left(self.ctx, C.c_void_p(d_z), _vp(L), l_size, r_size, _vp(out)))
        return out

    def bases_create(self, affine):
        affine = np.ascontiguousarray(affine, dtype=np.uint64).reshape(-1, 8)
        b = C.c_void_p()
        self._chk(self.lib.lasso_bases_create(self.ctx, _vp(affine), affine.shape[0], C.byref(b)))
        return b.value

    def bases_destroy(self, b):
        self.lib.lasso_bases_destroy(self.ctx, C.c_void_p(b))

    def hyrax_commit(self, d_z, l_size, r_size, bases):
        out = np.empty((l_size, 16), dtype=np.uint64)
        self._chk(self.lib.lasso_hyrax_commit(self.ctx, C.c_void_p(d_z), l_size, r_size, C.c_void_p(bases), _vp(out)))
        return out

    def hyrax_commit_compressed(self, d_z, l_size, r_size, bases):
        out = np.empty((l_size, 32), dtype=np.uint8)
        self._chk(self.lib.lasso_hyrax_commit_compressed(self.ctx, C.c_void_p(d_z), l_size, r_size, C.c_void_p(bases), _vp(out)))
        return out

    def hyrax_commit_compressed_u32(self, d_u32, max_value, l_size, r_size, bases):
        out = np.empty((l_size, 32), dtype=np.uint8)
        self._chk(self.lib.lasso_hyrax_commit_compressed_u32(self.ctx, C.c_void_p(d_u32), max_value, l_size, r_size, C.c_void_p(bases), _vp(out)))
        return out

    def materialize_subtable_u32(self, strategy, sub):
        m = 1 << strategy.log_m
        p = self.alloc(4 * m)
        self._chk(self.lib.lasso_materialize_subtable_u32(self.ctx, C.byref(strategy), sub, C.c_void_p(p)))
        out = self.download(p, (m,), dtype=np.uint32)
        self.free(p)
        return out

    def gather_u32(self, d_table, d_idx, n, d_out):
        self._chk(self.lib.lasso_gather_u32(self.ctx, C.c_void_p(d_table), C.c_void_p(d_idx), n, C.c_void_p(d_out)))

    def msm(self, bases, scalars):
        scalars = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
        out = np.empty((1, 16), dtype=np.uint64)
        self._chk(self.lib.lasso_msm(self.ctx, C.c_void_p(bases), _vp(scalars), scalars.shape[0], _vp(out)))
        return out

    def msm_dev(self, bases, d_scalars, n):
        out = np.empty((1, 16), dtype=np.uint64)
        self._chk(self.lib.lasso_msm_dev(self.ctx, C.c_void_p(bases), C.c_void_p(d_scalars), n, _vp(out)))
        return out

    def matvec_left_dev(self, d_z, d_l, l_size, r_size, d_out):
        self._chk(self.lib.lasso_matvec_left_dev(self.ctx, C.c_void_p(d_z), C.c_void_p(d_l), l_size, r_size, C.c_void_p(d_out)))

    def fr_to_bytes(self, d_src, n):
        out = np.empty((n, 32), dtype=np.uint8)
        self._chk(self.lib.lasso_fr_to_bytes(self.ctx, C.c_void_p(d_src), n, _vp(out)))
        return out

    def msm_dev_scaled(self, bases, d_scalars, n, scale, tail):
        scale = np.ascontiguousarray(scale, dtype=np.uint64); tail = np.ascontiguousarray(tail, dtype=np.uint64).reshape(2, 4)
        out = np.empty((1, 16), dtype=np.uint64)
        self._chk(self.lib.lasso_msm_dev_scaled(self.ctx, C.c_void_p(bases), C.c_void_p(d_scalars), n, _vp(scale), _vp(tail), _vp(out)))
        return out

    def inner_products_lr(self, d_a, d_b, nk):
        out = np.empty((2, 4), dtype=np.uint64)
        self._chk(self.lib.lasso_inner_products_lr(self.ctx, C.c_void_p(d_a), C.c_void_p(d_b), nk, _vp(out)))
        return out

    def bullet_lr(self, bases, n, d_a, nk, d_w, tail):
        tail = np.ascontiguousarray(tail, dtype=np.uint64).reshape(4, 4)
        out = np.empty((2, 16), dtype=np.uint64)
        self._chk(self.lib.lasso_bullet_lr(self.ctx, C.c_void_p(bases), n, C.c_void_p(d_a), nk, C.c_void_p(d_w), _vp(tail), _vp(out)))
        return out

    def densify_dim(self, d_indices, n_lookups, c, dim, s, log_m, d_dim_u32, d_dim, d_read, d_final):
        self._chk(self.lib.lasso_densify_dim(self.ctx, C.c_void_p(d_indices), n_lookups, c, dim, s, log_m, C.c_void_p(d_dim_u32), C.c_void_p(d_dim), C.c_void_p(d_read), C.c_void_p(d_final)))

    def bullet_round(self, bases, n, d_a_in, d_b_in, d_w_in, d_a_out, d_b_out, d_w_out, nk, u, u_inv, blinds):
        blinds = np.ascontiguousarray(blinds, dtype=np.uint64).reshape(2, 4)
        out = np.empty((2, 16), dtype=np.uint64)
        if u is None:
            pu = pui = None
        else:
            u = np.ascontiguousarray(u, dtype=np.uint64); ui = np.ascontiguousarray(u_inv, dtype=np.uint64)
            pu, pui = _vp(u), _vp(ui)
        self._chk(self.lib.lasso_bullet_round(self.ctx, C.c_void_p(bases), n, C.c_void_p(d_a_in), C.c_void_p(d_b_in), C.c_void_p(d_w_in), C.c_void_p(d_a_out), C.c_void_p(d_b_out),
                                              C.c_void_p(d_w_out), nk, pu, pui, _vp(blinds), _vp(out)))
        return out

    def bullet_fold(self, d_a, d_b, nk, d_w, nw, d_w_out, u, u_inv):
        u = np.ascontiguousarray(u, dtype=np.uint64); ui = np.ascontiguousarray(u_inv, dtype=np.uint64)
        self._chk(self.lib.lasso_bullet_fold(self.ctx, C.c_void_p(d_a), C.c_void_p(d_b), nk, C.c_void_p(d_w), nw, C.c_void_p(d_w_out), _vp(u), _vp(ui)))

    # ---- profiling
    def prof_enable(self, mask=0x3FF):
        self._chk(self.lib.lasso_prof_enable(self.ctx, int(mask)))

    def prof_reset(self):
        self._chk(self.lib.lasso_prof_reset(self.ctx))

    def prof_get(self, kid):
        n = C.c_uint64(); ms = C.c_double(); b = C.c_double()
        self._chk(self.lib.lasso_prof_get(self.ctx, kid, C.byref(n), C.byref(ms), C.byref(b)))
        return n.value, ms.value, b.value
