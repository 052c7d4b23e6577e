"""-m gpu: every entry point of liblasso_hip.so (include/lasso_hip.h) against the oracle's CPU statement of the same
call (oracle/mock_hip.cpp), bit-exact, on seeded inputs incl. the edge sizes the reference exercises (n = 2, ragged grids)."""
import ctypes as C
import os

import numpy as np
import pytest

from fieldref import L as FR_P, limbs, to_mont
from gpuutil import compress_points, gens, load_mock, rand_fr, small_fr
from lasso_amd import _abi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def devs():
    from lasso_amd import Device
    from fieldref import CURVE
    real = Device(0, curve=CURVE)     # raises loudly if the HIP library / GPU is missing (LASSO_TEST_CURVE=bn254: the BN254 build, tools/gpu.sh tests_bn254)
    mock = Device(0, lib=load_mock())
    yield real, mock
    real.close(); mock.close()


def canon(rows):
    """canonical representatives of Montgomery-form rows (any 256-bit value of the same residue -> the value in [0, p))"""
    flat = rows.reshape(-1, 4)
    out = np.empty_like(flat)
    for i, row in enumerate(flat):
        v = (int(row[0]) | int(row[1]) << 64 | int(row[2]) << 128 | int(row[3]) << 192) % FR_P
        out[i] = [(v >> (64 * k)) & (2**64 - 1) for k in range(4)]
    return out.reshape(rows.shape)


def both(devs, fn):
    return fn(devs[0]), fn(devs[1])


@pytest.mark.parametrize("n", [2, 4, 64, 1 << 10, 1 << 12, 1 << 17])
@pytest.mark.parametrize("npolys", [1, 3])
def test_bind_top(devs, n, npolys):
    rng = np.random.default_rng(n * 7 + npolys)
    data = [rand_fr(rng, n) for _ in range(npolys)]
    r = rand_fr(rng, 1, edge=False)[0]

    def run(d):
        ptrs = [d.upload(x) for x in data]
        d.bind_top(ptrs, n, r)
        out = [d.download(p, (n, 4)) for p in ptrs]
        for p in ptrs:
            d.free(p)
        return out
    a, b = both(devs, run)
    for x, y in zip(a, b):
        assert np.array_equal(x[: n // 2], y[: n // 2])
        assert np.array_equal(x[n // 2:], y[n // 2:])   # upper half untouched


def test_bind_top_chain_to_scalar(devs):
    """bind every variable: the last element standing is the MLE evaluation (dense_mlpoly.rs:435-458 style)."""
    rng = np.random.default_rng(5)
    n = 1 << 9
    z = rand_fr(rng, n)
    rs = rand_fr(rng, 9, edge=False)

    def run(d):
        p = d.upload(z)
        m = n
        for r in rs:
            d.bind_top([p], m, r)
            m //= 2
        out = d.download(p, (1, 4))
        d.free(p)
        return out
    a, b = both(devs, run)
    assert np.array_equal(a, b)


@pytest.mark.parametrize("ell", [0, 1, 5, 12, 13, 17])
def test_eq_evals(devs, ell):
    rng = np.random.default_rng(ell)
    r = rand_fr(rng, max(ell, 1), edge=False)[:ell]

    def run(d):
        p = d.alloc(32 << ell)
        d.eq_evals(r, p)
        out = d.download(p, (1 << ell, 4))
        d.free(p)
        return out
    a, b = both(devs, run)
    assert np.array_equal(a, b)


@pytest.mark.parametrize("n", [2, 8, 256, 4096])
def test_bullet_tail_chain_launched_ahead(devs, n):
    """lasso_bullet_tail_ahead + lasso_bullet_post + ONE lasso_result_wait of six values == lasso_bullet_fold (2 -> 1) + lasso_read_heads + lasso_msm_dev_scaled with the same
    challenge: the delta point, the two heads and the folded weights, on the device and in the mock; a chain that never gets its challenge is released by lasso_abort at once."""
    import time
    rng = np.random.default_rng(n * 91 + 5)
    mock_lib = devs[1].lib
    g = gens(mock_lib, b"gens_sparse_poly", n + 1)       # n + 2 points: G_0..G_{n-1}, Q, H
    nw = n // 2
    a = rand_fr(rng, 2, edge=False); b = rand_fr(rng, 2, edge=False); w = rand_fr(rng, max(nw, 1), edge=False)
    scale = rand_fr(rng, 1, edge=False); tail = rand_fr(rng, 2, edge=False)
    u, ui = rand_fr(rng, 2, edge=False)
    vp = lambda x: np.ascontiguousarray(x, dtype=np.uint64).ctypes.data_as(C.c_void_p)

    def run(d):
        bases = d.bases_create(g)
        if d.lib.lasso_bullet_tail_ahead_ok(d.ctx, bases) != 1:
            pytest.skip("lasso_bullet_tail_ahead not available in this configuration")
        pa = d.upload(a); pb = d.upload(b); pw = d.upload(w); pw2 = d.alloc(32 * n)
        # the three separate calls (they also size the context's buffers: the chain itself must not grow anything)
        d.bullet_fold(pa, pb, 2, pw, nw, pw2, u, ui)
        want_heads = d.read_heads([pa, pb]); want_w = d.download(pw2, (n, 4))
        want_pt = d.msm_dev_scaled(bases, pw2, n, scale, tail)
        for p, x in ((pa, a), (pb, b)):
            d.free(p)
        pa = d.upload(a); pb = d.upload(b)
        d._chk(d.lib.lasso_zero(d.ctx, C.c_void_p(pw2), 32 * n))
        # enqueued and abandoned
        d._chk(d.lib.lasso_bullet_tail_ahead(d.ctx, bases, n, C.c_void_p(pa), C.c_void_p(pb), C.c_void_p(pw), nw, C.c_void_p(pw2), vp(scale), vp(tail)))
        t0 = time.perf_counter()
        d._chk(d.lib.lasso_abort(d.ctx))
        assert time.perf_counter() - t0 < 1.0, "abort must not wait for a 5 s bail-out"
        d.free(pa); d.free(pb)
        pa = d.upload(a); pb = d.upload(b)
        d._chk(d.lib.lasso_zero(d.ctx, C.c_void_p(pw2), 32 * n))
        # the real thing on the same context
        d._chk(d.lib.lasso_bullet_tail_ahead(d.ctx, bases, n, C.c_void_p(pa), C.c_void_p(pb), C.c_void_p(pw), nw, C.c_void_p(pw2), vp(scale), vp(tail)))
        time.sleep(0.002)                                   # the chain is waiting on the device for these two scalars
        d._chk(d.lib.lasso_bullet_post(d.ctx, vp(u), vp(ui)))
        got = np.empty((6, 4), dtype=np.uint64)
        d._chk(d.lib.lasso_result_wait(d.ctx, got.ctypes.data_as(C.c_void_p), 6))
        got_w = d.download(pw2, (n, 4))
        for p in (pa, pb, pw, pw2):
            d.free(p)
        d.bases_destroy(bases)
        return want_pt, want_heads, want_w, got, got_w
    (pa_, ha, wa, ga, gwa), (pb_, hb, wb, gb, gwb) = both(devs, run)
    for want, got in ((pa_, ga), (pb_, gb), (pa_, gb)):
        assert compress_points(mock_lib, np.asarray(want).reshape(1, -1)) == compress_points(mock_lib, np.asarray(got[:4]).reshape(1, -1))
    assert np.array_equal(ha, ga[4:6]) and np.array_equal(hb, gb[4:6]) and np.array_equal(ha, hb)
    assert np.array_equal(wa, gwa) and np.array_equal(wb, gwb) and np.array_equal(wa, wb)


@pytest.mark.parametrize("n,ncirc", [(2, 1), (4, 2), (16, 33), (128, 5), (256, 2), (1 << 9, 2), (1 << 13, 8), (1 << 16, 3)])   # n <= 128: latency-shaped kernel
def test_sumcheck_cubic_round(devs, n, ncirc):
    rng = np.random.default_rng(n + ncirc)
    A = [rand_fr(rng, n) for _ in range(ncirc)]
    B = [rand_fr(rng, n) for _ in range(ncirc)]
    Cp = rand_fr(rng, n)

    def run(d):
        pa = [d.upload(x) for x in A]; pb = [d.upload(x) for x in B]; pc = d.upload(Cp)
        out = d.sumcheck_cubic_round(pa, pb, pc, n)
        for p in pa + pb + [pc]:
            d.free(p)
        return out
    a, b = both(devs, run)
    assert np.array_equal(a, b)


CONFIGS = [("and", 1, 16, 0), ("and", 4, 16, 0), ("xor", 8, 16, 0), ("or", 2, 4, 0), ("lt", 1, 4, 0), ("lt", 2, 4, 0), ("lt", 4, 4, 0), ("range", 3, 8, 40), ("range", 4, 16, 40),
           ("spark", 1, 4, 0), ("spark", 2, 4, 0), ("spark", 3, 4, 0), ("spark", 5, 4, 0), ("spark", 8, 4, 0), ("spark", 16, 4, 0)]   # LASSO_SPARK_UNCONFIRMED: g = prod E_m, degree C


@pytest.mark.parametrize("kind,c,log_m,log_r", CONFIGS)
@pytest.mark.parametrize("n", [2, 1 << 11])
def test_sumcheck_combine_round_and_claim(devs, kind, c, log_m, log_r, n):
    rng = np.random.default_rng(abs(hash((kind, c, n))) % 2**32)
    S = _abi.Strategy(_abi.KINDS[kind], c, log_m, log_r)
    alpha = 2 * c if kind == "lt" else c
    degree = c + 1 if kind in ("lt", "spark") else 2
    polys = [rand_fr(rng, n) for _ in range(alpha)]
    eq = rand_fr(rng, n)

    def run(d):
        pp = [d.upload(x) for x in polys]; pe = d.upload(eq)
        o1 = d.sumcheck_combine_round(S, pp, pe, n, degree)
        o2 = d.combine_claim(S, pp, pe, n)
        for p in pp + [pe]:
            d.free(p)
        return o1, o2
    (a1, a2), (b1, b2) = both(devs, run)
    assert np.array_equal(a1, b1)
    assert np.array_equal(a2, b2)


@pytest.mark.parametrize("c", [1, 2, 3, 5, 8, 16])
@pytest.mark.parametrize("n", [2, 64, 1 << 13])
@pytest.mark.parametrize("pattern", ["random", "ones", "alternating"])
def test_lt_first_round_from_bits(devs, c, n, pattern):
    """lasso_sumcheck_combine_round_lt_u32: the first round of the LT sumcheck from 0 / 1 integer values (exact 128-bit integer Horner walk + one field product per point) ==
    the literal round on the lifted field elements; 'ones' / 'alternating' drive the integer magnitudes to their extremes (|t| up to 17 (17^16 - 1) / 16 at C = 16)"""
    from fieldref import L as FR_P, limbs, to_mont
    rng = np.random.default_rng(c * 77 + n)
    if pattern == "random":
        U = [rng.integers(0, 2, size=n, dtype=np.uint32) for _ in range(2 * c)]
    elif pattern == "ones":        # lines 0 -> 1 everywhere: value x at point x
        U = [np.concatenate([np.zeros(n // 2, dtype=np.uint32), np.ones(n // 2, dtype=np.uint32)]) for _ in range(2 * c)]
    else:                           # lines 1 -> 0 / 0 -> 1 alternating over the memories: values 1 - x and x
        U = [np.concatenate([np.full(n // 2, (m + 1) & 1, dtype=np.uint32), np.full(n // 2, m & 1, dtype=np.uint32)]) for m in range(2 * c)]
    one, zero = limbs(to_mont(1, FR_P)), limbs(0)
    Ps = [np.array([one if x else zero for x in u], dtype=np.uint64).reshape(-1, 4) for u in U]
    eq = rand_fr(rng, n)
    S = _abi.Strategy(_abi.KINDS["lt"], c, 4, 0)

    def run(d):
        pu = [d.upload(u) for u in U]; pp = [d.upload(x) for x in Ps]; pe = d.upload(eq)
        a = d.sumcheck_combine_round_lt_u32(S, pu, pe, n, c + 1)
        b = d.sumcheck_combine_round(S, pp, pe, n, c + 1)
        for p in pu + pp + [pe]:
            d.free(p)
        return a, b
    (a1, b1), (a2, b2) = both(devs, run)
    assert np.array_equal(a1, b1) and np.array_equal(a1, a2) and np.array_equal(b1, b2)


@pytest.mark.parametrize("c,bad_mem,bad_at", [(1, 0, 0), (3, 5, 17), (16, 31, 63), (16, 0, 32), (2, 2, 1)])
def test_lt_first_round_refuses_entries_that_are_not_bits(devs, c, bad_mem, bad_at):
    """lasso_sumcheck_combine_round_lt_u32 is only exact for entries 0 / 1 and nothing used to enforce it at the ABI — a caller passing other integers got a silently
    wrong round polynomial.  Now the kernel checks what it reads (every memory, the last EQ memory included, which never enters the walk) and the call fails; same on the mock."""
    from lasso_amd import LassoError
    n = 64
    rng = np.random.default_rng(c + bad_mem)
    U = [rng.integers(0, 2, size=n, dtype=np.uint32) for _ in range(2 * c)]
    U[bad_mem][bad_at] = 2
    eq = rand_fr(rng, n)
    S = _abi.Strategy(_abi.KINDS["lt"], c, 4, 0)

    def run(d):
        pu = [d.upload(u) for u in U]; pe = d.upload(eq)
        try:
            d.sumcheck_combine_round_lt_u32(S, pu, pe, n, c + 1)
            return "accepted"
        except LassoError as e:
            return "refused"
        finally:
            for p in pu + [pe]:
                d.free(p)
    assert both(devs, run) == ("refused", "refused")
    U[bad_mem][bad_at] = 1      # and the same call is fine again afterwards (the flag does not stick)
    def run2(d):
        pu = [d.upload(u) for u in U]; pe = d.upload(eq)
        a = d.sumcheck_combine_round_lt_u32(S, pu, pe, n, c + 1)
        for p in pu + [pe]:
            d.free(p)
        return a
    a, b = both(devs, run2)
    assert np.array_equal(a, b)


@pytest.mark.parametrize("c", [1, 2, 3, 5, 8, 16])
@pytest.mark.parametrize("n", [2, 8, 1 << 13])
def test_lt_round_prescaled_horner(devs, c, n):
    """the prover's LT round: lasso_lt_prescale once, then lasso_sumcheck_combine_round_lt_scaled (Horner form, 1 / 2 / 3 lanes per index at degree <= 5 / 9 / 17) returns
    what the literal round returns on the unscaled arrays — on the device AND against the oracle's literal loop — and the scaled arrays are 32^-(C-1-m) LT_m"""
    rng = np.random.default_rng(c * 1000 + n)
    S = _abi.Strategy(_abi.KINDS["lt"], c, 4, 0)
    polys = [rand_fr(rng, n) for _ in range(2 * c)]
    eq = rand_fr(rng, n)

    def run(d):
        pp = [d.upload(x) for x in polys]; pe = d.upload(eq)
        literal = d.sumcheck_combine_round(S, pp, pe, n, c + 1)
        qq = [d.alloc(32 * n) for _ in pp]
        d.lt_prescale(S, qq, n, src=pp)          # out of place: clone + scaling in one pass; the sources stay as they were
        assert all(np.array_equal(d.download(p, (n, 4)), x) for p, x in zip(pp, polys))
        d.lt_prescale(S, pp, n)                  # in place
        scaled = d.sumcheck_combine_round_lt_scaled(S, pp, pe, n, c + 1)
        arrays = [d.download(p, (n, 4)) for p in pp]
        assert all(np.array_equal(d.download(q, (n, 4)), a) for q, a in zip(qq, arrays))
        for p in pp + qq + [pe]:
            d.free(p)
        return literal, scaled, arrays
    (l1, s1, a1), (l2, s2, a2) = both(devs, run)
    assert np.array_equal(l1, l2) and np.array_equal(s1, s2) and np.array_equal(l1, s1)
    for x, y in zip(a1, a2):
        assert np.array_equal(x, y)


@pytest.mark.parametrize("n,k", [(1, 1), (7, 2), (1 << 12, 5), (1 << 16, 1)])
def test_multi_dot(devs, n, k):
    rng = np.random.default_rng(n * 3 + k)
    polys = [rand_fr(rng, n) for _ in range(k)]
    w = rand_fr(rng, n)

    def run(d):
        pp = [d.upload(x) for x in polys]; pw = d.upload(w)
        out = d.multi_dot(pp, pw, n)
        for p in pp + [pw]:
            d.free(p)
        return out
    a, b = both(devs, run)
    assert np.array_equal(a, b)


@pytest.mark.parametrize("n", [2, 4, 8, 512, 1024, 1 << 13])
def test_gp_build(devs, n):
    rng = np.random.default_rng(n)
    leaves = rand_fr(rng, n)

    def run(d):
        tree = np.zeros((2 * n, 4), dtype=np.uint64); tree[:n] = leaves
        p = d.upload(tree)
        d.gp_build(p, n)
        out = d.download(p, (2 * n - 2, 4))
        d.free(p)
        return out
    a, b = both(devs, run)
    assert np.array_equal(a, b)


@pytest.mark.parametrize("s,log_m", [(16, 4), (1 << 12, 8), (1 << 14, 16)])
def test_fingerprints_gather_from_u32(devs, s, log_m):
    rng = np.random.default_rng(s)
    m = 1 << log_m
    table_u32 = rng.integers(0, 1 << 16, size=m, dtype=np.uint32)
    dim = rng.integers(0, m, size=s, dtype=np.uint32)
    read_u32 = rng.integers(0, 1000, size=s, dtype=np.uint32)
    final_u32 = rng.integers(0, 1000, size=m, dtype=np.uint32)
    table_u32[0] = 0xFFFFFFFF                      # from_u32 edge
    gamma, tau = rand_fr(rng, 2, edge=False)

    def run(d):
        pt32 = d.upload(table_u32); pdim = d.upload(dim); pr32 = d.upload(read_u32); pf32 = d.upload(final_u32)
        pt = d.alloc(32 * m); pr = d.alloc(32 * s); pf = d.alloc(32 * m); pe = d.alloc(32 * s)
        d.fr_from_u32(pt32, m, pt); d.fr_from_u32(pr32, s, pr); d.fr_from_u32(pf32, m, pf)
        d.gather(pt, pdim, s, pe)
        ro = d.alloc(32 * s); wo = d.alloc(32 * s); io = d.alloc(32 * m); fo = d.alloc(32 * m)
        d.fingerprint_ops(pt, pdim, pr, s, gamma, tau, ro, wo)
        d.fingerprint_mem(pt, pf, m, gamma, tau, io, fo)
        outs = [d.download(pt, (m, 4)), d.download(pe, (s, 4)), d.download(ro, (s, 4)), d.download(wo, (s, 4)), d.download(io, (m, 4)), d.download(fo, (m, 4))]
        for p in (pt32, pdim, pr32, pf32, pt, pr, pf, pe, ro, wo, io, fo):
            d.free(p)
        return outs
    a, b = both(devs, run)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


@pytest.mark.parametrize("s,log_m", [(4, 2), (16, 4), (1 << 10, 8), (1 << 13, 8), (1 << 17, 16)])
def test_fingerprint_ops_gp_fused(devs, s, log_m):
    """lasso_fingerprint_ops_gp (leaves + first product layer in one kernel, then the remaining layers of both trees) == lasso_fingerprint_ops + two lasso_gp_build,
    on the device AND against the mock's literal statement: every element of both 2s-element tree arenas"""
    rng = np.random.default_rng(s + log_m)
    m = 1 << log_m
    table = rand_fr(rng, m); read = small_fr(rng.integers(0, 1 << 20, size=s, dtype=np.uint64))
    dim = rng.integers(0, m, size=s, dtype=np.uint32); dim[0] = m - 1
    gamma, tau = rand_fr(rng, 2, edge=False)

    def run(d):
        pt = d.upload(table); pd = d.upload(dim); pr = d.upload(read)
        tr = d.alloc(64 * s); tw = d.alloc(64 * s); xr = d.alloc(64 * s); xw = d.alloc(64 * s)
        d.fingerprint_ops_gp(pt, pd, pr, s, gamma, tau, tr, tw)
        d.fingerprint_ops(pt, pd, pr, s, gamma, tau, xr, xw); d.gp_build(xr, s); d.gp_build(xw, s)
        n_used = 2 * s - 2      # leaves s, then s/2, ..., 2
        outs = [d.download(p, (2 * s, 4))[:n_used] for p in (tr, tw, xr, xw)]
        for p in (pt, pd, pr, tr, tw, xr, xw):
            d.free(p)
        return outs
    a, b = both(devs, run)
    assert np.array_equal(a[0], a[2]) and np.array_equal(a[1], a[3])          # fused == separate calls on the device
    for x, y in zip(a, b):
        assert np.array_equal(x, y)                                             # device == mock (the reference's loops)


@pytest.mark.parametrize("s,log_m", [(4, 2), (64, 4), (1 << 12, 8), (1 << 16, 16)])
def test_fingerprint_leafless_trees_and_strips(devs, s, log_m):
    """capacity mode's two entry points: lasso_fingerprint_ops_gp_upper == the layers above the leaves of lasso_fingerprint_ops_gp's trees, and lasso_fingerprint_ops_strips
    == the leaves at the positions a chunked round reads (2 strips s/4 apart for the first round, 4 strips s/8 apart for the second), on the device and against the mock"""
    rng = np.random.default_rng(s * 3 + log_m)
    m = 1 << log_m
    read_int = rng.integers(0, 1 << 20, size=s, dtype=np.uint64); read_int[0] = 0; read_int[-1] = (1 << 32) - 1
    table = rand_fr(rng, m); read = small_fr(read_int)
    dim = rng.integers(0, m, size=s, dtype=np.uint32); dim[-1] = m - 1
    gamma, tau = rand_fr(rng, 2, edge=False)
    picks = []      # (nstrips, i0, cs)
    for nstrips in (2, 4):
        stride = s // 2 // nstrips
        if stride >= 1:
            picks.append((nstrips, 0, stride))
            if stride >= 8:
                picks += [(nstrips, stride // 8 * 3, stride // 8), (nstrips, stride - 1, 1)]

    def run(d):
        pt = d.upload(table); pd = d.upload(dim); pr = d.upload(read)
        tr = d.alloc(64 * s); tw = d.alloc(64 * s); ur = d.alloc(32 * s); uw = d.alloc(32 * s)
        d.fingerprint_ops_gp(pt, pd, pr, s, gamma, tau, tr, tw)
        d.fingerprint_ops_gp_upper(pt, pd, pr, s, gamma, tau, ur, uw)
        full = [d.download(p, (2 * s, 4))[: 2 * s - 2] for p in (tr, tw)]
        upper = [d.download(p, (s, 4))[: s - 2] for p in (ur, uw)]
        # the compact form: the timestamps back as 32-bit integers (lasso_fr_to_u32), and the same two entry points reading those
        p32 = d.alloc(4 * s)
        assert d.fr_to_u32(pr, s, p32) == int(read_int.max())
        assert np.array_equal(d.download(p32, (s,), dtype=np.uint32), read_int.astype(np.uint32))
        d.fingerprint_ops_gp_upper(pt, pd, p32, s, gamma, tau, ur, uw, read_u32=True)
        upper32 = [d.download(p, (s, 4))[: s - 2] for p in (ur, uw)]
        strips = []
        for nstrips, i0, cs in picks:
            n = 2 * nstrips * cs
            o_r = d.alloc(32 * n); o_w = d.alloc(32 * n)
            d.fingerprint_ops_strips(pt, pd, pr, s, gamma, tau, nstrips, i0, cs, o_r, o_w)
            strips.append((d.download(o_r, (n, 4)), d.download(o_w, (n, 4))))
            d.fingerprint_ops_strips(pt, pd, p32, s, gamma, tau, nstrips, i0, cs, o_r, o_w, read_u32=True)
            assert np.array_equal(strips[-1][0], d.download(o_r, (n, 4))) and np.array_equal(strips[-1][1], d.download(o_w, (n, 4)))
            d.free(o_r); d.free(o_w)
        pbig = d.upload(small_fr(np.array([5, 1 << 32, 7], dtype=np.uint64)))
        with pytest.raises(Exception):          # a value that does not fit 32 bits is refused
            d.fr_to_u32(pbig, 3, p32)
        for p in (pt, pd, pr, tr, tw, ur, uw, p32, pbig):
            d.free(p)
        return full, upper, strips, upper32
    (full, upper, strips, upper32), (mfull, mupper, mstrips, mupper32) = both(devs, run)
    for c in range(2):
        assert np.array_equal(upper[c], full[c][s:]) and np.array_equal(upper[c], mupper[c])
        assert np.array_equal(upper32[c], upper[c]) and np.array_equal(mupper32[c], upper[c])
    for (nstrips, i0, cs), got, want in zip(picks, strips, mstrips):
        stride = s // 2 // nstrips
        for c in range(2):
            assert np.array_equal(got[c], want[c])
            for arr in range(2):
                for t in range(nstrips):
                    k0 = arr * (s // 2) + t * stride + i0
                    assert np.array_equal(got[c][(arr * nstrips + t) * cs:(arr * nstrips + t + 1) * cs], full[c][k0:k0 + cs])


@pytest.mark.parametrize("ls,rs", [(1, 1), (2, 4), (32, 64), (64, 300), (512, 1024)])
def test_matvec_left(devs, ls, rs):
    rng = np.random.default_rng(ls * 1000 + rs)
    Z = rand_fr(rng, ls * rs)
    Lv = rand_fr(rng, ls, edge=False)

    def run(d):
        p = d.upload(Z)
        out = d.matvec_left(p, Lv, ls, rs)
        d.free(p)
        return out
    a, b = both(devs, run)
    assert np.array_equal(a, b)


@pytest.fixture(scope="module")
def gens_300(devs):
    return gens(devs[1].lib, b"gens_sparse_poly", 300)


@pytest.mark.parametrize("ls,rs,maxv", [(1, 1, 5), (4, 8, 256), (16, 256, 1 << 16), (8, 300, 1 << 24), (3, 100, 1 << 32), (2, 64, None),
                                        # >= 32 rows of scalars <= 16 bits: the byte-table kernel (k_msm_rows8), one and two byte windows, ragged columns, a chunked row, and
                                        # just past its limits (17 and 24 bits: the bucket kernel again)
                                        (64, 256, 256), (40, 77, 2), (300, 300, 1 << 16), (33, 129, 1 << 12), (32, 5, 1 << 9), (64, 100, 1 << 17), (48, 64, 1 << 24),
                                        # >= 1024 rows: one wave per row (k_msm_rows8w), one and two byte windows, fewer columns than lanes, ragged rows (not a multiple of 4), all-zero values
                                        (1024, 64, 256), (1500, 100, 1 << 16), (2049, 33, 2), (1027, 300, 1 << 12), (1024, 8, 1)])
def test_hyrax_commit(devs, gens_300, ls, rs, maxv):
    rng = np.random.default_rng(ls * 31 + rs)
    if maxv is None:
        Z = rand_fr(rng, ls * rs)                      # full-width scalars: the general path
    else:
        vals = rng.integers(0, maxv, size=ls * rs, dtype=np.uint64)
        vals[0] = 0; vals[-1] = maxv - 1
        Z = small_fr(vals)

    def run(d):
        b = d.bases_create(gens_300)
        p = d.upload(Z)
        out = d.hyrax_commit(p, ls, rs, b)
        wire = d.hyrax_commit_compressed(p, ls, rs, b)
        d.free(p); d.bases_destroy(b)
        return out, wire
    (a, wa), (b, wb) = both(devs, run)
    mock_lib = devs[1].lib
    assert compress_points(mock_lib, a) == compress_points(mock_lib, b)
    # rows normalised + serialised on the device == the oracle's serialize_compressed of the same rows
    assert np.array_equal(wa, wb) and [bytes(x) for x in wa] == compress_points(mock_lib, b)


@pytest.mark.parametrize("ls,rs,shape", [(256, 300, "random"), (300, 257, "edges"), (256, 64, "equal"), (512, 300, "sparse"), (333, 300, "groups")])
def test_hyrax_commit_full_width_wide_windows(devs, gens_300, ls, rs, shape, monkeypatch):
    """>= 256 rows of FULL-WIDTH scalars: the 12-bit signed-window bucket form (k_msm_pip_sort / _accumulate / _reduce; by default from 2048 columns on, here forced down to
    the fixture's 300 generators) against the oracle's row commitments and against the nibble-bucket kernel (LASSO_MSM_PIP=0) on the same device.  Shapes: random scalars;
    canonical values built from the digit edge cases (digit 2048 stays positive, 2049 turns negative and carries, all-0xFFF carry chains, 0, 1, p - 1, 2^252 - 1); rows of
    EQUAL scalars (every pair of a row in 21 buckets: the size ranking's worst case); mostly-zero rows (empty buckets everywhere); rows in several groups (small scratch)."""
    rng = np.random.default_rng(ls * 131 + rs)
    if shape in ("random", "groups"):
        Z = rand_fr(rng, ls * rs)
        if shape == "groups":   # a 16 MB scratch: the rows go through in groups of 64 (the floor), the last one partly filled
            monkeypatch.setenv("LASSO_MSM_PIP_SCRATCH_MB", "16")
    elif shape == "edges":
        pats = [0, 1, 2048, 2049, 4095, 4096, FR_P - 1, FR_P - 2, 2**252 - 1, 2**252, sum(0x800 << (12 * w) for w in range(21)), sum(0x801 << (12 * w) for w in range(21)),
                sum(0xFFF << (12 * w) for w in range(21)), sum(0x7FF << (12 * w) for w in range(21)), (0x801 << 240) + 0x800, 2**251 + 2**11]
        vals = [pats[int(i)] % FR_P for i in rng.integers(0, len(pats), size=ls * rs)]
        Z = np.array([limbs(to_mont(v, FR_P)) for v in vals], dtype=np.uint64).reshape(-1, 4)
    elif shape == "equal":
        rows = rand_fr(rng, ls, edge=False)
        Z = np.repeat(rows, rs, axis=0)
    else:
        Z = rand_fr(rng, ls * rs, edge=False)
        Z[rng.random(ls * rs) < 0.97] = 0
    real, mock = devs
    monkeypatch.setenv("LASSO_MSM_PIP_MIN_COLS", "32")

    def run(d, pip):
        monkeypatch.setenv("LASSO_MSM_PIP", "1" if pip else "0")
        b = d.bases_create(gens_300)
        p = d.upload(Z)
        out = d.hyrax_commit(p, ls, rs, b)
        wire = d.hyrax_commit_compressed(p, ls, rs, b)
        d.free(p); d.bases_destroy(b)
        return out, wire
    (a, wa), (a0, wa0), (b, wb) = run(real, True), run(real, False), run(mock, True)
    want = compress_points(mock.lib, b)
    assert compress_points(mock.lib, a) == want and compress_points(mock.lib, a0) == want
    assert np.array_equal(wa, wb) and np.array_equal(wa0, wb) and [bytes(x) for x in wa] == want


@pytest.mark.parametrize("world,ls,rs,maxv", [(2, 8, 64, 300), (4, 64, 256, None), (8, 300, 256, 1 << 16), (2, 1, 2, 5)])
def test_slab_commitment_exchange_on_device(devs, gens_300, world, ls, rs, maxv):
    """Slab mode's device-side exchange (lasso_hip.h "slab mode"): rank g commits to the columns = g (mod P) of every row over its own generator table and
    leaves the L partial row sums on the device (lasso_hyrax_commit_rows_dev); lasso_rccl_allgather moves them; lasso_points_reduce_compress adds the P
    partials of each row and compresses.  One MI355X here, so the P ranks are P passes on one context and the all-gather is exercised with a one-rank
    RCCL communicator (the collective degenerates to a copy on the library's stream — the plumbing, dlopen and stream use are the real ones); the result
    must be the full commitment's wire bytes."""
    d = devs[0]
    rng = np.random.default_rng(world * 1000 + ls + rs)
    Z = rand_fr(rng, ls * rs) if maxv is None else small_fr(rng.integers(0, maxv, size=ls * rs, dtype=np.uint64))
    G = gens_300[:rs]
    b_full = d.bases_create(gens_300[: rs + 1])
    pz = d.upload(Z)
    want = d.hyrax_commit_compressed(pz, ls, rs, b_full)
    rb = d.lib.lasso_point_row_bytes()
    p_all = d.alloc(world * ls * rb); p_part = d.alloc(ls * rb)
    Zm = Z.reshape(ls, rs, 4)
    uid = (C.c_uint8 * 128)()
    have_rccl = d.lib.lasso_rccl_unique_id(uid) == 0 and d.lib.lasso_rccl_init(d.ctx, 0, 1, uid) == 0
    for g in range(world):
        bg = d.bases_create(np.ascontiguousarray(G[g::world]))
        pg = d.upload(np.ascontiguousarray(Zm[:, g::world, :]).reshape(-1, 4))
        d._chk(d.lib.lasso_hyrax_commit_rows_dev(d.ctx, C.c_void_p(pg), ls, rs // world, C.c_void_p(bg), C.c_void_p(p_part)))
        if have_rccl:      # world-1 all-gather: p_part -> slot g of p_all on the library's stream
            d._chk(d.lib.lasso_rccl_allgather(d.ctx, C.c_void_p(p_part), C.c_void_p(p_all + g * ls * rb), ls * rb))
        else:
            d._chk(d.lib.lasso_copy(d.ctx, C.c_void_p(p_all + g * ls * rb), C.c_void_p(p_part), ls * rb))
        d.sync(); d.free(pg); d.bases_destroy(bg)
    out = np.empty((ls, 32), dtype=np.uint8)
    d._chk(d.lib.lasso_points_reduce_compress(d.ctx, C.c_void_p(p_all), world, ls, out.ctypes.data_as(C.c_void_p)))
    if have_rccl:
        assert d.lib.lasso_rccl_ready(d.ctx) == 1
        d._chk(d.lib.lasso_rccl_shutdown(d.ctx))
        assert d.lib.lasso_rccl_ready(d.ctx) == 0
    for p in (pz, p_all, p_part):
        d.free(p)
    d.bases_destroy(b_full)
    assert [bytes(x) for x in out] == [bytes(x) for x in np.asarray(want).reshape(ls, 32)]
    assert have_rccl, "librccl could not be loaded / initialised on this box (the exchange itself was still checked through a plain copy)"


def test_mem_stats_accounts_allocations_tables_and_peak(gens_300):
    """lasso_mem_stats: bytes held through a context (buffers, generator tables), the high-water mark and its reset — what bench.py's peak_bytes_per_rank rests on"""
    from lasso_amd import Device
    d = Device(curve="bn254" if os.environ.get("LASSO_TEST_CURVE") == "bn254" else "curve25519")
    u64 = C.c_uint64

    def stats(reset=0):
        live, peak = u64(), u64()
        assert d.lib.lasso_mem_stats(d.ctx, C.byref(live), C.byref(peak), reset) == 0
        return live.value, peak.value
    l0, p0 = stats()
    assert p0 >= l0 > 0                      # the context's own scratch
    a = d.alloc(1 << 26); b = d.alloc(3 << 20)
    l1, p1 = stats()
    assert l1 == l0 + (1 << 26) + (3 << 20) and p1 >= l1
    d.free(a)
    l2, p2 = stats()
    assert l2 == l0 + (3 << 20) and p2 == p1  # the peak stays
    l3, p3 = stats(reset=1); l4, p4 = stats()
    assert p4 == l4 == l2                     # reset: the peak restarts at what is live
    bases = d.bases_create(np.ascontiguousarray(gens_300[:40]))
    l5, _ = stats()
    assert l5 > l4 + 40 * 64 * 100            # window table (+ digit multiples) of 40 generators
    d.bases_destroy(bases); d.free(b)
    l6, _ = stats()
    assert l6 == l0
    # the scratch buffer grows with the largest request served and stays; lasso_trim shrinks it back
    n = 1 << 16
    polys = [d.upload(np.zeros((n, 4), dtype=np.uint64)) for _ in range(40)]; w = d.upload(np.zeros((n, 4), dtype=np.uint64))
    d.multi_dot(polys, w, n)
    for p in polys + [w]:
        d.free(p)
    l7, _ = stats()
    assert d.lib.lasso_trim(d.ctx) == 0
    l8, _ = stats()
    assert l8 == l0 and l7 >= l8
    d.close()


def test_rccl_two_ranks_on_one_device(devs):
    """drive ncclAllGather with world > 1 if the box allows two communicators on one device.  Two contexts of the one MI355X join one communicator
    (lasso_rccl_init rank 0 / 1, one host thread each, as two ranks of a node would).  Either RCCL accepts — then a real 2-rank ncclAllGather runs on the two library streams and
    both ranks must hold [rank 0's bytes, rank 1's bytes] — or it refuses duplicate devices, consistently on BOTH ranks (which is what lets lasso_host_set_comm_shm's agreement
    protocol fall back to the shared-memory exchange without splitting the ranks).  The outcome is printed; a hang or a split decision fails."""
    import threading
    from lasso_amd import Device
    if devs[0].lib.lasso_rccl_available() != 1:
        pytest.skip("librccl cannot be loaded on this box")
    curve = "bn254" if os.environ.get("LASSO_TEST_CURVE") == "bn254" else "curve25519"
    ds = [Device(curve=curve), Device(curve=curve)]
    uid = (C.c_uint8 * 128)()
    assert ds[0].lib.lasso_rccl_unique_id(uid) == 0
    rc, msg = [None, None], [b"", b""]

    def init(k):
        rc[k] = ds[k].lib.lasso_rccl_init(ds[k].ctx, k, 2, uid)
        if rc[k] != 0:
            msg[k] = ds[k].lib.lasso_last_error(ds[k].ctx)
    ths = [threading.Thread(target=init, args=(k,), daemon=True) for k in range(2)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(timeout=90)
    assert not any(t.is_alive() for t in ths), "ncclCommInitRank hung with two ranks on one device"
    assert (rc[0] == 0) == (rc[1] == 0), f"the ranks disagree: {rc} {msg}"
    if rc[0] != 0:
        print(f"\n[rccl] two ranks on one device: refused on both ranks ({msg[0].decode()[:160]}) — a world > 1 ncclAllGather cannot run on this one-GPU box")
        for d in ds:
            d.close()
        return
    n = 1 << 16
    send = [np.full(n, 17 + k, dtype=np.uint8) for k in range(2)]
    ps = [d.upload(x) for d, x in zip(ds, send)]; pr = [d.alloc(2 * n) for d in ds]
    arc = [None, None]

    def gather(k):
        arc[k] = ds[k].lib.lasso_rccl_allgather(ds[k].ctx, C.c_void_p(ps[k]), C.c_void_p(pr[k]), n)
        ds[k].sync()
    ths = [threading.Thread(target=gather, args=(k,), daemon=True) for k in range(2)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(timeout=90)
    assert not any(t.is_alive() for t in ths), "ncclAllGather hung"
    assert arc == [0, 0]
    for k in range(2):
        got = ds[k].download(pr[k], (2 * n,), dtype=np.uint8)
        assert np.array_equal(got[:n], send[0]) and np.array_equal(got[n:], send[1])
    # round 6: the first-contact self-test lasso_host_set_comm_shm runs on a fresh communicator (1 KB all-gather, every slot checked, bounded wait)
    src = [None, None]

    def selftest(k):
        src[k] = ds[k].lib.lasso_rccl_selftest(ds[k].ctx)
    ths = [threading.Thread(target=selftest, args=(k,), daemon=True) for k in range(2)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(timeout=90)
    assert not any(t.is_alive() for t in ths), "lasso_rccl_selftest hung"
    assert src == [0, 0], [d.lib.lasso_last_error(d.ctx) for d in ds]
    print("\n[rccl] two ranks on one device: accepted; a 2-rank ncclAllGather of 64 KiB per rank ran on the two library streams; lasso_rccl_selftest ok on both")
    for d in ds:
        d._chk(d.lib.lasso_rccl_shutdown(d.ctx)); d.close()


@pytest.mark.parametrize("ls,rs,tbits", [(1, 1, 1), (4, 8, 8), (16, 256, 16), (8, 300, 24), (3, 100, 32), (128, 128, 8), (64, 300, 16), (32, 33, 1), (40, 64, 17)])
def test_hyrax_commit_u32(devs, gens_300, ls, rs, tbits):
    """commitment from the integer values (gathered from an integer table, as for E = T[dim]) == commitment of the same polynomial as field elements"""
    rng = np.random.default_rng(ls * 17 + rs)
    m = 64
    table = rng.integers(0, 1 << tbits, size=m, dtype=np.uint64).astype(np.uint32)
    table[0] = 0; table[-1] = (1 << tbits) - 1
    idx = rng.integers(0, m, size=ls * rs, dtype=np.uint64).astype(np.uint32)
    idx[0] = m - 1
    Z = small_fr(table[idx].astype(np.uint64))

    def run(d):
        b = d.bases_create(gens_300)
        pt = d.upload(table); pi = d.upload(idx); pu = d.alloc(4 * ls * rs)
        d.gather_u32(pt, pi, ls * rs, pu)
        wire = d.hyrax_commit_compressed_u32(pu, int(table.max()), ls, rs, b)
        pz = d.upload(Z)
        ref = d.hyrax_commit_compressed(pz, ls, rs, b)
        for p in (pt, pi, pu, pz):
            d.free(p)
        d.bases_destroy(b)
        return wire, ref
    (wa, ra), (wb, rb) = both(devs, run)
    assert np.array_equal(wa, wb) and np.array_equal(wa, ra) and np.array_equal(ra, rb)


@pytest.mark.parametrize("n", [1, 2, 33, 301])
def test_msm_full_width(devs, gens_300, n):
    rng = np.random.default_rng(n)
    sc = rand_fr(rng, n)
    if n > 2:
        sc[n // 2] = 0                                 # zero scalars are free and must not disturb the sum

    def run(d):
        b = d.bases_create(gens_300)
        out = d.msm(b, sc)
        d.bases_destroy(b)
        return out
    a, b = both(devs, run)
    mock_lib = devs[1].lib
    assert compress_points(mock_lib, a) == compress_points(mock_lib, b)


def test_msm_linearity_large(devs, oracle):
    """size-independent property at a size the CPU oracle would not finish quickly: msm(a) + msm(b) == msm(a+b) over 2^13
    generators; only the three resulting points go through the oracle."""
    real, mock = devs
    mock_lib = mock.lib
    n = 1 << 13
    g = gens(mock_lib, b"gens_sparse_poly", n)
    rng = np.random.default_rng(99)
    a = rand_fr(rng, n, edge=False); b2 = rand_fr(rng, n, edge=False)
    s = np.empty_like(a)
    for i in range(n):   # Montgomery form is linear: add the limb values mod p
        x = (sum(int(a[i, k]) << (64 * k) for k in range(4)) + sum(int(b2[i, k]) << (64 * k) for k in range(4))) % FR_P
        s[i] = limbs(x)
    bases = real.bases_create(g)
    pa, pb, ps = real.msm(bases, a), real.msm(bases, b2), real.msm(bases, s)
    real.bases_destroy(bases)
    ca, cb, cs = compress_points(mock_lib, pa)[0], compress_points(mock_lib, pb)[0], compress_points(mock_lib, ps)[0]
    U8 = C.c_uint64 * 8
    xa, xb, xo = U8(), U8(), U8()
    assert oracle.orc_pt_decompress(ca, xa) == 0 and oracle.orc_pt_decompress(cb, xb) == 0
    oracle.orc_pt_add(xa, xb, xo)
    buf = (C.c_uint8 * 32)()
    oracle.orc_pt_compress(xo, buf)
    assert bytes(buf) == cs


@pytest.mark.parametrize("nk", [2, 8, 256, 1 << 12])
def test_inner_products_lr(devs, nk):
    rng = np.random.default_rng(nk + 1)
    a = rand_fr(rng, nk); b = rand_fr(rng, nk)

    def run(d):
        pa = d.upload(a); pb = d.upload(b)
        out = d.inner_products_lr(pa, pb, nk)
        d.free(pa); d.free(pb)
        return out
    x, y = both(devs, run)
    assert np.array_equal(x, y)


@pytest.mark.parametrize("n,nk", [(2, 2), (8, 8), (8, 2), (64, 16), (256, 256), (256, 4)])
def test_bullet_lr_and_fold(devs, n, nk):
    """one bullet-reduction round on the virtually folded generators vs the oracle folding G explicitly (bullet.rs:84-132)"""
    rng = np.random.default_rng(n * 100 + nk)
    mock_lib = devs[1].lib
    g = gens(mock_lib, b"gens_sparse_poly", n + 1)       # n + 2 points: G_0..G_{n-1}, Q, H
    nw = n // nk
    a = rand_fr(rng, nk, edge=False); b = rand_fr(rng, nk, edge=False); w = rand_fr(rng, nw, edge=False)
    tail = rand_fr(rng, 4, edge=False)
    u, ui = rand_fr(rng, 2, edge=False)

    def run(d):
        bases = d.bases_create(g)
        pa = d.upload(a); pb = d.upload(b); pw = d.upload(w); pw2 = d.alloc(32 * 2 * nw)
        lr = d.bullet_lr(bases, n, pa, nk, pw, tail)
        d.bullet_fold(pa, pb, nk, pw, nw, pw2, u, ui)
        outs = (lr, d.download(pa, (nk // 2, 4)), d.download(pb, (nk // 2, 4)), d.download(pw2, (2 * nw, 4)))
        wsum = d.msm_dev(bases, pw2, 2 * nw) if 2 * nw <= n else None
        for p in (pa, pb, pw, pw2):
            d.free(p)
        d.bases_destroy(bases)
        return outs, wsum
    (ra, wa), (rb, wb) = both(devs, run)
    assert compress_points(mock_lib, ra[0]) == compress_points(mock_lib, rb[0])
    for x, y in zip(ra[1:], rb[1:]):
        assert np.array_equal(x, y)
    if wa is not None:
        assert compress_points(mock_lib, wa) == compress_points(mock_lib, wb)


@pytest.mark.parametrize("n,nk,fold", [(2, 2, False), (8, 8, False), (8, 2, True), (8, 4, True), (64, 16, True), (256, 256, False), (256, 128, True), (256, 2, True), (1 << 12, 1 << 11, True), (1 << 12, 4, True)])
def test_bullet_round_fused(devs, n, nk, fold):
    """the one-call round (fold of the previous challenge + c_L, c_R + L, R) vs the oracle's explicit fold-then-MSM (bullet.rs:66-132)"""
    rng = np.random.default_rng(n * 1000 + nk + int(fold))
    mock_lib = devs[1].lib
    g = gens(mock_lib, b"gens_sparse_poly", n + 1)
    len_in = 2 * nk if fold else nk
    nw_in = n // len_in
    a = rand_fr(rng, len_in, edge=False); b = rand_fr(rng, len_in, edge=False); w = rand_fr(rng, nw_in, edge=False)
    blinds = rand_fr(rng, 2, edge=False)
    u, ui = rand_fr(rng, 2, edge=False)

    def run(d):
        bases = d.bases_create(g)
        pa = d.upload(a); pb = d.upload(b); pw = d.upload(w)
        pa2 = d.alloc(32 * nk); pb2 = d.alloc(32 * nk); pw2 = d.alloc(32 * 2 * nw_in)
        if fold:
            lr = d.bullet_round(bases, n, pa, pb, pw, pa2, pb2, pw2, nk, u, ui, blinds)
            state = (d.download(pa2, (nk, 4)), d.download(pb2, (nk, 4)), d.download(pw2, (2 * nw_in, 4)))
        else:
            lr = d.bullet_round(bases, n, pa, pb, pw, 0, 0, 0, nk, None, None, blinds)
            state = ()
        for p in (pa, pb, pw, pa2, pb2, pw2):
            d.free(p)
        d.bases_destroy(bases)
        return lr, state
    (la, sa), (lb, sb) = both(devs, run)
    assert compress_points(mock_lib, la) == compress_points(mock_lib, lb)
    for x, y in zip(sa, sb):
        assert np.array_equal(x, y)


@pytest.mark.parametrize("n,nk", [(8, 2), (64, 16), (1024, 512), (4096, 256)])
def test_bullet_round_launched_ahead(devs, n, nk):
    """lasso_bullet_round_ahead + lasso_bullet_post + lasso_result_wait == lasso_bullet_round with the same challenge (L, R and the folded state), on the device and in the mock;
    and a round that never gets its challenge is released by lasso_abort at once (not by its 5 s bail-out), after which the context runs the same round normally."""
    import time
    rng = np.random.default_rng(n * 77 + nk)
    mock_lib = devs[1].lib
    g = gens(mock_lib, b"gens_sparse_poly", n + 1)
    nw_in = n // (2 * nk)
    a = rand_fr(rng, 2 * nk, edge=False); b = rand_fr(rng, 2 * nk, edge=False); w = rand_fr(rng, nw_in, edge=False)
    blinds = rand_fr(rng, 2, edge=False)
    u, ui = rand_fr(rng, 2, edge=False)
    vp = lambda x: np.ascontiguousarray(x, dtype=np.uint64).ctypes.data_as(C.c_void_p)

    def run(d):
        bases = d.bases_create(g)
        assert d.lib.lasso_bullet_ahead_ok(d.ctx, bases) == 1
        pa = d.upload(a); pb = d.upload(b); pw = d.upload(w)
        pa2 = d.alloc(32 * nk); pb2 = d.alloc(32 * nk); pw2 = d.alloc(32 * 2 * nw_in)
        want = d.bullet_round(bases, n, pa, pb, pw, pa2, pb2, pw2, nk, u, ui, blinds)
        want_state = (d.download(pa2, (nk, 4)), d.download(pb2, (nk, 4)), d.download(pw2, (2 * nw_in, 4)))
        # a round that is enqueued and then abandoned
        d._chk(d.lib.lasso_bullet_round_ahead(d.ctx, bases, n, C.c_void_p(pa), C.c_void_p(pb), C.c_void_p(pw), C.c_void_p(pa2), C.c_void_p(pb2), C.c_void_p(pw2), nk, vp(blinds)))
        t0 = time.perf_counter()
        d._chk(d.lib.lasso_abort(d.ctx))
        assert time.perf_counter() - t0 < 1.0, "abort must not wait for the kernel's 5 s bail-out"
        # ... and the real thing on the same context
        for p in (pa2, pb2, pw2):
            d._chk(d.lib.lasso_zero(d.ctx, C.c_void_p(p), 32))
        d._chk(d.lib.lasso_bullet_round_ahead(d.ctx, bases, n, C.c_void_p(pa), C.c_void_p(pb), C.c_void_p(pw), C.c_void_p(pa2), C.c_void_p(pb2), C.c_void_p(pw2), nk, vp(blinds)))
        time.sleep(0.002)                                   # the kernel is waiting on the device for these two scalars
        d._chk(d.lib.lasso_bullet_post(d.ctx, vp(u), vp(ui)))
        got = np.empty((2, 16), dtype=np.uint64)
        d._chk(d.lib.lasso_result_wait(d.ctx, got.ctypes.data_as(C.c_void_p), 8))
        got_state = (d.download(pa2, (nk, 4)), d.download(pb2, (nk, 4)), d.download(pw2, (2 * nw_in, 4)))
        for p in (pa, pb, pw, pa2, pb2, pw2):
            d.free(p)
        d.bases_destroy(bases)
        return want, want_state, got, got_state
    (wa, wsa, ga, gsa), (wb, wsb, gb, gsb) = both(devs, run)
    for want, got in ((wa, ga), (wb, gb), (wa, gb)):
        assert compress_points(mock_lib, np.asarray(want).reshape(2, -1)) == compress_points(mock_lib, np.asarray(got).reshape(2, -1))
    for x, y, z in zip(wsa, gsa, gsb):
        assert np.array_equal(x, y) and np.array_equal(x, z)


@pytest.mark.parametrize("n,ncirc", [(2, 1), (4, 2), (16, 33), (128, 5), (256, 2), (1 << 9, 2), (1 << 13, 8), (1 << 16, 3)])   # n <= 128: latency-shaped kernel
def test_sumcheck_cubic_eqw_round(devs, n, ncirc):
    """eq-weighted round sums: sum_i A(x)[i] B(x)[i] E[i] at x = 0, 2, 3, vs the oracle's loop"""
    rng = np.random.default_rng(n * 3 + ncirc)
    A = [rand_fr(rng, n) for _ in range(ncirc)]; B = [rand_fr(rng, n) for _ in range(ncirc)]; E = rand_fr(rng, n // 2)

    def run(d):
        pa = [d.upload(x) for x in A]; pb = [d.upload(x) for x in B]; pe = d.upload(E)
        out = d.sumcheck_cubic_eqw_round(pa, pb, pe, n)
        for p in pa + pb + [pe]:
            d.free(p)
        return out
    a, b = both(devs, run)
    assert np.array_equal(a, b)


@pytest.mark.parametrize("n,ncirc", [(4, 1), (8, 2), (32, 33), (256, 5), (512, 2), (1 << 10, 2), (1 << 14, 8), (1 << 17, 3)])   # n <= 256: latency-shaped kernel
def test_sumcheck_cubic_eqw_round_fused(devs, n, ncirc):
    """bind A, B with r then the eq-weighted sums of the next round in one pass == bind_top followed by the plain eq-weighted round (sumcheck.rs:49-120)"""
    rng = np.random.default_rng(n * 5 + ncirc)
    A = [rand_fr(rng, n) for _ in range(ncirc)]
    B = [rand_fr(rng, n) for _ in range(ncirc)]
    E = rand_fr(rng, n // 4)
    r = rand_fr(rng, 1, edge=False)[0]

    def run(d):
        pa = [d.upload(x) for x in A]; pb = [d.upload(x) for x in B]; pe = d.upload(E)
        out = d.sumcheck_cubic_eqw_round_fused(pa, pb, pe, n, r)
        again = d.sumcheck_cubic_eqw_round(pa, pb, pe, n // 2)        # the bound arrays must equal a separate bind
        res = (out, again, [d.download(p, (n // 2, 4)) for p in pa + pb])
        for p in pa + pb + [pe]:
            d.free(p)
        return res
    a, b = both(devs, run)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[0], a[1])
    for x, y in zip(a[2], b[2]):
        assert np.array_equal(x, y)
    # the bound arrays are bind_top's
    def bound(d):
        pa = [d.upload(x) for x in A[:1]]
        d.bind_top(pa, n, r)
        out = d.download(pa[0], (n // 2, 4)); d.free(pa[0]); return out
    assert np.array_equal(bound(devs[0]), a[2][0])


@pytest.mark.parametrize("n,ncirc", [(2, 1), (4, 1), (8, 2), (32, 33), (256, 5), (512, 2), (1 << 10, 2), (1 << 14, 8), (1 << 17, 3)])
def test_sumcheck_cubic_eqw2(devs, n, ncirc):
    """two-sum form of the eq-weighted round (begin + wait): (q(0), q_inf) per circuit, with and without the fused bind; q(0) must equal the
    three-sum form's value at 0 and q_inf must reproduce its values at 2 and 3 through q(x) = q(0) + (q(1) - q(0) - q_inf) x + q_inf x^2"""
    rng = np.random.default_rng(n * 11 + ncirc)
    A = [rand_fr(rng, n) for _ in range(ncirc)]
    B = [rand_fr(rng, n) for _ in range(ncirc)]
    E = rand_fr(rng, max(n // 2, 1))
    r = rand_fr(rng, 1, edge=False)[0]

    def run(d):
        pa = [d.upload(x) for x in A]; pb = [d.upload(x) for x in B]; pe = d.upload(E)
        first = d.sumcheck_cubic_eqw2(pa, pb, pe, n)
        three = d.sumcheck_cubic_eqw_round(pa, pb, pe, n)
        res = [first, three]
        if n >= 4:
            res.append(d.sumcheck_cubic_eqw2(pa, pb, pe, n, r))
            res.append(d.sumcheck_cubic_eqw_round(pa, pb, pe, n // 2))
            res.append(canon(np.stack([d.download(p, (n // 2, 4)) for p in pa + pb])))   # between rounds the device keeps a lazily reduced representative
        for p in pa + pb + [pe]:
            d.free(p)
        return res
    a, b = both(devs, run)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    # consistency with the three-sum form, in exact arithmetic on the host
    from fieldref import from_mont, unlimbs
    def ints(rows): return [from_mont(unlimbs(row), FR_P) for row in rows]
    for two, three in ([(a[0], a[1])] + ([(a[2], a[3])] if n >= 4 else [])):
        t2, t3 = ints(two), ints(three)
        for c in range(ncirc):
            q0, qi = t2[2 * c], t2[2 * c + 1]
            assert q0 == t3[3 * c]
            # q(2) = q0 + 2 l + 4 qi, q(3) = q0 + 3 l + 9 qi  =>  eliminate the linear coefficient l: 3 q(2) - 2 q(3) = q0 - 6 qi
            assert (3 * t3[3 * c + 1] - 2 * t3[3 * c + 2] - q0 + 6 * qi) % FR_P == 0


@pytest.mark.parametrize("n,ncirc,bind", [(2, 1, False), (4, 1, True), (4, 2, False), (8, 2, True), (64, 3, False), (128, 33, False), (256, 2, True), (256, 66, True), (16, 5, True), (512, 2, False), (1024, 3, True), (512, 40, True),
                                           (1024, 2, False), (2048, 2, True), (2048, 5, True), (1024, 33, False)])   # q = 512: the 512-thread / 147 KB form (round 3)
def test_sumcheck_cubic_tail(devs, n, ncirc, bind):
    """the resident tail kernel (all remaining rounds of a layer + the final bind in one launch, challenges through the mailbox) against the
    per-round two-sum calls: same sums every round, same heads"""
    rng = np.random.default_rng(n * 13 + ncirc)
    A = [rand_fr(rng, n) for _ in range(ncirc)]
    B = [rand_fr(rng, n) for _ in range(ncirc)]
    q = n // 4 if bind else n // 2
    E = rand_fr(rng, q)
    r0 = rand_fr(rng, 1, edge=False)[0] if bind else None
    turns = (2 * q).bit_length() - 1
    chal = rand_fr(rng, turns, edge=False)

    def run(d):
        pa = [d.upload(x) for x in A]; pb = [d.upload(x) for x in B]; pe = d.upload(E)
        outs = d.sumcheck_cubic_tail(pa, pb, pe, n, r0, chal)
        for p in pa + pb + [pe]:
            d.free(p)
        return outs

    def per_round(d):
        pa = [d.upload(x) for x in A]; pb = [d.upload(x) for x in B]; pe = d.upload(E)
        outs = [d.sumcheck_cubic_eqw2(pa, pb, pe, n, r0)]
        length = n // 2 if bind else n
        for t in range(turns - 1):
            outs.append(d.sumcheck_cubic_eqw2(pa, pb, pe, length, chal[t])); length //= 2
        d.bind_top(pa + pb, length, chal[turns - 1])
        heads = np.stack([d.download(p, (1, 4))[0] for p in pa + pb])
        outs.append(heads)
        for p in pa + pb + [pe]:
            d.free(p)
        return outs
    a, b = both(devs, run)
    assert len(a) == turns + 1
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    ref = per_round(devs[0])
    for x, y in zip(a, ref):
        assert np.array_equal(x, y)


def test_abort_releases_a_waiting_tail_kernel(devs):
    """a host that stops answering between tail_begin and the last tail_next used to leave the context unusable (tail_active / pending set) and a
    kernel spinning until its 5 s bail-out.  lasso_abort posts the poison tag: the kernel leaves at its next poll, the protocol state is reset, and the very
    same context runs a complete tail afterwards with the right answers."""
    import time
    d = devs[0]
    rng = np.random.default_rng(77)
    n, ncirc = 256, 2
    A = [rand_fr(rng, n) for _ in range(ncirc)]; B = [rand_fr(rng, n) for _ in range(ncirc)]; E = rand_fr(rng, n // 2)
    turns = n.bit_length() - 1
    chal = rand_fr(rng, turns, edge=False)
    pa = [d.upload(x) for x in A]; pb = [d.upload(x) for x in B]; pe = d.upload(E)
    d._chk(d.lib.lasso_sumcheck_cubic_tail_begin(d.ctx, d._ptrs(pa), d._ptrs(pb), ncirc, C.c_void_p(pe), n, None))
    out = np.empty((2 * ncirc, 4), dtype=np.uint64)
    d._chk(d.lib.lasso_result_wait(d.ctx, out.ctypes.data_as(C.c_void_p), 2 * ncirc))      # first round's sums arrive; the kernel now waits for a challenge
    t0 = time.perf_counter()
    d._chk(d.lib.lasso_abort(d.ctx))
    assert time.perf_counter() - t0 < 1.0, "abort must not wait for the kernel's 5 s bail-out"
    want = devs[1].sumcheck_cubic_tail([devs[1].upload(x) for x in A], [devs[1].upload(x) for x in B], devs[1].upload(E), n, None, chal)
    for p in pa + pb:
        d.free(p)
    pa = [d.upload(x) for x in A]; pb = [d.upload(x) for x in B]
    got = d.sumcheck_cubic_tail(pa, pb, pe, n, None, chal)
    assert len(got) == len(want)
    for x, y in zip(got, want):
        assert np.array_equal(x, y)
    for p in pa + pb + [pe]:
        d.free(p)


@pytest.mark.parametrize("n,ncirc,scaled", [(256, 2, False), (512, 3, True), (1 << 12, 2, False), (1 << 14, 2, True), (1 << 15, 5, False), (1 << 15, 16, True),
                                             (1 << 16, 2, True), (1 << 18, 3, False), (1 << 20, 1, True), (1 << 17, 9, False)])   # above 2^14 entries: factor tables in memory (EqGlobal)
def test_sumcheck_cubic_round0_with_inline_eq_table(devs, n, ncirc, scaled):
    """lasso_sumcheck_cubic_eqw2_begin_eq: the first round of a layer with the layer's eq table built inside the launch == lasso_eq_evals_scaled followed by the plain
    first round — same two sums per circuit AND the same table bytes left behind for the later rounds (real library and the oracle's mock)"""
    rng = np.random.default_rng(n + ncirc)
    A = [rand_fr(rng, n) for _ in range(ncirc)]; B = [rand_fr(rng, n) for _ in range(ncirc)]
    ell = (n // 2).bit_length() - 1
    point = rand_fr(rng, max(ell, 1), edge=False)[:ell]
    scale = rand_fr(rng, 1, edge=False)[0] if scaled else None

    def run(d):
        pa = [d.upload(x) for x in A]; pb = [d.upload(x) for x in B]
        e1 = d.alloc(32 * (n // 2)); e2 = d.alloc(32 * (n // 2))
        inline = d.sumcheck_cubic_eqw2_eq(pa, pb, e1, n, point, scale)
        t1 = d.download(e1, (n // 2, 4))
        d.eq_evals_scaled(point, scale, e2)
        plain = d.sumcheck_cubic_eqw2(pa, pb, e2, n)
        t2 = d.download(e2, (n // 2, 4))
        for p in pa + pb + [e1, e2]:
            d.free(p)
        return inline, plain, t1, t2
    (i1, p1, t1, u1), (i2, p2, t2, u2) = both(devs, run)
    assert np.array_equal(i1, p1) and np.array_equal(i1, i2) and np.array_equal(p1, p2)
    assert np.array_equal(t1, u1) and np.array_equal(t1, t2)


@pytest.mark.parametrize("n,ncirc,scaled", [(2, 1, False), (4, 2, True), (64, 3, False), (512, 2, True), (1024, 2, False), (1024, 33, True)])
def test_sumcheck_cubic_tail_with_inline_eq_table(devs, n, ncirc, scaled):
    """lasso_sumcheck_cubic_tail_begin_eq (no table: the resident kernel derives the eq factors from the point) == the tail over a table built by lasso_eq_evals_scaled"""
    rng = np.random.default_rng(n * 3 + ncirc)
    A = [rand_fr(rng, n) for _ in range(ncirc)]; B = [rand_fr(rng, n) for _ in range(ncirc)]
    q = n // 2; ell = q.bit_length() - 1
    point = rand_fr(rng, max(ell, 1), edge=False)[:ell]
    scale = rand_fr(rng, 1, edge=False)[0] if scaled else None
    turns = (2 * q).bit_length() - 1
    chal = rand_fr(rng, turns, edge=False)

    def run(d):
        pa = [d.upload(x) for x in A]; pb = [d.upload(x) for x in B]; pe = d.alloc(32 * q)
        inline = d.sumcheck_cubic_tail_eq(pa, pb, n, point, scale, chal)
        d.eq_evals_scaled(point, scale, pe)
        plain = d.sumcheck_cubic_tail(pa, pb, pe, n, None, chal)
        for p in pa + pb + [pe]:
            d.free(p)
        return inline, plain
    (i1, p1), (i2, p2) = both(devs, run)
    assert len(i1) == turns + 1
    for a, b, c in zip(i1, p1, i2):
        assert np.array_equal(a, b) and np.array_equal(a, c)


@pytest.mark.parametrize("n,alpha,bind", [(2, 1, False), (4, 1, True), (8, 3, False), (64, 2, True), (512, 1, False), (1024, 2, True), (256, 33, True), (16, 8, False), (1024, 1, False), (2048, 3, True)])   # the last two: q = 512
def test_sumcheck_linear_tail(devs, n, alpha, bind):
    """resident tail of the primary sumcheck (linear strategies) against the per-round eq-weighted calls: same dot products every round, same heads,
    and the source arrays are left untouched"""
    rng = np.random.default_rng(n * 17 + alpha)
    Ps = [rand_fr(rng, n) for _ in range(alpha)]
    q = n // 4 if bind else n // 2
    E = rand_fr(rng, q)
    r0 = rand_fr(rng, 1, edge=False)[0] if bind else None
    turns = (2 * q).bit_length() - 1
    chal = rand_fr(rng, turns, edge=False)

    def run(d):
        pp = [d.upload(x) for x in Ps]; pe = d.upload(E)
        outs = d.sumcheck_linear_tail(pp, pe, n, r0, chal)
        after = [d.download(p, (n, 4)) for p in pp]
        for p in pp + [pe]:
            d.free(p)
        return outs, after

    def per_round(d):
        pp = [d.upload(x) for x in Ps]; pe = d.upload(E)
        outs = [d.sumcheck_linear_eqw_round_fused(pp, pe, n, r0) if bind else d.sumcheck_linear_eqw_round(pp, pe, n)]
        length = n // 2 if bind else n
        for t in range(turns - 1):
            outs.append(d.sumcheck_linear_eqw_round_fused(pp, pe, length, chal[t])); length //= 2
        d.bind_top(pp, length, chal[turns - 1])
        outs.append(np.stack([d.download(p, (1, 4))[0] for p in pp]))
        for p in pp + [pe]:
            d.free(p)
        return outs
    (a, a_after), (b, b_after) = both(devs, run)
    assert len(a) == turns + 1
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    for x, y in zip(a_after, Ps):
        assert np.array_equal(x, y)
    ref = per_round(devs[0])
    for x, y in zip(a, ref):
        assert np.array_equal(x, y)


@pytest.mark.parametrize("kind,log_m,log_r,nsub", [("and", 16, 0, 1), ("or", 8, 0, 1), ("xor", 4, 0, 1), ("lt", 8, 0, 2), ("lt", 4, 0, 2), ("range", 16, 40, 3), ("range", 8, 13, 3), ("and", 2, 0, 1)])
def test_materialize_subtable_u32(devs, kind, log_m, log_r, nsub):
    """subtables written by the device == the oracle's restatement of and.rs / or.rs / xor.rs / lt.rs / range_check.rs (whose KATs oracle/kats.cpp pins)"""
    S = _abi.Strategy(_abi.KINDS[kind], 1, log_m, log_r)
    for sub in range(nsub):
        a, b = both(devs, lambda d: d.materialize_subtable_u32(S, sub))
        assert np.array_equal(a, b)
    bits = log_m // 2; i = np.arange(1 << log_m, dtype=np.uint64); l = (i >> bits) & ((1 << bits) - 1); r = i & ((1 << bits) - 1)
    t0 = devs[0].materialize_subtable_u32(S, 0)
    want = {"and": l & r, "or": l | r, "xor": l ^ r, "lt": (l < r).astype(np.uint64), "range": i}[kind]
    assert np.array_equal(t0.astype(np.uint64), want)


def test_launch_wait_protocol_errors(devs):
    """misuse of the launch/wait split is reported as an error code, never a hang or a wrong result: waiting with nothing pending, a wrong count,
    a second deferral, a tail challenge without a tail; and a deferred call still delivers the right values afterwards"""
    rng = np.random.default_rng(77)
    A = rand_fr(rng, 64); W = rand_fr(rng, 64)
    for d in devs:
        lib, ctx = d.lib, d.ctx
        out = np.empty((4, 4), dtype=np.uint64)
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        assert lib.lasso_result_wait(ctx, vp(out), 1) != 0                      # nothing pending
        r = np.ascontiguousarray(A[0])
        assert lib.lasso_sumcheck_cubic_tail_next(ctx, vp(r)) != 0              # no tail running
        pa = d.upload(A); pw = d.upload(W)
        want = d.multi_dot([pa], pw, 64)
        assert lib.lasso_defer_next(ctx) == 0
        assert lib.lasso_defer_next(ctx) != 0                                   # one deferral at a time
        dummy = np.empty((1, 4), dtype=np.uint64)
        ptrs = (C.c_void_p * 1)(pa)
        assert lib.lasso_multi_dot(ctx, ptrs, 1, C.c_void_p(pw), 64, vp(dummy)) == 0
        assert lib.lasso_result_wait(ctx, vp(out), 3) != 0                      # wrong count: still pending
        got = np.empty((1, 4), dtype=np.uint64)
        assert lib.lasso_result_wait(ctx, vp(got), 1) == 0
        assert np.array_equal(got, want)
        assert lib.lasso_result_wait(ctx, vp(out), 1) != 0                      # collected already
        d.free(pa); d.free(pw)


@pytest.mark.parametrize("n_lookups,c,log_m,mode", [(1, 1, 0, "rand"), (2, 1, 1, "rand"), (5, 2, 4, "rand"), (1000, 3, 8, "rand"), (4096, 1, 16, "rand"), (5000, 2, 12, "rand"),
                                                     (1 << 16, 1, 16, "rand"), (70000, 1, 17, "rand"), (9000, 1, 16, "same"), (1 << 15, 2, 3, "rand"), (12345, 1, 9, "sorted")])
def test_densify_dim(devs, n_lookups, c, log_m, mode):
    """device densify (stable radix sort -> timestamps) vs the reference's serial loop (densified.rs:32-57): dim, read, final polynomials and dim_usize, bit-exact;
    ragged n_lookups (zero-padded tail counts as address 0), one hot address, sorted addresses, 1..3 radix passes"""
    rng = np.random.default_rng(n_lookups * 31 + c * 7 + log_m)
    m = 1 << log_m
    s = 1 << max((n_lookups - 1).bit_length(), 0)
    if mode == "same":
        idx = np.full((n_lookups, c), m - 1, dtype=np.uint64)
    else:
        idx = rng.integers(0, m, size=(n_lookups, c), dtype=np.uint64)
        if mode == "sorted":
            idx = np.sort(idx, axis=0)

    def run(d):
        p_idx = d.upload(idx)
        res = []
        for dim in range(c):
            p_u32 = d.alloc(4 * s); p_dim = d.alloc(32 * s); p_read = d.alloc(32 * s); p_fin = d.alloc(32 * m)
            d.densify_dim(p_idx, n_lookups, c, dim, s, log_m, p_u32, p_dim, p_read, p_fin)
            res.append((d.download(p_u32, (s,), dtype=np.uint32), d.download(p_dim, (s, 4)), d.download(p_read, (s, 4)), d.download(p_fin, (m, 4))))
            for p in (p_u32, p_dim, p_read, p_fin):
                d.free(p)
        d.free(p_idx)
        return res
    a, b = both(devs, run)
    for x, y in zip(a, b):
        for u, v in zip(x, y):
            assert np.array_equal(u, v)
    # the oracle side is the reference loop; also check it against a direct numpy statement of the definition for dimension 0
    acc = np.zeros(s, dtype=np.uint64); acc[:n_lookups] = idx[:, 0]
    assert np.array_equal(b[0][0].astype(np.uint64), acc)


def test_densify_rejects_out_of_range(devs):
    d = devs[0]
    idx = np.array([[3], [16]], dtype=np.uint64)
    p_idx = d.upload(idx); p_u32 = d.alloc(8); p_dim = d.alloc(64); p_read = d.alloc(64); p_fin = d.alloc(32 * 16)
    with pytest.raises(Exception):
        d.densify_dim(p_idx, 2, 1, 0, 2, 4, p_u32, p_dim, p_read, p_fin)
    for p in (p_idx, p_u32, p_dim, p_read, p_fin):
        d.free(p)


def test_densify_far_out_of_range_stays_in_bounds(devs):
    """an index far beyond M used to reach the sort / run kernels as a key and index 2*m words of scratch with it (device writes GiBs out of
    bounds).  The key is clamped when flagged: the call must fail with the documented error AND leave the context usable — the same context then
    densifies a valid sequence bit-exactly."""
    d = devs[0]
    log_m, n = 8, 1 << 12
    m = 1 << log_m
    rng = np.random.default_rng(5)
    good = rng.integers(0, m, size=(n, 1), dtype=np.uint64)
    bad = good.copy(); bad[17, 0] = (1 << 40) + 3; bad[n - 1, 0] = (1 << 63); bad[100, 0] = 0xFFFFFFFF
    p_u32 = d.alloc(4 * n); p_dim = d.alloc(32 * n); p_read = d.alloc(32 * n); p_fin = d.alloc(32 * m)
    p_bad = d.upload(bad)
    with pytest.raises(Exception):
        d.densify_dim(p_bad, n, 1, 0, n, log_m, p_u32, p_dim, p_read, p_fin)
    p_good = d.upload(good)
    d.densify_dim(p_good, n, 1, 0, n, log_m, p_u32, p_dim, p_read, p_fin)
    got_read = d.download(p_read, (n, 4)); got_u32 = d.download(p_u32, (n,), dtype=np.uint32)
    assert np.array_equal(got_u32.astype(np.uint64), good[:, 0])
    counts = np.zeros(m, dtype=np.int64); want = np.zeros(n, dtype=np.int64)
    for k in range(n):
        want[k] = counts[good[k, 0]]; counts[good[k, 0]] += 1
    Rinv = pow(1 << 256, -1, FR_P)
    for k in (0, 1, 17, 100, n - 1):
        assert int.from_bytes(got_read[k].tobytes(), "little") * Rinv % FR_P == want[k]
    for p in (p_bad, p_good, p_u32, p_dim, p_read, p_fin):
        d.free(p)


@pytest.mark.parametrize("ls,rs", [(1, 1), (4, 8), (64, 300), (512, 1024)])
def test_matvec_left_dev_and_to_bytes(devs, ls, rs):
    """device-resident forms of the opening's mat-vec and of CanonicalSerialize: equal to the host-result forms / to the canonical integers"""
    rng = np.random.default_rng(ls * 13 + rs)
    Z = rand_fr(rng, ls * rs); L = rand_fr(rng, ls)

    def run(d):
        pz = d.upload(Z); pl = d.upload(L); po = d.alloc(32 * rs)
        d.matvec_left_dev(pz, pl, ls, rs, po)
        out = d.download(po, (rs, 4)); ref = d.matvec_left(pz, L, ls, rs); by = d.fr_to_bytes(po, rs)
        for p in (pz, pl, po):
            d.free(p)
        return out, ref, by
    (a, ra, ba), (b, rb, bb) = both(devs, run)
    assert np.array_equal(a, b) and np.array_equal(a, ra) and np.array_equal(b, rb) and np.array_equal(ba, bb)
    # bytes = the canonical integer of each element, little endian
    Rinv = pow(1 << 256, -1, FR_P)
    for i in range(min(rs, 5)):
        mont = int.from_bytes(a[i].tobytes(), "little")
        assert int.from_bytes(ba[i].tobytes(), "little") == mont * Rinv % FR_P


@pytest.mark.parametrize("n", [1, 2, 64, 298])
def test_msm_dev_scaled(devs, gens_300, n):
    """sum_j (scale*s_j) G_j + t0*Q + t1*H in one MSM (delta of dot_product.rs:219-224) vs the oracle"""
    rng = np.random.default_rng(n + 99)
    sc = rand_fr(rng, n, edge=False); scale = rand_fr(rng, 1, edge=False)[0]; tail = rand_fr(rng, 2, edge=False)

    def run(d):
        b = d.bases_create(gens_300[: n + 2])
        p = d.upload(sc)
        out = d.msm_dev_scaled(b, p, n, scale, tail)
        d.free(p); d.bases_destroy(b)
        return out
    a, b = both(devs, run)
    assert compress_points(devs[1].lib, a) == compress_points(devs[1].lib, b)


@pytest.mark.parametrize("n,alpha", [(2, 1), (4, 2), (64, 1), (1 << 12, 3), (1 << 15, 8)])
def test_sumcheck_linear_rounds_from_u32(devs, n, alpha):
    """the primary sumcheck's first round and first bind from the polynomials' 32-bit integer values == the same calls on the lifted 32-byte field elements: sums, bound arrays"""
    from fieldref import L as FR_P, limbs, to_mont
    rng = np.random.default_rng(n * 11 + alpha)
    U = [rng.integers(0, 2**32, size=n, dtype=np.uint32) for _ in range(alpha)]
    for u in U:
        u[:8] = [0, 1, 2**32 - 1, 2**31, 255, 2**16, 7, 2**32 - 2][: min(8, n)]
    if n > 16:
        U[0][n // 2:] = 2**32 - 1; U[0][: n // 2] = 0      # every difference hi - lo at its extreme
    Ps = [np.array([limbs(to_mont(int(x), FR_P)) for x in u], dtype=np.uint64).reshape(-1, 4) for u in U]
    E = rand_fr(rng, n // 2); r = rand_fr(rng, 1, edge=False)[0]

    def run(d):
        pu = [d.upload(u) for u in U]; pp = [d.upload(x) for x in Ps]; pe = d.upload(E)
        res = [d.sumcheck_linear_eqw_round_u32(pu, pe, n), d.sumcheck_linear_eqw_round(pp, pe, n)]
        if n >= 4:
            pd = [d.alloc(32 * (n // 2)) for _ in Ps]; pd2 = [d.alloc(32 * (n // 2)) for _ in Ps]
            res.append(d.sumcheck_linear_eqw_round_fused_from_u32(pu, pd, pe, n, r)); res.append([d.download(p, (n // 2, 4)) for p in pd])
            res.append(d.sumcheck_linear_eqw_round_fused_from(pp, pd2, pe, n, r)); res.append([d.download(p, (n // 2, 4)) for p in pd2])
            for p in pd + pd2:
                d.free(p)
        for p in pu + pp + [pe]:
            d.free(p)
        return res
    a, b = both(devs, run)
    assert np.array_equal(a[0], a[1]) and np.array_equal(a[0], b[0])
    if n >= 4:
        assert np.array_equal(a[2], a[4]) and np.array_equal(a[2], b[2])
        for x, y, z in zip(a[3], a[5], b[3]):
            assert np.array_equal(x, y) and np.array_equal(x, z)


@pytest.mark.parametrize("n,alpha", [(2, 1), (4, 3), (64, 8), (1 << 10, 2), (1 << 14, 16), (1 << 17, 1)])
def test_sumcheck_linear_eqw_rounds(devs, n, alpha):
    """eq-weighted primary-sumcheck rounds for linear strategies: two dot products per polynomial against the table prefix; fused = bind_top + plain"""
    rng = np.random.default_rng(n * 7 + alpha)
    Ps = [rand_fr(rng, n) for _ in range(alpha)]; E = rand_fr(rng, n // 2)
    r = rand_fr(rng, 1, edge=False)[0]

    def run(d):
        pp = [d.upload(x) for x in Ps]; pe = d.upload(E)
        first = d.sumcheck_linear_eqw_round(pp, pe, n)
        res = [first]
        if n >= 4:
            fused = d.sumcheck_linear_eqw_round_fused(pp, pe, n, r)
            again = d.sumcheck_linear_eqw_round(pp, pe, n // 2)
            res += [fused, again, [d.download(p, (n // 2, 4)) for p in pp]]
            # separate source arrays: same sums, same bound arrays, sources untouched
            ps = [d.upload(x) for x in Ps]; pd = [d.alloc(32 * (n // 2)) for _ in Ps]
            res.append(d.sumcheck_linear_eqw_round_fused_from(ps, pd, pe, n, r))
            res.append([d.download(p, (n // 2, 4)) for p in pd])
            res.append([d.download(p, (n, 4)) for p in ps])
            for p in ps + pd:
                d.free(p)
        for p in pp + [pe]:
            d.free(p)
        return res
    a, b = both(devs, run)
    assert np.array_equal(a[0], b[0])
    if n >= 4:
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[1], a[2])
        for x, y in zip(a[3], b[3]):
            assert np.array_equal(x, y)
        assert np.array_equal(a[4], a[1]) and np.array_equal(b[4], a[1])
        for x, y, z in zip(a[5], a[3], b[5]):
            assert np.array_equal(x, y) and np.array_equal(z, y)
        for x, y in zip(a[6], Ps):
            assert np.array_equal(x, y)


# ---- round 5: tails that hand their arrays to the host, rounds launched ahead of their challenge, the trees' tops in one read
@pytest.mark.parametrize("k,count", [(1, 1), (2, 62), (8, 30), (33, 6), (128, 2), (4, 4094)])
def test_read_runs(devs, k, count):
    rng = np.random.default_rng(k * 31 + count)
    arrs = [rand_fr(rng, count + 5) for _ in range(k)]

    def run(d):
        ptrs = [d.upload(a) for a in arrs]
        shifted = [p + 32 * 3 for p in ptrs]          # runs need not start at an allocation's first element (the tops sit at the end of a tree's arena)
        out = np.empty((k * count, 4), dtype=np.uint64)
        d._chk(d.lib.lasso_read_runs(d.ctx, d._ptrs(shifted), k, count, out.ctypes.data_as(C.c_void_p)))
        for p in ptrs:
            d.free(p)
        return out
    a, b = both(devs, run)
    assert np.array_equal(a, b)
    assert np.array_equal(a, np.concatenate([x[3:3 + count] for x in arrs]))


@pytest.mark.parametrize("n,ncirc,bind,m_stop", [(64, 2, False, 16), (256, 2, True, 16), (1024, 2, False, 2), (2048, 2, True, 16), (2048, 4, True, 8), (1024, 16, False, 2), (512, 3, True, 64),
                                                  (1024, 2, False, 128), (2048, 33, True, 4), (8, 1, False, 2), (16, 2, True, 2), (2048, 2, True, 128)])
def test_cubic_tail_hands_its_arrays_over(devs, n, ncirc, bind, m_stop):
    """lasso_tail_handover_next: the resident tail stops at m_stop elements per array; every round's sums equal the ordinary tail's, and the LAST publication is the arrays
    A_c[0..m_stop), B_c[0..m_stop) — which, bound with the remaining challenges, give the ordinary tail's heads.  Device == mock, and both == the per-round kernels."""
    rng = np.random.default_rng(n * 17 + ncirc * 3 + m_stop)
    A = [rand_fr(rng, n) for _ in range(ncirc)]; B = [rand_fr(rng, n) for _ in range(ncirc)]
    q = n // 4 if bind else n // 2
    E = rand_fr(rng, q)
    r0 = rand_fr(rng, 1, edge=False)[0] if bind else None
    turns_full = (2 * q).bit_length() - 1
    turns = turns_full - (m_stop.bit_length() - 1)            # rounds of sums before the hand-over
    assert turns >= 1
    chal = rand_fr(rng, turns_full, edge=False)
    vp = lambda x: np.ascontiguousarray(x, dtype=np.uint64).ctypes.data_as(C.c_void_p)

    def run(d):
        pa = [d.upload(x) for x in A]; pb = [d.upload(x) for x in B]; pe = d.upload(E)
        d._chk(d.lib.lasso_tail_handover_next(d.ctx, m_stop))
        d._chk(d.lib.lasso_sumcheck_cubic_tail_begin(d.ctx, d._ptrs(pa), d._ptrs(pb), ncirc, C.c_void_p(pe), n, None if r0 is None else vp(r0)))
        outs = []
        for t in range(turns):
            out = np.empty((2 * ncirc, 4), dtype=np.uint64); d._chk(d.lib.lasso_result_wait(d.ctx, vp(out), 2 * ncirc)); outs.append(out)
            d._chk(d.lib.lasso_sumcheck_cubic_tail_next(d.ctx, vp(chal[t])))
        arrs = np.empty((2 * ncirc * m_stop, 4), dtype=np.uint64); d._chk(d.lib.lasso_result_wait(d.ctx, vp(arrs), 2 * ncirc * m_stop))
        # the context is free again: the next tail is an ordinary one (the setting applies to ONE begin)
        full = d.sumcheck_cubic_tail(pa, pb, pe, n, r0, chal)
        for p in pa + pb + [pe]:
            d.free(p)
        return outs, arrs, full
    (oa, aa, fa), (ob, ab, fb) = both(devs, run)
    for x, y in zip(oa, ob):
        assert np.array_equal(x, y)
    assert np.array_equal(aa, ab)
    for x, y in zip(fa, fb):
        assert np.array_equal(x, y)
    for t in range(turns):
        assert np.array_equal(oa[t], fa[t])                   # same sums as the tail that runs to the heads
    # binding the handed-over arrays with the remaining challenges (python big ints) reproduces the heads
    def to_int(row): return (int(row[0]) | int(row[1]) << 64 | int(row[2]) << 128 | int(row[3]) << 192) % FR_P
    RINV = pow(1 << 256, -1, FR_P)
    vals = [[to_int(aa[c * m_stop + i]) * RINV % FR_P for i in range(m_stop)] for c in range(2 * ncirc)]
    for t in range(turns, turns_full):
        r = to_int(chal[t]) * RINV % FR_P
        vals = [[(v[i] + r * (v[i + len(v) // 2] - v[i])) % FR_P for i in range(len(v) // 2)] for v in vals]
    heads = [to_int(row) * RINV % FR_P for row in fa[-1]]
    assert [v[0] for v in vals] == heads


@pytest.mark.parametrize("n,ncirc", [(512, 2), (1 << 12, 2), (1 << 12, 8), (1 << 16, 2), (1 << 18, 3), (1 << 20, 2), (1 << 13, 16)])
def test_cubic_round_launched_ahead(devs, n, ncirc):
    """lasso_sumcheck_cubic_eqw2_begin_ahead + lasso_challenge_post == lasso_sumcheck_cubic_eqw2_begin with the same challenge — sums and the bound arrays — when the round is
    enqueued while the previous round's result is still pending (the prover's schedule), device and mock; a round that never gets its challenge is released by lasso_abort at once;
    and while a launch waits, the stream-synchronising entry points refuse instead of blocking."""
    import time
    rng = np.random.default_rng(n * 3 + ncirc)
    A = [rand_fr(rng, n) for _ in range(ncirc)]; B = [rand_fr(rng, n) for _ in range(ncirc)]
    E = rand_fr(rng, n // 2)
    r1, r2 = rand_fr(rng, 2, edge=False)
    vp = lambda x: np.ascontiguousarray(x, dtype=np.uint64).ctypes.data_as(C.c_void_p)

    def plain(d):
        pa = [d.upload(x) for x in A]; pb = [d.upload(x) for x in B]; pe = d.upload(E)
        outs = [d.sumcheck_cubic_eqw2(pa, pb, pe, n, None), d.sumcheck_cubic_eqw2(pa, pb, pe, n, r1), d.sumcheck_cubic_eqw2(pa, pb, pe, n // 2, r2)]
        state = [d.download(p, (n // 4, 4)) for p in pa + pb]
        for p in pa + pb + [pe]:
            d.free(p)
        return outs, state

    def ahead(d):
        assert d.lib.lasso_rounds_ahead_ok(d.ctx) == 1
        pa = [d.upload(x) for x in A]; pb = [d.upload(x) for x in B]; pe = d.upload(E)
        out0 = np.empty((2 * ncirc, 4), dtype=np.uint64); out1 = np.empty_like(out0); out2 = np.empty_like(out0)
        d._chk(d.lib.lasso_sumcheck_cubic_eqw2_begin(d.ctx, d._ptrs(pa), d._ptrs(pb), ncirc, C.c_void_p(pe), n, None))          # round 0 in flight
        d._chk(d.lib.lasso_sumcheck_cubic_eqw2_begin_ahead(d.ctx, d._ptrs(pa), d._ptrs(pb), ncirc, C.c_void_p(pe), n))           # round 1 behind it, no challenge yet
        if d is devs[0]:       # the device library: these would block behind the waiting gate for its whole bail-out — refused instead (the mock has nothing to wait for)
            assert d.lib.lasso_sync(d.ctx) != 0 and d.lib.lasso_trim(d.ctx) != 0
        d._chk(d.lib.lasso_result_wait(d.ctx, vp(out0), 2 * ncirc))
        time.sleep(0.001)
        d._chk(d.lib.lasso_challenge_post(d.ctx, vp(r1)))
        d._chk(d.lib.lasso_sumcheck_cubic_eqw2_begin_ahead(d.ctx, d._ptrs(pa), d._ptrs(pb), ncirc, C.c_void_p(pe), n // 2))      # round 2 behind round 1
        d._chk(d.lib.lasso_result_wait(d.ctx, vp(out1), 2 * ncirc))
        d._chk(d.lib.lasso_challenge_post(d.ctx, vp(r2)))
        d._chk(d.lib.lasso_result_wait(d.ctx, vp(out2), 2 * ncirc))
        state = [d.download(p, (n // 4, 4)) for p in pa + pb]
        # a round that is enqueued and abandoned: abort returns at once and the context works afterwards
        d._chk(d.lib.lasso_sumcheck_cubic_eqw2_begin_ahead(d.ctx, d._ptrs(pa), d._ptrs(pb), ncirc, C.c_void_p(pe), n // 4))
        t0 = time.perf_counter()
        d._chk(d.lib.lasso_abort(d.ctx))
        assert time.perf_counter() - t0 < 1.0, "abort must not wait for the kernel's 5 s bail-out"
        again = d.sumcheck_cubic_eqw2(pa, pb, pe, n // 4, None)
        for p in pa + pb + [pe]:
            d.free(p)
        return [out0, out1, out2], state, again
    if n // 8 <= 64:      # too short for the streaming kernel: the entry point says so and the caller launches the ordinary round
        d = devs[0]
        pa = [d.upload(x) for x in A]; pb = [d.upload(x) for x in B]; pe = d.upload(E)
        assert d.lib.lasso_sumcheck_cubic_eqw2_begin_ahead(d.ctx, d._ptrs(pa), d._ptrs(pb), ncirc, C.c_void_p(pe), 256) == -4      # LASSO_ERR_UNSUPPORTED
        out = d.sumcheck_cubic_eqw2(pa, pb, pe, 256, None)       # ... and nothing is left behind on the context
        assert out.shape == (2 * ncirc, 4)
        for p in pa + pb + [pe]:
            d.free(p)
        return
    (pa_o, pa_s), (pb_o, pb_s) = both(devs, plain)
    (aa_o, aa_s, aa_again), (ab_o, ab_s, ab_again) = both(devs, ahead)
    for x, y, z, w in zip(pa_o, pb_o, aa_o, ab_o):
        assert np.array_equal(x, y) and np.array_equal(x, z) and np.array_equal(x, w)
    for x, y, z in zip(pa_s, aa_s, ab_s):
        assert np.array_equal(canon(x), canon(y)) and np.array_equal(canon(x), canon(z))      # bound arrays: lazily reduced on the device (fr29_semi), canonical in the mock
    assert np.array_equal(aa_again, ab_again)


@pytest.mark.parametrize("n,ncirc,m_stop", [(2048, 2, 1), (2048, 2, 16), (1024, 8, 4), (256, 3, 1), (2048, 33, 2)])
def test_cubic_tail_launched_ahead(devs, n, ncirc, m_stop):
    """lasso_sumcheck_cubic_tail_begin_ahead (the resident tail enqueued before the challenge it binds first; the first lasso_sumcheck_cubic_tail_next posts it) == the ordinary
    tail begun with that challenge, with and without the hand-over of its arrays; enqueued while a round's result is pending, as the prover does."""
    rng = np.random.default_rng(n * 5 + ncirc + m_stop)
    A = [rand_fr(rng, 2 * n) for _ in range(ncirc)]; B = [rand_fr(rng, 2 * n) for _ in range(ncirc)]      # the round before the tail works on arrays of 2n
    E = rand_fr(rng, n)
    r_prev, r0 = rand_fr(rng, 2, edge=False)
    q = n // 4
    turns = (2 * q).bit_length() - 1 - (m_stop.bit_length() - 1)
    chal = rand_fr(rng, turns, edge=False)
    vp = lambda x: np.ascontiguousarray(x, dtype=np.uint64).ctypes.data_as(C.c_void_p)

    def run(d, ahead):
        pa = [d.upload(x) for x in A]; pb = [d.upload(x) for x in B]; pe = d.upload(E)
        out_prev = np.empty((2 * ncirc, 4), dtype=np.uint64)
        d._chk(d.lib.lasso_sumcheck_cubic_eqw2_begin(d.ctx, d._ptrs(pa), d._ptrs(pb), ncirc, C.c_void_p(pe), 2 * n, vp(r_prev)))     # the round before: binds 2n -> n, result pending
        if m_stop > 1:
            d._chk(d.lib.lasso_tail_handover_next(d.ctx, m_stop))
        if ahead:
            d._chk(d.lib.lasso_sumcheck_cubic_tail_begin_ahead(d.ctx, d._ptrs(pa), d._ptrs(pb), ncirc, C.c_void_p(pe), n))
            d._chk(d.lib.lasso_result_wait(d.ctx, vp(out_prev), 2 * ncirc))
            d._chk(d.lib.lasso_sumcheck_cubic_tail_next(d.ctx, vp(r0)))
        else:
            d._chk(d.lib.lasso_result_wait(d.ctx, vp(out_prev), 2 * ncirc))
            d._chk(d.lib.lasso_sumcheck_cubic_tail_begin(d.ctx, d._ptrs(pa), d._ptrs(pb), ncirc, C.c_void_p(pe), n, vp(r0)))
        outs = [out_prev]
        for t in range(turns):
            out = np.empty((2 * ncirc, 4), dtype=np.uint64); d._chk(d.lib.lasso_result_wait(d.ctx, vp(out), 2 * ncirc)); outs.append(out)
            d._chk(d.lib.lasso_sumcheck_cubic_tail_next(d.ctx, vp(chal[t])))
        fin = np.empty((2 * ncirc * m_stop, 4), dtype=np.uint64); d._chk(d.lib.lasso_result_wait(d.ctx, vp(fin), 2 * ncirc * m_stop)); outs.append(fin)
        for p in pa + pb + [pe]:
            d.free(p)
        return outs
    ref = run(devs[1], False)
    for d in devs:
        for ahead in (False, True):
            if d is devs[1] and not ahead:
                continue
            got = run(d, ahead)
            assert len(got) == len(ref)
            for x, y in zip(got, ref):
                assert np.array_equal(x, y)


@pytest.mark.parametrize("n2,ncirc2,m_stop2,mode", [(1 << 12, 2, 1, "post"), (1 << 15, 3, 1, "post"), (1 << 17, 2, 1, "post"), (1 << 18, 9, 1, "post"), (512, 2, 4, "post"), (1024, 2, 1, "post"),
                                                    (64, 9, 2, "post"), (1 << 13, 2, 1, "cancel"), (1 << 17, 2, 1, "cancel"), (512, 2, 8, "cancel")])
def test_layer_enqueued_ahead_of_its_point(devs, n2, ncirc2, m_stop2, mode):
    """lasso_sumcheck_cubic_eqw2_begin_eq_ahead / lasso_sumcheck_cubic_tail_begin_eq_ahead + lasso_point_post: the NEXT layer's first launch enqueued while the current layer's
    resident tail is still answering, its eq point delivered afterwards == the plain entry points called after the layer (sums, the table left behind, every turn of a resident
    layer, the handed-over arrays); lasso_point_cancel: nothing happens and the plain call that follows gives the same."""
    rng = np.random.default_rng(n2 + ncirc2 * 3 + m_stop2)
    n1, k1 = 256, 2                                                        # the current layer: a resident tail over arrays of 256, handing over at 4
    A1 = [rand_fr(rng, n1) for _ in range(k1)]; B1 = [rand_fr(rng, n1) for _ in range(k1)]
    ell1 = (n1 // 2).bit_length() - 1; pt1 = rand_fr(rng, ell1, edge=False); sc1 = rand_fr(rng, 1, edge=False)[0]; ms1 = 4
    turns1 = ell1 + 1 - (ms1.bit_length() - 1); chal1 = rand_fr(rng, turns1, edge=False)
    A2 = [rand_fr(rng, n2) for _ in range(ncirc2)]; B2 = [rand_fr(rng, n2) for _ in range(ncirc2)]
    ell2 = (n2 // 2).bit_length() - 1; pt2 = rand_fr(rng, max(ell2, 1), edge=False)[:ell2]; sc2 = rand_fr(rng, 1, edge=False)[0]
    tail2 = ell2 <= 9
    turns2 = ell2 + 1 - (m_stop2.bit_length() - 1); chal2 = rand_fr(rng, max(turns2, 1), edge=False)
    vp = lambda x: np.ascontiguousarray(x, dtype=np.uint64).ctypes.data_as(C.c_void_p)

    def layer2_rest(d, outs):          # after the first sums of layer 2 are pending
        out = np.empty((2 * ncirc2, 4), dtype=np.uint64); d._chk(d.lib.lasso_result_wait(d.ctx, vp(out), 2 * ncirc2)); outs.append(out)
        if tail2:
            for t in range(turns2):
                d._chk(d.lib.lasso_sumcheck_cubic_tail_next(d.ctx, vp(chal2[t])))
                cnt = 2 * ncirc2 * (m_stop2 if t == turns2 - 1 else 1)
                out = np.empty((cnt, 4), dtype=np.uint64); d._chk(d.lib.lasso_result_wait(d.ctx, vp(out), cnt)); outs.append(out)

    def run(d, ahead):
        pa1 = [d.upload(x) for x in A1]; pb1 = [d.upload(x) for x in B1]
        pa2 = [d.upload(x) for x in A2]; pb2 = [d.upload(x) for x in B2]
        e2 = d.alloc(32 * max(n2 // 2, 1))
        outs = []
        d._chk(d.lib.lasso_tail_handover_next(d.ctx, ms1))
        d._chk(d.lib.lasso_sumcheck_cubic_tail_begin_eq(d.ctx, d._ptrs(pa1), d._ptrs(pb1), k1, n1, vp(pt1), ell1, vp(sc1)))
        for t in range(turns1 + 1):
            if ahead and t == 1:       # in the middle of the current layer: a tail is active and a result is pending
                if tail2:
                    if m_stop2 > 1:
                        d._chk(d.lib.lasso_tail_handover_next(d.ctx, m_stop2))
                    d._chk(d.lib.lasso_sumcheck_cubic_tail_begin_eq_ahead(d.ctx, d._ptrs(pa2), d._ptrs(pb2), ncirc2, n2, ell2))
                else:
                    d._chk(d.lib.lasso_sumcheck_cubic_eqw2_begin_eq_ahead(d.ctx, d._ptrs(pa2), d._ptrs(pb2), ncirc2, C.c_void_p(e2), n2, ell2))
                if d is devs[0] and not getattr(d, "_dry_run_mock", False):       # the real library: nothing that synchronises the stream is legal now
                    assert d.lib.lasso_sync(d.ctx) != 0
            cnt = 2 * k1 * (ms1 if t == turns1 else 1)
            out = np.empty((cnt, 4), dtype=np.uint64); d._chk(d.lib.lasso_result_wait(d.ctx, vp(out), cnt)); outs.append(out)
            if t < turns1:
                d._chk(d.lib.lasso_sumcheck_cubic_tail_next(d.ctx, vp(chal1[t])))
        if ahead and mode == "post":
            d._chk(d.lib.lasso_point_post(d.ctx, vp(pt2) if ell2 else None, ell2, vp(sc2)))
        else:
            if ahead:
                d._chk(d.lib.lasso_point_cancel(d.ctx))
                assert d.lib.lasso_sync(d.ctx) == 0
            if tail2:
                if m_stop2 > 1:
                    d._chk(d.lib.lasso_tail_handover_next(d.ctx, m_stop2))
                d._chk(d.lib.lasso_sumcheck_cubic_tail_begin_eq(d.ctx, d._ptrs(pa2), d._ptrs(pb2), ncirc2, n2, vp(pt2) if ell2 else None, ell2, vp(sc2)))
            else:
                d._chk(d.lib.lasso_sumcheck_cubic_eqw2_begin_eq(d.ctx, d._ptrs(pa2), d._ptrs(pb2), ncirc2, C.c_void_p(e2), n2, vp(pt2), ell2, vp(sc2)))
        layer2_rest(d, outs)
        if not tail2:
            outs.append(d.download(e2, (n2 // 2, 4)))
        for p in pa1 + pb1 + pa2 + pb2 + [e2]:
            d.free(p)
        return outs
    ref = run(devs[1], False)
    for d in devs:
        got = run(d, True)
        assert len(got) == len(ref)
        for x, y in zip(got, ref):
            assert np.array_equal(x, y)
    got = run(devs[0], False)
    for x, y in zip(got, ref):
        assert np.array_equal(x, y)


def test_big_eq_tables_formed_inside_round0_in_their_own_process():
    """LASSO_EQ_INLINE_BIG=1 (read once per process): tables above 2^14 entries formed inside round 0 from factor tables in memory (EqGlobal; measured, not the default) —
    the round-0 and layer-enqueued-ahead cases of this file once more under that switch."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ, LASSO_EQ_INLINE_BIG="1")
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_kernels.py"), "-m", "gpu", "-q", "-x", "-k", "round0_with_inline_eq_table or layer_enqueued_ahead"],
                         env=e, cwd=root, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and " passed" in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]


@pytest.mark.parametrize("n", [1 << 12, 1 << 17, 512])
def test_layer_enqueued_ahead_and_never_posted_is_aborted_quickly(devs, n):
    """A layer enqueued ahead of its point whose host never posts (an exception in the prover): lasso_abort poisons the point mailbox, the gate and the kernels behind it leave at
    once (not after the gate's 5 s bound), and the context goes on working — the plain call gives what the mock gives."""
    import time
    d = devs[0]
    rng = np.random.default_rng(77)
    ncirc = 2
    A = [rand_fr(rng, n) for _ in range(ncirc)]; B = [rand_fr(rng, n) for _ in range(ncirc)]
    ell = (n // 2).bit_length() - 1; pt = rand_fr(rng, ell, edge=False); sc = rand_fr(rng, 1, edge=False)[0]
    pa = [d.upload(x) for x in A]; pb = [d.upload(x) for x in B]; e = d.alloc(32 * (n // 2))
    if ell <= 9:      # the resident-tail form (a small layer served whole)
        d._chk(d.lib.lasso_sumcheck_cubic_tail_begin_eq_ahead(d.ctx, d._ptrs(pa), d._ptrs(pb), ncirc, n, ell))
    else:
        d._chk(d.lib.lasso_sumcheck_cubic_eqw2_begin_eq_ahead(d.ctx, d._ptrs(pa), d._ptrs(pb), ncirc, C.c_void_p(e), n, ell))
    assert d.lib.lasso_sync(d.ctx) != 0           # waiting on the device: nothing that synchronises is legal
    t0 = time.time()
    assert d.lib.lasso_abort(d.ctx) == 0
    assert time.time() - t0 < 2.0                 # the poison tag, not the 5 s bail-out
    assert d.lib.lasso_sync(d.ctx) == 0
    m = devs[1]
    ma = [m.upload(x) for x in A]; mb = [m.upload(x) for x in B]; me = m.alloc(32 * (n // 2))
    if ell <= 9:
        chal = rand_fr(rng, ell + 1, edge=False)
        got = d.sumcheck_cubic_tail_eq(pa, pb, n, pt, sc, chal); want = m.sumcheck_cubic_tail_eq(ma, mb, n, pt, sc, chal)
        assert len(got) == len(want) and all(np.array_equal(x, y) for x, y in zip(got, want))
    else:
        got = d.sumcheck_cubic_eqw2_eq(pa, pb, e, n, pt, sc); want = m.sumcheck_cubic_eqw2_eq(ma, mb, me, n, pt, sc)
        assert np.array_equal(got, want)
    for p in pa + pb + [e]:
        d.free(p)
    for p in ma + mb + [me]:
        m.free(p)
