"""Pin the oracle's BN254 instantiation (oracle/bn254.hpp, ff.hpp under -DORC_BN254): fields against Python big integers, the group against
the published EIP-196 known answer and big-int affine formulas, ark-serialize's SW flags, and the reference's own acceptance criterion
(prove -> verify, the small-integer KATs of its unit tests) with G = BN254 G1 — the group BASELINE.json's configs[1] names."""
import ctypes

import numpy as np
import pytest

from fieldref import limbs, rng, unlimbs
from test_oracle_kats import NAMES

Q = 21888242871839275222246405745257275088696311157297823662689037894645226208583   # base field
RR = 21888242871839275222246405745257275088548364400416034343698204186575808495617  # scalar field = group order
MONT = 2**256
U4 = ctypes.c_uint64 * 4
U8 = ctypes.c_uint64 * 8
INF = None


def sw_add(P1, P2):
    if P1 is INF:
        return P2
    if P2 is INF:
        return P1
    (x1, y1), (x2, y2) = P1, P2
    if x1 == x2:
        if (y1 + y2) % Q == 0:
            return INF
        lam = 3 * x1 * x1 * pow(2 * y1, -1, Q) % Q
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, Q) % Q
    x3 = (lam * lam - x1 - x2) % Q
    return (x3, (lam * (x1 - x3) - y1) % Q)


def sw_mul(P, k):
    acc = INF
    while k:
        if k & 1:
            acc = sw_add(acc, P)
        P = sw_add(P, P)
        k >>= 1
    return acc


def _pt(P):
    return U8(*(limbs(P[0]) + limbs(P[1])))


def _unpt(v):
    P = (unlimbs(v[0:4]), unlimbs(v[4:8]))
    return INF if P == (0, 0) else P   # ark-ec's affine identity: x = y = 0 + infinity flag


@pytest.mark.parametrize("which,p", [(0, RR), (1, Q)])
def test_field_ops_vs_bigint(oracle_bn254, which, p):
    o_ = oracle_bn254
    r = rng(4321 + which)
    vals = [0, 1, 2, p - 1, p - 2, (p - 1) // 2, (p + 1) // 2, 2**64 - 1, 2**64, 2**128 - 1, 2**192 + 5, 2**253 % p] + [r.randrange(p) for _ in range(200)]
    for i in range(len(vals) - 1):
        a, b = vals[i], vals[(i * 7 + 3) % len(vals)]
        am, bm = U4(*limbs(a * MONT % p)), U4(*limbs(b * MONT % p))
        o = U4()
        back = lambda v: unlimbs(v) * pow(MONT, -1, p) % p
        o_.orc_f_mul(which, am, bm, o); assert back(o) == a * b % p
        o_.orc_f_add(which, am, bm, o); assert back(o) == (a + b) % p
        o_.orc_f_sub(which, am, bm, o); assert back(o) == (a - b) % p
        if a:
            o_.orc_f_inv(which, am, o); assert back(o) == pow(a, -1, p)
        o_.orc_f_from_canonical(which, U4(*limbs(a)), o); assert unlimbs(o) == a * MONT % p
        o_.orc_f_to_canonical(which, am, o); assert unlimbs(o) == a


def test_from_le_bytes_mod_order(oracle_bn254):
    r = rng(6)
    for n in (1, 31, 32, 33, 64):
        for _ in range(10):
            b = bytes(r.randrange(256) for _ in range(n))
            o = U4()
            oracle_bn254.orc_fr_from_le_bytes_mod_order(b, ctypes.c_size_t(n), o)
            assert unlimbs(o) * pow(MONT, -1, RR) % RR == int.from_bytes(b, "little") % RR


def test_generator_and_eip196_known_answer(oracle_bn254):
    """2*(1, 2) as published with EIP-196's ecadd test vectors: pins Fq, the curve equation's model and the doubling formula to an external value."""
    g = U8(); oracle_bn254.orc_pt_generator(g)
    assert _unpt(g) == (1, 2)
    o = U8(); oracle_bn254.orc_pt_dbl(g, o)
    want = (0x030644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd3, 0x15ed738c0e0a7c92e7845f96b2ae9c0a68a6a449e3538fc7ff3ebf7a5a18a2c4)
    assert _unpt(o) == want
    oracle_bn254.orc_pt_add(g, g, o)   # the addition must detect P = Q
    assert _unpt(o) == want
    oracle_bn254.orc_pt_mul(g, U4(*limbs(RR)), o)   # group order
    assert _unpt(o) is INF
    oracle_bn254.orc_pt_mul(g, U4(*limbs(RR - 1)), o)
    assert _unpt(o) == (1, Q - 2)


def test_curve_vs_bigint(oracle_bn254):
    r = rng(78)
    pts = [(1, 2)] + [sw_mul((1, 2), r.randrange(1, RR)) for _ in range(8)]
    o = U8()
    for i, P in enumerate(pts):
        P2 = pts[(i * 3 + 1) % len(pts)]
        assert (P[1] * P[1] - P[0] ** 3 - 3) % Q == 0
        oracle_bn254.orc_pt_add(_pt(P), _pt(P2), o); assert _unpt(o) == sw_add(P, P2)
        oracle_bn254.orc_pt_dbl(_pt(P), o); assert _unpt(o) == sw_add(P, P)
        oracle_bn254.orc_pt_add(_pt(P), _pt((P[0], Q - P[1])), o); assert _unpt(o) is INF    # P + (-P)
        k = r.randrange(RR)
        oracle_bn254.orc_pt_mul(_pt(P), U4(*limbs(k)), o); assert _unpt(o) == sw_mul(P, k)


def test_compress_flags_and_roundtrip(oracle_bn254):
    """ark-ec SWFlags: x in 32 bytes LE; bit 7 of the last byte iff y > -y as canonical integers; bit 6 = infinity."""
    r = rng(10)
    buf = (ctypes.c_uint8 * 32)(); o = U8()
    for _ in range(16):
        P = sw_mul((1, 2), r.randrange(1, RR))
        oracle_bn254.orc_pt_compress(_pt(P), buf)
        b = bytes(buf)
        assert int.from_bytes(b, "little") & (2**254 - 1) == P[0]
        assert (b[31] >> 7) == (1 if P[1] > Q - P[1] else 0) and not (b[31] & 0x40)
        assert oracle_bn254.orc_pt_decompress(buf, o) == 0 and _unpt(o) == P
    bad = (ctypes.c_uint8 * 32)(*([0xff] * 31 + [0x3f]))   # x >= q
    assert oracle_bn254.orc_pt_decompress(bad, o) != 0


def test_msm_matches_naive(oracle_bn254):
    r = rng(32)
    for n, small in ((1, False), (5, True), (40, True), (33, False)):
        bases = [sw_mul((1, 2), r.randrange(1, RR)) for _ in range(n)]
        if n > 4:
            bases[3] = bases[2]   # equal bases: an incomplete addition formula would fail here
        scalars = [r.randrange(2**16) if small else r.randrange(RR) for _ in range(n)]
        if n > 4:
            scalars[3] = scalars[2]
        B = (ctypes.c_uint64 * (8 * n))(*[w for P in bases for w in limbs(P[0]) + limbs(P[1])])
        S = (ctypes.c_uint64 * (4 * n))(*[w for s in scalars for w in limbs(s * MONT % RR)])
        o = U8()
        oracle_bn254.orc_msm(B, S, ctypes.c_size_t(n), o)
        acc = INF
        for P, s in zip(bases, scalars):
            acc = sw_add(acc, sw_mul(P, s))
        assert _unpt(o) == acc


def test_generators_are_distinct_curve_points(oracle_bn254):
    n = 9
    out = (ctypes.c_uint64 * (8 * (n + 1)))()
    assert oracle_bn254.orc_gens(b"gens_sparse_poly", ctypes.c_size_t(n), out) == 0
    pts = [(unlimbs(out[8 * i: 8 * i + 4]), unlimbs(out[8 * i + 4: 8 * i + 8])) for i in range(n + 1)]
    assert len(set(pts)) == n + 1
    for x, y in pts:
        assert (y * y - x ** 3 - 3) % Q == 0


@pytest.mark.parametrize("name", NAMES)
def test_reference_kat_over_bn254(oracle_bn254, name):
    """The reference's unit-test known answers are small integers mod p and its e2e criterion is prove -> verify: both hold for any field / group."""
    assert getattr(oracle_bn254, "orc_kat_" + name)() == 0


def test_session_prove_verify_and_tamper(oracle_bn254):
    from proverutil import OracleSession
    r = np.random.default_rng(5)
    idx = r.integers(0, 1 << 8, size=(64, 2), dtype=np.uint64)
    pt = np.array([limbs(int(v) * MONT % RR) for v in r.integers(1, 2**62, size=6)], dtype=np.uint64)
    s = OracleSession(oracle_bn254, 2, 2, 8, 0, idx, pt)   # XOR, C = 2, M = 2^8
    try:
        com = s.commit(); proof = s.prove()
        assert s.verify(proof, com) == 1
        bad = bytearray(proof); bad[40] ^= 1
        assert s.verify(bytes(bad), com) != 1
    finally:
        s.close()
