"""The caller's generators at the coarse boundary: SparsePolynomialEvaluationProof::prove / DensifiedRepresentation::commit take
`gens: &SparsePolyCommitmentGens<G>` (src/lasso/surge.rs:119-125, densified.rs:78-81) — lasso_host_gens_from_points is that argument.
 (i) points exported from the label-derived object and fed back give the same commitment and proof bytes (the label path is only a convenience);
 (ii) ARBITRARY valid points derived from no label (k * B for scripted k, through the oracle's own group law) give, on product and oracle alike,
      the same bytes — nothing of the library's restatement of arkworks' G::rand stream is in the trust base of commit / prove;
 (iii) a set of the wrong size is refused (surge.rs:39-47 fixes the sizes).
CPU: the host prover over the oracle's mock of the device ABI; -m gpu: the HIP library."""
import ctypes as C

import numpy as np
import pytest

from lasso_amd import _abi
from proverutil import HostProver, OracleSession, build_mock_prover

CASES = [("and", 1, 4, 0, 16), ("xor", 2, 4, 0, 32), ("lt", 2, 4, 0, 16), ("range", 3, 8, 40, 16), ("and", 1, 8, 0, 300)]


def _set_sizes(c, s, alpha, log_m):
    """n of the three PolyCommitmentGens (surge.rs:39-47, dense_mlpoly.rs:38-45, eq_poly.rs:40-42)"""
    def n_of(nv):
        return 1 << (nv - nv // 2)
    lg = lambda x: (x - 1).bit_length()
    return [n_of(lg(2 * c * s)), n_of(lg(c) + log_m), n_of(lg(alpha * s))]


def _scripted_points(oracle, count, salt):
    """count valid group elements that come from NO label: k_i * B, k_i scripted; returns (canonical xy for the oracle, Montgomery xy for the product ABI)"""
    u64p = C.POINTER(C.c_uint64)
    gen = (C.c_uint64 * 8)(); oracle.orc_pt_generator(gen)
    canon = np.empty((count, 8), dtype=np.uint64); mont = np.empty((count, 8), dtype=np.uint64)
    for i in range(count):
        k = (0x9E3779B97F4A7C15 * (i + 1) + salt * 0x1000193 + 7) % (1 << 64) | 1
        sc = (C.c_uint64 * 4)(k, (k * 3) % (1 << 64), i + 11, 5 + salt)      # a 196-bit scalar below both group orders
        out = (C.c_uint64 * 8)()
        oracle.orc_pt_mul(gen, sc, out)
        canon[i] = np.frombuffer(out, dtype=np.uint64)
        for h in range(2):
            o = (C.c_uint64 * 4)()
            oracle.orc_f_from_canonical(1, (C.c_uint64 * 4)(*canon[i, 4 * h:4 * h + 4]), o)
            mont[i, 4 * h:4 * h + 4] = np.frombuffer(o, dtype=np.uint64)
    return canon, mont


def _run(host, oracle, kind, c, log_m, log_r, lookups):
    s = 1 << (lookups - 1).bit_length()
    alpha = 2 * c if kind == "lt" else c
    idx = np.random.default_rng(lookups + c).integers(0, 1 << log_m, size=(lookups, c), dtype=np.uint64)
    r = host.gen_random_point(max(s.bit_length() - 1, 0))
    S = _abi.Strategy(_abi.KINDS[kind], c, log_m, log_r)
    sizes = _set_sizes(c, s, alpha, log_m)
    # (i) label-derived points, exported and fed back
    g_label = host.gens(c, s, alpha, log_m)
    sets = [host.gens_points(g_label, w) for w in range(3)]
    assert [a.shape[0] for a in sets] == [n + 2 for n in sizes]
    g_back = host.gens_from_points(c, s, alpha, log_m, *sets)
    dense = host.densify(idx, log_m)
    comm_a, proof_a = host.commit(dense, g_label), host.prove(dense, g_label, S, r)
    comm_b, proof_b = host.commit(dense, g_back), host.prove(dense, g_back, S, r)
    assert comm_a == comm_b and proof_a == proof_b
    # the three sets of the label path are prefixes of ONE stream (surge.rs:49-53: the same label three times)
    m = min(a.shape[0] for a in sets) - 2
    assert np.array_equal(sets[0][:m], sets[1][:m]) and np.array_equal(sets[0][:m], sets[2][:m])
    # (ii) points that come from no label: three unrelated sets
    canon, mont = [], []
    for w, n in enumerate(sizes):
        cx, mx = _scripted_points(oracle, n + 2, salt=w)
        canon.append(cx); mont.append(mx)
    g_own = host.gens_from_points(c, s, alpha, log_m, *mont)
    comm_c, proof_c = host.commit(dense, g_own), host.prove(dense, g_own, S, r)
    assert comm_c != comm_a and proof_c != proof_a
    assert host.verify(g_own, S, s, r, proof_c, comm_c)
    assert not host.verify(g_label, S, s, r, proof_c, comm_c)      # bound to ITS generators
    orc = OracleSession(oracle, _abi.KINDS[kind], c, log_m, log_r, idx, r)
    try:
        vp = lambda a: np.ascontiguousarray(a).ctypes.data_as(C.c_void_p)
        oracle.orc_session_set_gens.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        rc = oracle.orc_session_set_gens(C.c_void_p(orc.s), vp(canon[0]), canon[0].shape[0], vp(canon[1]), canon[1].shape[0], vp(canon[2]), canon[2].shape[0])
        assert rc == 0, oracle.orc_last_error().decode()
        assert comm_c == orc.commit()
        assert proof_c == orc.prove()
        assert orc.verify(proof_c, comm_c) == 1
    finally:
        orc.close()
    # (iii) wrong sizes are refused, with the needed count in the message
    with pytest.raises(Exception, match="points"):
        host.gens_from_points(c, s, alpha, log_m, mont[0][:-1], mont[1], mont[2])
    with pytest.raises(Exception, match="points"):
        host.gens_from_points(c, s, alpha, log_m, mont[0], np.concatenate([mont[1], mont[1][:1]]), mont[2])
    host.free(dense, g_label); host.free(None, g_back); host.free(None, g_own)


@pytest.fixture(scope="module")
def host_mock():
    hp = HostProver(C.CDLL(build_mock_prover()))
    yield hp
    hp.close()


@pytest.mark.parametrize("kind,c,log_m,log_r,lookups", CASES)
def test_callers_generators_cpu(host_mock, oracle, kind, c, log_m, log_r, lookups):
    _run(host_mock, oracle, kind, c, log_m, log_r, lookups)


@pytest.mark.gpu
@pytest.mark.parametrize("kind,c,log_m,log_r,lookups", CASES + [("and", 1, 16, 0, 1 << 12)])
def test_callers_generators_gpu(oracle, kind, c, log_m, log_r, lookups):
    hp = HostProver()
    try:
        _run(hp, oracle, kind, c, log_m, log_r, lookups)
    finally:
        hp.close()
