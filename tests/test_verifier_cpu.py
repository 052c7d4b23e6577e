"""The product-side verifier (lasso_amd/host/verifier.hpp behind lasso_host_verify: SparsePolynomialEvaluationProof::verify, surge.rs:214-271) on the CPU:
the host sources over the oracle's mock of the device ABI.  It must accept exactly what the oracle's verifier accepts: honest proofs of every strategy
family (the reference's own acceptance criterion, src/e2e_test.rs:54-59), and reject — or refuse to deserialize — tampered bytes, a foreign commitment
and a wrong evaluation point.  Both curve builds."""
import ctypes as C

import numpy as np
import pytest

from lasso_amd import _abi
from lasso_amd.device import LassoError
from proverutil import HostProver, OracleSession, build_mock_prover

CASES = [("lt", 4, 4, 0, 16), ("and", 4, 4, 0, 16), ("range", 3, 8, 40, 16), ("lt", 4, 4, 0, 128),          # e2e_test.rs:64-99
         ("and", 1, 16, 0, 1 << 10), ("xor", 3, 4, 0, 11), ("or", 2, 6, 0, 40), ("range", 2, 8, 12, 100), ("and", 1, 2, 0, 2), ("lt", 1, 4, 0, 3),
         ("spark", 2, 4, 0, 32), ("spark", 5, 4, 0, 20)]   # "spark" = LASSO_SPARK_UNCONFIRMED: the strategy BASELINE.json configs[4] names, restated (not in the reference snapshot)


@pytest.fixture(scope="module", params=["curve25519", "bn254"])
def setup(request):
    import conftest
    curve = request.param
    hp = HostProver(C.CDLL(build_mock_prover(curve)))
    orc = conftest._load_oracle(conftest._build_oracle_bn254() if curve == "bn254" else conftest._build_oracle())
    yield hp, orc
    hp.close()


def _prove(hp, kind, c, log_m, log_r, lookups, seed=3):
    s = 1 << max((lookups - 1).bit_length(), 0)
    alpha = 2 * c if kind == "lt" else c
    idx = np.random.default_rng(seed + lookups).integers(0, 1 << log_m, size=(lookups, c), dtype=np.uint64)
    r = hp.gen_random_point(max(s.bit_length() - 1, 0))
    S = _abi.Strategy(_abi.KINDS[kind], c, log_m, log_r)
    gens = hp.gens(c, s, alpha, log_m)
    dense = hp.densify(idx, log_m)
    comm = hp.commit(dense, gens)
    proof = hp.prove(dense, gens, S, r)
    hp.free(dense)
    return s, idx, r, S, gens, comm, proof


def _verdict(hp, gens, S, s, r, proof, comm):
    """True / False, or None when the bytes do not deserialize (the reference's CanonicalDeserialize would fail before verify is reached)"""
    try:
        return hp.verify(gens, S, s, r, proof, comm)
    except LassoError:
        return None


@pytest.mark.parametrize("kind,c,log_m,log_r,lookups", CASES)
def test_accepts_honest_rejects_tampered(setup, kind, c, log_m, log_r, lookups):
    hp, orc = setup
    s, idx, r, S, gens, comm, proof = _prove(hp, kind, c, log_m, log_r, lookups)
    o = OracleSession(orc, _abi.KINDS[kind], c, log_m, log_r, idx, r)
    try:
        assert hp.verify(gens, S, s, r, proof, comm) is True
        assert o.verify(proof, comm) == 1
        # one flipped bit at positions spread over the whole proof: never accepted, and whenever the oracle's verifier reaches a verdict it is the same one
        step = max(1, len(proof) // 41)
        for pos in range(0, len(proof), step):
            bad = bytearray(proof); bad[pos] ^= 1 << (pos % 8)
            got = _verdict(hp, gens, S, s, r, bytes(bad), comm)
            try:
                want = o.verify(bytes(bad), comm)
            except Exception:
                want = None
            # accepted only where ark-serialize itself sees an equivalent encoding (the sign bit of a point with x = 0, e.g. the identity row of an all-zero
            # memory: DESIGN.md 3.1) — then BOTH verifiers accept, as the reference's would
            assert got is not True or want == 1, f"tampered byte {pos} accepted by the product verifier only"
            if got is not None and want in (0, 1):
                assert got == (want == 1), f"byte {pos}: product verifier {got}, oracle verifier {want}"
        # a commitment with one row replaced by another valid point: rejected
        if len(comm) >= 8 + 64:
            badc = bytearray(comm); badc[8:40], badc[40:72] = comm[40:72], comm[8:40]
            if bytes(badc) != comm:
                assert _verdict(hp, gens, S, s, r, proof, bytes(badc)) is not True
        # a different evaluation point: rejected (unless there is no point: s == 1)
        if r.shape[0]:
            r2 = r.copy(); r2[0] = hp.gen_random_point(r.shape[0] + 1)[-1]
            assert _verdict(hp, gens, S, s, r2, proof, comm) is not True
        # truncated / extended bytes do not deserialize
        assert _verdict(hp, gens, S, s, r, proof[:-1], comm) is None
        assert _verdict(hp, gens, S, s, r, proof + b"\0", comm) is None
        # another transcript label: rejected
        assert hp.verify(gens, S, s, r, proof, comm, transcript=b"other") is False
    finally:
        o.close(); hp.free(None, gens)


def test_verifier_accepts_the_oracle_provers_proof(setup):
    """cross-implementation: a proof made by the ORACLE prover is accepted by the product verifier (and the product prover's by the oracle's verifier, above)"""
    hp, orc = setup
    kind, c, log_m, log_r, lookups = "and", 2, 8, 0, 200
    s, idx, r, S, gens, comm, proof = _prove(hp, kind, c, log_m, log_r, lookups)
    o = OracleSession(orc, _abi.KINDS[kind], c, log_m, log_r, idx, r)
    try:
        assert hp.verify(gens, S, s, r, o.prove(), o.commit()) is True
    finally:
        o.close(); hp.free(None, gens)


@pytest.mark.parametrize("name,curve", [("artefact_and_c1_2p10", "curve25519"), ("artefact_bn254_and_c4_2p8", "bn254"),
                                        ("artefact_and_c1_2p24", "curve25519")])   # the last one: the METRIC instance (AND, C=1, 2^24 lookups), proved on the MI355X
def test_committed_artefact_files_verify(name, curve):
    """tests/golden/<name>/ (tools/dump_proof.py --mock): the ark-serialize files an unmodified Rust `verify` would read.  File-based golden
    check: digests as recorded, accepted by the product verifier AND by the oracle's verifier, from the bytes on disk alone (no lookups, no prover)."""
    import conftest
    import hashlib
    import json
    import os
    d = os.path.join(os.path.dirname(__file__), "golden", name)
    meta = json.load(open(os.path.join(d, "meta.json")))
    files = {k: open(os.path.join(d, k), "rb").read() for k in ("proof.bin", "commitment.bin", "point.bin")}
    for k, v in files.items():
        assert hashlib.sha256(v).hexdigest() == meta["sha256"][k]
    assert meta["curve"] == curve
    p = {"curve25519": 2**252 + 27742317777372353535851937790883648493, "bn254": 21888242871839275222246405745257275088548364400416034343698204186575808495617}[curve]
    pt = files["point.bin"]
    r = np.array([[(int.from_bytes(pt[i:i + 32], "little") << 256) % p >> (64 * k) & (2**64 - 1) for k in range(4)] for i in range(0, len(pt), 32)], dtype=np.uint64).reshape(-1, 4)
    cf = files["commitment.bin"]
    comm, tail = cf[:-24], cf[-24:]
    s, log_m, m = (int.from_bytes(tail[i:i + 8], "little") for i in (0, 8, 16))
    assert (s, log_m, m) == (meta["s"], meta["log_m"], meta["M"])
    S = _abi.Strategy(_abi.KINDS[meta["strategy"]], meta["C"], meta["log_m"], meta["log_r"])
    hp = HostProver(C.CDLL(build_mock_prover(curve)))
    try:
        gens = hp.gens(meta["C"], s, meta["num_memories"], log_m, label=meta["gens_label"].encode())
        assert hp.verify(gens, S, s, r, files["proof.bin"], comm, transcript=meta["transcript_label"].encode()) is True
        hp.free(None, gens)
    finally:
        hp.close()
    orc = conftest._load_oracle(conftest._build_oracle_bn254() if curve == "bn254" else conftest._build_oracle())
    orc.orc_verify_only.argtypes = [C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
    rr = np.ascontiguousarray(r)
    assert orc.orc_verify_only(S.kind, S.c, 1 << log_m, S.log_r, s, rr.ctypes.data_as(C.c_void_p), files["proof.bin"], len(files["proof.bin"]), comm, len(comm)) == 1


def test_length_prefix_cannot_outgrow_the_bytes_behind_it(setup):
    """a vector length was bounded only by the whole input, so a crafted prefix made the reader reserve ~200 bytes per claimed 32-byte point.  Every length is
    now bounded by (remaining bytes) / (element size) before anything is reserved: the largest prefix the old bound let through is refused as 'implausible'."""
    hp, orc = setup
    s, idx, r, S, gens, comm, proof = _prove(hp, "and", 1, 8, 0, 64)
    try:
        assert hp.verify(gens, S, s, r, proof, comm) is True
        for claimed in (len(proof), (len(proof) - 8) // 32 + 1):      # the old bound (k <= n) let both through to reserve(k); (n - 8) / 32 + 1 is the first length the new one refuses
            bad = claimed.to_bytes(8, "little") + proof[8:]
            with pytest.raises(LassoError, match="implausible vector length"):
                hp.verify(gens, S, s, r, bad, comm)
        ok_len = ((len(proof) - 8) // 32).to_bytes(8, "little") + proof[8:]      # plausible by size: read on, and fail on content
        with pytest.raises(LassoError):
            hp.verify(gens, S, s, r, ok_len, comm)
    finally:
        hp.free(None, gens)


def test_bn254_infinity_flag_follows_ark_ec(setup):
    """ark-ec 0.4 (SWCurveConfig::deserialize_with_mode) reads the infinity flag BEFORE looking at x: with it set the point is the identity whatever canonical x the bytes hold,
    and the transcript absorbs the re-serialised identity — so such an encoding of an all-zero row's commitment verifies exactly like the canonical one; both flags set is no
    flag value at all and fails to deserialize.  The product verifier and the oracle's agree on both (they used to reject the first)."""
    hp, orc = setup
    if orc.orc_curve_id() != 1:
        pytest.skip("short-Weierstrass encoding only")
    kind, c, log_m, log_r, lookups = "range", 3, 8, 12, 16      # LOG_R < 2 log M: the third memory is all zeros, its rows commit to the identity
    s, idx, r, S, gens, comm, proof = _prove(hp, kind, c, log_m, log_r, lookups)
    o = OracleSession(orc, _abi.KINDS[kind], c, log_m, log_r, idx, r)
    try:
        ident = bytes(31) + b"\x40"
        rows = [i for i in range(8, len(proof) - 32, 32) if proof[i:i + 32] == ident]
        assert rows, "expected an identity row in comm_derefs"
        at = rows[0]
        alias = bytearray(proof); alias[at] = 1                   # x = 1 (canonical, not even on the curve) under the infinity flag
        assert hp.verify(gens, S, s, r, bytes(alias), comm) is True
        assert o.verify(bytes(alias), comm) == 1
        both = bytearray(proof); both[at + 31] = 0xC0             # negative AND infinity: not an SWFlags value
        assert _verdict(hp, gens, S, s, r, bytes(both), comm) is None
        try:
            assert o.verify(bytes(both), comm) != 1
        except Exception:
            pass
    finally:
        o.close(); hp.free(None, gens)
