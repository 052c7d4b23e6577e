"""CPU: the caller's LIVE transcript and random tape at the C ABI (include/lasso_prover.h lasso_host_prove_cb — VERDICT r3 "missing" 5: surge.rs:119-125 takes `&mut Transcript` and
`&mut RandomTape`, the label-replaying lasso_host_prove could not honour a transcript that already held state).  Host prover over the oracle's mock of the device ABI:
  * with the library's own Merlin objects behind the callbacks (fresh "example" / "proof") the proof is the label path's, byte for byte;
  * a transcript / tape the caller has already written to changes the proof exactly as it changes the ORACLE's (orc_session_prove_seeded), and the verifiers follow;
  * callbacks implemented by the caller (here: Python functions that forward to a Merlin object and record the calls) see the reference's schedule of labels."""
import ctypes as C

import numpy as np
import pytest

from lasso_amd import _abi
from lasso_amd.prover import APPEND_FN, CHALLENGE_FN, Transcript, TranscriptVtbl
from proverutil import HostProver, OracleSession, build_mock_prover


@pytest.fixture(scope="module")
def host():
    hp = HostProver(C.CDLL(build_mock_prover()))
    yield hp
    hp.close()


CASES = [("and", 1, 4, 0, 16), ("xor", 2, 4, 0, 24), ("lt", 2, 4, 0, 16), ("range", 2, 8, 12, 10)]


def instance(host, kind, c, log_m, log_r, lookups):
    s = 1 << (lookups - 1).bit_length()
    idx = np.random.default_rng(lookups + c).integers(0, 1 << log_m, size=(lookups, c), dtype=np.uint64)
    r = host.gen_random_point(s.bit_length() - 1)
    S = _abi.Strategy(_abi.KINDS[kind], c, log_m, log_r)
    return s, idx, r, S, (2 * c if kind == "lt" else c)


@pytest.mark.parametrize("kind,c,log_m,log_r,lookups", CASES)
def test_fresh_merlin_behind_the_callbacks_is_the_label_path(host, kind, c, log_m, log_r, lookups):
    s, idx, r, S, alpha = instance(host, kind, c, log_m, log_r, lookups)
    gens = host.gens(c, s, alpha, log_m); dense = host.densify(idx, log_m)
    t = Transcript(host.lib, b"example"); tape = Transcript(host.lib, b"proof", tape=True)
    try:
        by_label = host.prove(dense, gens, S, r)
        assert host.prove_with(dense, gens, S, r, t.pair(), tape.pair()) == by_label
        # the objects are LIVE: a second proof through the same (now advanced) transcript and tape is a different proof
        assert host.prove_with(dense, gens, S, r, t.pair(), tape.pair()) != by_label
    finally:
        t.close(); tape.close(); host.free(dense, gens)


@pytest.mark.parametrize("kind,c,log_m,log_r,lookups", CASES[:3])
@pytest.mark.parametrize("seed_tape", [False, True])
def test_preseeded_transcript_changes_the_proof_as_it_changes_the_oracles(host, oracle, kind, c, log_m, log_r, lookups, seed_tape):
    s, idx, r, S, alpha = instance(host, kind, c, log_m, log_r, lookups)
    gens = host.gens(c, s, alpha, log_m); dense = host.densify(idx, log_m)
    comm = host.commit(dense, gens)
    t = Transcript(host.lib, b"example"); tape = Transcript(host.lib, b"proof", tape=True)
    pre_label, pre_msg = b"outer protocol", b"state the caller absorbed before calling prove \x00\x01\x02"
    t.append_message(pre_label, pre_msg)
    if seed_tape:
        tape.append_message(b"tape state", b"drawn from already")
    orc = OracleSession(oracle, _abi.KINDS[kind], c, log_m, log_r, idx, r)
    try:
        proof = host.prove_with(dense, gens, S, r, t.pair(), tape.pair())
        oracle.orc_session_prove_seeded.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_size_t, C.c_char_p, C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        buf = (C.c_uint8 * (1 << 22))(); n = C.c_size_t()
        rc = oracle.orc_session_prove_seeded(C.c_void_p(orc.s), pre_label, pre_msg, len(pre_msg), b"tape state" if seed_tape else None, b"drawn from already" if seed_tape else None,
                                             len(b"drawn from already") if seed_tape else 0, buf, len(buf), C.byref(n))
        assert rc == 0, oracle.orc_last_error()
        assert proof == bytes(buf[: n.value])
        assert proof != host.prove(dense, gens, S, r)            # and it is NOT the fresh-transcript proof
        # verification needs the same transcript state: the product verifier through the callbacks, the oracle's seeded verifier
        v = Transcript(host.lib, b"example"); v.append_message(pre_label, pre_msg)
        assert host.verify_with(gens, S, s, r, proof, comm, v.pair()) is True
        v.close()
        v2 = Transcript(host.lib, b"example")
        assert host.verify_with(gens, S, s, r, proof, comm, v2.pair()) is False      # a fresh transcript does not accept it
        v2.close()
        oracle.orc_session_verify_seeded.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
        assert oracle.orc_session_verify_seeded(C.c_void_p(orc.s), pre_label, pre_msg, len(pre_msg), proof, len(proof)) == 1
        assert oracle.orc_session_verify_seeded(C.c_void_p(orc.s), None, None, 0, proof, len(proof)) == 0
    finally:
        orc.close(); t.close(); tape.close(); host.free(dense, gens)


def test_callbacks_implemented_by_the_caller_see_the_reference_schedule(host):
    """the embedder's own transcript object behind the vtbl (what integration/rust/hip.rs does with merlin::Transcript): Python callbacks that forward to a Merlin and log"""
    kind, c, log_m, log_r, lookups = "and", 2, 4, 0, 16      # C = 2: every n-to-1 reduction draws at least one challenge
    s, idx, r, S, alpha = instance(host, kind, c, log_m, log_r, lookups)
    gens = host.gens(c, s, alpha, log_m); dense = host.densify(idx, log_m)
    inner = Transcript(host.lib, b"example"); tape = Transcript(host.lib, b"proof", tape=True)
    log = []

    def on_append(user, label, ll, msg, n):
        lb = bytes(label[:ll]); log.append(("append", lb, n))
        inner.vt.contents.append_message(inner.m, label, ll, msg, n)

    def on_challenge(user, label, ll, dest, n):
        log.append(("challenge", bytes(label[:ll]), n))
        inner.vt.contents.challenge_bytes(inner.m, label, ll, dest, n)
    vt = TranscriptVtbl(APPEND_FN(on_append), CHALLENGE_FN(on_challenge))
    try:
        proof = host.prove_with(dense, gens, S, r, (C.pointer(vt), None), tape.pair())
        assert proof == host.prove(dense, gens, S, r)
    finally:
        inner.close(); tape.close(); host.free(dense, gens)
    # surge.rs:127-199's schedule as merlin sees it: the protocol name first, then the commitment of E, the claim, the sumcheck's round polynomials and challenges, ...
    assert log[0] == ("append", b"protocol-name", len(b"Lasso SparsePolynomialEvaluationProof"))
    labels = [l for _, l, _ in log]
    for expected in (b"subtable_evals_commitment", b"comm_poly_row_col_ops_val", b"poly_commitment_share", b"claim_eval_scalar_product", b"poly", b"coeff", b"challenge_nextround", b"evals_ops_val",
                     b"challenge_combine_n_to_one", b"joint_claim_eval", b"Cx", b"Cy", b"a", b"L", b"R", b"u", b"delta", b"beta", b"c", b"challenge_r_hash", b"claim_hash_init",
                     b"rand_coeffs_next_layer", b"claim_prod_left", b"challenge_r_layer", b"claim_evals_ops", b"claim_evals_mem", b"challenge_combine_two_to_one"):
        assert expected in labels, expected
    assert all(n == 64 for k, _, n in log if k == "challenge")      # challenge_scalar draws 64 bytes (utils/transcript.rs:64-68)
    assert labels.index(b"challenge_r_hash") > labels.index(b"joint_claim_eval") > labels.index(b"challenge_nextround") > labels.index(b"claim_eval_scalar_product")
