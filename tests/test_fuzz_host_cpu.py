"""CPU: seeded differential fuzz of the host prover (over the mock of the device ABI) against the oracle prover, both curve builds: random strategy,
C in 1..4, table sizes 2^1..2^8, ragged lookup counts up to 5000, three index distributions (independent per dimension, the harness's replicated
draw, one address hit every time).  Commitment and proof bytes must be identical, the oracle verifier and the product-side verifier must accept, and the product-side
verifier must reject the proof with one random bit flipped.  The open-ended version of this
loop ran 10,000 configurations clean (tools/fuzz_host.py); the bitwise tables need an even log_m (an address splits into two operands of log_m / 2 bits —
with an odd log_m the reference's own MLE formula disagrees with its table, and its verifier rejects its own proof)."""
import ctypes as C

import numpy as np
import pytest

from lasso_amd import _abi
from proverutil import HostProver, OracleSession, build_mock_prover


def random_config(rng):
    kind = ["and", "or", "xor", "lt", "range"][rng.integers(5)]
    c = int(rng.integers(1, 5)); log_m = int(rng.integers(1, 9))
    if kind != "range" and log_m % 2:
        log_m += 1
    lookups = int(rng.integers(2, 700)) if rng.integers(4) else int(rng.integers(2, 5000))   # one lookup (s = 1) is outside the reference's domain: GrandProductCircuit::new needs two leaves
    log_r = int(rng.integers(1, c * log_m + 1)) if kind == "range" else 0
    mode = int(rng.integers(3))
    if mode == 0:
        idx = rng.integers(0, 1 << log_m, size=(lookups, c), dtype=np.uint64)
    elif mode == 1:
        idx = np.repeat(rng.integers(0, 1 << log_m, size=(lookups, 1), dtype=np.uint64), c, axis=1).copy()
    else:
        idx = np.full((lookups, c), int(rng.integers(0, 1 << log_m)), dtype=np.uint64)
    return kind, c, log_m, log_r, lookups, idx


def run(host, oracle, seed, count):
    rng = np.random.default_rng(seed)
    for _ in range(count):
        kind, c, log_m, log_r, lookups, idx = random_config(rng)
        s = 1 << max((lookups - 1).bit_length(), 0)
        bits = max(s.bit_length() - 1, 0)
        r = host.gen_random_point(max(bits, 1))[:bits]
        S = _abi.Strategy(_abi.KINDS[kind], c, log_m, log_r)
        tag = f"{kind} C={c} log_m={log_m} log_r={log_r} lookups={lookups}"
        gens = host.gens(c, s, 2 * c if kind == "lt" else c, log_m); dense = host.densify(idx, log_m)
        comm = host.commit(dense, gens); proof = host.prove(dense, gens, S, r)
        accepted = host.verify(gens, S, s, r, proof, comm)           # the product-side verifier on every fuzzed proof ...
        bad = bytearray(proof); pos = int(rng.integers(len(bad))); bad[pos] ^= 1 << int(rng.integers(8))
        try:
            rejected = host.verify(gens, S, s, r, bytes(bad), comm) is False      # ... and on one corrupted bit of it
        except Exception:
            rejected = True                                                        # bytes that no longer deserialize
        host.free(dense, gens)
        assert accepted is True, tag
        o = OracleSession(oracle, _abi.KINDS[kind], c, log_m, log_r, idx, r)
        try:
            assert comm == o.commit(), tag
            assert proof == o.prove(), tag
            assert o.verify(proof, comm) == 1, tag
            # One flipped bit is rejected — unless it lands on an encoding ark-serialize itself treats as equivalent (the sign bit of a point with x = 0, e.g. the
            # identity that commits to an all-zero row: `get_point_from_y_unchecked` returns the same point for either flag value and the transcript absorbs the
            # RE-serialised bytes); then the reference's verifier accepts too, and so must both verifiers here.  (Found by the open-ended fuzz: RangeCheck with
            # LOG_R < log_m has all-zero memories.)
            if not rejected:
                try:
                    also = o.verify(bytes(bad), comm) == 1
                except Exception:
                    also = False
                assert also, tag + f": corrupted byte {pos} accepted by the product verifier but not by the oracle's"
        finally:
            o.close()


def test_fuzz_curve25519(oracle):
    hp = HostProver(C.CDLL(build_mock_prover()))
    try:
        run(hp, oracle, 2024, 40)
    finally:
        hp.close()


def test_fuzz_bn254(oracle_bn254):
    hp = HostProver(C.CDLL(build_mock_prover("bn254")))
    try:
        run(hp, oracle_bn254, 2025, 40)
    finally:
        hp.close()
