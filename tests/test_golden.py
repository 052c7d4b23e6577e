"""Committed fixtures (tests/golden/proofs.json, made by tests/golden/make_golden.py from the oracle): the oracle still reproduces them (CPU), the C++
host prover over the mock reproduces them (CPU), and the HIP path reproduces them (-m gpu).  They pin the three against drift; they do NOT pin the
oracle to the Rust binary, which cannot be built here and publishes no proof bytes (parity unpinned)."""
import ctypes as C
import hashlib
import json
import os
import sys

import numpy as np
import pytest

from lasso_amd import _abi
from proverutil import HostProver, OracleSession, build_mock_prover

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from make_golden import KINDS, instance  # noqa: E402

with open(os.path.join(HERE, "golden", "proofs.json")) as f:
    GOLDEN = json.load(f)
IDS = [f'{g["kind"]}-C{g["c"]}-m{g["log_m"]}-n{g["lookups"]}' for g in GOLDEN]


def _check(g, comm, proof):
    assert len(comm) == g["commitment_len"] and hashlib.sha256(comm).hexdigest() == g["commitment_sha256"]
    assert len(proof) == g["proof_len"] and proof[:64].hex() == g["proof_head"] and hashlib.sha256(proof).hexdigest() == g["proof_sha256"]


@pytest.mark.parametrize("g", GOLDEN, ids=IDS)
def test_oracle_reproduces_golden(oracle, g):
    idx, r = instance(oracle, g["kind"], g["c"], g["log_m"], g["lookups"], g["seed"])
    o = OracleSession(oracle, KINDS[g["kind"]], g["c"], g["log_m"], g["log_r"], idx, r)
    try:
        _check(g, o.commit(), o.prove())
    finally:
        o.close()


def _prove(hp, oracle, g):
    idx, r = instance(oracle, g["kind"], g["c"], g["log_m"], g["lookups"], g["seed"])
    s = 1 << max((g["lookups"] - 1).bit_length(), 0)
    alpha = 2 * g["c"] if g["kind"] == "lt" else g["c"]
    S = _abi.Strategy(_abi.KINDS[g["kind"]], g["c"], g["log_m"], g["log_r"])
    gens = hp.gens(g["c"], s, alpha, g["log_m"]); dense = hp.densify(idx, g["log_m"])
    comm = hp.commit(dense, gens); proof = hp.prove(dense, gens, S, r)
    hp.free(dense, gens)
    return comm, proof


@pytest.mark.parametrize("g", GOLDEN, ids=IDS)
def test_host_prover_over_mock_reproduces_golden(oracle, g):
    hp = HostProver(C.CDLL(build_mock_prover()))
    try:
        _check(g, *_prove(hp, oracle, g))
    finally:
        hp.close()


@pytest.mark.gpu
@pytest.mark.parametrize("g", GOLDEN, ids=IDS)
def test_hip_path_reproduces_golden(oracle, g):
    from lasso_amd import HostProver as HipProver
    hp = HipProver()          # product library: raises if the extension or the GPU is missing
    try:
        _check(g, *_prove(hp, oracle, g))
    finally:
        hp.close()


# ---- the same three-way pin for the BN254 build (tests/golden/proofs_bn254.json: oracle -DORC_BN254, host prover + mock -DLASSO_BN254, liblasso_*_bn254.so)
with open(os.path.join(HERE, "golden", "proofs_bn254.json")) as f:
    GOLDEN_BN254 = json.load(f)
IDS_BN254 = [f'bn254-{g["kind"]}-C{g["c"]}-m{g["log_m"]}-n{g["lookups"]}' for g in GOLDEN_BN254]


@pytest.mark.parametrize("g", GOLDEN_BN254, ids=IDS_BN254)
def test_bn254_oracle_reproduces_golden(oracle_bn254, g):
    idx, r = instance(oracle_bn254, g["kind"], g["c"], g["log_m"], g["lookups"], g["seed"])
    o = OracleSession(oracle_bn254, KINDS[g["kind"]], g["c"], g["log_m"], g["log_r"], idx, r)
    try:
        _check(g, o.commit(), o.prove())
    finally:
        o.close()


@pytest.mark.parametrize("g", GOLDEN_BN254, ids=IDS_BN254)
def test_bn254_host_prover_over_mock_reproduces_golden(oracle_bn254, g):
    hp = HostProver(C.CDLL(build_mock_prover("bn254")))
    try:
        _check(g, *_prove(hp, oracle_bn254, g))
    finally:
        hp.close()


@pytest.mark.gpu
@pytest.mark.parametrize("g", GOLDEN_BN254, ids=IDS_BN254)
def test_bn254_hip_path_reproduces_golden(oracle_bn254, g):
    from lasso_amd import HostProver as HipProver
    hp = HipProver(curve="bn254")
    try:
        _check(g, *_prove(hp, oracle_bn254, g))
    finally:
        hp.close()
