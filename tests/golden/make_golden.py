#!/usr/bin/env python3
"""Generates tests/golden/proofs.json (G = curve25519) and proofs_bn254.json (G = BN254 G1, the oracle built with -DORC_BN254): commitment and proof bytes (hex SHA-256 + lengths + the first 64 proof bytes) of the ORACLE prover for fixed,
seeded instances.  The reference is a Rust crate that cannot be built in this image and holds no golden vector for any commitment or proof byte
(SURVEY.md §8c), so these fixtures pin the oracle against its own drift, not against the Rust binary ("parity unpinned").
Regenerate with:  python tests/golden/make_golden.py   (needs only the CPU oracle)."""
import ctypes as C
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

KINDS = {"and": 0, "or": 1, "xor": 2, "lt": 3, "range": 4, "spark": 5}   # spark = LASSO_SPARK_UNCONFIRMED (round 4): pins the restated strategy against its own drift, like the rest
# (kind, C, log_m, log_r, lookups, seed): seed None = the bench harness's inputs (benches/bench.rs:13-34), else numpy default_rng(seed) indices
INSTANCES = [("and", 1, 16, 0, 1 << 10, None), ("and", 4, 4, 0, 16, None), ("lt", 4, 4, 0, 128, None), ("range", 3, 8, 40, 16, None),
             ("xor", 3, 4, 0, 11, 7), ("or", 2, 6, 0, 50, 3), ("and", 1, 8, 0, 300, 11), ("spark", 3, 6, 0, 100, 5), ("spark", 16, 4, 0, 64, None)]


def instance(orc, kind, c, log_m, lookups, seed):
    s = 1 << max((lookups - 1).bit_length(), 0)
    if seed is None:
        one = np.empty(lookups, dtype=np.uint64)
        orc.orc_gen_indices(C.c_size_t(lookups), C.c_size_t(1 << log_m), one.ctypes.data_as(C.c_void_p))
        idx = np.repeat(one[:, None], c, axis=1).copy()
    else:
        idx = np.random.default_rng(seed).integers(0, 1 << log_m, size=(lookups, c), dtype=np.uint64)
    bits = max(s.bit_length() - 1, 0)
    r = np.empty((max(bits, 1), 4), dtype=np.uint64)
    orc.orc_gen_random_point(C.c_size_t(bits), r.ctypes.data_as(C.c_void_p))
    return idx, r[:bits]


def main():
    import conftest
    from proverutil import OracleSession
    for build, name in ((conftest._build_oracle, "proofs.json"), (conftest._build_oracle_bn254, "proofs_bn254.json")):   # G = curve25519, G = BN254 G1
        make(OracleSession, C.CDLL(build()), name)


def make(OracleSession, orc, name):
    orc.orc_last_error.restype = C.c_char_p; orc.orc_session_new.restype = C.c_void_p
    out = []
    for kind, c, log_m, log_r, lookups, seed in INSTANCES:
        idx, r = instance(orc, kind, c, log_m, lookups, seed)
        o = OracleSession(orc, KINDS[kind], c, log_m, log_r, idx, r)
        comm, proof = o.commit(), o.prove()
        assert o.verify(proof, comm) == 1
        o.close()
        out.append({"kind": kind, "c": c, "log_m": log_m, "log_r": log_r, "lookups": lookups, "seed": seed,
                    "commitment_len": len(comm), "commitment_sha256": hashlib.sha256(comm).hexdigest(),
                    "proof_len": len(proof), "proof_sha256": hashlib.sha256(proof).hexdigest(), "proof_head": proof[:64].hex()})
    with open(os.path.join(HERE, name), "w") as f:
        json.dump(out, f, indent=1)
    print(f"{name}: wrote {len(out)} instances")


if __name__ == "__main__":
    main()
