#!/usr/bin/env python3
"""Write the cross-implementation acceptance artefact of SURVEY.md §8(f4): the files an UNMODIFIED Rust `verify` reads.

    python tools/dump_proof.py OUT_DIR [--kind and --c 1 --log-m 16 --log-s 24 --curve curve25519] [--mock]

OUT_DIR/proof.bin       SparsePolynomialEvaluationProof<G, C, M, S>   ark-serialize `serialize_compressed` (surge.rs:92-104)
OUT_DIR/commitment.bin  SparsePolynomialCommitment<G>                 ark-serialize `serialize_compressed` (surge.rs:61-68): the two PolyCommitments, then s, log_m, m as u64
OUT_DIR/point.bin       the evaluation point r: log2(s) scalars, 32 canonical little-endian bytes each (`Vec<Fr>::serialize_compressed` minus its u64 length)
OUT_DIR/meta.json       strategy, C, M, s, curve, transcript label, generator label, sha256 of the three files, and the product verifier's verdict

The inputs are the reference harness's (benches/bench.rs:13-34).  The Rust side (INTEGRATION.md "Checking a dumped proof with the unmodified crate"):
    let proof = SparsePolynomialEvaluationProof::<G, C, M, S>::deserialize_compressed(&proof_bytes[..])?;
    let commitment = SparsePolynomialCommitment::<G>::deserialize_compressed(&commitment_bytes[..])?;
    let gens = SparsePolyCommitmentGens::<G>::new(b"gens_sparse_poly", C, s, S::NUM_MEMORIES, log_m);
    proof.verify(&commitment, &r, &gens, &mut Transcript::new(b"example"))?;
--mock runs the host prover over the oracle's CPU mock of the device ABI (no GPU; small sizes only) — how the committed example under tests/golden/ was made."""
import argparse
import ctypes as C
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def fr_to_canonical_bytes(r, curve):
    p = {"curve25519": 2**252 + 27742317777372353535851937790883648493,
         "bn254": 21888242871839275222246405745257275088548364400416034343698204186575808495617}[curve]
    rinv = pow(1 << 256, -1, p)
    out = b""
    for row in np.ascontiguousarray(r, dtype=np.uint64).reshape(-1, 4):
        out += (int.from_bytes(row.tobytes(), "little") * rinv % p).to_bytes(32, "little")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out_dir")
    ap.add_argument("--kind", default="and", choices=["and", "or", "xor", "lt", "range"])
    ap.add_argument("--c", type=int, default=1)
    ap.add_argument("--log-m", type=int, default=16)
    ap.add_argument("--log-r", type=int, default=40)
    ap.add_argument("--log-s", type=int, default=24)
    ap.add_argument("--curve", default="curve25519", choices=["curve25519", "bn254"])
    ap.add_argument("--mock", action="store_true")
    a = ap.parse_args()
    from lasso_amd import HostProver, _abi
    if a.mock:
        from proverutil import build_mock_prover
        hp = HostProver(C.CDLL(build_mock_prover(a.curve)))
    else:
        hp = HostProver(curve=a.curve)
    s = 1 << a.log_s
    alpha = 2 * a.c if a.kind == "lt" else a.c
    log_r = a.log_r if a.kind == "range" else 0
    S = _abi.Strategy(_abi.KINDS[a.kind], a.c, a.log_m, log_r)
    idx = hp.gen_indices(s, 1 << a.log_m, a.c)
    r = hp.gen_random_point(a.log_s)
    gens = hp.gens(a.c, s, alpha, a.log_m)
    dense = hp.densify(idx, a.log_m)
    comm = hp.commit(dense, gens)
    proof = hp.prove(dense, gens, S, r)
    ok = hp.verify(gens, S, s, r, proof, comm)
    hp.free(dense, gens); hp.close()
    commitment_file = comm + s.to_bytes(8, "little") + a.log_m.to_bytes(8, "little") + (1 << a.log_m).to_bytes(8, "little")
    point_file = fr_to_canonical_bytes(r, a.curve)
    os.makedirs(a.out_dir, exist_ok=True)
    files = {"proof.bin": proof, "commitment.bin": commitment_file, "point.bin": point_file}
    for name, data in files.items():
        with open(os.path.join(a.out_dir, name), "wb") as f:
            f.write(data)
    strategy_type = {"and": "AndSubtableStrategy", "or": "OrSubtableStrategy", "xor": "XorSubtableStrategy", "lt": "LTSubtableStrategy", "range": f"RangeCheckSubtableStrategy<{log_r}>"}[a.kind]
    meta = {"strategy": a.kind, "rust_strategy_type": strategy_type, "C": a.c, "M": 1 << a.log_m, "log_m": a.log_m, "log_r": log_r, "s": s, "num_memories": alpha,
            "curve": a.curve, "G": "ark_curve25519::EdwardsProjective" if a.curve == "curve25519" else "ark_bn254::G1Projective",
            "transcript_label": "example", "gens_label": "gens_sparse_poly", "inputs": "benches/bench.rs:13-34 (gen_indices / gen_random_point from a fresh ark_std::test_rng())",
            "sha256": {k: hashlib.sha256(v).hexdigest() for k, v in files.items()}, "bytes": {k: len(v) for k, v in files.items()},
            "product_verifier_accepts": bool(ok), "backend": "oracle mock of the device ABI (CPU)" if a.mock else "liblasso_hip (MI355X)"}
    with open(os.path.join(a.out_dir, "meta.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print(json.dumps(meta))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
