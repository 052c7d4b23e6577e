#!/usr/bin/env python3
"""Byte parity GPU vs oracle at BASELINE.json's full-size configurations (VERDICT r2 "Next round" 1, row N1).

For every configuration named on the command line (default: configs[2] = XOR C=8 2^24 and configs[3] = RangeCheck C=4 2^26) the harness instance
(src/benches/bench.rs:13-34 inputs) is densified, committed and proved on the MI355X through liblasso_prover.so, and by the oracle prover
(oracle/, OpenMP on all physical cores); the sha256 of both commitments and both proofs are written SIDE BY SIDE to the output JSON, with
`equal` per pair.  The committed copy (tests/golden/full_config_digests.json, mirrored under profiles/) is what tests/test_gpu_prover.py and bench.py's
slab leg compare later proofs of the same instance with — the oracle's bytes do not depend on anything in lasso_amd/.

  python tools/parity_full_configs.py OUT.json [kind,c,log_m,log_r,log_s ...]
"""
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

DEFAULT = [("xor", 8, 16, 0, 24), ("range", 4, 16, 40, 26)]


def sha(b):
    return hashlib.sha256(b).hexdigest()


def main():
    import ctypes
    import subprocess
    from lasso_amd import HostProver, _abi
    from proverutil import oracle_harness_proof
    out_path = sys.argv[1]
    cfgs = [tuple(int(x) if i else x for i, x in enumerate(a.split(","))) for a in sys.argv[2:]] or DEFAULT
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "liblasso_oracle.so"])
    orc = ctypes.CDLL(os.path.join(ROOT, "oracle", "liblasso_oracle.so"))
    orc.orc_last_error.restype = ctypes.c_char_p
    hp = HostProver()
    res = {}
    for kind, c, log_m, log_r, log_s in cfgs:
        s = 1 << log_s
        alpha = 2 * c if kind == "lt" else c
        S = _abi.Strategy(_abi.KINDS[kind], c, log_m, log_r)
        idx = hp.gen_indices(s, 1 << log_m, c)
        r = hp.gen_random_point(log_s)
        gens = hp.gens(c, s, alpha, log_m)
        dense = hp.densify(idx, log_m)
        del idx
        comm = hp.commit(dense, gens)
        hp.prove(dense, gens, S, r)
        t0 = time.perf_counter(); proof = hp.prove(dense, gens, S, r); t_gpu = time.perf_counter() - t0
        hp.free(dense, gens)
        t0 = time.time()
        o_comm, o_proof, tm = oracle_harness_proof(orc, _abi.KINDS[kind], c, log_m, log_r, log_s)
        key = f"{kind},{c},{log_m},{log_r},{log_s}"
        res[key] = {"workload": f"{kind.upper()} subtable, C={c}, M=2^{log_m}, s=2^{log_s} lookups" + (f", LOG_R={log_r}" if kind == "range" else ""),
                    "gpu": {"commitment_sha256": sha(comm), "proof_sha256": sha(proof), "commitment_bytes": len(comm), "proof_bytes": len(proof), "prove_ms": round(t_gpu * 1e3, 2)},
                    "oracle": {"commitment_sha256": sha(o_comm), "proof_sha256": sha(o_proof), "commitment_bytes": len(o_comm), "proof_bytes": len(o_proof), "threads": tm["threads"],
                               "densify_s": round(tm["densify_s"], 2), "commit_s": round(tm["commit_s"], 2), "prove_s": round(tm["prove_s"], 2), "wall_s": round(time.time() - t0, 1)},
                    "commitment_equal": comm == o_comm, "proof_equal": proof == o_proof}
        print(key, json.dumps(res[key]), flush=True)
        with open(out_path, "w") as f:
            json.dump(res, f, indent=1)
    hp.close()
    return 0 if all(v["commitment_equal"] and v["proof_equal"] for v in res.values()) else 1


if __name__ == "__main__":
    sys.exit(main())
