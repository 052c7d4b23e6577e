"""smoke(): one small invocation of the hot path on cuda:0 (AND, C=1, M=2^8, 2^10 lookups), checked against the CPU oracle.
The oracle is used here only as the checker (allowed for __graft_entry__.smoke()); the proof itself comes from the HIP path."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def smoke():
    from . import HostProver, _abi
    hp = HostProver()                                    # raises if liblasso_hip.so / a GPU is missing: no fallback
    c, log_m, lookups = 1, 8, 1 << 10
    idx = hp.gen_indices(lookups, 1 << log_m, c)
    r = hp.gen_random_point(10)
    S = _abi.Strategy(_abi.KINDS["and"], c, log_m, 0)
    gens = hp.gens(c, lookups, c, log_m)
    dense = hp.densify(idx, log_m)
    comm = hp.commit(dense, gens)
    proof = hp.prove(dense, gens, S, r)
    hp.free(dense, gens)
    hp.close()
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    orc = C.CDLL(os.path.join(ROOT, "oracle", "liblasso_oracle.so"))
    orc.orc_last_error.restype = C.c_char_p
    orc.orc_session_new.restype = C.c_void_p
    from proverutil import OracleSession
    o = OracleSession(orc, 0, c, log_m, 0, idx, r)
    try:
        assert comm == o.commit(), "commitment differs from the oracle"
        assert proof == o.prove(), "proof differs from the oracle"
        assert o.verify(proof, comm) == 1, "oracle verifier rejected the GPU proof"
    finally:
        o.close()
    print(f"smoke ok: {len(proof)}-byte proof for 2^10 AND lookups, bit-identical to the oracle and verified")
