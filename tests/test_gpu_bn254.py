"""-m gpu: the BN254 build of the HIP path (liblasso_prover_bn254.so over liblasso_hip_bn254.so: the same kernels over ark-bn254's Fr / G1,
csrc/bn254_*.cuh + mont29.cuh) against the oracle's BN254 instantiation — G = BN254 is the group BASELINE.json's configs[1] names.
Commitment and proof bytes identical to the oracle prover's; at configs[1]'s full size the reference's own acceptance property
prove -> verify (src/e2e_test.rs:54-59) through the oracle verifier, tamper rejection and determinism."""
import ctypes as C

import numpy as np
import pytest

from lasso_amd import _abi
from proverutil import OracleSession

pytestmark = pytest.mark.gpu

CASES = [("lt", 4, 4, 0, 16), ("and", 4, 4, 0, 16), ("range", 3, 8, 40, 16), ("and", 1, 4, 0, 2), ("xor", 3, 4, 0, 11), ("or", 2, 4, 0, 8),
         ("and", 1, 16, 0, 1 << 10), ("and", 4, 16, 0, 1 << 12), ("xor", 8, 8, 0, 1 << 10), ("and", 1, 16, 0, 1 << 14),
         ("spark", 2, 8, 0, 300), ("spark", 4, 16, 0, 1 << 12), ("spark", 16, 4, 0, 64)]   # LASSO_SPARK_UNCONFIRMED over BN254


@pytest.fixture(scope="module")
def host():
    from lasso_amd import HostProver
    hp = HostProver(curve="bn254")            # product library (BN254 pair); raises if the extension or the GPU is missing
    yield hp
    hp.close()


@pytest.mark.parametrize("kind,c,log_m,log_r,lookups", CASES)
def test_gpu_bn254_proof_bit_exact_vs_oracle(host, oracle_bn254, kind, c, log_m, log_r, lookups):
    alpha = 2 * c if kind == "lt" else c
    s = 1 << max((lookups - 1).bit_length(), 0)
    idx = host.gen_indices(lookups, 1 << log_m, c)
    if (kind, c) == ("xor", 3):
        idx = np.random.default_rng(7).integers(0, 1 << log_m, size=(lookups, c), dtype=np.uint64)
    r = host.gen_random_point(max(s.bit_length() - 1, 0))
    S = _abi.Strategy(_abi.KINDS[kind], c, log_m, log_r)
    gens = host.gens(c, s, alpha, log_m)
    dense = host.densify(idx, log_m)
    comm = host.commit(dense, gens)
    proof = host.prove(dense, gens, S, r)
    proof2 = host.prove(dense, gens, S, r)
    accepted = host.verify(gens, S, s, r, proof, comm)      # product-side verifier, BN254 build
    host.free(dense, gens)
    assert proof == proof2
    assert accepted is True
    orc = OracleSession(oracle_bn254, _abi.KINDS[kind], c, log_m, log_r, idx, r)
    try:
        assert comm == orc.commit()
        assert proof == orc.prove()
        assert orc.verify(proof, comm) == 1
    finally:
        orc.close()


def test_gpu_bn254_baseline_config1_full_size(host, oracle_bn254):
    """BASELINE.json configs[1] as written: AND, C = 4, log_M = 16, 2^20 lookups, G = BN254."""
    kind, c, log_m, log_r, log_s = "and", 4, 16, 0, 20
    s = 1 << log_s
    idx = host.gen_indices(s, 1 << log_m, c)
    r = host.gen_random_point(log_s)
    S = _abi.Strategy(_abi.KINDS[kind], c, log_m, log_r)
    gens = host.gens(c, s, c, log_m)
    dense = host.densify(idx, log_m)
    comm = host.commit(dense, gens)
    import time
    proof = host.prove(dense, gens, S, r)
    t0 = time.perf_counter(); again = host.prove(dense, gens, S, r); dt = time.perf_counter() - t0
    host.free(dense, gens)
    assert proof == again
    print(f"\n[bn254] AND C=4 2^20 lookups: prove {dt * 1e3:.1f} ms ({s / dt:.3e} lookups/s), proof {len(proof)} bytes")
    rr = np.ascontiguousarray(r, dtype=np.uint64)
    o = oracle_bn254
    o.orc_verify_only.argtypes = [C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
    assert o.orc_verify_only(_abi.KINDS[kind], c, 1 << log_m, log_r, s, rr.ctypes.data_as(C.c_void_p), proof, len(proof), comm, len(comm)) == 1, o.orc_last_error()
    bad = bytearray(proof); bad[len(bad) // 2] ^= 0x04
    assert o.orc_verify_only(_abi.KINDS[kind], c, 1 << log_m, log_r, s, rr.ctypes.data_as(C.c_void_p), bytes(bad), len(bad), comm, len(comm)) != 1


def test_gpu_bn254_config1_bit_exact_at_full_size(host, oracle_bn254):
    """BASELINE.json configs[1] as written (AND, C = 4, log_M = 16, 2^20 lookups, G = BN254): commitment and proof BYTES identical to the oracle
    prover's on the harness inputs (the oracle runs on all host cores; bytes independent of the thread count)."""
    from proverutil import oracle_harness_proof
    kind, c, log_m, log_r, log_s = "and", 4, 16, 0, 20
    s = 1 << log_s
    idx = host.gen_indices(s, 1 << log_m, c)
    r = host.gen_random_point(log_s)
    S = _abi.Strategy(_abi.KINDS[kind], c, log_m, log_r)
    gens = host.gens(c, s, c, log_m)
    dense = host.densify(idx, log_m)
    comm = host.commit(dense, gens)
    proof = host.prove(dense, gens, S, r)
    host.free(dense, gens)
    o_comm, o_proof, tm = oracle_harness_proof(oracle_bn254, _abi.KINDS[kind], c, log_m, log_r, log_s)
    print(f"\n[oracle bn254] AND C=4 2^20: {tm['threads']} threads, commit {tm['commit_s']:.1f}s prove {tm['prove_s']:.1f}s")
    assert comm == o_comm
    assert proof == o_proof


def test_gpu_bn254_kernel_parity_suite():
    """Every entry point of the BN254 library against the BN254 mock: tests/test_gpu_kernels.py re-run in a child process with LASSO_TEST_CURVE=bn254
    (the switch is read at import time by tests/fieldref.py, so it cannot share this process)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LASSO_TEST_CURVE="bn254")
    res = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_kernels.py"), "-m", "gpu", "-x", "-q", "-p", "no:cacheprovider"],
                         cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-2000:]
    assert " passed" in res.stdout and "failed" not in res.stdout
