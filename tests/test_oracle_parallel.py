"""The oracle's OpenMP loops (oracle/par.hpp) are the reference's rayon sites restated; exact field / group arithmetic means the
bytes cannot depend on the thread count.  Held here as a test: commitment and proof at 1 thread == at 3 == at all threads, for every
strategy family, and the parallel prover's proofs still verify.  (The -m gpu suite relies on this to prove the metric-size instance,
2^24 lookups, on all host cores and compare it byte for byte with the GPU's proof.)"""
import ctypes as C

import pytest

from lasso_amd import _abi

CASES = [("and", 1, 16, 0, 13), ("xor", 3, 8, 0, 13), ("lt", 2, 8, 0, 12), ("range", 2, 8, 12, 13)]


def _run(oracle, kind, c, log_m, log_r, log_s, threads, verify):
    oracle.orc_set_threads(threads)
    td, tc, tp = C.c_double(), C.c_double(), C.c_double()
    cap = 1 << 21
    pb = (C.c_uint8 * cap)(); cb = (C.c_uint8 * cap)(); pl = C.c_size_t(); cl = C.c_size_t()
    rc = oracle.orc_bench_bytes(_abi.KINDS[kind], C.c_size_t(c), C.c_size_t(1 << log_m), C.c_size_t(log_r), C.c_size_t(1 << log_s), C.byref(td), C.byref(tc), C.byref(tp),
                                1 if verify else 0, pb, C.c_size_t(cap), C.byref(pl), cb, C.c_size_t(cap), C.byref(cl))
    assert rc == 0, oracle.orc_last_error()
    return bytes(pb[: pl.value]), bytes(cb[: cl.value])


@pytest.mark.parametrize("kind,c,log_m,log_r,log_s", CASES)
def test_bytes_independent_of_thread_count(oracle, kind, c, log_m, log_r, log_s):
    nmax = max(2, min(8, oracle.orc_max_threads()))
    try:
        ref = _run(oracle, kind, c, log_m, log_r, log_s, 1, verify=False)
        for threads in (3, nmax):
            assert _run(oracle, kind, c, log_m, log_r, log_s, threads, verify=(threads == nmax)) == ref
    finally:
        oracle.orc_set_threads(nmax)
