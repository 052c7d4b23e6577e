"""-m gpu: the full north-star path on the MI355X — from_lookup_indices -> commit -> prove through liblasso_prover.so /
liblasso_hip.so — against the CPU oracle: commitment bytes and proof bytes identical, oracle verifier accepts; at sizes the
oracle prover cannot reach in seconds, the size-independent property prove -> verify (the reference's own acceptance test,
src/e2e_test.rs:54-59) through the oracle verifier with the GPU's commitment."""
import numpy as np
import pytest

from lasso_amd import _abi
from proverutil import OracleSession, cubic_batched_case

pytestmark = pytest.mark.gpu

CASES = [("lt", 4, 4, 0, 16), ("lt", 4, 4, 0, 128), ("and", 4, 4, 0, 16), ("range", 3, 8, 40, 16),   # e2e_test.rs:64-99
         ("and", 1, 4, 0, 2), ("xor", 3, 4, 0, 11), ("or", 2, 4, 0, 8), ("and", 1, 16, 0, 1 << 10),   # BASELINE config 1 shape
         ("and", 1, 2, 0, 3), ("or", 1, 6, 0, 7), ("lt", 1, 4, 0, 4), ("range", 2, 4, 6, 9), ("and", 3, 2, 0, 4),
         ("and", 4, 16, 0, 1 << 12), ("xor", 8, 8, 0, 1 << 10), ("range", 4, 16, 40, 1 << 10), ("lt", 2, 8, 0, 1 << 9), ("and", 1, 16, 0, 1 << 14),
         ("lt", 16, 4, 0, 64), ("lt", 16, 8, 0, 1 << 10),   # C = 16: 32 memories, degree-17 sumcheck — the shape of BASELINE.json configs[4] with the one degree-C strategy the snapshot has
         ("spark", 1, 4, 0, 16), ("spark", 2, 8, 0, 300), ("spark", 3, 6, 0, 1 << 10), ("spark", 4, 16, 0, 1 << 12), ("spark", 8, 8, 0, 1 << 11), ("spark", 16, 4, 0, 64),
         ("spark", 16, 16, 0, 1 << 12)]   # configs[4] under its own name: SparkSubtableStrategy as SURVEY 8(f3) describes it (LASSO_SPARK_UNCONFIRMED: eq tables, product combine, degree C)


@pytest.fixture(scope="module")
def host():
    from lasso_amd import HostProver
    hp = HostProver()            # product library; raises if the extension or the GPU is missing
    yield hp
    hp.close()


def _instance(host, kind, c, log_m, lookups, seed=None):
    s = 1 << max((lookups - 1).bit_length(), 0)
    idx = host.gen_indices(lookups, 1 << log_m, c)
    if seed is not None:
        idx = np.random.default_rng(seed).integers(0, 1 << log_m, size=(lookups, c), dtype=np.uint64)
    r = host.gen_random_point(max(s.bit_length() - 1, 0))
    return s, idx, r


@pytest.mark.parametrize("kind,c,log_m,log_r,lookups", CASES)
def test_gpu_proof_bit_exact_vs_oracle(host, oracle, kind, c, log_m, log_r, lookups):
    alpha = 2 * c if kind == "lt" else c
    s, idx, r = _instance(host, kind, c, log_m, lookups, seed=7 if (kind, c) == ("xor", 3) else None)
    S = _abi.Strategy(_abi.KINDS[kind], c, log_m, log_r)
    gens = host.gens(c, s, alpha, log_m)
    dense = host.densify(idx, log_m)
    comm = host.commit(dense, gens)
    proof = host.prove(dense, gens, S, r)
    proof2 = host.prove(dense, gens, S, r)      # prove() must not consume its inputs: a second proof is identical
    accepted = host.verify(gens, S, s, r, proof, comm)      # the product-side verifier (surge.rs:214-271; its two MSMs per opening run on the device)
    bad = bytearray(proof); bad[len(bad) // 3] ^= 0x20
    try:
        tampered = host.verify(gens, S, s, r, bytes(bad), comm)
    except Exception:
        tampered = False
    host.free(dense, gens)
    assert proof == proof2
    assert accepted is True and tampered is False
    orc = OracleSession(oracle, _abi.KINDS[kind], c, log_m, log_r, idx, r)
    try:
        assert comm == orc.commit()
        assert proof == orc.prove()
        assert orc.verify(proof, comm) == 1
    finally:
        orc.close()


@pytest.mark.parametrize("kind,c,log_m,log_r,log_s", [("and", 1, 16, 0, 18), ("and", 4, 16, 0, 16), ("xor", 8, 16, 0, 14)])
def test_gpu_proof_verifies_at_scale(host, oracle, kind, c, log_m, log_r, log_s):
    lookups = 1 << log_s
    s, idx, r = _instance(host, kind, c, log_m, lookups)
    S = _abi.Strategy(_abi.KINDS[kind], c, log_m, log_r)
    gens = host.gens(c, s, c, log_m)
    dense = host.densify(idx, log_m)
    comm = host.commit(dense, gens)
    proof = host.prove(dense, gens, S, r)
    host.free(dense, gens)
    orc = OracleSession(oracle, _abi.KINDS[kind], c, log_m, log_r, idx, r)
    try:
        assert orc.verify(proof, comm) == 1
        bad = bytearray(proof); bad[40] ^= 0x10        # a corrupted commitment share must be rejected
        try:
            assert orc.verify(bytes(bad), comm) != 1
        except Exception:
            pass
    finally:
        orc.close()


def test_gpu_concurrent_proofs_identical_to_sequential():
    """Four provers (own context, stream, mapped result area and mailbox each) proving at once on the one device, one host thread each — bench.py's concurrent leg and a serving
    deployment: every proof must be the bytes the same prover produces alone.  Resident tail kernels of different contexts run side by side here."""
    import threading
    from lasso_amd import HostProver
    kind, c, log_m, log_s, T, reps = "and", 2, 16, 16, 4, 3
    S = _abi.Strategy(_abi.KINDS[kind], c, log_m, 0)
    workers = []
    for t in range(T):
        hp = HostProver()
        idx = (hp.gen_indices(1 << log_s, 1 << log_m, c) + 17 * t) % (1 << log_m)
        r = hp.gen_random_point(log_s)
        gens = hp.gens(c, 1 << log_s, c, log_m); dense = hp.densify(idx, log_m)
        workers.append((hp, dense, gens, r, hp.prove(dense, gens, S, r)))
    got, errors = [[] for _ in range(T)], []
    bar = threading.Barrier(T)

    def run(t):
        hp, dense, gens, r, _ = workers[t]
        try:
            bar.wait(timeout=60)
            for _ in range(reps):
                got[t].append(hp.prove(dense, gens, S, r))
        except Exception as e:
            errors.append(repr(e)); bar.abort()
    ths = [threading.Thread(target=run, args=(t,)) for t in range(T)]
    for th in ths:
        th.start()
    for th in ths:
        th.join(timeout=120)
    for hp, dense, gens, r, _ in workers:
        hp.free(dense, gens); hp.close()
    assert not errors, errors
    for t in range(T):
        assert len(got[t]) == reps and all(p == workers[t][4] for p in got[t])


def test_gpu_soak_alternating_shapes_on_one_prover(host):
    """Soak: 240 proofs on ONE prover, alternating four instances of different shapes and sizes (so every proof finds the buffer pool, the scratch areas, the generator tables' caches,
    the hand-off sequence numbers and the launched-ahead windows as ANOTHER shape left them).  Every proof must be the bytes its instance produced the first time; every 40th is
    also put through the product-side verifier.  The launched-ahead protocols (rounds, layers, bullet rounds, the opening's tail chain) are all on this path."""
    shapes = [("and", 1, 16, 0, 1 << 14), ("xor", 4, 8, 0, 1 << 12), ("lt", 2, 8, 0, 1 << 10), ("range", 2, 16, 40, 3000)]
    inst = []
    for kind, c, log_m, log_r, lookups in shapes:
        alpha = 2 * c if kind == "lt" else c
        s_, idx, r = _instance(host, kind, c, log_m, lookups, seed=1000 + lookups)
        S = _abi.Strategy(_abi.KINDS[kind], c, log_m, log_r)
        gens = host.gens(c, s_, alpha, log_m); dense = host.densify(idx, log_m)
        comm = host.commit(dense, gens)
        inst.append((S, s_, r, gens, dense, comm, host.prove(dense, gens, S, r)))
    try:
        import os
        for it in range(int(os.environ.get("LASSO_SOAK_ITERS", "240"))):   # LASSO_SOAK_ITERS=20000: the long form (profiles/r06_soak.txt)
            S, s_, r, gens, dense, comm, first = inst[(it * 7 + it // 4) % len(inst)]
            if it % 3 == 0:
                assert host.commit(dense, gens) == comm, f"commitment {it} differs"
            proof = host.prove(dense, gens, S, r)
            assert proof == first, f"proof {it} differs from its instance's first proof"
            if it % 40 == 0:
                assert host.verify(gens, S, s_, r, proof, comm) is True
    finally:
        for S, s_, r, gens, dense, comm, first in inst:
            host.free(dense, gens)


_SWITCH_SCRIPT = """
import hashlib, sys
import numpy as np
from lasso_amd import HostProver, _abi
hp = HostProver()
kind, c, log_m, log_s = "and", 2, 16, 14
S = _abi.Strategy(_abi.KINDS[kind], c, log_m, 0)
idx = hp.gen_indices(1 << log_s, 1 << log_m, c); r = hp.gen_random_point(log_s)
gens = hp.gens(c, 1 << log_s, c, log_m); dense = hp.densify(idx, log_m)
comm = hp.commit(dense, gens); proof = hp.prove(dense, gens, S, r)
print("DIGEST", hashlib.sha256(comm + proof).hexdigest())
"""


@pytest.mark.parametrize("env", [{"LASSO_TAGGED_RESULTS": "0"}, {"LASSO_DIRECT_NX": "0"}, {"LASSO_TAGGED_RESULTS": "0", "LASSO_CUBIC_TAIL": "0", "LASSO_LINEAR_TAIL": "0"},
                                 {"LASSO_EQ_INLINE": "0", "LASSO_SUMCHECK_U32": "0", "LASSO_MSM_FUSED": "0"}, {"LASSO_TAIL_Q": "256", "LASSO_LB_PIPELINE": "0"},
                                 {"LASSO_SEQ_START": "0xffefff9c"},     # the context's hand-off sequence numbers start 100 short of the end of an epoch: the proof crosses next_seq's restart
                                 {"LASSO_MSM_DIRECT8": "0"},                 # the openings' MSMs over the 4-bit digit-multiple tables instead of the byte-multiple ones
                                 {"LASSO_MSM_DIRECT8": "0", "LASSO_MSM_FUSED": "0"},
                                 {"LASSO_BULLET_AHEAD": "0"},                # every bullet round launched after its challenge is known (round 3's schedule)
                                 {"LASSO_MSM_DIRECT": "0"},                  # the bucket kernel serves the openings' MSMs too (incl. the deferred delta MSM)
                                 {"LASSO_CAPACITY": "1", "LASSO_LEAFLESS_MIN": "1024"},      # capacity mode: leafless trees, the bottom layer's two streaming rounds from recomputed leaves (8 chunks)
                                 {"LASSO_CAPACITY": "1", "LASSO_LEAFLESS_MIN": "1024", "LASSO_CUBIC_TAIL": "0", "LASSO_EQ_INLINE": "0"},
                                 {"LASSO_ROUNDS_AHEAD": "0"},                # round 5: no sumcheck round / resident tail is enqueued before its challenge is known
                                 {"LASSO_HOST_TAIL": "0"},                   # round 5: the host finishes no tail and proves no tree-top layer (every round through the device, as in round 4)
                                 {"LASSO_HOST_TAIL": "0", "LASSO_ROUNDS_AHEAD": "0", "LASSO_BULLET_AHEAD": "0"},
                                 {"LASSO_HOST_TAIL": "128"},                 # ... and takes over four times earlier than the default (arrays of 32 elements at two circuits... here of 32 / 16)
                                 {"LASSO_HOST_TAIL": "8", "LASSO_TAGGED_RESULTS": "0"},      # hand-over through the flag protocol
                                 {"LASSO_ROUNDS_AHEAD": "1", "LASSO_CUBIC_TAIL": "0"},      # every round of a layer launched ahead (no resident tail to end in)
                                 {"LASSO_CUBIC_THREE_SUMS": "1"},            # three sums per round from the device — and every streaming layer enqueued ahead is cancelled (lasso_point_cancel), then started the plain way
                                 {"LASSO_HOST_IFMA": "0"},                   # round 5: the host's rounds by the scalar loop (and its smaller take-over size)
                                 {"LASSO_HOST_IFMA": "0", "LASSO_HOST_TAIL": "128"}, {"LASSO_HOST_IFMA": "1", "LASSO_HOST_TAIL": "512"},
                                 {"LASSO_LAYER_AHEAD": "0"},                 # round 5: no layer's first launch is enqueued during the previous layer
                                 {"LASSO_LAYER_AHEAD": "0", "LASSO_EQ_INLINE_BIG": "1"}, {"LASSO_EQ_INLINE_BIG": "1"},      # the tables above 2^14 entries formed inside round 0 (measured, not the default)
                                 {"LASSO_HOST_TAIL": "0", "LASSO_LAYER_AHEAD": "1"},          # layers enqueued ahead behind tails that run to the heads
                                 {"LASSO_ROUNDS_AHEAD": "0", "LASSO_LAYER_AHEAD": "1", "LASSO_TAGGED_RESULTS": "0"},
                                 {"LASSO_CAPACITY": "1", "LASSO_LEAFLESS_MIN": "1024", "LASSO_HOST_TAIL": "0"},
                                 {"LASSO_MSM_TAGGED": "0"},                  # round 6: the few-row MSMs' points through the flag protocol (one lane converts, system fence, ticket, flag)
                                 {"LASSO_BULLET_TAIL_AHEAD": "0"},           # round 6: the openings' last fold, heads and delta MSM as separate calls instead of one chain enqueued ahead
                                 {"LASSO_BULLET_TAIL_AHEAD": "0", "LASSO_MSM_TAGGED": "0"},
                                 {"LASSO_MSM_FULL8": "1"},                   # round 6: full-width commitments over the signed byte-multiple table instead of the bucket kernel (measured, not the default)
                                 {"LASSO_MSM_ROWS8W_WAVES": "2048"}, {"LASSO_MSM_ROWS8W": "0"}])   # round 6: several rows per wave in the one-wave-per-row commitment (measured, not the default); the 256-lane form
def test_gpu_ab_switches_do_not_change_the_bytes(host, env):
    """The A/B switches the measurements in DESIGN.md rest on (flag protocol instead of tagged results, in-launch second stage, launch per round instead of the resident tails, ...)
    select other kernels / protocols for the same arithmetic: commitment and proof must be the bytes of the default configuration.  Each setting runs in its own process
    (the switches are read once per process)."""
    import hashlib, os, subprocess, sys
    kind, c, log_m, log_s = "and", 2, 16, 14
    S = _abi.Strategy(_abi.KINDS[kind], c, log_m, 0)
    idx = host.gen_indices(1 << log_s, 1 << log_m, c); r = host.gen_random_point(log_s)
    gens = host.gens(c, 1 << log_s, c, log_m); dense = host.densify(idx, log_m)
    want = hashlib.sha256(host.commit(dense, gens) + host.prove(dense, gens, S, r)).hexdigest()
    host.free(dense, gens)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ); e.update(env); e["PYTHONPATH"] = root + os.pathsep + e.get("PYTHONPATH", "")
    out = subprocess.run([sys.executable, "-c", _SWITCH_SCRIPT], env=e, cwd=root, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    got = [l.split()[1] for l in out.stdout.splitlines() if l.startswith("DIGEST")]
    assert got == [want], (env, got, want)


# BASELINE.json's configurations at FULL size (configs[1], configs[2] and the configuration the metric is quoted on).  The oracle prover cannot
# reach these sizes in seconds, so parity rests on the size-independent property the reference itself uses as its acceptance test
# (src/e2e_test.rs:54-59): prove -> verify, here through the oracle's verifier (a restatement of surge.rs:214-271) fed the GPU's commitment,
# plus rejection of a tampered proof and determinism of the proof bytes.
FULL = [("and", 4, 16, 0, 20), ("and", 1, 16, 0, 24), ("xor", 8, 16, 0, 24), ("range", 4, 16, 40, 26), ("and", 1, 16, 0, 28), ("lt", 16, 16, 0, 22), ("spark", 16, 16, 0, 20)]   # configs[1], the metric, configs[2], configs[3] (on one GPU: ~75 GiB), the largest lookup count of BASELINE.json (2^28, ~75 GiB) with the AND table, and LT C=16 (32 memories, degree 17: configs[4]'s shape, see CASES)


@pytest.mark.parametrize("kind,c,log_m,log_r,log_s", FULL)
def test_baseline_config_full_size(host, oracle, kind, c, log_m, log_r, log_s):
    import ctypes as C
    s = 1 << log_s
    idx = host.gen_indices(s, 1 << log_m, c)
    r = host.gen_random_point(log_s)
    S = _abi.Strategy(_abi.KINDS[kind], c, log_m, log_r)
    gens = host.gens(c, s, 2 * c if kind == "lt" else c, log_m)
    dense = host.densify(idx, log_m)
    del idx
    comm = host.commit(dense, gens)
    proof = host.prove(dense, gens, S, r)
    again = host.prove(dense, gens, S, r) if log_s <= 20 or c == 1 else proof
    host.free(dense)
    import time
    t0 = time.time(); accepted = host.verify(gens, S, s, r, proof, comm); tv = time.time() - t0      # product-side verifier at full size
    badp = bytearray(proof); badp[len(badp) // 2] ^= 0x04
    try:
        tampered = host.verify(gens, S, s, r, bytes(badp), comm)
    except Exception:
        tampered = False
    host.free(None, gens)
    print(f"\n[verify] {kind} C={c} 2^{log_s}: product verifier {tv * 1e3:.0f} ms, proof {len(proof)} B")
    assert proof == again
    # byte parity at full size against the ORACLE's digests for this harness instance (recorded by tools/parity_full_configs.py, which ran the oracle
    # prover at this size on the GPU box's host cores): configs[2] and configs[3] of BASELINE.json
    import hashlib, json, os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "full_config_digests.json")) as f:
        gold = json.load(f).get(f"{kind},{c},{log_m},{log_r},{log_s}")
    if gold:
        assert hashlib.sha256(comm).hexdigest() == gold["oracle"]["commitment_sha256"]
        assert hashlib.sha256(proof).hexdigest() == gold["oracle"]["proof_sha256"]
    assert accepted is True and tampered is False
    rr = np.ascontiguousarray(r, dtype=np.uint64)
    oracle.orc_verify_only.argtypes = [C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
    ok = oracle.orc_verify_only(_abi.KINDS[kind], c, 1 << log_m, log_r, s, rr.ctypes.data_as(C.c_void_p), proof, len(proof), comm, len(comm))
    assert ok == 1, oracle.orc_last_error()
    bad = bytearray(proof); bad[len(bad) // 2] ^= 0x04
    assert oracle.orc_verify_only(_abi.KINDS[kind], c, 1 << log_m, log_r, s, rr.ctypes.data_as(C.c_void_p), bytes(bad), len(bad), comm, len(comm)) != 1


# Byte-for-byte parity AT the sizes the claims are made on (VERDICT r1 "What's weak" #1): the oracle proves the same harness instance on all host
# cores (OpenMP over the reference's rayon sites; bytes independent of the thread count, tests/test_oracle_parallel.py) and the GPU's commitment and
# proof must be identical.  ("and", 1, 16, 0, 24) is the configuration BASELINE.json's metric is quoted on.
# ("xor", 8, 16, 0, 24) = BASELINE.json configs[2] at full size (oracle: ~85 s on the GPU box's 128 cores).  configs[3] (RangeCheck C=4 2^26, oracle ~150 s) is
# held to the digests tools/parity_full_configs.py recorded from the oracle at that size (tests/golden/full_config_digests.json) in test_baseline_config_full_size.
# ("spark", 4, 16, 0, 20): E is 2048 rows x 2048 columns of FULL-WIDTH scalars — the shape from which the commitment runs the 12-bit-window bucket kernels (k_msm_pip_*, round 6).
AT_SIZE = [("and", 1, 16, 0, 24), ("xor", 2, 16, 0, 22), ("range", 2, 16, 40, 21), ("lt", 1, 16, 0, 20), ("xor", 8, 16, 0, 24), ("spark", 4, 16, 0, 20)]


@pytest.mark.parametrize("kind,c,log_m,log_r,log_s", AT_SIZE)
def test_gpu_proof_bit_exact_at_baseline_size(host, oracle, kind, c, log_m, log_r, log_s):
    import time
    from proverutil import oracle_harness_proof
    s = 1 << log_s
    alpha = 2 * c if kind == "lt" else c
    idx = host.gen_indices(s, 1 << log_m, c)
    r = host.gen_random_point(log_s)
    S = _abi.Strategy(_abi.KINDS[kind], c, log_m, log_r)
    gens = host.gens(c, s, alpha, log_m)
    dense = host.densify(idx, log_m)
    del idx
    comm = host.commit(dense, gens)
    proof = host.prove(dense, gens, S, r)
    host.free(dense, gens)
    t0 = time.time()
    o_comm, o_proof, tm = oracle_harness_proof(oracle, _abi.KINDS[kind], c, log_m, log_r, log_s)
    print(f"\n[oracle] {kind} C={c} 2^{log_s}: {tm['threads']} threads, densify {tm['densify_s']:.1f}s commit {tm['commit_s']:.1f}s prove {tm['prove_s']:.1f}s (wall {time.time() - t0:.1f}s); "
          f"proof {len(proof)} B, commitment {len(comm)} B")
    assert comm == o_comm
    assert proof == o_proof


# Slab mode on the real device: ONE proof sharded over P ranks (include/lasso_prover.h lasso_host_set_comm).  The GPU box has one MI355X, so the P ranks
# are P contexts (own stream, own buffers) on the same device driven by P threads, with a shared-memory all-gather (tests/cpp/slab_threads.cpp) — the
# kernels, the slab variants (densify / fingerprints / scaled eq tables), the row-commitment exchange and the tails are the ones an N-GPU run executes.
def _build_slab_hip():
    import ctypes as C
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out_dir = os.path.join(root, "tests", "_build"); os.makedirs(out_dir, exist_ok=True)
    so = os.path.join(out_dir, "libslab_threads_hip.so")
    srcs = [os.path.join(root, "tests", "cpp", "slab_threads.cpp"), os.path.join(root, "lasso_amd", "host", "prover_capi.cpp")]
    lib_dir = os.path.join(root, "lasso_amd")
    if not os.path.exists(so) or any(os.path.getmtime(x) > os.path.getmtime(so) for x in srcs + [os.path.join(lib_dir, "liblasso_hip.so"), os.path.join(lib_dir, "host", "prover.hpp")]):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-Wno-unknown-pragmas", "-o", so] + srcs + ["-L" + lib_dir, "-llasso_hip", "-Wl,-rpath," + lib_dir])
    return C.CDLL(so)


@pytest.mark.parametrize("world", [2, 4, 8])
@pytest.mark.parametrize("kind,c,log_m,log_r,lookups", [("and", 1, 8, 0, 1 << 10), ("xor", 3, 6, 0, 500), ("lt", 2, 6, 0, 64), ("range", 3, 8, 40, 100), ("and", 2, 16, 0, 1 << 13),
                                                          ("spark", 4, 8, 0, 1 << 10)])
def test_gpu_slab_proof_bit_exact(host, oracle, world, kind, c, log_m, log_r, lookups):
    import ctypes as C
    lib = _build_slab_hip()
    s = 1 << (lookups - 1).bit_length()
    alpha = 2 * c if kind == "lt" else c
    idx = np.ascontiguousarray(np.random.default_rng(world * 77 + lookups).integers(0, 1 << log_m, size=(lookups, c), dtype=np.uint64))
    r = np.ascontiguousarray(host.gen_random_point(s.bit_length() - 1), dtype=np.uint64)
    S = _abi.Strategy(_abi.KINDS[kind], c, log_m, log_r)
    comm = (C.c_uint8 * (1 << 22))(); proof = (C.c_uint8 * (1 << 22))(); cl = C.c_size_t(); pl = C.c_size_t(); nc = C.c_size_t(); nb = C.c_size_t(); err = C.create_string_buffer(512)
    rc = lib.slab_prove_threads(world, C.byref(S), C.c_size_t(alpha), idx.ctypes.data_as(C.c_void_p), C.c_size_t(lookups), r.ctypes.data_as(C.c_void_p), C.c_size_t(r.shape[0]),
                                comm, C.c_size_t(len(comm)), C.byref(cl), proof, C.c_size_t(len(proof)), C.byref(pl), C.byref(nc), C.byref(nb), err, C.c_size_t(512))
    assert rc == 0, err.value.decode()
    comm_p, proof_p = bytes(comm[: cl.value]), bytes(proof[: pl.value])
    orc = OracleSession(oracle, _abi.KINDS[kind], c, log_m, log_r, idx, r)
    try:
        assert comm_p == orc.commit()
        assert proof_p == orc.prove()
        assert orc.verify(proof_p, comm_p) == 1
    finally:
        orc.close()


def run_slab_threads(lib, world, S, alpha, idx, r, shm_name=None, capacity=False, steps=1):
    """ONE proof over `world` contexts of the one device (tests/cpp/slab_threads.cpp): (commitment, proof, info)"""
    import ctypes as C
    idx = np.ascontiguousarray(idx, dtype=np.uint64); r = np.ascontiguousarray(r, dtype=np.uint64)
    comm = (C.c_uint8 * (1 << 23))(); proof = (C.c_uint8 * (1 << 23))(); cl = C.c_size_t(); pl = C.c_size_t(); nc = C.c_size_t(); nb = C.c_size_t(); err = C.create_string_buffer(512)
    peak = (C.c_uint64 * world)(); used = (C.c_uint64 * world)(); ms = C.c_double()
    rc = lib.slab_prove_threads_ex(world, C.byref(S), C.c_size_t(alpha), idx.ctypes.data_as(C.c_void_p), C.c_size_t(idx.shape[0]), r.ctypes.data_as(C.c_void_p), C.c_size_t(r.shape[0]),
                                   shm_name.encode() if shm_name else None, 1 if capacity else 0, steps, comm, C.c_size_t(len(comm)), C.byref(cl), proof, C.c_size_t(len(proof)), C.byref(pl),
                                   C.byref(nc), C.byref(nb), peak, used, C.byref(ms), err, C.c_size_t(512))
    assert rc == 0, err.value.decode()
    return bytes(comm[: cl.value]), bytes(proof[: pl.value]), {"peak_bytes_per_rank": list(peak), "prover_peak_bytes_per_rank": list(used), "ms_per_proof": ms.value}


# BASELINE.json configs[3] (RangeCheck, C=4, 2^26 lookups — the configuration the north star shards over 4 GPUs) as ONE proof over P = 4 and P = 8 ranks, at FULL size, through the
# transport an N-GPU run uses (lasso_host_set_comm_shm: the library's shared-memory exchange; RCCL declines when ranks share a device, so the partial row commitments take the
# same route).  The box has one MI355X: the P ranks are P contexts on it (own stream, buffers, generator tables; ~75 GiB in total), each rank holding 1/P of every polynomial.
# Commitment and proof must be the bytes the ORACLE produced for this instance (tests/golden/full_config_digests.json, re-derived by test_oracle_rederives_config3_digest).
@pytest.mark.parametrize("world,capacity", [(4, False), (8, True)])
def test_gpu_slab_config3_full_size_one_proof_over_P_ranks(host, world, capacity):
    import hashlib, json, os
    kind, c, log_m, log_r, log_s = "range", 4, 16, 40, 26
    lib = _build_slab_hip()
    s = 1 << log_s
    idx = host.gen_indices(s, 1 << log_m, c); r = host.gen_random_point(log_s)
    S = _abi.Strategy(_abi.KINDS[kind], c, log_m, log_r)
    comm, proof, info = run_slab_threads(lib, world, S, c, idx, r, shm_name=f"/lasso_test_cfg3_{os.getpid()}_{world}", capacity=capacity, steps=2)
    del idx
    print(f"\n[slab] configs[3] over {world} ranks on one device (capacity mode {capacity}): {info['ms_per_proof']:.1f} ms per proof, peak bytes per rank {max(info['peak_bytes_per_rank']) / 2**30:.2f} GiB "
          f"(prover in use at most {max(info['prover_peak_bytes_per_rank']) / 2**30:.2f} GiB)")
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "full_config_digests.json")) as f:
        gold = json.load(f)[f"{kind},{c},{log_m},{log_r},{log_s}"]["oracle"]
    assert hashlib.sha256(comm).hexdigest() == gold["commitment_sha256"]
    assert hashlib.sha256(proof).hexdigest() == gold["proof_sha256"]


def test_oracle_rederives_config3_digest(oracle):
    """The committed digests of configs[3] (tests/golden/full_config_digests.json) are the ORACLE's: re-derive them in-tree — the oracle prover at RangeCheck C=4 2^26 on one thread
    per physical core (~150 s on the GPU box's 128 cores) — so that the constant the GPU is held to at 2^26 cannot drift from the oracle (src/e2e_test.rs:17-62 is the reference's
    own acceptance shape).  Marked gpu because only the GPU box has the cores and the memory; it makes no device call."""
    import hashlib, json, os
    from proverutil import oracle_harness_proof
    kind, c, log_m, log_r, log_s = "range", 4, 16, 40, 26
    if (os.cpu_count() or 1) < 32:
        pytest.skip("needs the GPU box's host cores (the oracle proves 2^26 lookups)")
    o_comm, o_proof, tm = oracle_harness_proof(oracle, _abi.KINDS[kind], c, log_m, log_r, log_s)
    print(f"\n[oracle] configs[3]: {tm['threads']} threads, densify {tm['densify_s']:.1f}s commit {tm['commit_s']:.1f}s prove {tm['prove_s']:.1f}s")
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "full_config_digests.json")) as f:
        gold = json.load(f)[f"{kind},{c},{log_m},{log_r},{log_s}"]["oracle"]
    assert hashlib.sha256(o_comm).hexdigest() == gold["commitment_sha256"]
    assert hashlib.sha256(o_proof).hexdigest() == gold["proof_sha256"]


@pytest.mark.parametrize("k,ell,special", [(1, 1, {}), (2, 3, {}), (3, 5, {0: 0}), (2, 5, {2: 0}), (2, 5, {1: 1}), (1, 6, {0: 0, 1: 1, 2: 0, 5: 1}), (2, 8, {7: 0}), (33, 2, {1: 0}),
                                          (2, 12, {}), (2, 12, {0: 1, 3: 0, 11: 0}), (4, 14, {5: 0})])
def test_gpu_cubic_batched_scripted_eq_points(host, oracle, k, ell, special):
    """the two-sum rounds, the three-sum fallback (rand_t = 0) and the explicit-table path (rand_t = 1) on the device kernels, all sizes
    (latency-shaped and streaming kernels), against the oracle's literal loop"""
    cubic_batched_case(host, oracle, k, ell, special, seed=k * 100 + ell)


@pytest.mark.parametrize("kind,c,log_m,log_r,log_s", [("and", 2, 16, 0, 14), ("lt", 2, 8, 0, 10)])
def test_gpu_live_transcript_and_tape_through_the_callbacks(host, oracle, kind, c, log_m, log_r, log_s):
    """lasso_host_prove_cb on the device (surge.rs:119-125's `&mut Transcript`, `&mut RandomTape`): fresh Merlin objects behind the callbacks give the label path's bytes, and a
    transcript the caller has already written to gives the ORACLE's proof for the same pre-seeded transcript (tests/test_transcript_callbacks_cpu.py holds the host logic to
    more cases over the mock)."""
    import ctypes as C
    from lasso_amd.prover import Transcript
    s = 1 << log_s
    alpha = 2 * c if kind == "lt" else c
    idx = host.gen_indices(s, 1 << log_m, c); r = host.gen_random_point(log_s)
    S = _abi.Strategy(_abi.KINDS[kind], c, log_m, log_r)
    gens = host.gens(c, s, alpha, log_m); dense = host.densify(idx, log_m)
    t = Transcript(host.lib, b"example"); tape = Transcript(host.lib, b"proof", tape=True)
    t2 = Transcript(host.lib, b"example"); tape2 = Transcript(host.lib, b"proof", tape=True)
    pre_label, pre_msg = b"outer protocol", b"absorbed before prove"
    t2.append_message(pre_label, pre_msg)
    orc = OracleSession(oracle, _abi.KINDS[kind], c, log_m, log_r, idx, r)
    try:
        assert host.prove_with(dense, gens, S, r, t.pair(), tape.pair()) == host.prove(dense, gens, S, r)
        seeded = host.prove_with(dense, gens, S, r, t2.pair(), tape2.pair())
        oracle.orc_session_prove_seeded.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_size_t, C.c_char_p, C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        buf = (C.c_uint8 * (1 << 22))(); n = C.c_size_t()
        assert oracle.orc_session_prove_seeded(C.c_void_p(orc.s), pre_label, pre_msg, len(pre_msg), None, None, 0, buf, len(buf), C.byref(n)) == 0, oracle.orc_last_error()
        assert seeded == bytes(buf[: n.value])
    finally:
        orc.close()
        for x in (t, tape, t2, tape2):
            x.close()
        host.free(dense, gens)


def test_gpu_c99_program_through_the_c_abi(tmp_path):
    """examples/prove_c_abi.c linked against the PRODUCT libraries (liblasso_prover.so -> liblasso_hip.so): the boundary driven from plain C on the device — densify, commit,
    prove (labels and callbacks: identical bytes), verify."""
    import os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib_dir = os.path.join(root, "lasso_amd"); exe = str(tmp_path / "prove_c_abi")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I" + os.path.join(root, "include"), os.path.join(root, "examples", "prove_c_abi.c"), "-o", exe,
                           "-L" + lib_dir, "-llasso_prover", "-Wl,-rpath," + lib_dir])
    res = subprocess.run([exe, "14"], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "verified 1" in res.stdout and "callback path identical 1" in res.stdout


def test_gpu_slab_spark_c16_one_proof_over_8_ranks(host):
    """BASELINE.json configs[4]'s shape — the (restated, unconfirmed) Spark strategy, C = 16 — as ONE proof over 8 ranks in capacity mode, at the largest size that is quick on
    one device (2^22 lookups: 2^26 field-element table reads, a degree-17 sumcheck over 16 memories, 32 product trees): the 8 ranks (contexts of the one MI355X, the library's
    shared-memory exchange) must produce the bytes the single-GPU prover produces, which test_gpu_proof_bit_exact_vs_oracle holds to the oracle at the sizes the oracle reaches."""
    import os
    kind, c, log_m, log_s = "spark", 16, 16, 22
    lib = _build_slab_hip()
    s = 1 << log_s
    idx = host.gen_indices(s, 1 << log_m, c); r = host.gen_random_point(log_s)
    S = _abi.Strategy(_abi.KINDS[kind], c, log_m, 0)
    gens = host.gens(c, s, c, log_m); dense = host.densify(idx, log_m)
    comm1 = host.commit(dense, gens); proof1 = host.prove(dense, gens, S, r)
    host.free(dense, gens)
    comm, proof, info = run_slab_threads(lib, 8, S, c, idx, r, shm_name=f"/lasso_test_spark16_{os.getpid()}", capacity=True, steps=1)
    print(f"\n[slab] spark C=16 2^{log_s} over 8 ranks on one device (capacity mode): peak bytes per rank {max(info['peak_bytes_per_rank']) / 2**30:.2f} GiB")
    assert comm == comm1 and proof == proof1
