"""CPU, world_size = 2, gloo: the N > 1 path of bench.py — proofs sharded across ranks (one independent batch per rank), barrier +
max-over-ranks timing, digest gather — with the oracle's mock standing in for the GPUs.  Every rank's proof must verify."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent('''
    import ctypes as C, os, sys, time
    sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
    import numpy as np
    from lasso_amd import _abi
    from lasso_amd.parallel import Group, shard_indices
    from lasso_amd.prover import HostProver
    from proverutil import OracleSession, build_mock_prover
    grp = Group(backend="gloo")
    assert grp.world == 2
    hp = HostProver(C.CDLL(build_mock_prover()))
    c, log_m, s = 2, 4, 32
    idx = shard_indices(hp, s, 1 << log_m, c, grp.rank)
    r = hp.gen_random_point(5)
    S = _abi.Strategy(_abi.KINDS["and"], c, log_m, 0)
    gens = hp.gens(c, s, c, log_m); dense = hp.densify(idx, log_m)
    comm = hp.commit(dense, gens)
    grp.barrier(); t0 = time.perf_counter()
    proof = hp.prove(dense, gens, S, r)
    el = time.perf_counter() - t0; grp.barrier()
    mx = grp.max_over_ranks(el); total = grp.sum_over_ranks(s)
    assert mx >= el and total == 2 * s
    digests = grp.gather_digests(proof)
    assert len(digests) == 2 and digests[0] != digests[1]          # different batches -> different proofs
    orc = C.CDLL(os.path.join(%(root)r, "oracle", "liblasso_oracle.so")); orc.orc_last_error.restype = C.c_char_p; orc.orc_session_new.restype = C.c_void_p
    o = OracleSession(orc, 0, c, log_m, 0, idx, r)
    assert o.verify(proof, comm) == 1 and proof == o.prove()
    o.close(); hp.free(dense, gens); hp.close(); grp.close()
    print("rank", grp.rank, "ok")
''')


SLAB_WORKER = textwrap.dedent('''
    import ctypes as C, os, sys
    sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
    import numpy as np
    from lasso_amd import _abi
    from lasso_amd.parallel import Group
    from lasso_amd.prover import HostProver
    from proverutil import OracleSession, build_mock_prover
    grp = Group(backend="gloo")
    assert grp.world == 2
    hp = HostProver(C.CDLL(build_mock_prover()))
    hp.set_comm(grp)                                   # ONE proof over both ranks: torch.distributed all_gather is the only collective
    c, log_m, lookups = 2, 6, 100
    s = 128
    idx = np.random.default_rng(5).integers(0, 1 << log_m, size=(lookups, c), dtype=np.uint64)     # the SAME lookups on every rank
    r = hp.gen_random_point(7)
    S = _abi.Strategy(_abi.KINDS["xor"], c, log_m, 0)
    gens = hp.gens(c, s, c, log_m); dense = hp.densify(idx, log_m)
    comm = hp.commit(dense, gens)
    proof = hp.prove(dense, gens, S, r)
    digests = grp.gather_digests(proof)
    assert digests[0] == digests[1]                    # replicated transcript -> identical proof bytes on every rank
    orc = C.CDLL(os.path.join(%(root)r, "oracle", "liblasso_oracle.so")); orc.orc_last_error.restype = C.c_char_p; orc.orc_session_new.restype = C.c_void_p
    o = OracleSession(orc, _abi.KINDS["xor"], c, log_m, 0, idx, r)
    assert comm == o.commit() and proof == o.prove() and o.verify(proof, comm) == 1
    o.close(); hp.free(dense, gens); hp.close(); grp.close()
    print("rank", grp.rank, "ok")
''')


def _run_two(tmp_path, body, port):
    from proverutil import build_mock_prover
    build_mock_prover()
    script = tmp_path / "worker.py"
    script.write_text(body % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(rk), LOCAL_RANK=str(rk)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for rk in range(2)]
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    for rk, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert f"rank {rk} ok" in o


def test_one_proof_over_two_ranks_gloo(tmp_path, oracle):
    """slab mode through the real collective plumbing (lasso_amd.parallel.Group.allgather_callback over gloo): proof == oracle proof on both ranks"""
    _run_two(tmp_path, SLAB_WORKER, 29519)


def test_two_ranks_gloo(tmp_path, oracle):
    from proverutil import build_mock_prover
    build_mock_prover()
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29517", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(rk), LOCAL_RANK=str(rk)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for rk in range(2)]
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    for rk, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert f"rank {rk} ok" in o


SHM_WORKER = textwrap.dedent('''
    import ctypes as C, hashlib, os, sys
    sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
    import numpy as np
    from lasso_amd import _abi
    from lasso_amd.prover import HostProver
    from proverutil import OracleSession, build_mock_prover
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    hp = HostProver(C.CDLL(build_mock_prover()))
    hp.set_comm_shm(rank, world, os.environ["LASSO_SHM_NAME"])      # ONE proof over all ranks; the exchange is the library's own shared-memory all-gather
    kind, c, log_m, lookups = %(kind)r, %(c)d, %(log_m)d, %(lookups)d
    s = 1 << (lookups - 1).bit_length()
    alpha = 2 * c if kind == "lt" else c
    idx = np.random.default_rng(11).integers(0, 1 << log_m, size=(lookups, c), dtype=np.uint64)     # the SAME lookups on every rank
    r = hp.gen_random_point(s.bit_length() - 1)
    S = _abi.Strategy(_abi.KINDS[kind], c, log_m, 0)
    gens = hp.gens(c, s, alpha, log_m); dense = hp.densify(idx, log_m)
    comm = hp.commit(dense, gens)
    proof = hp.prove(dense, gens, S, r)
    proof2 = hp.prove(dense, gens, S, r)
    assert proof == proof2
    orc = C.CDLL(os.path.join(%(root)r, "oracle", "liblasso_oracle.so")); orc.orc_last_error.restype = C.c_char_p; orc.orc_session_new.restype = C.c_void_p
    o = OracleSession(orc, _abi.KINDS[kind], c, log_m, 0, idx, r)
    assert comm == o.commit() and proof == o.prove() and o.verify(proof, comm) == 1
    o.close(); hp.free(dense, gens); hp.close()
    print("rank", rank, "ok", hashlib.sha256(proof).hexdigest())
''')


import pytest  # noqa: E402


@pytest.mark.parametrize("world,kind,c,log_m,lookups", [(2, "and", 2, 6, 100), (4, "xor", 1, 8, 1 << 10), (4, "lt", 2, 6, 64), (8, "and", 1, 8, 1 << 9)])
def test_one_proof_over_ranks_native_shm_exchange(tmp_path, oracle, world, kind, c, log_m, lookups):
    """slab mode over lasso_host_set_comm_shm (lasso_amd/host/shm_comm.hpp): P processes, no torch, no callback — commitment and proof identical to the
    oracle's on every rank, and identical between ranks"""
    from proverutil import build_mock_prover
    build_mock_prover()
    script = tmp_path / "worker.py"
    script.write_text(SHM_WORKER % {"root": ROOT, "kind": kind, "c": c, "log_m": log_m, "lookups": lookups})
    env = dict(os.environ, WORLD_SIZE=str(world), LASSO_SHM_NAME=f"/lasso_test_{os.getpid()}_{world}_{kind}", OMP_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(rk)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for rk in range(world)]
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    digests = set()
    for rk, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert f"rank {rk} ok" in o
        digests.add(o.strip().split()[-1])
    assert len(digests) == 1


@pytest.mark.parametrize("mode", ["full", "partial", "world", "size"])
def test_shm_exchange_survives_a_stale_segment(tmp_path, mode):
    """a segment left behind by a crashed run must not capture a rank of the next run — whether it is fully attached, was abandoned before every rank attached
    (its creator's pid is gone), or belongs to a run of another world / slot size: tests/cpp/test_shm_stale.cpp starts rank 1 against the stale segment and rank 0 150 ms later;
    both must end up in the fresh segment and complete an all-gather."""
    import subprocess
    exe = str(tmp_path / "test_shm_stale")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-pthread", "-o", exe, os.path.join(ROOT, "tests", "cpp", "test_shm_stale.cpp"), "-lrt"])
    res = subprocess.run([exe, f"/lasso_test_stale_{os.getpid()}_{mode}", mode], capture_output=True, text=True, timeout=60)
    assert res.returncode == 0 and "OK" in res.stdout, res.stdout + res.stderr
