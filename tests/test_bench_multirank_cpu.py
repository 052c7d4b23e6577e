"""CPU: `python bench.py --gpus 2` outside any launcher spawns its own two ranks (torch.distributed.run on 127.0.0.1; gloo, because this box has fewer GPUs than
ranks), prints ONE JSON line with n_gpus = 2, the per-GPU-proof (weak) value AND the slab leg (ONE proof over both ranks, strong) — VERDICT r2 "What's missing" 4.
The product libraries need a GPU, so the harness is pointed (LASSO_PROVER_LIB / LASSO_DEVICE_LIB) at the host-prover sources linked against the test mock of the
device ABI: what is exercised is bench.py's own multi-rank plumbing, the shared-memory exchange and the slab prover's host logic — not a kernel."""
import json
import os
import subprocess
import sys

import numpy as np

from proverutil import OracleSession, build_mock_prover

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_gpus2_self_launch_prints_weak_and_slab(oracle):
    so = build_mock_prover()
    env = dict(os.environ, LASSO_PROVER_LIB=so, LASSO_DEVICE_LIB=so, OMP_NUM_THREADS="2")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--log-s", "8", "--log-m", "8", "--no-cpu-baseline", "--concurrent", "0",
           "--slab-kind", "range", "--slab-c", "2", "--slab-log-s", "7", "--slab-steps", "1", "--log-r", "12", "--slab-timeout", "240"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]                      # rank 0 prints, once
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["steps"] == 1
    assert out["config"]["distinct_proofs"] == 2                    # one independent proof per rank
    assert out["config"]["launched_by"].startswith("bench.py --gpus N itself")
    assert out["value"] > 0 and out["lib_sha"]["sources_sha256"]
    slab = out["slab_mode"]
    assert slab.get("n_gpus") == 2 and slab["scaling"] == "strong", slab
    assert slab["capacity_mode"]["same_bytes_as_pooled"] is True      # the leg's second pass (lasso_host_set_capacity) proves the same bytes
    assert slab["rccl_ranks"] == 0                                   # no GPU per rank here: the rows travel through the shared-memory exchange, consistently on both ranks
    # the sharded proof is the single-prover proof of the same instance: compare with the oracle
    from lasso_amd import _abi
    from lasso_amd.prover import HostProver
    import ctypes as C
    import hashlib
    hp = HostProver(lib=C.CDLL(so))
    idx = hp.gen_indices(1 << 7, 1 << 8, 2); r = hp.gen_random_point(7)
    hp.close()
    orc = OracleSession(oracle, _abi.KINDS["range"], 2, 8, 12, idx, r)
    try:
        assert hashlib.sha256(orc.prove()).hexdigest() == slab["proof_sha256"]
        assert hashlib.sha256(orc.commit()).hexdigest() == slab["commitment_sha256"]
    finally:
        orc.close()


def test_concurrent_leg_reports_a_failed_proof_instead_of_hanging():
    """bench.py's concurrent-proofs leg with a prover that fails in one of its threads: the leg must come back with an `error` entry (round 3: a lost hand-off in an
    experiment left the main thread waiting at the barrier until the outer time-out, and with it the whole bench line)."""
    import importlib.util
    import threading
    import time
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    from lasso_amd import _abi
    made = []

    class FakeProver:
        def __init__(self, curve="curve25519"):
            self.calls = 0; self.id = len(made); made.append(self)
        def set_throughput_mode(self, on=True): self.throughput = on
        def gen_indices(self, s, m, c): return np.zeros((s, c), dtype=np.uint64)
        def gen_random_point(self, n): return np.zeros((n, 4), dtype=np.uint64)
        def gens(self, c, s, alpha, log_m): return object()
        def densify(self, idx, log_m): return object()
        def prove(self, dense, gens, S, r):
            self.calls += 1
            if self.id == 1 and self.calls == 2:   # the second prover's first timed proof (call 1 is the warm-up)
                raise RuntimeError("a result was not delivered by the device")
            time.sleep(0.01); return b"proof"
        def free(self, *a): pass
        def close(self): pass
    S = _abi.Strategy(_abi.KINDS["and"], 1, 8, 0)
    box = {}
    th = threading.Thread(target=lambda: box.update(out=bench.concurrent_leg(FakeProver, _abi, 3, 2, S, 1, 8, 6)), daemon=True)
    th.start(); th.join(timeout=60)
    assert not th.is_alive(), "the leg hung"
    assert "error" in box["out"] and "not delivered" in box["out"]["error"]



def test_concurrent_sweep_checks_every_proof_against_its_sequential_bytes():
    """bench.py's throughput sweep: streams in {2, 4, 8}, every proof of every stream compared with the bytes the same prover produced alone; a stream whose
    concurrent proof differs is reported, and the best round is the headline of the leg."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test2", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    from lasso_amd import _abi
    made = []

    class FakeProver:
        flaky = False
        def __init__(self, curve="curve25519"):
            self.calls = 0; self.id = len(made); made.append(self)
        def set_throughput_mode(self, on=True): self.throughput = on
        def gen_indices(self, s, m, c): return np.zeros((s, c), dtype=np.uint64)
        def gen_random_point(self, n): return np.zeros((n, 4), dtype=np.uint64)
        def gens(self, c, s, alpha, log_m): return object()
        def densify(self, idx, log_m): return object()
        def prove(self, dense, gens, S, r):
            self.calls += 1
            import time as _t; _t.sleep(0.002)
            return b"other" if (FakeProver.flaky and self.id == 3 and self.calls == 3) else b"proof%d" % self.id
        def free(self, *a): pass
        def close(self): pass
    S = _abi.Strategy(_abi.KINDS["and"], 1, 8, 0)
    out = bench.concurrent_leg(FakeProver, _abi, 8, 2, S, 1, 8, 6, kernel_ms_per_proof=1.0)
    assert [x["streams"] for x in out["sweep"]] == [2, 4, 8] and out["all_proofs_identical_to_sequential"] is True
    assert out["value"] == max(x["value"] for x in out["sweep"]) and out["device_serial_bound"]["lookups_per_s_if_kernels_never_overlap"] == 64 / 1e-3
    made.clear(); FakeProver.flaky = True
    out = bench.concurrent_leg(FakeProver, _abi, 4, 2, S, 1, 8, 6)
    assert out["all_proofs_identical_to_sequential"] is False and sum(x["proofs_differing_from_sequential"] for x in out["sweep"]) == 1


def test_bench_gpus2_shard_proof_is_one_proof_over_both_ranks(oracle):
    """`bench.py --gpus 2 --shard-proof`: the timed proof itself is ONE proof over the two ranks (slab mode as `value`, "scaling": "strong") — both ranks must hold the same
    bytes, and they are the oracle's for that instance."""
    import hashlib
    so = build_mock_prover()
    env = dict(os.environ, LASSO_PROVER_LIB=so, LASSO_DEVICE_LIB=so, OMP_NUM_THREADS="2")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--shard-proof", "--steps", "1", "--warmup", "0", "--kind", "xor", "--c", "2", "--log-s", "8", "--log-m", "6",
           "--no-cpu-baseline", "--concurrent", "0", "--no-slab-leg", "--no-prof"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["config"]["distinct_proofs"] == 1 and out["value"] > 0
    assert out["config"]["per_rank"].startswith("one proof sharded")
    mg = out["multi_gpu"]
    assert mg["ranks"] == 2 and mg["mode"].startswith("one proof sharded") and mg["rccl_ranks"] == 0 and len(mg["peak_bytes_per_rank"]) == 2 and "shared-memory" in mg["exchange"]


def _oracle_digest(oracle, so, kind, c, log_m, log_r, log_s):
    import ctypes as C
    import hashlib
    from lasso_amd import _abi
    from lasso_amd.prover import HostProver
    hp = HostProver(lib=C.CDLL(so))
    idx = hp.gen_indices(1 << log_s, 1 << log_m, c); r = hp.gen_random_point(log_s)
    hp.close()
    orc = OracleSession(oracle, _abi.KINDS[kind], c, log_m, log_r, idx, r)
    try:
        return hashlib.sha256(orc.prove()).hexdigest()
    finally:
        orc.close()


def test_bench_world8_shard_proof_pooled_and_capacity(oracle):
    """The first real 8-GPU run will be `bench.py --gpus 8 [--shard-proof]`: the same command at world = 8 here — eight gloo ranks, ONE proof sharded over
    them — in the pooled mode and in capacity mode (LASSO_CAPACITY=1, leafless trees from 64 lookups per rank on), both byte-identical on all eight ranks, and the line
    describes itself: ranks, exchange, per-rank peak bytes.  The slab leg of the default (independent-proof) line at world = 8 is covered by its world = 2 twin above; this is
    the strong-scaling form."""
    import hashlib
    so = build_mock_prover()
    digests = []
    for cap in (False, True):
        env = dict(os.environ, LASSO_PROVER_LIB=so, LASSO_DEVICE_LIB=so, OMP_NUM_THREADS="1")
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
            env.pop(k, None)
        if cap:
            env.update(LASSO_CAPACITY="1", LASSO_LEAFLESS_MIN="64")
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--shard-proof", "--steps", "1", "--warmup", "0", "--kind", "range", "--c", "2", "--log-s", "10", "--log-m", "8",
               "--log-r", "12", "--no-cpu-baseline", "--concurrent", "0", "--no-slab-leg", "--no-prof"]
        res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
        assert res.returncode == 0, res.stderr[-3000:]
        lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, res.stdout[-2000:]
        out = json.loads(lines[0])
        assert out["n_gpus"] == 8 and out["scaling"] == "strong" and out["config"]["distinct_proofs"] == 1 and out["value"] > 0
        mg = out["multi_gpu"]
        assert mg["ranks"] == 8 and len(mg["peak_bytes_per_rank"]) == 8 and all(b > 0 for b in mg["peak_bytes_per_rank"]) and mg["rccl_ranks"] == 0
        digests.append(out["config"].get("proof_sha256"))
    want = _oracle_digest(oracle, so, "range", 2, 8, 12, 10)
    assert digests[0] == digests[1] == want, (digests, want)


def test_bench_world8_independent_proofs_line_is_self_describing():
    """`bench.py --gpus 8` (the driver's scaling command): eight independent proofs, weak scaling, and the multi_gpu object of the line"""
    so = build_mock_prover()
    env = dict(os.environ, LASSO_PROVER_LIB=so, LASSO_DEVICE_LIB=so, OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0", "--log-s", "7", "--log-m", "6", "--no-cpu-baseline", "--concurrent", "0", "--no-slab-leg", "--no-prof"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    out = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 8 and out["scaling"] == "weak" and out["config"]["distinct_proofs"] == 8
    mg = out["multi_gpu"]
    assert mg["ranks"] == 8 and mg["mode"].startswith("one independent proof per rank") and mg["exchange"].startswith("none on the data path") and len(mg["peak_bytes_per_rank"]) == 8
