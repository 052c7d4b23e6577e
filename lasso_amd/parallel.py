"""Multi-GPU plumbing (one process per GPU, torch.distributed; backend "nccl" = RCCL over xGMI on ROCm, "gloo" in CPU tests).

The north-star path shards by PROOF: each rank owns an independent batch of lookups (its own DensifiedRepresentation,
transcript and proof) — independent objects, so there is no data-path collective.  torch.distributed
is only used for the barrier / max-over-ranks timing contract of bench.py, to gather proof digests and to agree on a job nonce.
Slab mode (ONE proof over the ranks) exchanges through the library itself: HostProver.set_comm_shm (shared-memory all-gather of the per-round
sums, RCCL all-gather of the partial row commitments on the library's stream); `allgather_callback` below is the older transport-agnostic
hook (lasso_host_set_comm), kept for embedders and covered by the gloo test."""
import hashlib
import os


class Group:
    def __init__(self, backend=None):
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.dist = None
        self.backend = None
        if self.world > 1:
            import torch
            import torch.distributed as dist
            self.backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
            if self.backend == "nccl":
                torch.cuda.set_device(self.local_rank)
            dist.init_process_group(backend=self.backend, init_method="env://")
            self.dist = dist
            self.torch = torch

    @property
    def device_index(self):
        """One GPU per rank; with the gloo backend on a box with fewer GPUs than ranks (tests), ranks share devices."""
        if self.world == 1:
            return 0
        if self.backend == "gloo":
            try:
                import torch
                n = torch.cuda.device_count()
                return self.local_rank % n if n else 0
            except Exception:
                return 0
        return self.local_rank

    def _tensor(self, values, dtype):
        dev = "cuda" if self.backend == "nccl" else "cpu"
        return self.torch.tensor(values, dtype=dtype, device=dev)

    def barrier(self):
        if self.dist is not None:
            if self.backend == "nccl":
                self.torch.cuda.synchronize()
            self.dist.barrier()

    def max_over_ranks(self, x):
        if self.dist is None:
            return float(x)
        t = self._tensor([float(x)], self.torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, x):
        if self.dist is None:
            return float(x)
        t = self._tensor([float(x)], self.torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def gather_floats(self, x):
        """one number per rank, gathered on every rank (rank order): per-rank measurements (peak bytes, ...) for the bench line"""
        if self.dist is None:
            return [float(x)]
        t = self._tensor([float(x)], self.torch.float64)
        out = [self.torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        return [float(o.item()) for o in out]

    def shared_nonce(self):
        """One random 64-bit value, the same on every rank (drawn by rank 0): makes per-job names — e.g. the shared-memory segment of slab mode — that a
        crashed earlier job with the same MASTER_PORT cannot collide with."""
        import secrets
        x = secrets.randbits(62)
        if self.dist is None:
            return x
        t = self._tensor([x], self.torch.int64)
        self.dist.broadcast(t, src=0)
        return int(t.item())

    def gather_digests(self, payload: bytes):
        """sha256 of each rank's proof, gathered on every rank (32 bytes per rank — the only bytes that cross ranks)."""
        h = hashlib.sha256(payload).digest()
        if self.dist is None:
            return [h]
        t = self._tensor(list(h), self.torch.uint8)
        out = [self.torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        return [bytes(o.cpu().tolist()) for o in out]

    def allgather_callback(self):
        """The collective of slab mode (include/lasso_prover.h lasso_host_set_comm): gathers `nbytes` from every rank's host buffer `send` into `recv`
        in rank order.  RCCL over xGMI when the backend is "nccl" (staged through device tensors), gloo in the CPU tests."""
        import ctypes as C
        import numpy as np

        def cb(_user, send, recv, nbytes):
            try:
                if self.dist is None:
                    C.memmove(recv, send, nbytes)
                    return 0
                torch = self.torch
                src = np.ctypeslib.as_array(C.cast(send, C.POINTER(C.c_uint8)), shape=(nbytes,))
                dst = np.ctypeslib.as_array(C.cast(recv, C.POINTER(C.c_uint8)), shape=(nbytes * self.world,))
                dev = "cuda" if self.backend == "nccl" else "cpu"
                t_in = torch.from_numpy(src.copy()).to(dev)
                t_out = torch.empty(nbytes * self.world, dtype=torch.uint8, device=dev)
                self.dist.all_gather_into_tensor(t_out, t_in)
                dst[:] = t_out.cpu().numpy()
                return 0
            except Exception as e:      # never unwind into C
                import sys
                print("all-gather callback failed:", repr(e), file=sys.stderr)
                return -1
        return cb

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()


def shard_seed(rank):
    """each rank's batch of lookups is drawn from its own stream (rank 0 = the reference harness's stream)"""
    return rank


def shard_indices(hp, s, m, c, rank):
    """rank 0 reproduces benches/bench.rs gen_indices exactly; rank r > 0 rotates every index by r (a different, equally
    distributed batch) so the proofs of different ranks differ."""
    idx = hp.gen_indices(s, m, c)
    if rank:
        idx = (idx + rank) % m
    return idx
