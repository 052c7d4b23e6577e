#!/usr/bin/env python3
"""Throughput of T proofs proved CONCURRENTLY on one GPU (T contexts = T streams, T host threads), each over its own 2^log_s AND lookups.
A single proof is latency-bound by its ~470 sequential transcript rounds (the device is busy ~75% of the time at 2^24); interleaving independent
proofs fills the gaps.  Not the headline metric (bench.py times one proof at a time) — this is the serving-style number."""
import argparse
import sys
import threading
import time
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lasso_amd import HostProver, _abi


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--threads", type=int, default=2); ap.add_argument("--log-s", type=int, default=24); ap.add_argument("--steps", type=int, default=4)
    a = ap.parse_args()
    s = 1 << a.log_s
    S = _abi.Strategy(_abi.KINDS["and"], 1, 16, 0)
    workers = []
    for t in range(a.threads):
        hp = HostProver()
        idx = (hp.gen_indices(s, 1 << 16, 1) + t) % (1 << 16)
        r = hp.gen_random_point(a.log_s)
        gens = hp.gens(1, s, 1, 16); dense = hp.densify(idx, 16); hp.commit(dense, gens)
        hp.prove(dense, gens, S, r)     # warm-up
        workers.append((hp, dense, gens, r))
    for T in sorted({1, a.threads}):
        barrier = threading.Barrier(T + 1)
        def run(w):
            hp, dense, gens, r = w
            barrier.wait()
            for _ in range(a.steps):
                hp.prove(dense, gens, S, r)
            barrier.wait()
        ths = [threading.Thread(target=run, args=(workers[i],)) for i in range(T)]
        for th in ths: th.start()
        barrier.wait(); t0 = time.perf_counter(); barrier.wait(); el = time.perf_counter() - t0
        for th in ths: th.join()
        print(f"{T} concurrent proof stream(s): {T * a.steps} proofs of 2^{a.log_s} lookups in {el * 1e3:.1f} ms = {T * a.steps * s / el / 1e6:.1f} M lookups/s ({el / a.steps * 1e3:.1f} ms per round of {T})")
    for hp, dense, gens, r in workers:
        hp.free(dense, gens); hp.close()


if __name__ == "__main__":
    main()
