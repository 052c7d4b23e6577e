"""Open-ended differential fuzz of the host prover over the mock against the oracle prover (test infrastructure; tests/test_fuzz_host_cpu.py is the bounded,
seeded version).  usage: python tools/fuzz_host.py [curve25519|bn254] [seed] [seconds] [hip]"""
import ctypes as C, sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from lasso_amd import _abi
from proverutil import HostProver, OracleSession, build_mock_prover
import conftest
curve = sys.argv[1] if len(sys.argv) > 1 else "curve25519"
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
budget = float(sys.argv[3]) if len(sys.argv) > 3 else 120
orc = conftest._load_oracle(conftest._build_oracle_bn254() if curve == "bn254" else conftest._build_oracle())
if "hip" in sys.argv:      # the product library on a GPU box instead of the mock: python tools/fuzz_host.py curve25519 1 60 hip [capacity]
    from lasso_amd import HostProver as HipProver
    hp = HipProver(curve=curve)
    if "capacity" in sys.argv: hp.set_capacity(True)      # with LASSO_LEAFLESS_MIN=64 in the environment: the chunked leaf rounds on every instance large enough
else:
    hp = HostProver(C.CDLL(build_mock_prover(curve)))
rng = np.random.default_rng(seed0)
t0 = time.time(); n = 0
while time.time() - t0 < budget:
    kind = ["and", "or", "xor", "lt", "range", "spark"][rng.integers(6)]      # spark = LASSO_SPARK_UNCONFIRMED (round 4)
    c = int(rng.integers(1, 5)); log_m = int(rng.integers(1, 9));
    if kind not in ("range", "spark") and log_m % 2: log_m += 1   # the bitwise tables split an address into two operands of log_m / 2 bits
    lookups = int(rng.integers(2, 700)) if rng.integers(4) else int(rng.integers(2, 5000))   # one lookup (s = 1) is outside the reference's domain: GrandProductCircuit::new needs two leaves
    log_r = int(rng.integers(1, c * log_m + 1)) if kind == "range" else 0
    if kind == "range" and c * log_m > 63: continue
    s = 1 << max((lookups - 1).bit_length(), 0)
    if s < 2: s = 2 if False else s
    alpha = 2 * c if kind == "lt" else c
    mode = int(rng.integers(3))
    if mode == 0: idx = rng.integers(0, 1 << log_m, size=(lookups, c), dtype=np.uint64)
    elif mode == 1: idx = np.repeat(rng.integers(0, 1 << log_m, size=(lookups, 1), dtype=np.uint64), c, axis=1).copy()
    else: idx = np.full((lookups, c), int(rng.integers(0, 1 << log_m)), dtype=np.uint64)   # one address hit every time
    bits = max(s.bit_length() - 1, 0)
    r = hp.gen_random_point(max(bits, 1))[:bits]
    S = _abi.Strategy(_abi.KINDS[kind], c, log_m, log_r)
    tag = f"{kind} C={c} log_m={log_m} log_r={log_r} lookups={lookups} mode={mode}"
    try:
        gens = hp.gens(c, s, alpha, log_m); dense = hp.densify(idx, log_m)
        comm = hp.commit(dense, gens); proof = hp.prove(dense, gens, S, r)
        acc = hp.verify(gens, S, s, r, proof, comm)
        bad = bytearray(proof); bad[int(rng.integers(len(bad)))] ^= 1 << int(rng.integers(8))
        try: rej = hp.verify(gens, S, s, r, bytes(bad), comm) is False
        except Exception: rej = True
        hp.free(dense, gens)
        if acc is not True: print("VERIFIER REJECTS HONEST PROOF", tag)
    except Exception as e:
        print("HOST FAIL", tag, repr(e)[:300]); continue
    try:
        o = OracleSession(orc, _abi.KINDS[kind], c, log_m, log_r, idx, r)
        oc, op = o.commit(), o.prove(); ok = o.verify(proof, comm)
        if not rej:   # a flipped bit accepted: legitimate only for encodings ark itself treats as equivalent (sign bit of an x = 0 point) — then the oracle's verifier accepts too
            try: also = o.verify(bytes(bad), comm) == 1
            except Exception: also = False
            print("flipped bit accepted by both verifiers (malleable point encoding)" if also else "VERIFIER MISMATCH: corrupted proof accepted by the product verifier only", tag)
        o.close()
    except Exception as e:
        print("ORACLE FAIL", tag, repr(e)[:300]); continue
    if comm != oc or proof != op or ok != 1: print("MISMATCH", tag, comm == oc, proof == op, ok)
    n += 1
print("configs", n, "curve", curve)
