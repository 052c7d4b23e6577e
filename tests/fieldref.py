"""Python big-integer ground truth for the two prime fields and the Edwards curve (test helper)."""
import random

import os

# LASSO_TEST_CURVE=bn254 re-runs the GPU kernel parity tests (tests/test_gpu_kernels.py) on the BN254 build: Fr becomes the order of ark-bn254's G1
CURVE = os.environ.get("LASSO_TEST_CURVE", "curve25519")
L = (21888242871839275222246405745257275088548364400416034343698204186575808495617 if CURVE == "bn254"
     else 2**252 + 27742317777372353535851937790883648493)  # Fr: curve25519 scalar field
Q = 2**255 - 19                                       # Fq: curve25519 base field
R = 2**256
D = (-121665 * pow(121666, -1, Q)) % Q
GX = 15112221349535400772501151409588531511454012693041857206046113283949847762202
GY = 46316835694926478169428394003475163141307993866256225615783033603165251855960


def limbs(x):
    return [(x >> (64 * i)) & (2**64 - 1) for i in range(4)]


def unlimbs(v):
    return sum(int(v[i]) << (64 * i) for i in range(4))


def to_mont(x, p):
    return (x * R) % p


def from_mont(x, p):
    return (x * pow(R, -1, p)) % p


def ed_add(P1, P2):
    (x1, y1), (x2, y2) = P1, P2
    k = D * x1 * x2 * y1 * y2 % Q
    x3 = (x1 * y2 + y1 * x2) * pow(1 + k, -1, Q) % Q
    y3 = (y1 * y2 + x1 * x2) * pow(1 - k, -1, Q) % Q  # a = -1: y1y2 - a x1x2
    return (x3, y3)


def ed_mul(P, k):
    acc = (0, 1)
    while k:
        if k & 1:
            acc = ed_add(acc, P)
        P = ed_add(P, P)
        k >>= 1
    return acc


def rng(seed):
    return random.Random(seed)
