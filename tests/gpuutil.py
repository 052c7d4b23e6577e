"""Helpers shared by the GPU parity tests: the real device library next to the oracle's mock of the same C ABI."""
import ctypes
import os
import subprocess

import numpy as np

from fieldref import CURVE, L as FR_P, limbs, to_mont

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_mock():
    from lasso_amd import _abi
    name = "libmock_hip_bn254.so" if CURVE == "bn254" else "libmock_hip.so"
    so = os.path.join(ROOT, "oracle", name)
    srcs = [os.path.join(ROOT, "oracle", f) for f in ("mock_hip.cpp", "lasso_oracle.hpp", "ff.hpp", "ed25519.hpp", "bn254.hpp", "hashes.hpp")] + [os.path.join(ROOT, "include", "lasso_hip.h")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), name])
    lib = ctypes.CDLL(so)
    _abi.declare(lib)
    lib.mock_point_compress.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    lib.mock_point_on_curve.argtypes = [ctypes.c_void_p]
    lib.mock_point_on_curve.restype = ctypes.c_int
    lib.mock_gens.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p]
    return lib


def rand_fr(rng, n, edge=True):
    """(n,4) uint64 Montgomery-form field elements (any value < p is a valid Montgomery representation)."""
    a = rng.integers(0, 2**64, size=(n, 4), dtype=np.uint64)
    a[:, 3] &= np.uint64(2**60 - 1)   # < 2^252 < p
    if edge and n >= 8:
        a[0] = 0
        a[1] = limbs(FR_P - 1)
        a[2] = limbs(1)
        a[3] = limbs(FR_P - 2)
        a[4] = [2**64 - 1, 2**64 - 1, 2**64 - 1, 2**60 - 1]
    return a


def small_fr(vals):
    """small non-negative ints -> (n,4) Montgomery limbs via Python big ints"""
    return np.array([limbs(to_mont(int(v), FR_P)) for v in vals], dtype=np.uint64).reshape(-1, 4)


def compress_points(mock, pts):
    """(k,16) uint64 projective points -> list of 32-byte compressed encodings (via the oracle); also checks curve membership."""
    out = []
    buf = (ctypes.c_uint8 * 32)()
    for i in range(pts.shape[0]):
        p = np.ascontiguousarray(pts[i])
        assert mock.mock_point_on_curve(p.ctypes.data_as(ctypes.c_void_p)) == 1, "point not on curve / T inconsistent"
        mock.mock_point_compress(p.ctypes.data_as(ctypes.c_void_p), buf)
        out.append(bytes(buf))
    return out


def gens(mock, label, n):
    out = np.empty((n + 1, 8), dtype=np.uint64)
    mock.mock_gens(label, n, out.ctypes.data_as(ctypes.c_void_p))
    return out
