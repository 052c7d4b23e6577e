"""lasso_amd — MI355X-native hot path for a16z/Lasso's SparsePolynomialEvaluationProof.

The product is two native libraries:
  liblasso_hip.so     hand-written gfx950 kernels behind the C ABI in include/lasso_hip.h
  liblasso_prover.so  C++ mirror of the reference's Rust host (protocol + Merlin transcript), include/lasso_prover.h
This package is a thin ctypes binding used by tests/ and bench.py.  There is NO CPU fallback: importing the device
layer on a machine without the built extension or without a GPU raises.
"""
from .device import Device, LassoError, load_device_library  # noqa: F401
from .prover import HostProver, load_prover_library  # noqa: F401,E402
