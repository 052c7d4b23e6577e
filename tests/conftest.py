"""Shared fixtures. GPU tests are marked @pytest.mark.gpu; everything else runs on CPU.

The oracle (oracle/) is test infrastructure: it is built here on demand and used only as the checker.
"""
import ctypes
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def _make_oracle(name):
    """make decides staleness (oracle/Makefile lists every header, par.hpp included)"""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), name])
    return os.path.join(ROOT, "oracle", name)


def _build_oracle():
    return _make_oracle("liblasso_oracle.so")


def _build_oracle_bn254():
    return _make_oracle("liblasso_oracle_bn254.so")


def _load_oracle(path):
    lib = ctypes.CDLL(path)   # RTLD_LOCAL: the curve25519 and BN254 builds export the same names and must not see each other
    lib.orc_kat_names.restype = ctypes.c_char_p
    lib.orc_last_error.restype = ctypes.c_char_p
    lib.orc_session_new.restype = ctypes.c_void_p
    return lib


@pytest.fixture(scope="session")
def oracle_bn254():
    """The same CPU restatement instantiated over ark-bn254's G1 / Fr (oracle/bn254.hpp)."""
    lib = _load_oracle(_build_oracle_bn254())
    assert lib.orc_curve_id() == 1
    return lib


@pytest.fixture(scope="session")
def oracle():
    if os.environ.get("LASSO_TEST_CURVE") == "bn254":   # the GPU kernel parity tests re-run on the BN254 build (tests/fieldref.py)
        return _load_oracle(_build_oracle_bn254())
    return _load_oracle(_build_oracle())
