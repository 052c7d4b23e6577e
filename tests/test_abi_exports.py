"""CPU: the built device library exports exactly what include/lasso_hip.h declares, and the Python binding covers all of it.
Only dlopen/dlsym — no GPU call is made."""
import ctypes
import os
import re

from lasso_amd import _abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions(name):
    src = open(os.path.join(ROOT, "include", name)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(lasso_[a-z0-9_]+)\s*\(", src)))


import pytest


@pytest.mark.parametrize("suffix", ["", "_bn254"], ids=["curve25519", "bn254"])
def test_device_library_exports_every_symbol(suffix):
    import __graft_entry__ as g
    g.build()
    lib = ctypes.CDLL(os.path.join(ROOT, "lasso_amd", f"liblasso_hip{suffix}.so"))
    declared = _abi.declare(lib)          # AttributeError if a declared symbol is not exported
    assert declared == header_functions("lasso_hip.h")


@pytest.mark.parametrize("suffix", ["", "_bn254"], ids=["curve25519", "bn254"])
def test_prover_library_exports_every_symbol(suffix):
    """include/lasso_prover.h (the host prover's C ABI): every declared function is exported by liblasso_prover.so (and by the BN254 pair)"""
    import __graft_entry__ as g
    g.build()
    lib = ctypes.CDLL(os.path.join(ROOT, "lasso_amd", f"liblasso_prover{suffix}.so"))
    names = [n for n in header_functions("lasso_prover.h") if n.startswith("lasso_host_")]
    assert len(names) >= 12
    for n in names:
        getattr(lib, n)          # AttributeError = declared but not exported


def test_mock_exports_same_abi():
    from gpuutil import load_mock
    load_mock()


def test_no_cpu_fallback_when_library_missing(tmp_path):
    from lasso_amd import LassoError, load_device_library
    try:
        load_device_library(str(tmp_path / "nope.so"))
    except LassoError as e:
        assert "no CPU fallback" in str(e)
    else:
        raise AssertionError("missing extension must fail loudly")
