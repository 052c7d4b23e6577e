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


def test_rust_extern_block_mirrors_both_headers():
    """integration/rust/ffi.rs (the extern "C" blocks a maintainer of the reference adds) is the generator's output for the CURRENT headers, declares
    every function of include/lasso_hip.h and include/lasso_prover.h exactly once, with the same number of parameters, and declares nothing else."""
    import subprocess
    import sys
    assert subprocess.call([sys.executable, os.path.join(ROOT, "tools", "gen_rust_ffi.py"), "--check"]) == 0, "integration/rust/ffi.rs is stale: run tools/gen_rust_ffi.py"
    rs = open(os.path.join(ROOT, "integration", "rust", "ffi.rs")).read()
    rust = re.findall(r"pub fn (lasso_[a-z0-9_]+)\(([^)]*)\)", rs)
    names = [n for n, _ in rust]
    assert len(names) == len(set(names))
    declared = set(header_functions("lasso_hip.h")) | {n for n in header_functions("lasso_prover.h") if n.startswith("lasso_host_")}
    assert set(names) == declared
    # parameter counts against the C prototypes
    for header in ("lasso_hip.h", "lasso_prover.h"):
        src = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", header)).read(), flags=re.S)
        src = re.sub(r"typedef[^;]*;", "", src)
        for name, args in re.findall(r"\b(lasso_[a-z0-9_]+)\s*\(([^()]*)\)\s*;", src):
            n_c = 0 if args.strip() in ("", "void") else args.count(",") + 1
            n_rs = [a for nm, a in rust if nm == name]
            assert n_rs, name
            assert (0 if not n_rs[0].strip() else n_rs[0].count(",") + 1) == n_c, name
    for f in ("hip.rs", "bench_types.rs"):       # the shim only calls what the extern block declares
        used = set(re.findall(r"\b(lasso_[a-z0-9_]+)\s*\(", open(os.path.join(ROOT, "integration", "rust", f)).read()))
        assert used <= set(names), used - set(names)
