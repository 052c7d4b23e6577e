"""Build the HIP device library and the C++ host prover for gfx950 (in-tree .so files; they travel with gpurun)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _glob(d, exts):
    return [os.path.join(d, f) for f in sorted(os.listdir(d)) if f.endswith(exts)]


def build_device(force=False, verbose=False):
    """liblasso_hip.so: hand-written gfx950 kernels behind include/lasso_hip.h."""
    csrc = os.path.join(HERE, "csrc")
    target = os.path.join(HERE, "liblasso_hip.so")
    sources = _glob(csrc, (".hip", ".cuh")) + [os.path.join(ROOT, "include", "lasso_hip.h")]
    if force or _stale(target, sources):
        cmd = [HIPCC, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-result",
               "-o", target, os.path.join(csrc, "lasso_hip.hip")]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return target


def build_host(force=False, verbose=False):
    """liblasso_prover.so: the C++ mirror of the reference's Rust prover, linked against liblasso_hip.so."""
    hdir = os.path.join(HERE, "host")
    target = os.path.join(HERE, "liblasso_prover.so")
    sources = _glob(hdir, (".cpp", ".hpp")) + _glob(os.path.join(HERE, "csrc"), (".cuh",)) + [os.path.join(ROOT, "include", "lasso_hip.h"), os.path.join(ROOT, "include", "lasso_prover.h")]
    if not os.path.exists(os.path.join(hdir, "prover_capi.cpp")):
        return None
    dev = build_device(force=False, verbose=verbose)
    if force or _stale(target, sources + [dev]):
        cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function", "-Wno-unknown-pragmas", "-o", target,
               os.path.join(hdir, "prover_capi.cpp"), "-L" + HERE, "-llasso_hip", "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return target


def build_all(force=False, verbose=False):
    return build_device(force, verbose), build_host(force, verbose)


if __name__ == "__main__":
    print(build_all(force="--force" in sys.argv, verbose=True))
