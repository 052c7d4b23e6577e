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


CURVES = {"curve25519": ("", []), "bn254": ("_bn254", ["-DLASSO_BN254", "-Wl,-Bsymbolic"])}   # library suffix, extra compile flags


def build_device(force=False, verbose=False, curve="curve25519"):
    """liblasso_hip.so: hand-written gfx950 kernels behind include/lasso_hip.h.  curve="bn254" builds the same kernels over ark-bn254's
    Fr / G1 (csrc/bn254_*.cuh, mont29.cuh) into liblasso_hip_bn254.so — same C ABI, same symbol names, loaded side by side (RTLD_LOCAL)."""
    suffix, flags = CURVES[curve]
    csrc = os.path.join(HERE, "csrc")
    target = os.path.join(HERE, f"liblasso_hip{suffix}.so")
    sources = _glob(csrc, (".hip", ".cuh")) + [os.path.join(ROOT, "include", "lasso_hip.h")]
    if force or _stale(target, sources):
        extra = os.environ.get("LASSO_EXTRA_HIPCC_FLAGS", "").split()   # A/B builds of compile-time switches (e.g. -DLASSO_PLAIN_PARTIALS), on the GPU box
        cmd = [HIPCC, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-result", *flags, *extra,
               "-o", target, os.path.join(csrc, "lasso_hip.hip")]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return target


def host_march_flags():
    """-march=x86-64-v3 (AVX2, BMI1/2: what every x86 host a gfx950 ships in has) makes the transcript's Keccak-f 10-18 % faster — but a library built with it dies with SIGILL,
    and no diagnostic, on a host or VM without those extensions.  The flag is therefore used only when the build host's CPU reports them (LASSO_HOST_MARCH overrides:
    a -march value, or "none"), the library records the choice (-DLASSO_HOST_V3), and lasso_host_create checks the CPU it is LOADED on before anything else runs."""
    ov = os.environ.get("LASSO_HOST_MARCH")
    if ov:
        return [] if ov == "none" else [f"-march={ov}"] + (["-DLASSO_HOST_V3"] if ov == "x86-64-v3" else [])
    try:
        flags = set(next(l for l in open("/proc/cpuinfo") if l.startswith("flags")).split())
    except Exception:
        flags = set()
    if {"avx2", "bmi1", "bmi2", "fma", "movbe"} <= flags:
        return ["-march=x86-64-v3", "-DLASSO_HOST_V3"]
    return []


def build_host(force=False, verbose=False, curve="curve25519"):
    """liblasso_prover.so: the C++ mirror of the reference's Rust prover, linked against liblasso_hip.so (or the _bn254 pair)."""
    suffix, flags = CURVES[curve]
    hdir = os.path.join(HERE, "host")
    target = os.path.join(HERE, f"liblasso_prover{suffix}.so")
    sources = _glob(hdir, (".cpp", ".hpp")) + _glob(os.path.join(HERE, "csrc"), (".cuh",)) + [os.path.join(ROOT, "include", "lasso_hip.h"), os.path.join(ROOT, "include", "lasso_prover.h")]
    if not os.path.exists(os.path.join(hdir, "prover_capi.cpp")):
        return None
    dev = build_device(force=False, verbose=verbose, curve=curve)
    if force or _stale(target, sources + [dev]):
        # -fno-gnu-unique / -Bsymbolic: the two curve builds share C++ names and may be loaded side by side; nothing may be unified across them
        # -march=x86-64-v3 (AVX2, BMI1/2: every x86 host a gfx950 ships in): the transcript's Keccak-f is 10-18 % faster with andn / rorx and three-operand forms
        # (tools: 0.39 -> 0.32 us per permutation on the build container), and ~1.5 ms of a 19 ms proof is host Keccak
        cmd = ["g++", "-O2", *host_march_flags(), "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function", "-Wno-unknown-pragmas", "-fno-gnu-unique", "-Wl,-Bsymbolic", *flags, "-o", target,
               os.path.join(hdir, "prover_capi.cpp"), "-L" + HERE, f"-llasso_hip{suffix}", "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return target


def build_all(force=False, verbose=False):
    out = []
    for curve in CURVES:
        out += [build_device(force, verbose, curve), build_host(force, verbose, curve)]
    return tuple(out)


if __name__ == "__main__":
    print(build_all(force="--force" in sys.argv, verbose=True))
