"""The product's __host__ __device__ arithmetic headers (lasso_amd/csrc/fr.cuh, fq.cuh, fe29.cuh) checked on the CPU against the
oracle's independent 64-bit arithmetic (tests/cpp/*.cpp).  Both host forms are covered: the 64-bit-limb form g++ builds for the
host prover, and (-DLASSO_HOST_LIMBS32) the 32-bit-limb form the device executes."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("name", ["test_arith_host", "test_fe29_host", "test_fr29_host", "test_modinv_host"])
@pytest.mark.parametrize("flags", [[], ["-DLASSO_HOST_LIMBS32"]], ids=["limbs64", "limbs32"])
def test_cpp_arith(name, flags):
    src = os.path.join(ROOT, "tests", "cpp", name + ".cpp")
    if not os.path.exists(src):
        pytest.skip(f"{name}.cpp not present")
    out_dir = os.path.join(ROOT, "tests", "_build")
    os.makedirs(out_dir, exist_ok=True)
    exe = os.path.join(out_dir, name + ("_32" if flags else "_64"))
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wno-unknown-pragmas", *flags, "-o", exe, src])
    res = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    assert "OK" in res.stdout


@pytest.mark.parametrize("name", ["test_bn254_host", "test_fr29_host"])
@pytest.mark.parametrize("flags", [[], ["-DLASSO_HOST_LIMBS32"]], ids=["limbs64", "limbs32"])
def test_cpp_arith_bn254(name, flags):
    """The BN254 build of the same headers (-DLASSO_BN254: bn254_*.cuh over mont29.cuh) against the oracle's BN254 instantiation."""
    src = os.path.join(ROOT, "tests", "cpp", name + ".cpp")
    out_dir = os.path.join(ROOT, "tests", "_build")
    os.makedirs(out_dir, exist_ok=True)
    exe = os.path.join(out_dir, name + "_bn254" + ("_32" if flags else "_64"))
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wno-unknown-pragmas", "-DLASSO_BN254", "-DORC_BN254", *flags, "-o", exe, src])
    res = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    assert "OK" in res.stdout


@pytest.mark.parametrize("name,flags", [("test_fr29_host", []), ("test_fe29_host", []), ("test_fr29_host", ["-DLASSO_BN254"]), ("test_bn254_host", ["-DLASSO_BN254", "-DORC_BN254"])],
                         ids=["fr29", "fe29", "fr29-bn254", "curve-bn254"])
def test_limb_bounds_hold_under_ubsan(name, flags):
    """The 29-bit-limb forms rely on magnitudes never leaving their 32- / 64-bit containers (no carries between partial products).  The same tests
    built with -fsanitize=undefined turn any signed overflow on the way — in a product column, a lazy sum, a quotient estimate — into a failure, for both
    curve builds (it found the BN254 table conversion feeding a 2^259 value to a reduction that accepts 2^258)."""
    src = os.path.join(ROOT, "tests", "cpp", name + ".cpp")
    out_dir = os.path.join(ROOT, "tests", "_build")
    os.makedirs(out_dir, exist_ok=True)
    exe = os.path.join(out_dir, name + "_ubsan" + ("_bn254" if flags else ""))
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wno-unknown-pragmas", "-fsanitize=undefined", "-fno-sanitize-recover=undefined", *flags, "-o", exe, src])
    res = subprocess.run([exe], capture_output=True, text=True, timeout=900)
    assert res.returncode == 0 and "runtime error" not in res.stderr, res.stdout[-1000:] + res.stderr[-3000:]
    assert "OK" in res.stdout
