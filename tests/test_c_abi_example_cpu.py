"""CPU: the boundary from plain C.  examples/prove_c_abi.c — densify, commit, prove (labels and callbacks), verify through include/lasso_prover.h alone — is compiled as C99
(which also proves both headers are valid C, not just C++) and linked against the host prover built over the test mock of the device library; it must verify its own proof
and get identical bytes from the two proving paths."""
import os
import subprocess

from proverutil import build_mock_prover

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c99_program_proves_and_verifies_through_the_c_abi(tmp_path):
    so = build_mock_prover()
    exe = str(tmp_path / "prove_c_abi")
    lib_dir, lib_name = os.path.dirname(so), os.path.basename(so)
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "prove_c_abi.c"), "-o", exe,
                           "-L" + lib_dir, "-l:" + lib_name, "-Wl,-rpath," + lib_dir])
    res = subprocess.run([exe, "8"], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "verified 1" in res.stdout and "callback path identical 1" in res.stdout
