"""CPU: the MSM ceiling bench.py divides by (profiles/r04_madd_ceiling.json) is DERIVED from the kernels' ISA (tools/madd_isa_count.py: VALU instructions of one chained pt_madd,
`hipcc -S` for gfx950 — no GPU needed), so it must be re-derived whenever the mixed addition changes: this test re-runs the derivation and compares with the committed file."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("curve", ["curve25519", "bn254"])
def test_committed_ceiling_matches_the_isa_of_this_tree(curve):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import madd_isa_count
    if not os.path.exists(madd_isa_count.HIPCC):
        pytest.skip("hipcc not available")
    got = madd_isa_count.derive(curve)
    with open(os.path.join(ROOT, "profiles", "r04_madd_ceiling.json")) as f:
        want = json.load(f)[curve]
    assert got["valu_instructions_per_madd"] == want["valu_instructions_per_madd"], "pt_madd changed: run `python tools/madd_isa_count.py --write`"
    assert got["multiply_adds_per_madd"] == want["multiply_adds_per_madd"]
    assert abs(got["G_madd_per_s"] - want["G_madd_per_s"]) < 1e-6
    # the ceiling is what the issue peak allows for that many instructions, nothing else
    assert abs(want["G_madd_per_s"] - 256 * 64 * 2.4e9 / min(want["valu_instructions_per_madd"], want["second_difference"]) / 1e9) < 0.2
