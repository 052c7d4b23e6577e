#!/usr/bin/env python3
"""The MSM families' ceiling, derived instead of measured (the product kernels beat the microbenchmark they were normalised by).

A mixed point addition (`pt_madd`: accumulator += precomputed table entry — the one operation the commitment kernels and the opening MSMs' accumulation phase
execute per non-zero digit / byte) is a fixed straight-line VALU sequence.  Its instruction count is read off the gfx950 ISA: two kernels that differ by exactly one
chained pt_madd are compiled with `hipcc -S`, and the difference of their VALU instruction counts is the cost of one addition (loads, unpacking and the store cancel).
A CU issues at most one VALU instruction per SIMD per cycle for 16 lanes, i.e. 64 lane-instructions per CU per clock:
    peak lane-instructions/s = 256 CUs x 64 x 2.4 GHz (MI355X_MICROARCH.md: max clock)          = 3.93e13
    ceiling [additions/s]    = peak lane-instructions/s / VALU instructions per pt_madd
No schedule of the same arithmetic can exceed it (every multiply-add of the 29-bit-limb products is a full-rate VALU instruction: tools/microbench section 1), so
`roofline_msm.*.frac` = executed additions / time / ceiling is <= 1 by construction; bench.py asserts it.
Usage: tools/madd_isa_count.py [--write]   (writes profiles/r04_madd_ceiling.json)"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
CUS, LANES_PER_CU_CLK, CLOCK_HZ = 256, 64, 2.4e9

SRC = r"""
#include <hip/hip_runtime.h>
#include "%(csrc)s/msm_kernels.cuh"
extern "C" __global__ void k_chain1(pt29* acc, const niels29* tab) { pt29 P = acc[threadIdx.x]; P = pt_madd(P, tab[threadIdx.x]); acc[threadIdx.x] = P; }
extern "C" __global__ void k_chain2(pt29* acc, const niels29* tab) { pt29 P = acc[threadIdx.x]; P = pt_madd(P, tab[threadIdx.x]); P = pt_madd(P, tab[threadIdx.x + 64]); acc[threadIdx.x] = P; }
extern "C" __global__ void k_chain3(pt29* acc, const niels29* tab) {
  pt29 P = acc[threadIdx.x]; P = pt_madd(P, tab[threadIdx.x]); P = pt_madd(P, tab[threadIdx.x + 64]); P = pt_madd(P, tab[threadIdx.x + 128]); acc[threadIdx.x] = P; }
"""


def kernel_bodies(asm):
    out = {}
    for m in re.finditer(r"^k_chain(\d):[^\n]*\n(.*?)s_endpgm", asm, flags=re.S | re.M):
        out[int(m.group(1))] = m.group(2)
    return out


def count(body):
    ins = [l.strip().split()[0] for l in body.splitlines() if l.strip() and not l.strip().startswith((";", ".")) and not l.strip().endswith(":")]
    valu = [i for i in ins if i.startswith("v_")]
    return {"valu": len(valu), "mad_i64_i32": sum(1 for i in valu if i.startswith("v_mad_i64_i32") or i.startswith("v_mad_u64_u32")), "all": len(ins)}


def derive(curve):
    flags = ["-DLASSO_BN254"] if curve == "bn254" else []
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "chain.hip")
        open(src, "w").write(SRC % {"csrc": os.path.join(ROOT, "lasso_amd", "csrc")})
        asm = os.path.join(d, "chain.s")
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", *flags, "-o", asm, src])
        bodies = kernel_bodies(open(asm).read())
    c = {n: count(b) for n, b in bodies.items()}
    per = c[2]["valu"] - c[1]["valu"]; per2 = c[3]["valu"] - c[2]["valu"]
    peak_lane = CUS * LANES_PER_CU_CLK * CLOCK_HZ
    return {"valu_instructions_per_madd": max(per, per2), "second_difference": per2, "multiply_adds_per_madd": c[2]["mad_i64_i32"] - c[1]["mad_i64_i32"],
            "peak_lane_instructions_per_s": peak_lane, "G_madd_per_s": round(peak_lane / min(per, per2) / 1e9, 3),
            "source": f"tools/madd_isa_count.py: hipcc -S of k_chain<1..3> ({curve} build of lasso_amd/csrc), VALU instructions per chained pt_madd = {per} / {per2}; "
                      f"peak = {CUS} CUs x {LANES_PER_CU_CLK} lane-instructions per clock x {CLOCK_HZ / 1e9} GHz; the smaller count is used (the higher ceiling)"}


if __name__ == "__main__":
    out = {cv: derive(cv) for cv in ("curve25519", "bn254")}
    print(json.dumps(out, indent=1))
    if "--write" in sys.argv:
        with open(os.path.join(ROOT, "profiles", "r04_madd_ceiling.json"), "w") as f:
            json.dump(out, f, indent=1); f.write("\n")
