#!/bin/bash
# differential fuzz of the HIP path (all the kernels of this round: byte-table commitments, fused fingerprint / two-layer trees, streamed LT rounds) against the
# oracle prover, plus the product verifier on every proof; both curve builds
OUT=gpurun_out/r2y
mkdir -p $OUT
export OMP_NUM_THREADS=16
for curve in curve25519 bn254; do timeout 130 python tools/fuzz_host.py $curve 4242 75 hip > $OUT/fuzz_hip_$curve.log 2>&1; tail -3 $OUT/fuzz_hip_$curve.log; echo "mismatch lines: $(grep -c 'MISMATCH\|FAIL\|REJECTS' $OUT/fuzz_hip_$curve.log)"; done
exit 0
