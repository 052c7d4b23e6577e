#!/bin/bash
# One short box visit for the BN254 build: its GPU parity tests in the same process as the curve25519 golden tests (both library pairs loaded side by side).
OUT=gpurun_out/bn254
mkdir -p $OUT
timeout 85 python -m pytest tests/test_golden.py tests/test_gpu_bn254.py -m gpu -x -q -s > $OUT/pytest_bn254.log 2>&1; echo "rc=$?" | tee -a $OUT/pytest_bn254.log
grep -a "bn254\]\|passed\|failed\|Error\|error" $OUT/pytest_bn254.log | tail -15
