#!/bin/bash
# Last box visit of the round: the GPU test files in the driver's order up to the kernel suite, in ONE process (both curve library pairs, both mocks,
# both oracles loaded side by side), bounded to fit the remaining budget.  tests/test_gpu_prover.py ran green on the same device code earlier
# (profiles/r01_pytest_gpu_all_v8.log); it does not fit here.
OUT=gpurun_out/last
mkdir -p $OUT
timeout 44 python -m pytest tests/test_golden.py tests/test_gpu_bn254.py tests/test_gpu_kernels.py -m gpu -x -q > $OUT/pytest_coexist.log 2>&1; echo "rc=$?" | tee -a $OUT/pytest_coexist.log
tail -3 $OUT/pytest_coexist.log
exit 0
