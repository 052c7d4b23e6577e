#!/bin/bash
# PMC passes (separate, as MI355X_MICROARCH.md prescribes) of the headline bench command: HBM read / write bytes per dispatch.
# Only --kernel-trace accompanies --pmc (no hip/hsa/memory-copy tracing).  Outputs under gpurun_out/<tag>/.
TAG=${1:-pmc}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for CTR in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 900 rocprofv3 --pmc $CTR --kernel-trace --output-format csv -d /tmp/pmc_${TAG}_$CTR -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --concurrent 0 > $OUT/bench_$CTR.json 2> $OUT/rocprof_$CTR.err)
  f=$(find /tmp/pmc_${TAG}_$CTR -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && cp $f $OUT/bench_${CTR}_counter_collection.csv
done
ls -la $OUT
