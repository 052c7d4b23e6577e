#!/bin/bash
# rocprofv3 kernel trace + stats (csv) of the headline bench; outputs under gpurun_out/<tag>/
TAG=${1:-prof}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-prof "$@" > $OUT/bench_under_rocprof.json 2> $OUT/rocprof.err
find /tmp/prof_$TAG -name '*.csv' -exec cp {} $OUT/ \;
ls -la $OUT
