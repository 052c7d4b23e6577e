#!/bin/bash
OUT=gpurun_out/r2f
mkdir -p $OUT
export TMPDIR=/tmp
timeout 60 tools/latency_bench > $OUT/latency_bench.txt 2>&1; cat $OUT/latency_bench.txt
LASSO_TRACE=2 timeout 100 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --concurrent 0 --no-slab-leg --no-prof > /dev/null 2> $OUT/trace2.txt; grep "\[host\]" $OUT/trace2.txt | tail -12; grep "SparsePoly.prove\|HashLayer.prove\|ProductLayer.prove" $OUT/trace2.txt | tail -3
exit 0
