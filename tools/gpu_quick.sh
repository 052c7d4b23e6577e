#!/bin/bash
# Quick GPU visit: parity tests + traced bench.  Outputs under gpurun_out/<tag>/.
TAG=${1:-quick}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -4 $OUT/pytest_gpu.log
LASSO_TRACE=1 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-prof > $OUT/bench_trace.json 2> $OUT/bench_trace.err
grep trace $OUT/bench_trace.err | tail -20
python -c "import json;d=json.load(open('$OUT/bench_trace.json'));print('ms_per_step',d['ms_per_step'])"
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
python -c "import json;d=json.load(open('$OUT/bench.json'));print('ms_per_step',d['ms_per_step'], d.get('roofline'))"
