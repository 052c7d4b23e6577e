#!/bin/bash
# Round 2, final visit: PMC passes of HEAD (after the byte-table commitment, the fused fingerprint / tree kernels), then the closing validation (tools/gpu_r2g.sh)
OUT=gpurun_out/r2z
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for CTR in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 400 rocprofv3 --pmc $CTR --kernel-trace --output-format csv -d /tmp/pmcz_$CTR -o bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --concurrent 0 --no-slab-leg > $R/$OUT/bench_under_pmc_$CTR.json 2> $R/$OUT/rocprof_$CTR.err); echo "pmc $CTR rc=$?"
  f=$(find /tmp/pmcz_$CTR -name '*counter_collection.csv' | head -1); [ -n "$f" ] && cp $f $OUT/bench_${CTR}_counter_collection.csv
done
python tools/pmc_summary.py $OUT/bench_FETCH_SIZE_counter_collection.csv $OUT/bench_WRITE_SIZE_counter_collection.csv $OUT/bench_under_pmc_FETCH_SIZE.json $OUT/bench_traffic.json "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command on MI355X, HEAD of round 2 (profiles/r02_pmc/, tools/gpu_r2z.sh)" > $OUT/pmc_summary.log 2>&1; python -c "
import json; d=json.load(open('$OUT/bench_traffic.json'))
for k,v in d.items(): print(k, v['bytes_per_launch'], v['alg_bytes_per_launch'], v['traffic_over_algorithmic'], v['launches'])"
bash tools/gpu_r2g.sh
exit 0
