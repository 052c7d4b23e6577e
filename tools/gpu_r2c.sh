#!/bin/bash
# Round 2, third box visit: evidence on HEAD.  (1) the -m gpu suite after the clean-up build (incl. the device-side slab exchange test), (2) rocprofv3 kernel
# stats of the headline bench command, (3) the two PMC passes (FETCH_SIZE / WRITE_SIZE, separate, only --kernel-trace beside --pmc) of the same command,
# (4) the same kernel stats + bench line + CPU baseline for --curve bn254 at configs[1].
OUT=gpurun_out/r2c
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest_gpu.log
BARGS="--steps 2 --warmup 1 --no-cpu-baseline --concurrent 0 --no-slab-leg"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c25519 -o bench -- python $R/bench.py $BARGS > $R/$OUT/bench_under_rocprof.json 2> $R/$OUT/rocprof.err); echo "rocprof stats rc=$?"
f=$(find /tmp/prof_c25519 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $OUT/bench_2p24_kernel_stats.csv; head -12 $OUT/bench_2p24_kernel_stats.csv | cut -c1-200
for CTR in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 400 rocprofv3 --pmc $CTR --kernel-trace --output-format csv -d /tmp/pmc_$CTR -o bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --concurrent 0 --no-slab-leg > $R/$OUT/bench_under_pmc_$CTR.json 2> $R/$OUT/rocprof_$CTR.err); echo "pmc $CTR rc=$?"
  f=$(find /tmp/pmc_$CTR -name '*counter_collection.csv' | head -1); [ -n "$f" ] && cp $f $OUT/bench_${CTR}_counter_collection.csv
done
python tools/pmc_summary.py $OUT/bench_FETCH_SIZE_counter_collection.csv $OUT/bench_WRITE_SIZE_counter_collection.csv $OUT/bench_under_pmc_FETCH_SIZE.json $OUT/bench_traffic.json "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command on MI355X (profiles/r02_pmc/, tools/gpu_r2c.sh)" > $OUT/pmc_summary.log 2>&1; tail -40 $OUT/pmc_summary.log | head -60
# BN254: configs[1] as written, with its CPU baseline (oracle BN254 build, all cores at 2^20) and parity check, then kernel stats
timeout 300 python bench.py --curve bn254 --kind and --c 4 --log-s 20 --cpu-log-s 20 --cpu-1t-log-s 18 --steps 3 --warmup 1 --concurrent 0 --no-slab-leg > $OUT/bench_bn254_config1.json 2> $OUT/bench_bn254_config1.err; echo "bn254 config1 rc=$?"
python -c "
import json; d=json.loads(open('$OUT/bench_bn254_config1.json').read().strip().splitlines()[-1]); print('bn254 config1 ms', d['ms_per_step'], 'cpu', d.get('cpu_baseline',{}).get('sample'), 'parity', d.get('parity_checked'), 'msm', {k:(v or {}).get('frac') for k,v in d.get('roofline_msm',{}).items()})"
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bn254 -o bench -- python $R/bench.py --curve bn254 $BARGS > $R/$OUT/bench_bn254_2p24_under_rocprof.json 2> $R/$OUT/rocprof_bn254.err); echo "rocprof bn254 rc=$?"
f=$(find /tmp/prof_bn254 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $OUT/bench_bn254_2p24_kernel_stats.csv; head -8 $OUT/bench_bn254_2p24_kernel_stats.csv | cut -c1-200
LASSO_TRACE=1 timeout 100 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --concurrent 0 --no-slab-leg --no-prof > /dev/null 2> $OUT/trace_spans_2p24.txt; grep -c trace $OUT/trace_spans_2p24.txt
# drop the bulky per-dispatch CSV rows that are not part of the last proof?  keep: they are ~1500 rows each
ls -la $OUT
exit 0
