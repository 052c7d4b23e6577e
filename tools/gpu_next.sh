#!/bin/bash
# First box visit of the next round: what this round prepared but could not measure.  ~6 GPU-minutes, everything bounded.
#   1. differential fuzz of the HIP path against the oracle prover, both curve builds (tools/fuzz_host.py ... hip)
#   2. SURVEY 8(d)'s bind_top sweep (tools/microbench section 4)
#   3. BN254: LASSO_MSM_DIRECT_WGS sweep at the metric's shape, then kernel stats under rocprofv3
OUT=gpurun_out/next
mkdir -p $OUT
export TMPDIR=/tmp
for curve in curve25519 bn254; do timeout 100 python tools/fuzz_host.py $curve 31 60 hip > $OUT/fuzz_hip_$curve.log 2>&1; tail -2 $OUT/fuzz_hip_$curve.log; grep -c "FAIL\|MISMATCH" $OUT/fuzz_hip_$curve.log; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-result -Ilasso_amd/csrc -o tools/microbench tools/microbench.hip 2> /dev/null
timeout 240 tools/microbench > $OUT/microbench.txt 2>&1; grep -A30 "== 4" $OUT/microbench.txt | head -30
# 4. device-resident Fiat-Shamir groundwork: does the lane-distributed transcript match the host's on the GPU, and what does a round cost there
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Ilasso_amd/csrc -Iinclude -o tools/transcript_bench tools/transcript_bench.hip 2> /dev/null
timeout 30 tools/transcript_bench | tee $OUT/transcript_bench.txt
for wgs in 256 512 1024; do LASSO_MSM_DIRECT_WGS=$wgs timeout 40 python bench.py --curve bn254 --steps 3 --warmup 1 --no-cpu-baseline --concurrent 0 --no-prof > $OUT/bench_bn254_wgs$wgs.json 2> $OUT/bench_bn254_wgs$wgs.err; echo "bn254 wgs=$wgs $(python -c "import json;print(json.load(open('$OUT/bench_bn254_wgs$wgs.json'))['ms_per_step'])")"; done
for wgs in 256 512; do LASSO_MSM_DIRECT_WGS=$wgs timeout 40 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --concurrent 0 --no-prof > $OUT/bench_wgs$wgs.json 2> $OUT/bench_wgs$wgs.err; echo "curve25519 wgs=$wgs $(python -c "import json;print(json.load(open('$OUT/bench_wgs$wgs.json'))['ms_per_step'])")"; done
(cd /tmp && timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bn254 -o bench -- python $GRAFT_REPO_ROOT/bench.py --curve bn254 --steps 3 --warmup 1 --no-cpu-baseline --concurrent 0 > $GRAFT_REPO_ROOT/$OUT/bench_bn254_under_rocprof.json 2> $GRAFT_REPO_ROOT/$OUT/rocprof_bn254.err)
find /tmp/prof_bn254 -name '*kernel_stats*.csv' -exec cp {} $OUT/bench_bn254_kernel_stats.csv \;
ls $OUT
exit 0
