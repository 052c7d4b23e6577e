#!/bin/bash
# Round 2, first box visit: parity at the claimed sizes, the new bench line (all-core CPU baseline + parity check), and the measurements round 1 prepared
# but never ran (transcript on the device, SURVEY 8(d) bind_top sweep, streaming-ceiling exploration, BN254 MSM tuning knob).  Everything bounded.
OUT=gpurun_out/r2a
mkdir -p $OUT
export TMPDIR=/tmp
{ nproc; lscpu | grep -E "Model name|Socket|Core|Thread|^CPU\(s\)"; free -g | head -2; rocm-smi --showclocks 2>/dev/null | head -20; } > $OUT/host.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q -s -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/summary.txt; tail -3 $OUT/pytest_gpu.log; grep "^\[oracle" $OUT/pytest_gpu.log
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/summary.txt; python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r2a/bench.json").read().strip().splitlines()[-1])
    print("ms_per_step", d["ms_per_step"], "cpu", d.get("cpu_baseline", {}).get("sample"), d.get("cpu_baseline", {}).get("one_thread"), "parity", d.get("parity_checked"))
    print("roofline", d.get("roofline", {}).get("frac"), "bind", d.get("roofline_bind_top", {}).get("frac"), "msm", d.get("roofline_msm"))
except Exception as e:
    print("bench parse failed", e)
PY
timeout 60 tools/transcript_bench > $OUT/transcript_bench.txt 2>&1; tail -15 $OUT/transcript_bench.txt
timeout 400 tools/microbench 2,4,5 > $OUT/microbench.txt 2>&1; grep -E "MADD_CEILING|pt_madd" $OUT/microbench.txt; grep -A22 "== 4" $OUT/microbench.txt | head -24; grep -A40 "== 5" $OUT/microbench.txt
timeout 60 tools/microbench_bn254 2 > $OUT/microbench_bn254.txt 2>&1; grep -E "MADD_CEILING" $OUT/microbench_bn254.txt
for wgs in 256 512 1024; do LASSO_MSM_DIRECT_WGS=$wgs timeout 60 python bench.py --curve bn254 --steps 3 --warmup 1 --no-cpu-baseline --concurrent 0 --no-prof > $OUT/bench_bn254_wgs$wgs.json 2> $OUT/bench_bn254_wgs$wgs.err; echo "bn254 wgs=$wgs $(python -c "import json;print(json.loads(open('$OUT/bench_bn254_wgs$wgs.json').read().strip().splitlines()[-1])['ms_per_step'])")"; done
for wgs in 512; do LASSO_MSM_DIRECT_WGS=$wgs timeout 60 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --concurrent 0 --no-prof > $OUT/bench_wgs$wgs.json 2> $OUT/bench_wgs$wgs.err; echo "curve25519 wgs=$wgs $(python -c "import json;print(json.loads(open('$OUT/bench_wgs$wgs.json').read().strip().splitlines()[-1])['ms_per_step'])")"; done
ls $OUT
exit 0
