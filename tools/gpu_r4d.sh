#!/bin/bash
# Round 3, visit 4d: kernel trace of one proof on HEAD (timeline CSV for profiles/), trace spans
OUT=gpurun_out/r4d; mkdir -p $OUT
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --concurrent 0 --no-slab-leg --no-prof > $R/$OUT/bench_under_trace.json 2> $R/$OUT/trace.err)
f=$(find /tmp/kt -name '*kernel_trace.csv' | head -1)
python - "$f" > $OUT/kernel_trace_one_proof_2p24.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1]))); rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_gather_u32")]
rows = rows[idx[-1]:]
t0 = int(rows[0]["Start_Timestamp"]); prev = t0
print("start_us,dur_us,gap_us,grid,wg,kernel")
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%.2f,%.2f,%.2f,%s,%s,%s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")), r["Kernel_Name"].split("(")[0].replace(",", ";")[:60]))
    prev = e
PY
wc -l $OUT/kernel_trace_one_proof_2p24.csv; tail -1 $OUT/kernel_trace_one_proof_2p24.csv
LASSO_TRACE=1 timeout 100 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --concurrent 0 --no-slab-leg --no-prof > /dev/null 2> $OUT/trace_spans_2p24.txt; grep "\[trace\]" $OUT/trace_spans_2p24.txt | tail -20
exit 0
