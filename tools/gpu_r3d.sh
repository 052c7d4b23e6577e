#!/bin/bash
# Round 3, visit d: LT round in Horner form (parity on both curves, LT C=16 timing), exact executed-addition counters, kernel trace of one default proof
OUT=gpurun_out/r3d; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "combine or claim or lt or msm or hyrax or bullet" > $OUT/pytest_kernels.log 2>&1; echo "kernels rc=$?"; tail -2 $OUT/pytest_kernels.log
LASSO_TEST_CURVE=bn254 timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "combine or claim or lt" > $OUT/pytest_kernels_bn254.log 2>&1; echo "kernels bn254 rc=$?"; tail -2 $OUT/pytest_kernels_bn254.log
timeout 900 python -m pytest tests/test_gpu_prover.py tests/test_golden.py tests/test_gpu_bn254.py -x -q -m gpu -k "not at_baseline_size and not full_size and not slab and not verifies_at_scale" > $OUT/pytest_proofs.log 2>&1; echo "proofs rc=$?"; tail -2 $OUT/pytest_proofs.log
for LS in 22 24; do
  python bench.py --kind lt --c 16 --log-s $LS --steps 3 --warmup 1 --no-cpu-baseline --concurrent 0 --no-slab-leg > $OUT/bench_lt_c16_2p$LS.json 2> $OUT/bench_lt_c16_2p$LS.err; echo "bench lt 2^$LS rc=$?"
  python -c "
import json;d=json.load(open('$OUT/bench_lt_c16_2p$LS.json'));print('lt c16 2^$LS ms_per_step', d['ms_per_step'])
for k in d['kernels_one_profiled_step']: print('   ',k['kernel'],k['launches'],k['ms'],k['avg_launch_us'], k.get('alg_GBps'))"
done
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --concurrent 0 --no-slab-leg > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench default rc=$?"
python -c "
import json;d=json.load(open('$OUT/bench_default.json'));print('default ms_per_step', d['ms_per_step']); print(json.dumps(d.get('roofline_msm'), indent=1)[:1800]); print(d.get('lib_sha'))"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --concurrent 0 --no-slab-leg --no-prof > /dev/null 2> $GRAFT_REPO_ROOT/$OUT/rocprof_kt.err); echo "kernel trace rc=$?"
f=$(find /tmp/kt -name '*kernel_trace.csv' | head -1); [ -n "$f" ] && python - "$f" $OUT/kernel_trace_last_proof.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_gather_u32" in r["Kernel_Name"]]
rows = rows[idx[-1]:] if idx else rows
t0 = int(rows[0]["Start_Timestamp"])
with open(sys.argv[2], "w") as f:
    f.write("start_us,dur_us,gap_us,grid,wg,kernel\n"); prev = t0
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        f.write("%.2f,%.2f,%.2f,%s,%s,%s\n" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")), r["Kernel_Name"][:60].replace(",", ";")))
        prev = e
print("trace rows", len(rows))
PY
exit 0
