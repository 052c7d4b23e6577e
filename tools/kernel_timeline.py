#!/usr/bin/env python3
"""Print the kernel timeline of the last proof in a rocprofv3 --kernel-trace CSV (start offset, duration, name): where the gaps and the long kernels are."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
count = int(sys.argv[3]) if len(sys.argv) > 3 else 60
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_gather")]
rows = rows[idx[-1]:] if idx else rows
t0 = int(rows[0]["Start_Timestamp"])
prev_end = t0
for r in rows[first:first + count]:
    s = (int(r["Start_Timestamp"]) - t0) / 1e3
    e = (int(r["End_Timestamp"]) - t0) / 1e3
    gap = s - (prev_end - t0) / 1e3
    prev_end = int(r["End_Timestamp"])
    print("%10.1f  +%7.1f gap  %9.1f us  %s" % (s, gap, e - s, r["Kernel_Name"][:80]))
print("total span of the proof: %.1f us, kernels: %d" % ((int(rows[-1]["End_Timestamp"]) - t0) / 1e3, len(rows)))
