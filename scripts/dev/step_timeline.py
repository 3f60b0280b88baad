#!/usr/bin/env python3
"""Timeline of one sweep step from a rocprofv3 --kernel-trace csv (dev tool): start / end of every kernel of the median
step relative to the step's first kernel - shows how the two streams of SweepBatch.step_rest() really overlap."""
import csv
import glob
import sys

path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
big = [i for i, r in enumerate(rows) if "spmm_quad" in r["Kernel_Name"]]
i0 = big[len(big) // 2]
i1 = big[len(big) // 2 + 1]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i1]:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print(f"{s:8.1f} -> {e:8.1f} us ({e - s:6.1f})  queue {r.get('Queue_Id', '?'):>3}  {r['Kernel_Name'][:70]}")
print(f"step period: {(int(rows[i1]['Start_Timestamp']) - t0) / 1e3:.1f} us")
