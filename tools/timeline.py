#!/usr/bin/env python3
"""One EM iteration of a rocprofv3 kernel trace as a timeline (start, end, duration, gap to the previous end, queue):
    python tools/timeline.py gpurun_out/prof_quick/trace [iteration-index]
An iteration runs from one vary_kn_kernel to the next; index counts from the first (default: the 7th)."""
import csv
import glob
import os
import sys

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_quick/trace"
which = int(sys.argv[2]) if len(sys.argv) > 2 else 6
files = sorted(glob.glob(os.path.join(root, "**", "*_kernel_trace.csv"), recursive=True), key=os.path.getmtime)
rows = list(csv.DictReader(open(files[-1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "vary_kn" in r["Kernel_Name"]]
i0, i1 = idx[which], idx[which + 1]
t0 = int(rows[i0]["Start_Timestamp"])
prev_end = t0
busy = 0
for r in rows[i0:i1]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("%8.1f %8.1f %7.1f  gap %6.1f  q%s %s" % (s / 1e3, e / 1e3, (e - s) / 1e3, (s - prev_end + t0) / 1e3 if False else (s - (prev_end - t0)) / 1e3,
                                                    r["Queue_Id"], r["Kernel_Name"].split("(")[0][:60]))
    prev_end = max(prev_end, e + t0)
print("iteration: %.1f us, %d kernels" % ((int(rows[i1]["Start_Timestamp"]) - t0) / 1e3, i1 - i0))
