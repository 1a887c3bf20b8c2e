"""Timeline of one steady-state EM iteration from a rocprofv3 kernel trace: kernel, duration, idle gap before it.
usage: python tools/gaps.py <dir with *_kernel_trace.csv> [anchor kernel substring]"""
import csv, glob, sys
d = sys.argv[1]
anchor = sys.argv[2] if len(sys.argv) > 2 else "lpj_gram_kernel<0>|sssc_main_lpj_kernel<0"
import os
f = max(glob.glob(d + "/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if any(a in r["Kernel_Name"] for a in anchor.split("|"))]
if len(idx) < 4:
    sys.exit("anchor not found often enough")
k = len(idx) // 2
a, b = idx[k], idx[k + 1]  # one full iteration of the timed region (the last ones are the instrumented pass)
t0 = int(rows[a]["Start_Timestamp"])
prev_end = int(rows[a - 1]["End_Timestamp"])
busy = 0.0
gaps = 0.0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3
    dur = (e - s) / 1e3
    busy += dur
    gaps += max(gap, 0.0)
    print("%9.1f us  gap %7.1f  dur %8.1f  %s" % ((s - t0) / 1e3, gap, dur, r["Kernel_Name"].split("(")[0][:70]))
    prev_end = max(prev_end, e)
print("iteration %.1f us: kernels %.1f us, idle %.1f us, launches %d" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e3, busy, gaps, b - a))
