#!/bin/bash
# kernel trace of tools/bench_inverse.py: bash tools/prof_inverse.sh 512 1024
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
OUT=$R/gpurun_out/prof_inv
rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tools/bench_inverse.py "$@" > $OUT/out.txt 2> $OUT/trace.log
cat $OUT/out.txt
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:12]:
    n = r["Name"].split("(")[0][:60]
    print("%-62s calls %5s avg %9.1f us min %9.1f max %9.1f  %5.1f %%" % (n, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3, float(r["Percentage"])))
PY
