#!/bin/bash
# kernel-trace only, any config: bash tools/profile_quick.sh c5
CFG=${1:-c5}
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
OUT=$R/gpurun_out/quick_${CFG}
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --config $CFG --steps 12 --warmup 2 --no-cpu-baseline > $OUT/bench.json 2> $OUT/trace.log
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:22]:
    n = r["Name"].split("(")[0][:60]
    print("%-62s calls %5s avg %9.1f us  %5.1f %%" % (n, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
