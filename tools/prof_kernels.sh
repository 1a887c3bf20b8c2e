#!/bin/bash
# per-kernel durations of a short bench run: tools/prof_kernels.sh <config> <filter-regex> [bench options...]
R=${GRAFT_REPO_ROOT:-/root/repo}
CFG=${1:-c4}; FILT=${2:-.}; shift 2
export TMPDIR=/tmp
OUT=$R/gpurun_out/prof_quick
rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --config $CFG --steps 2 --warmup 1 --em-per-step 5 --no-cpu-baseline --inprocess-init "$@" > $OUT/bench.json 2> $OUT/trace.log || { tail -5 $OUT/trace.log; exit 1; }
python3 - "$OUT" "$FILT" <<'PY'
import csv, glob, re, sys
out, filt = sys.argv[1], sys.argv[2]
f = glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("%-70s %7s %10s %8s" % ("kernel", "calls", "avg us", "% time"))
for r in rows:
    name = r["Name"].split("(")[0]
    if re.search(filt, name):
        print("%-70s %7s %10.2f %8.2f" % (name[:70], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
