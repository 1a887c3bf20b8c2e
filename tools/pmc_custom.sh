#!/bin/bash
# one --pmc pass with a caller-given counter list: tools/pmc_custom.sh <config> <kernel substring> "<counters>" [tag]   (PMC_STEPS / PMC_EM: bench steps / iterations per step)
CFG=${1:-c4}; KSUB=$2; CTRS=$3; TAG=${4:-pmcx}
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
OUT=$R/gpurun_out/${TAG}_${CFG}
rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --pmc $CTRS --output-format csv -d $OUT/p -- python3 $R/bench.py --config $CFG --steps ${PMC_STEPS:-3} --warmup 1 --em-per-step ${PMC_EM:-2} --no-cpu-baseline --inprocess-init $BENCH_EXTRA > $OUT/p.json 2> $OUT/p.log || { echo "pass failed"; tail -5 $OUT/p.log; }
python3 - "$OUT" "$KSUB" <<'PY'
import csv, glob, sys, collections
out, ksub = sys.argv[1], sys.argv[2]
f = glob.glob(out + "/p/**/*counter_collection.csv", recursive=True)
if not f:
    print("no csv"); sys.exit(0)
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if ksub in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c, v in acc.items():
    print("%-40s avg/launch %.4g   (launches %d)" % (c, sum(v) / len(v), len(v)))
PY
