#!/bin/bash
# kernel trace of a LATE iteration of a bench configuration (the dense-state variant needs ~60 iterations to settle):
#   tools/prof_late.sh <config> <iterations-before> [bench options...]   -> gpurun_out/prof_quick/, timeline of the last full iteration
R=${GRAFT_REPO_ROOT:-/root/repo}
CFG=${1:-c4}; PRE=${2:-60}; shift 2
export TMPDIR=/tmp
OUT=$R/gpurun_out/prof_quick
rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $R/bench.py --config $CFG --steps 1 --warmup $PRE --em-per-step 1 --no-cpu-baseline --inprocess-init "$@" > $OUT/bench.json 2> $OUT/trace.log || { tail -5 $OUT/trace.log; exit 1; }
python3 $R/tools/timeline.py $OUT/trace $((PRE - 1))
