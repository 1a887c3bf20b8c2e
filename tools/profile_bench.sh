#!/bin/bash
# rocprofv3 passes for the bench workload (run on the GPU box through gpurun):
#   1. --kernel-trace --stats            per-kernel durations over a STEADY-STATE run (25 steps x 10 EM iterations after one
#                                        warm-up step: the same population of iterations bench.py's timed region averages)
#   2. --pmc FETCH_SIZE  / WRITE_SIZE    HBM bytes (separate passes, MI355X guide "rocprofv3 PMC slots"; short runs: the
#                                        counters serialise the kernels)
# usage: tools/profile_bench.sh <config> <tag> [extra bench options, e.g. --dense-states]
# Summaries are written under gpurun_out/prof_<tag>_<config>/ ; copy the ones to keep into profiles/.
# The program after `--` is python3 itself (no env/bash wrappers) and the CPU baseline leg is off:
# it spawns worker processes, which must not happen once the profiler has initialised the GPU.
set -e
CFG=${1:-c2}
TAG=${2:-r03}
shift 2 || true
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
OUT=$R/gpurun_out/prof_${TAG}_${CFG}
rm -rf $OUT
mkdir -p $OUT
cd /tmp
TS=${TRACE_STEPS:-25}
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --config $CFG --steps $TS --warmup 1 --em-per-step 10 --no-cpu-baseline --inprocess-init "$@" > $OUT/bench_trace.json 2> $OUT/trace.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --config $CFG --steps 2 --warmup 1 --em-per-step 3 --no-cpu-baseline --inprocess-init "$@" > $OUT/bench_fetch.json 2> $OUT/fetch.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --config $CFG --steps 2 --warmup 1 --em-per-step 3 --no-cpu-baseline --inprocess-init "$@" > $OUT/bench_write.json 2> $OUT/write.log
python3 $R/tools/summarize_prof.py $OUT > $OUT/summary.md
head -60 $OUT/summary.md
