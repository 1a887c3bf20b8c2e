#!/bin/bash
# default-length bench of one config with the per-class kernel times printed: tools/quick_bench.sh <config> <tag> [bench options...]
R=${GRAFT_REPO_ROOT:-/root/repo}
CFG=${1:-c4}; TAG=${2:-q}; shift 2
python3 $R/bench.py --config $CFG --steps 20 --warmup 5 --no-cpu-baseline "$@" > $R/gpurun_out/qb_${TAG}.json 2> $R/gpurun_out/qb_${TAG}.log || { tail -5 $R/gpurun_out/qb_${TAG}.log; exit 1; }
python3 - $R/gpurun_out/qb_${TAG}.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
k = d["config"]["kernel_ms"]
print("%s  ms/iteration %.4f  lpj frac %.3f  stats frac %.3f" % (d["config"]["workload"][:28], d["config"]["ms_per_em_iteration"], d["roofline"]["frac"], d["roofline_stats"]["frac"]))
print("  " + "  ".join("%s %.3f" % (n, v["avg_ms"] * v["launches_per_iteration"]) for n, v in k.items()))
PY
