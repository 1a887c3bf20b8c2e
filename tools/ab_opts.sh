#!/bin/bash
# interleaved A/B of whole option sets: tools/ab_opts.sh <config> <repeats> <steps> "<set A>" "<set B>" ...   (a set = "name=v name=v", "-" = none)
R=${GRAFT_REPO_ROOT:-/root/repo}
CFG=$1; REP=$2; STEPS=$3; shift 3
for r in $(seq 1 $REP); do
  for set in "$@"; do
    args=""
    if [ "$set" != "-" ]; then for o in $set; do args="$args --option $o"; done; fi
    python3 $R/bench.py --config $CFG --steps $STEPS --warmup 5 --no-cpu-baseline $args 2> /dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
k=d['config']['kernel_ms']
print('%-40s ms/iteration %.4f  stats %.3f ovf %.3f' % ('$set', d['config']['ms_per_em_iteration'], k.get('stats',{}).get('avg_ms',0), k.get('stats_overflow',{}).get('avg_ms',0)))"
  done
done
