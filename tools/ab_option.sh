#!/bin/bash
# A/B of one option on one config, interleaved repeats: tools/ab_option.sh <config> <option> "<values>" [repeats] [steps]
R=${GRAFT_REPO_ROOT:-/root/repo}
CFG=$1; OPT=$2; VALS=$3; REP=${4:-3}; STEPS=${5:-60}
for r in $(seq 1 $REP); do
  for v in $VALS; do
    python3 $R/bench.py --config $CFG --steps $STEPS --warmup 5 --no-cpu-baseline --option $OPT=$v 2> /dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$OPT=$v  ms/iteration %.4f' % d['config']['ms_per_em_iteration'])"
  done
done
