# A/B of one engine option on one config: bash tools/ab.sh c2 overlap_gemm [reps]
cd /root/repo; mkdir -p gpurun_out
CFG=$1; OPT=$2; REPS=${3:-3}
for i in $(seq $REPS); do for v in 1 0; do
  timeout -k 10 200 python bench.py --config $CFG --steps 40 --warmup 5 --no-cpu-baseline --option $OPT=$v 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$CFG $OPT=$v', round(d['ms_per_step'],4))"
done; done
