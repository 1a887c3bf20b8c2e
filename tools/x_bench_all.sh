#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python3 bench.py --steps 20 --warmup 5 > gpurun_out/bl_c4.json 2> gpurun_out/bl_c4.log || { tail -5 gpurun_out/bl_c4.log; exit 1; }
echo done c4
for c in c2 c3 c5 c5f32 c4half c4quarter c4shard; do
  python3 bench.py --config $c --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bl_$c.json 2> gpurun_out/bl_$c.log || { tail -5 gpurun_out/bl_$c.log; exit 1; }
  echo done $c
done
python3 bench.py --config c4 --dense-states --steps 10 --warmup 5 --no-cpu-baseline > gpurun_out/bl_c4_dense.json 2> gpurun_out/bl_c4_dense.log || { tail -5 gpurun_out/bl_c4_dense.log; exit 1; }
echo done dense
