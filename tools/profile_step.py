"""Host-side cProfile of model.step on the bench workload (where do the non-kernel ms go?)."""
import cProfile
import pstats
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from evo_amd.engine import Engine
from evo_amd.models import BSC, SSSC

cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "c2"]
eng = Engine()
cls = BSC if cfg["algo"] == "ebsc" else SSSC
model = cls(cfg["D"], cfg["H"], cfg["S"], rng="device", sync_host=False, engine=eng, seed=17)
my_data, theta, suff = bench.make_problem(cfg, 1236, model)
for _ in range(3):
    F, nu, nsub, theta = model.step(theta, suff, my_data)
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    F, nu, nsub, theta = model.step(theta, suff, my_data)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
