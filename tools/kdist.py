"""Distribution of active latents per state during the bench workload."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from evo_amd.engine import Engine
from evo_amd.models import BSC, SSSC
cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "c2"]
eng = Engine()
cls = BSC if cfg["algo"] == "ebsc" else SSSC
model = cls(cfg["D"], cfg["H"], cfg["S"], rng="device", sync_host=False, engine=eng, seed=17, device_mstep=True)
my_data, theta, suff = bench.make_problem(cfg, 1236, model)
for it in range(24):
    F, nu, nsub, theta = model.step(theta, suff, my_data)
    if it in (0, 3, 11, 23):
        ss = eng.download_states()
        k = ss.sum(axis=2).ravel()
        print("step", it + 1, "F %.4f" % F, "k hist", np.bincount(k, minlength=10)[:12].tolist(), "mean %.2f" % k.mean())
