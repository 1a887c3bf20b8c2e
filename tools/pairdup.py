"""How many k = 2 states of a datapoint share their latent pair (duplication factor the ES3C statistics scatter could exploit)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from evo_amd.engine import Engine
from evo_amd.models import SSSC
cfg = dict(bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "c2"])
cfg["N"] = min(cfg["N"], 4000)
eng = Engine()
model = SSSC(cfg["D"], cfg["H"], cfg["S"], rng="device", sync_host=False, engine=eng, seed=17, device_mstep=True)
my_data, theta, suff = bench.make_problem(cfg, 1236, model)
for it in range(25):
    F, nu, nsub, theta = model.step(theta, suff, my_data)
    if it in (0, 5, 24):
        ss = eng.download_states()
        k = ss.sum(axis=2)
        tot = dist = 0
        for n in range(0, ss.shape[0], 7):
            m = k[n] == 2
            if m.any():
                idx = np.array([np.flatnonzero(r) for r in ss[n][m]])
                keys = idx[:, 0] * cfg["H"] + idx[:, 1]
                tot += len(keys)
                dist += len(np.unique(keys))
        print("step %d: k hist %s  k=2 states per datapoint %.1f, distinct pairs %.1f (x%.2f)" % (
            it + 1, np.bincount(k.ravel(), minlength=6)[:7].tolist(), tot / (ss.shape[0] / 7), dist / (ss.shape[0] / 7), tot / max(dist, 1)), flush=True)
