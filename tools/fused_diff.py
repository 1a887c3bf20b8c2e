#!/usr/bin/env python3
"""Where do the fused E-step and the separate passes differ?  (diagnosis aid: python tools/fused_diff.py c2_small)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from evo_amd.engine import Engine  # noqa: E402
from evo_amd.models import SSSC  # noqa: E402
from evo_amd.variational import init_states  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "c2_small"
g = dict(np.load(os.path.join(ROOT, "tests", "golden", "shape_%s.npz" % name)))
D, H, S, N, seed = (int(g[k]) for k in ("D", "H", "S", "N", "seed"))
np.random.seed(seed)
Y = np.random.randn(N, D)
my_data = {"y": Y, "x_infr": np.ones_like(Y, dtype=bool)}
eng = Engine()
m0 = SSSC(D, H, S, engine=eng)
theta0 = m0.check_params(m0.standard_init(my_data))
ss0 = init_states(N, S, H, "fit", "randflip", 10, 1, 1)["ss"]
res = []
for fused in (0, 2):
    eng.set_option("fused_estep", fused)
    model = SSSC(D, H, S, rng="device", sync_host=True, engine=eng, seed=23, device_mstep=False, to_learn=[])
    theta = model.check_params({k: (np.array(v, copy=True) if isinstance(v, np.ndarray) else v) for k, v in theta0.items()})
    np.random.seed(1)
    suff = init_states(N, S, H, "fit", "randflip", 10, 1, 1)
    suff["ss"][:] = ss0
    before = suff["ss"].copy()
    F, nu, nsub, theta = model.step(theta, suff, my_data)
    res.append((suff["ss"].copy(), suff["lpj"].copy(), before))
(ssA, lA, b0), (ssB, lB, _) = res
print("K^n equal:", np.array_equal(ssA, ssB))
d = lA != lB
print("lpj entries that differ:", int(d.sum()), "of", d.size)
k = ssA.sum(axis=-1)
new = (ssA != b0).any(axis=-1)
for kk in range(0, 12):
    m = k == kk
    if m.any():
        print("k=%d: %d states, %d differ (of them %d new this step); new states total %d" %
              (kk, m.sum(), (d & m).sum(), (d & m & new).sum(), (m & new).sum()))
if d.any():
    rel = np.abs(lA - lB)[d] / np.abs(lA[d])
    print("max rel diff", rel.max())

# --- where does a difference come from?  (1) the separate passes twice; (2) the fused values against a re-evaluation
def one(fused):
    eng.set_option("fused_estep", fused)
    model = SSSC(D, H, S, rng="device", sync_host=True, engine=eng, seed=23, device_mstep=False, to_learn=[])
    theta = model.check_params({k: (np.array(v, copy=True) if isinstance(v, np.ndarray) else v) for k, v in theta0.items()})
    np.random.seed(1)
    suff = init_states(N, S, H, "fit", "randflip", 10, 1, 1)
    suff["ss"][:] = ss0
    model.step(theta, suff, my_data)
    l_step = suff["lpj"].copy()
    eng.lpj_resident()
    l_re = eng.download_lpj()
    return l_step, l_re


a1, a1r = one(0)
a2, a2r = one(0)
f1, f1r = one(2)
print("separate twice: step values differ in", int((a1 != a2).sum()), "; re-evaluations differ in", int((a1r != a2r).sum()))
print("separate: step vs re-evaluation differ in", int((a1 != a1r).sum()))
print("fused: step vs re-evaluation differ in", int((f1 != f1r).sum()))
print("re-evaluations separate vs fused differ in", int((a1r != f1r).sum()))
