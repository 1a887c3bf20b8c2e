"""lpj_resident kernel time vs the distribution of active latents per state (ES3C, c2 shape)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from evo_amd.engine import Engine

N, D, H, S = 10000, 256, 128, 64
rng = np.random.RandomState(0)
eng = Engine()
eng.configure("sssc", N, D, H, S, 0, 10)
Y = rng.normal(size=(N, D))
eng.upload_data(Y)
W = rng.normal(size=(D, H)) * 0.1
eng.set_params_sssc(W, rng.uniform(0.1, 0.5, H), rng.normal(size=H), np.eye(H), 1.0)


def states_with_k(kfun):
    ss = np.zeros((N, S, H), dtype=bool)
    ks = kfun((N, S))
    r = rng.random_sample((N, S, H)).argsort(axis=2)
    for k in range(0, ks.max() + 1):
        m = ks >= k + 1
        if not m.any():
            continue
        idx = r[..., k]
        nn, sss = np.nonzero(m)
        ss[nn, sss, idx[nn, sss]] = True
    return ss


cases = {
    "all k=1": lambda sh: np.ones(sh, int),
    "all k=2": lambda sh: np.full(sh, 2),
    "k in {0,1,2}": lambda sh: rng.randint(0, 3, sh),
    "92% k<=2, 8% k in {3,4}": lambda sh: np.where(rng.random_sample(sh) < 0.92, rng.randint(0, 3, sh), rng.randint(3, 5, sh)),
    "all k=4": lambda sh: np.full(sh, 4),
    "50% k<=2, 50% k=3..4": lambda sh: np.where(rng.random_sample(sh) < 0.5, rng.randint(0, 3, sh), rng.randint(3, 5, sh)),
    "all k=8": lambda sh: np.full(sh, 8),
}
for name, f in cases.items():
    eng.upload_states(states_with_k(f))
    for _ in range(3):
        eng.lpj_resident()
    eng.synchronize()
    eng.timing(True)
    eng.timing_reset()
    for _ in range(20):
        eng.lpj_resident()
    eng.synchronize()
    a, n = eng.kernel_time_ms("lpj_resident")
    b, m = eng.kernel_time_ms("lpj_overflow")
    eng.timing(False)
    print("%-28s main %.1f us   overflow chain %.1f us" % (name, a * 1e3, b * 1e3))
