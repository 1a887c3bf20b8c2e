#!/usr/bin/env python3
"""Distribution of the posterior weights q_ns = exp(lpj_ns - max_s lpj_ns) and of |s| over K^n after T EM
iterations of a bench workload: how many (n, s) terms of the statistics pass are numerically nothing?
    python tools/qdist.py --config c4 --iters 60 [--n 20000]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c4")
    ap.add_argument("--iters", type=int, default=60)
    ap.add_argument("--n", type=int, default=20000)
    a = ap.parse_args()
    cfg = dict(bench.CONFIGS[a.config])
    cfg["N"] = min(cfg["N"], a.n)
    from evo_amd.engine import Engine
    from evo_amd.models import BSC, SSSC
    np.random.seed(1236)
    Y = np.random.randn(cfg["N"], cfg["D"])
    my_data = {"y": Y, "x_infr": np.ones_like(Y, dtype=bool)}
    chunks = list(bench.init_states_packed(cfg, cfg["N"], 4321, bench.host_cores()))
    eng = Engine()
    model = (BSC if cfg["algo"] == "ebsc" else SSSC)(cfg["D"], cfg["H"], cfg["S"], rng="device", sync_host=False,
                                                      engine=eng, seed=17, device_mstep=True)
    np.random.seed(99)
    theta = model.check_params(model.standard_init(my_data))
    suff = bench.ea_suff(cfg)
    model.attach_resident_states(suff, my_data, chunks)
    for it in range(a.iters + 1):
        F, nu, nsub, theta = model.step(theta, suff, my_data)
        if it in (0, 5, 20, a.iters):
            lpj = eng.download_lpj()
            d = lpj - lpj.max(axis=1, keepdims=True)
            k = np.unpackbits(eng.download_states_packed(), axis=-1).sum(axis=-1)
            fr = [float((d < -t).mean()) for t in (20, 40, 69, 700)]
            print("iter %3d F %.4f  frac(q < e^-20, e^-40, e^-69, underflow) = %s  mean|s| %.3f  frac k>2 %.4f k>4 %.5f  "
                  "eff states/dp (q>e^-40) %.1f" % (it, F, np.round(fr, 4), k.mean(), (k > 2).mean(), (k > 4).mean(),
                                                   (d >= -40).sum(axis=1).mean()), flush=True)
    eng.close()


if __name__ == "__main__":
    main()
