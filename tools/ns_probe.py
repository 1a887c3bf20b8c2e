#!/usr/bin/env python3
"""How far do the H x H moment matrices of the Theta update move between EM iterations?
Runs a bench configuration with the host Theta update and prints, per iteration t, the residual
||I - A_t inv(A_{t-1})|| (2-norm and Frobenius) of every matrix the update inverts -- the starting error of a
Newton-Schulz iteration X <- X (2 I - A X) warm-started from the previous iteration's inverse (it converges
quadratically iff the 2-norm is below 1).      python tools/ns_probe.py c4shard 120"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "c4shard"
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 120
    cfg = bench.CONFIGS[name]
    from evo_amd.engine import Engine
    from evo_amd.models import BSC, SSSC
    from evo_amd.utils import parallel
    np.random.seed(1234 + 2)
    Y = np.ascontiguousarray(np.random.randn(cfg["N"], cfg["D"]))
    my_data = {"y": Y, "x_infr": np.ones_like(Y, dtype=bool)}
    chunks = list(bench.init_states_inprocess(cfg, cfg["N"], 4321))
    eng = Engine()
    comm = parallel.init_rccl_from_env(eng)
    cls = BSC if cfg["algo"] == "ebsc" else SSSC
    model = cls(cfg["D"], cfg["H"], cfg["S"], comm=comm, rng="device", sync_host=False, engine=eng, seed=17,
                device_mstep=False)
    np.random.seed(99)
    theta = model.check_params(model.standard_init(my_data))
    suff = bench.ea_suff(cfg)
    model.attach_resident_states(suff, my_data, chunks)
    seen = {}
    prev_inv = {}
    rows = []
    real_inv, real_lstsq = np.linalg.inv, np.linalg.lstsq

    def note(A, key):
        if A.shape[0] != cfg["H"]:
            return
        if key in prev_inv:
            E = np.eye(A.shape[0]) - A @ prev_inv[key]
            rows.append((key, np.linalg.norm(E, 2), np.linalg.norm(E, "fro"), np.linalg.cond(A)))
        prev_inv[key] = real_inv(A)

    def inv(A):
        k = seen["n"] = seen.get("n", 0) + 1
        note(np.array(A), "inv%d" % k)
        return real_inv(A)

    def lstsq(A, B, rcond=None):
        note(np.array(A), "lstsq")
        return real_lstsq(A, B, rcond=rcond)

    np.linalg.inv, np.linalg.lstsq = inv, lstsq
    for t in range(iters):
        seen["n"] = 0
        rows.clear()
        F, nu, nsub, theta = model.step(theta, suff, my_data)
        if t < 12 or t % 10 == 0:
            print("iteration %3d  F %.6f  " % (t, F) + "   ".join("%s: |E|2 %.3g |E|F %.3g cond %.2g" % r for r in rows), flush=True)


if __name__ == "__main__":
    main()
