#!/usr/bin/env python3
"""bench.py with clocks around the two library calls of an EM iteration (where does the host spend an iteration?):
    python tools/hostdiag.py [bench options]
prints the bench line on stdout and one [diag] line on stderr."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import bench
    from evo_amd import engine as E
    acc = {"estep": 0.0, "mstep": 0.0, "n": 0}
    _e, _m = E.Engine.estep, E.Engine.mstep_device

    delay = float(os.environ.get("HOSTDIAG_DELAY_US", "0")) * 1e-6  # artificial host work in front of every E-step call

    def estep(self, *a, **k):
        if delay:
            t_end = time.perf_counter() + delay
            while time.perf_counter() < t_end:
                pass
        t = time.perf_counter()
        r = _e(self, *a, **k)
        acc["estep"] += time.perf_counter() - t
        return r

    acc["probe1"] = acc["probe2"] = 0.0

    def mstep(self, *a, **k):
        t = time.perf_counter()
        r = _m(self, *a, **k)
        t1 = time.perf_counter()
        acc["mstep"] += t1 - t
        acc["n"] += 1
        x = 0
        for i in range(300):  # a fixed piece of interpreter work right behind the blocking call, and again
            x += i
        t2 = time.perf_counter()
        for i in range(300):
            x += i
        t3 = time.perf_counter()
        acc["probe1"] += t2 - t1
        acc["probe2"] += t3 - t2
        return r

    E.Engine.estep, E.Engine.mstep_device = estep, mstep
    from evo_amd.models import _models as M
    _s, _p = M.Model.step, M.Model._prepare
    acc["step"] = acc["prep"] = 0.0

    def step(self, *a, **k):
        t = time.perf_counter()
        r = _s(self, *a, **k)
        acc["step"] += time.perf_counter() - t
        return r

    def prep(self, *a, **k):
        t = time.perf_counter()
        r = _p(self, *a, **k)
        acc["prep"] += time.perf_counter() - t
        return r

    M.Model.step, M.Model._prepare = step, prep
    sys.argv = ["bench.py"] + sys.argv[1:]
    try:
        bench.main()
    finally:
        n = max(1, acc["n"])
        sys.stderr.write("[diag] 300-iteration Python loop right behind mstep_device: %.1f us, repeated: %.1f us\n"
                         % (1e6 * acc["probe1"] / n, 1e6 * acc["probe2"] / n))
        sys.stderr.write("[diag] calls %d: in estep %.3f ms, in mstep_device %.3f ms, whole step() %.3f ms, of it _prepare %.3f ms "
                         "per call\n" % (n, 1e3 * acc["estep"] / n, 1e3 * acc["mstep"] / n, 1e3 * acc["step"] / n,
                                         1e3 * acc["prep"] / n))


if __name__ == "__main__":  # (the CPU baseline's spawned workers import this module: they must not run it)
    main()
