"""cProfile of the host side of model.step() on the bench workload: python tools/pyprof_step.py c2 300"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from evo_amd.engine import Engine
from evo_amd.models import BSC, SSSC
cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "c2"]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
eng = Engine()
cls = BSC if cfg["algo"] == "ebsc" else SSSC
model = cls(cfg["D"], cfg["H"], cfg["S"], rng="device", sync_host=False, engine=eng, seed=17, device_mstep=True)
my_data, theta, suff = bench.make_problem(cfg, 1236, model)
for _ in range(5):
    F, nu, nsub, theta = model.step(theta, suff, my_data)
eng.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    F, nu, nsub, theta = model.step(theta, suff, my_data)
eng.synchronize()
print("plain: %.1f us/step" % ((time.perf_counter() - t0) / steps * 1e6))
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    F, nu, nsub, theta = model.step(theta, suff, my_data)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(18)
