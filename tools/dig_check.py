"""A/B of the state-digest path: run-to-run determinism vs digest on/off (GPU)."""
import sys
import numpy as np
sys.path.insert(0, ".")
from evo_amd.engine import Engine
from evo_amd.models import BSC, SSSC
from evo_amd.variational import init_states

engine = Engine()
def run(algo, H, S, use, learn):
    rng = np.random.RandomState(17)
    D, N = 40, 500
    W0 = rng.normal(size=(D, H))
    Y = (rng.random_sample((N, H)) < 3.0 / H).astype(float) @ W0.T + 0.3 * rng.normal(size=(N, D))
    my_data = {"y": Y, "x_infr": np.ones_like(Y, dtype=bool)}
    cls = BSC if algo == "ebsc" else SSSC
    engine.set_option("state_digest", use)
    np.random.seed(3)
    kw = {} if learn else {"to_learn": []}
    model = cls(D, H, S, rng="device", sync_host=True, engine=engine, seed=23, **kw)
    theta = model.check_params(model.standard_init(my_data))
    suff = init_states(N, S, H, "fit", "randflip", 6, 3, 1)
    out = []
    for _ in range(8):
        F, _, _, theta = model.step(theta, suff, my_data)
        out.append((F, suff["ss"].copy()))
    engine.set_option("state_digest", 1)
    return out
for algo, H, S in [("ebsc", 200, 40), ("es3c", 136, 30)]:
    for learn in (True, False):
        a = run(algo, H, S, 1, learn); b = run(algo, H, S, 1, learn); c = run(algo, H, S, 0, learn)
        for i in range(8):
            print(algo, "learn" if learn else "fixed", "step", i, "same-setting diff", int((a[i][1] != b[i][1]).sum()),
                  "dig on/off diff", int((a[i][1] != c[i][1]).sum()), "F", a[i][0] - b[i][0], a[i][0] - c[i][0])
