"""The fused per-datapoint E-step (csrc/kernels_fused.hpp, evoamd_estep) against the separate passes it replaces
(evoamd_lpj_resident + evoamd_evolve_randflip + evoamd_vary_kn): for the same device seed the two must leave the SAME
K^n, the same lpj bits, the same free-energy term and counters, and accumulators that agree to 1e-11 -- at the BASELINE
shapes (the inputs of the shape_* fixtures: same seeds, same init_states), on a dense K^n (states with 5..8 and more
latents: the FULL instantiation with the bit-word paths), and in exact mode (a duplicated latent: every state above two
latents on the pivoting form).  The separate passes are themselves pinned to the reference by test_shape_trajectory /
test_step_kernels; this file pins the fused kernel to them.  Run with ``-m gpu`` on an MI355X."""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine():
    from evo_amd.engine import Engine
    eng = Engine()
    yield eng
    eng.close()


def _run(engine, fused, cls, D, H, S, my_data, theta0, ss0, ea, n_steps, device_mstep, to_learn=None, screen=None):
    """n_steps of model.step() in rng="device" mode; returns per step (packed K^n, lpj, F, nu, nsub, acc or Theta)."""
    from evo_amd.variational import init_states
    N = ss0.shape[0]
    engine.set_option("fused_estep", 2 if fused else 0)
    if screen is not None:
        engine.set_option("lpj_singular_screen", screen)
    try:
        kw = {} if to_learn is None else {"to_learn": to_learn}
        model = cls(D, H, S, rng="device", sync_host=True, engine=engine, seed=23, device_mstep=device_mstep, **kw)
        theta = {k: (np.array(v, copy=True) if isinstance(v, np.ndarray) else v) for k, v in theta0.items()}
        theta = model.check_params(theta)
        np.random.seed(1)
        suff = init_states(N, S, H, ea[0], ea[1], ea[2], ea[3], 1)
        suff["ss"][:] = ss0
        out, used = [], []
        for _ in range(n_steps):
            F, nu, nsub, theta = model.step(theta, suff, my_data)
            used.append(model.last_estep_fused)
            rec = {"ss": np.packbits(suff["ss"], axis=-1), "lpj": suff["lpj"].copy(), "F": F, "nu": nu, "nsub": nsub}
            if device_mstep:
                rec["theta"] = {k: np.array(v) for k, v in theta.items() if isinstance(v, (np.ndarray, float, np.floating))}
            else:
                rec["acc"] = model.last_acc.copy()
            out.append(rec)
        return out, used
    finally:
        engine.set_option("fused_estep", 1)
        if screen is not None:
            engine.set_option("lpj_singular_screen", 1)


def _compare(sep, fus, device_mstep):
    for t, (a, b) in enumerate(zip(sep, fus)):
        assert np.array_equal(a["ss"], b["ss"]), "K^n differs at step %d" % t
        assert np.array_equal(a["lpj"], b["lpj"]), "lpj differs at step %d (max %g)" % (t, np.abs(a["lpj"] - b["lpj"]).max())
        assert a["F"] == b["F"], (t, a["F"], b["F"])
        assert a["nu"] == b["nu"] and a["nsub"] == b["nsub"], t
        if device_mstep:
            for k in a["theta"]:
                np.testing.assert_allclose(b["theta"][k], a["theta"][k], rtol=1e-9, atol=1e-12, err_msg="%s step %d" % (k, t))
        else:
            scale = max(1.0, float(np.abs(a["acc"]).max()))
            assert np.abs(a["acc"] - b["acc"]).max() <= 1e-11 * scale, t


@pytest.mark.parametrize("name,device_mstep", [("c2_small", False), ("c4_small", False), ("c2", False), ("c2", True),
                                               ("c4", False), ("c4", True)])
def test_fused_estep_matches_separate_passes_at_baseline_shapes(engine, name, device_mstep):
    """The inputs of the ES3C shape fixtures (true D, H, S of BASELINE configs[1] and configs[3]; N = 12 ... 1536), three EM
    iterations with the device generator: fused and separate paths bit-identical in K^n, lpj, F and the counters; the
    accumulators (host M-step) to 1e-11, Theta^new (device M-step) to 1e-9."""
    from evo_amd.models import SSSC
    from evo_amd.variational import init_states
    g = load_golden("shape_%s.npz" % name)
    D, H, S, N, seed = (int(g[k]) for k in ("D", "H", "S", "N", "seed"))
    np.random.seed(seed)
    Y = np.random.randn(N, D)
    my_data = {"y": Y, "x_infr": np.ones_like(Y, dtype=bool)}
    model0 = SSSC(D, H, S, engine=engine)
    theta0 = model0.check_params(model0.standard_init(my_data))
    ea = (str(g["ea_parent_selection"]), str(g["ea_mutation"]), int(g["ea_n_parents"]), int(g["ea_n_children"]))
    ss0 = init_states(N, S, H, ea[0], ea[1], ea[2], ea[3], 1)["ss"]
    n_steps = 3
    to_learn = [] if name.endswith("_small") else None  # N << H: the Theta update is ill-posed, Theta stays fixed
    sep, used_s = _run(engine, False, SSSC, D, H, S, my_data, theta0, ss0, ea, n_steps, device_mstep, to_learn=to_learn)
    fus, used_f = _run(engine, True, SSSC, D, H, S, my_data, theta0, ss0, ea, n_steps, device_mstep, to_learn=to_learn)
    assert not any(used_s) and all(used_f), (used_s, used_f)
    _compare(sep, fus, device_mstep)


@pytest.mark.parametrize("H,S,p_on", [(64, 40, 6.0), (136, 70, 5.0), (512, 200, 4.0)])
def test_fused_estep_dense_states(engine, H, S, p_on):
    """A K^n whose states hold ~p_on active latents (3..4, 5..8 and above eight all occur): the FULL instantiation --
    latents from the bit words, children's words written and read back inside the kernel, the 5..8 quad form and the
    pivoting form in-wave -- against the level chains of the separate passes."""
    from evo_amd.models import SSSC
    rng = np.random.RandomState(7)
    D, N = 24, 96
    W0 = rng.normal(size=(D, H))
    Y = (rng.random_sample((N, H)) < p_on / H).astype(float) @ W0.T + 0.3 * rng.normal(size=(N, D))
    my_data = {"y": Y, "x_infr": np.ones_like(Y, dtype=bool)}
    np.random.seed(5)
    model0 = SSSC(D, H, S, engine=engine)
    theta0 = model0.check_params(model0.standard_init(my_data))
    ss0 = np.zeros((N, S, H), dtype=bool)
    for n in range(N):
        seen = set()
        s = 0
        while s < S:
            k = min(H, max(0, int(rng.poisson(p_on)))) if s % 7 else int(rng.randint(0, 3))
            k = min(k, 12)
            row = np.zeros(H, dtype=bool)
            row[rng.choice(H, k, replace=False)] = True
            key = row.tobytes()
            if key in seen:
                continue
            seen.add(key)
            ss0[n, s] = row
            s += 1
    ea = ("fit", "randflip", 8, 2)
    sep, _ = _run(engine, False, SSSC, D, H, S, my_data, theta0, ss0, ea, 3, False)
    fus, used = _run(engine, True, SSSC, D, H, S, my_data, theta0, ss0, ea, 3, False)
    assert all(used)
    k = np.unpackbits(sep[0]["ss"], axis=-1).sum(axis=-1)
    assert (k >= 5).any() and (k >= 9).any() and ((k >= 3) & (k <= 4)).any()
    _compare(sep, fus, False)


def test_fused_estep_exact_mode(engine):
    """Psi with a duplicated latent (an exactly singular 2 x 2 principal block): the tables kernel stamps the Theta and
    every state above two latents goes through the pivoting form, which follows the reference's pinv branches
    (sssc.py:278-301).  Fused (FAST defers such datapoints, FULL pivots in-wave) against the separate passes."""
    from evo_amd.models import SSSC
    rng = np.random.RandomState(3)
    D, H, S, N = 20, 48, 30, 64
    Y = rng.normal(size=(N, D))
    my_data = {"y": Y, "x_infr": np.ones_like(Y, dtype=bool)}
    np.random.seed(2)
    model0 = SSSC(D, H, S, engine=engine)
    theta0 = model0.check_params(model0.standard_init(my_data))
    Psi = np.eye(H)
    Psi[3, 7] = Psi[7, 3] = 1.0  # latents 3 and 7 duplicated: [[1, 1], [1, 1]]
    theta0["Psi"] = Psi
    ss0 = np.zeros((N, S, H), dtype=bool)
    for n in range(N):
        seen = set()
        s = 0
        while s < S:
            k = int(rng.randint(0, 6))
            row = np.zeros(H, dtype=bool)
            row[rng.choice(H, k, replace=False)] = True
            if s % 3 == 0:
                row[[3, 7]] = True
            key = row.tobytes()
            if key in seen:
                continue
            seen.add(key)
            ss0[n, s] = row
            s += 1
    ea = ("fit", "randflip", 6, 1)
    sep, _ = _run(engine, False, SSSC, D, H, S, my_data, theta0, ss0, ea, 2, False, to_learn=[])
    fus, used = _run(engine, True, SSSC, D, H, S, my_data, theta0, ss0, ea, 2, False, to_learn=[])
    assert all(used)
    assert (sep[0]["lpj"] == 0.0).any()  # B_max: the reference's +inf for an exactly singular Psi_A
    _compare(sep, fus, False)


def test_fused_estep_automatic_choice(engine):
    """Option "fused_estep" = 1 (default): the first E-step of a geometry runs the separate passes (no census yet), the
    following ones the fused kernel while K^n is sparse; a K^n with states above eight latents goes back to the separate
    passes (whose levels hold 64 latents)."""
    from evo_amd.models import SSSC
    from evo_amd.variational import init_states
    rng = np.random.RandomState(9)
    D, H, S, N = 16, 64, 24, 200
    Y = rng.normal(size=(N, D))
    my_data = {"y": Y, "x_infr": np.ones_like(Y, dtype=bool)}
    np.random.seed(4)
    model = SSSC(D, H, S, rng="device", sync_host=True, engine=engine, seed=3, device_mstep=True)
    theta = model.check_params(model.standard_init(my_data))
    suff = init_states(N, S, H, "fit", "randflip", 6, 1, 1)
    used = []
    for _ in range(3):
        _, _, _, theta = model.step(theta, suff, my_data)
        used.append(model.last_estep_fused)
    assert used == [False, True, True], used
    for s in range(S):  # every state of datapoint 0 gets 12 active latents (at most 6 of them are replaced per step)
        suff["ss"][0, s] = False
        suff["ss"][0, s, s:s + 12] = True
    _, _, _, theta = model.step(theta, suff, my_data)   # the census of THIS step sees it ...
    _, _, _, theta = model.step(theta, suff, my_data)   # ... so the next one takes the separate passes
    assert model.last_estep_fused is False
