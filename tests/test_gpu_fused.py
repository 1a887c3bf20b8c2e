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


def _lockstep(engine, cls, D, H, S, my_data, theta0, ss0, ea, n_steps, to_learn=None):
    """n_steps EM iterations (rng="device", host M-step).  Every E-step is run TWICE from the same K^n and the same
    Theta -- by the separate passes on a copy of the state bag, then by the fused kernel inside model.step(), which also
    carries the trajectory on -- and compared: K^n, lpj, F and the counters bit for bit, the accumulators to 1e-11 (the
    statistics pass adds with f64 atomics: its sums are reproducible to ~1e-16 only, so two TRAJECTORIES drift apart in
    the last bits of Theta after the first M-step; the E-step itself has no such freedom).  Returns K^n after each step."""
    from evo_amd.variational import init_states
    N = ss0.shape[0]
    kw = {} if to_learn is None else {"to_learn": to_learn}
    model = cls(D, H, S, rng="device", sync_host=True, engine=engine, seed=23, device_mstep=False, **kw)
    theta = {k: (np.array(v, copy=True) if isinstance(v, np.ndarray) else v) for k, v in theta0.items()}
    theta = model.check_params(theta)
    np.random.seed(1)
    suff = init_states(N, S, H, ea[0], ea[1], ea[2], ea[3], 1)
    suff["ss"][:] = ss0
    out = []
    # (bit for bit means: the same arithmetic per state.  The separate passes hand the 5..8-latent states to the pivoting
    # wavefront kernel when there are only a few of them -- option "merge_small_levels" --, the fused kernel always
    # eliminates them four lanes per state: same values to ~1e-13, not the same bits.  Off for this comparison.)
    engine.set_option("merge_small_levels", 0)
    try:
        for t in range(n_steps):
            sep = dict(suff)
            sep["ss"], sep["lpj"] = suff["ss"].copy(), suff["lpj"].copy()
            engine.set_option("fused_estep", 0)
            Fa, nua, nsuba = model.E_step(theta, sep, my_data)
            assert model.last_estep_fused is False
            acc_a = model.last_acc.copy()
            model._n_steps -= 1  # the same device seed for the second evaluation of this E-step
            engine.set_option("fused_estep", 2)
            Fb, nub, nsubb, theta = model.step(theta, suff, my_data)
            assert model.last_estep_fused is True
            acc_b = model.last_acc
            assert np.array_equal(sep["ss"], suff["ss"]), "K^n differs at step %d" % t
            same = sep["lpj"] == suff["lpj"]
            assert same.all(), "lpj differs at step %d in %d entries (max %g)" % (t, (~same).sum(), np.abs(sep["lpj"] - suff["lpj"]).max())
            assert Fa == Fb and nua == nub and nsuba == nsubb, (t, Fa, Fb, nua, nub, nsuba, nsubb)
            scale = max(1.0, float(np.abs(acc_a).max()))
            assert np.abs(acc_a - acc_b).max() <= 1e-11 * scale, t
            out.append(suff["ss"].copy())
    finally:
        engine.set_option("fused_estep", 0)
        engine.set_option("merge_small_levels", 1)
    return out


@pytest.mark.parametrize("name", ["c2_small", "c4_small", "c2", "c4"])
def test_fused_estep_matches_separate_passes_at_baseline_shapes(engine, name):
    """The inputs of the ES3C shape fixtures (true D, H, S of BASELINE configs[1] and configs[3]; N = 12 ... 1536), three EM
    iterations with the device generator: every E-step by both paths from the same K^n and Theta -- K^n, lpj, F and the
    counters bit-identical, the accumulators to 1e-11.  (configs[2] and configs[4] are EBSC: no fused kernel yet.)"""
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
    to_learn = [] if name.endswith("_small") else None  # N << H: the Theta update is ill-posed, Theta stays fixed
    _lockstep(engine, SSSC, D, H, S, my_data, theta0, ss0, ea, 3, to_learn=to_learn)


def test_fused_estep_device_mstep_trajectory(engine):
    """The configuration bench.py times (device M-step, K^n resident) over five iterations, fused against separate: the
    two trajectories share every E-step decision while Theta agrees (first step: bit for bit) and stay within the
    reproducibility of the atomic sums afterwards -- F to 1e-10, Theta to 1e-8 after five steps."""
    from evo_amd.models import SSSC
    from evo_amd.variational import init_states
    g = load_golden("shape_c2.npz")
    D, H, S, N, seed = (int(g[k]) for k in ("D", "H", "S", "N", "seed"))
    np.random.seed(seed)
    Y = np.random.randn(N, D)
    my_data = {"y": Y, "x_infr": np.ones_like(Y, dtype=bool)}
    res = []
    engine.set_option("merge_small_levels", 0)  # (one arithmetic per state in both runs, see _lockstep)
    try:
        for opt in (0, 2):
            engine.set_option("fused_estep", opt)
            np.random.seed(seed + 1)
            model = SSSC(D, H, S, rng="device", sync_host=False, engine=engine, seed=5, device_mstep=True)
            theta = model.check_params(model.standard_init(my_data))
            suff = init_states(N, S, H, "fit", "randflip", 10, 1, 1)
            Fs, used = [], []
            for _ in range(5):
                F, nu, nsub, theta = model.step(theta, suff, my_data)
                Fs.append(F)
                used.append(model.last_estep_fused)
            assert used == [opt == 2] * 5
            res.append((Fs, {k: np.array(v) for k, v in theta.items()}))
    finally:
        engine.set_option("fused_estep", 0)
        engine.set_option("merge_small_levels", 1)
    assert res[0][0][0] == res[1][0][0]
    np.testing.assert_allclose(res[1][0], res[0][0], rtol=1e-10)
    for k in ("W", "pies", "mus", "Psi", "sigma2"):
        np.testing.assert_allclose(res[1][1][k], res[0][1][k], rtol=1e-8, atol=1e-10, err_msg=k)


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
            if s % 11 == 5 and n % 3 == 0:
                k = 17 + (s % 5)  # above the 16 latents of the first FULL launch: the datapoint moves on to the second
            row = np.zeros(H, dtype=bool)
            row[rng.choice(H, k, replace=False)] = True
            key = row.tobytes()
            if key in seen:
                continue
            seen.add(key)
            ss0[n, s] = row
            s += 1
    ea = ("fit", "randflip", 8, 2)
    kn = _lockstep(engine, SSSC, D, H, S, my_data, theta0, ss0, ea, 3)
    k = ss0.sum(axis=-1)  # (what the first E-step evaluates)
    assert (k >= 5).any() and (k >= 9).any() and ((k >= 3) & (k <= 4)).any() and (k >= 17).any()
    assert (kn[-1].sum(axis=-1) >= 5).any()


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
    _lockstep(engine, SSSC, D, H, S, my_data, theta0, ss0, ea, 2, to_learn=[])
    engine.lpj_resident()
    assert (engine.download_lpj() == 0.0).any()  # B_max: the reference's +inf for an exactly singular Psi_A


def test_fused_estep_automatic_choice(engine):
    """Option "fused_estep" = 1 (automatic): the first E-step of a geometry (or after a K^n upload) runs the separate passes
    (no census yet), the following ones the fused kernel while K^n is sparse -- states above four latents in at most a
    quarter of the datapoints; a dense K^n goes back to the separate passes."""
    from evo_amd.models import SSSC
    from evo_amd.variational import init_states
    rng = np.random.RandomState(9)
    D, H, S, N = 16, 64, 24, 200
    Y = rng.normal(size=(N, D))
    my_data = {"y": Y, "x_infr": np.ones_like(Y, dtype=bool)}
    np.random.seed(4)
    engine.set_option("fused_estep", 1)
    try:
        _automatic_choice_body(engine, D, H, S, N, my_data)
    finally:
        engine.set_option("fused_estep", 0)


def _automatic_choice_body(engine, D, H, S, N, my_data):
    from evo_amd.models import SSSC
    from evo_amd.variational import init_states
    model = SSSC(D, H, S, rng="device", sync_host=False, engine=engine, seed=3, device_mstep=True)
    theta = model.check_params(model.standard_init(my_data))
    suff = init_states(N, S, H, "fit", "randflip", 6, 1, 1)
    used = []
    for _ in range(3):
        _, _, _, theta = model.step(theta, suff, my_data)
        used.append(model.last_estep_fused)
    assert used == [False, True, True], used
    ss = engine.download_states()
    for s in range(S):  # every state of datapoint 0 gets 12 active latents (at most 6 of them are replaced per step)
        ss[0, s] = False
        ss[0, s, s:s + 12] = True
    engine.upload_states(ss)  # (a K^n from the host: its census is unknown -> separate passes, which then count it)
    _, _, _, theta = model.step(theta, suff, my_data)
    assert model.last_estep_fused is False
    _, _, _, theta = model.step(theta, suff, my_data)  # 24 dense states in 200 datapoints: still sparse -> fused
    assert model.last_estep_fused is True
    ss = engine.download_states()
    ss[:, :, :6] = True  # every state of every datapoint above four latents: the automatic choice goes back to the separate passes
    engine.upload_states(ss)
    for _ in range(2):
        _, _, _, theta = model.step(theta, suff, my_data)
        assert model.last_estep_fused is False
