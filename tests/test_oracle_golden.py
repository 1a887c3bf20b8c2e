"""Pins the CPU oracle (oracle/evo_oracle.py) against fixtures generated FROM THE REFERENCE
(tests/golden/make_golden.py).  CPU only."""
import hashlib

import numpy as np
import pytest

from conftest import load_golden, unpack_bits
from oracle import evo_oracle as orc

STEP_FIXTURES = ["ebsc_bars", "es3c_bars", "ebsc_mid", "es3c_mid", "es3c_dense", "ebsc_dense",
                 "ebsc_sparseflip", "es3c_cross", "ebsc_gen2", "ebsc_perm", "es3c_perm", "es3c_f32",
                 "ebsc_bg", "es3c_bg", "es3c_bg_cross", "ebsc_exact", "es3c_exact_bg"]  # r4: background unit, exact E-steps
BSC_KEYS = ("W", "pi", "sigma")
SSSC_KEYS = ("W", "pies", "mus", "Psi", "sigma2")


def suff_from_fixture(g, ss_bool):
    N, S, H = ss_bool.shape
    bf = float(g["ea_bitflip_prob"])
    S_perm = int(g["S_perm"]) if "S_perm" in g else 0  # permanent all-zero state (variational/utils.py:39-54)
    return {
        "ss": ss_bool.copy(), "lpj": np.empty((N, S + S_perm)), "S_perm": S_perm, "incl": np.zeros((S_perm, H), dtype=bool),
        "permanent": {"background": bool(g["background"]) if "background" in g else False, "allzero": S_perm == 1,
                      "singletons": False}, "sm": None,
        "n_parents": int(g["ea_n_parents"]), "n_children": int(g["ea_n_children"]),
        "n_generations": int(g["ea_n_generations"]),
        "parent_selection": orc.PARENT_SELECTION[str(g["ea_parent_selection"])],
        "mutation_algorithm": orc.MUTATION[str(g["ea_mutation"])],
        "bitflip_prob": None if np.isnan(bf) else bf, "Mprime": int(g["ea_Mprime"]),
    }


def theta_in(g, t, keys):
    th = {k: np.array(g["t%d_in_%s" % (t, k)]) for k in keys}
    for k in ("pi", "sigma", "sigma2"):
        if k in th:
            th[k] = np.float64(th[k])
    return th


@pytest.mark.parametrize("name", STEP_FIXTURES)
def test_step_replay(name):
    g = load_golden("step_%s.npz" % name)
    algo = str(g["algo"])
    H, N, S = int(g["H"]), int(g["N"]), int(g["S"])
    Y = g["Y"]
    for t in range(int(g["n_steps"])):
        keys = BSC_KEYS if algo == "ebsc" else SSSC_KEYS
        theta = theta_in(g, t, keys)
        suff = suff_from_fixture(g, unpack_bits(g["t%d_ss_in" % t], H))
        trace = []
        np.random.seed(1000 + int(g["seed"]) + t)
        if algo == "ebsc":
            F, nu, nsub, theta, sums = orc.bsc_step(theta, suff, Y, trace=trace)
            sum_names = ("Wp", "Wq", "pies", "sigma", "Fs")
        else:
            prec = np.float32 if ("precision32" in g and bool(g["precision32"])) else np.float64  # sssc.py:49
            if prec is np.float32 and t > 0:
                # the reference's own Theta^new carries pies as a float32 ARRAY in this mode (sssc.py:714: float32 sums / N),
                # so its next precompute takes the logs in float32; the fixture stores values as float64 (lossless)
                theta["pies"] = theta["pies"].astype(np.float32)
            F, nu, nsub, theta, sums = orc.sssc_step(theta, suff, Y, use_storage=bool(g["use_storage"]), trace=trace,
                                                     precision=prec)
            sum_names = ("xpt_s", "xpt_ss", "xpt_sz", "xpt_szsz", "s_sz_outer", "sz_sz_outer", "Wp", "y_outer_diag", "Fs")
        # candidate stream: bit-exact states, lpj to rounding
        counts = np.array([c[1].shape[0] for c in trace])
        assert np.array_equal(counts, g["t%d_cand_counts" % t])
        if counts.sum():
            got = np.concatenate([c[1] for c in trace], axis=0)
            assert np.array_equal(np.packbits(got, axis=-1), g["t%d_cand_states" % t])
            np.testing.assert_allclose(np.concatenate([c[2] for c in trace]), g["t%d_cand_lpj" % t], rtol=1e-12)
        # selection: bit-exact
        assert np.array_equal(np.packbits(suff["ss"], axis=-1), g["t%d_ss_out" % t])
        np.testing.assert_allclose(suff["lpj"], g["t%d_lpj_out" % t], rtol=1e-12)
        assert nu == float(g["t%d_S_nunique" % t]) and nsub == float(g["t%d_S_sub" % t])
        np.testing.assert_allclose(F, float(g["t%d_F" % t]), rtol=1e-13)
        for nm in sum_names:
            np.testing.assert_allclose(sums[nm], g["t%d_sum_%s" % (t, nm)], rtol=1e-11, atol=1e-13, err_msg=nm)
        for k in keys:
            np.testing.assert_allclose(theta[k], g["t%d_out_%s" % (t, k)], rtol=1e-9, atol=1e-11, err_msg=k)


def _shape_problem(g, n_sub=None):
    """Inputs of a shape_*.npz fixture, regenerated from its seed through the oracle's own standard_init /
    init_states (the fixture holds hashes of what the reference drew)."""
    import _sketch
    algo = str(g["algo"])
    D, H, S, N, seed = int(g["D"]), int(g["H"]), int(g["S"]), int(g["N"]), int(g["seed"])
    np.random.seed(seed)
    Y = np.random.randn(N, D)
    assert _sketch.array_sha1(Y) == str(g["Y_sha1"])
    if algo == "ebsc":
        theta = orc.check_params(orc.bsc_standard_init(Y, H), orc.BSC_POLICY)
        keys = BSC_KEYS
    else:
        theta = orc.check_params(orc.sssc_standard_init(Y, H), orc.SSSC_POLICY)
        keys = SSSC_KEYS
    for k in keys:
        assert _sketch.array_sha1(np.asarray(theta[k], dtype=np.float64)) == str(g["in_sha1_" + k]), k
    suff = orc.init_states(N, S, H, str(g["ea_parent_selection"]), str(g["ea_mutation"]), int(g["ea_n_parents"]),
                           int(g["ea_n_children"]), int(g["ea_n_generations"]))
    assert np.array_equal(_sketch.state_hashes(suff["ss"]), g["ss_in_hash"])
    if n_sub is not None:  # datapoints are independent given Theta and consume np.random in order: a prefix replays alone
        Y = Y[:n_sub]
        suff["ss"], suff["lpj"] = suff["ss"][:n_sub].copy(), suff["lpj"][:n_sub].copy()
    return algo, keys, Y, theta, suff


def _check_shape_estep(g, t, suff, trace, n):
    import _sketch
    assert np.array_equal(_sketch.state_hashes(suff["ss"]), g["t%d_ss_hash" % t][:n]), "K^n differs"
    np.testing.assert_allclose(_sketch.lpj_rows(suff["lpj"]), g["t%d_lpj_rows" % t][:n], rtol=1e-11)
    assert np.array_equal([c[1].shape[0] for c in trace], g["t%d_cand_counts" % t][:n])
    assert np.array_equal(_sketch.ragged_hashes([c[1] for c in trace]), g["t%d_cand_hash" % t][:n])
    np.testing.assert_allclose([c[2].sum() for c in trace], g["t%d_cand_lpj_sum" % t][:n], rtol=1e-11, atol=1e-9)


@pytest.mark.parametrize("name", ["c2_small", "c3_small", "c4_small", "c5_small", "c3_wide"])
def test_shape_replay_small(name):
    """Reference EM steps at the TRUE (D, H, S) of BASELINE.json configs[1..4] with small N: the oracle must
    reproduce candidate streams, selection, lpj, F, every all-reduced accumulator and -- where the Theta
    update is well posed (condition number stored with the fixture) -- Theta^new.  A step whose Theta^in
    came out of an ill-posed update (N << H) is not replayed."""
    import _sketch
    from conftest import sketch_close
    g = load_golden("shape_%s.npz" % name)
    algo, keys, Y, theta, suff = _shape_problem(g)
    N, seed = int(g["N"]), int(g["seed"])
    for t in range(int(g["n_steps"])):
        trace = []
        np.random.seed(1000 + seed + t)
        if algo == "ebsc":
            F, nu, nsub, theta, sums = orc.bsc_step(theta, suff, Y, trace=trace)
            sum_names = ("Wp", "Wq", "pies", "sigma", "Fs")
        else:
            F, nu, nsub, theta, sums = orc.sssc_step(theta, suff, Y, use_storage=False, trace=trace)
            sum_names = ("xpt_s", "xpt_ss", "xpt_sz", "xpt_szsz", "s_sz_outer", "sz_sz_outer", "Wp", "y_outer_diag", "Fs")
        _check_shape_estep(g, t, suff, trace, N)
        np.testing.assert_allclose(F, float(g["t%d_F" % t]), rtol=1e-13)
        assert nu == float(g["t%d_S_nunique" % t]) and nsub == float(g["t%d_S_sub" % t])
        for nm in sum_names:
            sketch_close(_sketch.sketch(sums[nm]), g["t%d_sum_%s" % (t, nm)], 1e-10, nm)
        cond = float(g["t%d_cond" % t])
        if cond > 1e8:
            break  # Theta^new of this step is rounding noise times the condition number: nothing to chain on
        for k in keys:
            sketch_close(_sketch.sketch(theta[k]), g["t%d_out_%s" % (t, k)], max(1e-9, 1e-14 * cond), k)


@pytest.mark.parametrize("name,n_sub", [("c2", 12), ("c3", 48), ("c4", 4), ("c5", 10), ("c2x5", 10), ("c3x5", 40)])
def test_shape_replay_prefix(name, n_sub):
    """The N ~ 3H fixtures (minutes of reference time): the oracle replays the E-step of the first n_sub
    datapoints of step 0 (same Theta, same np.random prefix) -- candidate stream, selection and lpj."""
    g = load_golden("shape_%s.npz" % name)
    algo, keys, Y, theta, suff = _shape_problem(g, n_sub)
    trace = []
    np.random.seed(1000 + int(g["seed"]))
    if algo == "ebsc":
        orc.bsc_E_step(theta, suff, Y, trace)
    else:
        orc.sssc_EM_accumulate(theta, suff, Y, use_storage=False, trace=trace)
    _check_shape_estep(g, 0, suff, trace, n_sub)


def test_kat_bars_from_seed():
    """Whole pipeline (generate -> standard_init -> init_states -> 3 EM steps) from seed 42;
    numbers also quoted in BASELINE.md section 2."""
    g = load_golden("kat_bars.npz")
    H, D, N, S = 10, 25, 500, 32
    for algo in ("ebsc", "es3c"):
        np.random.seed(42)
        W = 10.0 * orc.bars_dictionary(H)
        if algo == "ebsc":
            Y, _ = orc.bsc_generate({"W": W, "pi": 2.0 / H, "sigma": 1.0}, N)
            theta = orc.check_params(orc.bsc_standard_init(Y, H), orc.BSC_POLICY)
        else:
            Y, _, _ = orc.sssc_generate({"W": W, "pies": np.ones(H) * 2.0 / H, "sigma2": np.array(1.0),
                                         "mus": np.zeros(H), "Psi": np.eye(H)}, N)
            theta = orc.check_params(orc.sssc_standard_init(Y, H), orc.SSSC_POLICY)
        assert hashlib.sha1(Y.tobytes()).hexdigest() == str(g[algo + "_Y_sha1"])
        suff = orc.init_states(N, S, H, "fit", "randflip", 10, 1, 1)
        Fs = []
        for _ in range(3):
            if algo == "ebsc":
                F, _, _, theta, _ = orc.bsc_step(theta, suff, Y)
            else:
                F, _, _, theta, _ = orc.sssc_step(theta, suff, Y)
            Fs.append(F)
        np.testing.assert_allclose(Fs, g[algo + "_F"], rtol=1e-12)
        assert hashlib.sha1(np.packbits(suff["ss"], axis=-1).tobytes()).hexdigest() == str(g[algo + "_ss_sha1"])
    np.testing.assert_allclose(g["ebsc_F"], [-78.7824265109, -76.6595629116, -74.1714405461], rtol=1e-11)
    np.testing.assert_allclose(g["es3c_F"], [-83.4044473038, -78.1001689500, -73.7140907703], rtol=1e-11)


def test_lpj_bsc():
    g = load_golden("lpj_bsc.npz")
    H = int(g["H"])
    theta = {"W": g["W"], "pi": float(g["pi"]), "sigma": float(g["sigma"])}
    cnt = orc.bsc_precompute(theta, g["Y"].shape[1], H)
    states = unpack_bits(g["states"], H)
    for n in range(g["Y"].shape[0]):
        np.testing.assert_allclose(orc.bsc_lpj(theta, states, g["Y"][n], cnt), g["lpj"][n], rtol=1e-14)
    np.testing.assert_allclose(theta["ljc"], float(g["ljc"]), rtol=1e-15)


def test_lpj_sssc():
    g = load_golden("lpj_sssc.npz")
    H = int(g["H"])
    theta = {k: g[k] for k in SSSC_KEYS}
    theta["sigma2"] = np.float64(theta["sigma2"])
    cnt = orc.sssc_precompute(theta, g["Y"].shape[1])
    states = unpack_bits(g["states"], H)
    for n in range(g["Y"].shape[0]):
        np.testing.assert_allclose(orc.sssc_lpj(theta, states, g["Y"][n], cnt, {}), g["lpj"][n], rtol=1e-13)
    np.testing.assert_allclose(theta["ljc"], float(g["ljc"]), rtol=1e-15)


@pytest.mark.parametrize("fixture,n_sing", [("lpj_sssc_singular.npz", 7), ("lpj_sssc_singular_k3.npz", 11),
                                            ("lpj_sssc_indefinite.npz", 15), ("lpj_sssc_dense.npz", 0)])
def test_lpj_sssc_singular_psi(fixture, n_sing):
    """Exactly singular Psi_s (sssc.py:278-301): pinv branches, lpj = +inf -> B_max with the isinf counter, and the
    lambda_s / kappa_s the reference's statistics loop reads from its storage -- all from the reference itself
    (states with at most two active latents; `_k3`: three to ten, and one M_s that is exactly singular as well).
    `_indefinite` (round 4): a REGULAR Psi_s whose M_s is exactly singular (sssc.py:295-300), inside an indefinite Psi;
    `_dense`: up to 150 active latents, nothing singular."""
    g = load_golden(fixture)
    H = int(g["H"])
    theta = {k: g[k] for k in SSSC_KEYS}
    theta["sigma2"] = np.float64(theta["sigma2"])
    states = unpack_bits(g["states"], H)
    Y = g["Y"]
    assert int((g["lpj"][0] == 0.0).sum()) == n_sing  # states that hold an exactly singular Psi_s
    for n in range(Y.shape[0]):
        cnt = orc.sssc_precompute(theta, Y.shape[1])
        cache = {}
        with np.errstate(all="ignore"):
            got = orc.sssc_lpj(theta, states, Y[n], cnt, cache)
        np.testing.assert_allclose(got, g["lpj"][n], rtol=1e-13, atol=0)
        assert [cnt["isnan"], cnt["smaller_eps"], cnt["isinf"]] == list(g["reset_counts"][n])
        for c in range(states.shape[0]):
            k = int(states[c].sum())
            if k == 0:
                continue
            t = cache[states[c].tobytes()]
            np.testing.assert_allclose(t["lam"], g["lam"][c, :k, :k], rtol=1e-12, atol=1e-14)
            kap = np.dot(t["lam_Wt"], Y[n] - t["Wmu"]) + theta["mus"][states[c]]
            np.testing.assert_allclose(kap, g["kappa"][n, c, :k], rtol=1e-11, atol=1e-13)


def test_lpj_clamp():
    g = load_golden("lpj_clamp.npz")
    for i in range(4):
        cnt = orc.new_counters()
        with np.errstate(all="ignore"):
            out = orc.lpj_clamp(g["in%d" % i].copy(), cnt)
        assert np.array_equal(out, g["out%d" % i])
        assert [cnt["isnan"], cnt["smaller_eps"], cnt["isinf"]] == list(g["cnt%d" % i])


def test_vary_kn():
    g = load_golden("vary_kn.npz")
    for i in range(int(g["n_cases"])):
        H, S, Mp = int(g["c%d_H" % i]), int(g["c%d_S" % i]), int(g["c%d_Mprime" % i])
        states = unpack_bits(g["c%d_old" % i], H).copy()
        new = unpack_bits(g["c%d_new" % i], H).reshape(-1, H)
        lpj_out = np.zeros(S)
        ret = orc.vary_Kn(g["c%d_lpj_old" % i].copy(), g["c%d_lpj_new" % i].copy(), lpj_out, states, new, H, S, 0,
                          np.zeros((0, H), dtype=bool), Mp)
        assert list(ret) == list(g["c%d_ret" % i])
        assert np.array_equal(np.packbits(states, axis=-1), g["c%d_states_out" % i])
        assert np.array_equal(lpj_out, g["c%d_lpj_out" % i])
    # SURVEY 8c(i) known answers, stated independently of the fixture file
    assert list(g["c0_ret"]) == [3, 2] and list(g["c0_lpj_out"]) == [-2.0, -1.0, -0.5, -3.0]
    assert list(g["c1_ret"]) == [3, 1] and list(g["c1_lpj_out"]) == [-5.0, -1.0, -0.5, -3.0]
    assert list(g["c2_ret"]) == [0, 0]


def test_full_free_energy():
    g = load_golden("full_F.npz")
    H, S = 8, 10
    suff = {"sm": orc.all_states_matrix(H)}
    th = {k: g["ebsc_" + k] for k in BSC_KEYS}
    th["pi"], th["sigma"] = float(th["pi"]), float(th["sigma"])
    np.testing.assert_allclose(orc.bsc_free_energy_full(th, suff, g["ebsc_Y"]), float(g["ebsc_L"]), rtol=1e-13)
    th = {k: g["es3c_" + k] for k in SSSC_KEYS}
    th["sigma2"] = np.float64(th["sigma2"])
    np.testing.assert_allclose(orc.sssc_free_energy_full(th, suff, g["es3c_Y"]), float(g["es3c_L"]), rtol=1e-13)


@pytest.mark.parametrize("algo", ["ebsc", "es3c"])
def test_reconstruction_replay(algo):
    """step(..., do_reconstruction=True) on complete data (_models.py:193-194,614-665; sssc.py:500-507,
    613-627): the oracle's y_reconstructed against the reference's, two chained steps."""
    g = load_golden("recon_%s.npz" % algo)
    H = int(g["H"])
    Y, x = g["Y"], g["x"]
    keys = BSC_KEYS if algo == "ebsc" else SSSC_KEYS
    theta = theta_in(g, 0, keys)
    suff = suff_from_fixture(g, unpack_bits(g["t0_ss_in"], H))
    for t in range(int(g["n_steps"])):
        np.random.seed(1000 + int(g["seed"]) + t)
        if algo == "ebsc":
            F, nu, nsub, theta, sums = orc.bsc_step(theta, suff, Y, reconstruct_x=x)
        else:
            F, nu, nsub, theta, sums = orc.sssc_step(theta, suff, Y, reconstruct_x=x)
        np.testing.assert_allclose(F, float(g["t%d_F" % t]), rtol=1e-13)
        assert np.array_equal(np.packbits(suff["ss"], axis=-1), g["t%d_ss_out" % t])
        want = g["t%d_y_reconstructed" % t]
        np.testing.assert_allclose(sums["y_reconstructed"], want, rtol=1e-11, atol=1e-12)
        assert np.array_equal(sums["y_reconstructed"][x], Y[x])          # kept entries untouched
        assert np.array_equal(sums["y_reconstructed"][1], Y[1])          # x[1] all True


def test_missing_data_replay_ebsc():
    """EBSC on incomplete data (x_infr not all True, NaN at the missing entries): masked lpj
    (bsc.py:59-97), ljc over the reliable entries (bsc.py:113-118), M-step on y_reconstructed with the
    masked residual and the incomplete-data sigma (bsc.py:184-223,266-272), reconstruction
    (_models.py:614-665) -- three chained steps, the last one without a fresh reconstruction."""
    g = load_golden("missing_ebsc.npz")
    H = int(g["H"])
    Y, x_infr = g["Y"], g["x_infr"]
    assert np.isnan(Y[~x_infr]).all() and not np.isnan(Y[x_infr]).any()
    theta = theta_in(g, 0, BSC_KEYS)
    suff = suff_from_fixture(g, unpack_bits(g["t0_ss_in"], H))
    y_rec = None
    for t in range(int(g["n_steps"])):
        np.random.seed(1000 + int(g["seed"]) + t)
        do_rec = bool(g["t%d_do_rec" % t])
        F, nu, nsub, theta, sums = orc.bsc_step(theta, suff, Y, x_infr=x_infr,
                                                reconstruct_x=x_infr if do_rec else None, y_rec_prev=y_rec)
        if do_rec:
            y_rec = sums["y_reconstructed"]
        np.testing.assert_allclose(F, float(g["t%d_F" % t]), rtol=1e-13)
        assert np.array_equal(np.packbits(suff["ss"], axis=-1), g["t%d_ss_out" % t])
        np.testing.assert_allclose(suff["lpj"], g["t%d_lpj_out" % t], rtol=1e-12)
        np.testing.assert_allclose(y_rec, g["t%d_y_reconstructed" % t], rtol=1e-11, atol=1e-12)
        for k in BSC_KEYS:
            np.testing.assert_allclose(theta[k], g["t%d_out_%s" % (t, k)], rtol=1e-9, atol=1e-11, err_msg=k)


def test_standard_init_incomplete_data():
    """_models.py:246-267 on the fixture's data: same RNG draw, same Theta^init as the reference."""
    g = load_golden("missing_ebsc.npz")
    np.random.seed(int(g["seed"]))
    D, H, S, N = int(g["D"]), int(g["H"]), int(g["S"]), int(g["N"])
    # replay the RNG consumption that precedes standard_init in the generator (data + mask draws)
    W = 10.0 * orc.bars_dictionary(H)
    orc.bsc_generate({"W": W, "pi": 2.0 / H, "sigma": 1.0}, N)
    np.random.random_sample((N, D))
    th = orc.check_params(orc.bsc_standard_init(g["Y"], H, g["x_infr"]), orc.BSC_POLICY)
    for k in BSC_KEYS:
        np.testing.assert_allclose(th[k], g["t0_in_%s" % k], rtol=1e-12, atol=1e-13, err_msg=k)


def test_missing_data_replay_es3c():
    """ES3C on incomplete data: per-datapoint W_obs in the state terms (sssc.py:276-318), ljc over the
    reliable entries (:352-357), Wp from the reconstructed rows (:631), the incomplete-data sigma2
    (:747-755), reconstruction (:613-627) -- two chained steps against the reference."""
    g = load_golden("missing_es3c.npz")
    H = int(g["H"])
    Y, x_infr = g["Y"], g["x_infr"]
    theta = theta_in(g, 0, SSSC_KEYS)
    suff = suff_from_fixture(g, unpack_bits(g["t0_ss_in"], H))
    for t in range(int(g["n_steps"])):
        np.random.seed(1000 + int(g["seed"]) + t)
        F, nu, nsub, theta, acc = orc.sssc_step(theta, suff, Y, use_storage=False, reconstruct_x=x_infr, x_infr=x_infr)
        np.testing.assert_allclose(F, float(g["t%d_F" % t]), rtol=1e-12)
        assert np.array_equal(np.packbits(suff["ss"], axis=-1), g["t%d_ss_out" % t])
        np.testing.assert_allclose(suff["lpj"], g["t%d_lpj_out" % t], rtol=1e-11)
        np.testing.assert_allclose(acc["y_reconstructed"], g["t%d_y_reconstructed" % t], rtol=1e-10, atol=1e-11)
        for k in SSSC_KEYS:
            np.testing.assert_allclose(theta[k], g["t%d_out_%s" % (t, k)], rtol=1e-8, atol=1e-10, err_msg=k)


def test_init_states_background_and_exact():
    """Round 4: init_states with the permanent background unit and with exact E-steps (S == 2^H_) against the
    reference's own outputs -- K^n, shapes, the state table and where np.random stands afterwards
    (variational/utils.py:42-47, 55, 71-98, 140-141)."""
    g = load_golden("background.npz")
    for nm in ("bg", "exact", "exact_bg", "exact_zero", "bg_zero_ignored"):
        N, S, H = int(g[nm + "_N"]), int(g[nm + "_S"]), int(g[nm + "_H"])
        p0 = float(g[nm + "_p0"])
        perm = dict(zip(("background", "allzero", "singletons"), (bool(v) for v in g[nm + "_perm"])))
        np.random.seed(31)
        suff = orc.init_states(N, S, H, "fit", "randflip", 3, 2, 1, None, None, None if np.isnan(p0) else p0, perm)
        assert np.array_equal(suff["ss"], g[nm + "_ss"]), nm
        assert list(suff["lpj"].shape) == list(g[nm + "_lpj_shape"]), nm
        assert suff["S_perm"] == int(g[nm + "_S_perm"]) and list(suff["incl"].shape) == list(g[nm + "_incl_shape"]), nm
        assert np.array_equal(suff["sm"], g[nm + "_sm"]), nm
        assert np.random.random() == float(g[nm + "_next_random"]), nm
        if perm["background"]:
            assert suff["ss"][:, :, -1].all()


@pytest.mark.parametrize("algo", ["ebsc", "es3c"])
def test_full_free_energy_background(algo):
    g = load_golden("background.npz")
    keys = BSC_KEYS if algo == "ebsc" else SSSC_KEYS
    theta = {k: np.array(g["full_%s_%s" % (algo, k)]) for k in keys}
    for k in ("pi", "sigma", "sigma2"):
        if k in theta:
            theta[k] = np.float64(theta[k])
    Y = g["full_%s_Y" % algo]
    H = theta["W"].shape[1]
    suff = {"sm": orc.all_states_matrix(H - 1), "permanent": {"background": True, "allzero": False, "singletons": False}}
    if algo == "ebsc":
        theta = orc.check_params(theta, orc.BSC_POLICY)
        L = orc.bsc_free_energy_full(theta, suff, Y)
    else:
        theta = orc.check_params(theta, orc.SSSC_POLICY)
        L = orc.sssc_free_energy_full(theta, suff, Y)
    np.testing.assert_allclose(L, float(g["full_%s_L" % algo]), rtol=1e-12)
