"""GPU parity of the host-side mirror (evo_amd.models / evo_amd.variational) against fixtures
generated from the reference: whole EM trajectories in rng="reference" mode, written the way a
reference user would write them (model.step(theta, my_suff_stat, my_data))."""
import hashlib

import numpy as np
import pytest

from conftest import load_golden, unpack_bits

pytestmark = pytest.mark.gpu

BSC_KEYS = ("W", "pi", "sigma")
SSSC_KEYS = ("W", "pies", "mus", "Psi", "sigma2")
STEP_FIXTURES = ["ebsc_bars", "es3c_bars", "ebsc_mid", "es3c_mid", "es3c_dense", "ebsc_dense",
                 "ebsc_sparseflip", "es3c_cross", "ebsc_gen2"]


@pytest.fixture(scope="module")
def engine():
    from evo_amd.engine import Engine
    eng = Engine()
    yield eng
    eng.close()


def make_suff(g, ss):
    from evo_amd.variational.utils import MUTATION, PARENT_SELECTION
    N, S, H = ss.shape
    bf = float(g["ea_bitflip_prob"])
    return {
        "ss": ss.copy(), "lpj": np.empty((N, S)), "S_perm": 0, "incl": np.zeros((0, H), dtype=bool),
        "permanent": {"background": False, "allzero": False, "singletons": False}, "sm": None,
        "n_parents": int(g["ea_n_parents"]), "n_children": int(g["ea_n_children"]),
        "n_generations": int(g["ea_n_generations"]),
        "parent_selection": PARENT_SELECTION[str(g["ea_parent_selection"])],
        "mutation_algorithm": MUTATION[str(g["ea_mutation"])],
        "bitflip_prob": None if np.isnan(bf) else bf, "Mprime": int(g["ea_Mprime"]),
    }


@pytest.mark.parametrize("name", ["ebsc_bars", "ebsc_mid", "ebsc_dense"])
def test_trajectory_bsc_direct_form(engine, name):
    """Same trajectories with the cancellation-free direct EBSC kernel selected."""
    try:
        engine.set_option("bsc_direct", 1)
        test_trajectory_reference_rng(engine, name)
    finally:
        engine.set_option("bsc_direct", 0)


@pytest.mark.parametrize("name", STEP_FIXTURES)
def test_trajectory_reference_rng(engine, name):
    """Theta and K^n are carried from step to step (not reloaded), so errors would compound:
    after every step K^n must be bit-identical and F / Theta within 1e-8 of the reference."""
    from evo_amd.models import BSC, SSSC
    g = load_golden("step_%s.npz" % name)
    algo = str(g["algo"])
    D, H, S, N = int(g["D"]), int(g["H"]), int(g["S"]), int(g["N"])
    keys = BSC_KEYS if algo == "ebsc" else SSSC_KEYS
    model = (BSC(D, H, S, engine=engine) if algo == "ebsc"
             else SSSC(D, H, S, use_storage=bool(g["use_storage"]), engine=engine))
    Y = g["Y"]
    my_data = {"y": Y, "x_infr": np.ones_like(Y, dtype=bool)}
    theta = {k: np.array(g["t0_in_%s" % k]) for k in keys}
    for k in ("pi", "sigma", "sigma2"):
        if k in theta:
            theta[k] = np.float64(theta[k])
    suff = make_suff(g, unpack_bits(g["t0_ss_in"], H))
    for t in range(int(g["n_steps"])):
        np.random.seed(1000 + int(g["seed"]) + t)
        F, nu, nsub, theta = model.step(theta, suff, my_data)
        assert np.array_equal(np.packbits(suff["ss"], axis=-1), g["t%d_ss_out" % t]), "K^n differs at step %d" % t
        np.testing.assert_allclose(suff["lpj"], g["t%d_lpj_out" % t], rtol=1e-8, atol=1e-8)
        np.testing.assert_allclose(F, float(g["t%d_F" % t]), rtol=1e-9)
        assert nu == float(g["t%d_S_nunique" % t]) and nsub == float(g["t%d_S_sub" % t])
        for k in keys:
            ref = g["t%d_out_%s" % (t, k)]
            np.testing.assert_allclose(theta[k], ref, rtol=1e-6, atol=1e-7 * max(1.0, np.abs(ref).max()), err_msg=k)


def test_kat_bars_from_seed(engine):
    """examples/bars-test set-up from seed 42: data generation, standard_init,
    init_states and three EM steps through evo_amd only; F must match the reference's numbers."""
    from evo_amd.models import BSC, SSSC
    from evo_amd.variational import init_states
    g = load_golden("kat_bars.npz")
    H, D, N, S = 10, 25, 500, 32
    R = H // 2
    W = np.zeros((R, R, H))
    for i in range(R):
        W[i, :, i] = 1.0
        W[:, i, R + i] = 1.0
    W = 10.0 * W.reshape(D, H)
    for algo in ("ebsc", "es3c"):
        np.random.seed(42)
        if algo == "ebsc":
            model = BSC(D, H, S, engine=engine)
            gen = {"W": W, "pi": 2.0 / H, "sigma": 1.0}
        else:
            model = SSSC(D, H, S, engine=engine)
            gen = {"W": W, "pies": np.ones(H) * 2.0 / H, "sigma2": np.array(1.0), "mus": np.zeros(H), "Psi": np.eye(H)}
        Y = model.generate_data(gen, N)["y"]
        assert hashlib.sha1(Y.tobytes()).hexdigest() == str(g[algo + "_Y_sha1"])
        my_data = {"y": Y, "x_infr": np.ones_like(Y, dtype=bool)}
        theta = model.check_params(model.standard_init(my_data))
        suff = init_states(N, S, H, "fit", "randflip", 10, 1, 1)
        Fs = []
        for _ in range(3):
            F, _, _, theta = model.step(theta, suff, my_data)
            Fs.append(F)
        np.testing.assert_allclose(Fs, g[algo + "_F"], rtol=1e-9)
        assert hashlib.sha1(np.packbits(suff["ss"], axis=-1).tobytes()).hexdigest() == str(g[algo + "_ss_sha1"])


def test_full_free_energy(engine):
    """free_energy(full=True): exact likelihood over all 2^H states (examples/bars-test/main.py:126)."""
    from evo_amd.models import BSC, SSSC
    from evo_amd.variational import init_states
    g = load_golden("full_F.npz")
    H, D, N, S = 8, 16, 30, 10
    for algo, cls, keys in (("ebsc", BSC, BSC_KEYS), ("es3c", SSSC, SSSC_KEYS)):
        Y = g[algo + "_Y"]
        theta = {k: np.array(g["%s_%s" % (algo, k)]) for k in keys}
        for k in ("pi", "sigma", "sigma2"):
            if k in theta:
                theta[k] = np.float64(theta[k])
        np.random.seed(0)
        suff = init_states(N, S, H, "fit", "randflip", 5, 1, 1)
        model = cls(D, H, S, engine=engine)
        L = model.free_energy({"y": Y, "x_infr": np.ones_like(Y, dtype=bool)}, theta, suff, full=True)
        np.testing.assert_allclose(L, float(g[algo + "_L"]), rtol=1e-11)
        assert suff["S_perm"] == 0 and not suff["permanent"]["allzero"]


def test_per_datapoint_operator(engine):
    """log_pseudo_joint with the reference's scratch-key protocol."""
    from evo_amd.models import SSSC
    g = load_golden("lpj_sssc.npz")
    H = int(g["H"])
    Y = g["Y"]
    model = SSSC(Y.shape[1], H, 8, engine=engine)
    theta = {k: g[k] for k in SSSC_KEYS}
    theta["sigma2"] = np.float64(theta["sigma2"])
    my_data = {"y": Y, "x_infr": np.ones_like(Y, dtype=bool)}
    suff = {}
    model.E_step_precompute(theta, suff, my_data)
    np.testing.assert_allclose(theta["ljc"], float(g["ljc"]), rtol=1e-15)
    states = unpack_bits(g["states"], H)
    for n in range(Y.shape[0]):
        my_data["this_y"] = Y[n]
        my_data["this_x_infr"] = my_data["x_infr"][n]
        suff["this_states"] = states
        np.testing.assert_allclose(model.log_pseudo_joint(theta, suff, my_data), g["lpj"][n], rtol=1e-10)


@pytest.mark.parametrize("algo", ["ebsc", "es3c"])
def test_device_rng_mode(engine, algo):
    """rng='device': candidates generated on the GPU.  No stream parity is claimed; check the
    invariants vary_Kn guarantees (variational/utils.py:318): every K^n stays duplicate-free, lpj
    rows match a re-evaluation of the states, and for FIXED Theta the free energy never decreases."""
    from evo_amd.models import BSC, SSSC
    from evo_amd.variational import init_states
    rng = np.random.RandomState(5)
    D, H, S, N = 32, 70, 24, 300
    Y = rng.normal(size=(N, D))
    my_data = {"y": Y, "x_infr": np.ones_like(Y, dtype=bool)}
    np.random.seed(3)
    cls = BSC if algo == "ebsc" else SSSC
    model = cls(D, H, S, to_learn=[], rng="device", sync_host=True, engine=engine, seed=11)
    theta = model.check_params(model.standard_init(my_data))
    suff = init_states(N, S, H, "fit", "randflip", 8, 2, 1)
    Fs = []
    for _ in range(6):
        F, nu, nsub, theta = model.step(theta, suff, my_data)
        Fs.append(F)
        assert 0 <= nsub <= nu <= 16
    assert all(b >= a - 1e-12 for a, b in zip(Fs, Fs[1:])), Fs
    assert Fs[-1] > Fs[0]
    for n in range(N):
        assert np.unique(np.packbits(suff["ss"][n], axis=-1), axis=0).shape[0] == S
    lpj_after = suff["lpj"].copy()
    F2 = model.free_energy(my_data, theta, suff, full=False)
    np.testing.assert_allclose(F2, Fs[-1], rtol=1e-12)
    engine.lpj_resident()
    np.testing.assert_allclose(engine.download_lpj(), lpj_after, rtol=1e-12)


@pytest.mark.parametrize("algo,H,S", [("ebsc", 70, 24), ("es3c", 70, 24), ("ebsc", 200, 40), ("es3c", 136, 30)])
def test_state_digest_matches_word_path(engine, algo, H, S):
    """The lpj / statistics kernels read one 8-byte digest per state (k and the first active
    latents) that pack, evolve and vary_Kn keep next to the bit words.  Same run with
    state_digest = 0 (kernels extract from the words).  Fixed Theta: K^n and lpj bit-identical over
    8 device-RNG steps (init_states starts at k = 1..2, the mutations grow states past the digest's
    four slots).  Learning: the statistics scatter uses atomics, whose order moves Theta by an ulp
    and with it exact lpj ties in vary_Kn (also between two runs of the SAME setting), so there the
    first M-step's Theta and every F are compared to 1e-10 / 1e-9."""
    from evo_amd.models import BSC, SSSC
    from evo_amd.variational import init_states
    rng = np.random.RandomState(17)
    D, N = 40, 500
    W0 = rng.normal(size=(D, H))
    Y = (rng.random_sample((N, H)) < 3.0 / H).astype(float) @ W0.T + 0.3 * rng.normal(size=(N, D))
    my_data = {"y": Y, "x_infr": np.ones_like(Y, dtype=bool)}
    cls = BSC if algo == "ebsc" else SSSC

    def run(use, learn):
        engine.set_option("state_digest", use)
        try:
            np.random.seed(3)
            kw = {} if learn else {"to_learn": []}
            model = cls(D, H, S, rng="device", sync_host=True, engine=engine, seed=23, **kw)
            theta = model.check_params(model.standard_init(my_data))
            suff = init_states(N, S, H, "fit", "randflip", 6, 3, 1)
            # every third state dense (k up to ~12 > the digest's four slots): word fall-back, and
            # children of dense parents exercise it for the candidates as well
            suff["ss"][:, ::3, :] = np.random.RandomState(5).random_sample((N, len(range(0, S, 3)), H)) < 6.0 / H
            Fs, theta1 = [], None
            for it in range(8):
                F, _, _, theta = model.step(theta, suff, my_data)
                Fs.append(F)
                if it == 0:
                    theta1 = {k: np.array(v) for k, v in theta.items()}
            return np.array(Fs), suff["ss"].copy(), suff["lpj"].copy(), theta1
        finally:
            engine.set_option("state_digest", 1)

    F1, ss1, l1, _ = run(1, False)
    F0, ss0, l0, _ = run(0, False)
    np.testing.assert_array_equal(ss1, ss0)
    np.testing.assert_array_equal(l1, l0)
    np.testing.assert_array_equal(F1, F0)
    F1, _, _, t1 = run(1, True)
    F0, _, _, t0 = run(0, True)
    np.testing.assert_allclose(F1, F0, rtol=1e-9)
    for k in t1:
        np.testing.assert_allclose(t1[k], t0[k], rtol=1e-10, atol=1e-12)


@pytest.mark.parametrize("algo", ["ebsc", "es3c"])
def test_prefetched_lpj_pass(engine, algo):
    """evoamd_mstep_device enqueues the next iteration's pass over the resident K^n behind its mailbox
    kernel (into a second lpj buffer).  Same trajectory with prefetch_lpj = 0; the lpj rows a user reads
    after step() are still those of the E-step that just ended (the reference's semantics), and a host
    write to K^n or Theta between two steps drops the prefetched pass."""
    from evo_amd.models import BSC, SSSC
    from evo_amd.variational import init_states
    rng = np.random.RandomState(4)
    D, H, S, N = 24, 40, 16, 300
    Y = rng.normal(size=(N, D))
    my_data = {"y": Y, "x_infr": np.ones_like(Y, dtype=bool)}
    cls = BSC if algo == "ebsc" else SSSC
    out = []
    for pf in (1, 0):
        engine.set_option("prefetch_lpj", pf)
        try:
            np.random.seed(3)
            model = cls(D, H, S, rng="device", sync_host=False, engine=engine, seed=9)  # K^n stays resident
            theta = model.check_params(model.standard_init(my_data))
            suff = init_states(N, S, H, "fit", "randflip", 5, 2, 1)
            rec = []
            for it in range(5):
                if it == 3:  # host edits K^n between two steps: datapoint 0 gets fresh random states
                    suff["ss"][0] = np.random.RandomState(99).random_sample((S, H)) < 0.1
                    suff["ss"][0, np.arange(S), np.arange(S)] = True  # distinct rows
                    engine.upload_states(suff["ss"])
                F, _, _, theta = model.step(theta, suff, my_data)
                model.sync_to_host(suff)  # downloads only: the prefetched pass stays valid
                rec.append((F, suff["lpj"].copy(), suff["ss"].copy()))
            out.append(rec)
        finally:
            engine.set_option("prefetch_lpj", 1)
    for (F1, l1, s1), (F0, l0, s0) in zip(*out):
        np.testing.assert_array_equal(s1, s0)
        np.testing.assert_allclose(F1, F0, rtol=1e-10)
        np.testing.assert_allclose(l1, l0, rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("algo", ["ebsc", "es3c"])
def test_overlap_gemm_option(engine, algo):
    """evoamd_mstep_device forks the K = N statistics contraction onto a second stream beside the
    H x H inverses (default).  The serial schedule (overlap_gemm = 0) must give the same Theta and F:
    a missing join would show up as a stale or half-written sum_n y <s z>^T block."""
    from evo_amd.models import BSC, SSSC
    from evo_amd.variational import init_states
    rng = np.random.RandomState(2)
    D, H, S, N = 48, 96, 20, 4000
    Y = rng.normal(size=(N, D))
    my_data = {"y": Y, "x_infr": np.ones_like(Y, dtype=bool)}
    cls = BSC if algo == "ebsc" else SSSC
    out = []
    for ov in (2, 0):  # 2 = always (1 = only for the shapes where it was measured to pay)
        engine.set_option("overlap_gemm", ov)
        try:
            np.random.seed(3)
            model = cls(D, H, S, rng="device", sync_host=True, engine=engine, seed=5)
            theta = model.check_params(model.standard_init(my_data))
            suff = init_states(N, S, H, "fit", "randflip", 6, 2, 1)
            Fs = []
            for _ in range(3):
                F, _, _, theta = model.step(theta, suff, my_data)
                Fs.append(F)
            out.append((np.array(Fs), {k: np.array(v) for k, v in theta.items()}))
        finally:
            engine.set_option("overlap_gemm", 1)
    np.testing.assert_allclose(out[0][0], out[1][0], rtol=1e-9)
    for k in out[0][1]:
        np.testing.assert_allclose(out[0][1][k], out[1][1][k], rtol=1e-7, atol=1e-10)


@pytest.mark.parametrize("name", ["ebsc_mid", "es3c_mid", "es3c_bars", "ebsc_dense"])
def test_device_mstep_matches_host(engine, name):
    """device_mstep=True: the Theta update, clamps and precompute run on the GPU (Gauss-Jordan solves
    instead of LAPACK).  Same inputs as the host path => Theta within 1e-8, F and K^n identical
    for this step (the E-step is shared), and the reference's Theta within 1e-6."""
    from evo_amd.models import BSC, SSSC
    g = load_golden("step_%s.npz" % name)
    algo = str(g["algo"])
    D, H, S, N = int(g["D"]), int(g["H"]), int(g["S"]), int(g["N"])
    keys = BSC_KEYS if algo == "ebsc" else SSSC_KEYS
    Y = g["Y"]
    my_data = {"y": Y, "x_infr": np.ones_like(Y, dtype=bool)}
    out = {}
    for mode in (False, True):
        model = (BSC if algo == "ebsc" else SSSC)(D, H, S, engine=engine, device_mstep=mode)
        theta = {k: np.array(g["t0_in_%s" % k]) for k in keys}
        for k in ("pi", "sigma", "sigma2"):
            if k in theta:
                theta[k] = np.float64(theta[k])
        suff = make_suff(g, unpack_bits(g["t0_ss_in"], H))
        res = []
        for t in range(int(g["n_steps"])):
            np.random.seed(1000 + int(g["seed"]) + t)
            F, nu, nsub, theta = model.step(theta, suff, my_data)
            res.append((F, nu, nsub, {k: np.array(theta[k]) for k in keys}, suff["ss"].copy()))
        out[mode] = res
    for t, (h, d) in enumerate(zip(out[False], out[True])):
        np.testing.assert_allclose(d[0], h[0], rtol=1e-10, err_msg="F step %d" % t)
        assert d[1:3] == h[1:3]
        assert np.array_equal(d[4], h[4]), "K^n differs at step %d" % t
        for k in keys:
            ref = g["t%d_out_%s" % (t, k)]
            scale = max(1.0, float(np.abs(ref).max()))
            np.testing.assert_allclose(d[3][k], h[3][k], rtol=1e-7, atol=1e-9 * scale, err_msg="%s vs host" % k)
            np.testing.assert_allclose(d[3][k], ref, rtol=1e-6, atol=1e-7 * scale, err_msg="%s vs reference" % k)


def test_bench_path_with_rccl_communicator():
    """The exact configuration bench.py runs per rank (rng='device', device M-step, K^n resident,
    RcclComm) with a 1-rank RCCL communicator: same F trajectory as without a communicator."""
    from evo_amd.engine import Engine
    from evo_amd.models import SSSC
    from evo_amd.utils import parallel
    from evo_amd.variational import init_states
    rng = np.random.RandomState(9)
    D, H, S, N = 24, 40, 16, 200
    Y = rng.normal(size=(N, D))
    res = []
    for use_comm, overlap in ((False, 1), (True, 1), (True, 2)):
        # overlap = 2: the statistics contraction runs beside the inverses and the packed accumulator is
        # all-reduced in two pieces (scattered moments before the inverses, the GEMM block at the join)
        eng = Engine()
        eng.set_option("overlap_gemm", overlap)
        try:
            comm = parallel.RcclComm(eng, 0, 1, Engine.comm_unique_id()) if use_comm else None
            np.random.seed(4)
            model = SSSC(D, H, S, comm=comm, rng="device", sync_host=False, engine=eng, seed=5, device_mstep=True)
            my_data = {"y": Y, "x_infr": np.ones_like(Y, dtype=bool)}
            theta = model.check_params(model.standard_init(my_data))
            suff = init_states(N, S, H, "fit", "randflip", 6, 1, 1)
            Fs = []
            for _ in range(4):
                F, nu, nsub, theta = model.step(theta, suff, my_data)
                Fs.append(F)
            res.append(Fs)
            assert np.isfinite(Fs).all() and Fs[-1] > Fs[0]
            model.sync_to_host(suff)
            for n in range(0, N, 37):
                assert np.unique(np.packbits(suff["ss"][n], axis=-1), axis=0).shape[0] == S
            if comm is not None:
                comm.close()
        finally:
            eng.close()
    np.testing.assert_allclose(res[1], res[0], rtol=1e-9)
    np.testing.assert_allclose(res[2], res[0], rtol=1e-9)


@pytest.mark.parametrize("D,H,S,N,device_mstep", [(64, 64, 20, 300, False), (64, 64, 20, 300, True),
                                                    (192, 128, 24, 160, True)])
def test_es3c_tile_aligned_shapes_against_oracle(engine, D, H, S, N, device_mstep):
    """ES3C EM steps at shapes whose accumulator blocks are 64-aligned (D + H a multiple of 64), which
    switches the Ez^T Ez contraction to its upper-tiles-plus-mirror form and the GEMM loaders to their
    16-byte form; the golden fixtures have ragged shapes and never reach those paths.  Oracle =
    oracle.evo_oracle.sssc_step (sssc.py:407-813) on the same seeds."""
    from oracle import evo_oracle as orc
    from evo_amd.models import SSSC
    from evo_amd.variational import init_states
    rng = np.random.RandomState(D + H)
    gen = {"W": rng.normal(size=(D, H)), "pies": np.full(H, 2.0 / H), "mus": rng.normal(size=H) + 2.0,
           "Psi": np.eye(H), "sigma2": np.float64(0.5)}
    np.random.seed(5)
    Y = orc.sssc_generate(gen, N)[0]
    my_data = {"y": Y, "x_infr": np.ones_like(Y, dtype=bool)}
    np.random.seed(6)
    theta0 = orc.sssc_standard_init(Y, H)
    np.random.seed(7)
    suff_a = init_states(N, S, H, "fit", "randflip", 6, 2, 1)
    np.random.seed(7)
    suff_o = orc.init_states(N, S, H, "fit", "randflip", 6, 2, 1)
    assert np.array_equal(suff_a["ss"], suff_o["ss"])
    model = SSSC(D, H, S, engine=engine, device_mstep=device_mstep)
    th_a = {k: np.array(v) for k, v in theta0.items()}
    th_o = {k: np.array(v) for k, v in theta0.items()}
    for k in ("sigma2",):
        th_a[k], th_o[k] = np.float64(th_a[k]), np.float64(th_o[k])
    for t in range(2):
        np.random.seed(100 + t)
        Fa, nua, nsa, th_a = model.step(th_a, suff_a, my_data)
        np.random.seed(100 + t)
        Fo, nuo, nso, th_o, _ = orc.sssc_step(th_o, suff_o, Y)
        np.testing.assert_allclose(Fa, Fo, rtol=1e-9, err_msg="F step %d" % t)
        assert (nua, nsa) == (nuo, nso)
        assert np.array_equal(suff_a["ss"], suff_o["ss"]), "K^n differs at step %d" % t
        for k in ("W", "pies", "mus", "Psi", "sigma2"):
            ref = np.asarray(th_o[k])
            np.testing.assert_allclose(th_a[k], ref, rtol=1e-6, atol=1e-7 * max(1.0, float(np.abs(ref).max())),
                                       err_msg="%s step %d" % (k, t))


@pytest.mark.parametrize("algo", ["ebsc", "es3c"])
@pytest.mark.parametrize("device_mstep", [False, True])
def test_reconstruction_against_reference(engine, algo, device_mstep):
    """model.step(..., do_reconstruction=True) on complete data -- the image-denoising use
    (examples/image-denoising/main.py:100-110,162-169) -- against my_data["y_reconstructed"] recorded
    from the reference (tests/golden/recon_*.npz), two chained steps, host and device M-step."""
    from evo_amd.models import BSC, SSSC
    g = load_golden("recon_%s.npz" % algo)
    D, H, S = int(g["D"]), int(g["H"]), int(g["S"])
    keys = BSC_KEYS if algo == "ebsc" else SSSC_KEYS
    Y, x = g["Y"], g["x"]
    my_data = {"y": Y, "x_infr": np.ones_like(Y, dtype=bool), "x": x}
    model = (BSC if algo == "ebsc" else SSSC)(D, H, S, engine=engine, device_mstep=device_mstep)
    theta = {k: np.array(g["t0_in_%s" % k]) for k in keys}
    for k in ("pi", "sigma", "sigma2"):
        if k in theta:
            theta[k] = np.float64(theta[k])
    suff = make_suff(g, unpack_bits(g["t0_ss_in"], H))
    for t in range(int(g["n_steps"])):
        np.random.seed(1000 + int(g["seed"]) + t)
        F, nu, nsub, theta = model.step(theta, suff, my_data, do_reconstruction=True)
        np.testing.assert_allclose(F, float(g["t%d_F" % t]), rtol=1e-9)
        assert np.array_equal(np.packbits(suff["ss"], axis=-1), g["t%d_ss_out" % t])
        want = g["t%d_y_reconstructed" % t]
        got = my_data["y_reconstructed"]
        np.testing.assert_allclose(got, want, rtol=1e-8, atol=1e-9 * max(1.0, float(np.abs(want).max())))
        assert np.array_equal(got[x], Y[x])


@pytest.mark.parametrize("device_mstep", [False, True])
def test_missing_data_ebsc_against_reference(engine, device_mstep):
    """EBSC on incomplete data (image-inpainting set-up: NaN at the missing entries, x_infr = x = ~isnan):
    standard_init, masked lpj / selection, reconstruction feeding the same step's M-step, the step that
    reuses an older y_reconstructed -- against tests/golden/missing_ebsc.npz recorded from the reference."""
    from evo_amd.models import BSC
    g = load_golden("missing_ebsc.npz")
    D, H, S, N = int(g["D"]), int(g["H"]), int(g["S"]), int(g["N"])
    Y, x_infr = g["Y"], g["x_infr"]
    my_data = {"y": Y, "x_infr": x_infr, "x": x_infr.copy()}
    # device_mstep: the Theta update of bsc.py:226-277 on the device too (reliable-entry count in the sigma
    # update and in ljc, bsc.py:113-118,266-272)
    model = BSC(D, H, S, engine=engine, device_mstep=device_mstep)
    theta = {k: np.array(g["t0_in_%s" % k]) for k in BSC_KEYS}
    for k in ("pi", "sigma"):
        theta[k] = np.float64(theta[k])
    suff = make_suff(g, unpack_bits(g["t0_ss_in"], H))
    for t in range(int(g["n_steps"])):
        np.random.seed(1000 + int(g["seed"]) + t)
        F, nu, nsub, theta = model.step(theta, suff, my_data, do_reconstruction=bool(g["t%d_do_rec" % t]))
        np.testing.assert_allclose(F, float(g["t%d_F" % t]), rtol=1e-9, err_msg="F step %d" % t)
        assert np.array_equal(np.packbits(suff["ss"], axis=-1), g["t%d_ss_out" % t]), "K^n step %d" % t
        np.testing.assert_allclose(suff["lpj"], g["t%d_lpj_out" % t], rtol=1e-9)
        want = g["t%d_y_reconstructed" % t]
        np.testing.assert_allclose(my_data["y_reconstructed"], want, rtol=1e-8, atol=1e-9)
        for k in BSC_KEYS:
            ref = g["t%d_out_%s" % (t, k)]
            np.testing.assert_allclose(theta[k], ref, rtol=1e-7, atol=1e-9 * max(1.0, float(np.abs(ref).max())), err_msg=k)
    # per-datapoint operator with this_x_infr (bsc.py:80-95) against the E-step's own numbers
    model.E_step_precompute(theta, suff, my_data)
    n = 3
    my_data["this_y"], my_data["this_x_infr"] = Y[n], x_infr[n]
    suff["this_states"] = suff["ss"][n]
    one = model.log_pseudo_joint(theta, suff, my_data)
    from oracle import evo_oracle as orc
    th = dict(theta)
    orc.bsc_precompute(th, D, H, x_infr)
    np.testing.assert_allclose(one, orc.bsc_lpj(th, suff["ss"][n], Y[n], orc.new_counters(), x_infr[n]), rtol=1e-10)


@pytest.mark.parametrize("device_mstep", [False, True])
def test_missing_data_es3c_against_reference(engine, device_mstep):
    """ES3C on incomplete data (per-datapoint W_obs^T W_obs formed inside the wavefront kernel): F, K^n, lpj,
    y_reconstructed and Theta of two chained steps against tests/golden/missing_es3c.npz (reference with
    use_storage=False, do_reconstruction=True)."""
    from evo_amd.models import SSSC
    g = load_golden("missing_es3c.npz")
    D, H, S, N = int(g["D"]), int(g["H"]), int(g["S"]), int(g["N"])
    Y, x_infr = g["Y"], g["x_infr"]
    my_data = {"y": Y, "x_infr": x_infr, "x": x_infr.copy()}
    model = SSSC(D, H, S, engine=engine, device_mstep=device_mstep)  # device: sssc.py:352-357,747-755 in the kernels
    theta = {k: np.array(g["t0_in_%s" % k]) for k in SSSC_KEYS}
    theta["sigma2"] = np.float64(theta["sigma2"])
    suff = make_suff(g, unpack_bits(g["t0_ss_in"], H))
    for t in range(int(g["n_steps"])):
        np.random.seed(1000 + int(g["seed"]) + t)
        F, nu, nsub, theta = model.step(theta, suff, my_data, do_reconstruction=True)
        np.testing.assert_allclose(F, float(g["t%d_F" % t]), rtol=1e-9, err_msg="F step %d" % t)
        assert np.array_equal(np.packbits(suff["ss"], axis=-1), g["t%d_ss_out" % t]), "K^n step %d" % t
        np.testing.assert_allclose(suff["lpj"], g["t%d_lpj_out" % t], rtol=1e-9)
        np.testing.assert_allclose(my_data["y_reconstructed"], g["t%d_y_reconstructed" % t], rtol=1e-8, atol=1e-9)
        for k in SSSC_KEYS:
            ref = g["t%d_out_%s" % (t, k)]
            np.testing.assert_allclose(theta[k], ref, rtol=1e-6, atol=1e-8 * max(1.0, float(np.abs(ref).max())), err_msg=k)
    with pytest.raises(ValueError):
        model.step(theta, suff, my_data, do_reconstruction=False)
    # per-datapoint operator with this_x_infr (sssc.py:245-322) against the oracle
    from oracle import evo_oracle as orc
    model.E_step_precompute(theta, suff, my_data)
    n = 5
    my_data["this_y"], my_data["this_x_infr"] = Y[n], x_infr[n]
    suff["this_states"] = suff["ss"][n]
    one = model.log_pseudo_joint(theta, suff, my_data)
    th = dict(theta)
    orc.sssc_precompute(th, D, x_infr)
    np.testing.assert_allclose(one, orc.sssc_lpj(th, suff["ss"][n], Y[n], orc.new_counters(), {}, x_infr[n]), rtol=1e-9)
