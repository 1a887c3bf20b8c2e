"""GPU parity tests of the C-ABI kernels against the golden fixtures (which were generated
from the reference itself) -- run with ``-m gpu`` on an MI355X.  Every call goes through
libevo_amd.so via ctypes; there is no CPU fallback to fall back to."""
import numpy as np
import pytest

from conftest import load_golden, unpack_bits

pytestmark = pytest.mark.gpu

LPJ_RTOL = 1e-9   # north_star allows 1e-5; the kernels are expected to sit near 1e-13
SUM_RTOL = 1e-9


@pytest.fixture(scope="module")
def engine():
    from evo_amd.engine import Engine
    eng = Engine()
    yield eng
    eng.close()


def _close(a, b, rtol, name=""):
    a, b = np.asarray(a), np.asarray(b)
    scale = max(1.0, float(np.abs(b).max())) if b.size else 1.0
    np.testing.assert_allclose(a, b, rtol=rtol, atol=rtol * scale, err_msg=name)


def test_lpj_bsc_kat(engine):
    g = load_golden("lpj_bsc.npz")
    H = int(g["H"])
    states = unpack_bits(g["states"], H)
    C = states.shape[0]
    Y = g["Y"]
    N, D = Y.shape
    engine.configure("bsc", N, D, H, C, 0, 8)
    engine.upload_data(Y)
    engine.upload_states(np.tile(states[None], (N, 1, 1)))
    ljc = engine.set_params_bsc(g["W"], float(g["pi"]), float(g["sigma"]))
    np.testing.assert_allclose(ljc, float(g["ljc"]), rtol=1e-14)
    engine.lpj_resident()                      # Gram-form batch kernel (k from 0 to H)
    _close(engine.download_lpj(), g["lpj"], 1e-11, "resident, Gram form")
    _close(engine.lpj_shared(states), g["lpj"], 1e-13, "shared")
    for n in range(N):
        out, flags = engine.lpj_single(Y[n], states)
        _close(out, g["lpj"][n], 1e-13, "single")
        assert not flags.any()
    try:                                       # direct residual form: the reference's own arithmetic
        engine.set_option("bsc_direct", 1)
        engine.set_params_bsc(g["W"], float(g["pi"]), float(g["sigma"]))
        engine.lpj_resident()
        _close(engine.download_lpj(), g["lpj"], 1e-13, "resident, direct form")
    finally:
        engine.set_option("bsc_direct", 0)


def test_lpj_sssc_kat(engine):
    """k = 0 ... 24 active latents with a dense non-symmetric Psi: exercises the K=4 register
    kernel, the K=8 overflow kernel and the LDS wavefront kernel."""
    g = load_golden("lpj_sssc.npz")
    H = int(g["H"])
    states = unpack_bits(g["states"], H)
    C = states.shape[0]
    Y = g["Y"]
    N, D = Y.shape
    engine.configure("sssc", N, D, H, C, 0, 8)
    engine.upload_data(Y)
    engine.upload_states(np.tile(states[None], (N, 1, 1)))
    ljc = engine.set_params_sssc(g["W"], g["pies"], g["mus"], g["Psi"], float(g["sigma2"]))
    np.testing.assert_allclose(ljc, float(g["ljc"]), rtol=1e-14)
    engine.lpj_resident()
    got = engine.download_lpj()
    k = states.sum(axis=1)
    for lo, hi in ((0, 4), (5, 8), (9, 64)):
        m = (k >= lo) & (k <= hi)
        assert m.any()
        _close(got[:, m], g["lpj"][:, m], 1e-11, "k in [%d,%d]" % (lo, hi))
    _close(engine.lpj_shared(states), g["lpj"], 1e-11, "shared")
    out, flags = engine.lpj_single(Y[1], states)
    _close(out, g["lpj"][1], 1e-11, "single")


@pytest.mark.parametrize("fixture,n_sing,census", [("lpj_sssc_singular.npz", 7, 1), ("lpj_sssc_singular_k3.npz", 11, 1),
                                                   ("lpj_sssc_singular_k3.npz", 11, 0), ("lpj_sssc_indefinite.npz", 15, 1),
                                                   ("lpj_sssc_indefinite.npz", 15, 0), ("lpj_sssc_dense.npz", 0, 1),
                                                   ("lpj_sssc_dense.npz", 0, 0)])
def test_lpj_sssc_singular_psi(engine, fixture, n_sing, census):
    """States whose Psi_A is EXACTLY singular (np.linalg.inv raises: a zero variance, two equal rows, rows in a
    power-of-two ratio, a rank-2 3 x 3 block): the reference goes on with pinv(Psi_A) and slogdet = -inf, i.e. lpj = +inf
    -> B_max with the isinf counter, and its statistics use Lam = inv(G_A / sigma2 + pinv(Psi_A)) (sssc.py:278-301).
    The fixtures hold the reference's lpj, lambda_s and kappa_s.  At most two active latents (`lpj_sssc_singular`): served
    by the state-term tables (resident pass, per-datapoint operator) and by the K = 2 register path (shared sets).  Three
    to ten (`_k3`, with one M_A that is exactly singular as well): the levels pass every state on to the pivoting
    wavefront kernel, which screens Psi_A with LAPACK's elimination and takes the pinv branches (option
    "lpj_singular_screen", on by itself here because the tables kernel has seen a dead / a duplicated latent); census = 0:
    the same through the level chains of the register kernels instead of the census lists + quad kernels.
    `_indefinite` (round 4): the other pinv branch (sssc.py:295-300) -- Psi_A REGULAR, M_A = G_A / sigma2 + inv(Psi_A)
    exactly singular, inside an indefinite Psi; 1 to 20 active latents, the tables, the K = 2 register path and the
    wavefront kernel's screen of M_A (switched on by the tables kernel, which has seen a singular 1 x 1 / 2 x 2 M block).
    `_dense`: up to 150 of H = 150 latents active -- above SSSC_KCAP = 64 the wavefront kernel works in global memory.
    The statistics against the oracle's restatement of the reference loop on the same K^n."""
    from oracle import evo_oracle as orc
    g = load_golden(fixture)
    engine.set_option("census_lists", census)
    try:
        _singular_psi_body(engine, orc, g, fixture, n_sing)
    finally:
        engine.set_option("census_lists", 1)


def _singular_psi_body(engine, orc, g, fixture, n_sing):
    H = int(g["H"])
    states = unpack_bits(g["states"], H)
    C = states.shape[0]
    Y = g["Y"]
    N, D = Y.shape
    theta = {k: np.array(g[k]) for k in ("W", "pies", "mus", "Psi")}
    theta["sigma2"] = np.float64(g["sigma2"])
    engine.configure("sssc", N, D, H, C, 0, 8)
    engine.upload_data(Y)
    engine.upload_states(np.tile(states[None], (N, 1, 1)))
    engine.set_params_sssc(theta["W"], theta["pies"], theta["mus"], theta["Psi"], float(theta["sigma2"]))
    engine.lpj_resident()
    got = engine.download_lpj()
    sing = g["lpj"][0] == 0.0
    assert sing.sum() == n_sing
    assert (got[:, sing] == 0.0).all()  # B_max exactly
    _close(got[:, ~sing], g["lpj"][:, ~sing], 1e-11, "regular states beside singular ones")
    shared = engine.lpj_shared(states)  # K = 2 register path (no tables)
    assert (shared[:, sing] == 0.0).all()
    _close(shared[:, ~sing], g["lpj"][:, ~sing], 1e-11, "shared set")
    out, flags = engine.lpj_single(Y[1], states)
    assert (out[sing] == 0.0).all()
    _close(out[~sing], g["lpj"][1][~sing], 1e-11, "single")
    # statistics of this K^n: the singular states carry weight exp(0 - max) = the dominant weight of their row
    engine.lpj_resident()
    v = engine.acc_views(engine.stats())
    suff = {"ss": np.tile(states[None], (N, 1, 1)), "lpj": np.empty((N, C)), "S_perm": 0, "incl": np.zeros((0, H), dtype=bool),
            "Mprime": C}
    with np.errstate(all="ignore"):
        want = orc.sssc_EM_accumulate(dict(theta), suff, Y, use_storage=False, evolve=False)
    for name in ("xpt_s", "xpt_ss", "xpt_sz", "xpt_szsz", "Wp", "s_sz_outer", "sz_sz_outer"):
        ref = want[name]
        assert np.abs(v[name] - ref).max() <= 1e-10 * max(1.0, np.abs(ref).max()), name
    if n_sing:
        assert int(v["reset_isinf"]) >= 1  # the +inf values were counted (_models.py:589-590)
    if "k3" in fixture:
        # the screen off: the Gram form's continuous limit (finite values) for the singular states above two latents
        big = sing & (states.sum(axis=1) > 2)
        engine.set_option("lpj_singular_screen", 0)
        try:
            engine.lpj_resident()
            off = engine.download_lpj()
        finally:
            engine.set_option("lpj_singular_screen", 1)
        assert (off[:, big] != 0.0).all() and np.isfinite(off).all()
        _close(off[:, ~big], got[:, ~big], 1e-13, "states the screen does not concern")
        engine.set_option("lpj_singular_screen", 2)  # always on: same values
        try:
            engine.lpj_resident()
            _close(engine.download_lpj(), got, 0.0, "screen forced")
        finally:
            engine.set_option("lpj_singular_screen", 1)


def test_lpj_sssc_singular_psi_incomplete_data(engine):
    """The same exactly singular Psi_A with incomplete data (sssc.py:276: W[this_x_infr, :] -- the Gram block belongs to the
    datapoint, every state runs on the wavefront kernel): B_max exactly for the singular states whatever their number of
    latents, the others against the oracle's masked lpj (pinned by step_*_missing.npz)."""
    from oracle import evo_oracle as orc
    g = load_golden("lpj_sssc_singular_k3.npz")
    H = int(g["H"])
    states = unpack_bits(g["states"], H)
    C = states.shape[0]
    Y = g["Y"]
    N, D = Y.shape
    theta = {k: np.array(g[k]) for k in ("W", "pies", "mus", "Psi")}
    theta["sigma2"] = np.float64(g["sigma2"])
    mask = np.random.RandomState(5).random_sample((N, D)) < 0.7
    mask[:, :4] = True
    engine.configure("sssc", N, D, H, C, 0, 8)
    try:
        engine.upload_data(Y)
        engine.upload_masks(mask)
        engine.upload_states(np.tile(states[None], (N, 1, 1)))
        engine.set_params_sssc(theta["W"], theta["pies"], theta["mus"], theta["Psi"], float(theta["sigma2"]))
        engine.lpj_resident()
        got = engine.download_lpj()
    finally:
        engine.configure("sssc", 1, D, H, C, 0, 8)  # drops the masks for the tests that follow on this engine
    sing = g["lpj"][0] == 0.0
    want = np.zeros((N, C))
    for n in range(N):
        th = dict(theta)
        cnt = orc.sssc_precompute(th, D, mask)
        with np.errstate(all="ignore"):
            want[n] = orc.sssc_lpj(th, states, Y[n], cnt, {}, obs=mask[n])
    assert (want[:, sing] == 0.0).all() and (got[:, sing] == 0.0).all()
    _close(got[:, ~sing], want[:, ~sing], 1e-11, "regular states, incomplete data")


def test_reconfigure_after_masks_drops_them(engine):
    """Round-2 abort (gpurun_out/r2_tests1.txt: `Fatal Python error: Aborted` at the first synchronisation after the lpj
    pass of a shape test that followed the missing-data tests on the shared engine): evoamd_configure kept mask_infr /
    mask_x / Yrec of the PREVIOUS geometry -- N x D buffers of a smaller shard -- and the next, larger and complete
    data set ran the masked kernels over them, out of bounds.  (The same commit also added a dynamic-LDS attribute for
    two new instantiations; a missing attribute is a launch error code, not a process abort, and the attribution run
    of the isolated test passed: the order of the tests was the trigger.)  Masks on a small geometry, then a larger
    unmasked configure on the same context: the pass must equal the reference values, i.e. run unmasked."""
    g = load_golden("lpj_sssc.npz")
    H = int(g["H"])
    states = unpack_bits(g["states"], H)
    C = states.shape[0]
    Y = g["Y"]
    N, D = Y.shape
    rng = np.random.RandomState(2)
    engine.configure("sssc", 1, D, H, C, 0, 8)
    engine.upload_data(Y[:1])
    engine.upload_masks(rng.random_sample((1, D)) < 0.6)
    assert engine.has_masks
    engine.upload_states(states[None])
    engine.set_params_sssc(g["W"], g["pies"], g["mus"], g["Psi"], float(g["sigma2"]))
    engine.lpj_resident()
    masked = engine.download_lpj()
    assert not np.allclose(masked, g["lpj"][:1], rtol=1e-6)  # the holes do change the values
    reps = 400  # large enough that rows of a stale 1 x D mask would be far out of bounds
    engine.configure("sssc", N * reps, D, H, C, 0, 8)
    assert not engine.has_masks
    engine.upload_data(np.tile(Y, (reps, 1)))
    engine.upload_states(np.tile(states[None], (N * reps, 1, 1)))
    engine.set_params_sssc(g["W"], g["pies"], g["mus"], g["Psi"], float(g["sigma2"]))
    engine.lpj_resident()
    engine.synchronize()
    _close(engine.download_lpj(), np.tile(g["lpj"], (reps, 1)), 1e-11, "unmasked pass after a masked geometry")


def test_lpj_sssc_dense_candidates(engine):
    """States with up to 24 active latents as a CANDIDATE batch (kernel TAG 1): the level chain of the candidates
    ends in sssc_big_kernel<0, 1> at SSSC_KCAP (98 KB of dynamic LDS -- the instantiation whose missing
    MaxDynamicSharedMemorySize attribute was the second change of the round-2 fix); test_lpj_sssc_kat runs the same
    states as resident K^n (TAG 0) and as a shared set (TAG 2)."""
    g = load_golden("lpj_sssc.npz")
    H = int(g["H"])
    states = unpack_bits(g["states"], H)
    C = states.shape[0]
    Y = g["Y"]
    N, D = Y.shape
    k = states.sum(axis=1)
    assert (k > 8).any() and k.max() <= 64
    engine.configure("sssc", N, D, H, 4, 0, C)
    engine.upload_data(Y)
    engine.upload_states(np.tile(states[None, :4], (N, 1, 1)))
    engine.set_params_sssc(g["W"], g["pies"], g["mus"], g["Psi"], float(g["sigma2"]))
    counts = np.array([C, C - 5, C][:N], dtype=np.int32)  # ragged
    got = engine.lpj_candidates(np.tile(states[None], (N, 1, 1)), counts)
    for n in range(N):
        _close(got[n, :counts[n]], g["lpj"][n, :counts[n]], 1e-11, "candidates of datapoint %d" % n)


def test_lpj_candidates_unstaged_rows(engine):
    """A candidate batch whose B rows do not fit the LDS of the table-driven lpj kernel (1024 / Cmax datapoints per
    workgroup at H = 256: 0.5 MB): it runs on that kernel with the B values gathered from global memory (option
    "lpj_main_unstaged", default), before on the K = 2 register kernel.  Both against the oracle, and against each other."""
    from oracle import evo_oracle as orc
    rng = np.random.RandomState(5)
    N, D, H, S, C = 300, 12, 256, 4, 4
    Y = rng.normal(size=(N, D))
    A = rng.normal(size=(H, H)) * 0.05
    theta = {"W": rng.normal(size=(D, H)), "pies": rng.uniform(0.05, 0.4, H), "mus": rng.normal(size=H),
             "Psi": np.eye(H) + A @ A.T + rng.normal(size=(H, H)) * 0.01, "sigma2": np.float64(0.8)}
    cand = np.zeros((N, C, H), dtype=bool)
    ks = rng.randint(0, 4, size=(N, C))  # 0..3 active latents: the table path and its overflow level
    for n in range(N):
        for c in range(C):
            cand[n, c, rng.choice(H, ks[n, c], replace=False)] = True
    counts = rng.randint(1, C + 1, size=N).astype(np.int32)
    engine.configure("sssc", N, D, H, S, 0, C)
    engine.upload_data(Y)
    engine.upload_states(np.zeros((N, S, H), dtype=bool))
    engine.set_params_sssc(theta["W"], theta["pies"], theta["mus"], theta["Psi"], float(theta["sigma2"]))
    got = {}
    for opt in (1, 0):
        engine.set_option("lpj_main_unstaged", opt)
        try:
            got[opt] = engine.lpj_candidates(cand, counts)
        finally:
            engine.set_option("lpj_main_unstaged", 1)
    orc.sssc_precompute(theta, D)
    for n in range(0, N, 37):
        want = orc.sssc_lpj(theta, cand[n, :counts[n]], Y[n], orc.new_counters(), {})
        _close(got[1][n, :counts[n]], want, 1e-10, "unstaged table kernel, datapoint %d" % n)
    for n in range(N):
        _close(got[1][n, :counts[n]], got[0][n, :counts[n]], 1e-11, "table kernel vs K = 2 register kernel")


def test_vary_kn_kat(engine):
    g = load_golden("vary_kn.npz")
    for i in range(int(g["n_cases"])):
        H, S, Mp = int(g["c%d_H" % i]), int(g["c%d_S" % i]), int(g["c%d_Mprime" % i])
        old = unpack_bits(g["c%d_old" % i], H)
        new = unpack_bits(g["c%d_new" % i], H).reshape(-1, H)
        C = new.shape[0]
        Cmax = max(C, 1)
        engine.configure("bsc", 1, 4, H, S, 0, Cmax)
        engine.upload_states(old[None])
        engine.upload_lpj(g["c%d_lpj_old" % i][None])
        cand = np.zeros((1, Cmax, H), dtype=bool)
        cand[0, :C] = new
        lpj_new = np.zeros((1, Cmax))
        lpj_new[0, :C] = g["c%d_lpj_new" % i]
        engine.set_candidates(cand, np.array([C], dtype=np.int32), lpj_new)
        sums = engine.vary_kn(Mp)
        assert list(sums) == list(g["c%d_ret" % i]), i
        assert np.array_equal(np.packbits(engine.download_states()[0], axis=-1), g["c%d_states_out" % i]), i
        assert np.array_equal(engine.download_lpj()[0], g["c%d_lpj_out" % i]), i
        engine.set_estep_counts(0.0, 0.0)


STEP_FIXTURES = ["ebsc_bars", "es3c_bars", "ebsc_mid", "es3c_mid", "es3c_dense", "ebsc_dense",
                 "ebsc_sparseflip", "es3c_cross", "ebsc_gen2", "ebsc_perm", "es3c_perm"]


@pytest.mark.parametrize("name", STEP_FIXTURES)
def test_step_kernels(engine, name):
    """One EM step at the C-ABI level with the candidate batches the reference generated:
    lpj(K^n) -> lpj(candidates) -> vary_Kn -> statistics, every intermediate compared."""
    g = load_golden("step_%s.npz" % name)
    algo = str(g["algo"])
    D, H, S, N = int(g["D"]), int(g["H"]), int(g["S"]), int(g["N"])
    Y = g["Y"]
    for t in range(int(g["n_steps"])):
        counts = g["t%d_cand_counts" % t].astype(np.int32)
        Cmax = max(int(counts.max()), 1)
        S_perm = int(g["S_perm"]) if "S_perm" in g else 0  # permanent all-zero state: lpj column 0
        engine.configure(algo[1:] if algo == "ebsc" else "sssc", N, D, H, S, S_perm, Cmax)
        engine.upload_data(Y)
        engine.upload_states(unpack_bits(g["t%d_ss_in" % t], H))
        if algo == "ebsc":
            engine.set_params_bsc(g["t%d_in_W" % t], float(g["t%d_in_pi" % t]), float(g["t%d_in_sigma" % t]))
        else:
            engine.set_params_sssc(g["t%d_in_W" % t], g["t%d_in_pies" % t], g["t%d_in_mus" % t],
                                   g["t%d_in_Psi" % t], float(g["t%d_in_sigma2" % t]))
        engine.lpj_resident()
        # candidates: ragged -> padded
        flat = unpack_bits(g["t%d_cand_states" % t], H).reshape(-1, H)
        cand = np.zeros((N, Cmax, H), dtype=bool)
        ref_lpj = np.zeros((N, Cmax))
        off = 0
        for n in range(N):
            c = int(counts[n])
            cand[n, :c] = flat[off:off + c]
            ref_lpj[n, :c] = g["t%d_cand_lpj" % t][off:off + c]
            off += c
        got = engine.lpj_candidates(cand, counts)
        mask = np.arange(Cmax)[None, :] < counts[:, None]
        _close(got[mask], ref_lpj[mask], LPJ_RTOL, "candidate lpj")
        sums = engine.vary_kn(int(g["ea_Mprime"]))
        if int(g["ea_n_generations"]) == 1:
            # with >1 generation the reference's trace holds all generations at once; selection
            # below is still exact because vary_Kn sees the same union of candidates
            pass
        assert np.array_equal(np.packbits(engine.download_states(), axis=-1), g["t%d_ss_out" % t])
        _close(engine.download_lpj(), g["t%d_lpj_out" % t], LPJ_RTOL, "lpj after selection")
        assert sums[0] == float(g["t%d_S_nunique" % t]) * N and sums[1] == float(g["t%d_S_sub" % t]) * N
        v = engine.acc_views(engine.stats())
        names = (("Wp", "Wq", "pies", "sigma", "Fs") if algo == "ebsc" else
                 ("xpt_s", "xpt_ss", "xpt_sz", "xpt_szsz", "s_sz_outer", "sz_sz_outer", "Wp", "y_outer_diag", "Fs"))
        for nm in names:
            _close(v[nm], g["t%d_sum_%s" % (t, nm)], SUM_RTOL, nm)
        assert float(v["N"]) == N and float(v["sum_nunique"]) == sums[0] and float(v["sum_sub"]) == sums[1]
        F = engine.ljc + float(v["Fs"]) / N
        np.testing.assert_allclose(F, float(g["t%d_F" % t]), rtol=1e-10)


def test_free_energy_kernel(engine):
    rng = np.random.RandomState(0)
    lpj = rng.normal(size=(37, 19)) * 30 - 100
    from scipy.special import logsumexp
    want = logsumexp(lpj, axis=1).sum()
    np.testing.assert_allclose(engine.free_energy_sum(lpj), want, rtol=1e-13)


def test_clamp_flags(engine):
    """NaN / inf handling of lpj_reset_check through the single-datapoint operator."""
    H, D = 6, 4
    engine.configure("bsc", 1, D, H, 2, 0, 1)
    W = np.ones((D, H))
    W[0, 0] = np.inf
    engine.set_params_bsc(W, 0.2, 1.0)
    st = np.zeros((2, H), dtype=bool)
    st[0, 0] = True   # residual inf -> pre1*inf = -inf -> clamped to B_max = 0.0
    st[1, 1] = True
    out, flags = engine.lpj_single(np.zeros(D), st)
    assert out[0] == 0.0 and np.isfinite(out[1])
    assert list(flags) == [0, 1, 1]
    W[0, 0] = np.nan
    engine.set_params_bsc(W, 0.2, 1.0)
    out, flags = engine.lpj_single(np.zeros(D), st)
    assert out[0] == np.finfo(np.float64).min
    assert flags[0] == 1


def test_rccl_single_rank_allreduce(engine):
    """RCCL is reached through dlopen inside libevo_amd; a 1-rank communicator must initialise on
    the GPU box and leave the packed accumulator unchanged (sum over one rank).  Multi-rank runs
    are the driver's (8-GPU node); the N>1 host logic is covered by test_multirank_gloo.py."""
    from evo_amd.engine import Engine
    from evo_amd.utils import parallel
    g = load_golden("step_es3c_bars.npz")
    D, H, S, N = int(g["D"]), int(g["H"]), int(g["S"]), int(g["N"])
    eng = Engine()
    try:
        eng.configure("sssc", N, D, H, S, 0, 4)
        eng.upload_data(g["Y"])
        eng.upload_states(unpack_bits(g["t0_ss_in"], H))
        eng.set_params_sssc(g["t0_in_W"], g["t0_in_pies"], g["t0_in_mus"], g["t0_in_Psi"], float(g["t0_in_sigma2"]))
        eng.lpj_resident()
        before = eng.stats()
        comm = parallel.RcclComm(eng, 0, 1, Engine.comm_unique_id())
        assert comm.device_reduces and comm.size == 1
        after = eng.stats()
        np.testing.assert_allclose(after, before, rtol=1e-12, atol=1e-14)
        np.testing.assert_array_equal(comm.allreduce_array(np.arange(5.0)), np.arange(5.0))
        assert comm.allreduce_max(3.5) == 3.5 and comm.allreduce(7) == 7
        assert comm.bcast(np.float64(2.5)) == 2.5
        comm.Barrier()
        comm.close()
    finally:
        eng.close()


def _moment_matrix(rng, n, cond_boost=0.05):
    X = rng.standard_normal((n, 3 * n))
    return X @ X.T / (3 * n) + cond_boost * np.eye(n)


@pytest.mark.parametrize("n", [7, 16, 31, 32, 33, 40, 48, 64, 65, 100, 128, 130, 192, 200, 500])
def test_inverse_spd_block_path(engine, n):
    """The M-step's H x H solver on Gram-type (SPD) matrices -- the shape np.linalg.inv is given
    at sssc.py:693,738 and lstsq at bsc.py:237 -- against numpy, two matrices per call."""
    rng = np.random.default_rng(100 + n)
    engine.configure("bsc", 8, 4, n, 4, 0, 4)
    A, B = _moment_matrix(rng, n), _moment_matrix(rng, n, 1e-3) + np.diag(rng.uniform(0, 2, n))
    engine.set_option("inverse_spd", 1)
    Ai, Bi, _ = engine.inverse(A, B)
    for M, Mi in ((A, Ai), (B, Bi)):
        ref = np.linalg.inv(M)
        assert np.abs(Mi - ref).max() <= 1e-10 * np.abs(ref).max()
        assert np.abs(Mi @ M - np.eye(n)).max() < 1e-9
    # the partially pivoted path gives the same inverse to rounding
    engine.set_option("inverse_spd", 0)
    Ap, Bp, _ = engine.inverse(A, B)
    engine.set_option("inverse_spd", 1)
    assert np.abs(Ap - Ai).max() <= 1e-11 * np.abs(Ai).max()
    assert np.abs(Bp - Bi).max() <= 1e-11 * np.abs(Bi).max()
    # n >= 32 ran with 32-column block steps (32 x 32 pivot block inverted in registers through its Schur
    # complement); the 16-column steps must agree to rounding
    engine.set_option("inverse_block", 16)
    A16, B16, _ = engine.inverse(A, B)
    engine.set_option("inverse_block", 32)
    A32, B32, _ = engine.inverse(A, B)
    engine.set_option("inverse_block", 0)  # default: 32 columns from n = 256 on
    for Mi2 in (A16, A32):
        assert np.abs(Mi2 - Ai).max() <= 1e-11 * np.abs(Ai).max()
    for Mi2 in (B16, B32):
        assert np.abs(Mi2 - Bi).max() <= 1e-11 * np.abs(Bi).max()


@pytest.mark.parametrize("n", [40, 96, 256, 300])
def test_inverse_block32_nonsymmetric(engine, n):
    """xpt_szsz is not symmetric once Psi is not (quirk Q2): the 32-column steps invert the 32 x 32 pivot block
    through its Schur complement WITHOUT assuming symmetry.  Diagonally dominant non-symmetric matrices."""
    rng = np.random.default_rng(50 + n)
    engine.configure("bsc", 8, 4, n, 4, 0, 4)
    A = _moment_matrix(rng, n) + 0.05 * rng.standard_normal((n, n)) / np.sqrt(n)
    B = _moment_matrix(rng, n, 0.2) + 0.1 * np.triu(rng.standard_normal((n, n))) / np.sqrt(n)
    engine.set_option("inverse_block", 32)
    try:
        Ai, Bi, _ = engine.inverse(A, B)
    finally:
        engine.set_option("inverse_block", 0)
    for M, Mi in ((A, Ai), (B, Bi)):
        ref = np.linalg.inv(M)
        assert np.abs(Mi - ref).max() <= 1e-10 * np.abs(ref).max()


@pytest.mark.parametrize("n", [33, 128, 200, 300])
def test_inverse_falls_back_to_pivoting(engine, n):
    """A matrix that is not SPD (zero leading entry, general entries) makes the block path report a
    bad pivot; the library repeats with partial pivoting and still returns the inverse."""
    rng = np.random.default_rng(7 + n)
    engine.configure("bsc", 8, 4, n, 4, 0, 4)
    G = rng.standard_normal((n, n))
    G[0, 0] = 0.0
    A = _moment_matrix(rng, n)
    engine.set_option("inverse_spd", 1)
    Ai, Gi, _ = engine.inverse(A, G)
    assert np.abs(Gi @ G - np.eye(n)).max() < 1e-8
    assert np.abs(Ai @ A - np.eye(n)).max() < 1e-9
    ref = np.linalg.inv(G)
    assert np.abs(Gi - ref).max() <= 1e-9 * np.abs(ref).max()


def test_inverse_singular_is_reported(engine):
    from evo_amd._lib import EvoAmdError
    n = 48
    rng = np.random.default_rng(3)
    engine.configure("bsc", 8, 4, n, 4, 0, 4)
    A = _moment_matrix(rng, n)
    A[:, 5] = 0.0
    A[5, :] = 0.0  # a latent that never occurs: exactly singular moment matrix
    with pytest.raises(EvoAmdError):
        engine.inverse(A)


@pytest.mark.parametrize("H,S", [(256, 40), (384, 24), (512, 200)])
def test_lpj_sssc_wide_states(engine, H, S):
    """ES3C lpj of resident states with several 64-bit words per state (the LDS-staged main kernel,
    closed 2 x 2 form, plus the overflow levels) against the oracle's sssc.py:241-326 restatement."""
    from oracle import evo_oracle as orc
    rng = np.random.RandomState(H + S)
    N, D = 12, 24
    Y = rng.normal(size=(N, D))
    A = rng.normal(size=(H, H)) * 0.05
    theta = {"W": rng.normal(size=(D, H)) * 0.3, "pies": rng.uniform(0.05, 0.4, H), "mus": rng.normal(size=H),
             "Psi": np.eye(H) + A + 0.3 * A.T, "sigma2": np.float64(0.7)}   # Psi deliberately not symmetric (Q2)
    orc.sssc_precompute(theta, D)
    ks = rng.choice([0, 1, 1, 2, 2, 2, 3, 5, 9], size=(N, S))
    ss = np.zeros((N, S, H), dtype=bool)
    for n in range(N):
        for s in range(S):
            ss[n, s, rng.choice(H, ks[n, s], replace=False)] = True
    engine.configure("sssc", N, D, H, S, 0, 4)
    engine.upload_data(Y)
    engine.upload_states(ss)
    engine.set_params_sssc(theta["W"], theta["pies"], theta["mus"], theta["Psi"], theta["sigma2"])
    engine.lpj_resident()
    got = engine.download_lpj()
    want = np.empty((N, S))
    for n in range(N):
        want[n] = orc.sssc_lpj(theta, ss[n], Y[n], orc.new_counters(), {})
    _close(got, want, LPJ_RTOL, "wide-state ES3C lpj")


@pytest.mark.parametrize("K,M,Nc,sym", [(9000, 384, 256, False), (8200, 640, 256, True), (300, 192, 64, True),
                                        (8192, 258, 256, False), (130, 70, 34, False),
                                        (20000, 1280, 512, True), (16000, 1024, 256, False), (20010, 1152, 512, True)])
def test_gemm_tn_dispatch(engine, K, M, Nc, sym):
    """C = A^T B through the statistics pass's MFMA dispatch: 64- and 128-tile kernels, split K over
    the XCDs, 16-byte and scalar loaders, upper-tiles-plus-mirror for a trailing X^T X block."""
    rng = np.random.default_rng(K + M)
    A = rng.standard_normal((K, M))
    B = rng.standard_normal((K, Nc))
    if sym:
        A[:, M - Nc:] = B
    C = engine.gemm_tn(A, B, M - Nc if sym else -1)
    ref = A.T @ B
    assert np.abs(C - ref).max() <= 1e-11 * np.abs(ref).max()
    if sym:
        blk = C[M - Nc:]
        assert np.abs(blk - blk.T).max() <= 1e-11 * np.abs(ref).max()
    if K >= 8192:
        # stream-K grid: partial tiles through the workspace + reduce kernel (above; a fixed order of additions, so a
        # second call returns the same bits) against the f64 atomic epilogue and against the split-K grid
        C2 = engine.gemm_tn(A, B, M - Nc if sym else -1)
        np.testing.assert_array_equal(C2, C)
        # (the last three shapes fill the resident grid with whole K chunks per tile: grouped split-K, 34 real tiles x 15
        # chunks / 16 x 32 / 30 x 17 -- the north-star contraction's and the EBSC c5 contraction's tilings; option
        # "gemm_grouped" = 0 sends them through stream-K)
        for opt in ("gemm_workspace", "gemm_streamk", "gemm_grouped"):
            engine.set_option(opt, 0)
            try:
                Ca = engine.gemm_tn(A, B, M - Nc if sym else -1)
            finally:
                engine.set_option(opt, 1)
            assert np.abs(Ca - ref).max() <= 1e-11 * np.abs(ref).max()


def test_b_transposed_matches_rowmajor(engine):
    """From N = 8192 datapoints on the context keeps Y^T and computes B = Y W on the 128-tile kernel (option
    b_transposed); the lpj of a resident K^n must not care which product fed it."""
    from oracle import evo_oracle as orc
    rng = np.random.default_rng(3)
    N, D, H, S = 8300, 40, 128, 8
    Y = rng.standard_normal((N, D))
    ss = rng.random((N, S, H)) < 1.5 / H
    W = rng.standard_normal((D, H))
    got = {}
    for opt in (2, 0):  # 2: the transposed product at this (small) H as well
        engine.set_option("b_transposed", opt)
        try:
            engine.configure("bsc", N, D, H, S, 0, 4)
            engine.upload_data(Y)
            engine.upload_states(ss)
            engine.set_params_bsc(W, 0.02, 1.3)
            engine.lpj_resident()
            got[opt] = engine.download_lpj()
        finally:
            engine.set_option("b_transposed", 1)
    np.testing.assert_allclose(got[2], got[0], rtol=1e-12, atol=0)
    th = {"W": W, "pre1": -1.0 / 2.0 / 1.3 / 1.3, "pil_bar": np.log(0.02 / (1.0 - 0.02))}
    want = np.array([orc.bsc_lpj(th, ss[n], Y[n], orc.new_counters()) for n in range(0, N, 997)])
    np.testing.assert_allclose(got[2][::997], want, rtol=1e-10)


def test_inverse_above_blocked_limit(engine):
    """n > 1024: the SPD block path has no size limit; a general matrix falls back to the unblocked
    pivoted elimination (2 n launches)."""
    n = 1040
    rng = np.random.default_rng(11)
    engine.configure("bsc", 8, 4, n, 4, 0, 4)
    A = _moment_matrix(rng, n)
    G = rng.standard_normal((n, n)) + 3.0 * np.eye(n)
    G[0, 0] = 0.0
    Ai, Gi, _ = engine.inverse(A, G)
    assert np.abs(Ai @ A - np.eye(n)).max() < 1e-8
    assert np.abs(Gi @ G - np.eye(n)).max() < 1e-7


def test_stats_twice_in_exact_mode(engine):
    """in census mode the statistics pass appended to the on-the-fly lists (the quad levels' hand-over, the
    16-latent wavefront launch) but marked their counters clean, so a SECOND statistics pass with no chain in between
    served the stale entries again and added their moments twice.  Exact mode (screen = 2) sends every state above two
    latents through those lists: two evoamd_stats calls in a row must give the same accumulator."""
    g = load_golden("lpj_sssc_singular_k3.npz")
    H = int(g["H"])
    states = unpack_bits(g["states"], H)
    C = states.shape[0]
    Y = g["Y"]
    N, D = Y.shape
    engine.set_option("lpj_singular_screen", 2)
    try:
        engine.configure("sssc", N, D, H, C, 0, 8)
        engine.upload_data(Y)
        engine.upload_states(np.tile(states[None], (N, 1, 1)))
        engine.set_params_sssc(g["W"], g["pies"], g["mus"], g["Psi"], float(g["sigma2"]))
        engine.lpj_resident()
        a1 = engine.stats().copy()
        a2 = engine.stats().copy()   # what _step_device_singular / reconstruct() do: statistics again, no chain between
        a3 = engine.stats().copy()
    finally:
        engine.set_option("lpj_singular_screen", 1)
    # (the last 8 entries are the E-step tail: the clamp counters of the lpj pass are consumed by the first call)
    m1, m2, m3 = a1[:-8], a2[:-8], a3[:-8]
    scale = max(1.0, float(np.abs(m1).max()))
    assert np.abs(m2 - m1).max() <= 1e-12 * scale
    assert np.abs(m3 - m1).max() <= 1e-12 * scale
    assert a1[-8] == a2[-8] == a3[-8]  # Fs


def test_stats_large_H_falls_back_to_chains(engine):
    """above H = 1745 the 4-wave statistics kernel's LDS rows do not fit and the census lists cannot be used by
    the statistics pass; it must take the round-2 chains (decided before any level runs) instead of failing."""
    from oracle import evo_oracle as orc
    rng = np.random.RandomState(11)
    N, D, H, S = 12, 24, 1800, 10
    Y = rng.normal(size=(N, D))
    W = rng.normal(size=(D, H))
    pies = rng.uniform(0.1, 0.4, H)
    mus = rng.normal(size=H)
    A = rng.normal(size=(H, 6)) * 0.05
    Psi = np.eye(H) + A @ A.T
    ss = np.zeros((N, S, H), dtype=bool)
    for n in range(N):
        for s in range(S):
            k = 1 + (s % 5)  # 1 .. 5 active latents: every level
            ss[n, s, rng.choice(H, k, replace=False)] = True
    theta = {"W": W, "pies": pies, "mus": mus, "Psi": Psi, "sigma2": np.float64(1.3)}
    engine.configure("sssc", N, D, H, S, 0, 4)
    engine.upload_data(Y)
    engine.upload_states(ss)
    engine.set_params_sssc(W, pies, mus, Psi, 1.3)
    engine.lpj_resident()
    lpj = engine.download_lpj()
    v = engine.acc_views(engine.stats())
    suff = {"ss": ss, "lpj": np.empty((N, S)), "S_perm": 0, "incl": np.zeros((0, H), dtype=bool), "Mprime": S}
    want = orc.sssc_EM_accumulate(dict(theta), suff, Y, use_storage=False, evolve=False)
    _close(lpj, suff["lpj"], 1e-10, "lpj at H = 1800")
    for name in ("xpt_s", "xpt_sz", "xpt_ss", "xpt_szsz", "Wp"):
        ref = want[name]
        assert np.abs(v[name] - ref).max() <= 1e-9 * max(1.0, np.abs(ref).max()), name


def test_out_of_range_list_entry_is_an_error_not_a_fault(engine):
    """list entries and latent indices that cross LDS are range-checked before they become addresses.
    A census list with an out-of-range (datapoint, state) entry must end in the error word, not in a GPU memory fault;
    the context stays usable and the rebuilt census gives the right values."""
    from evo_amd._lib import EvoAmdError
    g = load_golden("lpj_sssc.npz")
    H = int(g["H"])
    states = unpack_bits(g["states"], H)
    C = states.shape[0]
    Y = g["Y"]
    N, D = Y.shape
    engine.configure("sssc", N, D, H, C, 0, 8)
    engine.upload_data(Y)
    engine.upload_states(np.tile(states[None], (N, 1, 1)))
    engine.set_params_sssc(g["W"], g["pies"], g["mus"], g["Psi"], float(g["sigma2"]))
    engine.set_option("debug_poison_list", 1)
    engine.lpj_resident()
    with pytest.raises(EvoAmdError, match="out of range"):
        engine.stats()
    engine.lpj_resident()  # fresh census
    _close(engine.download_lpj(), g["lpj"], 1e-11, "after the poisoned pass")
    engine.stats()
