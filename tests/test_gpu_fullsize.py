"""Size-independent properties at BASELINE.json's FULL sizes (run with ``-m gpu`` on an MI355X).

The oracle finishes in seconds only at small N, so at the north-star shape (ES3C D=256 H=512 S=200, N = 100k on one
GPU), at configs[1] (ES3C H=128, N = 10k), configs[2] (EBSC D=64 H=256 S=128, N = 50k) and configs[4] (EBSC H=1024,
N = 200k on one GPU, float64) the device path is checked through what must hold at any size:

* idempotence -- the lpj pass and the statistics pass leave the same numbers when they run again;
* a checksum of checksums -- the first and second moments the statistics pass accumulates, summed, must equal
  sum_n sum_s q_ns |s| and sum_n sum_s q_ns |s| (|s| - 1) recomputed on the host from the downloaded K^n and lpj
  (softmax weights and popcounts; sssc.py:553-611, bsc.py:193-223), and the free-energy term must equal the
  log-sum-exp of the downloaded rows (_models.py:540-547);
* structure -- E[s s^T] symmetric with E[s] on its diagonal, every datapoint's K^n free of duplicate states
  (variational/utils.py:279-290);
* path independence -- pair bins vs global atomics, stream-K workspace vs atomic epilogue give the same sums;
* monotonicity -- with Theta fixed an E-step can only raise the free energy (vary_Kn keeps the best S states,
  variational/utils.py:292-337).
Everything goes through the C ABI; the workload is bench.py's (same seeds)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

_POP8 = np.array([bin(i).count("1") for i in range(256)], dtype=np.uint8)


def _setup(name, N=None, dense=False, **model_kw):
    import bench
    from evo_amd.engine import Engine
    from evo_amd.models import BSC, SSSC
    cfg = dict(bench.CONFIGS[name])
    if N is not None:
        cfg["N"] = N
    np.random.seed(1234 + 2)
    Y = np.ascontiguousarray(np.random.randn(cfg["N"], cfg["D"]))
    my_data = {"y": Y, "x_infr": np.ones_like(Y, dtype=bool)}
    chunks = list(bench.init_states_packed(cfg, cfg["N"], 4321, max(1, bench.host_cores()), dense=dense))
    eng = Engine()
    cls = BSC if cfg["algo"] == "ebsc" else SSSC
    kw = dict(device_mstep=False)
    kw.update(model_kw)
    model = cls(cfg["D"], cfg["H"], cfg["S"], rng="device", sync_host=False, engine=eng, seed=17, **kw)
    np.random.seed(99)
    theta = model.check_params(model.standard_init(my_data))
    suff = bench.ea_suff(cfg)
    model.attach_resident_states(suff, my_data, chunks)
    return cfg, eng, model, theta, suff, my_data


def _weights_and_counts(eng, cfg):
    """q (N, S) posterior weights of the truncated posterior and k (N, S) = |s| from the device's K^n and lpj."""
    lpj = eng.download_lpj()
    m = lpj.max(axis=1, keepdims=True)
    e = np.exp(lpj - m)
    z = e.sum(axis=1, keepdims=True)
    q = e / z
    Fs = float((np.log(z[:, 0]) + m[:, 0]).sum())
    k = np.empty((cfg["N"], cfg["S"]), dtype=np.int32)
    step = 10000
    for n0 in range(0, cfg["N"], step):  # packed rows, popcount by table
        st = eng.download_states_packed(n0, min(step, cfg["N"] - n0))
        k[n0:n0 + st.shape[0]] = _POP8[st].sum(axis=2, dtype=np.int32)
        if n0 % (7 * step) == 0:  # a sample of datapoints: no duplicate state in K^n
            for r in range(0, st.shape[0], 997):
                assert np.unique(st[r], axis=0).shape[0] == cfg["S"]
    return lpj, q, k, Fs


def _rel(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / max(1e-300, float(np.abs(np.asarray(b)).max())))


@pytest.mark.parametrize("name", ["c2", "c4"])
def test_es3c_full_size_properties(name):
    cfg, eng, model, theta, suff, my_data = _setup(name)
    try:
        N, S, H = cfg["N"], cfg["S"], cfg["H"]
        Fseq = []
        for _ in range(2):  # two full EM iterations: Theta moves, K^n grows denser
            F, nu, nsub, theta = model.step(theta, suff, my_data)
            Fseq.append(F)
        assert np.isfinite(Fseq).all()
        # --- Theta fixed from here on: an E-step can only raise F
        Fe = [model.E_step(theta, suff, my_data)[0] for _ in range(3)]
        assert Fe[1] >= Fe[0] - 1e-12 * abs(Fe[0]) and Fe[2] >= Fe[1] - 1e-12 * abs(Fe[1]), Fe
        assert Fe[2] > Fe[0]
        v1 = eng.acc_views(model.last_acc.copy())
        # --- idempotence: lpj pass (bit for bit), statistics pass (order of the f64 additions only)
        eng.lpj_resident()
        l1 = eng.download_lpj()
        eng.lpj_resident()
        assert np.array_equal(l1, eng.download_lpj())
        v2 = eng.acc_views(eng.stats())
        for name in ("xpt_s", "xpt_ss", "xpt_sz", "xpt_szsz", "Wp", "s_sz_outer", "sz_sz_outer", "y_outer_diag"):
            assert _rel(v2[name], v1[name]) <= 1e-11, name
        # (Fs of the E-step comes out of the selection kernel's row statistics, here out of the row pass: other tree)
        assert abs(float(v2["Fs"]) - float(v1["Fs"])) <= 1e-14 * abs(float(v1["Fs"])) and float(v2["N"]) == N
        # --- checksums from the raw K^n and lpj
        lpj, q, k, Fs = _weights_and_counts(eng, cfg)
        assert abs(Fs - float(v2["Fs"])) <= 1e-11 * abs(Fs)
        s1 = float((q * k).sum())
        s2 = float((q * (k * (k - 1.0))).sum())
        xs, xss = v2["xpt_s"], v2["xpt_ss"]
        assert abs(float(xs.sum()) - s1) <= 1e-10 * s1
        assert np.array_equal(np.diag(xss), xs)
        assert np.array_equal(xss, xss.T)
        assert abs(float(xss.sum() - np.trace(xss)) - s2) <= 1e-10 * max(s2, 1.0)
        # --- path independence
        for opt, val in (("pair_bins", 0), ("gemm_workspace", 0), ("gemm_streamk", 0), ("gemm_grouped", 0)):
            eng.set_option(opt, val)
            try:
                v3 = eng.acc_views(eng.stats())
            finally:
                eng.set_option(opt, 1)
            for name in ("xpt_s", "xpt_ss", "xpt_sz", "xpt_szsz", "Wp", "s_sz_outer", "sz_sz_outer"):
                assert _rel(v3[name], v2[name]) <= 1e-10, (opt, name)
        # sz_sz_outer is a Gram matrix: symmetric, non-negative diagonal
        G = v2["sz_sz_outer"]
        assert _rel(G, G.T) <= 1e-12 and (np.diag(G) >= 0).all()
    finally:
        eng.close()


@pytest.mark.parametrize("name", ["c3", "c5"])
def test_ebsc_full_size_properties(name):
    cfg, eng, model, theta, suff, my_data = _setup(name)
    try:
        N, S, H = cfg["N"], cfg["S"], cfg["H"]
        for _ in range(2):
            F, nu, nsub, theta = model.step(theta, suff, my_data)
            assert np.isfinite(F)
        Fe = [model.E_step(theta, suff, my_data)[0] for _ in range(3)]
        assert Fe[1] >= Fe[0] - 1e-12 * abs(Fe[0]) and Fe[2] >= Fe[1] - 1e-12 * abs(Fe[1]), Fe
        assert Fe[2] > Fe[0]
        v1 = eng.acc_views(model.last_acc.copy())
        eng.lpj_resident()
        l1 = eng.download_lpj()
        eng.lpj_resident()
        assert np.array_equal(l1, eng.download_lpj())
        v2 = eng.acc_views(eng.stats())
        for name in ("Wp", "Wq", "pies", "sigma"):
            assert _rel(v2[name], v1[name]) <= 1e-11, name
        pies, Wq = v2["pies"], v2["Wq"]
        assert np.array_equal(np.diag(Wq), pies)
        assert np.array_equal(Wq, Wq.T)
        if N * S * ((H + 7) // 8) <= 2 << 30:  # (configs[4]: 6.5 GB of packed K^n -- the checks above and below only)
            lpj, q, k, Fs = _weights_and_counts(eng, cfg)
            assert abs(Fs - float(v2["Fs"])) <= 1e-11 * abs(Fs)
            s1 = float((q * k).sum())
            s2 = float((q * (k * (k - 1.0))).sum())
            assert abs(float(pies.sum()) - s1) <= 1e-10 * s1
            assert abs(float(Wq.sum() - np.trace(Wq)) - s2) <= 1e-10 * max(s2, 1.0)
            # sigma accumulates sum_ns q ||y - W s||^2 = sum_ns q (lpj - pil_bar |s|) / pre1 (bsc.py:214-218 through lpj's terms)
            pre1, pil_bar = float(theta["pre1"]), float(theta["pil_bar"])  # left in the dict by E_step_precompute (bsc.py:99-125)
            sig = float((q * ((lpj - pil_bar * k) / pre1)).sum())
            assert abs(float(v2["sigma"]) - sig) <= 1e-9 * abs(sig)
        for opt, val in (("pair_bins", 0), ("bsc_stats_wave", 0), ("gemm_grouped", 0)):
            eng.set_option(opt, val)
            try:
                v3 = eng.acc_views(eng.stats())
            finally:
                eng.set_option(opt, 1)
            for name in ("Wp", "Wq", "pies", "sigma"):
                assert _rel(v3[name], v2[name]) <= 1e-10, (opt, name)
    finally:
        eng.close()


def test_ebsc_float32_full_size():
    """BASELINE configs[4] in the float32 mode at its full one-GPU size (N = 200k): the E_q[s] rows are float and
    Wp = Es^T Y runs on the f32 matrix cores as a grouped split-K (16 tiles x 32 chunks; partial tiles in f32 over
    6 250 rows each, summed in double).  With Theta fixed the E-step only raises F, the passes are idempotent, the
    grouped and the stream-K contraction agree to float32 accumulation accuracy, and the double sums (Wq, pies, sigma)
    do not depend on the contraction's form."""
    cfg, eng, model, theta, suff, my_data = _setup("c5", dtype=np.float32)
    try:
        for _ in range(2):
            F, nu, nsub, theta = model.step(theta, suff, my_data)
            assert np.isfinite(F)
        Fe = [model.E_step(theta, suff, my_data)[0] for _ in range(2)]
        assert Fe[1] >= Fe[0] - 1e-9 * abs(Fe[0]), Fe
        eng.lpj_resident()
        l1 = eng.download_lpj()
        eng.lpj_resident()
        assert np.array_equal(l1, eng.download_lpj())
        v1 = {k: np.array(v) for k, v in eng.acc_views(eng.stats()).items() if k in ("Wp", "Wq", "pies", "sigma")}
        eng.set_option("gemm_grouped", 0)
        try:
            v2 = eng.acc_views(eng.stats())
            assert _rel(v2["Wp"], v1["Wp"]) <= 5e-6          # two float32 summation orders
            for name in ("Wq", "pies", "sigma"):              # double sums: untouched by the contraction's form
                assert _rel(v2[name], v1[name]) <= 1e-10, name
        finally:
            eng.set_option("gemm_grouped", 1)
        assert np.array_equal(np.diag(v1["Wq"]), v1["pies"]) and np.array_equal(v1["Wq"], v1["Wq"].T)
    finally:
        eng.close()


def _unpack_rows(st, H):
    return np.unpackbits(st, axis=-1)[..., :H].astype(bool)


def test_es3c_dense_states_full_size():
    """SURVEY 8d's dense-state stress variant (K^n initialised with p_init_Kn = 8 / H: 97 % of the states above two
    active latents, more than half above four) at the north-star shape, N = 100k: the census lists and the
    four-lanes-per-state kernels carry nearly every state.  Idempotence, the checksums of the first / second moments
    from the raw K^n and lpj, E[s s^T] symmetric with E[s] on the diagonal, the census against host popcounts, and --
    path independence -- the rows of a sample of datapoints against evoamd_lpj_single, which runs the round-2
    register / wavefront kernels (LU with row exchanges) on the same states."""
    cfg, eng, model, theta, suff, my_data = _setup("c4", dense=True)
    try:
        N, S, H = cfg["N"], cfg["S"], cfg["H"]
        F0, _, _, theta = model.step(theta, suff, my_data)
        Fe = [model.E_step(theta, suff, my_data)[0] for _ in range(2)]
        assert np.isfinite([F0] + Fe).all() and Fe[1] >= Fe[0] - 1e-12 * abs(Fe[0])
        v1 = eng.acc_views(model.last_acc.copy())
        eng.lpj_resident()
        l1 = eng.download_lpj()
        eng.lpj_resident()
        assert np.array_equal(l1, eng.download_lpj())
        v2 = eng.acc_views(eng.stats())
        for name in ("xpt_s", "xpt_ss", "xpt_sz", "xpt_szsz", "Wp", "s_sz_outer", "sz_sz_outer"):
            assert _rel(v2[name], v1[name]) <= 1e-11, name
        lpj, q, k, Fs = _weights_and_counts(eng, cfg)
        assert (k > 2).mean() > 0.5 and (k > 4).mean() > 0.2, ((k > 2).mean(), (k > 4).mean())  # the regime under test
        assert abs(Fs - float(v2["Fs"])) <= 1e-11 * abs(Fs)
        s1 = float((q * k).sum())
        s2 = float((q * (k * (k - 1.0))).sum())
        xs, xss = v2["xpt_s"], v2["xpt_ss"]
        assert abs(float(xs.sum()) - s1) <= 1e-10 * s1
        assert np.array_equal(np.diag(xss), xs) and np.array_equal(xss, xss.T)
        assert abs(float(xss.sum() - np.trace(xss)) - s2) <= 1e-10 * max(s2, 1.0)
        # independent kernels on the same states: per-datapoint operator (TAG 2 chain: K = 4 / K = 8 register kernels)
        Y = my_data["y"]
        for n in (0, 1, N // 2, N - 1):
            st = _unpack_rows(eng.download_states_packed(n, 1)[0], H)
            ref, flags = eng.lpj_single(Y[n], st)
            np.testing.assert_allclose(lpj[n], ref, rtol=1e-11, atol=0)
            assert not flags.any()
    finally:
        eng.close()


def test_bench_configuration_full_size():
    """The exact configuration bench.py times: ES3C north-star shape, rng="device",
    device_mstep=True, prefetched pass over K^n, LazyTheta.  After three EM iterations: the F the step returned equals
    ljc + log-sum-exp of the downloaded rows (the rows of THAT E-step: the prefetched pass of the next one sits in the
    alternate buffer), no duplicate state in a K^n, the overflow census the device reports (n_gt2 / n_gt4 / n_gt8,
    from the census lists) equals host popcounts of the downloaded K^n, the prefetched pass is the one a fresh pass
    would compute, and F does not decrease over further iterations."""
    cfg, eng, model, theta, suff, my_data = _setup("c4", device_mstep=True, lazy_theta=True)
    try:
        N, S, H = cfg["N"], cfg["S"], cfg["H"]
        Fs_seq = []
        for _ in range(3):
            F, nu, nsub, theta = model.step(theta, suff, my_data)
            Fs_seq.append(F)
        assert np.isfinite(Fs_seq).all() and Fs_seq[2] >= Fs_seq[1] >= Fs_seq[0], Fs_seq
        assert not theta.materialised
        lpj, q, k, Fs = _weights_and_counts(eng, cfg)
        d = model.last_dpar
        assert abs((d["ljc_estep"] + Fs / N) - Fs_seq[2]) <= 1e-11 * abs(Fs_seq[2])
        assert (int(d["n_gt2"]), int(d["n_gt4"]), int(d["n_gt8"])) == (int((k > 2).sum()), int((k > 4).sum()), int((k > 8).sum()))
        # the pass prefetched behind the M-step (new Theta, conservative levels) == a fresh pass under the same Theta
        eng.lpj_resident()               # consumes the prefetched pass (buffer swap)
        pre = eng.download_lpj()
        eng.set_option("prefetch_lpj", 1)  # any option drops a pending prefetch: the next call really launches
        eng.lpj_resident()
        np.testing.assert_array_equal(pre, eng.download_lpj())
        for _ in range(2):
            F, nu, nsub, theta = model.step(theta, suff, my_data)
            Fs_seq.append(F)
        assert Fs_seq[4] >= Fs_seq[3] >= Fs_seq[2], Fs_seq
        W = theta["W"]
        assert theta.materialised and W.shape == (cfg["D"], H) and np.isfinite(W).all()
    finally:
        eng.close()
