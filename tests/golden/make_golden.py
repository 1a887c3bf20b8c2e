#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ FROM THE REFERENCE ITSELF.

Runs only in the build container (needs /root/reference); the GPU box and the test-suite
only ever read the resulting ``*.npz`` files.  The reference is imported unmodified with a
single-rank ``mpi4py`` stand-in (``_mpi_standin.py``, our code) whose communicator records
every all-reduce so that the per-rank accumulators -- function locals in the reference --
can be stored.  No reference source text is copied: fixtures hold inputs and outputs only.

    python tests/golden/make_golden.py            # rewrites every fixture

Fixture families
  kat_bars.npz            seed-42 bars run of examples/bars-test (F after 3 EM steps, EBSC+ES3C)
  step_<name>.npz         one or more full ``model.step()`` calls with everything observable
  lpj_<model>.npz         direct ``log_pseudo_joint`` calls on hand-made states (k = 0 ... dense)
  vary_kn.npz             ``vary_Kn`` known answers + random cases
  full_F.npz              ``free_energy(full=True)`` (exact enumeration) on bars data
"""
import hashlib
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, "/root/reference")

import numpy as np  # noqa: E402

import _mpi_standin  # noqa: E402
import _sketch  # noqa: E402

COMM = _mpi_standin.install()

import evo.models._models as ref_models  # noqa: E402
import evo.models.sssc as ref_sssc  # noqa: E402
from evo.models import BSC, SSSC  # noqa: E402
from evo.variational.eas import evolve_states as ref_evolve  # noqa: E402
from evo.variational.utils import init_states, vary_Kn  # noqa: E402


def pack(bits):
    """bool (..., H) -> uint8 (..., ceil(H/8)), h=0 in the MSB of byte 0."""
    return np.packbits(np.asarray(bits, dtype=bool), axis=-1)


def bars(H):
    R = H // 2
    W = np.zeros((R, R, H))
    for i in range(R):
        W[i, :, i] = 1.0
        W[:, i, R + i] = 1.0
    return W.reshape(R * R, H)


# ---- candidate-batch recorder: wrap the evolve_states symbol the models call -------------
TRACE = []


def _recording_evolve(my_suff_stat, model_params, eval_lpj):
    out_states, out_lpj = ref_evolve(my_suff_stat, model_params, eval_lpj)
    TRACE.append((out_states.copy(), out_lpj.copy()))
    return out_states, out_lpj


ref_models.evolve_states = _recording_evolve
ref_sssc.evolve_states = _recording_evolve


def theta_arrays(prefix, theta, keys):
    return {prefix + k: np.array(theta[k], dtype=np.float64) for k in keys}


BSC_KEYS = ("W", "pi", "sigma")
SSSC_KEYS = ("W", "pies", "mus", "Psi", "sigma2")
BSC_SUMS = ("Wp", "Wq", "pies", "sigma")
PERM_ZERO = {"background": False, "allzero": True, "singletons": False}


def run_steps(model, keys, theta, suff, my_data, n_steps, seed0):
    """Run n_steps reference steps, re-seeding np.random before each so a replay can follow."""
    out = {}
    for t in range(n_steps):
        np.random.seed(seed0 + t)
        theta = model.check_params(theta)  # step() does it again (idempotent); lets us save Theta-in
        out.update(theta_arrays("t%d_in_" % t, theta, keys))
        out["t%d_ss_in" % t] = pack(suff["ss"])
        TRACE.clear()
        COMM.log.clear()
        COMM.recording = True
        F, nu, nsub, theta = model.step(theta, suff, my_data)
        COMM.recording = False
        out["t%d_F" % t] = np.float64(F)
        out["t%d_S_nunique" % t] = np.float64(nu)
        out["t%d_S_sub" % t] = np.float64(nsub)
        out["t%d_ss_out" % t] = pack(suff["ss"])
        out["t%d_lpj_out" % t] = suff["lpj"].copy()
        out.update(theta_arrays("t%d_out_" % t, theta, keys))
        # ragged candidate batches
        counts = np.array([s.shape[0] for s, _ in TRACE], dtype=np.int64)
        out["t%d_cand_counts" % t] = counts
        H = suff["ss"].shape[2]
        out["t%d_cand_states" % t] = pack(np.concatenate([s for s, _ in TRACE], axis=0)) if counts.sum() else np.zeros((0, (H + 7) // 8), np.uint8)
        out["t%d_cand_lpj" % t] = np.concatenate([l for _, l in TRACE]) if counts.sum() else np.zeros(0)
        # recorded buffer all-reduces, in call order
        bufs = [p for kind, p in COMM.log if kind == "Allreduce"]
        if isinstance(model, BSC):
            names = ["Wp", "Wq", "pies"]  # bsc.py:230,231,257
            scal = [float(p) for kind, p in COMM.log if kind == "allreduce" and np.ndim(p) == 0]
            # scalar order in BSC.step: N (E_step), N (precompute), S_nunique, S_sub, Fs, N (M_step),
            # 3 reset counters, my_sigma  -> Fs is index 4, my_sigma the last
            out["t%d_sum_Fs" % t] = np.float64(scal[4])
            out["t%d_sum_sigma" % t] = np.float64(scal[-1])
        else:
            # sssc.py:671-674, 677, 682, 691, 763
            names = ["xpt_s", "xpt_ss", "xpt_sz", "xpt_szsz", "s_sz_outer", "sz_sz_outer", "Wp", "y_outer_diag"]
            scal = [float(p) for kind, p in COMM.log if kind == "allreduce" and np.ndim(p) == 0]
            # scalar order in SSSC.EM_step: N, N(precompute), S_nunique, S_sub, Fs, 4 counters
            out["t%d_sum_Fs" % t] = np.float64(scal[4])
        assert len(bufs) == len(names), (len(bufs), names)
        for nm, b in zip(names, bufs):
            out["t%d_sum_%s" % (t, nm)] = b
    return out


def make_step_fixture(name, algo, D, H, S, N, seed, n_steps=2, data="randn", ea=("fit", "randflip", 10, 1, 1),
                      bitflip_prob=None, Mprime=None, p_init=None, use_storage=True, permanent=None,
                      precision=np.float64):
    np.random.seed(seed)
    if algo == "ebsc":
        model = BSC(D, H, S)
        keys = BSC_KEYS
    else:
        model = SSSC(D, H, S, use_storage=use_storage, precision=precision)
        keys = SSSC_KEYS
    if data == "bars":
        gen = {"W": 10.0 * bars(H), "pi": 2.0 / H, "sigma": 1.0}
        Y = BSC(D, H, S).generate_data(gen, N)["y"]
    else:
        Y = np.random.randn(N, D)
    my_data = {"y": Y, "x_infr": np.ones_like(Y, dtype=bool)}
    theta = model.check_params(model.standard_init(my_data))
    suff = init_states(N, S, H, ea[0], ea[1], ea[2], ea[3], ea[4], bitflip_prob, Mprime, p_init, permanent)
    out = {
        "algo": np.array(algo), "D": np.int64(D), "H": np.int64(H), "S": np.int64(S), "N": np.int64(N),
        "seed": np.int64(seed), "n_steps": np.int64(n_steps), "Y": Y, "S_perm": np.int64(suff["S_perm"]),
        "ea_parent_selection": np.array(ea[0]), "ea_mutation": np.array(ea[1]),
        "ea_n_parents": np.int64(suff["n_parents"]), "ea_n_children": np.int64(suff["n_children"]),
        "ea_n_generations": np.int64(suff["n_generations"]),
        "ea_bitflip_prob": np.float64(np.nan if bitflip_prob is None else bitflip_prob),
        "ea_Mprime": np.int64(suff["Mprime"]), "use_storage": np.bool_(use_storage),
        "precision32": np.bool_(np.dtype(precision) == np.float32),
        "background": np.bool_(bool(permanent) and bool(permanent["background"])),
    }
    out.update(run_steps(model, keys, theta, suff, my_data, n_steps, seed0=1000 + seed))
    path = os.path.join(HERE, "step_%s.npz" % name)
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def make_kat_bars():
    """BASELINE.md section 2 determinism check: seed 42, H=10, D=25, N=500, S=32, 3 EM steps."""
    out = {}
    for algo in ("ebsc", "es3c"):
        np.random.seed(42)
        H, D, N, S = 10, 25, 500, 32
        if algo == "ebsc":
            model = BSC(D, H, S)
            gen = {"W": 10.0 * bars(H), "pi": 2.0 / H, "sigma": 1.0}
        else:
            model = SSSC(D, H, S)
            gen = {"W": 10.0 * bars(H), "pies": np.ones(H) * 2.0 / H, "sigma2": np.array(1.0),
                   "mus": np.ones(H) * 0.0, "Psi": np.eye(H) * 1.0}
        Y = model.generate_data(gen, N)["y"]
        model.check_params(gen)
        my_data = {"y": Y, "x_infr": np.ones_like(Y, dtype=bool)}
        theta = model.check_params(model.standard_init(my_data))
        suff = init_states(N, S, H, "fit", "randflip", 10, 1, 1)
        Fs = []
        for _ in range(3):
            F, nu, nsub, theta = model.step(theta, suff, my_data)
            Fs.append(F)
        out[algo + "_F"] = np.array(Fs)
        out[algo + "_ss_sha1"] = np.array(hashlib.sha1(pack(suff["ss"]).tobytes()).hexdigest())
        out[algo + "_Y_sha1"] = np.array(hashlib.sha1(Y.tobytes()).hexdigest())
        print(algo, Fs)
    np.savez_compressed(os.path.join(HERE, "kat_bars.npz"), **out)


def learned_like_sssc_theta(D, H, rng):
    """A Theta that looks like one after a few M-steps: dense NON-symmetric Psi."""
    A = rng.normal(size=(H, H)) * 0.15
    Psi = np.eye(H) + A @ A.T * 0.5 + rng.normal(size=(H, H)) * 0.03
    return {"W": rng.normal(size=(D, H)), "pies": rng.uniform(0.05, 0.4, H), "mus": rng.normal(size=H),
            "Psi": Psi, "sigma2": np.float64(0.7)}


def make_lpj_fixtures():
    rng = np.random.RandomState(7)
    # --- BSC
    D, H, C = 20, 70, 40
    model = BSC(D, H, C)
    theta = {"W": rng.normal(size=(D, H)), "pi": 0.07, "sigma": 1.3}
    states = rng.random_sample((C, H)) < rng.uniform(0, 0.3, size=(C, 1))
    states[0] = False  # k = 0
    states[1] = True   # k = H
    Y = rng.normal(size=(3, D))
    my_data = {"y": Y, "x_infr": np.ones_like(Y, dtype=bool)}
    suff = {}
    model.E_step_precompute(theta, suff, my_data)
    lpj = np.zeros((3, C))
    for n in range(3):
        my_data["this_y"] = Y[n]
        my_data["this_x_infr"] = my_data["x_infr"][n]
        suff["this_states"] = states
        lpj[n] = model.log_pseudo_joint(theta, suff, my_data)
    np.savez_compressed(os.path.join(HERE, "lpj_bsc.npz"), W=theta["W"], pi=np.float64(theta["pi"]),
                        sigma=np.float64(theta["sigma"]), states=pack(states), H=np.int64(H), Y=Y, lpj=lpj,
                        ljc=np.float64(theta["ljc"]))
    # --- SSSC, learned-like Theta, k from 0 to 24
    D, H, C = 24, 70, 48
    model = SSSC(D, H, C, use_storage=False)
    theta = learned_like_sssc_theta(D, H, rng)
    ks = [0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 6, 7, 8, 9, 12, 16, 20, 24] + list(rng.randint(0, 7, size=C - 18))
    states = np.zeros((C, H), dtype=bool)
    for c, k in enumerate(ks):
        states[c, rng.choice(H, size=k, replace=False)] = True
    Y = rng.normal(size=(3, D))
    my_data = {"y": Y, "x_infr": np.ones_like(Y, dtype=bool)}
    suff = {}
    model.E_step_precompute(theta, suff, my_data)
    lpj = np.zeros((3, C))
    for n in range(3):
        my_data["this_y"] = Y[n]
        my_data["this_x_infr"] = my_data["x_infr"][n]
        suff["this_states"] = states
        lpj[n] = model.log_pseudo_joint(theta, suff, my_data)
        suff["storage"] = {"storagekeys": (), "counts_norm": 0, "counts": 0}
    np.savez_compressed(os.path.join(HERE, "lpj_sssc.npz"), H=np.int64(H), states=pack(states), Y=Y, lpj=lpj,
                        ljc=np.float64(theta["ljc"]), **theta_arrays("", theta, SSSC_KEYS))
    # --- SSSC, exactly singular Psi_s (sssc.py:278-301: pinv branches; lpj = +inf -> B_max).  Direct operator calls, so
    # check_params (which floors the diagonal of Psi) is not in the way of a zero variance.  k <= 2 only: those are the
    # states evo_amd serves the reference's way; per state the lambda_s / kappa the reference's statistics use
    D, H = 16, 14
    model = SSSC(D, H, 20, use_storage=True)
    theta = learned_like_sssc_theta(D, H, rng)
    Psi = theta["Psi"]
    for a, b in ((2, 5),):                    # two equal rows / columns inside every A that holds both
        Psi[a, a] = Psi[b, b] = Psi[a, b] = Psi[b, a] = 1.0
    for h in (7, 9, 11):                      # zero variance, uncorrelated with everything
        Psi[h, :] = 0.0
        Psi[:, h] = 0.0
    Psi[3, 8] = 2.0 * Psi[3, 3]               # rows in a power-of-two ratio: [[p, 2p], [p/2... ]] made exact below
    Psi[8, 3] = 0.5 * Psi[3, 3]
    Psi[8, 8] = Psi[3, 3]                     # [[p, 2p], [p/2, p]]: row 1 = row 0 / 2 exactly
    sets = [(), (2,), (5,), (2, 5), (7,), (2, 7), (7, 12), (9, 11), (9,), (3, 8), (3,), (8,), (1, 4), (0, 13), (6, 10),
            (1,), (12,), (4, 6), (10, 13), (0,)]
    states = np.zeros((len(sets), H), dtype=bool)
    for c, on in enumerate(sets):
        states[c, list(on)] = True
    Y = rng.normal(size=(3, D))
    my_data = {"y": Y, "x_infr": np.ones_like(Y, dtype=bool)}
    suff = {}
    model.E_step_precompute(theta, suff, my_data)
    lpj = np.zeros((3, len(sets)))
    lam = np.zeros((len(sets), 2, 2))
    kappa = np.zeros((3, len(sets), 2))
    cnt = np.zeros((3, 3), dtype=np.int64)
    for n in range(3):
        my_data["this_y"] = Y[n]
        my_data["this_x_infr"] = my_data["x_infr"][n]
        suff["this_states"] = states
        for key in ("reset_lpj_isnan", "reset_lpj_smaller_eps_lpj", "reset_lpj_isinf"):
            suff[key] = 0
        with np.errstate(all="ignore"):
            lpj[n] = model.log_pseudo_joint(theta, suff, my_data)
        cnt[n] = [suff["reset_lpj_isnan"], suff["reset_lpj_smaller_eps_lpj"], suff["reset_lpj_isinf"]]
        for c, on in enumerate(sets):            # what the statistics loop reads from `storage` (sssc.py:566-575)
            if not on:
                continue
            ent = suff["storage"][str((model.s_ids * states[c]).sum())]
            k = len(on)
            lam[c, :k, :k] = ent["lambda_s"]
            kappa[n, c, :k] = np.dot(ent["lambda_s_W_s_sigma2_inv"], Y[n] - ent["W_s_mus_s"]) + theta["mus"][states[c]]
    np.savez_compressed(os.path.join(HERE, "lpj_sssc_singular.npz"), H=np.int64(H), states=pack(states), Y=Y, lpj=lpj,
                        lam=lam, kappa=kappa, reset_counts=cnt, psi_s_pinv=np.int64(suff["Psi_s_pinv"]),
                        ljc=np.float64(theta["ljc"]), **theta_arrays("", theta, SSSC_KEYS))
    # --- clamp behaviour (_models.py:567-596)
    model = BSC(4, 4, 4)
    cases = [np.array([1.0, np.nan, -np.inf, np.inf]), np.array([-np.inf, 2.0, -3.0]), np.array([np.inf, 1.0]),
             np.array([0.5, -0.5])]
    res = {}
    for i, c in enumerate(cases):
        cnt = {"reset_lpj_isnan": 0, "reset_lpj_smaller_eps_lpj": 0, "reset_lpj_isinf": 0}
        res["in%d" % i] = c.copy()
        with np.errstate(all="ignore"):
            res["out%d" % i] = model.lpj_reset_check(c.copy(), cnt)
        res["cnt%d" % i] = np.array([cnt["reset_lpj_isnan"], cnt["reset_lpj_smaller_eps_lpj"], cnt["reset_lpj_isinf"]])
    np.savez_compressed(os.path.join(HERE, "lpj_clamp.npz"), **res)


def make_lpj_singular_k3():
    """SSSC states with THREE OR MORE active latents whose Psi_A is exactly singular (sssc.py:278-301), round 3: equal
    rows / columns, zero variances, a rank-2 3 x 3 block whose 2 x 2 minors are all regular, and one set whose
    M_A = G_A / sigma2 + pinv(Psi_A) is exactly singular as well (Psi_A = 0 and two equal columns of W: pinv(M_A))."""
    rng = np.random.RandomState(77)
    D, H = 16, 16
    model = SSSC(D, H, 24, use_storage=True)
    theta = learned_like_sssc_theta(D, H, rng)
    Psi, W = theta["Psi"], theta["W"]
    Psi[5, :] = Psi[2, :]                     # latent 5 a copy of latent 2 (row, then column: Psi[5,5] = Psi[2,2])
    Psi[:, 5] = Psi[:, 2]
    for h in (7, 9, 10):                      # zero variance, uncorrelated with everything
        Psi[h, :] = 0.0
        Psi[:, h] = 0.0
    W[:, 9] = W[:, 7]                         # G_A of a set holding 7 and 9 has two equal rows / columns
    blk = (0, 1, 3)                           # V V^T, V = [[1,0],[0,1],[1,1]]: rank 2, every 2 x 2 minor equals 1
    for h in blk:
        Psi[h, :] = 0.0
        Psi[:, h] = 0.0
    Psi[np.ix_(blk, blk)] = np.array([[1.0, 0.0, 1.0], [0.0, 1.0, 1.0], [1.0, 1.0, 2.0]])
    sets = [(2, 5, 12), (1, 4, 7), (0, 1, 3), (0, 1, 3, 6), (7, 9, 10), (4, 6, 8), (4, 8, 12, 13), (4, 6, 8, 11, 13),
            (4, 6, 8, 10, 12, 14), (0, 1, 3, 4, 6, 8, 12, 13, 2), (4, 6, 8, 11, 12, 13, 14, 15, 2), (0, 1), (1, 3), (2, 5),
            (7,), (4,), (), (6, 11, 14), (2, 4, 6, 8, 11, 12, 13, 14, 15, 1), (0, 3, 4), (5, 11, 15), (7, 9),
            (2, 6, 11, 12, 13, 14, 15), (7, 9, 11, 12)]
    states = np.zeros((len(sets), H), dtype=bool)
    for c, on in enumerate(sets):
        states[c, list(on)] = True
    KM = max(len(on) for on in sets)
    N = 3
    Y = rng.normal(size=(N, D))
    my_data = {"y": Y, "x_infr": np.ones_like(Y, dtype=bool)}
    suff = {}
    model.E_step_precompute(theta, suff, my_data)
    lpj = np.zeros((N, len(sets)))
    lam = np.zeros((len(sets), KM, KM))
    kappa = np.zeros((N, len(sets), KM))
    cnt = np.zeros((N, 3), dtype=np.int64)
    for n in range(N):
        my_data["this_y"] = Y[n]
        my_data["this_x_infr"] = my_data["x_infr"][n]
        suff["this_states"] = states
        for key in ("reset_lpj_isnan", "reset_lpj_smaller_eps_lpj", "reset_lpj_isinf"):
            suff[key] = 0
        with np.errstate(all="ignore"):
            lpj[n] = model.log_pseudo_joint(theta, suff, my_data)
        cnt[n] = [suff["reset_lpj_isnan"], suff["reset_lpj_smaller_eps_lpj"], suff["reset_lpj_isinf"]]
        for c, on in enumerate(sets):            # what the statistics loop reads from `storage` (sssc.py:566-575)
            if not on:
                continue
            ent = suff["storage"][str((model.s_ids * states[c]).sum())]
            k = len(on)
            lam[c, :k, :k] = ent["lambda_s"]
            kappa[n, c, :k] = np.dot(ent["lambda_s_W_s_sigma2_inv"], Y[n] - ent["W_s_mus_s"]) + theta["mus"][states[c]]
    print("singular k>=3 fixture: %d of %d states at B_max, Psi_s_pinv = %d" % ((lpj[0] == 0.0).sum(), len(sets), suff["Psi_s_pinv"]))
    np.savez_compressed(os.path.join(HERE, "lpj_sssc_singular_k3.npz"), H=np.int64(H), states=pack(states), Y=Y, lpj=lpj,
                        lam=lam, kappa=kappa, reset_counts=cnt, psi_s_pinv=np.int64(suff["Psi_s_pinv"]),
                        ljc=np.float64(theta["ljc"]), **theta_arrays("", theta, SSSC_KEYS))


def _operator_fixture(path, model, theta, sets, H, Y, extra=None):
    """log_pseudo_joint on hand-made states, with lambda_s / kappa_s of every state as the statistics loop reads them."""
    states = np.zeros((len(sets), H), dtype=bool)
    for c, on in enumerate(sets):
        states[c, list(on)] = True
    KM = max(len(on) for on in sets)
    N = Y.shape[0]
    my_data = {"y": Y, "x_infr": np.ones_like(Y, dtype=bool)}
    suff = {}
    model.E_step_precompute(theta, suff, my_data)
    lpj = np.zeros((N, len(sets)))
    lam = np.zeros((len(sets), KM, KM))
    kappa = np.zeros((N, len(sets), KM))
    cnt = np.zeros((N, 3), dtype=np.int64)
    for n in range(N):
        my_data["this_y"] = Y[n]
        my_data["this_x_infr"] = my_data["x_infr"][n]
        suff["this_states"] = states
        for key in ("reset_lpj_isnan", "reset_lpj_smaller_eps_lpj", "reset_lpj_isinf"):
            suff[key] = 0
        with np.errstate(all="ignore"):
            lpj[n] = model.log_pseudo_joint(theta, suff, my_data)
        cnt[n] = [suff["reset_lpj_isnan"], suff["reset_lpj_smaller_eps_lpj"], suff["reset_lpj_isinf"]]
        for c, on in enumerate(sets):            # what the statistics loop reads from `storage` (sssc.py:566-575)
            if not on:
                continue
            ent = suff["storage"][str((model.s_ids * states[c]).sum())]
            k = len(on)
            lam[c, :k, :k] = ent["lambda_s"]
            kappa[n, c, :k] = np.dot(ent["lambda_s_W_s_sigma2_inv"], Y[n] - ent["W_s_mus_s"]) + theta["mus"][states[c]]
    np.savez_compressed(path, H=np.int64(H), states=pack(states), Y=Y, lpj=lpj, lam=lam, kappa=kappa, reset_counts=cnt,
                        psi_s_pinv=np.int64(suff["Psi_s_pinv"]), ljc=np.float64(theta["ljc"]),
                        **theta_arrays("", theta, SSSC_KEYS), **(extra or {}))
    return lpj, suff


def make_lpj_indefinite():
    """Round 4: the remaining branch of sssc.py:295-300 -- a REGULAR Psi_s whose M_s = W_s^T W_s / sigma2 + inv(Psi_s) is
    exactly singular (inv raises, the reference goes on with pinv(M_s) and slogdet(M_s) = -inf, lpj = +inf -> B_max) --
    inside an INDEFINITE Psi (mixed-sign spectrum, T = I + Psi_A G_A / sigma2 regular but not positive).  sigma2 = 1/2 and
    planted latents whose arithmetic is exact in binary:
      3        W = e_0, Psi_33 = -1/2, uncorrelated:  M = 2 - 2 = 0 for every set that holds it (a decoupled zero row);
      (6, 11)  W = (e_1, e_1 + e_2), Psi block [[-1, 1/2], [1/2, -1/2]] -> inv = -[[2, 2], [2, 4]] = -G / sigma2: M = 0;
      (8, 15)  W = (e_3, e_3 + e_4), Psi block [[-1/4, 1/2], [1/2, -1/2]] -> inv = [[4, 4], [4, 2]]: M = [[6, 6], [6, 6]],
               rank one, first pivot non-zero.
    Each of 6, 11, 8, 15 alone has a regular M.  The other latents share the rows 5.. of W and a dense indefinite Psi."""
    rng = np.random.RandomState(404)
    D, H = 18, 20
    model = SSSC(D, H, 40, use_storage=True)
    theta = learned_like_sssc_theta(D, H, rng)
    theta["sigma2"] = np.float64(0.5)
    W, Psi = theta["W"], theta["Psi"]
    Q = np.linalg.qr(rng.normal(size=(H, H)))[0]
    lamb = np.concatenate([rng.uniform(0.4, 1.6, H - 7), -rng.uniform(0.3, 1.2, 7)])
    Psi[:] = (Q * lamb) @ Q.T + rng.normal(size=(H, H)) * 0.02   # indefinite, slightly non-symmetric
    planted = (3, 6, 11, 8, 15)
    W[:5, :] = 0.0
    for h in planted:
        W[:, h] = 0.0
        Psi[h, :] = 0.0
        Psi[:, h] = 0.0
    W[0, 3] = 1.0
    Psi[3, 3] = -0.5
    W[1, 6] = 1.0
    W[1, 11] = W[2, 11] = 1.0
    Psi[6, 6], Psi[6, 11], Psi[11, 6], Psi[11, 11] = -1.0, 0.5, 0.5, -0.5
    W[3, 8] = 1.0
    W[3, 15] = W[4, 15] = 1.0
    Psi[8, 8], Psi[8, 15], Psi[15, 8], Psi[15, 15] = -0.25, 0.5, 0.5, -0.5
    sets = [(), (3,), (6,), (11,), (8,), (15,), (6, 11), (8, 15), (3, 6), (0, 1), (2, 4), (5, 19), (0,), (7,),
            (0, 3), (3, 8, 15), (6, 11, 0), (8, 15, 1, 2), (0, 1, 2), (1, 2, 4, 5), (6, 8, 0), (11, 15, 7, 9),
            (0, 1, 2, 4, 5), (3, 0, 1, 2, 4, 5), (6, 11, 0, 1, 2, 4, 5, 7), (0, 1, 2, 4, 5, 7, 9, 10),
            (8, 15, 0, 1, 2, 4, 5, 7, 9), (0, 1, 2, 4, 5, 7, 9, 10, 12, 13), (3, 6, 11, 8, 15),
            (0, 1, 2, 4, 5, 7, 9, 10, 12, 13, 14, 16), (3, 0, 1, 2, 4, 5, 7, 9, 10, 12, 13, 14, 16, 17),
            (6, 8, 11, 0, 1, 2, 4, 5, 7, 9, 10, 12), tuple(range(H)), (6, 15), (8, 11)]
    Y = rng.normal(size=(3, D))
    lpj, suff = _operator_fixture(os.path.join(HERE, "lpj_sssc_indefinite.npz"), model, theta, sets, H, Y)
    print("indefinite Psi fixture: %d of %d states at B_max, Psi_s_pinv = %d, eig(Psi_sym) in [%.2f, %.2f]"
          % ((lpj[0] == 0.0).sum(), len(sets), suff["Psi_s_pinv"], lamb.min(), lamb.max()))


def make_lpj_dense():
    """Round 4: states with MORE than 64 active latents (H = 150, k up to 150) beside sparse ones -- the reference's loop
    has no limit on |s| (sssc.py:261-324)."""
    rng = np.random.RandomState(405)
    D, H = 40, 150
    model = SSSC(D, H, 24, use_storage=True)
    theta = learned_like_sssc_theta(D, H, rng)
    sets = []
    for k in (0, 1, 2, 3, 7, 12, 40, 63, 64, 65, 66, 80, 97, 128, 129, 149, 150):
        sets.append(tuple(sorted(rng.choice(H, size=k, replace=False).tolist())))
    sets += [tuple(range(0, 130)), tuple(range(20, 150)), tuple(range(0, 150, 2))]
    Y = rng.normal(size=(2, D))
    lpj, suff = _operator_fixture(os.path.join(HERE, "lpj_sssc_dense.npz"), model, theta, sets, H, Y)
    print("dense-state fixture: k = %s, lpj in [%.1f, %.1f]" % ([len(x) for x in sets], lpj.min(), lpj.max()))


PERM_BG = {"background": True, "allzero": False, "singletons": False}


def make_background_fixtures():
    """Round 4: the permanent background unit (latent H-1 on in every state: variational/utils.py:42-47,96-98,
    eas.py:213-239, bsc.py:259, sssc.py:718) and exact E-steps (S == 2^H_, variational/utils.py:55,71-88) --
    init_states outputs with their np.random stream, full steps, and the exact log-likelihood with a background unit."""
    out = {}
    cases = [("bg", 5, 8, 7, dict(PERM_BG), 0.25), ("exact", 3, 16, 4, None, None), ("exact_bg", 3, 16, 5, dict(PERM_BG), None),
             ("exact_zero", 3, 16, 4, dict(PERM_ZERO), None), ("bg_zero_ignored", 4, 6, 9,
                                                               {"background": True, "allzero": True, "singletons": False}, None)]
    for nm, N, S, H, perm, p0 in cases:
        np.random.seed(31)
        suff = init_states(N, S, H, "fit", "randflip", 3, 2, 1, None, None, p0, perm)
        out[nm + "_N"], out[nm + "_S"], out[nm + "_H"] = np.int64(N), np.int64(S), np.int64(H)
        out[nm + "_p0"] = np.float64(np.nan if p0 is None else p0)
        out[nm + "_perm"] = np.array([bool(perm and perm[k]) for k in ("background", "allzero", "singletons")])
        out[nm + "_ss"] = suff["ss"].copy()
        out[nm + "_lpj_shape"] = np.array(suff["lpj"].shape, dtype=np.int64)
        out[nm + "_S_perm"] = np.int64(suff["S_perm"])
        out[nm + "_incl_shape"] = np.array(suff["incl"].shape, dtype=np.int64)
        out[nm + "_sm"] = suff["sm"].copy() if suff["sm"] is not None else np.zeros((0, 0), dtype=bool)
        out[nm + "_next_random"] = np.float64(np.random.random())  # where the stream stands afterwards
    # exact log-likelihood with a background unit (_models.py:389-390)
    for algo in ("ebsc", "es3c"):
        np.random.seed(12)
        H, D, N, S = 7, 9, 20, 10
        if algo == "ebsc":
            model = BSC(D, H, S)
            gen = {"W": 6.0 * np.random.randn(D, H), "pi": 2.0 / H, "sigma": 1.0}
            keys = BSC_KEYS
        else:
            model = SSSC(D, H, S)
            gen = {"W": 6.0 * np.random.randn(D, H), "pies": np.ones(H) * 2.0 / H, "sigma2": np.array(1.0),
                   "mus": np.ones(H) * 0.3, "Psi": np.eye(H) * 1.0}
            keys = SSSC_KEYS
        Y = model.generate_data(gen, N)["y"]
        gen = model.check_params(gen)
        my_data = {"y": Y, "x_infr": np.ones_like(Y, dtype=bool)}
        suff = init_states(N, S, H, "fit", "randflip", 5, 1, 1, permanent=dict(PERM_BG))
        L = model.free_energy(my_data, dict(gen), suff, full=True)
        out["full_" + algo + "_Y"] = Y
        out["full_" + algo + "_L"] = np.float64(L)
        out.update(theta_arrays("full_" + algo + "_", gen, keys))
        print(algo, "L with background unit", L)
    np.savez_compressed(os.path.join(HERE, "background.npz"), **out)
    make_step_fixture("ebsc_bg", "ebsc", 20, 24, 12, 30, seed=81, n_steps=2, ea=("fit", "randflip", 4, 2, 1), permanent=dict(PERM_BG))
    make_step_fixture("es3c_bg", "es3c", 20, 24, 12, 30, seed=82, n_steps=2, ea=("fit", "randflip", 4, 2, 1), permanent=dict(PERM_BG))
    make_step_fixture("es3c_bg_cross", "es3c", 16, 12, 10, 24, seed=83, n_steps=2, ea=("fit", "cross_randflip", 4, 1, 2),
                      permanent=dict(PERM_BG))
    make_step_fixture("ebsc_exact", "ebsc", 12, 5, 32, 20, seed=84, n_steps=2, ea=("fit", "randflip", 4, 2, 1))
    make_step_fixture("es3c_exact_bg", "es3c", 12, 6, 32, 20, seed=85, n_steps=2, ea=("fit", "randflip", 4, 2, 1),
                      permanent=dict(PERM_BG))


def make_vary_kn():
    out = {}
    cases = []
    # SURVEY 8c(i) known answers
    H, S = 6, 4
    old = np.eye(H, dtype=bool)[:3]
    old = np.concatenate((old, np.array([[1, 1, 0, 0, 0, 0]], dtype=bool)))
    new = np.concatenate((np.eye(H, dtype=bool)[3:], np.eye(H, dtype=bool)[:1]))
    for Mp in (4, 1):
        cases.append((H, S, Mp, old.copy(), np.array([-5.0, -1.0, -9.0, -3.0]), new.copy(), np.array([-2.0, -20.0, -0.5, -5.0])))
    cases.append((H, S, 4, old.copy(), np.array([-5.0, -1.0, -9.0, -3.0]), np.zeros((0, H), dtype=bool), np.zeros(0)))
    rng = np.random.RandomState(3)
    for (H, S, C, Mp) in [(9, 6, 5, 6), (70, 12, 10, 12), (70, 12, 10, 3), (130, 40, 30, 40), (33, 8, 20, 8)]:
        for rep in range(3):
            o = np.zeros((0, H), dtype=bool)
            while o.shape[0] < S:
                o = np.unique(np.concatenate((o, rng.random_sample((S, H)) < 2.0 / H)), axis=0)
            o = o[rng.permutation(o.shape[0])[:S]]
            nw = rng.random_sample((C, H)) < 2.0 / H
            nw[0] = o[1]  # duplicate of an old state
            if C > 3:
                nw[3] = nw[2]  # duplicate inside the new batch
            cases.append((H, S, Mp, o, rng.normal(size=S) * 5, nw, rng.normal(size=C) * 5))
    out["n_cases"] = np.int64(len(cases))
    for i, (H, S, Mp, o, lo, nw, ln) in enumerate(cases):
        states = o.copy()
        lpj_old = lo.copy()
        lpj_out = np.zeros(S)
        a, b = vary_Kn(lpj_old, ln.copy(), lpj_out, states, nw.copy(), H, S, 0, np.zeros((0, H), dtype=bool), Mp)
        out.update({"c%d_H" % i: np.int64(H), "c%d_S" % i: np.int64(S), "c%d_Mprime" % i: np.int64(Mp),
                    "c%d_old" % i: pack(o), "c%d_lpj_old" % i: lo, "c%d_new" % i: pack(nw), "c%d_lpj_new" % i: ln,
                    "c%d_states_out" % i: pack(states), "c%d_lpj_out" % i: lpj_out,
                    "c%d_ret" % i: np.array([a, b], dtype=np.int64)})
    np.savez_compressed(os.path.join(HERE, "vary_kn.npz"), **out)
    print("vary_kn cases:", len(cases))


def make_full_F():
    out = {}
    for algo in ("ebsc", "es3c"):
        np.random.seed(11)
        H, D, N, S = 8, 16, 30, 10
        if algo == "ebsc":
            model = BSC(D, H, S)
            gen = {"W": 10.0 * bars(H), "pi": 2.0 / H, "sigma": 1.0}
            keys = BSC_KEYS
        else:
            model = SSSC(D, H, S)
            gen = {"W": 10.0 * bars(H), "pies": np.ones(H) * 2.0 / H, "sigma2": np.array(1.0),
                   "mus": np.ones(H) * 0.3, "Psi": np.eye(H) * 1.0}
            keys = SSSC_KEYS
        Y = model.generate_data(gen, N)["y"]
        gen = model.check_params(gen)
        my_data = {"y": Y, "x_infr": np.ones_like(Y, dtype=bool)}
        suff = init_states(N, S, H, "fit", "randflip", 5, 1, 1)
        L = model.free_energy(my_data, dict(gen), suff, full=True)
        out[algo + "_Y"] = Y
        out[algo + "_L"] = np.float64(L)
        out.update(theta_arrays(algo + "_", gen, keys))
        print(algo, "L_gen", L)
    np.savez_compressed(os.path.join(HERE, "full_F.npz"), **out)


def make_learn_bars(n_seeds=6, n_iter=40):
    """Learning-level fixture (the reference validates by TRAINING -- bars recovery and F -> L_gen,
    examples/bars-test/main.py:123-135, 156-162).  The bars set-up of SURVEY 8c(ii) (H = 10, D = 25, N = 500, S = 32,
    fit / randflip 10 x 1 x 1), EBSC and ES3C, `n_seeds` seeds, `n_iter` EM iterations each, run by the reference itself:
    per seed the F trajectory, the exact log-likelihood under the generating Theta (free_energy(full=True)), the learned W
    and how many bars it recovered.  The GPU test trains the same data sets in the configuration bench.py times
    (rng="device", device M-step) and compares distributions; it needs no reference run."""
    H, D, N, S = 10, 25, 500, 32
    out = {"n_seeds": np.int64(n_seeds), "n_iter": np.int64(n_iter), "H": np.int64(H), "D": np.int64(D), "N": np.int64(N),
           "S": np.int64(S)}
    Wg = 10.0 * bars(H)
    for algo in ("ebsc", "es3c"):
        Fs, Ls, Ws, rec, shas = [], [], [], [], []
        for seed in range(n_seeds):
            np.random.seed(1000 + seed)
            if algo == "ebsc":
                model = BSC(D, H, S)
                gen = {"W": Wg.copy(), "pi": 2.0 / H, "sigma": 1.0}
            else:
                model = SSSC(D, H, S)
                gen = {"W": Wg.copy(), "pies": np.ones(H) * 2.0 / H, "sigma2": np.array(1.0),
                       "mus": np.ones(H) * 0.0, "Psi": np.eye(H) * 1.0}
            Y = model.generate_data(gen, N)["y"]
            gen = model.check_params(gen)
            my_data = {"y": Y, "x_infr": np.ones_like(Y, dtype=bool)}
            theta = model.check_params(model.standard_init(my_data))
            suff = init_states(N, S, H, "fit", "randflip", 10, 1, 1)
            L_gen = model.free_energy(my_data, {k: (np.array(v, copy=True) if isinstance(v, np.ndarray) else v) for k, v in gen.items()},
                                      suff, full=True)
            F_tr = []
            for _ in range(n_iter):
                F, nu, nsub, theta = model.step(theta, suff, my_data)
                F_tr.append(F)
            Fs.append(F_tr)
            Ls.append(L_gen)
            Ws.append(np.array(theta["W"]))
            rec.append(_sketch.bars_recovered(theta["W"], Wg))
            shas.append(hashlib.sha1(Y.tobytes()).hexdigest())
            print(algo, "seed", seed, "L_gen %.4f" % L_gen, "F_end %.4f" % F_tr[-1], "bars", rec[-1], flush=True)
        out[algo + "_F"] = np.array(Fs)
        out[algo + "_L_gen"] = np.array(Ls)
        out[algo + "_W"] = np.array(Ws)
        out[algo + "_bars"] = np.array(rec, dtype=np.int64)
        out[algo + "_Y_sha1"] = np.array(shas)
    np.savez_compressed(os.path.join(HERE, "learn_bars.npz"), **out)


def make_recon_fixture(name, algo, D, H, S, N, seed, n_steps=2):
    """model.step(..., do_reconstruction=True) on complete data (the image-denoising use,
    examples/image-denoising/main.py:100-110,162-169): my_data["x"] marks the entries that keep their
    value; every other entry of y_reconstructed is the posterior-predictive estimate."""
    np.random.seed(seed)
    model = BSC(D, H, S) if algo == "ebsc" else SSSC(D, H, S)
    keys = BSC_KEYS if algo == "ebsc" else SSSC_KEYS
    gen = {"W": 10.0 * bars(H), "pi": 2.0 / H, "sigma": 1.0}
    Y = BSC(D, H, S).generate_data(gen, N)["y"]
    x = np.random.random_sample(Y.shape) < 0.4
    x[0] = False  # one datapoint reconstructed completely
    x[1] = True   # one not at all
    my_data = {"y": Y, "x_infr": np.ones_like(Y, dtype=bool), "x": x}
    theta = model.check_params(model.standard_init(my_data))
    suff = init_states(N, S, H, "fit", "randflip", 5, 1, 1)
    out = {"algo": np.array(algo), "D": np.int64(D), "H": np.int64(H), "S": np.int64(S), "N": np.int64(N),
           "seed": np.int64(seed), "n_steps": np.int64(n_steps), "Y": Y, "x": x,
           "ea_parent_selection": np.array("fit"), "ea_mutation": np.array("randflip"),
           "ea_n_parents": np.int64(5), "ea_n_children": np.int64(1), "ea_n_generations": np.int64(1),
           "ea_bitflip_prob": np.float64(np.nan), "ea_Mprime": np.int64(suff["Mprime"])}
    out["t0_ss_in"] = pack(suff["ss"])
    out.update(theta_arrays("t0_in_", theta, keys))
    for t in range(n_steps):
        np.random.seed(1000 + seed + t)
        F, nu, nsub, theta = model.step(theta, suff, my_data, do_reconstruction=True)
        out["t%d_F" % t] = np.float64(F)
        out["t%d_y_reconstructed" % t] = my_data["y_reconstructed"].copy()
        out["t%d_ss_out" % t] = pack(suff["ss"])
        out.update(theta_arrays("t%d_out_" % t, theta, keys))
    path = os.path.join(HERE, "recon_%s.npz" % name)
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def make_missing_fixture_es3c(D=25, H=10, S=8, N=36, seed=41, n_steps=2):
    """ES3C on incomplete data (NaN holes, x_infr = x = ~isnan(y)); the reference needs
    use_storage=False (the cached state terms depend on the datapoint's reliable entries) and
    do_reconstruction=True in every step (sssc.py:630-633 reads this_y_rec)."""
    np.random.seed(seed)
    model = SSSC(D, H, S, use_storage=False)
    gen = {"W": 10.0 * bars(H), "pi": 2.0 / H, "sigma": 1.0}
    Y = BSC(D, H, S).generate_data(gen, N)["y"]
    miss = np.random.random_sample(Y.shape) < 0.25
    miss[0] = False
    Y = Y.copy()
    Y[miss] = np.nan
    x_infr = np.logical_not(np.isnan(Y))
    my_data = {"y": Y, "x_infr": x_infr, "x": x_infr.copy()}
    theta = model.check_params(model.standard_init(my_data))
    suff = init_states(N, S, H, "fit", "randflip", 5, 1, 1)
    out = {"algo": np.array("es3c"), "D": np.int64(D), "H": np.int64(H), "S": np.int64(S), "N": np.int64(N),
           "seed": np.int64(seed), "n_steps": np.int64(n_steps), "Y": Y, "x_infr": x_infr,
           "ea_parent_selection": np.array("fit"), "ea_mutation": np.array("randflip"),
           "ea_n_parents": np.int64(5), "ea_n_children": np.int64(1), "ea_n_generations": np.int64(1),
           "ea_bitflip_prob": np.float64(np.nan), "ea_Mprime": np.int64(suff["Mprime"])}
    out["t0_ss_in"] = pack(suff["ss"])
    out.update(theta_arrays("t0_in_", theta, SSSC_KEYS))
    for t in range(n_steps):
        np.random.seed(1000 + seed + t)
        F, nu, nsub, theta = model.step(theta, suff, my_data, do_reconstruction=True)
        out["t%d_F" % t] = np.float64(F)
        out["t%d_y_reconstructed" % t] = my_data["y_reconstructed"].copy()
        out["t%d_ss_out" % t] = pack(suff["ss"])
        out["t%d_lpj_out" % t] = suff["lpj"].copy()
        out.update(theta_arrays("t%d_out_" % t, theta, SSSC_KEYS))
    path = os.path.join(HERE, "missing_es3c.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def make_missing_fixture(name="ebsc", D=25, H=10, S=8, N=40, seed=31, n_steps=3):
    """EBSC on incomplete data, the image-inpainting use (examples/image-inpainting/main.py:105-111):
    y holds NaN where a value is missing, x_infr = x = ~isnan(y); step 0 and 1 reconstruct, step 2
    reuses the y_reconstructed of step 1 for its M-step (bsc.py:184-189)."""
    np.random.seed(seed)
    model = BSC(D, H, S)
    gen = {"W": 10.0 * bars(H), "pi": 2.0 / H, "sigma": 1.0}
    Y = BSC(D, H, S).generate_data(gen, N)["y"]
    miss = np.random.random_sample(Y.shape) < 0.3
    miss[0] = False                 # one complete datapoint
    Y = Y.copy()
    Y[miss] = np.nan
    x_infr = np.logical_not(np.isnan(Y))
    my_data = {"y": Y, "x_infr": x_infr, "x": x_infr.copy()}
    theta = model.check_params(model.standard_init(my_data))
    suff = init_states(N, S, H, "fit", "randflip", 5, 1, 1)
    out = {"algo": np.array("ebsc"), "D": np.int64(D), "H": np.int64(H), "S": np.int64(S), "N": np.int64(N),
           "seed": np.int64(seed), "n_steps": np.int64(n_steps), "Y": Y, "x_infr": x_infr,
           "ea_parent_selection": np.array("fit"), "ea_mutation": np.array("randflip"),
           "ea_n_parents": np.int64(5), "ea_n_children": np.int64(1), "ea_n_generations": np.int64(1),
           "ea_bitflip_prob": np.float64(np.nan), "ea_Mprime": np.int64(suff["Mprime"])}
    out["t0_ss_in"] = pack(suff["ss"])
    out.update(theta_arrays("t0_in_", theta, BSC_KEYS))
    for t in range(n_steps):
        np.random.seed(1000 + seed + t)
        do_rec = t < 2
        F, nu, nsub, theta = model.step(theta, suff, my_data, do_reconstruction=do_rec)
        out["t%d_do_rec" % t] = np.bool_(do_rec)
        out["t%d_F" % t] = np.float64(F)
        out["t%d_y_reconstructed" % t] = my_data["y_reconstructed"].copy()
        out["t%d_ss_out" % t] = pack(suff["ss"])
        out["t%d_lpj_out" % t] = suff["lpj"].copy()
        out.update(theta_arrays("t%d_out_" % t, theta, BSC_KEYS))
    path = os.path.join(HERE, "missing_%s.npz" % name)
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def make_shape_fixture(name, algo, D, H, S, N, seed, n_steps=2, ea=("fit", "randflip", 10, 1, 1)):
    """EM steps of the reference at a BASELINE.json shape (true D, H, S).  Raw outputs would be tens of MB
    (W 1-2 MB, K^n up to 100 MB), so the fixture holds what _sketch.py makes of them: one hash per
    datapoint for K^n and for the candidate batch (bit-exactness is an equality test), three numbers per
    lpj row, probe products of every accumulator and of Theta^new.  Inputs are not stored either: Y, Theta^init and K^n(0)
    come from np.random.seed(seed) through randn / standard_init / init_states (hashes stored, so a
    replay that draws differently fails loudly)."""
    np.random.seed(seed)
    if algo == "ebsc":
        model = BSC(D, H, S)
        keys = BSC_KEYS
    else:
        model = SSSC(D, H, S, use_storage=False)  # the only memory-scalable mode
        keys = SSSC_KEYS
    Y = np.random.randn(N, D)
    my_data = {"y": Y, "x_infr": np.ones_like(Y, dtype=bool)}
    theta = model.check_params(model.standard_init(my_data))
    suff = init_states(N, S, H, ea[0], ea[1], ea[2], ea[3], ea[4])
    out = {
        "algo": np.array(algo), "D": np.int64(D), "H": np.int64(H), "S": np.int64(S), "N": np.int64(N),
        "seed": np.int64(seed), "n_steps": np.int64(n_steps), "Y_sha1": np.array(_sketch.array_sha1(Y)),
        "ea_parent_selection": np.array(ea[0]), "ea_mutation": np.array(ea[1]),
        "ea_n_parents": np.int64(suff["n_parents"]), "ea_n_children": np.int64(suff["n_children"]),
        "ea_n_generations": np.int64(suff["n_generations"]), "ea_bitflip_prob": np.float64(np.nan),
        "ea_Mprime": np.int64(suff["Mprime"]), "ss_in_hash": _sketch.state_hashes(suff["ss"]),
    }
    for k in keys:
        out["in_sha1_" + k] = np.array(_sketch.array_sha1(np.asarray(theta[k], dtype=np.float64)))
    import time
    for t in range(n_steps):
        np.random.seed(1000 + seed + t)
        TRACE.clear()
        COMM.log.clear()
        COMM.recording = True
        t0 = time.time()
        F, nu, nsub, theta = model.step(theta, suff, my_data)
        dt = time.time() - t0
        COMM.recording = False
        out["t%d_F" % t] = np.float64(F)
        out["t%d_S_nunique" % t] = np.float64(nu)
        out["t%d_S_sub" % t] = np.float64(nsub)
        out["t%d_ss_hash" % t] = _sketch.state_hashes(suff["ss"])
        out["t%d_k_sum" % t] = suff["ss"].sum(axis=(1, 2)).astype(np.int64)
        out["t%d_lpj_rows" % t] = _sketch.lpj_rows(suff["lpj"])
        out["t%d_cand_counts" % t] = np.array([s.shape[0] for s, _ in TRACE], dtype=np.int64)
        out["t%d_cand_hash" % t] = _sketch.ragged_hashes([s for s, _ in TRACE])
        out["t%d_cand_lpj_sum" % t] = np.array([l.sum() for _, l in TRACE])
        bufs = [p for kind, p in COMM.log if kind == "Allreduce"]
        scal = [float(p) for kind, p in COMM.log if kind == "allreduce" and np.ndim(p) == 0]
        if algo == "ebsc":
            names = ["Wp", "Wq", "pies"]
            out["t%d_sum_sigma" % t] = np.float64(scal[-1])
        else:
            names = ["xpt_s", "xpt_ss", "xpt_sz", "xpt_szsz", "s_sz_outer", "sz_sz_outer", "Wp", "y_outer_diag"]
        out["t%d_sum_Fs" % t] = np.float64(scal[4])
        assert len(bufs) == len(names), (len(bufs), names)
        conds = []
        worst = 1.0
        for nm, b in zip(names, bufs):
            out["t%d_sum_%s" % (t, nm)] = _sketch.sketch(b)
            if nm in ("Wq", "xpt_szsz", "xpt_ss"):  # the matrices the Theta update solves with
                cn = np.linalg.cond(b)
                worst = max(worst, cn)
                conds.append("%s cond %.2e" % (nm, cn))
        out["t%d_cond" % t] = np.float64(worst)  # tests compare Theta^new only where the update is well posed
        for k in keys:
            out["t%d_out_%s" % (t, k)] = _sketch.sketch(theta[k])
        print("  %s step %d: %.1f s, F %.6f, %s" % (name, t, dt, F, ", ".join(conds)), flush=True)
    path = os.path.join(HERE, "shape_%s.npz" % name)
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB", flush=True)


SHAPES = {  # name: (algo, D, H, S, N, seed, ea) -- BASELINE.json configs[1..4] at their true D, H, S
    # small N (seconds per step): E-step, selection and every accumulator
    "c2_small": ("es3c", 256, 128, 64, 24, 52, ("fit", "randflip", 10, 1, 1)),
    "c3_small": ("ebsc", 64, 256, 128, 48, 53, ("fit", "randflip", 10, 1, 1)),
    "c4_small": ("es3c", 256, 512, 200, 12, 54, ("fit", "randflip", 10, 1, 1)),
    "c5_small": ("ebsc", 256, 1024, 256, 24, 55, ("fit", "randflip", 10, 1, 1)),
    # more than 64 candidates per datapoint (12 parents x 6 children) at the c3 shape
    "c3_wide": ("ebsc", 64, 256, 128, 40, 56, ("fit", "randflip", 12, 6, 1)),
    # N a few times H: the Theta update is well posed, so Theta^new and a second, chained step are pinned too
    "c2": ("es3c", 256, 128, 64, 512, 62, ("fit", "randflip", 10, 1, 1)),
    "c3": ("ebsc", 64, 256, 128, 1024, 63, ("fit", "randflip", 10, 1, 1)),
    "c4": ("es3c", 256, 512, 200, 1536, 64, ("fit", "randflip", 10, 1, 1)),
    "c5": ("ebsc", 256, 1024, 256, 3072, 65, ("fit", "randflip", 10, 1, 1)),
    # five chained EM steps (SURVEY 8d metric 3: free energy after T = 5 iterations from identical init and streams)
    "c2x5": ("es3c", 256, 128, 64, 512, 66, ("fit", "randflip", 10, 1, 1), 5),
    "c3x5": ("ebsc", 64, 256, 128, 1024, 67, ("fit", "randflip", 10, 1, 1), 5),
}


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "learn":  # learning-level fixture (round 4): bars training runs of the reference
        make_learn_bars()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "missing":  # only the incomplete-data fixture (added later)
        make_missing_fixture()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "missing_es3c":
        make_missing_fixture_es3c()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "shape":  # BASELINE-shape fixtures (minutes each; run selected names in parallel)
        for nm in (sys.argv[2:] or sorted(SHAPES)):
            a, D, H, S, N, seed, ea = SHAPES[nm][:7]
            make_shape_fixture(nm, a, D, H, S, N, seed, n_steps=(SHAPES[nm][7] if len(SHAPES[nm]) > 7 else 2), ea=ea)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "perm":  # permanent all-zero state (S_perm = 1), added later
        make_step_fixture("ebsc_perm", "ebsc", 20, 24, 12, 30, seed=71, n_steps=2, ea=("fit", "randflip", 4, 2, 1), permanent=PERM_ZERO)
        make_step_fixture("es3c_perm", "es3c", 20, 24, 12, 30, seed=72, n_steps=2, ea=("fit", "randflip", 4, 2, 1), permanent=PERM_ZERO)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "lpj":  # the direct-operator fixtures only (lpj_*.npz)
        make_lpj_fixtures()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "lpj_k3":  # exactly singular Psi_A with three or more active latents (round 3)
        make_lpj_singular_k3()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "lpj_r4":  # round 4: singular M_s inside an indefinite Psi; more than 64 active latents
        make_lpj_indefinite()
        make_lpj_dense()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "bg":  # round 4: permanent background unit, exact E-steps
        make_background_fixtures()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "prec32":  # SSSC(precision=np.float32) (sssc.py:49), added in round 3
        make_step_fixture("es3c_f32", "es3c", 24, 72, 30, 40, seed=4, n_steps=2, precision=np.float32)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "recon":  # only the reconstruction fixtures (added later)
        make_recon_fixture("ebsc", "ebsc", 25, 10, 8, 30, seed=21)
        make_recon_fixture("es3c", "es3c", 25, 10, 8, 30, seed=22)
        sys.exit(0)
    make_kat_bars()
    make_lpj_fixtures()
    make_vary_kn()
    make_full_F()
    make_step_fixture("ebsc_bars", "ebsc", 25, 10, 8, 24, seed=1, n_steps=3, data="bars", ea=("fit", "randflip", 5, 1, 1))
    make_step_fixture("es3c_bars", "es3c", 25, 10, 8, 24, seed=2, n_steps=3, data="bars", ea=("fit", "randflip", 5, 1, 1))
    make_step_fixture("ebsc_mid", "ebsc", 48, 100, 40, 60, seed=3, n_steps=2)
    make_step_fixture("es3c_mid", "es3c", 24, 72, 30, 40, seed=4, n_steps=2)
    make_step_fixture("es3c_dense", "es3c", 32, 40, 24, 12, seed=5, n_steps=2, p_init=6.0 / 40, use_storage=False)
    make_step_fixture("ebsc_dense", "ebsc", 32, 130, 24, 160, seed=6, n_steps=2, p_init=8.0 / 130)
    make_step_fixture("ebsc_sparseflip", "ebsc", 20, 24, 12, 30, seed=7, n_steps=2, ea=("rand", "sparseflip", 4, 2, 1), bitflip_prob=0.1)
    make_step_fixture("es3c_cross", "es3c", 20, 24, 12, 30, seed=8, n_steps=2, ea=("fit", "cross_randflip", 4, 1, 1))
    make_step_fixture("ebsc_gen2", "ebsc", 20, 24, 12, 30, seed=9, n_steps=2, ea=("fit", "randflip", 4, 2, 2), Mprime=5)
    make_recon_fixture("ebsc", "ebsc", 25, 10, 8, 30, seed=21)
    make_recon_fixture("es3c", "es3c", 25, 10, 8, 30, seed=22)
    make_missing_fixture()
    make_missing_fixture_es3c()
    make_step_fixture("ebsc_perm", "ebsc", 20, 24, 12, 30, seed=71, n_steps=2, ea=("fit", "randflip", 4, 2, 1), permanent=PERM_ZERO)
    make_step_fixture("es3c_perm", "es3c", 20, 24, 12, 30, seed=72, n_steps=2, ea=("fit", "randflip", 4, 2, 1), permanent=PERM_ZERO)
    make_step_fixture("es3c_f32", "es3c", 24, 72, 30, 40, seed=4, n_steps=2, precision=np.float32)
    make_lpj_singular_k3()
    make_lpj_indefinite()
    make_lpj_dense()
    make_background_fixtures()
    make_learn_bars()
    for nm in sorted(SHAPES):
        a, D, H, S, N, seed, ea = SHAPES[nm][:7]
        make_shape_fixture(nm, a, D, H, S, N, seed, n_steps=(SHAPES[nm][7] if len(SHAPES[nm]) > 7 else 2), ea=ea)
