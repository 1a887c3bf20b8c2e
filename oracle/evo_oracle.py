"""CPU oracle for the EVO hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

This module is a loop-faithful NumPy restatement of the reference algorithm
(tvlearn/evo, /root/reference) for the path named by BASELINE.json:north_star:
log-pseudo-joint evaluation over candidate binary states (EBSC / ES3C), the
log-sum-exp free energy, K^n selection, and the M-step sufficient statistics.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import it, and only as the checker / the timed CPU baseline.  Nothing under
``evo_amd/`` imports it; the product path fails loudly when the HIP library is absent.

Parity pinning: the reference ships no tests or golden vectors, so
this oracle is pinned against outputs of the reference itself, generated in the build
container by ``tests/golden/make_golden.py`` (reference imported from /root/reference
with a single-rank mpi4py stand-in) and committed as ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` replays every fixture through this file.

Every function cites the reference lines it restates (paths relative to /root/reference).
The numerics deliberately keep the reference's operation order (same NumPy/SciPy calls on
the same operands) so trajectories agree bit-for-bit on the fixture inputs; the code
organisation (pure functions over explicit arguments instead of the reference's class +
scratch-dict protocol) is our own.

Permanent states: ``allzero`` (the exact-likelihood path ``free_energy(full=True)`` needs it) and, since round 4,
the ``background`` unit (last latent on in every state) and exact E-steps (S == 2^H_); pinned by
tests/golden/background.npz and step_*_bg / step_*_exact*.npz.
"""
from __future__ import annotations

import warnings
from itertools import combinations

import numpy as np
from scipy.special import logsumexp

F64_MIN = np.finfo(np.float64).min  # eps_lpj / "log_tiny", bsc.py:23-24, sssc.py:37,42
F64_TINY = np.finfo(np.float64).tiny  # eps_pjc_sum, sssc.py:36,43
F64_EPS = np.finfo(np.float64).eps  # eps_mus, sssc.py:38,46
B_MAX = 0.0  # _models.py:55
B_MAX_SHFT = np.inf  # _models.py:56


# ---------------------------------------------------------------------------------------
# small helpers
# ---------------------------------------------------------------------------------------
def new_counters():
    """Reset counters written by E_step_precompute (bsc.py:123-125, sssc.py:363-366)."""
    return {"isnan": 0, "smaller_eps": 0, "isinf": 0, "psi_pinv": 0}


def lpj_clamp(lpj, counters):
    """_models.py:567-596.  NaN -> finfo.min, values < finfo.min (only -inf) -> finfo.min,
    then EVERY +-inf of the *original* array -> B_max (0.0).  Exactly one counter per call
    (if / elif chain).  Operates in place and returns the array."""
    nan_m = np.isnan(lpj)
    low_m = lpj < F64_MIN
    inf_m = np.isinf(lpj)
    if nan_m.any():
        counters["isnan"] += 1
    elif low_m.any():
        counters["smaller_eps"] += 1
    elif inf_m.any():
        counters["isinf"] += 1
    lpj[nan_m] = F64_MIN
    lpj[low_m] = F64_MIN
    lpj[inf_m] = B_MAX
    return lpj


def _row_keys(rows_int, H):
    """View an (R, H) C-contiguous int64 0/1 matrix as R opaque byte strings, the trick the
    reference uses to de-duplicate rows with np.unique (eas.py:252-256,
    variational/utils.py:107-111,279-282).  memcmp order == lexicographic, h=0 most
    significant."""
    return rows_int.view(np.dtype((np.void, rows_int.dtype.itemsize * H)))


def _first_unique(rows_bool_list, H):
    """Concatenate bool row blocks, return (int matrix, sorted-unique first-occurrence idx)."""
    conc = np.ascontiguousarray(np.concatenate(rows_bool_list, axis=0), dtype=int)
    _, first = np.unique(_row_keys(conc, H), return_index=True)
    return conc, first


# ---------------------------------------------------------------------------------------
# evolutionary operators  (evo/variational/eas.py)
# ---------------------------------------------------------------------------------------
def randflip(parents, n_children, sparseness=None, p_bf=None):
    """eas.py:10-43: every parent is repeated n_children times; child c of parent p gets
    one bit flipped, the flipped positions of one parent being distinct (argpartition of a
    uniform matrix).  Consumes np.random.rand(n_parents, H)."""
    n_par, H = parents.shape
    kids = np.repeat(parents, n_children, axis=0)
    pick = np.argpartition(np.random.rand(n_par, H), n_children - 1, axis=1)[:, :n_children]
    rows = np.arange(n_children * n_par)
    cols = pick.flatten()
    kids[rows, cols] = np.logical_not(kids[rows, cols])
    return kids


def sparseflip(parents, n_children, sparseness, p_bf):
    """eas.py:46-100: sparsity-driven independent bit flips; consumes
    np.random.random((n_parents*n_children, H))."""
    assert p_bf is not None, "Please specify the bitflip probability"
    n_par, H = parents.shape
    n_on = parents.sum(axis=1)
    kids = np.repeat(parents, n_children, axis=0)
    eps = 1e-100
    alpha = (H - n_on) * ((H * p_bf) - (sparseness - n_on)) / (
        (sparseness - n_on + H * p_bf) * n_on + eps
    )
    p_off = (H * p_bf) / (H + (alpha - 1.0) * n_on + eps)
    p_on = alpha * p_off
    p_off = np.repeat(np.repeat(p_off[:, None], H, axis=1), n_children, axis=0)
    p_on = np.repeat(np.repeat(p_on[:, None], H, axis=1), n_children, axis=0)
    p = np.empty_like(p_off, dtype=float)
    p[kids] = p_on[kids]
    p[np.logical_not(kids)] = p_off[np.logical_not(kids)]
    flips = np.random.random((n_par * n_children, H)) < p
    kids[flips] = np.logical_not(kids[flips])
    return kids


def cross(parents, *_unused):
    """eas.py:103-125: one-point crossover of every unordered parent pair, two children per
    pair; one np.random.randint(1, H) per pair."""
    n_par, H = parents.shape
    kids = np.empty((n_par * (n_par - 1), H), dtype=bool)
    slot = np.arange(2)
    for pair in combinations(range(n_par), 2):
        cut = np.random.randint(low=1, high=H)
        kids[slot] = parents[pair, :]
        kids[slot, cut:] = parents[pair[-1::-1], cut:]
        slot += 2
    return kids


def cross_randflip(parents, n_children, sparseness, p_bf):
    """eas.py:128-130."""
    return randflip(cross(parents), 1, sparseness, p_bf)


def cross_sparseflip(parents, n_children, sparseness, p_bf):
    """eas.py:133-135."""
    return sparseflip(cross(parents), 1, sparseness, p_bf)


def fitparents(candidates, n_parents, lpj):
    """eas.py:138-146: fitness-proportional sampling without replacement."""
    fit = lpj - 2 * np.min([np.min(lpj), 0.0])
    fit = fit / fit.sum()
    return candidates[np.random.choice(candidates.shape[0], size=n_parents, replace=False, p=fit)]


def randparents(candidates, n_parents, lpj=None):
    """eas.py:149-150."""
    return candidates[np.random.choice(candidates.shape[0], size=n_parents, replace=False)]


PARENT_SELECTION = {"fit": fitparents, "rand": randparents}
MUTATION = {
    "randflip": randflip,
    "sparseflip": sparseflip,
    "cross": cross,
    "cross_randflip": cross_randflip,
    "cross_sparseflip": cross_sparseflip,
}


def evolve_states(states, lpj, ea, piH, eval_lpj):
    """eas.py:153-313.  ``states`` (S,H) bool, ``lpj`` (S,), ``ea`` the hyper-parameter dict
    made by init_states.  Returns (new unique states in lexicographic order per generation,
    their lpj).  Includes the reference's re-use of old lpj values with the index shift of
    eas.py:284-293.  Permanent background unit (eas.py:213-239): the last latent is not mutated -- the
    operators see the parents without it and every child gets it back switched on."""
    background = bool(ea["permanent"]["background"])
    incl = ea["incl"]
    n_par, n_child, n_gen = ea["n_parents"], ea["n_children"], ea["n_generations"]
    select, mutate = ea["parent_selection"], ea["mutation_algorithm"]
    K, H = states.shape
    known = np.concatenate((incl, states), axis=0)
    known_lpj = lpj
    n_known = known.shape[0]
    cursor = 0
    out_states = out_lpj = fresh = None
    pool = None
    for g in range(n_gen):
        if g == 0:
            parents = select(states, np.min([K, n_par]), lpj)
        else:
            parents = select(out_states[pool], np.min([len(pool), n_par]), out_lpj[pool])
        kids = mutate(parents[:, :(H - 1 if background else H)], n_child, piH, ea["bitflip_prob"])
        if background:
            kids = np.concatenate((kids, np.ones((kids.shape[0], 1), dtype=bool)), axis=1)
        if g == 0:
            per_gen = kids.shape[0]
            out_states = np.zeros((per_gen * n_gen, H), dtype=bool)
            out_lpj = np.zeros(per_gen * n_gen)
            fresh = np.zeros(out_lpj.size, dtype=bool)
        conc, first = _first_unique([known, kids], H)
        first = first[first >= n_known]
        n_fresh = first.size
        end_fresh = cursor + n_fresh
        if n_fresh > 0:
            sl = range(cursor, end_fresh)
            out_states[sl] = conc[first, :].astype(np.bool_)
            out_lpj[sl] = eval_lpj(out_states[sl])
            fresh[sl] = True
        # children that duplicate an already-known state are copied with "their" lpj
        rev = np.ascontiguousarray(conc[::-1], dtype=int)
        _, idx_old = np.unique(_row_keys(rev, H), return_index=True)
        idx_old = idx_old[np.logical_and(idx_old >= per_gen, idx_old < (per_gen + n_known - 1))] - per_gen
        if idx_old.size > 0:
            base = np.arange(n_known - 1)
            idx_old = base[::-1][idx_old]
            idx_old = np.setdiff1d(base, idx_old)
            end = end_fresh + idx_old.size
            sl_old = range(end_fresh, end)
            out_states[sl_old] = known[idx_old + 1].astype(bool)
            out_lpj[sl_old] = known_lpj[idx_old]
        else:
            end = end_fresh
        if n_fresh > 0:
            known = np.append(known, out_states[sl], axis=0)
            known_lpj = np.append(known_lpj, out_lpj[sl])
            n_known = known.shape[0]
        if cursor == end:
            warnings.warn("No new and unique states. Skipping evolutionary loop.")
            break
        pool = range(cursor, end)
        cursor = end
    return out_states[fresh], out_lpj[fresh]


# ---------------------------------------------------------------------------------------
# K^n initialisation and selection  (evo/variational/utils.py)
# ---------------------------------------------------------------------------------------
def all_states_matrix(H):
    """variational/utils.py:58-67: all 2^H states ordered by |s| then lexicographic combos."""
    combos = []
    for g in range(H + 1):
        for c in combinations(range(H), g):
            combos.append(np.array(c, dtype=np.int8))
    sm = np.zeros((len(combos), H), dtype=bool)
    for i, c in enumerate(combos):
        sm[i, c] = True
    return sm


def init_states(N, S, H, parent_selection, mutation_algorithm, no_parents, no_children,
                no_generations, bitflip_prob=None, Mprime=None, p_init_Kn=None, permanent=None):
    """variational/utils.py:19-228.  Draws S Bernoulli(p_init_Kn) rows per datapoint, tops up until S unique rows
    exist (first-occurrence order of the sorted-unique result), and stores the EA knobs.  Permanent background unit
    (:42-47, :96-98, :140-141): the draws cover the first H - 1 latents, the last one is on in every state, no permanent
    all-zero state whatever ``allzero`` says.  Exact E-steps (:55, :71-88): S == 2^H_ makes K^n the full state table for
    every datapoint (with the all-zero permanent state: its 2^H - 1 other rows, i.e. ONE ROW FEWER than S)."""
    if permanent is None:
        permanent = {"background": False, "allzero": False, "singletons": False}
    background = bool(permanent["background"])
    Hd = H - 1 if background else H  # latents that are drawn / enumerated
    S_perm = 0 if background else (1 if (permanent["allzero"] == 1 and permanent["singletons"] == 0) else 0)
    incl = np.zeros((S_perm, Hd), dtype=bool)
    sm = all_states_matrix(Hd) if Hd < 12 else None
    if S == 2 ** Hd:
        assert Hd < 12, "Exact E-steps too expensive for H={})".format(Hd)
        if background:
            table = np.concatenate((sm, np.ones((sm.shape[0], 1), dtype=bool)), axis=1)
            lpj = np.empty((N, 2 ** Hd))
        else:
            lpj = np.empty((N, S + S_perm))
            table = sm[1:, :].copy() if S_perm == 1 else sm.copy()
        ss = np.tile(table[None, :, :], (N, 1, 1))
    else:
        if p_init_Kn is None:
            p_init_Kn = 1.0 / H
        lpj = np.empty((N, S + S_perm))
        ss = np.empty((N, S, H), dtype=bool)
        if background:
            ss[:, :, -1] = True
        for n in range(N):
            draw = np.random.random(size=(S, Hd)) < p_init_Kn
            conc, first = _first_unique([incl, draw], Hd)
            first = first[first >= S_perm]
            have = conc[first, :].astype(np.bool_)
            while have.shape[0] < S:
                more = np.random.random(size=(S, Hd)) < p_init_Kn
                conc, first = _first_unique([incl, have, more], Hd)
                first = first[first >= (S_perm + have.shape[0])]
                have = np.concatenate((have, conc[first, :].astype(np.bool_)), axis=0)
            ss[n, :, :Hd] = have[:S]
    if background:
        incl = np.zeros((S_perm, H), dtype=bool)
    if "cross" in mutation_algorithm:
        no_children = no_parents - 1
    assert no_parents <= S
    if Mprime is None:
        Mprime = S
    else:
        assert Mprime <= S
    return {
        "ss": ss, "lpj": lpj, "permanent": permanent, "incl": incl, "S_perm": S_perm, "sm": sm,
        "n_parents": no_parents, "n_children": no_children, "n_generations": no_generations,
        "parent_selection": PARENT_SELECTION[parent_selection],
        "mutation_algorithm": MUTATION[mutation_algorithm],
        "bitflip_prob": bitflip_prob, "Mprime": Mprime,
    }


def vary_Kn(lpj_old, lpj_new, lpj_out, states, states_new, H, S, S_perm, incl, Mprime):
    """variational/utils.py:231-337, unification branch (the only one the models use).
    In-place: ``states`` (S,H) rows and ``lpj_out`` (S,) are overwritten; ``lpj_old`` is also
    modified (the reference aliases it).  Returns (#new unique, #swapped)."""
    conc, first = _first_unique([incl, states, states_new], H)
    fresh = first[first >= (S + S_perm)]
    states_new = conc[fresh, :].astype(np.bool_)
    lpj_new = lpj_new[fresh - (S_perm + S)]
    m = min([lpj_new.size, Mprime])
    top_new = np.argpartition(lpj_new, -m)[-m:]
    low_old = np.argpartition(lpj_old, m - 1)[:m]
    pool = np.stack((lpj_new[top_new], lpj_old[low_old])) if len(low_old) > 0 else lpj_new[top_new]
    order = np.array(np.unravel_index(np.argsort(pool, axis=None)[::-1], (2, top_new.size)))
    best = order[:, :m]
    good = top_new[best[1, best[0] == 0]]
    worst = order[:, -1:-1 - m:-1]
    bad = low_old[worst[1, worst[0] == 1]]
    for j in range(good.size):
        states[bad[j]] = states_new[good[j]]
        assert lpj_new[good[j]] >= lpj_old[bad[j]]
        lpj_old[bad[j]] = lpj_new[good[j]]
    lpj_out[:] = lpj_old
    return fresh.size, good.size


# ---------------------------------------------------------------------------------------
# shared free-energy reduction
# ---------------------------------------------------------------------------------------
def free_energy_sum(lpj):
    """_models.py:544-546 / 433-435 / sssc.py:777-779: Fs = sum_n (logsumexp(lpj_n+B_n)-B_n)
    with B_n = min(B_max - max_s lpj_ns, B_max_shft)."""
    B = np.minimum(B_MAX - lpj.max(axis=1), B_MAX_SHFT)
    return (logsumexp(lpj + B[:, None], axis=1) - B).sum()


def check_params(theta, policy):
    """_models.py:101-159 for one rank: clamp each parameter into [low, up] (+ diagonal floor)."""
    for name, (low, up, absify, low_diag) in policy.items():
        v = theta[name]
        if np.isscalar(v):
            if v < low:
                v = low
            if v >= up:
                v = up
            if absify:
                v = np.abs(v)
            if low_diag is not None and v < low_diag:
                v = low_diag
        else:
            v = np.minimum(up, np.maximum(low, v))
            if absify:
                v = np.abs(v)
            if low_diag is not None:
                mask = np.diag(v) < low_diag
                v[np.diag(mask)] = low_diag
        theta[name] = v
    return theta


def standard_init_common(Y):
    """_models.py:240-255 (complete data, one rank): data mean and mean variance."""
    y_mean = np.sum(Y, 0) / Y.shape[0]
    var = np.sum((Y - y_mean) ** 2, 0) / Y.shape[0]
    return y_mean, var


# ---------------------------------------------------------------------------------------
# BSC / EBSC   (evo/models/bsc.py + evo/models/_models.py)
# ---------------------------------------------------------------------------------------
TOL = 1e-5
BSC_POLICY = {  # _models.py:47-52
    "W": (-np.inf, +np.inf, False, None),
    "pi": (TOL, 1.0 - TOL, False, None),
    "sigma": (TOL, +np.inf, False, None),
}


def bsc_standard_init(Y, H, x_infr=None):
    """_models.py:205-283 default branch: W = y_mean + N(0, (sigma/4)^2), pi=1/H.  With missing
    entries (x_infr not all True) mean and variance run over the reliable entries only
    (_models.py:246-267; note the mean divides by my_N, not by the per-dimension count)."""
    D = Y.shape[1]
    if x_infr is None or x_infr.all():
        y_mean, var = standard_init_common(Y)
        sigma = np.sqrt(var.sum() / D)
    else:
        N = Y.shape[0]
        y_mean = np.zeros(D)
        for n in range(N):
            y_mean[x_infr[n]] += Y[n][x_infr[n]]
        y_mean = y_mean / N
        tmp = np.zeros(D)
        for n in range(N):
            tmp[x_infr[n]] += (Y[n][x_infr[n]] - y_mean[x_infr[n]]) ** 2
        sigma = np.sqrt(tmp.sum() / x_infr.flatten().sum())
    noise = np.random.normal(scale=sigma / 4.0, size=[D, H])
    return {"W": y_mean[:, None] + noise, "pi": 1.0 / H, "sigma": sigma}


def bsc_precompute(theta, D, H, x_infr=None):
    """bsc.py:99-125.  Adds pre1, pil_bar, piH, ljc to ``theta`` (incomplete data: the Gaussian
    normaliser counts the reliable entries, bsc.py:113-118)."""
    pi, sigma = theta["pi"], theta["sigma"]
    theta["piH"] = pi * H
    theta["pre1"] = -1.0 / 2.0 / sigma / sigma
    theta["pil_bar"] = np.log(pi / (1.0 - pi))
    if x_infr is not None and not x_infr.all():
        sum_n_d = x_infr.sum()
        theta["ljc"] = H * np.log(1.0 - pi) - np.log(2 * np.pi * sigma * sigma) * sum_n_d / x_infr.shape[0] / 2
    else:
        theta["ljc"] = H * np.log(1.0 - pi) - D / 2 * np.log(2 * np.pi * sigma * sigma)
    return new_counters()


def bsc_lpj(theta, states, y, counters, x_infr=None):
    """bsc.py:78-97: lpj_c = pil_bar*|s_c| + pre1*sum_d (sum_h s_ch W_dh - y_d)^2."""
    Wt = theta["W"].T
    if x_infr is None:
        x_infr = np.ones(y.shape[0], dtype=bool)
    n_on = states.sum(axis=1)
    Wbar = np.dot(states, Wt[:, x_infr])
    lpj = theta["pre1"] * ((Wbar - y[x_infr]) ** 2).sum(axis=1) + theta["pil_bar"] * n_on
    return lpj_clamp(lpj, counters)


def bsc_lpj_allzero(theta, y, counters, x_infr=None):
    """bsc.py:59-76 with permanent['allzero']: lpj = pre1*||y_obs||^2."""
    lpj = np.empty((1,))
    if x_infr is None:
        x_infr = np.ones(y.shape[0], dtype=bool)
    lpj[0] = theta["pre1"] * (y[x_infr] ** 2).sum()
    return lpj_clamp(lpj, counters)


def bsc_E_step(theta, suff, Y, trace=None, x_infr=None):
    """_models.py:453-565 on one rank.  Mutates suff['ss'], suff['lpj'] in place; returns
    (Fs, sum_nunique, sum_sub, counters) -- the *un-normalised* per-rank quantities that the
    reference all-reduces (_models.py:540-547).  ``trace`` (list) optionally receives
    (n, candidate states, their lpj) for the golden fixtures."""
    N, D = Y.shape
    H = theta["W"].shape[1]
    S = suff["ss"].shape[1]
    counters = bsc_precompute(theta, D, H, x_infr)
    S_perm, incl, Mprime = suff["S_perm"], suff["incl"], suff["Mprime"]
    n_uniq = n_sub = 0.0
    for n in range(N):
        y = Y[n]
        xi = None if x_infr is None else x_infr[n]
        cur = suff["ss"][n]
        if S_perm > 0:
            suff["lpj"][n, 0:S_perm] = bsc_lpj_allzero(theta, y, counters, xi)
        cur_lpj = bsc_lpj(theta, cur, y, counters, xi)
        new_s, new_l = evolve_states(cur, cur_lpj, suff, theta["piH"],
                                     lambda st: bsc_lpj(theta, st, y, counters, xi))
        if trace is not None:
            trace.append((n, new_s.copy(), new_l.copy()))
        a, b = vary_Kn(cur_lpj, new_l, suff["lpj"][n, S_perm:], cur, new_s, H, S, S_perm, incl, Mprime)
        n_uniq += a
        n_sub += b
    Fs = free_energy_sum(suff["lpj"])
    return Fs, n_uniq, n_sub, counters


def bsc_accumulate(theta, suff, Y, x_infr=None, y_rec=None):
    """bsc.py:176-223: per-rank M-step sums (my_Wp (H,D), my_Wq (H,H), my_pies (H,), my_sigma).
    Incomplete data (bsc.py:184-189): Wp uses y_reconstructed, the residual the reliable entries."""
    N, D = Y.shape
    incmpl = x_infr is not None and not x_infr.all()
    if incmpl:
        assert y_rec is not None  # bsc.py:186
        Y = y_rec
    Wt = theta["W"].T
    H = Wt.shape[0]
    lpj, ss, S_perm = suff["lpj"], suff["ss"], suff["S_perm"]
    B = np.minimum(B_MAX - lpj.max(axis=1), B_MAX_SHFT)
    pjc = np.exp(lpj + B[:, None])
    Wp = np.zeros_like(Wt)
    Wq = np.zeros((H, H))
    pies = np.zeros(H)
    sig = 0.0
    for n in range(N):
        y = Y[n]
        obs = x_infr[n] if x_infr is not None else np.ones(D, dtype=bool)  # the reference always indexes with the mask
        q = pjc[n]
        st = ss[n]
        t_Wp = np.zeros_like(Wp)
        t_Wq = np.zeros_like(Wq)
        t_pies = np.zeros(H)
        t_sig = 0.0
        if suff["permanent"]["allzero"]:
            t_sig += q[0] * (y[obs] ** 2).sum()
        t_pies += (q[S_perm:].T * st.T).sum(axis=1)
        t_Wp += np.outer((q[S_perm:].T * st.T).sum(axis=1), y)
        t_Wq += np.dot(q[S_perm:].T * st.T, st)
        t_sig += (q[S_perm:] * ((y[obs] - np.dot(st, Wt[:, obs])) ** 2).sum(axis=1)).sum()
        qs = q.sum()
        pies += t_pies / qs
        Wp += t_Wp / qs
        Wq += t_Wq / qs
        sig += t_sig / qs
    return {"Wp": Wp, "Wq": Wq, "pies": pies, "sigma": sig}


def bsc_update(theta, sums, N, D, H, to_learn=("W", "pi", "sigma"), n_reliable=None, background=False):
    """bsc.py:226-277: Theta update from the all-reduced sums.  Mutates and returns theta.
    rcond follows the reference's version test, which yields -1 on NumPy 2.x.
    n_reliable = x_infr.sum() selects the incomplete-data sigma (bsc.py:266-272: the OLD sigma
    enters with the count of RELIABLE entries, as the reference writes it)."""
    if "W" in to_learn:
        rcond = None if float(np.__version__[2:]) >= 14.0 else -1
        try:
            theta["W"] = np.linalg.lstsq(sums["Wq"], sums["Wp"], rcond=rcond)[0].T
        except np.linalg.LinAlgError:  # bsc.py:238-250 (lstsq raises only when its SVD does not converge)
            try:
                noise = np.random.normal(0, EPS_W, H)
                theta["W"] = np.dot(np.linalg.pinv(sums["Wq"] + np.outer(noise, noise)), sums["Wp"]).T
            except np.linalg.LinAlgError:
                theta["W"] = (theta["W"].T + (EPS_W * np.random.normal(0, 1, [H, D]))).T
    if "pi" in to_learn:
        pies_new = sums["pies"] / N
        if background:  # bsc.py:259-260
            pies_new[-1] = 1.0 - 1.1e-5
        theta["pi"] = pies_new.sum() / H
        theta["pies"] = pies_new
    if "sigma" in to_learn:
        if n_reliable is not None:
            theta["sigma"] = np.sqrt((sums["sigma"] + n_reliable * theta["sigma"] ** 2) / N / D)
        else:
            theta["sigma"] = np.sqrt(sums["sigma"] / N / D)
    return theta


def bsc_reconstruct(theta, suff, Y, x, x_infr=None):
    """Model.reconstruct (_models.py:614-665) with BSC.modelmean (bsc.py:279-287): entries with x
    False become sum_s q_s (W s)_d / sum_s q_s under the current Theta and K^n; datapoints without a
    single reliable entry are skipped (_models.py:648-649)."""
    lpj, ss, S_perm = suff["lpj"], suff["ss"], suff["S_perm"]
    B = np.minimum(B_MAX - lpj.max(axis=1), B_MAX_SHFT)
    pjc = np.exp(lpj + B[:, None])
    y_rec = Y.copy()
    Wt = theta["W"].T
    for n in range(Y.shape[0]):
        if x_infr is not None and np.logical_not(x_infr[n]).all():
            continue
        this_x = x[n]
        this_W = Wt[:, np.logical_not(this_x)]
        this_mu = np.dot(ss[n], this_W).T                       # (D_miss, S)
        this_pjc = pjc[n]
        est = (this_mu * this_pjc[None, S_perm:]).sum(axis=1) / this_pjc.sum()
        y_rec[n][np.logical_not(this_x)] = est
    return y_rec


def bsc_step(theta, suff, Y, to_learn=("W", "pi", "sigma"), trace=None, reconstruct_x=None, x_infr=None,
             y_rec_prev=None):
    """_models.py:161-203 for BSC on one rank: check_params -> E_step [-> reconstruct] -> M_step.
    Returns (F, S_nunique, S_sub, theta) like the reference, plus the raw sums dict
    (sums["y_reconstructed"] when reconstruct_x, the my_data["x"] mask, is given)."""
    N, D = Y.shape
    H = theta["W"].shape[1]
    theta = check_params(theta, BSC_POLICY)
    Fs, nu, nsub, _ = bsc_E_step(theta, suff, Y, trace, x_infr)
    F = theta["ljc"] + Fs / N
    y_rec = bsc_reconstruct(theta, suff, Y, reconstruct_x, x_infr) if reconstruct_x is not None else None
    # incomplete data: the M-step reads my_data["y_reconstructed"], i.e. this step's or an older one
    sums = bsc_accumulate(theta, suff, Y, x_infr, y_rec if y_rec is not None else y_rec_prev)
    if y_rec is not None:
        sums["y_reconstructed"] = y_rec
    sums["Fs"] = Fs
    if len(to_learn) > 0:
        incmpl = x_infr is not None and not x_infr.all()
        theta = bsc_update(theta, sums, N, D, H, to_learn, n_reliable=x_infr.sum() if incmpl else None,
                           background=bool(suff["permanent"]["background"]))
    return F, nu / N, nsub / N, theta, sums


def bsc_free_energy_full(theta, suff, Y):
    """_models.py:333-451 with full=True (allzero permanent forced on, S_perm=1): exact
    log-likelihood by enumerating all 2^H states; H < 12 only."""
    N, D = Y.shape
    H = theta["W"].shape[1]
    sm = suff["sm"]
    assert sm is not None
    counters = bsc_precompute(theta, D, H)
    if suff.get("permanent", {}).get("background", False):  # _models.py:389-390: every state of the H - 1 others with the unit on, no all-zero state
        states = np.concatenate((sm, np.ones((sm.shape[0], 1), dtype=bool)), axis=1)
        lpj = np.zeros((N, states.shape[0]))
        for n in range(N):
            lpj[n] = bsc_lpj(theta, states, Y[n], counters)
        return theta["ljc"] + free_energy_sum(lpj) / N
    states = sm[1:, :].astype(bool)
    lpj = np.zeros((N, states.shape[0] + 1))
    for n in range(N):
        lpj[n, 0:1] = bsc_lpj_allzero(theta, Y[n], counters)
        lpj[n, 1:] = bsc_lpj(theta, states, Y[n], counters)
    return theta["ljc"] + free_energy_sum(lpj) / N


# ---------------------------------------------------------------------------------------
# SSSC / ES3C   (evo/models/sssc.py)
# ---------------------------------------------------------------------------------------
SSSC_POLICY = {  # sssc.py:51-58
    "W": (-np.inf, +np.inf, False, None),
    "pies": (TOL, 1.0 - TOL, False, None),
    "mus": (-np.inf, +np.inf, False, None),
    "Psi": (-np.inf, +np.inf, False, TOL),
    "sigma2": (TOL, +np.inf, False, None),
}
EPS_W = 5e-5
EPS_PIES = 5e-5
EPS_PSI = TOL
EPS_SIGMA2 = TOL


def sssc_standard_init(Y, H, to_learn=("W", "pies", "mus", "sigma2", "Psi"), x_infr=None):
    """sssc.py:104-197, default branch: pies~U(0.1,0.5), mus~N(0,1), Psi=I, sigma2 = mean diag
    cov + 1e-3, W = y_mean + N(0, sigma2/16).  RNG order kept.  Incomplete data (sssc.py:144-176): mean
    over the reliable entries divided by N, sigma2 = sum of squared deviations / number of reliable
    entries + 1e-3."""
    D = Y.shape[1]
    theta = {"pies": np.random.uniform(low=0.1, high=0.5, size=[H])}
    theta["mus"] = np.random.normal(0, 1, [H]) if "mus" in to_learn else np.ones(H)
    theta["Psi"] = np.diag(np.ones(H))
    if x_infr is None or x_infr.all():
        y_mean, _ = standard_init_common(Y)
        theta["sigma2"] = np.mean(np.diag(np.cov(Y.T))) + 0.001
    else:
        N = Y.shape[0]
        y_mean = np.zeros(D)
        for n in range(N):
            y_mean[x_infr[n]] += Y[n][x_infr[n]]
        y_mean = y_mean / N
        tmp = np.zeros(D)
        for n in range(N):
            tmp[x_infr[n]] += (Y[n][x_infr[n]] - y_mean[x_infr[n]]) ** 2
        theta["sigma2"] = tmp.sum() / x_infr.flatten().sum() + 0.001
    theta["W"] = y_mean[:, None] + np.random.normal(scale=np.sqrt(theta["sigma2"]) / 4.0, size=[D, H])
    return theta


def sssc_precompute(theta, D, x_infr=None, precision=np.float64):
    """sssc.py:328-366.  sigma2 goes through longdouble like the reference; incomplete data
    (sssc.py:352-357): the Gaussian normaliser counts the reliable entries.  precision = the model's
    dtype_precision (sssc.py:49, 346-349)."""
    pies = theta["pies"]
    s2 = np.asarray(theta["sigma2"]).astype("longdouble")
    theta["ljc"] = np.log(1.0 - pies).sum() - D / 2 * np.log(2 * np.pi)
    theta["piH"] = pies.sum()
    theta["pil_bar"] = np.log(pies / (1.0 - pies))
    theta["sigma2_inv"] = (1.0 / s2).astype(precision)
    theta["ljc"] -= 0.5 * (D * np.log(s2).astype(precision))
    if x_infr is not None and not x_infr.all():
        sum_n_d = x_infr.sum()
        theta["ljc"] = (np.log(1.0 - pies).sum()
                        + (-np.log(2 * np.pi) - np.log(theta["sigma2"])) * sum_n_d / x_infr.shape[0] / 2)
    return new_counters()


def sssc_state_terms(theta, state, obs=None):
    """sssc.py:276-318: everything the reference caches per state id (obs = this datapoint's x_infr:
    with incomplete data the terms belong to the datapoint, i.e. use_storage must be False)."""
    W, Psi, mus, s2i = theta["W"], theta["Psi"], theta["mus"], theta["sigma2_inv"]
    if obs is None:
        obs = np.ones(W.shape[0], dtype=bool)  # complete data; same indexing expression as the reference
    W_s = W[obs, :][:, state]
    Psi_s = Psi[state, :][:, state]
    try:
        Psi_s_inv = np.linalg.inv(Psi_s)
    except np.linalg.LinAlgError:          # sssc.py:278-283: exactly singular Psi_s
        Psi_s_inv = np.linalg.pinv(Psi_s)
    logdet_Psi = np.linalg.slogdet(Psi_s)[1]   # -inf then: C_det = -inf, lpj = +inf -> B_max (lpj_clamp)
    Wmu = np.dot(W_s, mus[state])
    sW = s2i * W_s
    M = np.dot(W_s.T, sW) + Psi_s_inv
    logdet_M = np.linalg.slogdet(M)[1]
    try:
        lam = np.linalg.inv(M)
    except np.linalg.LinAlgError:          # sssc.py:295-300
        lam = np.linalg.pinv(M)
    lam_Wt = np.dot(lam, W_s.T) * s2i
    C_inv = -np.dot(sW, lam_Wt) + s2i * np.eye(int(obs.sum()))
    return {"Wmu": Wmu, "C_det": logdet_M + logdet_Psi, "C_inv": C_inv, "lam": lam, "lam_Wt": lam_Wt}


def sssc_lpj(theta, states, y, counters, cache, obs=None):
    """sssc.py:241-326.  ``cache`` maps state bytes -> sssc_state_terms (the reference's
    ``storage``; numerically a pure memo for complete data)."""
    if obs is not None:
        y = y[obs]
    C = states.shape[0]
    quad = np.zeros(C)
    prior = np.zeros(C)
    for c in range(C):
        st = states[c]
        key = st.tobytes()
        prior[c] = theta["pil_bar"][st].sum()
        if key not in cache:
            cache[key] = sssc_state_terms(theta, st, obs)
        t = cache[key]
        r = y - t["Wmu"]
        quad[c] = -0.5 * (t["C_det"] + (r * np.dot(t["C_inv"], r)).sum())
    return lpj_clamp(quad + prior, counters)


def sssc_lpj_allzero(theta, y, counters, obs=None):
    """sssc.py:224-239."""
    lpj = np.empty((1,))
    if obs is not None:
        y = y[obs]
    lpj[0] = -0.5 * (y ** 2).sum() * theta["sigma2_inv"]
    return lpj_clamp(lpj, counters)


def sssc_EM_accumulate(theta, suff, Y, use_storage=True, trace=None,
                       to_learn=("W", "pies", "mus", "sigma2", "Psi"), evolve=True, reconstruct_x=None, x_infr=None,
                       precision=np.float64):
    """sssc.py:419-656: the fused per-datapoint loop (E-step + sufficient statistics) on one
    rank.  Returns the dict of per-rank sums the reference all-reduces (sssc.py:671-691,763,
    773-780).  With evolve=False the EA / selection is skipped (statistics of the resident K^n)."""
    N, D = Y.shape
    H = theta["W"].shape[1]
    S = suff["ss"].shape[1]
    incmpl = x_infr is not None and not x_infr.all()
    if incmpl:
        assert not use_storage and reconstruct_x is not None  # the reference needs both (sssc.py:630-633)
    counters = sssc_precompute(theta, D, x_infr, precision)
    S_perm, incl, Mprime = suff["S_perm"], suff["incl"], suff["Mprime"]
    lpj_all, ss = suff["lpj"], suff["ss"]
    mus = theta["mus"]
    cache = {}
    trace_WW = 0.0   # trace of sum_n outer(W_obs xpt_sz) (sssc.py:640-645,751)
    acc = {
        # dtype_precision arrays (sssc.py:484-498); my_Wp is float64 whatever the precision (sssc.py:490)
        "xpt_s": np.zeros(H, dtype=precision), "xpt_ss": np.zeros((H, H), dtype=precision),
        "xpt_sz": np.zeros(H, dtype=precision), "xpt_szsz": np.zeros((H, H), dtype=precision), "Wp": np.zeros((D, H)),
        "s_sz_outer": np.zeros((H, H), dtype=precision), "sz_sz_outer": np.zeros((H, H), dtype=precision),
    }
    n_uniq = n_sub = 0.0
    y_rec = Y.copy() if reconstruct_x is not None else None   # sssc.py:500-507
    for n in range(N):
        y = Y[n]
        obs = x_infr[n] if incmpl else None
        cur = ss[n]
        if S_perm > 0:
            lpj_all[n, 0:S_perm] = sssc_lpj_allzero(theta, y, counters, obs)
        cur_lpj = sssc_lpj(theta, cur, y, counters, cache, obs)
        if evolve:
            new_s, new_l = evolve_states(cur, cur_lpj, suff, theta["piH"],
                                         lambda st: sssc_lpj(theta, st, y, counters, cache, obs))
            if trace is not None:
                trace.append((n, new_s.copy(), new_l.copy()))
            a, b = vary_Kn(cur_lpj, new_l, lpj_all[n, S_perm:], cur, new_s, H, S, S_perm, incl, Mprime)
            n_uniq += a
            n_sub += b
        else:
            lpj_all[n, S_perm:] = cur_lpj
        # sufficient statistics, sssc.py:553-611
        B = np.minimum(B_MAX - lpj_all[n].max(), B_MAX_SHFT)
        q = np.exp(lpj_all[n] + B)
        e_s = np.zeros(H, dtype=precision)          # sssc.py:556-559
        e_ss = np.zeros((H, H), dtype=precision)
        e_sz = np.zeros(H, dtype=precision)
        e_szsz = np.zeros((H, H), dtype=precision)
        for s in range(S):
            st = cur[s]
            w = q[s + S_perm]
            t = cache[st.tobytes()]
            kappa = np.dot(t["lam_Wt"], (y[obs] if incmpl else y) - t["Wmu"])
            kappa += mus[st]
            second = t["lam"] + np.outer(kappa, kappa)
            k_tmp = np.array(kappa, dtype=precision)     # sssc.py:580-584: cast, then scaled in that dtype
            k_tmp *= w
            e_sz[st] += k_tmp
            sec_tmp = np.array(second, dtype=precision)  # sssc.py:586-595
            sec_tmp *= w
            tmp = np.zeros((H, H), dtype=precision)
            tmp[np.outer(st, st)] = sec_tmp.flatten()
            e_szsz += tmp
        e_s += (q[S_perm:][:, None] * cur).sum(axis=0)
        e_ss += np.dot(q[S_perm:].T * cur.T, cur)
        qs = q.sum() + F64_TINY
        e_s /= qs
        e_ss /= qs
        e_sz /= qs
        e_szsz /= qs
        if y_rec is not None:
            # sssc.py:613-627 with SSSC.modelmean (sssc.py:368-405), complete data
            this_x = reconstruct_x[n]
            this_sz = np.zeros((H, S))
            for s in range(S):
                st = cur[s]
                t = cache[st.tobytes()]
                kappa_s = np.dot(t["lam_Wt"], (y[obs] if incmpl else y) - t["Wmu"])
                kappa_s += mus[st]
                this_sz[st, s] = kappa_s
            this_mus = np.dot(theta["W"][np.logical_not(this_x), :], this_sz)   # (D_miss, S)
            est = (this_mus * q[None, S_perm:]).sum(axis=1) / q.sum()
            y_rec[n][np.logical_not(this_x)] = est
        acc["xpt_s"] += e_s
        acc["xpt_ss"] += e_ss
        acc["xpt_sz"] += e_sz
        acc["xpt_szsz"] += e_szsz
        if incmpl:
            acc["Wp"] += e_sz[None, :] * y_rec[n][:, None]   # sssc.py:631: the reconstructed row
            Wx = np.dot(theta["W"][obs, :], e_sz)            # sssc.py:640-645
            trace_WW += (Wx ** 2).sum()
        else:
            acc["Wp"] += e_sz[None, :] * y[:, None]      # sssc.py:634
        acc["s_sz_outer"] += np.outer(e_s, e_sz)         # sssc.py:637
        acc["sz_sz_outer"] += np.outer(e_sz, e_sz)       # sssc.py:646
        if not use_storage:
            cache = {}
    if incmpl:
        acc["y_inner"] = (Y[x_infr] ** 2).sum()          # sssc.py:748
        acc["trace_WW"] = trace_WW
        acc["n_reliable"] = x_infr.sum()
        acc["y_outer_diag"] = np.where(x_infr, Y, 0.0).__pow__(2).sum(axis=0)
    else:
        acc["y_outer_diag"] = (Y ** 2).sum(axis=0)       # sssc.py:761
    acc["Fs"] = free_energy_sum(lpj_all)                 # sssc.py:777-779
    acc["n_uniq"] = n_uniq
    acc["n_sub"] = n_sub
    acc["counters"] = counters
    if y_rec is not None:
        acc["y_reconstructed"] = y_rec
    return acc


def sssc_update(theta, acc, N, D, H, to_learn=("W", "pies", "mus", "sigma2", "Psi"), background=False):
    """sssc.py:687-770: Theta update from the all-reduced sums, including the reference's
    element-wise Psi product and dead '+eps' statement and the sigma2 formula
    built from first moments and the *new* W (Q4).  Mutates and returns theta."""
    sigma2_old = theta["sigma2"]
    if "W" in to_learn:
        try:
            theta["W"] = np.dot(acc["Wp"], np.linalg.inv(acc["xpt_szsz"]))
        except np.linalg.LinAlgError:  # sssc.py:696-708: pinv of the sum plus a rank-one noise term, else W + noise
            try:
                noise = np.random.normal(0, EPS_W, H)
                theta["W"] = np.dot(acc["Wp"], np.linalg.pinv(acc["xpt_szsz"] + np.outer(noise, noise)))
            except np.linalg.LinAlgError:
                theta["W"] = theta["W"] + (EPS_W * np.random.normal(0, 1, [D, H]))
    if "pies" in to_learn:
        pies = acc["xpt_s"] / N
        pies[pies <= EPS_PIES] = EPS_PIES
        pies[pies >= (1 - EPS_PIES)] = 1 - EPS_PIES
        if background:  # sssc.py:718-719
            pies[H - 1] = 1.0 - 1.1e-5
        theta["pies"] = pies
    if "mus" in to_learn:
        theta["mus"] = acc["xpt_sz"] * 1.0 / (acc["xpt_s"] + F64_EPS)
    if "Psi" in to_learn:
        Psi = np.zeros((H, H))
        Psi += np.outer(theta["mus"], theta["mus"]) * acc["xpt_ss"]
        Psi += acc["xpt_szsz"]
        Psi -= 2 * theta["mus"][:, None] * acc["s_sz_outer"]
        theta["Psi"] = Psi * np.linalg.inv(acc["xpt_ss"] + EPS_PSI * np.eye(H))
    if "sigma2" in to_learn:
        if "trace_WW" in acc:  # incomplete data, sssc.py:747-755 (OLD sigma2 x count of reliable entries)
            s2 = acc["y_inner"] - acc["trace_WW"]
            theta["sigma2"] = ((s2 + acc["n_reliable"] * sigma2_old) / N / D) + EPS_SIGMA2
        else:
            WtW = np.dot(theta["W"].T, theta["W"])
            s2 = 0.0
            s2 += acc["y_outer_diag"].sum()
            s2 -= np.trace(np.dot(acc["sz_sz_outer"], WtW))
            theta["sigma2"] = (s2 / N / D) + EPS_SIGMA2
    return theta


def sssc_step(theta, suff, Y, use_storage=True, to_learn=("W", "pies", "mus", "sigma2", "Psi"),
              trace=None, reconstruct_x=None, x_infr=None, precision=np.float64):
    """sssc.py:407-417 + EM_step on one rank.  Returns (F, S_nunique, S_sub, theta, acc).
    F uses the *old* Theta's ljc (sssc.py:472,780)."""
    N, D = Y.shape
    H = theta["W"].shape[1]
    theta = check_params(theta, SSSC_POLICY)
    acc = sssc_EM_accumulate(theta, suff, Y, use_storage, trace, to_learn, reconstruct_x=reconstruct_x,
                             x_infr=x_infr, precision=precision)
    ljc = theta["ljc"]
    theta = sssc_update(theta, acc, N, D, H, to_learn, background=bool(suff["permanent"]["background"]))
    F = ljc + acc["Fs"] / N
    return F, acc["n_uniq"] / N, acc["n_sub"] / N, theta, acc


def sssc_free_energy_full(theta, suff, Y):
    """_models.py:333-451 with full=True for SSSC."""
    N, D = Y.shape
    sm = suff["sm"]
    assert sm is not None
    counters = sssc_precompute(theta, D)
    if suff.get("permanent", {}).get("background", False):
        states = np.concatenate((sm, np.ones((sm.shape[0], 1), dtype=bool)), axis=1)
        lpj = np.zeros((N, states.shape[0]))
        cache = {}
        for n in range(N):
            lpj[n] = sssc_lpj(theta, states, Y[n], counters, cache)
        return theta["ljc"] + free_energy_sum(lpj) / N
    states = sm[1:, :].astype(bool)
    lpj = np.zeros((N, states.shape[0] + 1))
    cache = {}
    for n in range(N):
        lpj[n, 0:1] = sssc_lpj_allzero(theta, Y[n], counters)
        lpj[n, 1:] = sssc_lpj(theta, states, Y[n], counters, cache)
    return theta["ljc"] + free_energy_sum(lpj) / N


# ---------------------------------------------------------------------------------------
# data generation used by fixtures / bench (bars test, examples/bars-test/utils.py:7-36)
# ---------------------------------------------------------------------------------------
def bars_dictionary(H):
    """H/2 horizontal + H/2 vertical bars on an (H/2)x(H/2) grid, as a (D, H) matrix."""
    R = H // 2
    W = np.zeros((R, R, H))
    for i in range(R):
        W[i, :, i] = 1.0
        W[:, i, R + i] = 1.0
    return W.reshape((R * R, H))


def bsc_generate(theta, N):
    """_models.py:73-99 + bsc.py:26-57: s ~ Bern(pi), y = W s + N(0, sigma^2)."""
    D, H = theta["W"].shape
    s = np.random.random(size=(N, H)) <= theta["pi"]
    y = np.zeros((N, D))
    Wt = theta["W"].T
    for n in range(N):
        for h in range(H):
            if s[n, h]:
                y[n] += Wt[h]
    y += np.random.normal(scale=theta["sigma"], size=(N, D))
    return y, s


def sssc_generate(theta, N):
    """_models.py:73-99 + sssc.py:65-102: s ~ Bern(pies); z_A ~ N(mus_A, Psi_AA); y = W_A z_A +
    N(0, sigma2).  RNG call order as in the reference (one multivariate_normal then one randn
    per datapoint)."""
    W = theta["W"]
    D, H = W.shape
    s = np.random.random(size=(N, H)) <= theta["pies"]
    y = np.zeros((N, D))
    z = np.zeros((N, H))
    sd = np.sqrt(theta["sigma2"]) * np.ones(D)
    for n in range(N):
        mean = np.zeros(D)
        if np.sum(s[n]) > 0:
            z_n = np.random.multivariate_normal(theta["mus"][s[n]], (theta["Psi"][s[n], :])[:, s[n]], 1).flatten()
            z[n, s[n]] = z_n
            mean = np.dot(np.array(W[:, s[n]], order="C"), z_n[:, None]).flatten()
        y[n] = mean + sd * np.random.randn(D)
    return y, s, z
