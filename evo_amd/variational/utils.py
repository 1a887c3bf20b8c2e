"""K^n initialisation and host-side selection with the reference's signatures
(evo/variational/utils.py).  The accelerated E-step uses the device kernel
(csrc/kernels_common.hpp: vary_kn_kernel); ``vary_Kn`` here is the same rule on NumPy arrays for
callers that drive the per-datapoint API."""
from itertools import combinations

import numpy as np

from .eas import (cross, cross_randflip, cross_sparseflip, fitparents, randflip, randparents, row_keys,
                  sparseflip)
from ..utils.parallel import pprint

PARENT_SELECTION = {"fit": fitparents, "rand": randparents}
MUTATION = {"randflip": randflip, "sparseflip": sparseflip, "cross": cross,
            "cross_randflip": cross_randflip, "cross_sparseflip": cross_sparseflip}


def enumerate_states(H):
    """All 2^H binary states ordered by |s|, then by combination order (variational/utils.py:58-67)."""
    rows = [np.fromiter(c, dtype=np.intp, count=g) for g in range(H + 1) for c in combinations(range(H), g)]
    sm = np.zeros((len(rows), H), dtype=bool)
    for i, on in enumerate(rows):
        sm[i, on] = True
    return sm


def _unique_after(blocks, n_before):
    """Rows of concat(blocks) that are first occurrences located at index >= n_before, in
    lexicographic order (what np.unique(..., return_index=True) yields in the reference)."""
    both = np.concatenate(blocks, axis=0)
    _, first = np.unique(row_keys(both), return_index=True)
    return both[first[first >= n_before]]


def init_states(N, S, H, parent_selection, mutation_algorithm, no_parents, no_children, no_generations,
                bitflip_prob=None, Mprime=None, p_init_Kn=None, permanent=None):
    """Build ``my_suff_stat`` (variational/utils.py:19-228): S unique Bernoulli(p_init_Kn) states per
    datapoint (RNG: np.random.random((S,H)) per round, rounds repeated until S unique rows exist)
    plus the EA hyper-parameters.  Keys and dtypes are the reference's.

    ``permanent["background"]`` (:42-47, :96-98, :140-141): the last latent is a background unit, on in every state --
    the draws and the state table ``sm`` cover the other H - 1 latents, and there is no permanent all-zero state
    whatever ``allzero`` says.  ``S == 2 ** H_`` (H_ = latents that vary; < 12) means exact E-steps (:55, :71-88): K^n is
    the whole state table for every datapoint and no random number is drawn (with the permanent all-zero state the
    table's other 2^H - 1 rows: K^n then has one row fewer than S, as in the reference)."""
    permanent = permanent or {"background": False, "allzero": False, "singletons": False}
    background = bool(permanent["background"])
    Hv = H - 1 if background else H
    S_perm = 0 if background else (1 if (permanent["allzero"] == 1 and permanent["singletons"] == 0) else 0)
    incl = np.zeros((S_perm, Hv), dtype=bool)
    sm = enumerate_states(Hv) if Hv < 12 else None
    if S == 2 ** Hv:
        assert Hv < 12, "Exact E-steps too expensive for H={})".format(Hv)
        pprint("Computing exact E-steps")
        if background:
            table = np.concatenate((sm, np.ones((sm.shape[0], 1), dtype=bool)), axis=1)
            lpj = np.empty((N, 2 ** Hv))
        else:
            table = (sm[1:] if S_perm else sm).copy()
            lpj = np.empty((N, S + S_perm))
        ss = np.tile(table[None], (N, 1, 1))
    else:
        p0 = 1.0 / H if p_init_Kn is None else p_init_Kn
        lpj = np.empty((N, S + S_perm))
        ss = np.empty((N, S, H), dtype=bool)
        if background:
            ss[:, :, -1] = True
        for n in range(N):
            have = _unique_after([incl, np.random.random(size=(S, Hv)) < p0], S_perm)
            while have.shape[0] < S:
                more = np.random.random(size=(S, Hv)) < p0
                have = np.concatenate((have, _unique_after([incl, have, more], S_perm + have.shape[0])), axis=0)
            ss[n, :, :Hv] = have[:S]
    if background:
        incl = np.zeros((S_perm, H), dtype=bool)
    if "cross" in mutation_algorithm:
        no_children = no_parents - 1
        pprint("Setting no_children to pre-determined value `no_parents - 1` ({}) when using crossover".format(
            no_parents - 1))
    assert no_parents <= S
    if Mprime is None:
        Mprime = S
    assert Mprime <= S
    return {
        "ss": ss, "lpj": lpj, "permanent": permanent, "incl": incl, "S_perm": S_perm, "sm": sm,
        "n_parents": no_parents, "n_children": no_children, "n_generations": no_generations,
        "parent_selection": PARENT_SELECTION[parent_selection],
        "mutation_algorithm": MUTATION[mutation_algorithm],
        "bitflip_prob": bitflip_prob, "Mprime": Mprime,
    }


def vary_Kn(lpj_old, lpj_new, lpj, states, states_new, H, S, S_perm, incl, Mprime, unification=True,
            reject_worse=True):
    """Selection step (variational/utils.py:231-337) on host arrays, in place.

    Same rule as the device kernel: drop candidates equal to a permanent state, to a member of
    K^n or to an earlier candidate; with M' = min(#kept, Mprime) the j-th best kept candidate
    replaces the j-th worst old state while it is strictly better (j = 1..M').  For tie-free
    inputs this is exactly the reference's argpartition / argsort construction; for exact ties
    NumPy's order is unspecified and the rule here is "no swap on equality, lowest index first".
    Returns (#new unique, #swapped).  ``unification=False`` keeps the reference's set-replacement
    behaviour (variational/utils.py:325-335)."""
    kept = _unique_after_idx([incl, states, states_new], S + S_perm)
    if not unification:
        if reject_worse and (lpj_new.sum() < lpj_old.sum()):
            lpj[:] = lpj_old
            return 0, 0
        lpj[:] = lpj_new
        states[:, :] = states_new
        return kept.size, kept.size
    cand = states_new[kept]
    cand_lpj = lpj_new[kept]
    m = min(cand_lpj.size, Mprime)
    best_new = np.argsort(-cand_lpj, kind="stable")[:m]
    worst_old = np.argsort(lpj_old, kind="stable")[:m]
    n_sub = 0
    for j in range(m):
        if not cand_lpj[best_new[j]] > lpj_old[worst_old[j]]:
            break
        states[worst_old[j]] = cand[best_new[j]]
        lpj_old[worst_old[j]] = cand_lpj[best_new[j]]
        n_sub += 1
    lpj[:] = lpj_old
    return kept.size, n_sub


def _unique_after_idx(blocks, n_before):
    both = np.concatenate(blocks, axis=0)
    _, first = np.unique(row_keys(both), return_index=True)
    return first[first >= n_before] - n_before
