"""Host-side evolutionary operators with the reference's names and signatures
(evo/variational/eas.py).  They exist for the ``rng="reference"`` mode, where candidate states
must come from NumPy's global Mersenne-Twister stream in the reference's per-datapoint order so
that whole training trajectories can be compared bit for bit with the reference; every
``np.random`` call below therefore has the same shape / arguments / order as its counterpart
(cited per function).  Row de-duplication works on bit-packed rows (MSB first), whose byte order
is the lexicographic row order the reference obtains from int64 rows.

The fast path (``rng="device"``) does not use this module: see csrc/kernels_evolve.hpp.
"""
import warnings
from itertools import combinations

import numpy as np


def row_keys(rows):
    """(R,H) bool -> (R,) opaque keys whose sort order is lexicographic in the rows (h = 0 first)."""
    packed = np.packbits(np.asarray(rows, dtype=bool), axis=1)
    return np.ascontiguousarray(packed).view(np.dtype((np.void, packed.shape[1]))).ravel()


def _flip(children, rows, cols):
    children[rows, cols] ^= True
    return children


def randflip(parents, n_children, sparseness=None, p_bf=None):
    """One uniform bit flip per child, distinct positions per parent (eas.py:10-43).
    RNG: one np.random.rand(n_parents, H)."""
    P, H = parents.shape
    noise = np.random.rand(P, H)
    which = np.argpartition(noise, n_children - 1, axis=1)[:, :n_children]
    children = np.repeat(parents, n_children, axis=0)
    return _flip(children, np.arange(P * n_children), which.reshape(-1))


def sparseflip(parents, n_children, sparseness, p_bf):
    """Sparsity-driven flips (eas.py:46-100): ON bits flip with p1, OFF bits with p0 chosen so the
    expected child has ``sparseness`` ON bits and H*p_bf flips.  RNG: one
    np.random.random((n_parents*n_children, H))."""
    if p_bf is None:
        raise AssertionError("Please specify the bitflip probability")
    P, H = parents.shape
    on = parents.sum(axis=1)
    tiny = 1e-100
    alpha = (H - on) * ((H * p_bf) - (sparseness - on)) / ((sparseness - on + H * p_bf) * on + tiny)
    p0 = (H * p_bf) / (H + (alpha - 1.0) * on + tiny)
    p1 = alpha * p0
    children = np.repeat(parents, n_children, axis=0)
    prob = np.where(children, np.repeat(p1, n_children)[:, None], np.repeat(p0, n_children)[:, None])
    flips = np.random.random((P * n_children, H)) < prob
    children[flips] ^= True
    return children


def cross(parents, *_ignored):
    """One-point crossover of every unordered parent pair, two children each (eas.py:103-125).
    RNG: one np.random.randint(1, H) per pair, pairs in itertools.combinations order."""
    P, H = parents.shape
    out = np.empty((P * (P - 1), H), dtype=bool)
    for j, (a, b) in enumerate(combinations(range(P), 2)):
        cut = np.random.randint(low=1, high=H)
        out[2 * j, :cut], out[2 * j, cut:] = parents[a, :cut], parents[b, cut:]
        out[2 * j + 1, :cut], out[2 * j + 1, cut:] = parents[b, :cut], parents[a, cut:]
    return out


def cross_randflip(parents, n_children, sparseness, p_bf):
    """eas.py:128-130."""
    return randflip(cross(parents), 1, sparseness, p_bf)


def cross_sparseflip(parents, n_children, sparseness, p_bf):
    """eas.py:133-135."""
    return sparseflip(cross(parents), 1, sparseness, p_bf)


def fitness_probabilities(lpj):
    """p_s of fitparents (eas.py:139-141): shift by twice the (non-positive) minimum, normalise."""
    fit = lpj - 2 * np.min([np.min(lpj), 0.0])
    return fit / fit.sum()


def fitparents(candidates, n_parents, lpj):
    """Fitness-proportional parents without replacement (eas.py:138-146)."""
    pick = np.random.choice(candidates.shape[0], size=n_parents, replace=False, p=fitness_probabilities(lpj))
    return candidates[pick]


def randparents(candidates, n_parents, lpj=None):
    """Uniform parents without replacement (eas.py:149-150)."""
    return candidates[np.random.choice(candidates.shape[0], size=n_parents, replace=False)]


def evolve_states(my_suff_stat, model_params, eval_lpj):
    """Reference signature (eas.py:153-313): evolve ``my_suff_stat["this_states"]`` for
    n_generations and return (new unique states, their lpj); ``eval_lpj(states)`` is called once
    per generation on the not-yet-known children in lexicographic order.

    Semantics restated: per generation g, parents come from K^n (g = 0) or from the previous
    generation's pool; children already known (permanent states, K^n, earlier generations) are
    not evaluated again.  The next pool is [fresh children] + [known states that some child hit,
    index >= 1, in index order], the latter paired with ``lpj_unique[index-1]`` exactly as the
    reference does (that pairing is only aligned when one permanent state precedes K^n;
    SURVEY Q5) -- the pool only steers parent selection of the following generation.
    """
    lpj0 = my_suff_stat["this_lpj"]
    kn = my_suff_stat["this_states"]
    background = bool(my_suff_stat["permanent"]["background"])  # eas.py:213-239: the last latent is not mutated
    select = my_suff_stat["parent_selection"]
    mutate = my_suff_stat["mutation_algorithm"]
    n_par, n_child = my_suff_stat["n_parents"], my_suff_stat["n_children"]
    n_gen, p_bf = my_suff_stat["n_generations"], my_suff_stat["bitflip_prob"]
    sparseness = model_params["piH"]
    S, H = kn.shape

    known = np.concatenate((my_suff_stat["incl"], kn), axis=0)
    known_lpj = lpj0
    out_s, out_l = [], []
    pool_s, pool_l = kn, lpj0
    for g in range(n_gen):
        n_pick = np.min([pool_s.shape[0], n_par]) if g else np.min([S, n_par])
        kids = mutate(select(pool_s, n_pick, pool_l)[:, :(H - 1 if background else H)], n_child, sparseness, p_bf)
        if background:
            kids = np.concatenate((kids, np.ones((kids.shape[0], 1), dtype=bool)), axis=1)
        n_known = known.shape[0]
        both = np.concatenate((known, kids), axis=0)
        _, first, inverse = np.unique(row_keys(both), return_index=True, return_inverse=True)
        fresh_idx = first[first >= n_known]
        fresh = both[fresh_idx]
        fresh_lpj = eval_lpj(fresh) if fresh_idx.size else np.zeros(0)
        group_hit = np.zeros(first.size, dtype=bool)
        group_hit[inverse[n_known:]] = True
        hit = np.flatnonzero(group_hit[inverse[1:n_known]]) + 1  # known rows (index >= 1) met by a child
        if fresh_idx.size:
            out_s.append(fresh)
            out_l.append(fresh_lpj)
        if fresh_idx.size + hit.size == 0:
            warnings.warn("No new and unique states. Skipping evolutionary loop.")
            break
        pool_s = np.concatenate((fresh, known[hit]), axis=0)
        pool_l = np.concatenate((fresh_lpj, known_lpj[hit - 1]))
        if fresh_idx.size:
            known = np.concatenate((known, fresh), axis=0)
            known_lpj = np.append(known_lpj, fresh_lpj)
    if not out_s:
        return np.zeros((0, H), dtype=bool), np.zeros(0)
    return np.concatenate(out_s, axis=0), np.concatenate(out_l)


def first_generation_candidates(kn, lpj, my_suff_stat, sparseness):
    """Generation-0 half of evolve_states without the evaluation: the not-yet-known children of
    one datapoint in lexicographic order.  Identical RNG consumption; used by the batched E-step
    (all datapoints are generated first, then evaluated in ONE kernel launch), which is
    stream-exact for n_generations == 1 because evaluation draws no random numbers."""
    S, H = kn.shape
    n_pick = np.min([S, my_suff_stat["n_parents"]])
    parents = my_suff_stat["parent_selection"](kn, n_pick, lpj)
    background = bool(my_suff_stat["permanent"]["background"])
    kids = my_suff_stat["mutation_algorithm"](parents[:, :(H - 1 if background else H)], my_suff_stat["n_children"],
                                              sparseness, my_suff_stat["bitflip_prob"])
    if background:
        kids = np.concatenate((kids, np.ones((kids.shape[0], 1), dtype=bool)), axis=1)
    known = np.concatenate((my_suff_stat["incl"], kn), axis=0)
    both = np.concatenate((known, kids), axis=0)
    _, first = np.unique(row_keys(both), return_index=True)
    return both[first[first >= known.shape[0]]]
