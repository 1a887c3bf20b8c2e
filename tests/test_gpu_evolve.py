"""Device-side evolutionary algorithm (rng="device": evoamd_evolve_randflip / evoamd_evolve_states) against the
reference's samplers.  The device uses a counter-based generator, so parity is DISTRIBUTIONAL: every datapoint of a
batch of identical datapoints is one independent run of evolve_states (eas.py:153-313), and the empirical laws of
10^5 runs are compared with (i) the exact law of the reference's operator where it has a closed form and (ii) the
empirical law of the oracle's restatement of the same operator driven by np.random (oracle = checker)."""
import itertools

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

Z_MAX = 5.0  # |z| bound of every frequency comparison (fixed seeds: deterministic outcome, 5 sigma leaves room)


@pytest.fixture(scope="module")
def engine():
    from evo_amd.engine import Engine
    eng = Engine()
    yield eng
    eng.close()


def _setup(engine, states, N, S_perm=0, Cmax=8, seed=0, D=6):
    """N identical datapoints with K^n = states (S, H) under a fixed random BSC Theta; returns the lpj row."""
    S, H = states.shape
    rng = np.random.RandomState(seed)
    W = rng.normal(size=(D, H))
    y = rng.normal(size=D)
    engine.configure("bsc", N, D, H, S, S_perm, Cmax)
    engine.upload_data(np.tile(y, (N, 1)))
    engine.upload_states(np.tile(states[None], (N, 1, 1)))
    engine.set_params_bsc(W, 0.2, 1.3)
    engine.lpj_resident()
    lpj = engine.download_lpj()
    assert np.all(lpj == lpj[0])
    return W, y, lpj[0]


def _fit_probs(lpj):
    f = lpj - 2 * min(lpj.min(), 0.0)  # eas.py:139
    return f / f.sum()


def _inclusion_probs(p, n_draw):
    """Exact inclusion probabilities of successive sampling without replacement (what np.random.choice(...,
    replace=False, p=p) implements: duplicates of a batch are redrawn from the renormalised rest)."""
    inc = np.zeros(len(p))
    for seq in itertools.permutations(range(len(p)), n_draw):
        pr, rest = 1.0, 1.0
        for i in seq:
            pr *= p[i] / rest
            rest -= p[i]
        inc[list(seq)] += pr
    return inc


def _z(count, n, prob):
    count, prob = np.asarray(count, dtype=float), np.asarray(prob, dtype=float)
    return (count - n * prob) / np.sqrt(np.maximum(n * prob * (1.0 - prob), 1e-300))


def _chi2_ok(obs, exp, slack=1.6):
    """Pearson statistic against the multinomial expectation: fixed seed, so a generous multiple of the degrees of
    freedom (mean df, sd sqrt(2 df)) is a deterministic pass / fail."""
    obs, exp = np.asarray(obs, float), np.asarray(exp, float)
    df = len(obs) - 1
    stat = ((obs - exp) ** 2 / exp).sum()
    assert stat < slack * df + 5 * np.sqrt(2.0 * df), (stat, df)
    return stat


def test_fitparents_and_randflip_law(engine):
    """fitparents (eas.py:138-146) + randflip (eas.py:10-43): first-drawn parent ~ p, inclusion frequencies =
    successive sampling, flipped position uniform over H -- for the fast kernel and the general kernel, and the
    oracle's np.random operators obey the same law."""
    from oracle import evo_oracle as orc
    H, S, P, N = 32, 8, 3, 120000
    states = np.zeros((S, H), dtype=bool)
    for j in range(S):
        states[j, 4 * j:4 * j + 3] = True  # pairwise Hamming distance 6: a child names its parent and its flip
    W, y, lpj = _setup(engine, states, N, Cmax=P)
    p = _fit_probs(lpj)
    inc = _inclusion_probs(p, P)
    assert p.max() / p.min() > 1.3  # the law is visibly non-uniform
    # the oracle's fitparents on np.random: same law
    np.random.seed(5)
    n_or = 40000
    cnt_or = np.zeros(S)
    first_or = np.zeros(S)
    ids = np.arange(S)[:, None]
    for _ in range(n_or):
        sel = orc.fitparents(ids, P, lpj)[:, 0]
        cnt_or[sel] += 1
        first_or[sel[0]] += 1
    assert np.abs(_z(cnt_or, n_or, inc)).max() < Z_MAX
    _chi2_ok(first_or, n_or * p)

    def analyse(cand, counts):
        par = np.full((N, cand.shape[1]), -1)
        flip = np.full((N, cand.shape[1]), -1)
        for j in range(S):
            d = cand ^ states[j][None, None, :]
            one = d.sum(axis=2) == 1
            par[one] = j
            flip[one] = np.argmax(d, axis=2)[one]
        valid = np.arange(cand.shape[1])[None, :] < counts[:, None]
        assert np.all(par[valid] >= 0)
        return par, flip, valid

    # fast kernel: slot i holds the child of the i-th drawn parent
    engine.evolve_randflip(P, 1, 1234, True)
    cand, counts, _ = engine.download_candidates()
    assert np.all(counts == P)
    par, flip, valid = analyse(cand, counts)
    _chi2_ok(np.bincount(par[:, 0], minlength=S), N * p)
    cnt = np.array([(par == j).any(axis=1).sum() for j in range(S)])
    assert np.abs(_z(cnt, N, inc)).max() < Z_MAX, (cnt / N, inc)
    assert np.all(np.sort(par, axis=1)[:, 1:] != np.sort(par, axis=1)[:, :-1])  # without replacement
    _chi2_ok(np.bincount(flip[valid], minlength=H), np.full(H, valid.sum() / H))
    # general kernel (lexicographic output order: inclusion frequencies and flips)
    engine.evolve_states("randflip", P, 1, 1, 4321, True)
    cand, counts, _ = engine.download_candidates()
    assert np.all(counts == P)
    par, flip, valid = analyse(cand, counts)
    cnt = np.array([(par == j).any(axis=1).sum() for j in range(S)])
    assert np.abs(_z(cnt, N, inc)).max() < Z_MAX, (cnt / N, inc)
    _chi2_ok(np.bincount(flip[valid], minlength=H), np.full(H, valid.sum() / H))
    # rows come out in np.unique's order (unsigned big-endian words == lexicographic over latents)
    keys = np.packbits(cand[:2000], axis=-1)
    for n in range(2000):
        rows = [bytes(keys[n, c]) for c in range(counts[n])]
        assert rows == sorted(rows)
    # randparents (eas.py:149-150): uniform inclusion P / S
    engine.evolve_states("randflip", P, 1, 1, 99, False)
    cand, counts, _ = engine.download_candidates()
    par, _, _ = analyse(cand, counts)
    cnt = np.array([(par == j).any(axis=1).sum() for j in range(S)])
    assert np.abs(_z(cnt, N, np.full(S, P / S))).max() < Z_MAX
    # distinct flips for the children of one parent (eas.py:31-33): two children of one parent never coincide
    engine.configure("bsc", N, 6, H, S, 0, 4)
    _setup(engine, states, N, Cmax=4)
    engine.evolve_randflip(2, 2, 77, True)
    cand, counts, _ = engine.download_candidates()
    par, flip, valid = analyse(cand, counts)
    assert np.all(par[:, 0] == par[:, 1]) and np.all(flip[:, 0] != flip[:, 1])


@pytest.mark.parametrize("k", [0, 1, 3, 7])
def test_sparseflip_law(engine, k):
    """sparseflip (eas.py:46-100): a 0 flips with p_0, a 1 with p_1 = alpha p_0, independently; a child equal to
    its parent is dropped by the de-duplication, so the survivors follow the law conditioned on >= 1 flip."""
    from oracle import evo_oracle as orc
    H, N = 40, 150000
    sparseness, p_bf = 2.5, 0.08
    parent = np.zeros((1, H), dtype=bool)
    parent[0, [3, 11, 12, 20, 29, 30, 38][:k]] = True
    _setup(engine, parent, N, Cmax=1)
    # the reference's probabilities (eas.py:75-83)
    s_abs = float(k)
    eps = 1e-100
    alpha = (H - s_abs) * ((H * p_bf) - (sparseness - s_abs)) / ((sparseness - s_abs + H * p_bf) * s_abs + eps)
    p0 = (H * p_bf) / (H + (alpha - 1.0) * s_abs + eps)
    p1 = alpha * p0
    pbit = np.where(parent[0], p1, p0)
    pbit = np.clip(pbit, 0.0, 1.0)
    none = np.prod(1.0 - pbit)  # child == parent
    # the oracle's operator on np.random: raw children follow pbit
    np.random.seed(3)
    n_or = 30000
    kids = orc.sparseflip(np.repeat(parent, n_or, axis=0), 1, sparseness, p_bf)
    fl = kids ^ parent
    assert np.abs(_z(fl.sum(axis=0), n_or, pbit)).max() < Z_MAX
    # device
    engine.evolve_states("sparseflip", 1, 1, 1, 2024 + k, False, sparseness, p_bf)
    cand, counts, _ = engine.download_candidates()
    assert abs(_z((counts == 1).sum(), N, 1.0 - none)) < Z_MAX
    got = cand[counts == 1, 0] ^ parent
    assert got.any(axis=1).all()
    n_s = got.shape[0]
    assert np.abs(_z(got.sum(axis=0), n_s, pbit / (1.0 - none))).max() < Z_MAX
    # independence check on one pair of zero bits: P(both | survive) = p0^2 / (1 - none)
    z0 = np.flatnonzero(~parent[0])[:2]
    both = (got[:, z0[0]] & got[:, z0[1]]).sum()
    pz = min(max(p0, 0.0), 1.0)  # the reference compares a uniform draw with p_0: negative never flips, > 1 always
    assert abs(_z(both, n_s, pz * pz / (1.0 - none))) < Z_MAX


def test_cross_law(engine):
    """cross (eas.py:103-125): both tail exchanges of every parent pair at a cut uniform on 1..H-1."""
    H, N = 24, 100000
    states = np.zeros((2, H), dtype=bool)
    states[0] = True  # all ones x all zeros: the children are 1^cp 0^(H-cp) and 0^cp 1^(H-cp)
    _setup(engine, states, N, Cmax=2)
    engine.evolve_states("cross", 2, 1, 1, 31, False)
    cand, counts, _ = engine.download_candidates()
    assert np.all(counts == 2)
    k = cand.sum(axis=2)
    assert np.all(k[:, 0] + k[:, 1] == H)
    head = np.where(cand[:, :, 0], k, H - k)  # the cut, from either child
    assert np.all(head[:, 0] == head[:, 1])
    cp = head[:, 0]
    assert cp.min() == 1 and cp.max() == H - 1
    ones_first = cand[np.arange(N), np.argmax(cand[:, :, 0], axis=1)]
    assert np.all(ones_first == (np.arange(H)[None, :] < cp[:, None]))
    _chi2_ok(np.bincount(cp, minlength=H)[1:], np.full(H - 1, N / (H - 1.0)))
    # cross_randflip: every child is one flip away from a pure crossing of the same cut family
    engine.evolve_states("cross_randflip", 2, 1, 1, 32, False)
    cand, counts, _ = engine.download_candidates()
    pure = np.concatenate([np.arange(H)[None, :] < np.arange(1, H)[:, None], np.arange(H)[None, :] >= np.arange(1, H)[:, None]])
    for n in range(0, 3000):
        for c in range(counts[n]):
            assert ((pure ^ cand[n, c][None, :]).sum(axis=1) == 1).any()


def _oracle_runs(states, lpj, ea, piH, eval_lpj, n_runs, seed):
    from oracle import evo_oracle as orc
    np.random.seed(seed)
    out = []
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for _ in range(n_runs):
            st, lp = orc.evolve_states(states.copy(), lpj.copy(), ea, piH, eval_lpj)
            out.append(frozenset(tuple(np.flatnonzero(r)) for r in st))
    return out


@pytest.mark.parametrize("mutation,S_perm,n_gen,n_par,n_child", [
    ("randflip", 0, 3, 2, 2), ("randflip", 1, 3, 2, 2), ("sparseflip", 0, 2, 3, 2), ("cross_randflip", 0, 2, 3, 2),
    ("cross_sparseflip", 1, 2, 3, 2), ("cross", 0, 2, 3, 2)])
def test_generations_match_oracle_law(engine, mutation, S_perm, n_gen, n_par, n_child):
    """n_generations > 1 incl. the next-generation parent pools of eas.py:278-293 (with S_perm = 0 a
    duplicated known state enters the pool with its predecessor's lpj): on a tiny latent space (H = 5: children
    collide with known states all the time) the device's law of the returned set of new states -- which state is
    returned how often, how many states a run returns -- matches the oracle's evolve_states on np.random."""
    from oracle import evo_oracle as orc
    H, S, N = 5, 4, 200000
    rng = np.random.RandomState(11)
    all_states = np.array(list(itertools.product([False, True], repeat=H)))[1:]  # without the all-zero state
    states = all_states[rng.permutation(len(all_states))[:S]]
    crossing = "cross" in mutation
    if crossing:
        n_child = n_par - 1  # init_states forces it (variational/utils.py:202-207)
    per_gen = n_par * (n_par - 1) if crossing else n_par * n_child
    W, y, lpj_row = _setup(engine, states, N, S_perm=S_perm, Cmax=per_gen * n_gen, seed=21)
    lpj = lpj_row[S_perm:]
    theta = {"W": W, "pi": 0.2, "sigma": 1.3}
    orc.bsc_precompute(theta, W.shape[0], H)
    cnt = orc.new_counters()
    ea = {"permanent": {"background": False, "allzero": S_perm == 1, "singletons": False},
          "incl": np.zeros((S_perm, H), dtype=bool), "n_parents": n_par, "n_children": n_child, "n_generations": n_gen,
          "parent_selection": orc.fitparents, "mutation_algorithm": orc.MUTATION[mutation], "bitflip_prob": 0.15}
    runs = _oracle_runs(states, lpj, ea, theta["piH"], lambda st: orc.bsc_lpj(theta, st, y, cnt), 4000, 7)
    engine.evolve_states(mutation, n_par, n_child, n_gen, 555, True, theta["piH"], 0.15)
    cand, counts, clpj = engine.download_candidates()
    # what comes back is new, unique, and carries its own lpj
    known = {tuple(np.flatnonzero(r)) for r in states} | ({()} if S_perm else set())
    for n in range(300):
        rows = [tuple(np.flatnonzero(cand[n, c])) for c in range(counts[n])]
        assert len(set(rows)) == len(rows) and not (set(rows) & known)
        np.testing.assert_allclose(clpj[n, :counts[n]], orc.bsc_lpj(theta, cand[n, :counts[n]], y, cnt), rtol=1e-9)
    # law of the returned set: per-state inclusion frequency and size histogram, device vs oracle (two samples)
    code = (cand * (1 << np.arange(H))[None, None, :]).sum(axis=2)
    valid = np.arange(cand.shape[1])[None, :] < counts[:, None]
    n_or = len(runs)
    for st in all_states.tolist() + [[False] * H]:
        key = tuple(np.flatnonzero(st))
        c_or = sum(1 for r in runs if key in r)
        c_dev = int(((code == sum(1 << h for h in key)) & valid).any(axis=1).sum())
        pool = (c_or + c_dev) / float(n_or + N)
        if pool in (0.0, 1.0):
            assert c_or / n_or == c_dev / N
            continue
        z = (c_dev / N - c_or / n_or) / np.sqrt(pool * (1 - pool) * (1.0 / N + 1.0 / n_or))
        assert abs(z) < Z_MAX, (key, c_dev / N, c_or / n_or, z)
    sizes_or = np.bincount([len(r) for r in runs], minlength=per_gen * n_gen + 1) / n_or
    sizes_dev = np.bincount(counts, minlength=per_gen * n_gen + 1) / N
    for a, b in zip(sizes_dev, sizes_or):
        pool = (a * N + b * n_or) / (N + n_or)
        if 0.0 < pool < 1.0:
            assert abs(a - b) / np.sqrt(pool * (1 - pool) * (1.0 / N + 1.0 / n_or)) < Z_MAX, (sizes_dev, sizes_or)


@pytest.mark.parametrize("algo", ["ebsc", "es3c"])
@pytest.mark.parametrize("mutation,n_gen", [("sparseflip", 1), ("cross_randflip", 1), ("randflip", 2), ("cross_sparseflip", 2)])
def test_device_ea_through_the_model(engine, algo, mutation, n_gen):
    """model.step with rng='device' for the operators / generation counts that used to raise: K^n stays
    duplicate-free, lpj rows equal a re-evaluation of the states, F never decreases for fixed Theta."""
    from evo_amd.models import BSC, SSSC
    from evo_amd.variational import init_states
    rng = np.random.RandomState(5)
    D, H, S, N = 24, 70, 20, 400
    Y = rng.normal(size=(N, D))
    my_data = {"y": Y, "x_infr": np.ones_like(Y, dtype=bool)}
    np.random.seed(3)
    cls = BSC if algo == "ebsc" else SSSC
    model = cls(D, H, S, to_learn=[], rng="device", sync_host=True, engine=engine, seed=11)
    theta = model.check_params(model.standard_init(my_data))
    suff = init_states(N, S, H, "fit", mutation, 4, 2, n_gen, bitflip_prob=0.05)
    Fs = []
    for _ in range(5):
        F, nu, nsub, theta = model.step(theta, suff, my_data)
        Fs.append(F)
        assert 0 <= nsub <= nu
    assert all(b >= a - 1e-12 for a, b in zip(Fs, Fs[1:])), Fs
    assert Fs[-1] > Fs[0]
    for n in range(0, N, 7):
        assert np.unique(np.packbits(suff["ss"][n], axis=-1), axis=0).shape[0] == S
    lpj_after = suff["lpj"].copy()
    engine.lpj_resident()
    np.testing.assert_allclose(engine.download_lpj(), lpj_after, rtol=1e-12)
