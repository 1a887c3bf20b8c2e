"""Compact, order-sensitive summaries of arrays for fixtures whose raw outputs would be tens of MB
(EM steps at the BASELINE shapes: W alone is 1-2 MB, K^n 20-100 MB).  Shared by the generator
(make_golden.py, run against the reference) and by the tests (run against the oracle / the GPU).

  sketch(A)        0-d / 1-d: the values themselves; 2-d (r, c): [A v1, A v2, u1 A, u2 A] with fixed
                   pseudo-random probe vectors -- 2r + 2c doubles that move with every entry of A.
  state_hashes(ss) one 64-bit hash per datapoint of its bit-packed K^n rows (bit-exactness is an
                   equality test, so a hash carries it).
  lpj_rows(lpj)    per datapoint [sum, max, positively weighted sum] of its lpj row.
"""
import hashlib

import numpy as np


def _probe(n, which):
    # fixed probe vectors, entries in [0.5, 1.5]: positive, so sums of same-signed entries do not cancel
    rs = np.random.RandomState(7919 + 104729 * which + n)
    return 0.5 + rs.random_sample(n)


def sketch(A):
    A = np.asarray(A, dtype=np.float64)
    if A.ndim <= 1:
        return A.copy()
    assert A.ndim == 2
    r, c = A.shape
    return np.concatenate((A @ _probe(c, 0), A @ _probe(c, 1), _probe(r, 2) @ A, _probe(r, 3) @ A))


def state_hashes(ss_bool):
    """uint64 (N,): first 8 bytes of sha1 over each datapoint's packed (S, ceil(H/8)) rows."""
    packed = np.packbits(np.asarray(ss_bool, dtype=bool), axis=-1)
    out = np.empty(packed.shape[0], dtype=np.uint64)
    for n in range(packed.shape[0]):
        out[n] = np.frombuffer(hashlib.sha1(packed[n].tobytes()).digest()[:8], dtype=np.uint64)[0]
    return out


def ragged_hashes(batches):
    """uint64 per entry of a list of bool (C_n, H) arrays (the ragged candidate batches)."""
    out = np.empty(len(batches), dtype=np.uint64)
    for n, b in enumerate(batches):
        p = np.packbits(np.asarray(b, dtype=bool), axis=-1)
        out[n] = np.frombuffer(hashlib.sha1(p.tobytes()).digest()[:8], dtype=np.uint64)[0]
    return out


def lpj_rows(lpj):
    lpj = np.asarray(lpj, dtype=np.float64)
    w = _probe(lpj.shape[1], 4)
    return np.stack((lpj.sum(axis=1), lpj.max(axis=1), lpj @ w), axis=1)


def array_sha1(A):
    return hashlib.sha1(np.ascontiguousarray(A).tobytes()).hexdigest()


def bars_recovered(W, W_gen, thresh=0.9):
    """Number of generating fields (columns of W_gen, D x H) matched by DISTINCT learned columns of W with a cosine
    similarity above `thresh` (greedy over the similarity matrix, best pairs first) -- the bars test's success measure
    (examples/bars-test: learned W against the ground-truth bars)."""
    W, W_gen = np.asarray(W, dtype=np.float64), np.asarray(W_gen, dtype=np.float64)
    a = W / np.maximum(np.linalg.norm(W, axis=0, keepdims=True), 1e-300)
    b = W_gen / np.maximum(np.linalg.norm(W_gen, axis=0, keepdims=True), 1e-300)
    sim = np.abs(b.T @ a)  # (H_gen, H); the sign of a spike-and-slab field is not identified (mu = 0, z -> -z)
    found = 0
    sim = sim.copy()
    for _ in range(sim.shape[0]):
        i, j = np.unravel_index(np.argmax(sim), sim.shape)
        if sim[i, j] < thresh:
            break
        found += 1
        sim[i, :] = -np.inf
        sim[:, j] = -np.inf
    return found
