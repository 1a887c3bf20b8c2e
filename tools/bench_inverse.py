"""Accuracy and device time of the M-step's H x H Gauss-Jordan inverse (evoamd_inverse)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from evo_amd.engine import Engine

rng = np.random.default_rng(5)
for n in [int(a) for a in sys.argv[1:]] or [40, 100, 128, 200, 256, 500, 512, 1000, 1024]:
    eng = Engine()
    eng.configure("bsc", N=64, D=8, H=n, S=4, Cmax=4)
    X = rng.standard_normal((n, 3 * n))
    A = X @ X.T / (3 * n) + 0.05 * np.eye(n)       # SPD like the M-step's moment matrices
    B = rng.standard_normal((n, n))                # general: forces interchanges
    B[0, 0] = 0.0
    best = 1e9
    for rep in range(4):
        Ai, Bi, ms = eng.inverse(A, B)
        best = min(best, ms)
    # both SPD (the M-step's case): SPD block path; then the same through the pivoted path
    A2 = A + np.diag(rng.uniform(0.0, 1.0, n))
    Ai_s, A2i_s, ms_spd = min((eng.inverse(A, A2) for _ in range(4)), key=lambda r: r[2])
    eng.set_option("inverse_spd", 0)
    Ai_p, A2i_p, ms_piv = min((eng.inverse(A, A2) for _ in range(4)), key=lambda r: r[2])
    eng.set_option("inverse_spd", 1)
    eng.set_option("inverse_block", 16)
    ms_16 = min((eng.inverse(A, A2) for _ in range(4)), key=lambda r: r[2])[2]
    eng.set_option("inverse_block", 0)
    print("n %5d  SPD pair with 16-column steps %8.3f ms" % (n, ms_16))
    print("n %5d  SPD pair: block path %8.3f ms  pivoted %8.3f ms   |AiA-I| %.2e %.2e   spd vs pivoted rel %.2e"
          % (n, ms_spd, ms_piv, np.abs(Ai_s @ A - np.eye(n)).max(), np.abs(A2i_s @ A2 - np.eye(n)).max(),
             np.abs(Ai_s - Ai_p).max() / np.abs(Ai_p).max()), flush=True)
    Ai1, _, ms1 = eng.inverse(A)
    ea = np.abs(Ai @ A - np.eye(n)).max()
    eb = np.abs(Bi @ B - np.eye(n)).max()
    ra = np.abs(Ai - np.linalg.inv(A)).max() / np.abs(Ai).max()
    rb = np.abs(Bi - np.linalg.inv(B)).max() / np.abs(Bi).max()
    print("n %5d  two matrices %8.3f ms  one %8.3f ms   |AiA-I| %.2e  |BiB-I| %.2e  rel vs numpy %.2e %.2e  single==double %s"
          % (n, best, ms1, ea, eb, ra, rb, np.array_equal(Ai1, Ai)), flush=True)
    eng.close() if hasattr(eng, "close") else None
