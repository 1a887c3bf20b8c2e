"""Does the M-step contraction's time depend on the DATA?  Same launch (K = 100k, [Y|Es|Ez]^T Ez shape, symmetric block),
operands all zero / 45 % non-zero / dense random: the instruction stream is identical, only the switching activity
(power, hence clock) differs.     python tools/bench_gemm_data.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from evo_amd.engine import Engine
K, M, Nc, sym = 100000, 1280, 512, 768
rng = np.random.default_rng(0)
eng = Engine(); eng.configure("bsc", 8, 4, 8, 4, 0, 4)
fl = 2.0 * K * (M * Nc - 6 * 128 * 128)  # executed: 34 of 40 tiles
for name, dens in (("all zero", 0.0), ("45 % non-zero", 0.45), ("dense random", 1.0), ("all zero again", 0.0)):
    A = rng.standard_normal((K, M)) * (rng.random((K, M)) < dens)
    B = np.ascontiguousarray(A[:, sym:])
    eng.gemm_tn(A, B, sym)
    eng.timing(["gemm_f64"]); eng.timing_reset()
    for _ in range(4):
        eng.gemm_tn(A, B, sym)
    ms, n = eng.kernel_time_ms("gemm_f64")
    print("%-16s %.3f ms per launch (contraction + reduce + mirror), %.1f TFLOP/s executed" % (name, ms, fl / ms / 1e9), flush=True)
