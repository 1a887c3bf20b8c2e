"""Device time of the M-step contraction C = A^T B (evoamd_gemm_tn) at the c4/100k shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from evo_amd.engine import Engine
K, M, Nc = [int(a) for a in sys.argv[1:4]] if len(sys.argv) > 3 else (100000, 1280, 512)
sym = int(sys.argv[4]) if len(sys.argv) > 4 else -1
rng = np.random.default_rng(0)
A = rng.standard_normal((K, M)); B = rng.standard_normal((K, Nc))
eng = Engine(); eng.configure("bsc", 8, 4, 8, 4, 0, 4)
eng.gemm_tn(A, B, sym)
eng.timing(["gemm_f64"]); eng.timing_reset()
for _ in range(3):
    C = eng.gemm_tn(A, B, sym)
ms, n = eng.kernel_time_ms("gemm_f64")
fl = 2.0 * K * M * Nc
print("K %d M %d Nc %d sym %d: %.3f ms per launch, %.1f TFLOP/s nominal (%d launches)" % (K, M, Nc, sym, ms, fl / ms / 1e9, n))
