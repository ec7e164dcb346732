import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xmca_amd import _hip
h = _hip.Handle(0)
for (M, K, dt, name) in [(1200, 1036800, np.float32, "C5 Gram f32"), (2920, 10000, np.float64, "C2 Gram f64"), (5000, 20000, np.float32, "C3 Gram f32"),
                         (5000, 20000, np.float64, "C3 Gram f64")]:
    for (akf, bnf, lab) in [(True, False, "NT (field as stored, T x N)"), (False, True, "TN (field transposed, N x T)")]:
        for splits in (0,):
            ms = h.bench_gemm(M, M, K, dt, a_kfast=akf, b_nfast=bnf, upper_only=True, splits=splits, reps=5)
            fl = M * (M + 1.0) * K
            peak = 78.6 if dt == np.float64 else 157.3
            print("%-14s %-30s %.3f ms  %.1f TF  %.1f %%" % (name, lab, ms, fl / ms / 1e9, 100 * fl / ms / 1e9 / peak))
