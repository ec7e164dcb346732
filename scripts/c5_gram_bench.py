#!/usr/bin/env python3
"""The covariance GEMM of BASELINE configs[4] (Gram matrix of a T = 1200 x N = 1 036 800 float32 field) on device-resident random
operands: ms and fraction of the f32 MFMA peak, tile x K-slab schedule (default) against stream-K (XMCA_NT_SLABS=0 in a
second process), plus the C2 / C3 Gram shapes."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xmca_amd import _hip
h = _hip.Handle(0)
cases = [("C5 f32", 1200, 1036800, np.float32, 157.3), ("C3 f32", 5000, 20000, np.float32, 157.3),
         ("C5-scaled f32", 1200, 41472, np.float32, 157.3), ("C2 f64", 2920, 10000, np.float64, 78.6)]
if len(sys.argv) > 1:
    cases = cases[:int(sys.argv[1])]
for name, M, K, dt, peak in cases:
    ms = h.bench_gemm(M, M, K, dt, True, False, True, 0, 3)
    fl = float(M) * (M + 1) * K
    print(json.dumps({"case": name, "ms": ms, "TF": fl / ms / 1e9, "frac": fl / ms / 1e9 / peak, "slabs": os.environ.get("XMCA_NT_SLABS", "1"), "count": os.environ.get("XMCA_NT_SLAB_COUNT")}))
