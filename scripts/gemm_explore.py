#!/usr/bin/env python3
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xmca_amd import _hip
h = _hip.Handle(0)
cases = [
  ("f32 NN 4096^3", 4096, 4096, 4096, np.float32, True, True, False, 1),
  ("f32 NT 4096^3", 4096, 4096, 4096, np.float32, True, False, False, 1),
  ("f32 TN 4096^3", 4096, 4096, 4096, np.float32, False, True, False, 1),
  ("f64 NN 4096^3", 4096, 4096, 4096, np.float64, True, True, False, 1),
  ("f64 NT 4096^3", 4096, 4096, 4096, np.float64, True, False, False, 1),
  ("f64 TN 4096^3", 4096, 4096, 4096, np.float64, False, True, False, 1),
  ("f64 NT gram-like 2944x2944x10000 full", 2944, 2944, 10000, np.float64, True, False, False, 0),
  ("f64 NT gram-like small K=2048 (L2/MALL resident)", 2944, 2944, 2048, np.float64, True, False, False, 1),
]
for name, M, N, K, dt, ak, bn, up, sp in cases:
    ms = h.bench_gemm(M, N, K, dt, ak, bn, up, sp, 5)
    fl = 2.0 * M * N * K
    peak = 157.3 if dt == np.float32 else 78.6
    print(json.dumps({"case": name, "ms": ms, "TF": fl / ms / 1e9, "frac": fl / ms / 1e9 / peak}))
