#!/usr/bin/env python3
"""Gram GEMM micro-benchmark through the C ABI (hipEvents on the library stream): TF/s vs the MFMA peak."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xmca_amd import _hip
h = _hip.Handle(0)
rng = np.random.default_rng(0)
for name, T, N, dt, peak in [("C2 f64", 2920, 10000, np.float64, 78.6), ("C3 f64", 5000, 20000, np.float64, 78.6),
                             ("C5/8 f32", 1200, 129600, np.float32, 157.3), ("C2 f32", 2920, 10000, np.float32, 157.3)]:
    X = rng.standard_normal((T, N)).astype(dt)
    h.set_field(0, X)
    g = h.bench_gram(0, 5)
    tf = g["flops"] / (g["avg_ms"] * 1e-3) / 1e12
    print(json.dumps({"case": name, "ms": g["avg_ms"], "TF": tf, "frac": tf / peak}))
