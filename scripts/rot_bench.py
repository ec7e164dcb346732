#!/usr/bin/env python3
"""Varimax/Promax timing on synthetic loadings (C2: N=1e4,p=10 real; C3: N=35000,p=20 complex)."""
import os, sys, json, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xmca_amd import _hip
h = _hip.Handle(0)
rng = np.random.default_rng(0)
CASES = [("C2-like", 10000, 10, False, 1), ("C3-like", 35000, 20, True, 4), ("C5-like", 1036800, 10, False, 1)]
if len(sys.argv) > 1:
    CASES = [c for c in CASES if c[0] in sys.argv[1:]]
for name, N, p, cplx, power in CASES:
    L = 0.2 * rng.standard_normal((N, p))
    w = N // p
    for j in range(p):
        L[j * w:(j + 1) * w, j] += np.hanning(w) * (3 - 0.1 * j)
    if cplx:
        L = L * np.exp(1j * rng.uniform(0, 0.5, (N, 1))) + 0.05j * rng.standard_normal((N, p))
        Q, _ = np.linalg.qr(rng.standard_normal((p, p)) + 1j * rng.standard_normal((p, p)))
    else:
        Q, _ = np.linalg.qr(rng.standard_normal((p, p)))
    L = L @ Q
    h.rotate_loadings(L, n_left=N // 2, power=power)
    h.reset_timings()
    t0 = time.perf_counter()
    out = h.rotate_loadings(L, n_left=N // 2, power=power)
    dt = time.perf_counter() - t0
    tm = h.timings()
    print(json.dumps({"case": name, "iters": out["n_iter"], "wall_ms": 1e3 * dt, "varimax_ms": tm.get("varimax"),
                      "us_per_iter": 1e3 * tm.get("varimax") / out["n_iter"], "promax_ms": tm.get("promax")}))
