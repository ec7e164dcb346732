#!/usr/bin/env python3
"""Eigenvalues-only solves through the tridiagonal route (csrc/tridiag.h) against numpy.linalg.eigvalsh, with device times.
usage: trd_probe.py [n,n,...]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xmca_amd import _hip

sizes = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [5, 64, 193, 300, 1000, 2920]
h = _hip.Handle(0)
out = []
for n in sizes:
    for cplx in (False, True):
        if cplx and n > 2600:
            n_ = 2501
        else:
            n_ = n
        rng = np.random.default_rng(n_)
        N = 3 * n_
        X = (rng.standard_normal((n_, 20)) * np.linspace(10, 1, 20)) @ rng.standard_normal((20, N)) + rng.standard_normal((n_, N))
        if cplx:
            X = X + 1j * rng.standard_normal((n_, N))
        X -= X.mean(axis=0)
        G = X @ X.conj().T
        ref = np.linalg.eigvalsh(G)[::-1]
        rec = {"n": n_, "cplx": cplx}
        lam, _ = h.eigh(G, vectors=False)          # warm-up (allocations)
        h.reset_timings()
        t0 = time.perf_counter()
        lam, _ = h.eigh(G, vectors=False)
        wall = time.perf_counter() - t0
        tm = h.timings()
        rec.update(err=float(np.max(np.abs(lam - ref)) / ref[0]), ms=tm.get("eigh_values"), wall_ms=wall * 1e3, info=h.last_eigh_info, trd={k: round(v, 3) for k, v in tm.items() if k.startswith("trd")})
        out.append(rec)
        print(json.dumps(rec), flush=True)
