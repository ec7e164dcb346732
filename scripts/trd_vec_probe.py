#!/usr/bin/env python3
"""Eigen-decompositions with vectors through the tridiagonal route (csrc/tridiag_vec.h) against numpy: eigenvalues, orthonormality,
residuals, device time; plus spectra with clusters, which must come back through the Jacobi fallback.
usage: trd_vec_probe.py [n,n,...]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xmca_amd import _hip

sizes = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [70, 200, 1000, 2920]
h = _hip.Handle(0)


def check(tag, G, reps=2):
    n = G.shape[0]
    ref = np.linalg.eigvalsh(G)[::-1]
    sc = max(abs(ref[0]), abs(ref[-1]))
    for _ in range(reps):
        h.reset_timings()
        t0 = time.perf_counter()
        lam, U = h.eigh(G)
        wall = time.perf_counter() - t0
    tm = h.timings()
    sel = np.r_[0:min(40, n), n // 2:min(n // 2 + 40, n), max(n - 40, 0):n]
    Us = U[:, sel]
    orth = float(np.max(np.abs(Us.conj().T @ U - np.eye(n)[sel])))
    res = float(np.max(np.linalg.norm(G @ Us - Us * lam[sel], axis=0)) / sc)
    rec = dict(tag=tag, n=n, cplx=bool(np.iscomplexobj(G)), err=float(np.max(np.abs(lam - ref)) / sc), orth=orth, res=res,
               ms=tm.get("eigh_vectors"), wall_ms=wall * 1e3, info=h.last_eigh_info)
    print(json.dumps(rec), flush=True)


for n in sizes:
    for cplx in (False, True):
        n_ = 2501 if (cplx and n > 2600) else n
        rng = np.random.default_rng(n_)
        N = 3 * n_
        X = (rng.standard_normal((n_, 20)) * np.linspace(10, 1, 20)) @ rng.standard_normal((20, N)) + rng.standard_normal((n_, N))
        if cplx:
            X = X + 1j * ((rng.standard_normal((n_, 20)) * np.linspace(6, 1, 20)) @ rng.standard_normal((20, N)) + rng.standard_normal((n_, N)))
        X -= X.mean(axis=0)
        check("c2like", X @ X.conj().T)
n = min(sizes[-1], 600)
rng = np.random.default_rng(5)
Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
check("graded12", (Q * np.logspace(0, -12, n)) @ Q.T, reps=1)
Y = rng.standard_normal((n, 20))
check("lowrank", Y @ Y.T, reps=1)
check("identity", np.eye(n), reps=1)
check("diag", np.diag(np.arange(1.0, n + 1)), reps=1)
