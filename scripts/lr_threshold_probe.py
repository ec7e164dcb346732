"""Which matrices profit from the Cholesky LR step?  Gram matrices of synthetic climate-like fields (smooth spatial
modes with power-law variance, AR(1) in time, white measurement noise): sweeps / time with the step forced on and off
and the diagonal-spread statistic the solver decides on (run once per XMCA_JACOBI_LR setting)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from xmca_amd import _hip

h = _hip.default_handle()
T, N, k = 2920, 4000, 400
rng = np.random.default_rng(1)
x = np.linspace(0, 1, N)
modes = np.cos(np.pi * np.arange(1, k + 1)[:, None] * x[None, :])
for alpha in (1.0, 2.0, 3.0):
    for noise in (1e-1, 1e-2, 1e-3):
        pcs = rng.standard_normal((T, k))
        for t in range(1, T):
            pcs[t] = 0.8 * pcs[t - 1] + 0.6 * pcs[t]
        X = (pcs * np.arange(1, k + 1) ** (-alpha / 2)) @ modes + noise * rng.standard_normal((T, N))
        X -= X.mean(axis=0)
        A = X @ X.T
        h.eigh(A[:300, :300])
        t0 = time.perf_counter(); lam, U = h.eigh(A); dt = time.perf_counter() - t0
        print("alpha %.0f noise %.0e: sweeps %2d lr %d  %.1f ms  orth %.1e" % (alpha, noise, h.last_eigh_info["sweeps"], h.last_eigh_info["lr_step"],
              dt * 1e3, np.abs(U.T @ U - np.eye(T)).max()), flush=True)
