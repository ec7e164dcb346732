"""CPU experiment: sweeps of a parallel-order cyclic Jacobi on G versus on M = R R^T (G = R^T R, one Cholesky LR step)."""
import sys
import numpy as np

def sweeps_to_converge(G, tol=1e-10, max_sweeps=60):
    n = G.shape[0]
    G = G.copy()
    scale = np.abs(np.diag(G)).max()
    idx = list(range(n))
    hist = []
    for sweep in range(max_sweeps):
        for r in range(n - 1):
            p = np.array(idx[: n // 2]); q = np.array(idx[n // 2:][::-1])
            lo = np.minimum(p, q); hi = np.maximum(p, q)
            gpp = G[lo, lo]; gqq = G[hi, hi]; gpq = G[lo, hi]
            act = np.abs(gpq) > 1e-300
            tau = np.where(act, (gqq - gpp) / (2 * np.where(act, gpq, 1.0)), 0.0)
            t = np.where(act, np.sign(tau + (tau == 0)) / (np.abs(tau) + np.sqrt(1 + tau * tau)), 0.0)
            c = 1 / np.sqrt(1 + t * t); s = t * c
            J = np.eye(n)
            J[lo, lo] = c; J[hi, hi] = c; J[lo, hi] = s; J[hi, lo] = -s
            G = J.T @ G @ J
            idx = [idx[0]] + [idx[-1]] + idx[1:-1]
        off = np.abs(G - np.diag(np.diag(G))).max() / scale
        hist.append(off)
        if off < tol:
            break
    return len(hist), hist

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rng = np.random.default_rng(0)
Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
for name, lam in [("decay12", np.logspace(0, -12, n)), ("decay6", np.logspace(0, -6, n)), ("flat", np.linspace(1, 0.05, n))]:
    G = (Q * lam) @ Q.T
    G = (G + G.T) / 2
    k0, h0 = sweeps_to_converge(G)
    d = np.argsort(-np.diag(G))
    Gs = G[np.ix_(d, d)]
    R = np.linalg.cholesky(Gs + 1e-13 * np.diag(Gs).max() * np.eye(n)).T
    M = R @ R.T
    k1, h1 = sweeps_to_converge(M)
    R2 = np.linalg.cholesky(M + 1e-13 * np.diag(M).max() * np.eye(n)).T
    M2 = R2 @ R2.T
    k2, h2 = sweeps_to_converge(M2)
    print(name, "plain", k0, "LR1", k1, "LR2", k2)
    print("  plain", " ".join("%.0e" % x for x in h0))
    print("  LR1  ", " ".join("%.0e" % x for x in h1))
