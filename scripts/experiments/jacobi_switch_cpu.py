"""CPU experiment: when to insert a Cholesky LR step into a cyclic Jacobi run, and what the diagonal says beforehand."""
import sys
import numpy as np

def jacobi(G, tol=1e-10, max_sweeps=60, stop_after=None):
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
        if off < tol or (stop_after and sweep + 1 == stop_after):
            break
    return G, hist

def lr_step(G):
    n = G.shape[0]
    d = np.argsort(-np.diag(G)); G = G[np.ix_(d, d)]
    R = np.linalg.cholesky(G + 1e-13 * np.diag(G).max() * np.eye(n)).T
    return R @ R.T

def stat(G):
    d = np.sort(np.abs(np.diag(G)))[::-1]
    n = len(d)
    return "q25/q75=%.1e q10/q90=%.1e" % (d[n // 4] / d[3 * n // 4], d[n // 10] / d[9 * n // 10])

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rng = np.random.default_rng(0)
Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
i = np.arange(1, n + 1)
mp = np.linalg.eigvalsh(np.cov(rng.standard_normal((n, 4 * n))))[::-1]
spiked = mp.copy(); spiked[:8] = np.linspace(1e5, 1e3, 8)
cases = [("decay12", np.logspace(0, -12, n)), ("decay6", np.logspace(0, -6, n)), ("decay3", np.logspace(0, -3, n)),
         ("pow2", i ** -2.0), ("pow1", i ** -1.0), ("flat", np.linspace(1, 0.05, n)), ("mp", mp), ("spiked", spiked)]
for name, lam in cases:
    G = (Q * lam) @ Q.T
    G = (G + G.T) / 2
    _, h0 = jacobi(G)
    out = [name, "plain %d" % len(h0)]
    for k in (0, 2, 3, 4):
        Gk, hk = (G, []) if k == 0 else jacobi(G, stop_after=k)
        _, h1 = jacobi(lr_step(Gk))
        out.append("LR@%d: %d+%d %s" % (k, k, len(h1), stat(Gk)))
    print(" | ".join(out), flush=True)
