"""CPU model: rounds of two-sided block Jacobi (exact 64 x 64 pair solves, half-blocks of 32) until the largest
off-diagonal entry is below 1e-10 max|diag| - round-robin tournament against a greedy dynamic ordering (every round pairs
the half-blocks by descending Frobenius weight of their coupling block; `lag` = how many rounds old the weights are that
the matching uses - lag 1 is what a look-ahead launch structure could afford).  Input: a C2-shaped Gram matrix."""
import sys, time
import numpy as np
from scipy.optimize import linear_sum_assignment

def closest_to_identity_eigh(M):
    lam, J = np.linalg.eigh(M)
    r, c = linear_sum_assignment(-np.abs(J))
    perm = np.empty(len(lam), int); perm[r] = c
    J = J[:, perm]
    return J * np.sign(np.where(np.diag(J) == 0, 1, np.diag(J)))

def rr_pairs(n, step):
    m = n - 1
    out = [(n - 1, step)]
    for k in range(1, n // 2):
        out.append(((step + k) % m, (step - k) % m))
    return out

def weights(A, b, nb):
    B = (A * A).reshape(nb, b, nb, b).sum(axis=(1, 3))
    np.fill_diagonal(B, 0.0)
    return B

def greedy_matching(W):
    nb = W.shape[0]
    order = np.dstack(np.unravel_index(np.argsort(-W, axis=None), W.shape))[0]
    used = np.zeros(nb, bool); pairs = []
    for i, j in order:
        if i < j and not used[i] and not used[j]:
            used[i] = used[j] = True; pairs.append((i, j))
            if len(pairs) == nb // 2: break
    return pairs

def run(G, b, mode, lag=0, max_rounds=3000, tol=1e-10):
    A = G.copy(); n = A.shape[0]; nb = n // b
    scale = np.max(np.abs(np.diag(A)))
    hist = [weights(A, b, nb)]
    recent = []
    rounds = 0
    while rounds < max_rounds:
        if mode == "rr":
            pairs = rr_pairs(nb, rounds % (nb - 1))
        else:
            W = hist[max(0, len(hist) - 1 - lag)].copy()
            for old in recent[-lag:] if lag else []:          # pairs solved since those weights were measured are done
                for (i, j) in old:
                    W[i, j] = W[j, i] = 0.0
            pairs = greedy_matching(W)
            recent.append(pairs)
        for (i, j) in pairs:
            idx = np.r_[i * b:(i + 1) * b, j * b:(j + 1) * b]
            J = closest_to_identity_eigh(A[np.ix_(idx, idx)])
            A[idx, :] = J.T @ A[idx, :]
            A[:, idx] = A[:, idx] @ J
        rounds += 1
        hist.append(weights(A, b, nb)); hist = hist[-(lag + 2):]
        off = np.max(np.abs(A - np.diag(np.diag(A)))) / scale
        if off < tol: break
    return rounds, off

if __name__ == "__main__":
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 1536     # (an even number of half-blocks of 32)
    rng = np.random.default_rng(0)
    k = 20
    X = (rng.standard_normal((T, k)) * np.linspace(10, 1, k)) @ rng.standard_normal((k, 4 * T)) + rng.standard_normal((T, 4 * T))
    X -= X.mean(axis=0)
    G = X @ X.T
    b = 32
    nb = T // b
    print("n = %d, %d half-blocks, %d rounds per round-robin sweep" % (T, nb, nb - 1))
    for mode, lag in (("rr", 0), ("dyn", 0), ("dyn", 1), ("dyn", 2)):
        t = time.time()
        r, off = run(G, b, mode, lag)
        print("%-3s lag %d: %4d rounds (= %.1f sweeps), off %.1e   [%.0f s]" % (mode, lag, r, r / (nb - 1), off, time.time() - t), flush=True)
