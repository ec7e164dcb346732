"""numpy model of the DEVICE formulation (debug aid, TEST INFRASTRUCTURE only).

Mirrors, step for step, what xmca_amd/csrc does on the GPU so kernels can be
debugged stage by stage against fp64 numpy:

* two-sided block Jacobi EVD of a Hermitian matrix with round-robin pair slots
  (`block_jacobi_evd`), whose diagonal-tile solver is a parallel-order scalar
  Jacobi (`tile_evd`) using exactly the rotation formulas of jacobi.hip.h;
* the Gram ("dual") / covariance ("primal") routes of solve().

It is NOT the oracle (that is ref_numpy.py, which follows the reference's
algorithm); it shares no code with the product.
"""
import numpy as np


def rr_pairs(n, step):
    """circle-method 1-factorisation of K_n (n even): pairs met at `step` in [0, n-1)."""
    m = n - 1
    out = [(n - 1, step)]
    for k in range(1, n // 2):
        out.append(((step + k) % m, (step - k) % m))
    return out


def rotation(a, b, g):
    """Unitary 2x2 [[c, s e^{i phi}], [-s e^{-i phi}, c]] diagonalising [[a, g], [conj g, b]]."""
    ag = abs(g)
    ph = g / ag
    tau = (b - a) / (2.0 * ag)
    t = (1.0 if tau >= 0 else -1.0) / (abs(tau) + np.sqrt(1.0 + tau * tau))
    c = 1.0 / np.sqrt(1.0 + t * t)
    s = t * c
    return c, s * ph        # (c real, sigma = s e^{i phi})


def tile_evd(M, tol=1e-15, abs_floor=0.0, max_sweeps=30):
    """Parallel-order cyclic Jacobi on a small Hermitian tile.  Returns (lam desc, J)."""
    M = np.array(M, dtype=complex if np.iscomplexobj(M) else float)
    n = M.shape[0]
    V = np.eye(n, dtype=M.dtype)
    sweeps = 0
    for sweep in range(max_sweeps):
        rotated = False
        for step in range(n - 1):
            for (p, q) in rr_pairs(n, step):
                if p > q:
                    p, q = q, p
                g = M[p, q]
                a, b = M[p, p].real, M[q, q].real
                ag = abs(g)
                if ag == 0.0 or ag <= abs_floor or ag * ag <= tol * tol * abs(a * b):
                    continue
                rotated = True
                c, sg = rotation(a, b, g)
                # columns: new_p = c col_p - conj(sg) col_q ; new_q = sg col_p + c col_q
                cp, cq = M[:, p].copy(), M[:, q].copy()
                M[:, p], M[:, q] = c * cp - np.conj(sg) * cq, sg * cp + c * cq
                vp, vq = V[:, p].copy(), V[:, q].copy()
                V[:, p], V[:, q] = c * vp - np.conj(sg) * vq, sg * vp + c * vq
                # rows: J^H from the left
                rp, rq = M[p, :].copy(), M[q, :].copy()
                M[p, :], M[q, :] = c * rp - sg * rq, np.conj(sg) * rp + c * rq
                M[p, q] = M[q, p] = 0.0
        sweeps += 1
        if not rotated:
            break
    # NOT sorted: J must stay close to the identity or the outer iteration loses its
    # quadratic convergence (sorting = a large permutation every round).
    return np.diag(M).real.copy(), V, sweeps


def dest_block(p, h, S):
    """round-robin move of half h (0=top,1=bottom) of slot p -> destination block index."""
    if h == 0:
        if p == 0:
            return 0
        if p == S - 1:
            return 2 * (S - 1) + 1
        return 2 * (p + 1)
    if p == 0:
        return 2 * 1 if S > 1 else 1
    return 2 * (p - 1) + 1


def block_jacobi_evd(G, n2=32, tol=1e-10, max_sweeps=30, use_tile_jacobi=False, verbose=False):
    """Two-sided block Jacobi.  G Hermitian (n x n).  Returns (lam desc, U, sweeps)."""
    n = G.shape[0]
    b = n2 // 2
    S = max((n + n2 - 1) // n2, 1)
    npad = S * n2
    dt = complex if np.iscomplexobj(G) else float
    A = np.zeros((npad, npad), dtype=dt)
    A[:n, :n] = G
    gscale = float(np.max(np.abs(np.diag(G)).real)) if n else 1.0
    pad_val = -gscale if gscale > 0 else -1.0
    for i in range(n, npad):
        A[i, i] = pad_val
    Z = np.eye(npad, dtype=dt)            # Z = Q^H
    abs_floor = 1e-13 * gscale
    sweeps = 0
    if S == 1:
        lam, J, _ = tile_evd(A, abs_floor=abs_floor) if use_tile_jacobi else _eigh_desc(A)
        Z = J.conj().T
        return _finish(lam, Z, n, npad)
    nb = 2 * S
    for sweep in range(max_sweeps):
        sweep_off = 0.0
        for rnd in range(nb - 1):
            Js = []
            lams = []
            for P in range(S):
                tile = A[P * n2:(P + 1) * n2, P * n2:(P + 1) * n2]
                d = np.abs(np.diag(tile).real)
                off = np.abs(tile - np.diag(np.diag(tile)))
                denom = np.sqrt(np.outer(d, d))
                mask = off > abs_floor
                if mask.any():
                    with np.errstate(divide="ignore", invalid="ignore"):
                        ratio = np.where(mask, off / np.where(denom > 0, denom, np.inf), 0.0)
                    sweep_off = max(sweep_off, float(ratio.max()))
                if use_tile_jacobi:
                    lam, J, _ = tile_evd(tile, abs_floor=abs_floor)
                else:
                    lam, J = _eigh_desc(tile)[:2]
                Js.append(J)
                lams.append(lam)
            A2 = np.empty_like(A)
            Z2 = np.empty_like(Z)
            for P in range(S):
                JP = Js[P]
                rows_out = JP.conj().T @ Z[P * n2:(P + 1) * n2, :]
                for hi in range(2):
                    db = dest_block(P, hi, S)
                    Z2[db * b:(db + 1) * b, :] = rows_out[hi * b:(hi + 1) * b, :]
                for Q in range(P, S):
                    if P == Q:
                        out = np.diag(lams[P]).astype(dt)
                    else:
                        out = JP.conj().T @ A[P * n2:(P + 1) * n2, Q * n2:(Q + 1) * n2] @ Js[Q]
                    for hi in range(2):
                        for hj in range(2):
                            r, c = dest_block(P, hi, S), dest_block(Q, hj, S)
                            blk = out[hi * b:(hi + 1) * b, hj * b:(hj + 1) * b]
                            A2[r * b:(r + 1) * b, c * b:(c + 1) * b] = blk
                            A2[c * b:(c + 1) * b, r * b:(r + 1) * b] = blk.conj().T
            A, Z = A2, Z2
        sweeps += 1
        if verbose:
            print("sweep", sweeps, "off", sweep_off)
        if sweep_off < tol:
            break
    lam = np.diag(A).real.copy()
    return _finish(lam, Z, n, npad) + (sweeps,)


def _eigh_desc(M):
    """eigh re-ordered / re-phased to be as close to the identity as possible: the
    fast stand-in for `tile_evd` (same fixed point, LAPACK speed)."""
    from scipy.optimize import linear_sum_assignment
    lam, J = np.linalg.eigh(M)
    r, c = linear_sum_assignment(-np.abs(J))
    perm = np.empty(len(lam), int)
    perm[r] = c
    J, lam = J[:, perm], lam[perm]
    d = np.diag(J)
    ph = np.where(np.abs(d) > 0, d / np.where(np.abs(d) > 0, np.abs(d), 1), 1)
    return lam, J * np.conj(ph), 1


def _finish(lam, Z, n, npad):
    order = np.argsort(-lam, kind="stable")[:n]
    U = Z[order, :n].conj().T
    return lam[order], U
