#!/usr/bin/env python3
"""CPU model of the device tridiagonalisation (csrc/tridiag.h): unblocked Householder reduction of a Hermitian matrix in
the fused "update-and-multiply" form - ONE pass over the trailing matrix per column applies the rank-2 update of the
previous column and multiplies by the new Householder vector; the new vector itself comes from row j of the not yet
updated matrix plus the two vectors of the previous step (so every workgroup can form it redundantly: no gather).
Checks the tridiagonal's eigenvalues against numpy.linalg.eigvalsh, Sturm bisection, twisted-factorisation vectors,
back-transformation and a Newton-Schulz clean-up."""
import sys
import numpy as np


def larfg(x):
    """H = I - tau v v^H with v[0] = 1 and H^H x = beta e_1, beta real (LAPACK zlarfg)."""
    alpha = x[0]
    xnorm2 = float(np.sum(np.abs(x[1:]) ** 2))
    if xnorm2 == 0.0 and alpha.imag == 0.0:
        v = np.zeros_like(x); v[0] = 1.0
        return v, 0.0, alpha.real
    beta = -np.copysign(np.sqrt(abs(alpha) ** 2 + xnorm2), alpha.real)
    tau = (beta - alpha) / beta
    v = x / (alpha - beta)
    v[0] = 1.0
    return v, tau, beta


def tridiagonalise_fused(A, keep_reflectors=False):
    """returns d (n), e (n-1), and optionally (V rows = reflectors, taus).  A: full Hermitian (copied)."""
    A = np.array(A, dtype=np.complex128 if np.iscomplexobj(A) else np.float64)
    cplx = np.iscomplexobj(A)
    n = A.shape[0]
    d = np.zeros(n); e = np.zeros(max(n - 1, 0))
    taus = np.zeros(max(n - 1, 0), dtype=A.dtype)
    Vs = np.zeros((max(n - 1, 0), n), dtype=A.dtype)
    vprev = np.zeros(n, dtype=A.dtype)      # v_{j-1}, global row index
    pprev = np.zeros(n, dtype=A.dtype)      # p_{j-1} = tau A v
    tau_prev = 0.0
    gamma_prev = 0.0                        # sum conj(p) v
    for j in range(n - 1):
        # ---- prologue (every workgroup, redundantly) ----
        alpha = -0.5 * tau_prev * gamma_prev
        w = pprev + alpha * vprev                                   # w_{j-1}, entries i >= j matter
        # column j of the CURRENT matrix from row j of the stored one (updated through step j-2)
        rowj = A[j, j:].copy()                                      # A[j, k], k >= j
        colj = np.conj(rowj)                                        # A[k, j]
        x = colj - vprev[j:] * np.conj(w[j]) - w[j:] * np.conj(vprev[j])
        d[j] = x[0].real
        v, tau, beta = larfg(x[1:].copy())
        e[j] = beta
        taus[j] = tau
        vj = np.zeros(n, dtype=A.dtype); vj[j + 1:] = v
        Vs[j] = vj
        # ---- pass: rows i >= j+1, columns k >= j+1: apply the previous update, multiply by v ----
        sl = slice(j + 1, n)
        A[sl, sl] -= np.outer(vprev[sl], np.conj(w[sl])) + np.outer(w[sl], np.conj(vprev[sl]))
        p = np.zeros(n, dtype=A.dtype)
        p[sl] = tau * (A[sl, sl] @ vj[sl])
        gamma_prev = np.vdot(p[sl], vj[sl])                          # sum conj(p) v
        vprev, pprev, tau_prev = vj, p, tau
    # last diagonal entry: one more (1 x 1) update
    j = n - 1
    alpha = -0.5 * tau_prev * gamma_prev
    w = pprev + alpha * vprev
    d[j] = (A[j, j] - vprev[j] * np.conj(w[j]) - w[j] * np.conj(vprev[j])).real
    if keep_reflectors:
        return d, e, Vs, taus
    return d, e


def sturm_count(d, e2, lam, pivmin):
    """number of eigenvalues < lam (LDL^T recurrence with the pivmin safeguard of dstebz)."""
    cnt = 0
    q = d[0] - lam
    if abs(q) < pivmin: q = -pivmin
    cnt += q < 0
    for i in range(1, len(d)):
        q = d[i] - lam - e2[i - 1] / q
        if abs(q) < pivmin: q = -pivmin
        cnt += q < 0
    return cnt


def bisect_all(d, e):
    n = len(d)
    e2 = e * e
    bnd = max(np.max(np.abs(d)) + 2 * (np.max(np.abs(e)) if n > 1 else 0.0), 1e-300)
    pivmin = np.finfo(float).tiny * max(1.0, np.max(e2) if n > 1 else 1.0)
    lo = np.full(n, -bnd * 1.0000001); hi = np.full(n, bnd * 1.0000001)
    for _ in range(64):
        mid = 0.5 * (lo + hi)
        for k in range(n):                       # eigenvalue k (ascending) : count(mid) <= k -> lo = mid
            if sturm_count(d, e2, mid[k], pivmin) <= k: lo[k] = mid[k]
            else: hi[k] = mid[k]
    return 0.5 * (lo + hi)


def twisted_vector(d, e, lam):
    """eigenvector of the tridiagonal for the (accurate) eigenvalue lam: forward and backward stationary factorisations of
    T - lam I, twist at the smallest |gamma_r| (Parlett & Dhillon)."""
    n = len(d)
    if n == 1:
        return np.ones(1)
    tiny = np.finfo(float).tiny * 1e20
    dp = np.zeros(n); lp = np.zeros(n - 1)       # T - lam = L D+ L^T
    dp[0] = d[0] - lam
    for i in range(n - 1):
        if abs(dp[i]) < tiny: dp[i] = tiny if dp[i] >= 0 else -tiny
        lp[i] = e[i] / dp[i]
        dp[i + 1] = d[i + 1] - lam - lp[i] * e[i]
    dm = np.zeros(n); um = np.zeros(n - 1)       # T - lam = U D- U^T
    dm[n - 1] = d[n - 1] - lam
    for i in range(n - 2, -1, -1):
        if abs(dm[i + 1]) < tiny: dm[i + 1] = tiny if dm[i + 1] >= 0 else -tiny
        um[i] = e[i] / dm[i + 1]
        dm[i] = d[i] - lam - um[i] * e[i]
    gam = dp + dm - (d - lam)
    r = int(np.argmin(np.abs(gam)))
    z = np.zeros(n); z[r] = 1.0
    for i in range(r - 1, -1, -1):
        z[i] = -lp[i] * z[i + 1]
    for i in range(r, n - 1):
        z[i + 1] = -um[i] * z[i]
    return z / np.linalg.norm(z)


def back_transform(Vs, taus, Y):
    """Q Y with Q = H_0 H_1 ... H_{n-2}, H_j = I - tau_j v_j v_j^H (reflectors in the rows of Vs)."""
    Z = np.array(Y, dtype=Vs.dtype)
    for j in range(len(taus) - 1, -1, -1):
        v = Vs[j]
        Z -= taus[j] * np.outer(v, np.conj(v) @ Z)
    return Z


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    rng = np.random.default_rng(1)
    for cplx in (False, True):
        X = (rng.standard_normal((n, 20)) * np.linspace(10, 1, 20)) @ rng.standard_normal((20, 3 * n)) + rng.standard_normal((n, 3 * n))
        if cplx:
            X = X + 1j * rng.standard_normal((n, 3 * n))
        X -= X.mean(axis=0)
        G = X @ X.conj().T
        d, e, Vs, taus = tridiagonalise_fused(G, keep_reflectors=True)
        ref = np.linalg.eigvalsh(G)
        T = np.diag(d) + np.diag(e, 1) + np.diag(e, -1)
        lam_t = np.linalg.eigvalsh(T)
        print("cplx", cplx, "n", n, "tridiagonal eigenvalues vs eigvalsh(G): %.2e (rel. to lam_max)" % (np.max(np.abs(lam_t - ref)) / ref[-1]))
        lam_b = bisect_all(d, e) if n <= 200 else lam_t
        print("   bisection vs eigvalsh(T): %.2e" % (np.max(np.abs(lam_b - lam_t)) / ref[-1]))
        Y = np.stack([twisted_vector(d, e, l) for l in lam_b], axis=1)
        print("   twisted vectors: orthogonality %.2e, residual %.2e" % (np.max(np.abs(Y.T @ Y - np.eye(n))), np.max(np.linalg.norm(T @ Y - Y * lam_b, axis=0)) / ref[-1]))
        Z = back_transform(Vs, taus, Y)
        print("   back-transformed: orthogonality %.2e, residual %.2e" % (np.max(np.abs(Z.conj().T @ Z - np.eye(n))), np.max(np.linalg.norm(G @ Z - Z * lam_b, axis=0)) / ref[-1]))
        for it in range(3):
            S = Z.conj().T @ Z
            Z = Z @ (1.5 * np.eye(n) - 0.5 * S)
            print("   Newton-Schulz %d: orthogonality %.2e, residual %.2e" % (it + 1, np.max(np.abs(Z.conj().T @ Z - np.eye(n))), np.max(np.linalg.norm(G @ Z - Z * lam_b, axis=0)) / ref[-1]))




# ----------------------------------------------------------------------------------------------------------------------
# Two-stage (dense -> band -> tridiagonal) cost model, round 5.  Step counts of the algorithm x unit
# costs MEASURED on the MI355X with the kernels such a reduction would be built from (profiles/r05_two_stage_gate.md holds the
# rocprofv3 numbers they come from).  `python scripts/experiments/tridiag_model.py two-stage` prints the table.
# ----------------------------------------------------------------------------------------------------------------------
MEASURED_US = {
    # 64 x 64 factorisation + solve of the attached panel, one launch (chol64_panel_kernel): avg of the Cholesky panel loops
    "factor64": {"real": 20.6, "complex": 39.6},
    # split-K product with a 64-wide result + fixed-order sum of the partial tiles (chol64_rowupdate_kernel): the shape of the
    # panel Gram matrices P^H P, of Z = A Y and of Y^H Z
    "skinny64": {"real": 23.2, "complex": 40.3},
    # rank-64 update of the whole trailing matrix (gemm.h, right-looking Cholesky of round 4: traffic-bound); a Hermitian rank-2b
    # update moves the same bytes once per real plane product
    "trailing": {"real": 41.2, "complex": 82.4},
    "launch_gap": 1.4,          # dependent kernel boundary (MI355X_MICROARCH.md price list)
    "handoff": 1.0,             # cross-CU hand-off of a <= 4 KB record, idle chip (same list)
    "cu_l2_GBs": 100.0,         # what one CU streams from L2 / MALL with loads in flight (62-122 GB/s in the list)
}


def two_stage_cost(n, b=64, cplx=False, look_ahead=True, lag=2):
    """Wall-clock model (ms) of stage 1 (panel QR by two Cholesky-QR passes + reconstruction of the block reflector, Z = A Y,
    W, rank-2b update) and stage 2 (bulge chasing, one workgroup per sweep, `lag` block steps between consecutive sweeps)."""
    k = "complex" if cplx else "real"
    u = MEASURED_US
    panels = (n - 1) // b
    # ---- stage 1, chain of one panel ----
    gram = u["skinny64"][k]
    fact = u["factor64"][k] * max(b, 16) / 64.0   # a chain of b pivots (+ ~4 us of launch, loads and stores, kept in the scaling)
    panel = 2 * (gram + fact) + fact            # CholQR2 + inverse of I - S^H Q_top (one more 64-step elimination)
    zprod = u["skinny64"][k]                    # Z = A22 Y, split-K, 64 columns
    small = u["skinny64"][k] * 0.5 + 10.0       # Y^H Z (64 x 64), W = Z K^H - Y (K M K^H) / 2
    colup = u["skinny64"][k] * 0.5              # update of the next panel's columns ahead of the rest (look-ahead)
    trail = u["trailing"][k]
    gaps = 9 * u["launch_gap"]
    chain = panel + zprod + small + gaps + (colup + max(0.0, trail - panel) if look_ahead else trail)
    stage1 = panels * chain * 1e-3
    # ---- stage 2: n sweeps, sweep s+1 starts `lag` block steps behind sweep s; a block step touches three b x b blocks ----
    e = 16 if cplx else 8
    step_bytes = 2 * 3 * b * b * e                                   # read + write of D_k, B_k, B_{k-1}
    tau = step_bytes / (u["cu_l2_GBs"] * 1e3)                        # us per block step when the blocks stream through one CU
    stage2_chain = n * (lag * tau + u["handoff"]) * 1e-3             # sweeps on different CUs: one hand-off per sweep on the chain
    steps = n * n / (2.0 * b)
    in_flight = max(1.0, n / (lag * b) / 2.0)                        # average number of sweeps alive
    stage2_tput = steps * tau / in_flight * 1e-3
    return {"panels": panels, "panel_chain_us": chain, "stage1_ms": stage1, "step_us": tau, "stage2_ms": max(stage2_chain, stage2_tput),
            "total_ms": stage1 + max(stage2_chain, stage2_tput)}


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "two-stage":
    print("%-22s %4s %10s %12s %10s %10s %9s   (one-stage, measured)" % ("problem", "b", "panel us", "stage 1 ms", "step us", "stage 2 ms", "total ms"))
    for name, n, cplx, now in (("C2  n=2920 real", 2920, False, 21.6), ("C4  n=2501 complex", 2501, True, 26.9)):
        for b in (32, 64, 128):
            for la in (False, True):
                r = two_stage_cost(n, b, cplx, la)
                print("%-22s %4d %10.0f %12.2f %10.2f %10.2f %9.1f   %.1f %s" % (name, b, r["panel_chain_us"], r["stage1_ms"], r["step_us"], r["stage2_ms"],
                                                                                r["total_ms"], now, "look-ahead" if la else "serial"))
    sys.exit(0)

if __name__ == "__main__":
    main()
