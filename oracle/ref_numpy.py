"""numpy restatement of the reference hot path (TEST INFRASTRUCTURE, not product).

Every function restates the *algorithm* of nicrie/xmca v1.4.2 for the path
``MCA.solve`` / ``MCA.rotate`` / ``MCA.rule_n`` and cites the reference
file:line it follows.  All arithmetic of the reference lives in un-vendored
third-party code (numpy ``linalg.svd``/``@`` -> LAPACK gesdd / BLAS gemm,
``scipy.signal.hilbert``), so this restatement calls the same numpy/scipy
entry points in the same order.

Parity pin: ``oracle/make_goldens.py`` imports the real reference from
``/root/reference`` (build container only), checks every function here against
it on seeded inputs (<= 1e-12) and against the reference's own netCDF golden
vectors (tests/integration/fixtures/{std,cplx}), and commits the resulting
vectors under ``tests/golden/``.  ``tests/test_oracle_goldens.py`` re-checks the
oracle against those vectors on every run.
"""
from __future__ import annotations

import numpy as np
from scipy.signal import hilbert as _hilbert

__all__ = [
    "flatten_and_center", "analytic_signal", "solve", "varimax", "promax",
    "rotate", "variance_of", "rule_n", "OracleModel",
]


# ----------------------------------------------------------------------------
# constructor-side preprocessing (xmca/array.py:191-240, :199-207)
# ----------------------------------------------------------------------------
def flatten_and_center(field):
    """T x ... -> (T x N' centered, valid-column mask, column mean, column std).

    reshape to 2-D (array.py:230-240), drop columns holding any NaN
    (array.py:217-228, tools/array.py:14-62), column mean / std with ddof=0
    (array.py:209-215), subtract the mean (array.py:199-207).  dtype is kept.
    """
    field = np.asarray(field)
    t = field.shape[0]
    flat = field.reshape(t, int(np.prod(field.shape[1:])))
    valid = ~np.isnan(flat).any(axis=0)
    flat = flat[:, valid]
    mean = flat.mean(axis=0)
    std = flat.std(axis=0)
    return flat - mean, valid, mean, std


def analytic_signal(x):
    """Hilbert complexify along time without extension (array.py:455-464)."""
    return _hilbert(np.asarray(x).real, axis=0)


# ----------------------------------------------------------------------------
# solve (xmca/array.py:509-603)
# ----------------------------------------------------------------------------
def solve(fields, complexify=False):
    """Reference solve on already centered 2-D fields (list of 1 or 2 arrays).

    Steps (array.py): :546-547 complexify; :552 thin SVD per field;
    :553 PC scores U*s; :555-566 kernel = R_l^H R_r / (T-1) (R_l^H R_l for one
    field); :569-578 SVD of the kernel; :580-584 back-projection Vt^H @ P.
    Returns a dict with the state block of :590-603.
    """
    fields = [np.asarray(f) for f in fields]
    if complexify:
        fields = [analytic_signal(f) for f in fields]
    dof = fields[0].shape[0] - 1
    scores, vts = [], []
    for f in fields:
        u, s, vt = np.linalg.svd(f, full_matrices=False)
        scores.append(u * s)
        vts.append(vt)
    left = scores[0]
    right = scores[1] if len(scores) == 2 else scores[0]
    kernel = left.conj().T @ right / dof
    p, sigma, qh = np.linalg.svd(kernel, full_matrices=False)
    small = [p, qh.conj().T]
    V = [vts[k].conj().T @ small[k] for k in range(len(fields))]
    return {
        "fields": fields,
        "V": V,
        "singular_values": sigma,
        "norm": [np.sqrt(sigma) for _ in fields],
        "variance": sigma,
        "var_idx": np.argsort(sigma)[::-1],
        "total_covariance": sigma.sum(),
        "total_squared_covariance": (sigma ** 2).sum(),
        "rank": len(sigma),
    }


# ----------------------------------------------------------------------------
# Varimax / Promax (xmca/tools/rotation.py:15-78, :84-149)
# ----------------------------------------------------------------------------
def varimax(A, gamma=1.0, max_iter=1000, tol=1e-8):
    """Kaiser-normalised Varimax.  Returns (B, R, n_iter).

    rotation.py:46-48 row normalisation; :52-64 loop
    ``Z=AR; G=A^H(Z^2 conj(Z) - gamma/n Z diag(colsum|Z|^2)); R=U V^H; d=sum(s)``
    stopping at the first iteration with ``|d-d_old|/d < tol``;
    :66-71 RuntimeError when ``max_iter`` is exhausted; :74-77 de-normalise, B=(hA)R.
    """
    A = np.array(A, copy=True)
    n, p = A.shape
    h = np.sqrt(np.sum(A * A.conj(), axis=1))
    A = (1.0 / h)[:, None] * A          # operand order kept: complex64 products are not bitwise commutative
    R = np.eye(p)
    d = 0.0
    n_iter = 0
    for it in range(max_iter):
        d_prev = d
        Z = A @ R
        colsq = np.sum(Z * Z.conj(), axis=0)
        G = A.conj().T @ (Z ** 2 * Z.conj() - (gamma / n) * (Z @ np.diag(colsq)))
        u, s, vh = np.linalg.svd(G)
        R = u @ vh
        d = np.sum(s)
        n_iter = it + 1
        if abs(d - d_prev) / d < tol:
            break
    else:
        raise RuntimeError("Rotation process did not converge.")
    B = (h[:, None] * A) @ R
    return B, R, n_iter


def promax(A, power=1, max_iter=1000, tol=1e-8):
    """Promax = Varimax + oblique Procrustes fit.  Returns (B, R, Phi, n_iter).

    rotation.py:112 varimax; :115-117 row normalise; :121 column-max
    normalise; :124 target P = Xn |Xn|^(power-1); :128 L = (X^H X)^-1 X^H P;
    :131-137 rescale by sqrt(diag((L^H L)^-1)) with pinv fallback;
    :138-147 B = h (X L), R <- R L, Phi = L^-1 L^-H.
    """
    X, R, n_iter = varimax(A, max_iter=max_iter, tol=tol)
    h = np.sqrt(np.sum(X * X.conj(), axis=1))
    X = (1.0 / h)[:, None] * X
    Xn = X / np.max(np.abs(X), axis=0)
    P = Xn * np.abs(Xn) ** (power - 1)
    L = np.linalg.inv(X.conj().T @ X) @ X.conj().T @ P
    try:
        scale = np.diag(np.diag(np.linalg.inv(L.conj().T @ L)))
    except np.linalg.LinAlgError:
        scale = np.diag(np.diag(np.linalg.pinv(L.conj().T @ L)))
    L = L @ np.sqrt(scale)
    B = h[:, None] * (X @ L)
    R = R @ L
    Linv = np.linalg.inv(L)
    Phi = Linv @ Linv.conj().T
    return B, R, Phi, n_iter


# ----------------------------------------------------------------------------
# rotate driver (xmca/array.py:781-844)
# ----------------------------------------------------------------------------
def rotate(V, singular_values, n_rot, power=1, tol=1e-8):
    """array.py:815-833: loadings L=[V_l;V_r][:, :p] sqrt(s[:p]); promax; block norms."""
    if n_rot < 2:
        raise ValueError("`n_rot` must be > 1")
    if power < 1:
        raise ValueError("`power` must be >=1")
    s = np.asarray(singular_values)[:n_rot]
    blocks = [np.asarray(v)[:, :n_rot] for v in V]
    n_left = blocks[0].shape[0]
    L = np.concatenate(blocks) * np.sqrt(s)
    L_rot, R, Phi, n_iter = promax(L, power, max_iter=1000, tol=tol)
    norm_left = np.linalg.norm(L_rot[:n_left], axis=0)
    norm_right = np.linalg.norm(L_rot[n_left:], axis=0) if len(blocks) == 2 else norm_left
    variance = norm_left * norm_right
    return {
        "R": R, "Phi": Phi, "norm": [norm_left, norm_right][:len(blocks)],
        "variance": variance, "var_idx": np.argsort(variance)[::-1],
        "n_iter": n_iter, "L_rot": L_rot,
    }


def variance_of(norms, bivariate, var_idx):
    """array.py:755-779 with sorted=True, n=None."""
    nl = norms[0][var_idx]
    if bivariate:
        return nl * norms[1][var_idx]
    return nl ** 2


# ----------------------------------------------------------------------------
# A tiny model object so the rule_n restatement reads like the reference
# ----------------------------------------------------------------------------
class OracleModel:
    """Holds exactly the state rule_n needs (array.py:1744-1749, :1764-1769)."""

    def __init__(self, *fields):
        self.raw_shapes = [np.asarray(f).shape for f in fields]
        prepared = [flatten_and_center(f) for f in fields]
        self.fields = [p[0] for p in prepared]
        self.valid = [p[1] for p in prepared]
        self.bivariate = len(fields) == 2
        self.is_complex = False
        self.is_rotated = False
        self.n_rot = 0
        self.power = 0

    def solve(self, complexify=False):
        out = solve(self.fields, complexify)
        self.fields = out["fields"]
        self.is_complex = complexify
        self.V = out["V"]
        self.singular_values = out["singular_values"]
        self.norm = out["norm"]
        self.var_idx = out["var_idx"]
        self.rank = out["rank"]
        self.total_covariance = out["total_covariance"]
        self.is_rotated = False
        self.n_rot = self.rank
        self.power = 0
        return out

    def rotate(self, n_rot, power=1, tol=1e-8):
        out = rotate(self.V, self.singular_values, n_rot, power, tol)
        self.norm = out["norm"]
        self.var_idx = out["var_idx"]
        self.R, self.Phi = out["R"], out["Phi"]
        self.is_rotated, self.n_rot, self.power = True, n_rot, power
        self.n_iter = out["n_iter"]
        return out

    def variance(self):
        return variance_of(self.norm, self.bivariate, self.var_idx)


def rule_n(model, n_runs, n_modes=None, normal=None):
    """array.py:1716-1771.  ``normal(shape)`` defaults to the global numpy stream.

    Per run: standard_normal([T, n_variables]) per field, left then right
    (:1755-1756; NaN columns are counted, :1745); MCA(); solve(complexify);
    rotate(n_rot, power) if rotated, dropping the run on RuntimeError
    (:1759-1763); keep the sorted variance (:1764).  Then scale every run so
    its sum equals the sum of the model's own variance (:1767-1769) and slice.
    """
    if normal is None:
        normal = np.random.standard_normal
    T = model.raw_shapes[0][0]
    widths = [int(np.prod(s[1:])) for s in model.raw_shapes]
    kept = []
    for _ in range(n_runs):
        data = [normal([T, w]) for w in widths]
        sur = OracleModel(*data)
        sur.solve(complexify=model.is_complex)
        if model.is_rotated:
            try:
                sur.rotate(model.n_rot, model.power)
            except RuntimeError:
                continue
        kept.append(sur.variance())
    sv = np.array(kept).T
    ref = model.variance()
    sv /= sv.sum(axis=0) / ref.sum()
    n_modes = model.rank if n_modes is None else n_modes
    return sv[:n_modes]
