"""Varimax / Promax rotation on the MI355X (drop-in for xmca/tools/rotation.py).

Same signatures and return values as the reference functions; the arithmetic runs in
`xmca_rotate_loadings` (include/xmca_hip.h): the whole Varimax loop in one persistent launch (per iteration an
MFMA accumulation of G = A^H (|Z|^2 Z - gamma Z c / N), an epoch-flag exchange of the per-workgroup partials and a
Newton-Schulz polar factor R = U V^H; csrc/rotate.h), the reference's stopping rule evaluated on the device, then the
Promax passes.  There is no CPU fallback: without the HIP library / a GPU these functions raise.
"""
import numpy as np

from .. import _hip


def varimax(A, gamma=1, maxIter=1000, tol=1e-8, handle=None):
    """Kaiser-normalised Varimax rotation (xmca/tools/rotation.py:15-78).  Returns (B, R).
    `gamma`: 1 = Varimax, 0 = Quartimax (rotation.py:56-57)."""
    A = np.asarray(A)
    h = handle or _hip.default_handle()
    out = h.rotate_loadings(A, n_left=A.shape[0], power=1, tol=tol, max_iter=maxIter, varimax_only=True, want_B=True,
                            gamma=gamma)
    return out["B"], out["R"]


def promax(A, power=1, maxIter=1000, tol=1e-8, handle=None):
    """Promax rotation (xmca/tools/rotation.py:84-149).  Returns (B, R, phi)."""
    A = np.asarray(A)
    n, p = A.shape
    if p < 2:
        # same early exit as the reference (rotation.py:107-109)
        print('Cannot rotate 1 PC. No rotation performed.')
        X = A.copy()
        return X, np.eye(n), X.conjugate().T @ X
    h = handle or _hip.default_handle()
    out = h.rotate_loadings(A, n_left=n, power=power, tol=tol, max_iter=maxIter, varimax_only=False, want_B=True)
    return out["B"], out["R"], out["Phi"]
