"""Varimax / Promax rotation on the MI355X (drop-in for xmca/tools/rotation.py).

Same signatures and return values as the reference functions; the arithmetic runs in
`xmca_rotate_loadings` (include/xmca_hip.h): a fused single-pass Varimax step kernel plus an on-device
p x p Jacobi SVD per iteration, the reference's stopping rule evaluated on the device.
There is no CPU fallback: without the HIP library / a GPU these functions raise.
"""
import numpy as np

from .. import _hip


def _check_gamma(gamma):
    if gamma != 1:
        raise NotImplementedError("only the Varimax criterion (gamma=1) is implemented on the device")


def varimax(A, gamma=1, maxIter=1000, tol=1e-8, handle=None):
    """Kaiser-normalised Varimax rotation (xmca/tools/rotation.py:15-78).  Returns (B, R)."""
    _check_gamma(gamma)
    A = np.asarray(A)
    h = handle or _hip.default_handle()
    out = h.rotate_loadings(A, n_left=A.shape[0], power=1, tol=tol, max_iter=maxIter, varimax_only=True, want_B=True)
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
