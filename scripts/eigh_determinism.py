"""Runs the device eigensolver several times on the same matrix and reports whether the results are bitwise equal."""
import sys
import numpy as np
sys.path.insert(0, ".")
from xmca_amd import _hip

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
rng = np.random.default_rng(3)
X = rng.standard_normal((n, 3 * n))
X -= X.mean(axis=1, keepdims=True)
A = X @ X.T
h = _hip.default_handle()
ref = None
for rep in range(4):
    lam, U = h.eigh(A)
    if ref is None:
        ref = (lam.copy(), U.copy())
    print("rep", rep, "sweeps", h.last_eigh_info["sweeps"], "lam bitwise equal:", np.array_equal(lam, ref[0]),
          "U bitwise equal:", np.array_equal(U, ref[1]), "max |dlam|/|lam|max", np.max(np.abs(lam - ref[0])) / np.abs(lam).max())
