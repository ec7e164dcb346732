import os, sys
os.environ["XMCA_TRIDIAG_VEC_MIN_N"] = "2"
os.environ["XMCA_TRIDIAG_MIN_N"] = "2"
sys.path.insert(0, ".")
import numpy as np
from xmca_amd import _hip
h = _hip.Handle(0)
rng = np.random.default_rng(0)
bad = 0
for n in (2, 3, 5, 17, 63, 64, 65, 66, 127, 128, 129, 191, 192, 256, 257, 511, 512, 513, 515):
    for cplx in (False, True):
        X = rng.standard_normal((n, 3 * n + 5))
        if cplx: X = X + 1j * rng.standard_normal((n, 3 * n + 5))
        G = X @ X.conj().T
        for rep in range(2):
            lam, U = h.eigh(G)
            ref = np.linalg.eigvalsh(G)[::-1]
            e1 = np.max(np.abs(lam - ref)) / ref[0]
            e2 = np.max(np.abs(U.conj().T @ U - np.eye(n)))
            e3 = np.max(np.linalg.norm(G @ U - U * lam, axis=0)) / ref[0]
            ok = e1 < 1e-13 and e2 < 1e-12 and e3 < 1e-12
            if not ok or h.last_eigh_info["tridiag"] != 1:
                bad += 1
                print("n", n, cplx, rep, "tridiag", h.last_eigh_info["tridiag"], e1, e2, e3)
print("bad", bad)
