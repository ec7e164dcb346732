"""GPU box: accuracy of the C2 eigen-decomposition as a function of the stopping tolerance of the Jacobi sweeps
(XMCA_JACOBI_TOL), against LAPACK: eigenvalues (relative, every non-null one), residuals and orthogonality of leading, bulk
and trailing vectors."""
import json, os, subprocess, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    from xmca_amd import _hip
    d = np.load("/tmp/eigh_tol_probe.npz")
    G, ref = d["G"], d["ref"]
    h = _hip.Handle(0)
    lam, U = h.eigh(G)
    n = len(ref)
    nn = ref > 1e-9 * ref[0]
    sel = np.r_[0:20, 20:60, n // 2:n // 2 + 40, n - 60:n - 20]
    Us = U[:, sel]
    res = np.linalg.norm(G @ Us - Us * lam[sel], axis=0) / ref[0]
    orth = np.max(np.abs(Us.T @ U - np.eye(n)[sel]))
    print(json.dumps({"tol": os.environ.get("XMCA_JACOBI_TOL", "default"), "sweeps": h.last_eigh_info["sweeps"],
                      "lam_rel_err_max_nonnull": float(np.max(np.abs(lam[nn] - ref[nn]) / ref[nn])),
                      "lam_abs_err_over_lam0": float(np.max(np.abs(lam - ref)) / ref[0]),
                      "resid_over_norm_leading20": float(res[:20].max()), "resid_bulk": float(res[20:].max()), "orth": float(orth)}))
    sys.exit(0)
sys.path.insert(0, os.path.join(REPO, "tests"))
from golden_inputs import gen_A
X = gen_A()
X -= X.mean(axis=0)
G = X @ X.T
ref = np.linalg.eigvalsh(G)[::-1]
np.savez("/tmp/eigh_tol_probe.npz", G=G, ref=ref)
for tol in ["1e-10", "3e-8", "2e-6", "1e-4"]:
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, XMCA_JACOBI_TOL=tol), capture_output=True, text=True)
    print(r.stdout.strip() or r.stderr[-500:])
