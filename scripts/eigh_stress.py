"""Randomised stress of the device eigensolver against numpy: sizes around the tile boundaries, real / complex, flat and
graded spectra, rank-deficient and indefinite input.  Prints the worst errors; exit code 1 on a violated bound."""
import sys
import numpy as np
sys.path.insert(0, ".")
from xmca_amd import _hip

h = _hip.default_handle()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
worst = {"lam": 0.0, "orth": 0.0, "res": 0.0}
bad = 0
sizes = [1, 2, 31, 32, 33, 63, 64, 65, 127, 128, 129, 130, 191, 193, 200, 255, 256, 257, 320, 385, 500, 641, 777]
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 60):
    n = int(rng.choice(sizes))
    cplx = bool(rng.integers(2))
    kind = rng.choice(["flat", "graded", "deficient", "indefinite", "spiked"])
    Q = rng.standard_normal((n, n)) + (1j * rng.standard_normal((n, n)) if cplx else 0)
    Q, _ = np.linalg.qr(Q)
    if kind == "flat":
        lam = rng.uniform(0.1, 1.0, n)
    elif kind == "graded":
        lam = np.logspace(0, -rng.uniform(3, 13), n)
    elif kind == "deficient":
        lam = np.where(np.arange(n) < max(1, n // 3), rng.uniform(0.1, 1, n), 0.0)
    elif kind == "indefinite":
        lam = rng.standard_normal(n) * np.logspace(0, -rng.uniform(0, 8), n)
    else:
        lam = np.concatenate([rng.uniform(1e4, 1e6, min(5, n)), rng.uniform(0.5, 1.0, max(n - 5, 0))])[:n]
    A = (Q * lam) @ Q.conj().T
    A = (A + A.conj().T) / 2 * 10.0 ** rng.integers(-6, 7)
    w, U = h.eigh(A)
    ref = np.linalg.eigvalsh(A)[::-1]
    scale = np.abs(ref).max() if n else 1.0
    e_lam = np.max(np.abs(w - ref)) / scale
    e_orth = np.max(np.abs(U.conj().T @ U - np.eye(n)))
    e_res = np.max(np.abs(A @ U - U * w)) / scale
    info = h.last_eigh_info
    ok = e_lam < 2e-11 and e_orth < 2e-10 and e_res < 2e-10 and np.all(np.diff(w) <= 0)
    if not ok:
        bad += 1
        print("FAIL", n, cplx, kind, "lam %.2e orth %.2e res %.2e" % (e_lam, e_orth, e_res), info)
    for k, v in (("lam", e_lam), ("orth", e_orth), ("res", e_res)):
        worst[k] = max(worst[k], float(v))
print("trials done, failures:", bad, "worst:", worst)
sys.exit(1 if bad else 0)
