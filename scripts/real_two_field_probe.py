#!/usr/bin/env python3
"""C3's fields WITHOUT complexify (real two-field model, T = 5000 x (20 000, 15 000)): the general one-sided route with the
Cholesky factor of G_a against the eigen-factor (XMCA_CHOLESKY_FACTOR=0) - time, sweeps, and the difference of the results."""
import json, os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from golden_inputs import make_input
from xmca_amd.array import MCA
m = MCA(*make_input("c3_full"))
m.solve()                                   # warm-up (pool, tile maps)
m = MCA(*make_input("c3_full"))
m._device().reset_timings()
t0 = time.perf_counter(); m.solve(); dt = time.perf_counter() - t0
s = m._singular_values.astype(np.float64)
V = np.asarray(m._V["left"][:, :40])
out = {"factor": os.environ.get("XMCA_CHOLESKY_FACTOR", "cholesky"), "solve_s": dt, "stages_ms": m._device().timings(),
       "evd": m._device().solve_info(), "orth_left_40": float(np.abs(V.conj().T @ V - np.eye(40)).max())}
np.save("/tmp/sigma_%s.npy" % out["factor"], s)
other = "/tmp/sigma_%s.npy" % ("cholesky" if out["factor"] == "0" else "0")
if os.path.exists(other):
    o = np.load(other); keep = o > 1e-9 * o[0]
    out["sigma_rel_diff_between_factors"] = float(np.max(np.abs(s[keep] - o[keep]) / o[keep]))
print(json.dumps(out))
