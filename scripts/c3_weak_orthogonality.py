#!/usr/bin/env python3
"""Full-size C3: orthonormality of the LEFT and RIGHT singular vectors over all non-null modes (computed here on the host
from the downloaded vectors, in blocks) - for the choice of the field factor (XMCA_CHOLESKY_FACTOR)."""
import json, os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from golden_inputs import make_input
from xmca_amd.array import MCA
m = MCA(*make_input("c3_full"))
t0 = time.perf_counter(); m.solve(complexify=True); dt = time.perf_counter() - t0
out = {"factor": os.environ.get("XMCA_CHOLESKY_FACTOR", "default"), "solve_s": dt}
for key in ("left", "right"):
    V = np.asarray(m._V[key][:, :2500])
    G = V.conj().T @ V
    d = np.abs(G - np.eye(2500))
    out[key] = {"max_all": float(d.max()), "max_lead20_vs_all": float(d[:20].max()), "max_weak_weak": float(d[20:, 20:].max()),
                "argmax": [int(x) for x in np.unravel_index(d.argmax(), d.shape)]}
print(json.dumps(out))
