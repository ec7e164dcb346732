#!/usr/bin/env python3
"""Full-size C3 (T = 5000 x (20 000, 15 000), complexify): singular values against the reference's (tests/golden/
config_c3_full.npz) - relative error by magnitude class, for DESIGN.md 1 (accuracy envelope of the analytic path)."""
import json, os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from golden_inputs import GOLDEN_DIR, make_input
from xmca_amd.array import MCA
name = "c3_real_full" if "real" in sys.argv[1:] else "c3_full"          # `real`: the same fields without complexify
g = np.load(os.path.join(GOLDEN_DIR, "config_%s.npz" % name))
gs = g[name + "__singular_values"]
m = MCA(*make_input(name))
m.solve(complexify=name == "c3_full")                                   # warm-up
m = MCA(*make_input(name))
t0 = time.perf_counter(); m.solve(complexify=name == "c3_full"); dt = time.perf_counter() - t0
s = m._singular_values.astype(np.float64)
keep = gs > 1e-9 * gs[0]
rel = np.abs(s[keep] - gs[keep]) / gs[keep]
ratio = gs[keep] / gs[0]
out = {"case": name, "solve_s": dt, "n_nonnull": int(keep.sum()), "sigma_1": float(gs[0]), "sigma_min_nonnull": float(gs[keep][-1]),
       "max_rel": float(rel.max()), "n_above_1e-5": int((rel > 1e-5).sum()), "n_above_1e-6": int((rel > 1e-6).sum()),
       "classes": [], "stages": m._handle.timings() if hasattr(m, "_handle") else None}
for lo, hi in [(1e-1, 2), (1e-2, 1e-1), (1e-3, 1e-2), (1e-4, 1e-3), (3e-5, 1e-4), (0, 3e-5)]:
    sel = (ratio >= lo) & (ratio < hi)
    if sel.any():
        out["classes"].append({"sigma_over_sigma1": [lo, hi], "modes": int(sel.sum()), "max_rel_err": float(rel[sel].max()),
                               "median_rel_err": float(np.median(rel[sel]))})
print(json.dumps(out))
