#!/usr/bin/env python3
"""Which form of the reduction ran (resident / launch per column) and how long, for small complex problems."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xmca_amd import _hip

h = _hip.Handle(0)
for n in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "400,500,1000,1024,1100").split(",")]:
    for cplx in (False, True):
        rng = np.random.default_rng(n)
        X = rng.standard_normal((n, 2 * n)) + (1j * rng.standard_normal((n, 2 * n)) if cplx else 0)
        G = X @ X.conj().T
        h.eigh(G, vectors=False)
        h.reset_timings()
        h.eigh(G, vectors=False)
        tm = h.timings()
        print(json.dumps({"n": n, "cplx": cplx, **{k: v for k, v in tm.items() if k.startswith("trd") or k.startswith("eigh")}}), flush=True)
