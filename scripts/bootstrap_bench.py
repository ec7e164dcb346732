#!/usr/bin/env python3
"""MCA.bootstrapping throughput on one GPU (replicates resampled, centered, solved on the device; several in flight)."""
import os, sys, json, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from golden_inputs import gen_A, gen_B
from xmca_amd.array import MCA
cases = [("C2-shaped EOF (T=2920, N=10000)", [gen_A(2920, 10000)], False, None, 8),
         ("MCA T=2000 x (6000, 4000), complexify, rotate(10, 1)", list(gen_B(2000, 6000, 4000, geometric=True)), True, (10, 1), 8),
         ("air_temperature-shaped (T=2920, 1325 x 675)", list(gen_B(2920, 1325, 675)), False, None, 16)]
for name, fields, cplx, rot, runs in cases:
    m = MCA(*fields)
    m.solve(complexify=cplx)
    if rot:
        m.rotate(*rot)
    np.random.seed(1)
    m.bootstrapping(4, n_modes=5, block_size=5)          # warm-up (every lane)
    np.random.seed(1)
    t0 = time.perf_counter()
    out = m.bootstrapping(runs, n_modes=5, block_size=5)
    dt = time.perf_counter() - t0
    print(json.dumps({"case": name, "runs": runs, "lanes": os.environ.get("XMCA_RULE_N_LANES", "default"),
                      "s_per_replicate": dt / runs, "replicates_per_s": runs / dt, "head": [float(x) for x in out[:2, 0]]}))
