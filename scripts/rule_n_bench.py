#!/usr/bin/env python3
"""rule_n throughput on one GPU (surrogates generated, centered, complexified, solved on the device)."""
import os, sys, json, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xmca_amd import _hip
h = _hip.Handle(0)
cases = [("C4 (T=5000, 20000x15000, complex, unrotated)", 5000, 20000, 15000, 2, True, False, 0, 0, 12, np.float64),
         ("C4 f32 surrogates", 5000, 20000, 15000, 2, True, False, 0, 0, 12, np.float32),
         ("C2-shaped EOF (T=2920, N=10000, real, unrotated)", 2920, 10000, 0, 1, False, False, 0, 0, 12, np.float64),
         ("C2-shaped EOF rotated n_rot=10", 2920, 10000, 0, 1, False, True, 10, 1, 12, np.float64),
         ("C1 air_temperature-shaped (T=2920, 1325x675, real)", 2920, 1325, 675, 2, False, False, 0, 0, 24, np.float64)]
if len(sys.argv) > 1:
    cases = [c for i, c in enumerate(cases) if str(i) in sys.argv[1].split(",")]
for name, T, Nx, Ny, nf, cplx, rot, p, power, runs, dt in cases:
    rank = min(T, Nx if nf == 1 else min(Nx, Ny))
    n_out = p if rot else rank
    h.rule_n(T, Nx, Ny, nf, cplx, rot, p, power, 1e-8, 0, 4, 1, dt, n_out)       # warm-up (every lane)
    h.reset_timings()
    t0 = time.perf_counter()
    sp, kept = h.rule_n(T, Nx, Ny, nf, cplx, rot, p, power, 1e-8, 0, runs, 1, dt, n_out)
    dtm = time.perf_counter() - t0
    tm = {k: round(v / runs, 2) for k, v in h.timings().items()}
    print(json.dumps({"case": name, "runs": runs, "s_per_surrogate": dtm / runs, "surrogates_per_s": runs / dtm,
                      "kept": int(kept.sum()), "lanes_env": os.environ.get("XMCA_RULE_N_LANES"), "resident": os.environ.get("XMCA_TRD_RESIDENT", "1"), "stages_ms_per_run": tm, "sigma_head": [float(x) for x in sp[0][:3]]}))
