"""A few C4 surrogates through xmca_rule_n (for kernel traces):  rule_n_once.py [runs]   (XMCA_RULE_N_LANES as set)"""
import sys, time, numpy as np
sys.path.insert(0, '.')
from xmca_amd import _hip
h = _hip.Handle(0)
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 2
h.rule_n(5000, 20000, 15000, 2, True, False, 0, 0, 1e-8, 0, 2, 1, np.float64, 5000)
t0 = time.perf_counter()
h.rule_n(5000, 20000, 15000, 2, True, False, 0, 0, 1e-8, 0, runs, 1, np.float64, 5000)
print("s per surrogate", (time.perf_counter() - t0) / runs)
