import sys, time, numpy as np
sys.path.insert(0, '.')
from xmca_amd import _hip
h = _hip.Handle(0)
h.rule_n(5000, 20000, 15000, 2, True, False, 0, 0, 1e-8, 0, 2, 1, np.float64, 5000)
h.reset_timings()
t0 = time.perf_counter()
h.rule_n(5000, 20000, 15000, 2, True, False, 0, 0, 1e-8, 0, 6, 1, np.float64, 5000)
dt = time.perf_counter() - t0
print("ms per surrogate", 1e3 * dt / 6, {k: round(v / 6, 2) for k, v in h.timings().items()})
