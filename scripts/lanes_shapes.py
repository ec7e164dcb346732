import sys, time, numpy as np
sys.path.insert(0, '.')
from xmca_amd import _hip
h = _hip.Handle(0)
cases = {"c2eof": (2920, 10000, 0, 1, False, False, 0, 0, 1e-8), "c2rot": (2920, 10000, 0, 1, False, True, 10, 1, 1e-8),
         "c4rot": (5000, 20000, 15000, 2, True, True, 20, 4, 1e-8), "mid": (2000, 6000, 4000, 2, True, False, 0, 0, 1e-8)}
a = cases[sys.argv[1]]; runs = int(sys.argv[2])
n_out = a[6] if a[5] else (a[0] if not a[4] or a[3] == 1 else a[0] // 2 + 1 if False else a[0])
try:
    h.rule_n(*a, 0, 4, 1, np.float64, n_out)
except ValueError:
    n_out = a[0] // 2 + 1
    h.rule_n(*a, 0, 4, 1, np.float64, n_out)
t0 = time.perf_counter(); h.rule_n(*a, 0, runs, 1, np.float64, n_out)
print(sys.argv[1], "ms per surrogate %.2f" % (1e3 * (time.perf_counter() - t0) / runs))
