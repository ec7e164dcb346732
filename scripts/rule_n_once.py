import sys, numpy as np
sys.path.insert(0, '.')
from xmca_amd import _hip
h = _hip.Handle(0)
h.rule_n(5000, 20000, 15000, 2, True, False, 0, 0, 1e-8, 0, 1, 1, np.float64, 5000)
