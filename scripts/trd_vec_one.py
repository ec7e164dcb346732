#!/usr/bin/env python3
"""One eigen-decomposition with vectors of a C2-sized Gram matrix (for rocprofv3 kernel traces): trd_vec_one.py [n] [c]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xmca_amd import _hip
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2920
cplx = len(sys.argv) > 2
rng = np.random.default_rng(0)
X = (rng.standard_normal((n, 20)) * np.linspace(10, 1, 20)) @ rng.standard_normal((20, 3 * n)) + rng.standard_normal((n, 3 * n))
if cplx:
    X = X + 1j * rng.standard_normal((n, 3 * n))
X -= X.mean(axis=0)
G = X @ X.conj().T
h = _hip.Handle(0)
for _ in range(3):
    h.reset_timings()
    h.eigh(G)
print(h.timings(), h.last_eigh_info)
