#!/usr/bin/env python3
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xmca_amd import _hip
h = _hip.Handle(0)
T, N = 2920, 10000
dt = np.float32 if len(sys.argv) > 1 and sys.argv[1] == "f32" else np.float64
X = np.random.default_rng(0).standard_normal((T, N)).astype(dt)
h.set_field(0, X)
g = h.bench_gram(0, 3)
print(json.dumps(g))
