#!/usr/bin/env python3
"""Only the covariance (Gram) GEMM of a configuration, for counter passes: gram_only.py c2|c5 [reps]
c2: X X^T of the T = 2920 x N = 10 000 float64 field of generator A; c5: T = 1200 x N = 1 036 800 float32."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xmca_amd import _hip
which = sys.argv[1] if len(sys.argv) > 1 else "c2"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
h = _hip.Handle(0)
if which == "c2":
    import bench
    X = bench.gen_A(2920, 10000)
    X -= X.mean(axis=0)
else:
    X = np.random.default_rng(5).standard_normal((1200, 1_036_800), dtype=np.float32)
h.set_field(0, X)
T, N = X.shape
del X
h.bench_gram(0, 1)
g = h.bench_gram(0, reps)
print(json.dumps({"config": which, "T": T, "N": N, "reps": reps, **g}))
