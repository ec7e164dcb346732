"""Why is the C4 Gram slower inside rule_n (9.1 ms) than alone (7.7 ms)?  The same product alone, back to back, and right after a
tridiagonal reduction (a 22-27 ms latency-bound kernel: clocks / power state)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, '.')
from xmca_amd import _hip
h = _hip.Handle(0)
rng = np.random.default_rng(1)
X = rng.standard_normal((5000, 20000))
h.set_field(0, X)
h.bench_gram(0, 1)
print("back to back x4:", h.bench_gram(0, 4)["kernel_ms"])
A = rng.standard_normal((2920, 3000)); G = A @ A.T
h2 = _hip.Handle(0)
for i in range(4):
    h2.eigh(G, vectors=False)
    print("after a reduction on another handle:", h.bench_gram(0, 1)["kernel_ms"])
for i in range(3):
    time.sleep(0.05)
    print("after 50 ms idle:", h.bench_gram(0, 1)["kernel_ms"])
