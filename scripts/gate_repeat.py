"""Rotated rule_n at n = 800 with four lanes, repeated: give-ups of the persistent kernels per call (tests/test_gpu_rule_n.py gate test)."""
import sys, time, json, os
import numpy as np
sys.path.insert(0, '.')
from xmca_amd import _hip
h = _hip.Handle(0)
lib = _hip.load_library()
args = (800, 2000, 0, 1, False, True, 6, 1, 1e-8)
h.rule_n(*args, 0, 4, 5, np.float64, 6)
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    g0 = lib.xmca_persistent_giveups(); t0 = time.perf_counter()
    sp, kept = h.rule_n(*args, 0, 16, 5, np.float64, 6)
    print(rep, "seconds %.3f giveups %d kept %d" % (time.perf_counter() - t0, lib.xmca_persistent_giveups() - g0, int(kept.sum())), flush=True)
