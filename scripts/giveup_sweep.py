"""Give-ups of the persistent reduction in two-lane rule_n over a few sizes (0 expected):  giveup_sweep.py [calls]"""
import sys, numpy as np
sys.path.insert(0, ".")
from xmca_amd import _hip
h = _hip.Handle(0); lib = _hip.load_library()
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 6
for T, Nx, Ny, cplx in ((900, 2200, 1700, True), (900, 2200, 1700, False), (700, 2200, 1700, True), (1300, 3000, 2600, True), (2000, 5000, 4000, True), (1500, 4000, 3000, False)):
    args = (T, Nx, Ny, 2, cplx, False, 0, 0, 1e-8)
    ref, _ = h.rule_n(*args, 0, 8, 3, np.float64, T)
    g0 = lib.xmca_persistent_giveups()
    same = True
    for r in range(calls):
        sp, _ = h.rule_n(*args, 0, 8, 3, np.float64, T)
        same = same and np.array_equal(sp, ref)
    print(T, cplx, "giveups in %d runs:" % (8 * calls), lib.xmca_persistent_giveups() - g0, "same bits:", same, flush=True)
