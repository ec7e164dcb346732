"""rule_n of a mid-sized complex two-field model (m = 451: the persistent complex reduction) repeated: every call must return
the same bits (generator keyed by (seed, run, side), fixed summation orders).  A mismatch is a race."""
import sys, time, numpy as np
sys.path.insert(0, '.')
from xmca_amd import _hip
h = _hip.Handle(0)
lib = _hip.load_library()
T, Nx, Ny = (int(a) for a in (sys.argv[1:4] if len(sys.argv) > 3 else (900, 2200, 1700)))
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 30
args = (T, Nx, Ny, 2, True, False, 0, 0, 1e-8)
ref, _ = h.rule_n(*args, 0, 8, 3, np.float64, T)
bad = 0
for r in range(reps):
    sp, kept = h.rule_n(*args, 0, 8, 3, np.float64, T)
    if not np.array_equal(sp, ref):
        bad += 1
        d = np.abs(sp - ref)
        print("rep", r, "MISMATCH max abs", d.max(), "rel", (d / np.maximum(np.abs(ref), 1e-300)).max(), "runs", np.nonzero(d.max(axis=1))[0], flush=True)
print("mismatches", bad, "of", reps, "giveups", lib.xmca_persistent_giveups())
