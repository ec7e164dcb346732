"""Bit-equality of repeated rule_n calls at one shape:  det_probe.py T Nx Ny cplx calls"""
import sys, numpy as np
sys.path.insert(0, ".")
from xmca_amd import _hip
h = _hip.Handle(0); lib = _hip.load_library()
T, Nx, Ny, cplx, calls = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), bool(int(sys.argv[4])), int(sys.argv[5])
args = (T, Nx, Ny, 2, cplx, False, 0, 0, 1e-8)
ref, _ = h.rule_n(*args, 0, 8, 3, np.float64, T)
g0 = lib.xmca_persistent_giveups()
bad = 0
for r in range(calls):
    h.reset_timings()
    sp, _ = h.rule_n(*args, 0, 8, 3, np.float64, T)
    tm = h.timings()
    if 'eigh' in tm or tm.get('trd_reduce_calls', 0) != 8:
        print('  call', r, 'route:', {k: round(v, 2) for k, v in tm.items() if k in ('eigh', 'trd_reduce_calls', 'cholesky', 'trd_resident_calls')}, flush=True)
    if not np.array_equal(sp, ref):
        bad += 1
        d = np.abs(sp - ref) / np.maximum(np.abs(ref), 1e-300)
        rows = np.where(np.any(sp != ref, axis=1))[0]
        print("  call", r, "differs in runs", rows.tolist(), "max rel", float(d.max()), flush=True)
print(T, cplx, "calls", calls, "giveups", lib.xmca_persistent_giveups() - g0, "differing calls", bad, flush=True)
