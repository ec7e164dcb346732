import sys, time, json, numpy as np
sys.path.insert(0, '.')
from xmca_amd import _hip
h = _hip.Handle(0)
args = (800, 2000, 0, 1, False, True, 6, 1, 1e-8)
h.rule_n(*args, 0, 4, 5, np.float64, 6)
for rep in range(4):
    g0 = _hip.load_library().xmca_persistent_giveups(); h.reset_timings(); t0 = time.perf_counter()
    sp, kept = h.rule_n(*args, 0, 16, 5, np.float64, 6); dt = time.perf_counter() - t0
    tm = h.timings()
    print(json.dumps({'seconds': dt, 'giveups': _hip.load_library().xmca_persistent_giveups() - g0, 'kept': int(kept.sum()), 'tm': {k: round(v, 2) for k, v in tm.items()}}))
