"""Bit-equality of concurrent values-only eigh calls (one handle = one stream per thread): eigh_race.py n cplx threads calls"""
import sys, threading, numpy as np
sys.path.insert(0, ".")
from xmca_amd import _hip
n, cplx, nthr, calls = int(sys.argv[1]), bool(int(sys.argv[2])), int(sys.argv[3]), int(sys.argv[4])
rng = np.random.default_rng(1)
X = rng.standard_normal((n, 2 * n))
if cplx:
    X = X + 1j * rng.standard_normal((n, 2 * n))
G = X @ X.conj().T
hs = [_hip.Handle(0) for _ in range(nthr)]
ref, _ = hs[0].eigh(G, vectors=False)
bad = [0] * nthr
def work(i):
    for c in range(calls):
        lam, _ = hs[i].eigh(G, vectors=False)
        if not np.array_equal(lam, ref):
            bad[i] += 1
            print("  thread", i, "call", c, "max rel", float(np.max(np.abs(lam - ref) / np.abs(ref))), flush=True)
ts = [threading.Thread(target=work, args=(i,)) for i in range(nthr)]
[t.start() for t in ts]; [t.join() for t in ts]
print(n, cplx, "threads", nthr, "calls", calls, "differing", sum(bad), "giveups", _hip.load_library().xmca_persistent_giveups(), flush=True)
