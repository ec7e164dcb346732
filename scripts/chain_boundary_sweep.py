"""Eigenvalues by the chained persistent reduction at sizes around every instantiation / cut boundary: against numpy and against the
one-launch form (bit for bit).  chain_boundary_sweep.py"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xmca_amd import _hip
h = _hip.Handle(0)
bad = 0
for cplx in (False, True):
    sizes = [384, 385, 511, 512, 513, 767, 768, 769, 1023, 1024, 1025, 1279, 1281, 1535, 1536, 1537, 1791, 2047, 2048, 2049, 2303, 2305, 2559, 2560]
    if not cplx:
        sizes += [2561, 2815, 3071, 3072]
    for n in sizes:
        rng = np.random.default_rng(n + cplx)
        X = rng.standard_normal((n, n + 50))
        if cplx:
            X = X + 1j * rng.standard_normal((n, n + 50))
        G = X @ X.conj().T
        os.environ["XMCA_TRD_CHAIN"] = "1"
        lam, _ = h.eigh(G, vectors=False)
        info = h.reduction_info()
        os.environ["XMCA_TRD_CHAIN"] = "0"
        one, _ = h.eigh(G, vectors=False)
        ref = np.linalg.eigvalsh(G)[::-1]
        err = float(np.max(np.abs(lam - ref)) / ref[0])
        ok = np.array_equal(lam, one) and err < 1e-13
        bad += not ok
        print(n, "complex" if cplx else "real", "links", info.count("trd_resident_kernel<"), "err %.1e" % err, "same bits" if np.array_equal(lam, one) else "DIFFERENT BITS", "" if ok else "<-- FAIL", flush=True)
print("failures:", bad, "giveups:", _hip.load_library().xmca_persistent_giveups())
