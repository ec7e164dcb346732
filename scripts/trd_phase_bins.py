#!/usr/bin/env python3
"""Phase stamps of the (chained) resident tridiagonalisation by bins of columns (XMCA_TRD_PROF=file, see trd_phase_summary.py).
usage: trd_phase_bins.py file [bin=256]"""
import sys
import numpy as np

NAMES = ["gather+w", "col,norm", "reflector", "pass", "drain+flag", "poll"]
a = np.loadtxt(sys.argv[1])
w = int(sys.argv[2]) if len(sys.argv) > 2 else 256
n = a.shape[0]
st = a[:, 1:7].copy()
st[st > 1e15] = np.nan
d = np.diff(np.concatenate([np.zeros((n, 1)), st], axis=1), axis=1)
tot = a[:, 0].copy()
tot[tot > 1e9] = np.nan          # (first column of a later launch: the gap spans the launch boundary)
print("%s: n = %d, %.0f cycles per column (median %.0f)" % (sys.argv[1], n, np.nanmean(tot[1:]), np.nanmedian(tot[1:])))
for lo in range(0, n - 1, w):
    hi = min(lo + w, n - 1)
    cols = "  ".join("%s %5.0f" % (nm, np.nanmedian(d[lo:hi, q])) for q, nm in enumerate(NAMES) if not np.all(np.isnan(d[lo:hi, q])))
    print("  columns %4d-%4d: %s | column %5.0f" % (lo, hi, cols, np.nanmedian(tot[lo + 1:hi + 1])))
