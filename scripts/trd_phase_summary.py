#!/usr/bin/env python3
"""Phase stamps of the resident tridiagonalisation (XMCA_TRD_PROF=file: s_memtime of workgroup 0, one line per column - the gap
to the previous column's first stamp, then the stamps of the column relative to its first) -> mean core-clock cycles per phase
and quarter of the columns.  usage: trd_phase_summary.py file..."""
import sys
import numpy as np

NAMES = ["gather + w", "column, norm", "reflector", "pass", "drain + flag", "poll"]
for f in sys.argv[1:]:
    a = np.loadtxt(f)
    n = a.shape[0]
    st = a[:, 1:7].copy()
    st[st > 1e15] = np.nan                      # (a stamp that was not taken: the tagged exchange has no drain / poll)
    d = np.diff(np.concatenate([np.zeros((n, 1)), st], axis=1), axis=1)
    print("%s: n = %d, %.0f cycles per column" % (f, n, a[1:, 0].mean()))
    for lo, hi in ((0, n // 4), (n // 4, n // 2), (n // 2, 3 * n // 4), (3 * n // 4, n - 1)):
        cols = "  ".join("%s %5.0f" % (nm, np.nanmean(d[lo:hi, q])) for q, nm in enumerate(NAMES) if not np.all(np.isnan(d[lo:hi, q])))
        print("  columns %4d-%4d: %s   | column %5.0f" % (lo, hi, cols, a[lo + 1:hi + 1, 0].mean()))
