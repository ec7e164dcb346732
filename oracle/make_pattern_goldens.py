#!/usr/bin/env python3
"""Golden vectors of the correlation maps (xmca/array.py:1188-1261, tools/array.py:76-88) from the REAL reference.

Run (build container only):  PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python oracle/make_pattern_goldens.py

`homogeneous_patterns(6)` / `heterogeneous_patterns(6)` of nicrie/xmca v1.4.2 on seeded inputs - real, complexified,
Varimax- and Promax-rotated, one field, and the reference's own sst/prcp fixture with its NaN columns - written to
tests/golden/pattern_cases.npz (r and p maps, field-shaped, NaN columns re-inserted as the reference does).
Same import shims as oracle/make_goldens.py.  Nothing but the .npz travels.
"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
sys.path.insert(0, os.path.join(REPO, "oracle"))

from make_goldens import import_reference, OUT  # noqa: E402
from golden_inputs import make_input  # noqa: E402

CASES = [  # tag, input, complexify, rotate(n_rot, power, tol) or None
    ("wide_both_std", "wide_both", False, None),
    ("wide_both_cplx", "wide_both", True, None),
    ("wide_both_std_rot6p1", "wide_both", False, (6, 1, 1e-8)),
    ("wide_both_cplx_rot6p4", "wide_both", True, (6, 4, 1e-8)),
    ("unit_both_std_rot10p4", "unit_both", False, (10, 4, 1e-8)),
    ("wide_left_std", "wide_left", False, None),
    ("sst_prcp_std_rot10p1", "sst_prcp", False, (10, 1, 1e-5)),
    ("sst_prcp_cplx", "sst_prcp", True, None),
]


def main():
    MCA, _, _ = import_reference()
    out = {}
    for tag, name, cplx, rot in CASES:
        m = MCA(*make_input(name))
        m.solve(complexify=cplx)
        if rot:
            m.rotate(*rot)
        kinds = [("hom", m.homogeneous_patterns)]
        if len(m._keys) == 2:
            kinds.append(("het", m.heterogeneous_patterns))
        for kind, fn in kinds:
            r, p = fn(6)
            for k in m._keys:
                out["%s__%s_r_%s" % (tag, kind, k)] = r[k]
                out["%s__%s_p_%s" % (tag, kind, k)] = p[k]
        print(tag, {k: v.shape for k, v in r.items()})
    dst = os.path.join(OUT, "pattern_cases.npz")
    np.savez_compressed(dst, **out)
    print("written", dst, "%.2f MB" % (os.path.getsize(dst) / 1e6))


if __name__ == "__main__":
    main()
