#!/usr/bin/env python3
"""Generate tests/golden/*.npz by importing the REAL reference (build container only).

Run:  PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python oracle/make_goldens.py

* imports nicrie/xmca v1.4.2 unmodified from /root/reference with two import
  shims (statsmodels ThetaModel stub; ``np.product`` alias on numpy>=2),
* runs the cases of SURVEY.md section 8(c) on seeded inputs,
* asserts that ``oracle/ref_numpy.py`` reproduces every number (pins the oracle),
* converts the reference's own netCDF golden vectors
  (tests/integration/fixtures/{sst,prcp}.nc and {std,cplx}/*.nc) to .npz through
  /opt/conda/bin/python3.9 + h5py (the system interpreter has no netCDF reader),
* writes the vectors (data only: inputs by seed where possible, outputs always).

Nothing here travels to the GPU box except the .npz files it writes.
"""
import os
import subprocess
import sys
import types

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)


def import_reference():
    for name in ["statsmodels", "statsmodels.tsa", "statsmodels.tsa.forecasting",
                 "statsmodels.tsa.forecasting.theta"]:
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["statsmodels.tsa.forecasting.theta"].ThetaModel = object
    if not hasattr(np, "product"):
        np.product = np.prod
    sys.path.insert(0, REF)
    from xmca.array import MCA
    from xmca.tools.rotation import varimax, promax
    return MCA, varimax, promax


class SvdCounter:
    """Counts np.linalg.svd calls (= Varimax iterations inside promax)."""

    def __enter__(self):
        self.n = 0
        self._orig = np.linalg.svd

        def counting(*a, **k):
            self.n += 1
            return self._orig(*a, **k)
        np.linalg.svd = counting
        return self

    def __exit__(self, *exc):
        np.linalg.svd = self._orig


def rel(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


# ----------------------------------------------------------------------------
# seeded input generators (also used verbatim by the tests; keep in sync with
# tests/golden_inputs.py which is the copy the tests import)
# ----------------------------------------------------------------------------
sys.path.insert(0, os.path.join(REPO, "tests"))
from golden_inputs import make_input  # noqa: E402


def solve_case(MCA, name, complexify):
    from oracle import ref_numpy as O
    fields = make_input(name)
    m = MCA(*fields)
    m.solve(complexify=complexify)
    om = O.OracleModel(*fields)
    om.solve(complexify=complexify)
    # pin the oracle on this case
    assert rel(om.singular_values, m._singular_values) < 1e-12, name
    keys = list(m._V.keys())
    nkeep = min(20, m._analysis["rank"])
    for i, k in enumerate(keys):
        assert rel(np.abs(om.V[i][:, :nkeep]), np.abs(m._V[k][:, :nkeep])) < 1e-8, (name, k)
    out = {
        "singular_values": m._singular_values,
        "total_covariance": np.asarray(m._analysis["total_covariance"]),
        "total_squared_covariance": np.asarray(m._analysis["total_squared_covariance"]),
        "rank": np.asarray(m._analysis["rank"]),
    }
    for k in keys:
        out["V_" + k] = m._V[k][:, :nkeep]
    return m, om, out


def rotate_case(MCA, name, complexify, n_rot, power, tol):
    from oracle import ref_numpy as O
    fields = make_input(name)
    m = MCA(*fields)
    m.solve(complexify=complexify)
    with SvdCounter() as cnt:
        m.rotate(n_rot, power, tol)
    n_iter = cnt.n
    om = O.OracleModel(*fields)
    om.solve(complexify=complexify)
    oo = om.rotate(n_rot, power, tol)
    assert oo["n_iter"] == n_iter, (name, oo["n_iter"], n_iter)
    assert rel(oo["R"], m._rotation_matrix) < 1e-10
    assert rel(oo["Phi"], m._correlation_matrix) < 1e-10
    assert rel(oo["variance"], m._variance) < 1e-10
    keys = list(m._V.keys())
    out = {
        "n_iter": np.asarray(n_iter),
        "singular_values": m._singular_values[:n_rot],
        "R": m._rotation_matrix, "Phi": m._correlation_matrix,
        "variance": m._variance, "var_idx": m._var_idx,
        "explained_variance": m.explained_variance(),
        "correlation_matrix": m.correlation_matrix(),
    }
    eofs = m.eofs(n_rot)
    pcs = m.pcs(n_rot)
    for k in keys:
        out["V_" + k] = m._V[k][:, :n_rot]
        out["norm_" + k] = m._norm[k]
        out["eofs_" + k] = eofs[k]
        out["pcs_" + k] = pcs[k]
    return out


def direct_rotation_cases(varimax, promax):
    from oracle import ref_numpy as O
    out = {}
    specs = [("r4", 300, 4, False, 1), ("r10", 300, 10, False, 1), ("r10p4", 300, 10, False, 4),
             ("c4", 300, 4, True, 1), ("c10p4", 300, 10, True, 4), ("c10p2", 300, 10, True, 2)]
    for tag, n, p, cplx, power in specs:
        A = make_input("loadings_%s" % tag)[0]
        with SvdCounter() as cnt:
            B, R, Phi = promax(A, power)
        oB, oR, oPhi, oit = O.promax(A, power)
        assert oit == cnt.n and rel(oB, B) < 1e-10 and rel(oR, R) < 1e-10 and rel(oPhi, Phi) < 1e-10
        with SvdCounter() as cnt2:
            Bv, Rv = varimax(A)
        oBv, oRv, oitv = O.varimax(A)
        assert oitv == cnt2.n and rel(oBv, Bv) < 1e-10
        out.update({tag + "_B": B, tag + "_R": R, tag + "_Phi": Phi, tag + "_n_iter": np.asarray(cnt.n),
                    tag + "_Bv": Bv, tag + "_Rv": Rv, tag + "_power": np.asarray(power)})
    # a case that does NOT converge in 1000 iterations (pure complex noise, p=20)
    A = make_input("loadings_noconv")[0]
    try:
        promax(A, 4)
        raise AssertionError("expected the reference to raise")
    except RuntimeError:
        pass
    try:
        O.promax(A, 4)
        raise AssertionError("expected the oracle to raise")
    except RuntimeError:
        pass
    out["noconv_expected"] = np.asarray(1)
    return out


def rule_n_cases(MCA):
    from oracle import ref_numpy as O
    out = {}
    for tag, name, cplx, rot in [("eof_std", "unit_left", False, None),
                                 ("mca_std", "unit_both", False, None),
                                 ("mca_rot", "small_both", False, (4, 1)),
                                 ("mca_cplx", "small_both", True, None)]:
        fields = make_input(name)
        m = MCA(*fields)
        m.solve(complexify=cplx)
        om = O.OracleModel(*fields)
        om.solve(complexify=cplx)
        if rot:
            m.rotate(*rot)
            om.rotate(*rot)
        np.random.seed(1234)
        ref = m.rule_n(3)
        np.random.seed(1234)
        mine = O.rule_n(om, 3)
        assert ref.shape == mine.shape and rel(mine, ref) < 1e-10, (tag, ref.shape, mine.shape)
        out[tag] = ref
    return out


def convert_reference_fixtures():
    """Reference's own goldens -> npz via the conda interpreter (has h5py)."""
    code = r'''
import h5py, numpy as np, sys
fx = "/root/reference/tests/integration/fixtures/"
out = {}
with h5py.File(fx + "sst.nc", "r") as h: out["sst"] = h["sst"][...]
with h5py.File(fx + "prcp.nc", "r") as h: out["prcp"] = h["prcp"][...]
for case in ["std", "cplx"]:
    with h5py.File(fx + case + "/singular_values.nc", "r") as h:
        out[case + "_singular_values"] = h["singular values"][...]
    for f in ["sst", "prcp"]:
        with h5py.File(fx + case + "/" + f + "_eofs.nc", "r") as h:
            e = h[f + " eofs"][...]
            if e.dtype.names:  # compound complex
                e = e[e.dtype.names[0]] + 1j * e[e.dtype.names[1]]
            out[case + "_" + f + "_eofs"] = e
np.savez_compressed(sys.argv[1], **out)
'''
    dst = os.path.join(OUT, "reference_fixtures.npz")
    subprocess.run(["/opt/conda/bin/python3.9", "-c", code, dst], check=True)
    return dst


def main():
    os.makedirs(OUT, exist_ok=True)
    MCA, varimax, promax = import_reference()
    from oracle import ref_numpy as O

    # --- 6. the reference's own golden vectors + oracle pin on them ---------
    dst = convert_reference_fixtures()
    fx = np.load(dst)
    for case, cplx in [("std", False), ("cplx", True)]:
        om = O.OracleModel(fx["sst"], fx["prcp"])
        om.solve(complexify=cplx)
        gold_s = fx[case + "_singular_values"]
        err = rel(om.singular_values[:100], gold_s[:100])
        print("reference fixture %-4s: oracle sigma rel err (first 100) = %.2e" % (case, err))
        assert err < 1e-3            # the reference's own tolerance (test_integration_xarray.py:33-35)
        for i, f in enumerate(["sst", "prcp"]):
            gold = fx[case + "_" + f + "_eofs"].reshape(162, -1)[om.valid[i]]
            mod = np.abs(om.V[i][:, :100]) - np.abs(gold[:, :100])
            assert np.max(np.abs(mod)) < 1e-3, (case, f, np.max(np.abs(mod)))

    # --- 1. solve cases -----------------------------------------------------
    solve_out = {}
    for name in ["unit_left", "unit_both", "wide_left", "wide_both", "wide_both_f32", "mixed_both", "sst_prcp"]:
        for cplx in [False, True]:
            _, _, out = solve_case(MCA, name, cplx)
            tag = name + ("_cplx" if cplx else "_std")
            for k, v in out.items():
                solve_out[tag + "__" + k] = v
            print("solve", tag, "rank", int(out["rank"]))
    np.savez_compressed(os.path.join(OUT, "solve_cases.npz"), **solve_out)

    # --- 2. rotate cases ----------------------------------------------------
    rot_out = {}
    for name, cplx, n_rot, power, tol in [
            ("unit_both", False, 10, 1, 1e-8), ("unit_both", False, 10, 4, 1e-8),
            ("unit_left", False, 10, 1, 1e-8), ("unit_both", False, 10, 1, 1e-5),
            ("wide_both", False, 6, 1, 1e-8), ("wide_both", False, 6, 4, 1e-8),
            ("wide_both", True, 6, 4, 1e-8), ("wide_left", True, 6, 2, 1e-8),
            ("unit_both", True, 10, 4, 1e-5), ("sst_prcp", False, 10, 1, 1e-5),
            ("sst_prcp", True, 10, 4, 1e-5)]:
        tag = "%s_%s_n%d_p%d_t%g" % (name, "cplx" if cplx else "std", n_rot, power, tol)
        try:
            out = rotate_case(MCA, name, cplx, n_rot, power, tol)
        except RuntimeError as e:
            print("rotate", tag, "-> RuntimeError (not stored):", str(e)[:40])
            continue
        for k, v in out.items():
            rot_out[tag + "__" + k] = v
        print("rotate", tag, "iters", int(out["n_iter"]))
    np.savez_compressed(os.path.join(OUT, "rotate_cases.npz"), **rot_out)

    # --- 3. direct varimax / promax ----------------------------------------
    np.savez_compressed(os.path.join(OUT, "rotation_direct.npz"), **direct_rotation_cases(varimax, promax))

    # --- 5. rule_n ----------------------------------------------------------
    np.savez_compressed(os.path.join(OUT, "rule_n_cases.npz"), **rule_n_cases(MCA))

    tot = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
    print("golden fixtures written: %.2f MB" % (tot / 1e6))


if __name__ == "__main__":
    main()
