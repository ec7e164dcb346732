"""CPU: the numpy oracle (oracle/ref_numpy.py) against the golden vectors generated from the REAL reference
(oracle/make_goldens.py) and against the reference's own netCDF goldens (converted to
tests/golden/reference_fixtures.npz).  This is what pins the oracle; it runs without a GPU."""
import os

import numpy as np
import pytest

from conftest import align_modes
from golden_inputs import GOLDEN_DIR, make_input
from oracle import ref_numpy as O


def _rel(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def _load(name):
    return np.load(os.path.join(GOLDEN_DIR, name))


@pytest.mark.parametrize("case,cplx", [("std", False), ("cplx", True)])
def test_reference_own_fixtures(case, cplx):
    """tests/integration/test_integration_xarray.py:49-85: first 100 modes, atol = rtol = 1e-3."""
    fx = _load("reference_fixtures.npz")
    om = O.OracleModel(fx["sst"], fx["prcp"])
    om.solve(complexify=cplx)
    gold_s = fx[case + "_singular_values"]
    assert om.rank == 155 and gold_s.shape == (155,)
    assert np.allclose(om.singular_values[:100], gold_s[:100], rtol=1e-3, atol=1e-3)
    for i, f in enumerate(["sst", "prcp"]):
        gold = fx[case + "_" + f + "_eofs"].reshape(162, -1)
        assert np.array_equal(~np.isnan(gold[:, 0].real), om.valid[i])          # NaN columns (7 of 162 for sst)
        gold = gold[om.valid[i]][:, :100]
        mine, _ = align_modes(om.V[i][:, :100], gold)
        # modulus always; aligned values for the leading, well separated modes
        assert np.allclose(np.abs(om.V[i][:, :100]), np.abs(gold), atol=1e-3)
        assert np.allclose(mine[:, :10], gold[:, :10], atol=1e-3)


SOLVE = ["unit_left", "unit_both", "wide_left", "wide_both", "wide_both_f32", "mixed_both", "sst_prcp"]


@pytest.mark.parametrize("cplx", [False, True])
@pytest.mark.parametrize("name", SOLVE)
def test_solve_goldens(name, cplx):
    g = _load("solve_cases.npz")
    tag = name + ("_cplx" if cplx else "_std") + "__"
    om = O.OracleModel(*make_input(name))
    om.solve(complexify=cplx)
    assert om.rank == int(g[tag + "rank"])
    assert _rel(om.singular_values, g[tag + "singular_values"]) < 1e-10
    assert _rel(om.total_covariance, g[tag + "total_covariance"]) < 1e-10
    for i, k in enumerate(["left", "right"][:len(om.V)]):
        gv = g[tag + "V_" + k]
        assert _rel(np.abs(om.V[i][:, :gv.shape[1]]), np.abs(gv)) < 1e-7


ROT = [("unit_both", False, 10, 1, 1e-8), ("unit_both", False, 10, 4, 1e-8), ("unit_left", False, 10, 1, 1e-8),
       ("unit_both", False, 10, 1, 1e-5), ("wide_both", False, 6, 1, 1e-8), ("wide_both", False, 6, 4, 1e-8),
       ("wide_both", True, 6, 4, 1e-8), ("wide_left", True, 6, 2, 1e-8), ("unit_both", True, 10, 4, 1e-5),
       ("sst_prcp", False, 10, 1, 1e-5), ("sst_prcp", True, 10, 4, 1e-5)]


@pytest.mark.parametrize("name,cplx,n_rot,power,tol", ROT)
def test_rotate_goldens(name, cplx, n_rot, power, tol):
    g = _load("rotate_cases.npz")
    tag = "%s_%s_n%d_p%d_t%g__" % (name, "cplx" if cplx else "std", n_rot, power, tol)
    om = O.OracleModel(*make_input(name))
    om.solve(complexify=cplx)
    out = om.rotate(n_rot, power, tol)
    assert out["n_iter"] == int(g[tag + "n_iter"])
    assert _rel(out["R"], g[tag + "R"]) < 1e-9
    assert _rel(out["Phi"], g[tag + "Phi"]) < 1e-9
    assert _rel(out["variance"], g[tag + "variance"]) < 1e-9
    assert np.array_equal(out["var_idx"], g[tag + "var_idx"])


@pytest.mark.parametrize("tag", ["r4", "r10", "r10p4", "c4", "c10p4", "c10p2"])
def test_direct_rotation_goldens(tag):
    g = _load("rotation_direct.npz")
    A = make_input("loadings_" + tag)[0]
    B, R, Phi, it = O.promax(A, int(g[tag + "_power"]))
    assert it == int(g[tag + "_n_iter"])
    assert _rel(B, g[tag + "_B"]) < 1e-10 and _rel(R, g[tag + "_R"]) < 1e-10 and _rel(Phi, g[tag + "_Phi"]) < 1e-10
    Bv, Rv, _ = O.varimax(A)
    assert _rel(Bv, g[tag + "_Bv"]) < 1e-10 and _rel(Rv, g[tag + "_Rv"]) < 1e-10


def test_non_convergence_raises():
    with pytest.raises(RuntimeError):
        O.promax(make_input("loadings_noconv")[0], 4)
    with pytest.raises(ValueError):
        O.rotate([np.eye(4)], np.ones(4), 1)
    with pytest.raises(ValueError):
        O.rotate([np.eye(4)], np.ones(4), 2, power=0)


@pytest.mark.parametrize("tag,name,cplx,rot", [("eof_std", "unit_left", False, None), ("mca_std", "unit_both", False, None),
                                               ("mca_rot", "small_both", False, (4, 1)), ("mca_cplx", "small_both", True, None)])
def test_rule_n_goldens(tag, name, cplx, rot):
    """same numpy stream as the reference (np.random.seed(1234); left then right per run, array.py:1755-1756)."""
    g = _load("rule_n_cases.npz")
    om = O.OracleModel(*make_input(name))
    om.solve(complexify=cplx)
    if rot:
        om.rotate(*rot)
    np.random.seed(1234)
    mine = O.rule_n(om, 3)
    assert mine.shape == g[tag].shape and _rel(mine, g[tag]) < 1e-10


@pytest.mark.parametrize("name,cplx,n_rot,power", [("c1_standin", False, 10, 1), ("c2_full", False, 10, 1), ("c3_reduced", True, 20, 4),
                                                   ("c5_scaled", False, 10, 1)])
def test_config_goldens(name, cplx, n_rot, power):
    """the BASELINE.json configurations (C2 at full size, C3 / C5 scaled) from the real reference
    (oracle/make_config_goldens.py): sigma, leading loadings, R, variance and the Varimax iteration count."""
    g = _load("config_cases.npz")
    g = {k[len(name) + 2:]: g[k] for k in g.files if k.startswith(name + "__")}
    fields = make_input(name)
    f32 = fields[0].dtype == np.float32
    om = O.OracleModel(*fields)
    om.solve(complexify=cplx)
    assert om.rank == int(g["rank"])
    assert _rel(om.singular_values, g["singular_values"]) < (1e-5 if f32 else 1e-10)
    stride = int(g["stride"])
    for i, k in enumerate(["left", "right"][:len(om.V)]):
        gv = g["V_" + k]
        mine, _ = align_modes(om.V[i][::stride, :gv.shape[1]], gv)
        assert _rel(mine, gv) < (1e-3 if f32 else 1e-6)           # (stored as float32 / complex64)
    out = om.rotate(n_rot, power)
    assert out["n_iter"] == int(g["n_iter"])
    assert _rel(out["variance"], g["variance"]) < (1e-4 if f32 else 1e-9)
    assert np.array_equal(out["var_idx"], g["var_idx"])
