"""`xmca_amd.array.MCA` end to end on the GPU, written like the reference's own tests
(tests/unit/test_array.py, tests/integration/test_integration_xarray.py) plus parity against golden vectors
generated from the reference (tests/golden/rotate_cases.npz, rule_n_cases.npz).

Gauge: singular vectors are defined up to a per-mode sign/phase shared by the left and right vector; with
V = V_ref D the rotation matrices transform as R = D^H R_ref D, Phi = D^H Phi_ref D while norms, variance,
mode order and the Varimax iteration count are invariant.  Tolerance: 1e-5 (north_star).
"""
import os

import numpy as np
import pytest

from conftest import align_modes
from golden_inputs import GOLDEN_DIR, make_input
from oracle import ref_numpy as O
from xmca_amd.array import MCA

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _rel(a, b):
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


@pytest.fixture(scope="module")
def rot_gold():
    return np.load(os.path.join(GOLDEN_DIR, "rotate_cases.npz"))


# ----------------------------------------------------------------------------------------------
# reference unit tests (tests/unit/test_array.py:20-50)
# ----------------------------------------------------------------------------------------------
def test_input_validation():
    left, right = make_input("unit_both")
    MCA()
    MCA(left)
    MCA(left, right)
    with pytest.raises(ValueError):
        MCA(left, right, right)
    with pytest.raises(ValueError):
        MCA(left[:-1], right)
    bad = left.copy()
    bad[3] = np.nan
    with pytest.raises(ValueError):
        MCA(bad)
    with pytest.raises(TypeError):
        MCA(list(left))


def test_shapes_like_reference_unit_test():
    left, right = make_input("unit_both")
    m = MCA(left, right)
    m.solve()
    rank = min(np.prod(left.shape[1:]), np.prod(right.shape[1:]))
    pcs, eofs = m.pcs(), m.eofs()
    assert pcs['left'].shape == (500, rank) and pcs['right'].shape == (500, rank)
    assert eofs['left'].shape == left.shape[1:] + (rank,)
    assert eofs['right'].shape == right.shape[1:] + (rank,)
    assert m._analysis['rank'] == rank and m._analysis['n_rot'] == rank
    assert np.array_equal(m.rotation_matrix(), np.eye(rank)) and np.array_equal(m.correlation_matrix(), np.eye(rank))


def test_getters_before_solve_raise():
    m = MCA(make_input("unit_left")[0])
    with pytest.raises(RuntimeError):
        m.singular_values()
    with pytest.raises(RuntimeError):
        m.eofs()
    with pytest.raises(RuntimeError):
        MCA().solve()


# ----------------------------------------------------------------------------------------------
# rotate(): golden parity incl. the iteration count
# ----------------------------------------------------------------------------------------------
ROT_CASES = [("unit_both", False, 10, 1, 1e-8), ("unit_both", False, 10, 4, 1e-8), ("unit_left", False, 10, 1, 1e-8),
             ("unit_both", False, 10, 1, 1e-5), ("wide_both", False, 6, 1, 1e-8), ("wide_both", False, 6, 4, 1e-8),
             ("wide_both", True, 6, 4, 1e-8), ("wide_left", True, 6, 2, 1e-8), ("unit_both", True, 10, 4, 1e-5),
             ("sst_prcp", False, 10, 1, 1e-5), ("sst_prcp", True, 10, 4, 1e-5)]


@pytest.mark.parametrize("name,cplx,n_rot,power,tol", ROT_CASES)
def test_rotate_matches_reference(rot_gold, name, cplx, n_rot, power, tol):
    tag = "%s_%s_n%d_p%d_t%g__" % (name, "cplx" if cplx else "std", n_rot, power, tol)
    g = {k[len(tag):]: rot_gold[k] for k in rot_gold.files if k.startswith(tag)}
    fields = make_input(name)
    f32 = fields[0].dtype == np.float32
    m = MCA(*fields)
    m.solve(complexify=cplx)
    m.rotate(n_rot, power, tol)
    keys = m._keys
    # gauge of the unrotated vectors
    _, ph = align_modes(m._V['left'][:, :n_rot], g['V_left'])
    D = np.diag(ph)
    t = 1e-3 if f32 else TOL
    if f32:
        # the float32 reference solve feeds a visibly different (1e-6) matrix into Varimax: the stop
        # iteration may move by a few steps; everything else is compared at the reference's own 1e-3
        assert abs(m._varimax_iterations - int(g['n_iter'])) <= 3
    else:
        assert m._varimax_iterations == int(g['n_iter'])
    assert _rel(D @ m._rotation_matrix @ D.conj().T, g['R']) < t
    assert _rel(D @ m._correlation_matrix @ D.conj().T, g['Phi']) < t
    assert _rel(m._variance, g['variance']) < t
    assert np.array_equal(m._var_idx, g['var_idx'])
    for k in keys:
        assert _rel(m._norm[k], g['norm_' + k]) < t
    assert _rel(m.explained_variance(), g['explained_variance']) < (1e-3 if f32 else TOL)
    # rotated EOFs / PCs, aligned per (re-ordered) mode
    eofs, pcs = m.eofs(n_rot), m.pcs(n_rot)
    for k in keys:
        ge = g['eofs_' + k].reshape(-1, n_rot)
        me = eofs[k].reshape(-1, n_rot)
        ok = ~np.isnan(ge[:, 0])
        assert np.array_equal(np.isnan(me[:, 0]), ~ok)                   # NaN grid points are put back
        al, _ = align_modes(me[ok], ge[ok])
        assert _rel(al, ge[ok]) < (5e-3 if f32 else 10 * TOL)
        al, _ = align_modes(pcs[k], g['pcs_' + k])
        assert _rel(al, g['pcs_' + k]) < (5e-3 if f32 else 10 * TOL)


def test_rotate_argument_errors():
    m = MCA(*make_input("wide_both"))
    m.solve()
    with pytest.raises(ValueError):
        m.rotate(1)
    with pytest.raises(ValueError):
        m.rotate(5, power=0)


# ----------------------------------------------------------------------------------------------
# property tests of the reference's integration suite (test_integration_xarray.py:150-341, :368-502)
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cplx", [False, True])
def test_orthogonality_and_correlation(cplx):
    m = MCA(*make_input("sst_prcp"))
    m.solve(complexify=cplx)
    V = m._V
    k = 40
    for key in m._keys:
        assert np.allclose(V[key][:, :k].conj().T @ V[key][:, :k], np.eye(k), atol=1e-3)
    pcs = m.pcs(k)
    T = pcs['left'].shape[0]
    assert np.allclose(pcs['left'].conj().T @ pcs['right'] / (T - 1), np.eye(k), atol=1e-3)
    m.rotate(10, 1, tol=1e-5)
    pcs = m.pcs(10)
    assert np.allclose(pcs['left'].conj().T @ pcs['right'] / (T - 1), np.eye(10), atol=1e-3)       # Varimax keeps PCs uncorrelated
    eofs = m.eofs(10)['left'].reshape(-1, 10)
    eofs = eofs[~np.isnan(eofs[:, 0])]
    assert not np.allclose(eofs.conj().T @ eofs, np.eye(10), atol=1e-3)                            # ... but not the EOFs
    m.rotate(10, 4, tol=1e-5)
    pcs = m.pcs(10)
    assert not np.allclose(pcs['left'].conj().T @ pcs['right'] / (T - 1), np.eye(10), atol=1e-3)   # Promax does not


@pytest.mark.parametrize("normalize", [False, True])
def test_fields_round_trip(normalize):
    left, right = make_input("sst_prcp")
    m = MCA(left, right)
    if normalize:
        m.normalize()
    m.solve()
    m.rotate(10)
    back = m.fields(original_scale=True)
    assert np.allclose(back['left'], left, rtol=1e-3, atol=1e-3, equal_nan=True)
    assert np.allclose(back['right'], right, rtol=1e-3, atol=1e-3, equal_nan=True)


@pytest.mark.parametrize("cplx,n_rot,power", [(False, 0, 0), (False, 10, 1), (False, 10, 2)])
def test_predict_reproduces_pcs(cplx, n_rot, power):
    left, right = make_input("sst_prcp")
    m = MCA(left, right)
    m.solve(complexify=cplx)
    if n_rot:
        m.rotate(n_rot, power, tol=1e-5)
    n = n_rot or 20
    pcs = m.pcs(n)
    new = m.predict(left[:20], right[:20], n=n)
    for k in m._keys:
        assert np.allclose(new[k], pcs[k][:20], rtol=1e-3, atol=1e-3)
    with pytest.raises(ValueError):
        m.predict(left[0], right[0])


def test_truncate_and_patterns():
    m = MCA(*make_input("sst_prcp"))
    m.solve()
    m.rotate(10)
    with pytest.raises(ValueError):
        m.truncate(5)
    m.truncate(30)
    assert m._V['left'].shape[1] == 30 and m._analysis['is_truncated']
    r, p = m.heterogeneous_patterns(5)
    assert np.nanmax(np.abs(r['left'])) <= 1 + 1e-6 and np.nanmin(p['left']) >= 0
    r, _ = m.homogeneous_patterns(5)
    assert r['right'].shape == (9, 18, 5)
    rec = m.reconstructed_fields(slice(1, 5))
    assert rec['left'].shape == (492, 9, 18)
    assert m.rule_north(3).shape == (3,)


def test_apply_weights_and_extend_exp():
    left, right = make_input("wide_both")
    w = np.linspace(0.5, 1.5, left.shape[1])[None, :]
    m = MCA(left, right)
    m.apply_weights(left=w)
    m.solve()
    om = O.solve([O.flatten_and_center(left)[0] * w, O.flatten_and_center(right)[0]])
    assert _rel(m.singular_values(10), om["singular_values"][:10]) < TOL
    m2 = MCA(left, right)
    m2.solve(complexify=True, extend='exp', period=12)         # host extension + complex upload path
    assert m2._fields['left'].dtype == np.complex128 and m2.singular_values(3).shape == (3,)


# ----------------------------------------------------------------------------------------------
# rule_n: shape / normalisation like the reference; distribution vs the oracle fed with numpy normals
# ----------------------------------------------------------------------------------------------
def test_rule_n_shape_normalisation_and_reproducibility():
    m = MCA(*make_input("small_both"))
    m.solve()
    np.random.seed(5)
    a = m.rule_n(6)
    np.random.seed(5)
    b = m.rule_n(6)
    assert a.shape == (m._analysis['rank'], 6) and np.array_equal(a, b)
    assert np.allclose(a.sum(axis=0), m._get_variance().sum())
    assert m.rule_n(4, n_modes=3).shape == (3, 4)
    c = m.rule_n(6, seed=123)
    d = m.rule_n(3, seed=123)
    assert np.array_equal(c[:, :3], d)                           # a run's stream depends on (seed, run) only


@pytest.mark.parametrize("cplx,rot", [(False, None), (True, None), (False, (4, 1))])
def test_rule_n_distribution_matches_oracle(cplx, rot):
    """per-mode mean of 120 device surrogates vs 120 oracle surrogates: agreement within Monte-Carlo error."""
    fields = make_input("small_both")
    m = MCA(*fields)
    m.solve(complexify=cplx)
    om = O.OracleModel(*fields)
    om.solve(complexify=cplx)
    if rot:
        m.rotate(*rot)
        om.rotate(*rot)
    n = 120
    mine = m.rule_n(n, seed=7)
    rng = np.random.default_rng(11)
    ref = O.rule_n(om, n, normal=lambda shape: rng.standard_normal(shape))
    assert mine.shape[0] == ref.shape[0] and mine.shape[1] >= 0.9 * n
    k = min(6, ref.shape[0])
    se = np.sqrt(mine[:k].var(axis=1) / mine.shape[1] + ref[:k].var(axis=1) / ref.shape[1])
    assert np.all(np.abs(mine[:k].mean(axis=1) - ref[:k].mean(axis=1)) < 5 * se + 1e-12)


def test_bootstrapping_runs():
    m = MCA(*make_input("small_both"))
    m.solve()
    np.random.seed(0)
    out = m.bootstrapping(3, n_modes=4, on_left=True, on_right=True, block_size=2)
    assert out.shape == (4, 3) and np.all(out > 0)
    single = MCA(make_input("small_both")[0])
    single.solve()
    with pytest.raises(ValueError):
        single.bootstrapping(2, on_left=False, on_right=True)


# ----------------------------------------------------------------------------------------------
# PC projection on the device (`_get_U`, array.py:648-674)
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name,cplx,analytic", [("wide_both", False, "1"), ("wide_both", True, "1"), ("wide_both", True, "0"),
                                                ("wide_both_f32", True, "1"), ("small_both", True, "1"), ("unit_left", False, "1")])
def test_pcs_projection_matches_host_formula(name, cplx, analytic, monkeypatch):
    """pcs() = X~ V / sqrt(sigma) (array.py:391): the product runs on the fields resident on the device - real fields,
    the implicit analytic signal of the subspace path (U = W + i Ht W), stored complex planes (general path), f32."""
    monkeypatch.setenv("XMCA_ANALYTIC", analytic)
    fields = make_input(name)
    m = MCA(*fields)
    m.solve(complexify=cplx)
    k = 12
    pcs = m.pcs(k)
    X = m._get_X()                                           # host copy (scipy.signal.hilbert when complex)
    for key in m._keys:
        ref = X[key] @ m._V[key][:, :k] / np.sqrt(m._singular_values[:k])
        tol = 2e-4 if X[key].real.dtype == np.float32 else 1e-9
        assert pcs[key].shape == ref.shape
        assert _rel(pcs[key], ref) < tol


def test_pcs_projection_after_the_handle_was_used_elsewhere():
    """another model's solve() and rule_n() overwrite the resident fields: the projection uploads its own again."""
    a = MCA(*make_input("wide_both"))
    a.solve(complexify=True)
    a.rotate(6, 2)
    first = a.pcs(6)
    b = MCA(*make_input("small_both"))
    b.solve()
    b.rule_n(2, seed=3)
    again = a.pcs(6)
    for key in a._keys:
        assert _rel(again[key], first[key]) < 1e-12
    X = a._get_X()
    R = a.rotation_matrix(inverse_transpose=True)
    for key in a._keys:
        ref = ((X[key] @ a._V[key][:, :6] / np.sqrt(a._singular_values[:6])) @ R)[:, a._var_idx]
        assert _rel(first[key], ref) < 1e-9


# ----------------------------------------------------------------------------------------------
# correlation maps on the device (homogeneous / heterogeneous patterns, array.py:1188-1261)
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name,cplx", [("sst_prcp", False), ("sst_prcp", True), ("wide_both", False), ("wide_both_f32", True)])
def test_correlation_maps_match_pearsonr(name, cplx):
    """r = corr(real field, real PCs) as a device GEMM + column moments; the reference forms np.corrcoef of all N + m
    columns (tools/array.py:76-88).  NaN columns come back as NaN, p-values from the same beta distribution."""
    from xmca_amd.tools.array import pearsonr
    m = MCA(*make_input(name))
    m.solve(complexify=cplx)
    m.rotate(6, 2)
    X = m._get_X(real=True)
    pcs = m.pcs(6)
    other = {'left': 'right', 'right': 'left'}
    for maps, pair in ((m.homogeneous_patterns(6), {k: k for k in m._keys}), (m.heterogeneous_patterns(6), other)):
        rv, pv = maps
        for k in m._keys:
            r_ref, p_ref = pearsonr(X[k], pcs[pair[k]].real)
            valid = m._no_nan_index[k]
            r = rv[k].reshape(-1, 6)
            p = pv[k].reshape(-1, 6)
            tol = 2e-5 if X[k].dtype == np.float32 else 1e-10
            assert np.all(np.isnan(r[~valid])) and np.all(np.isnan(p[~valid]))
            assert np.max(np.abs(r[valid] - r_ref)) < tol
            assert np.max(np.abs(p[valid] - p_ref)) < 50 * tol


PATTERN_CASES = [  # tag of tests/golden/pattern_cases.npz, input, complexify, rotate(...) or None, where the gauge comes from
    ("wide_both_std", "wide_both", False, None, "solve:wide_both_std"),
    ("wide_both_cplx", "wide_both", True, None, "solve:wide_both_cplx"),
    ("wide_both_std_rot6p1", "wide_both", False, (6, 1, 1e-8), "rot:wide_both_std_n6_p1_t1e-08"),
    ("wide_both_cplx_rot6p4", "wide_both", True, (6, 4, 1e-8), "rot:wide_both_cplx_n6_p4_t1e-08"),
    ("unit_both_std_rot10p4", "unit_both", False, (10, 4, 1e-8), "rot:unit_both_std_n10_p4_t1e-08"),
    ("wide_left_std", "wide_left", False, None, "solve:wide_left_std"),
    ("sst_prcp_std_rot10p1", "sst_prcp", False, (10, 1, 1e-5), "rot:sst_prcp_std_n10_p1_t1e-05"),
    ("sst_prcp_cplx", "sst_prcp", True, None, "solve:sst_prcp_cplx"),
]


@pytest.mark.parametrize("tag,name,cplx,rot,gauge", PATTERN_CASES)
def test_correlation_maps_match_the_reference(rot_gold, tag, name, cplx, rot, gauge):
    """`homogeneous_patterns(6)` / `heterogeneous_patterns(6)` against the REFERENCE's own output
    (oracle/make_pattern_goldens.py -> tests/golden/pattern_cases.npz; xmca/array.py:1188-1261, tools/array.py:76-88):
    real, complexified, Varimax / Promax rotated, one field, NaN columns.  The maps correlate with the REAL part of the
    PCs, which depends on the sign / phase the SVD happened to give each mode - so the model's unrotated vectors are
    first put into the reference's gauge (one unit factor per mode, the same on both sides; vectors from the solve /
    rotate goldens), then everything is compared as is."""
    gold = np.load(os.path.join(GOLDEN_DIR, "pattern_cases.npz"))
    kind, gtag = gauge.split(":")
    gv = rot_gold if kind == "rot" else np.load(os.path.join(GOLDEN_DIR, "solve_cases.npz"))
    f32 = make_input(name)[0].dtype == np.float32
    m = MCA(*make_input(name))
    m.solve(complexify=cplx)
    Vref = gv[gtag + "__V_left"]
    nal = Vref.shape[1]
    V = {k: np.array(m._V[k]) for k in m._keys}
    _, ph = align_modes(V['left'][:, :nal], Vref)
    for k in m._keys:
        V[k][:, :nal] = V[k][:, :nal] / ph
    m._V = V
    if rot:
        m.rotate(*rot)
    maps = [("hom", m.homogeneous_patterns(6))]
    if len(m._keys) == 2:
        maps.append(("het", m.heterogeneous_patterns(6)))
    tol = 1e-3 if f32 else 1e-8
    for kind, (rv, pv) in maps:
        for k in m._keys:
            r_ref, p_ref = gold["%s__%s_r_%s" % (tag, kind, k)], gold["%s__%s_p_%s" % (tag, kind, k)]
            assert rv[k].shape == r_ref.shape and pv[k].shape == p_ref.shape
            nan = np.isnan(r_ref)
            assert np.array_equal(np.isnan(rv[k]), nan) and np.array_equal(np.isnan(pv[k]), np.isnan(p_ref))
            assert np.max(np.abs(rv[k][~nan] - r_ref[~nan])) < tol, (tag, kind, k)
            assert np.allclose(pv[k][~nan], p_ref[~nan], rtol=1e4 * tol, atol=1e-12), (tag, kind, k)


# ----------------------------------------------------------------------------------------------
# bootstrapping replicates on the device (array.py:1813-1952)
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name,single,cplx,rot,kw", [
    ("small_both", False, False, None, dict(on_left=True, on_right=True, block_size=2)),
    ("wide_both", False, False, (5, 2), dict(on_left=True, on_right=False, block_size=1)),
    ("wide_both", True, True, None, dict(on_left=True, on_right=False, block_size=4, replace=False)),
    ("wide_both", False, True, (4, 1), dict(on_left=False, on_right=True, block_size=1)),
    ("sst_prcp", False, False, None, dict(on_left=True, on_right=True, block_size=3, strategy='iterative'))])
def test_bootstrapping_device_replicates_equal_host_loop(name, single, cplx, rot, kw):
    """same numpy seed -> same block draws: the device replicates (cumulative row gather, centering, solve, rotate)
    must reproduce the reference's host loop (one MCA per replicate) to rounding."""
    fields = make_input(name)
    if single:
        fields = fields[:1]
    out = {}
    for host in (False, True):
        m = MCA(*fields)
        m.solve(complexify=cplx)
        if rot:
            m.rotate(*rot)
        m._bootstrap_on_host = host
        np.random.seed(5)
        out[host] = m.bootstrapping(3, n_modes=4, **kw)
    assert out[False].shape == out[True].shape
    scale = np.abs(out[True]).max()
    tol = 2e-5 if fields[0].dtype == np.float32 else 1e-8
    assert np.max(np.abs(out[False] - out[True])) < tol * scale


# ----------------------------------------------------------------------------------------------
# constructor preprocessing on the device (array.py:199-215) - opt-in
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name,cplx", [("wide_both", False), ("wide_both", True), ("wide_both_f32", False), ("unit_left", True)])
def test_device_preprocessing_equals_host_constructor(name, cplx):
    """MCA(..., preprocess='device'): means / stds / centered fields computed on the GPU and kept resident; the model
    must be indistinguishable from the host-preprocessed one (rounding of the float64 column means aside)."""
    fields = make_input(name)
    ref = MCA(*fields)
    dev = MCA(*fields, preprocess='device')
    assert dev._store_is_raw                               # nothing was centered on the host
    f32 = fields[0].dtype == np.float32
    tol = 1e-5 if f32 else 1e-12
    for k in ref._keys:
        assert np.array_equal(dev._no_nan_index[k], ref._no_nan_index[k])
        assert dev._field_means[k].dtype == ref._field_means[k].dtype
        assert np.allclose(dev._field_means[k], ref._field_means[k], rtol=tol, atol=tol * np.abs(ref._field_means[k]).max())
        assert np.allclose(dev._field_stds[k], ref._field_stds[k], rtol=tol)
    dev.solve(complexify=cplx)
    assert dev._store_is_raw                               # ... nor uploaded again for solve()
    ref.solve(complexify=cplx)                             # (same default handle: takes the resident fields over)
    k = 8
    assert _rel(dev._singular_values[:k], ref._singular_values[:k]) < (1e-4 if f32 else 1e-10)
    pd, pr = dev.pcs(k), ref.pcs(k)
    for key in ref._keys:
        ph = np.sum(np.conj(pr[key]) * pd[key], axis=0)
        ph /= np.abs(ph)
        assert _rel(pd[key] / ph, pr[key]) < (2e-3 if f32 else 1e-8)
    X = dev._get_X()                                       # first host access: fetched / recomputed, analytic signal built lazily
    assert not dev._store_is_raw
    Xr = ref._get_X()
    for key in ref._keys:
        assert X[key].dtype == Xr[key].dtype and _rel(X[key], Xr[key]) < (1e-5 if f32 else 1e-12)
    dev.solve()                                            # real again, fields now come from the host copy
    ref.solve()
    assert _rel(dev._singular_values[:k], ref._singular_values[:k]) < (1e-4 if f32 else 1e-10)


@pytest.mark.parametrize("own_handle", [False, True])
def test_device_preprocessing_drops_nan_columns_on_the_device(own_handle):
    """fields with NaN columns (land / sea masks): mask, compaction, means and centering all on the device
    (xmca_compact_field + xmca_center_field) - same model as the host constructor (array.py:191-215)."""
    from xmca_amd import _hip
    sst, prcp = make_input("sst_prcp")                     # NaN columns in both fields
    ref = MCA(sst, prcp)
    dev = MCA(sst, prcp, preprocess='device', **({"handle": _hip.Handle(0)} if own_handle else {}))
    assert dev._store_is_raw
    f32 = sst.dtype == np.float32
    for k in ref._keys:
        assert np.array_equal(dev._no_nan_index[k], ref._no_nan_index[k])
        assert np.allclose(dev._field_means[k], ref._field_means[k], rtol=1e-5 if f32 else 1e-12, atol=1e-5 if f32 else 1e-12)
        assert np.allclose(dev._field_stds[k], ref._field_stds[k], rtol=1e-5 if f32 else 1e-12)
    dev.solve()
    ref.solve()                                            # (default handle: takes the resident fields over unless dev has its own)
    assert _rel(dev._singular_values[:8], ref._singular_values[:8]) < (1e-4 if f32 else 1e-10)
    assert dev._V['left'].shape == ref._V['left'].shape and dev._V['right'].shape == ref._V['right'].shape
    X, Xr = dev._get_X(), ref._get_X()                     # downloaded (own handle) or recomputed from the raw input
    for key in ref._keys:
        assert X[key].shape == Xr[key].shape == (492, int(ref._no_nan_index[key].sum()))
        assert _rel(X[key], Xr[key]) < (1e-5 if f32 else 1e-12)
    pd, pr = dev.eofs(3), ref.eofs(3)                      # NaN columns come back as NaN in the spatial patterns
    for key in ref._keys:
        assert pd[key].shape == pr[key].shape and np.array_equal(np.isnan(pd[key]), np.isnan(pr[key]))


@pytest.mark.parametrize("own_handle", [False, True])
def test_device_preprocessing_weights_and_normalize_on_the_device(own_handle):
    """normalize() and per-column apply_weights() of a device-preprocessed model act on the resident fields
    (xmca_scale_field); the model equals the host one (array.py:317-365)."""
    from xmca_amd import _hip
    sst, prcp = make_input("sst_prcp")
    rng = np.random.default_rng(2)
    ref = MCA(sst, prcp)
    dev = MCA(sst, prcp, preprocess='device', **({"handle": _hip.Handle(0)} if own_handle else {}))
    wl = rng.uniform(0.5, 1.5, ref._fields['left'].shape[1]).astype(sst.dtype)
    wr = rng.uniform(0.5, 1.5, (1, ref._fields['right'].shape[1])).astype(prcp.dtype)
    for m in (ref, dev):
        m.normalize()
        m.apply_weights(left=wl, right=wr)
    assert dev._store_is_raw and dev._analysis['is_normalized']
    dev.solve()
    assert dev._store_is_raw                               # nothing came back to the host
    ref.solve()
    f32 = sst.dtype == np.float32
    assert _rel(dev._singular_values[:8], ref._singular_values[:8]) < (1e-4 if f32 else 1e-10)
    X, Xr = dev._get_X(), ref._get_X()
    for key in ref._keys:
        # (float32 input: the host's float32 column means carry ~3e-5 K of rounding at 300 K, the device's float64 ones do not)
        assert X[key].dtype == Xr[key].dtype and _rel(X[key], Xr[key]) < (1e-4 if f32 else 1e-12)
    full = MCA(sst, prcp, preprocess='device')
    full.apply_weights(left=np.ones_like(ref._fields['left']))          # (T, N') weights: the host path, silently
    assert not full._store_is_raw


def test_device_preprocessing_falls_back_and_lost_ownership():
    bad = np.random.default_rng(0).standard_normal((30, 12))
    bad[:, :] = np.where(np.arange(12) % 2 == 0, np.nan, bad)     # half the columns NaN: fine
    bad[5, :] = np.nan                                              # ... and one NaN time step: every column has a NaN
    with pytest.raises(ValueError):
        MCA(bad, preprocess='device')                               # the host path raises the reference's error
    fields = make_input("wide_both")
    a = MCA(*fields, preprocess='device')
    b = MCA(*make_input("small_both"))
    b.solve()                                              # takes the handle: a's resident fields are gone
    a.solve()
    r = MCA(*fields)
    r.solve()
    assert _rel(a._singular_values[:8], r._singular_values[:8]) < 1e-10
    with pytest.raises(ValueError):
        MCA(*fields, preprocess='gpu')


def test_device_preprocessing_downloads_the_centered_field_on_demand():
    """with its own handle the model keeps the resident fields: the first host access fetches the centered field from
    the device (and builds the analytic signal from it) instead of recomputing it."""
    from xmca_amd import _hip
    fields = make_input("wide_both")
    c = MCA(*fields, handle=_hip.Handle(0), preprocess='device')
    c.solve(complexify=True)
    assert c._store_is_raw and c._owns_device_fields(c._device())
    X = c._get_X()
    assert not c._store_is_raw
    r = MCA(*fields)
    r.solve(complexify=True)
    Xr = r._get_X()
    for key in r._keys:
        assert np.iscomplexobj(X[key]) and _rel(X[key], Xr[key]) < 1e-12
    c.solve()                                              # and a real solve afterwards
    r.solve()
    assert _rel(c._singular_values[:8], r._singular_values[:8]) < 1e-10


def test_singular_vectors_are_fetched_lazily_and_survive_other_models():
    """`_V` stays on the device after solve(); reading the leading modes fetches only those, and a second model solving
    on the same handle makes the first one fetch everything before its result is overwritten."""
    fields = make_input("wide_both")
    a = MCA(*fields)
    a.solve()
    assert set(a._V._pending) == {"left", "right"}
    head = a._V.head("left", 3)
    assert head.shape[1] == 3 and set(a._V._pending) == {"left", "right"}
    e3 = a.eofs(3)                                          # public getter of three modes: still nothing fetched in full
    assert set(a._V._pending) == {"left", "right"} and e3["left"].shape[-1] == 3
    b = MCA(*make_input("small_both"))
    b.solve()                                               # same default handle
    assert not a._V._pending                                # a holds its own copy now
    assert np.array_equal(a._V["left"][:, :3], head)
    r = MCA(*fields)
    r.solve()
    assert _rel(a._singular_values, r._singular_values) < 1e-12
    assert np.array_equal(a._V["right"], r._V["right"])    # the solver is deterministic
    a.solve(complexify=True)                                # re-solving drops the pending fetch instead of doing it
    assert np.iscomplexobj(a._V.head("left", 2))


# ----------------------------------------------------------------------------------------------
# regression tests of the round-1 review
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name,cplx,n_rot,power", [("unit_both", False, 8, 1), ("wide_both", True, 6, 4)])
def test_varimax_hand_over_from_the_persistent_kernel(monkeypatch, name, cplx, n_rot, power):
    """When the single-launch Varimax loop cannot finish (a workgroup of its grid never becomes resident: partitioned
    device, CUs held by another process) the per-iteration launches continue from the state it left.  Simulated by
    stopping the persistent launch after 7 iterations: same R, same stop iteration as the uninterrupted loop."""
    fields = make_input(name)
    m = MCA(*fields)
    m.solve(complexify=cplx)
    m.rotate(n_rot, power)
    n_iter, R, var = m._varimax_iterations, m._rotation_matrix.copy(), m._variance.copy()
    assert n_iter > 20
    monkeypatch.setenv("XMCA_VARIMAX_TEST_GIVEUP", "7")
    m.rotate(n_rot, power)
    assert m._varimax_iterations == n_iter
    assert _rel(m._rotation_matrix, R) < 1e-9 and _rel(m._variance, var) < 1e-9


@pytest.mark.parametrize("name,cplx,n_rot,power", [("unit_both", False, 8, 1), ("wide_both", True, 6, 4), ("unit_left", False, 5, 2),
                                                   ("unit_left", True, 18, 1)])
def test_rotate_on_resident_vectors_equals_rotate_on_fetched_loadings(name, cplx, n_rot, power):
    """rotate() right after solve() builds the stacked loadings V sqrt(s) (array.py:818-822) from the singular vectors that
    are still on the device (xmca_rotate_solved); once the vectors have been fetched it uploads the loadings built on the
    host (xmca_rotate_loadings).  Same rotation either way: iteration count, R, Phi, norms."""
    from xmca_amd.array import _LazyVectors
    fields = make_input(name)
    m = MCA(*fields)
    m.solve(complexify=cplx)
    assert isinstance(m._V, _LazyVectors) and m._V._pending == set(m._keys)       # nothing fetched yet: the device path
    m.rotate(n_rot, power)
    assert m._V._pending == set(m._keys)                                          # ... and rotate() fetched nothing
    got = (m._varimax_iterations, m.rotation_matrix().copy(), m.correlation_matrix().copy(), m.norm(), m.variance().copy())
    m._V.materialize()                                                            # now the host path
    m.rotate(n_rot, power)
    assert m._varimax_iterations == got[0]
    assert _rel(m.rotation_matrix(), got[1]) < 1e-10 and _rel(m.correlation_matrix(), got[2]) < 1e-10
    assert _rel(m.variance(), got[4]) < 1e-10
    for k in m._keys:
        assert _rel(m.norm()[k], got[3][k]) < 1e-10


def test_float32_model_keeps_float32_vectors_and_rotates_them_on_the_device():
    """A real float32 field decomposed on its dual side (the C5 route): `_V` has the input's dtype in the reference
    (array.py:584), so the back-projection writes float32 (half the store at C5) and rotate() right after solve() builds the
    float32 loadings float32(V) * float32(sqrt(s)) on the device - the product the reference's host code forms
    (array.py:818-822).  Same iteration count, R, norms as the host path on the fetched float32 vectors; and EVERY mode has
    unit norm (advisor, round 3: the 1 / sqrt(lambda) scale of the GEMM epilogue is only used where lambda is clear of the
    float32 noise of the Gram matrix, the weak modes are normalised by their measured norm)."""
    from xmca_amd.array import _LazyVectors
    rng = np.random.default_rng(12)
    T, N, k = 120, 900, 6
    X = ((rng.standard_normal((T, k)) * np.linspace(9, 2, k)) @ rng.standard_normal((k, N)) + 0.3 * rng.standard_normal((T, N))).astype(np.float32)
    X[:, 100:400] *= np.float32(1e-3)                       # a graded field: weak modes far below the leading ones
    m = MCA(X)
    m.solve()
    dev = m._device()
    assert dev.vectors_are_f32(0) and isinstance(m._V, _LazyVectors) and m._V._pending == set(m._keys)
    m.rotate(5, 1)
    assert m._V._pending == set(m._keys)                                          # rotate() fetched nothing
    got = (m._varimax_iterations, m.rotation_matrix().copy(), m.norm()['left'].copy(), m.variance().copy())
    m._V.materialize()                                                            # now the host path
    V = m._V['left']
    assert V.dtype == np.float32 and V.shape == (N, T)
    nrm = np.linalg.norm(V.astype(np.float64), axis=0)
    assert np.max(np.abs(nrm - 1.0)) < 2e-6, (np.argmax(np.abs(nrm - 1.0)), nrm.min(), nrm.max())      # all T modes, null mode included
    m.rotate(5, 1)
    assert m._varimax_iterations == got[0]
    assert _rel(m.rotation_matrix(), got[1]) < 1e-9 and _rel(m.norm()['left'], got[2]) < 1e-9 and _rel(m.variance(), got[3]) < 1e-9
    # ... and against the oracle (float32 model: 2e-5 on sigma, the reference's own 1e-3 on vectors)
    om = O.OracleModel(X)
    om.solve()
    lead = om.singular_values > 1e-2 * om.singular_values[0]
    assert np.max(np.abs(m._singular_values[lead] - om.singular_values[lead]) / om.singular_values[lead]) < 2e-5
    # weak modes of a float32 field: sgesdd itself is only good to eps32 sigma_1 there - an absolute bar
    assert np.max(np.abs(m._singular_values - om.singular_values)) < 2e-6 * om.singular_values[0]
    Va, _ = align_modes(V[:, :5].astype(np.float64), om.V[0][:, :5].astype(np.float64))
    assert np.max(np.abs(Va - om.V[0][:, :5])) < 1e-3


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_few_modes_on_a_long_grid_have_unit_norm(dtype):
    """T = 10 samples on 70 000 grid points: the null mode of the centered field (and every mode under the dtype's cut) is
    normalised by its measured norm, with one workgroup per CHUNK of a row when the rows are few and long (csrc/kernels.h
    normalize_rows; one workgroup per row took 2.5 ms on C5's 10^6-point rows).  All modes: unit norm; the non-null ones
    agree with the oracle (array.py:580-584)."""
    rng = np.random.default_rng(31)
    T, N = 10, 70000
    X = (rng.standard_normal((T, 3)) @ rng.standard_normal((3, N)) * 4 + rng.standard_normal((T, N))).astype(dtype)
    m = MCA(X)
    m.solve()
    V = np.asarray(m._V['left'])
    assert V.shape == (N, T) and V.dtype == dtype
    nrm = np.linalg.norm(V.astype(np.float64), axis=0)
    # (float32: the modes above the cut are scaled by 1 / sqrt(lambda), lambda from float32 products: ~1e-7 lambda_0 / lambda)
    assert np.max(np.abs(nrm - 1.0)) < (1e-5 if dtype == np.float32 else 1e-12), nrm
    assert abs(nrm[-1] - 1.0) < (2e-7 if dtype == np.float32 else 1e-12)             # the null mode: measured norm
    om = O.OracleModel(X)
    om.solve()
    k = T - 1
    Va, _ = align_modes(V[:, :k].astype(np.float64), om.V[0][:, :k].astype(np.float64))
    assert np.max(np.abs(Va - om.V[0][:, :k])) < (1e-3 if dtype == np.float32 else 1e-9)


def test_unconverged_eigensolver_raises_like_gesdd(monkeypatch):
    """the Jacobi sweeps either reach their stopping rule or the solve raises LinAlgError (numpy's 'SVD did not
    converge'): an unconverged basis is never returned as singular vectors."""
    rng = np.random.default_rng(8)
    m = MCA(rng.standard_normal((200, 520)))                # T = 200: four pair slots of 64 x 64 tiles
    monkeypatch.setenv("XMCA_JACOBI_MAX_SWEEPS", "2")
    with pytest.raises(np.linalg.LinAlgError):
        m.solve()
    monkeypatch.delenv("XMCA_JACOBI_MAX_SWEEPS")
    m.solve()
    assert m._analysis['rank'] == 200


def test_device_preprocessed_model_rescaled_after_a_complex_solve():
    """preprocess='device', solve(complexify=True) on narrow fields (N <= T: the general path leaves an imaginary plane
    on the device), then normalize() / apply_weights(): the reference simply rescales the fields and solves again."""
    from xmca_amd import _hip
    rng = np.random.default_rng(4)
    left, right = rng.standard_normal((60, 20)) * np.linspace(1, 3, 20), rng.standard_normal((60, 14)) * np.linspace(2, 1, 14)
    dev = MCA(left, right, handle=_hip.Handle(0), preprocess='device')
    ref = MCA(left, right, preprocess='host')
    for m in (dev, ref):
        m.solve(complexify=True)
        m.normalize()
        m.apply_weights(left=np.linspace(0.5, 1.5, 20))
        m.solve(complexify=True)
    assert dev._store_is_raw                                       # still never came back to the host
    assert _rel(dev._singular_values[:10], ref._singular_values[:10]) < 1e-10
    om = O.OracleModel(left, right)
    om.fields = [om.fields[0] / om.fields[0].std(axis=0) * np.linspace(0.5, 1.5, 20), om.fields[1] / om.fields[1].std(axis=0)]
    om.solve(complexify=True)
    assert _rel(dev._singular_values[:10], om.singular_values[:10]) < TOL


def test_resident_fields_follow_host_side_rescaling():
    """solve(), then normalize() on the host path: pcs() must project the UPDATED fields like the reference
    (`_get_U` reads `_fields`), not the copy the device still holds from the solve."""
    left, right = make_input("wide_both")
    m = MCA(left, right, preprocess='host')
    m.solve()
    before = m.pcs(3)
    m.normalize()
    after = m.pcs(3)
    X = m._get_X()
    for k in m._keys:
        ref = X[k] @ m._V[k][:, :3] / np.sqrt(m._singular_values[:3])
        assert _rel(after[k], ref) < 1e-9
        assert _rel(after[k], before[k]) > 1e-3                    # it did change


def test_model_identity_is_not_its_address():
    """a new model that happens to get the address of a collected one must not take over that one's resident fields"""
    import gc
    fields = make_input("wide_both")
    a = MCA(*fields)
    a.solve()
    dev = a._device()
    key = dev.fields_owner
    assert a._owns_device_fields(dev)
    del a
    gc.collect()
    for _ in range(50):
        b = MCA(*make_input("small_both"), preprocess='host')          # (nothing uploaded: it cannot own anything yet)
        assert not b._owns_device_fields(dev)
        assert dev.fields_owner is key
        del b


def test_device_memory_pool_is_reported_and_trimmed(hip):
    """The solver's temporaries stay with the handle for re-use (csrc/common.h DevPool); `pool_bytes` reports them,
    `trim_pool` returns them to the driver, and results do not depend on what the pool holds."""
    from golden_inputs import make_input
    from xmca_amd import _hip
    if os.environ.get("XMCA_POOL") == "0":
        pytest.skip("pool switched off")
    h = _hip.Handle(0)
    fields = make_input("wide_both")
    a = MCA(*fields, handle=h)
    a.solve(complexify=True)
    s1 = a._singular_values.copy()
    held = h.pool_bytes()
    assert held > 0
    h.trim_pool()
    assert h.pool_bytes() == 0
    b = MCA(*fields, handle=h)
    b.solve(complexify=True)
    assert np.array_equal(b._singular_values, s1)
    assert h.pool_bytes() > 0


@pytest.mark.parametrize("name,cplx,rot", [("sst_prcp", False, None), ("sst_prcp", True, (6, 2)), ("wide_both", False, (5, 1)),
                                           ("wide_both_f32", False, None), ("wide_both_f32", True, (4, 1)), ("c5_scaled", False, (10, 1))])
def test_eofs_assembled_on_the_device_equal_the_host_path(name, cplx, rot):
    """Round 6: while the vectors of solve() are still resident, `eofs()` is mixed on the device into its
    final (space..., modes) layout - `(V sqrt(s)) @ R / norm`, variance order, slice; array.py:615-646, 676-721 - by
    `xmca_get_eofs`; once they have been fetched the numpy path of `_get_V` runs.  Same numbers, shape, dtype and NaN mask
    (sst has masked grid points), for counts, slices, scalings and a phase shift."""
    fields = make_input(name)
    m = MCA(*fields)
    m.solve(complexify=cplx)
    if rot:
        m.rotate(*rot)
    asks = [dict(n=3), dict(n=None), dict(n=slice(2, 4)), dict(n=4, rotated=False), dict(n=3, scaling='max'), dict(n=3, scaling='eigen'),
            dict(n=2, scaling='std', phase_shift=0.7 if cplx else 0)]
    assert set(m._V._pending) == set(m._keys)
    fast = [m.eofs(**kw) for kw in asks]
    assert set(m._V._pending) == set(m._keys)               # nothing was fetched in full: every call took the device path
    m._V.materialize()
    for kw, f in zip(asks, fast):
        slow = m.eofs(**kw)                                 # vectors on the host now: the numpy path
        for k in m._keys:
            assert f[k].shape == slow[k].shape and f[k].dtype == slow[k].dtype, (kw, k, f[k].dtype, slow[k].dtype)
            assert np.array_equal(np.isnan(f[k]), np.isnan(slow[k]))
            ok = ~np.isnan(slow[k])
            # (float32 models: the host path mixes the vectors as fetched - rounded to float32 -, the device the resident ones)
            tol = 2e-6 if m._V._dtype == np.float32 else 1e-12
            assert np.max(np.abs(f[k][ok] - slow[k][ok])) <= tol * max(np.max(np.abs(slow[k][ok])), 1e-300), (kw, k)


def test_bootstrapping_and_rule_n_through_a_process_group_equal_the_plain_calls():
    """Round 6: `bootstrapping` shards its replicates over the ranks of a torch.distributed job like `rule_n` (dist.sharded_bootstrap:
    rank 0's composed row indices are broadcast, every rank runs its block on its GPU, ONE all_gather with a status row per rank).
    A one-rank gloo group in a subprocess - all a one-GPU box allows - takes exactly that path on the real device: the numbers are
    those of the plain calls (xmca/array.py:1935-1947, :1753-1769)."""
    import subprocess
    import sys
    REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, os, numpy as np; sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'tests'));"
            "from golden_inputs import make_input; from xmca_amd.array import MCA;"
            "f = make_input('wide_both'); m = MCA(*f); m.solve(complexify=True); m.rotate(4, 2);"
            "np.random.seed(5); b0 = m.bootstrapping(4, n_modes=3, on_left=True, on_right=True, block_size=2); r0 = m.rule_n(5, seed=9);"
            "import torch.distributed as td; os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = sys.argv[1];"
            "td.init_process_group('gloo', rank=0, world_size=1);"
            "np.random.seed(5); b1 = m.bootstrapping(4, n_modes=3, on_left=True, on_right=True, block_size=2); r1 = m.rule_n(5, seed=9);"
            "td.destroy_process_group();"
            "print('EQUAL', bool(np.array_equal(b0, b1)), bool(np.array_equal(r0, r1)), b0.shape, r0.shape)" % (REPO, REPO))
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    r = subprocess.run([sys.executable, "-c", code, str(port)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("EQUAL")][-1]
    assert line.split()[1] == "True" and line.split()[2] == "True", line
