"""CPU: the C-ABI library loads and exports every symbol of include/xmca_hip.h; host logic of the drop-in class;
the product path fails loudly without a GPU (no fallback)."""
import os
import re

import numpy as np
import pytest

from golden_inputs import make_input
from oracle import ref_numpy as O

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(REPO, "include", "xmca_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(xmca_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound():
    from xmca_amd import _hip
    lib = _hip.load_library()
    names = _declared_symbols()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), "symbol %s declared in include/xmca_hip.h is not exported" % n
    assert set(names) == set(_hip.SIGNATURES), set(names) ^ set(_hip.SIGNATURES)
    assert b"gfx950" in lib.xmca_version()


def test_library_is_built_for_gfx950_only():
    from xmca_amd import _hip
    blob = open(_hip.library_path(), "rb").read()
    targets = set(re.findall(rb"amdgcn-amd-amdhsa--(gfx[0-9a-f]+)", blob))       # code objects of the offload bundle
    assert targets == {b"gfx950"}, targets
    assert not re.search(rb"nvptx|sm_[0-9]{2}\b", blob)                           # no CUDA device code rides along


def test_stale_library_is_refused(monkeypatch):
    """a library built from another revision of include/xmca_hip.h (ABI number) is not bound silently"""
    from xmca_amd import _hip
    _hip.load_library()
    header = open(os.path.join(REPO, "include", "xmca_hip.h")).read()
    assert int(re.search(r"#define XMCA_ABI_VERSION (\d+)", header).group(1)) == _hip.ABI_VERSION
    monkeypatch.setattr(_hip, "_lib", None)
    monkeypatch.setattr(_hip, "ABI_VERSION", _hip.ABI_VERSION + 1)
    with pytest.raises(ImportError, match="ABI"):
        _hip.load_library()


def test_no_gpu_means_loud_failure():
    """without a device the product raises instead of falling back to numpy (checked only where no GPU exists)."""
    from xmca_amd import _hip
    if _hip.load_library().xmca_device_count() > 0:
        pytest.skip("a GPU is visible here")
    from xmca_amd.array import MCA
    m = MCA(np.random.default_rng(0).standard_normal((30, 8)))
    with pytest.raises(RuntimeError):
        m.solve()
    from xmca_amd.tools.rotation import varimax
    with pytest.raises(RuntimeError):
        varimax(np.random.default_rng(0).standard_normal((30, 4)))


def test_product_does_not_import_the_oracle():
    for root, _, files in os.walk(os.path.join(REPO, "xmca_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                assert "oracle" not in src.replace("# oracle", ""), f


# ------------------------------------------------------------------------------------------------
# host logic of MCA (no device call)
# ------------------------------------------------------------------------------------------------
def test_constructor_matches_oracle_preprocessing():
    from xmca_amd.array import MCA
    sst, prcp = make_input("sst_prcp")
    m = MCA(sst, prcp)
    for k, f in zip(["left", "right"], [sst, prcp]):
        c, valid, mean, std = O.flatten_and_center(f)
        assert np.array_equal(m._no_nan_index[k], valid)
        assert np.array_equal(m._fields[k], c) and m._fields[k].dtype == f.dtype
        assert np.array_equal(m._field_means[k], mean) and np.array_equal(m._field_stds[k], std)
    assert m._n_variables == {'left': 162, 'right': 162} and m._fields['left'].shape == (492, 155)
    assert m._analysis['is_bivariate'] and m._analysis['method'] == 'mca'
    assert MCA(sst)._analysis['method'] == 'pca'


def test_slices_like_reference():
    from xmca_amd.array import MCA
    m = MCA()
    m._analysis['rank'] = 50
    assert m._get_slice(None) == slice(0, 50) and m._get_slice(7) == slice(0, 7)
    assert m._get_slice(slice(3, 9)) == slice(2, 9, None)
    assert m._get_slice(slice(None, 80)) == slice(0, 50, None)
    with pytest.raises(ValueError):
        m._get_slice("3")


@pytest.mark.parametrize("cplx,rot", [(False, None), (False, (6, 4)), (True, (6, 2))])
def test_getters_on_injected_state(cplx, rot):
    """feed the oracle's solve/rotate state into the class: getters must reproduce the reference formulas."""
    from xmca_amd.array import MCA
    fields = make_input("wide_both")
    om = O.OracleModel(*fields)
    so = om.solve(complexify=cplx)
    m = MCA(*fields)
    if cplx:
        m._fields = {k: f for k, f in zip(m._keys, om.fields)}
    m._analysis['is_complex'] = cplx
    sv = so["singular_values"]
    m._V = {k: v for k, v in zip(m._keys, so["V"])}
    m._singular_values, m._variance = sv, sv
    m._var_idx = np.argsort(sv)[::-1]
    m._norm = {k: np.sqrt(sv) for k in m._keys}
    m._analysis.update({'rank': len(sv), 'n_rot': len(sv), 'total_covariance': sv.sum(),
                        'total_squared_covariance': (sv ** 2).sum()})
    m._rotation_matrix = m._correlation_matrix = np.eye(len(sv))
    if rot:
        ro = om.rotate(*rot)
        m._norm = {k: n for k, n in zip(m._keys, ro["norm"])}
        m._variance, m._var_idx = ro["variance"], ro["var_idx"]
        m._rotation_matrix, m._correlation_matrix = ro["R"], ro["Phi"]
        m._analysis.update({'is_rotated': True, 'n_rot': rot[0], 'power': rot[1]})
    n = rot[0] if rot else 5
    # the product X @ V of _get_U runs on the device (tests/test_gpu_mca.py); here only the host algebra around it
    m._project_on_device = lambda V: {k: f @ V[k] for k, f in zip(m._keys, om.fields)}
    eofs, pcs = m.eofs(n), m.pcs(n)
    X = om.fields
    for i, k in enumerate(m._keys):
        V = so["V"][i][:, :len(sv) if not rot else rot[0]]
        if rot:
            s = sv[:rot[0]]
            Vr = (V * np.sqrt(s) @ ro["R"] / ro["norm"][i])[:, ro["var_idx"]]
            Rinv = np.linalg.pinv(ro["R"]).conj().T if rot[1] > 1 else ro["R"]
            U = ((X[i] @ V / np.sqrt(s)) @ Rinv)[:, ro["var_idx"]]
        else:
            Vr, U = V[:, :n], X[i] @ V[:, :n] / np.sqrt(sv[:n])
        assert np.allclose(eofs[k].reshape(-1, n), Vr[:, :n]) and np.allclose(pcs[k], U[:, :n])
    assert np.allclose(m.explained_variance().sum(), m._get_variance().sum() / sv.sum() * 100)
    assert np.allclose(m.scf(3), m._variance[m._var_idx][:3] ** 2 / (sv ** 2).sum() * 100)
    amp = m.spatial_amplitude(n)['left']
    assert np.allclose(amp, np.abs(eofs['left']))
    assert m.rule_north(4).shape == (4,)


def test_tools():
    from xmca_amd.tools.array import block_bootstrap, has_nan_time_steps, pearsonr, remove_nan_cols
    x = np.arange(24.).reshape(12, 2)
    np.random.seed(0)
    b = block_bootstrap(x, block_size=3)
    assert b.shape == x.shape and set(b[::3, 0]) <= set(x[::3, 0])
    np.random.seed(0)
    p = block_bootstrap(x, axis=1, replace=False)
    assert sorted(p[0]) == sorted(x[0])
    with pytest.raises(ValueError):
        block_bootstrap(x, block_size=5)
    with pytest.raises(ValueError):
        block_bootstrap(x, axis=2)
    y = x.copy()
    y[3, 1] = np.nan
    assert remove_nan_cols(y).shape == (12, 1) and not has_nan_time_steps(y)
    y[3] = np.nan
    assert has_nan_time_steps(y)
    rng = np.random.default_rng(1)
    a, c = rng.standard_normal((50, 3)), rng.standard_normal((50, 2))
    r, pv = pearsonr(a, c)
    assert r.shape == (3, 2) and np.allclose(r[1, 0], np.corrcoef(a[:, 1], c[:, 0])[0, 1]) and np.all((pv >= 0) & (pv <= 1))


@pytest.mark.parametrize("T", [9, 10, 64, 101])
def test_hilbert_operator_equals_scipy(T):
    from scipy.signal import hilbert
    from xmca_amd._hip import hilbert_imag_column, hilbert_imag_operator
    x = np.random.default_rng(T).standard_normal((T, 6))
    H = hilbert_imag_operator(T)
    assert np.allclose(H @ x, hilbert(x, axis=0).imag, atol=1e-13)
    assert np.allclose(H[:, 0], hilbert_imag_column(T))


def test_shard_ranges_cover_all_runs():
    from xmca_amd.dist import shard_range
    for n in [0, 1, 7, 200]:
        for w in [1, 2, 3, 8]:
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            assert max(e - b for b, e in spans) - min(e - b for b, e in spans) <= 1


def test_lazy_vectors_and_raw_field_stand_in():
    """host logic of the device-resident state: `_LazyVectors` fetches leading modes / everything on demand and exactly
    once; `_RawField` replays the device's preprocessing (NaN columns, centering, weights) on the host."""
    from xmca_amd.array import _LazyVectors, _RawField

    class FakeDev:
        def __init__(self):
            self.calls = []
            self.Vt = {0: np.arange(5 * 7, dtype=np.float64).reshape(5, 7), 1: -np.arange(5 * 4, dtype=np.float64).reshape(5, 4)}

        def vectors(self, side, n_modes, N, dtype):
            self.calls.append((side, n_modes))
            return self.Vt[side][:n_modes].astype(dtype)

    dev = FakeDev()
    V = _LazyVectors(dev, {"left": (0, 7), "right": (1, 4)}, 5, np.float64)
    assert list(V) == ["left", "right"] and dev.calls == []
    assert np.array_equal(V.head("left", 2), dev.Vt[0][:2].T) and dev.calls == [(0, 2)] and V._pending == {"left", "right"}
    assert np.array_equal(V["left"], dev.Vt[0].T) and V._pending == {"right"}
    assert np.array_equal(V.head("left", 3), dev.Vt[0][:3].T) and dev.calls == [(0, 2), (0, 5)]      # served from the host copy
    assert np.array_equal(V.head("right", 9), dev.Vt[1].T) and not V._pending                       # more than rank: everything
    assert [k for k, _ in V.items()] == ["left", "right"] and len(dev.calls) == 3
    V2 = _LazyVectors(dev, {"left": (0, 7)}, 5, np.float64)
    V2["left"] = np.zeros((7, 2))                                                                   # truncate(): assignment wins
    assert not V2._pending and V2["left"].shape == (7, 2)

    rng = np.random.default_rng(0)
    raw = rng.standard_normal((20, 6))
    raw[3, 1] = np.nan
    raw[:, 4] = np.nan
    keep = ~np.isnan(raw).any(axis=0)
    f = _RawField(raw, keep)
    assert f.shape == (20, 4) and f.dtype == raw.dtype and f.real is f and not np.iscomplexobj(f)
    w = np.array([1.0, 2.0, 3.0, 4.0])
    f.ops.append((True, np.nanstd(raw[:, keep], axis=0)))
    f.ops.append((False, w))
    want = (raw[:, keep] - raw[:, keep].mean(axis=0)) / raw[:, keep].std(axis=0) * w
    assert np.allclose(f.centered(), want, rtol=1e-14, atol=1e-14)


def test_plot_methods_exist_and_say_what_to_use_instead():
    """the reference exposes plot / save_plot on both classes (xmca/array.py:1430, xarray.py): here they exist and raise a
    NotImplementedError that names the getters to plot from - not an AttributeError."""
    import pytest
    from xmca_amd.array import MCA
    m = MCA()
    for fn in (m.plot, m.save_plot):
        with pytest.raises(NotImplementedError, match="eofs"):
            fn(mode=1)
