"""`xmca_amd.xarray.xMCA` (the facade of xmca/xarray.py:23-1488) executed end to end.  The real `xarray` package is
not part of the image; tests/fake_xarray/ provides a minimal DataArray stand-in (values / dims / coords / name / attrs,
ufuncs, `*` with broadcasting by dimension name), used only when the real package cannot be imported.  Checked: the hot-
path pass-throughs (`solve` xarray.py:183-207, `rotate` :209-238, `rule_n` :1447-1488) give the numbers of the array
class, and every getter wraps them with the reference's dims / 1-based mode coords / names / attrs (xarray.py:286-297,
:455-467, :495-512, :1479-1488); `apply_coslat` (:167-181) equals explicit sqrt(cos(lat)) weights."""
import os
import sys

import numpy as np
import pytest

try:
    import xarray as xr                      # the real package, where it exists
except Exception:
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "fake_xarray"))
    import xarray as xr

from xmca_amd.array import MCA
from xmca_amd.xarray import xMCA

pytestmark = pytest.mark.gpu


def _fields():
    rng = np.random.default_rng(12)
    T, nlat, nlon = 72, 9, 14
    time = np.arange(T)
    lat = np.linspace(-60, 60, nlat)
    lon = np.linspace(0, 130, nlon)
    pcs = rng.standard_normal((T, 4)) * np.array([6.0, 4.0, 2.5, 1.5])
    pat = rng.standard_normal((4, nlat * nlon))
    a = (pcs @ pat + 0.4 * rng.standard_normal((T, nlat * nlon))).reshape(T, nlat, nlon)
    b = (pcs @ rng.standard_normal((4, nlat * (nlon - 3))) + 0.4 * rng.standard_normal((T, nlat * (nlon - 3)))).reshape(T, nlat, nlon - 3)
    a[:, 2, 3] = np.nan                                        # a masked grid point
    mk = lambda v, lo: xr.DataArray(v, dims=['time', 'lat', 'lon'], coords={'time': time, 'lat': lat, 'lon': lo})
    return mk(a, lon), mk(b, lon[:-3])


def test_constructor_validation():
    left, right = _fields()
    xMCA()
    xMCA(left)
    with pytest.raises(ValueError):
        xMCA(left, right, right)
    with pytest.raises(TypeError):
        xMCA(left.values)


@pytest.mark.parametrize("cplx", [False, True])
def test_facade_wraps_the_array_class(cplx):
    left, right = _fields()
    xm = xMCA(left, right)
    xm.set_field_names('sst', 'prcp')
    xm.apply_coslat()
    xm.solve(complexify=cplx)
    xm.rotate(4, 2)

    m = MCA(left.values, right.values)
    for k, f in zip(['left', 'right'], [left, right]):
        w = np.sqrt(np.cos(np.deg2rad(f.coords['lat'].values)) + 1e-6)[None, :, None] * np.ones(f.shape)
        m.apply_weights(**{k: w.reshape(f.shape[0], -1)[:, m._no_nan_index[k]]})
    m.solve(complexify=cplx)
    m.rotate(4, 2)
    assert xm._analysis['is_coslat_corrected'] and xm._varimax_iterations == m._varimax_iterations

    sv = xm.singular_values(6)
    assert sv.dims == ('mode',) and list(sv.coords['mode'].values) == [1, 2, 3, 4, 5, 6] and sv.name == 'singular values'
    assert np.allclose(sv.values, m.singular_values(6), rtol=1e-12)
    assert sv.attrs['is_rotated'] == 'True' and sv.attrs['n_rot'] == '4' and sv.attrs['power'] == '2'
    ev = xm.explained_variance(4)
    assert ev.name == 'covariance fraction' and np.allclose(ev.values, m.explained_variance(4), rtol=1e-10)

    pcs, ref_pcs = xm.pcs(4), m.pcs(4)
    eofs, ref_eofs = xm.eofs(4), m.eofs(4)
    for k, name, f in zip(['left', 'right'], ['sst', 'prcp'], [left, right]):
        assert pcs[k].dims == ('time', 'mode') and pcs[k].name == name + ' pcs'
        assert np.array_equal(pcs[k].coords['time'].values, f.coords['time'].values)
        assert np.allclose(pcs[k].values, ref_pcs[k], rtol=1e-9, atol=1e-12)
        assert eofs[k].dims == ('lat', 'lon', 'mode') and eofs[k].name == name + ' eofs'
        assert eofs[k].shape == f.shape[1:] + (4,)
        assert np.array_equal(eofs[k].coords['lon'].values, f.coords['lon'].values)
        assert np.allclose(eofs[k].values, ref_eofs[k], rtol=1e-9, atol=1e-12, equal_nan=True)
    assert np.isnan(eofs['left'].values[2, 3]).all()                 # the masked grid point comes back as NaN
    amp = xm.spatial_amplitude(2)
    assert amp['left'].name == 'sst spatial amplitude' and amp['left'].shape == left.shape[1:] + (2,)
    hom, pv = xm.homogeneous_patterns(3)
    assert hom['right'].name == 'prcp homogeneous patterns' and pv['right'].shape == right.shape[1:] + (3,)

    back = xm.fields(original_scale=True)
    assert back['left'].dims == ('time', 'lat', 'lon')
    # (apply_coslat weights with sqrt(cos(lat) + 1e-6), the inverse scaling divides by sqrt(cos(lat)): xarray.py:167-181 vs
    #  :110-126 - the reference's own round-trip test uses rtol = 1e-3, test_integration_xarray.py:284-341)
    assert np.allclose(back['left'].values.real, left.values, rtol=1e-4, atol=1e-4, equal_nan=True)

    runs = xm.rule_n(5, seed=3)
    ref_runs = m.rule_n(5, seed=3)
    assert runs.dims == ('mode', 'run') and runs.name == 'singular values'
    assert list(runs.coords['run'].values) == [1, 2, 3, 4, 5][:runs.shape[1]] and list(runs.coords['mode'].values) == [1, 2, 3, 4]
    assert np.array_equal(runs.values, ref_runs)
    north = xm.rule_north(3)
    assert north.dims == ('mode',) and np.allclose(north.values, m.rule_north(3))
    np.random.seed(2)
    boot = xm.bootstrapping(3, n_modes=3, on_left=True, on_right=True, block_size=2)
    np.random.seed(2)
    assert boot.dims == ('mode', 'run') and np.allclose(boot.values, m.bootstrapping(3, n_modes=3, on_left=True, on_right=True, block_size=2))

    first = xr.DataArray(left.values[:10], dims=left.dims, coords={'time': left.coords['time'].values[:10],
                                                                   'lat': left.coords['lat'].values, 'lon': left.coords['lon'].values})
    new = xm.predict(first, n=3)
    assert new['left'].dims == ('time', 'mode') and new['left'].shape == (10, 3)
    if not cplx:          # (a complex model's PCs belong to the analytic signal of the whole series, not to 10 real time steps)
        assert np.allclose(new['left'].values, pcs['left'].values[:10, :3], rtol=1e-6, atol=1e-8)
