"""`xmca_amd.xarray.xMCA` - drop-in for `xmca.xarray.xMCA` (xmca/xarray.py:23-1488).

A thin facade over `xmca_amd.array.MCA`: the hot path (`solve`, `rotate`, `rule_n`) is inherited unchanged (the
reference's versions are pure pass-throughs, xarray.py:183-238, :1447-1488); every getter re-wraps the numpy result
as an `xarray.DataArray` with the reference's dims / coords / names (1-based `mode`, `time`, `lat`, `lon`, `run`) and
`attrs = str(self._analysis)` items.  xarray (and h5netcdf for IO, cartopy for maps) are imported lazily: they are
not part of the build image, so this module is only exercised where they exist (`pytest.importorskip('xarray')`).
"""
import os

import numpy as np

from .array import MCA, secure_str


def _xr():
    try:
        import xarray as xr
    except Exception as err:        # pragma: no cover - depends on the environment
        raise ImportError("xmca_amd.xarray needs the `xarray` package") from err
    return xr


class xMCA(MCA):
    """MCA / EOF analysis of one or two `xarray.DataArray` (time, lat, lon)."""

    def __init__(self, *fields, handle=None, preprocess=None):
        xr = _xr()
        if len(fields) > 2:
            raise ValueError("Too many fields. Pass 1 or 2 fields.")
        if not all(isinstance(f, xr.DataArray) for f in fields):
            raise TypeError('''One or more fields are not `xarray.DataArray`.
            Please provide `xarray.DataArray` only.''')
        self._field_dims = {}
        self._field_coords = {}
        for key, field in zip(['left', 'right'], fields):
            self._field_dims[key] = field.dims
            self._field_coords[key] = field.coords
        super().__init__(*[f.values for f in fields], handle=handle, preprocess=preprocess)

    # ------------------------------------------------------------------ scaling incl. coslat weights
    def _coslat_weights(self, k):
        coslat = np.sqrt(np.cos(np.deg2rad(self._field_coords[k]['lat']))).values
        weights = np.ones(self._fields_spatial_shape[k]) * coslat.reshape(coslat.size, 1)
        return weights.flatten()[self._no_nan_index[k]]

    def _scale_X(self, data_dict):
        k = None
        for k in data_dict:
            data_dict[k] -= self._field_means[k]
        # normalisation / coslat outside the loop: only the last field, as in the reference (xarray.py:97-108)
        if k is not None and self._analysis['is_normalized']:
            data_dict[k] /= self._field_stds[k]
        if k is not None and self._analysis['is_coslat_corrected']:
            data_dict[k] *= self._coslat_weights(k)
        return data_dict

    def _scale_X_inverse(self, data_dict):
        for k, field in data_dict.items():
            if self._analysis['is_coslat_corrected']:
                field /= self._coslat_weights(k)
            if self._analysis['is_normalized']:
                field *= self._field_stds[k]
            field += self._field_means[k]
        return data_dict

    def apply_weights(self, **weights):
        """Weights as DataArrays broadcastable against the fields (keys `left` / `right`)."""
        fields = self.fields()
        store = self._fields
        for k, weight in weights.items():
            try:
                new = (fields[k] * weight).data
            except KeyError as err:
                raise KeyError('Key `{:}` not found. Please use `left` or `right`'.format(k)) from err
            try:
                new = new.reshape(self._n_observations[k], self._n_variables[k])[:, self._no_nan_index[k]]
            except ValueError as err:
                msg = ('Error for {:} weights. Mismatch between dimensions of weights ({:}) and original field ({:}).')
                raise ValueError(msg.format(k, weight.shape, fields[k].shape)) from err
            store[k] = new
        self._fields = store              # (through the setter: the copy a device may still hold is stale now)

    def apply_coslat(self):
        """Weight by sqrt(cos(lat)) (area weighting on a regular grid)."""
        eps = 1e-6
        self.apply_weights(**{k: np.sqrt(np.cos(np.deg2rad(c['lat'])) + eps) for k, c in self._field_coords.items()})
        self._analysis['is_coslat_corrected'] = True

    # ------------------------------------------------------------------ wrapping helpers
    def _attrs(self):
        return {k: str(v) for k, v in self._analysis.items()}

    def _modes(self, n, length):
        sl = self._get_slice(n)
        return list(range(sl.start + 1, sl.stop + 1))[:length]

    def _mode_array(self, values, n, name):
        xr = _xr()
        return xr.DataArray(values, dims=['mode'], coords={'mode': self._modes(n, len(values))}, name=name,
                            attrs=self._attrs())

    def _time_array(self, key, values, n, what):
        xr = _xr()
        return xr.DataArray(values, dims=['time', 'mode'],
                            coords={'time': self._field_coords[key]['time'], 'mode': self._modes(n, values.shape[1])},
                            name=' '.join([self._field_names[key], what]), attrs=self._attrs())

    def _space_array(self, key, values, n, what):
        xr = _xr()
        c = self._field_coords[key]
        return xr.DataArray(values, dims=['lat', 'lon', 'mode'],
                            coords={'lon': c['lon'], 'lat': c['lat'], 'mode': self._modes(n, values.shape[-1])},
                            name=' '.join([self._field_names[key], what]), attrs=self._attrs())

    # ------------------------------------------------------------------ getters
    def fields(self, original_scale=False):
        xr = _xr()
        out = super().fields(original_scale)
        return {k: xr.DataArray(out[k], dims=self._field_dims[k], coords=self._field_coords[k], name=self._field_names[k])
                for k in self._keys}

    def singular_values(self, n=None):
        return self._mode_array(super().singular_values(n), n, 'singular values')

    def norm(self, n=None, sorted=True):
        return {k: self._mode_array(v, n, ' '.join([self._field_names[k], 'norm']))
                for k, v in super().norm(n=n, sorted=sorted).items()}

    def variance(self, n=None, sorted=True):
        return self._mode_array(super().variance(n, sorted), n, 'variance')

    def explained_variance(self, n=None):
        return self._mode_array(super().explained_variance(n), n, 'covariance fraction')

    def scf(self, n=None):
        return self._mode_array(super().scf(n), n, 'squared covariance fraction')

    def pcs(self, n=None, scaling='None', phase_shift=0, rotated=True):
        return {k: self._time_array(k, v, n, 'pcs') for k, v in super().pcs(n, scaling, phase_shift, rotated).items()}

    def eofs(self, n=None, scaling='None', phase_shift=0, rotated=True):
        return {k: self._space_array(k, v, n, 'eofs') for k, v in super().eofs(n, scaling, phase_shift, rotated).items()}

    def spatial_amplitude(self, n=None, scaling='None', rotated=True):
        return {k: self._space_array(k, v, n, 'spatial amplitude')
                for k, v in MCA.spatial_amplitude(self._plain(), n, scaling, rotated).items()}

    def spatial_phase(self, n=None, phase_shift=0, rotated=True):
        return {k: self._space_array(k, v, n, 'spatial phase')
                for k, v in MCA.spatial_phase(self._plain(), n, phase_shift, rotated).items()}

    def temporal_amplitude(self, n=None, scaling='None', rotated=True):
        return {k: self._time_array(k, v, n, 'temporal amplitude')
                for k, v in MCA.temporal_amplitude(self._plain(), n, scaling, rotated).items()}

    def temporal_phase(self, n=None, phase_shift=0, rotated=True):
        return {k: self._time_array(k, v, n, 'temporal phase')
                for k, v in MCA.temporal_phase(self._plain(), n, phase_shift, rotated).items()}

    def _plain(self):
        """View of this object whose getters return numpy arrays (the base-class implementations call
        self.eofs()/self.pcs(), which are overridden here)."""
        return _NumpyView(self)

    def homogeneous_patterns(self, n=None, phase_shift=0):
        r, p = super().homogeneous_patterns(n, phase_shift)
        return ({k: self._space_array(k, v, n, 'homogeneous patterns') for k, v in r.items()},
                {k: self._space_array(k, v, n, 'pvalues homogeneous patterns') for k, v in p.items()})

    def heterogeneous_patterns(self, n=None, phase_shift=0):
        r, p = super().heterogeneous_patterns(n, phase_shift)
        return ({k: self._space_array(k, v, n, 'heterogeneous patterns') for k, v in r.items()},
                {k: self._space_array(k, v, n, 'pvalues heterogeneous patterns') for k, v in p.items()})

    def reconstructed_fields(self, mode=None, original_scale=True):
        xr = _xr()
        out = super().reconstructed_fields(mode, original_scale)
        return {k: xr.DataArray(v, dims=self._field_dims[k], coords=self._field_coords[k],
                                name='reconstructed_{:}_field'.format(k)) for k, v in out.items()}

    def predict(self, left=None, right=None, n=None, scaling='None', phase_shift=0):
        xr = _xr()
        data = {k: d for k, d in zip(['left', 'right'], [left, right]) if d is not None}
        if not all(isinstance(d, xr.DataArray) for d in data.values()):
            raise TypeError('Data must be `xarray.DataArray`.')
        out = super().predict(left=None if left is None else left.values, right=None if right is None else right.values,
                              n=n, scaling=scaling, phase_shift=phase_shift)
        res = {}
        for k, v in out.items():
            res[k] = xr.DataArray(v, dims=['time', 'mode'],
                                  coords={'time': data[k].coords['time'], 'mode': list(range(1, v.shape[1] + 1))})
        return res

    # ------------------------------------------------------------------ significance
    def rule_north(self, n=None):
        return self._mode_array(super().rule_north(n), n, 'singular values')

    def _run_array(self, values):
        xr = _xr()
        return xr.DataArray(values, dims=['mode', 'run'],
                            coords={'mode': list(range(1, values.shape[0] + 1)), 'run': list(range(1, values.shape[1] + 1))},
                            name='singular values')

    def rule_n(self, n_runs, n_modes=None, **kwargs):
        return self._run_array(super().rule_n(n_runs, n_modes, **kwargs))

    def bootstrapping(self, n_runs, n_modes=20, axis=0, on_left=True, on_right=False, block_size=1, replace=True,
                      strategy='standard', disable_progress=False):
        # like the reference the facade always resamples along time (xarray.py:1418-1419)
        return self._run_array(_NumpyView(self).bootstrapping_base(n_runs, n_modes, 0, on_left, on_right, block_size,
                                                                    replace, strategy, disable_progress))

    # ------------------------------------------------------------------ persistence (netCDF through xarray)
    def _save_data(self, data, path, engine='h5netcdf', *args, **kwargs):
        out = os.path.join(path, secure_str('.'.join([data.name, 'nc'])))
        data.to_netcdf(path=out, engine=engine, invalid_netcdf=(engine == 'h5netcdf'), *args, **kwargs)

    def plot(self, *args, **kwargs):
        """see `xmca_amd.array.MCA.plot`"""
        return MCA.plot(self, *args, **kwargs)

    def save_plot(self, *args, **kwargs):
        return MCA.plot(self, *args, **kwargs)

    def save_analysis(self, path=None, engine='h5netcdf'):
        """info.xmca + original-scale real fields, UNROTATED eofs and singular values (xarray.py:1253-1279)."""
        path = self._get_analysis_path(path)
        self._create_analysis_path(path)
        self._create_info_file(path)
        fields = self.fields(original_scale=True)
        eofs = self.eofs(rotated=False)
        self._save_data(self.singular_values(), path, engine)
        for key in self._keys:
            self._save_data(eofs[key], path, engine)
            self._save_data(fields[key].real, path, engine)

    def load_analysis(self, path, engine='h5netcdf'):
        xr = _xr()
        self._set_info_from_file(path)
        folder, _ = os.path.split(path)
        names = self._get_file_names(format='nc')
        singular_values = xr.open_dataarray(os.path.join(folder, names['singular']), engine=engine).data
        fields, eofs = {}, {}
        self._field_coords = {}
        for key in self._field_names.keys():
            eofs[key] = xr.open_dataarray(os.path.join(folder, names['eofs'][key]), engine=engine).data
            f = xr.open_dataarray(os.path.join(folder, names['fields'][key]), engine=engine)
            self._field_coords[key] = f.coords
            self._field_dims[key] = f.dims
            fields[key] = f.data
        MCA.load_analysis(self, path=path, fields=fields, eofs=eofs, singular_values=singular_values)
        if self._analysis['is_coslat_corrected']:
            self.apply_coslat()


class _NumpyView:
    """Delegates attribute access to an xMCA but resolves the public getters to the numpy base-class versions."""

    _BASE = ('eofs', 'pcs', 'spatial_amplitude', 'spatial_phase', 'temporal_amplitude', 'temporal_phase', 'explained_variance',
             'singular_values', 'norm', 'variance', 'fields')

    def __init__(self, obj):
        object.__setattr__(self, '_obj', obj)

    def __getattr__(self, name):
        obj = object.__getattribute__(self, '_obj')
        if name in _NumpyView._BASE:
            return lambda *a, **k: getattr(MCA, name)(self, *a, **k)
        if name == 'bootstrapping_base':
            return lambda *a, **k: MCA.bootstrapping(self, *a, **k)
        return getattr(obj, name)

    def __setattr__(self, name, value):
        setattr(object.__getattribute__(self, '_obj'), name, value)
