"""`xmca_amd.array.MCA` - drop-in for `xmca.array.MCA` with solve()/rotate()/rule_n() on the MI355X.

Host side (this file, numpy): validation, flattening, NaN masking, centering, weights, getters, IO - the same
public methods, argument meanings, state attributes and exceptions as xmca/array.py (v1.4.2), so the
reference's tests read the same against this class.
Device side (libxmca_hip.so through `_hip.Handle`): the numerical core of
  * `solve`   xmca/array.py:549-584   (per-field SVD, kernel, kernel SVD, back-projection)
  * `rotate`  xmca/array.py:821-823 + xmca/tools/rotation.py (Varimax/Promax loop)
  * `rule_n`  xmca/array.py:1753-1765 (surrogate loop; run-sharded over ranks when torch.distributed is up)
There is no numpy fallback for these three: without the library or a GPU they raise.
"""
import cmath
import os
import warnings
from datetime import datetime

import numpy as np

from . import __version__, _hip
from .tools.array import (block_bootstrap, get_nan_cols, has_nan_time_steps, pearsonr, remove_mean, remove_nan_cols)

_SCALINGS_MSG = ('The scaling option {:} is not valid. Please choose one of the following: None, eigen, std, max')


def _two_sided_p(r, n_obs):
    """Two-sided p-value of a Pearson correlation under the exact null distribution, as tools/array.py:86-88:
    `2 * scipy.stats.beta(n/2 - 1, n/2 - 1, loc=-1, scale=2).cdf(-abs(r))` = 2 I_{(1-|r|)/2}(n/2 - 1, n/2 - 1): the same regularised
    incomplete beta function (bit for bit) without the frozen distribution's argument handling.  The function itself is what
    `homogeneous_patterns` costs on the host (18 of 21.7 ms at C2 for 10^5 values; it holds the GIL, so threads do not help)."""
    import scipy.special
    a = n_obs / 2 - 1
    return 2 * scipy.special.betainc(a, a, (1.0 - np.abs(np.asarray(r, dtype=np.float64))) / 2)


class _LazyVectors(dict):
    """`MCA._V` after solve(): (N', rank) singular vectors per field, fetched from the device when first read.  `head`
    fetches only the leading modes while the full array has not been asked for."""

    def __init__(self, dev, where, rank, dtype):
        super().__init__({k: None for k in where})
        self._dev, self._where, self._rank, self._dtype = dev, dict(where), rank, dtype
        self._pending = set(where)

    def _load(self, k):
        if k in self._pending:
            side, n_k = self._where[k]
            dict.__setitem__(self, k, self._dev.vectors(side, self._rank, n_k, self._dtype).T)   # view of the mode-major result
            self._pending.discard(k)

    def materialize(self):
        for k in list(self._pending):
            self._load(k)

    def head(self, k, m):
        if k in self._pending:
            side, n_k = self._where[k]
            m = self._rank if m is None else min(m, self._rank)
            if m < self._rank:
                return self._dev.vectors(side, m, n_k, self._dtype).T
            self._load(k)
        return dict.__getitem__(self, k)[:, :m]

    def __getitem__(self, k):
        self._load(k)
        return dict.__getitem__(self, k)

    def __setitem__(self, k, v):
        self._pending.discard(k)
        dict.__setitem__(self, k, v)

    def get(self, k, default=None):
        return self[k] if k in self else default

    def items(self):
        self.materialize()
        return dict.items(self)

    def values(self):
        self.materialize()
        return dict.values(self)


class _RawField:
    """Stand-in for a field that was preprocessed on the device (MCA(..., preprocess='device')): the raw input, the
    mask of its NaN-free columns, and the shape / dtype the centered field has."""

    def __init__(self, raw, keep):
        self.raw = raw
        self.keep = keep
        self.shape = (raw.shape[0], int(np.count_nonzero(keep)))
        self.dtype = raw.dtype
        self.ops = []            # (divide, per-column factors) applied on the device after centering, in order

    @property
    def real(self):
        return self

    def kept_columns(self):
        return self.raw if self.shape[1] == self.raw.shape[1] else self.raw[:, self.keep]

    def centered(self):
        """What the device holds, recomputed on the host."""
        f = np.ascontiguousarray(remove_mean(self.kept_columns()))
        for divide, w in self.ops:
            f = f / w if divide else f * w
        return f


class MCA:
    """Maximum Covariance Analysis of one (EOF/PCA) or two `numpy.ndarray` fields; time is axis 0."""

    def __init__(self, *fields, handle=None, preprocess=None):
        """fields: one or two numpy arrays, time first.  `handle`: a `_hip.Handle` (default: one per device).
        `preprocess` (extension): where the constructor's NaN-column / mean / std / centering passes (array.py:191-215)
        run.  `'device'`: on the GPU over the uploaded raw field - column means in float64 - and the centered field stays
        resident for solve(); the host copy `_fields` is fetched on first use.  `'host'`: the reference's numpy path
        (bit-identical means for float32 input).  Default (None): `'device'` when a GPU is visible and the fields are
        plain real float32 / float64 arrays of one dtype, `'host'` otherwise (XMCA_PREPROCESS=host|device overrides)."""
        if preprocess is None:
            preprocess = os.environ.get('XMCA_PREPROCESS') or 'auto'
        if preprocess not in ('host', 'device', 'auto'):
            raise ValueError("preprocess must be 'host' or 'device'")
        if preprocess == 'auto':
            preprocess = 'device' if (handle is not None or _gpu_visible()) else 'host'
        if len(fields) > 2:
            raise ValueError("Too many fields. Pass 1 or 2 fields.")
        if len(fields) == 2 and fields[0].shape[0] != fields[1].shape[0]:
            raise ValueError('''Time dimensions of given fields are different.
                Time series should have same time lengths.''')
        if not all(isinstance(f, np.ndarray) for f in fields):
            raise TypeError('''One or more fields are not `numpy.ndarray`.
            Please provide `numpy.ndarray` only.''')
        self._handle_override = handle
        self._preprocess = preprocess
        self._store_is_raw = False
        self._keys = ['left', 'right']
        if len(fields) == 1:
            self._keys.pop()
        self._fields_store = {}
        self._pending_hilbert = False
        self._device_hilbert = False
        self._upload_serial = 0
        self._token = object()              # identity of this model as owner of a handle's resident fields (never reused, unlike id())
        self._shape = {}
        self._field_names = {}
        self._field_means = {}
        self._field_stds = {}
        self._fields_spatial_shape = {}
        self._n_variables = {}
        self._no_nan_index = {}
        self._n_observations = {}

        data = {k: f for k, f in zip(self._keys, fields)}
        if not (preprocess == 'device' and self._ingest_on_device(data)):
            if any(has_nan_time_steps(f) for f in fields):
                raise ValueError('''One or more fields contain NaN time steps.
            Please remove these prior to analysis.''')
            self._ingest(data)

        self._analysis = {
            'version': __version__,
            'is_bivariate': len(self._fields_store) > 1,
            'is_normalized': False,
            'is_coslat_corrected': False,
            'method': 'pca',
            'is_complex': False,
            'extend': False,
            'theta_period': 365,
            'is_rotated': False,
            'n_rot': 0,
            'power': 0,
            'is_truncated': False,
            'is_truncated_at': 0,
            'rank': 0,
            'total_covariance': 0.0,
            'total_squared_covariance': 0.0,
        }
        self._analysis['method'] = self._get_method_id()

    # ------------------------------------------------------------------------------------------
    # `_fields`: the reference replaces it by the analytic signal inside solve(complexify=True)
    # (array.py:546-547).  The device only needs the real field, so the host copy of the analytic signal is
    # materialised lazily, the first time anything reads `_fields`.
    # ------------------------------------------------------------------------------------------
    @property
    def _fields(self):
        if self._store_is_raw:
            self._materialize_fields()
        if self._pending_hilbert:
            from scipy.signal import hilbert
            self._fields_store = {k: hilbert(f.real, axis=0) for k, f in self._fields_store.items()}
            self._pending_hilbert = False
        return self._fields_store

    @_fields.setter
    def _fields(self, value):
        self._fields_store = value
        self._pending_hilbert = False
        self._store_is_raw = False
        self._upload_serial = getattr(self, '_upload_serial', 0) + 1    # whatever the device still holds is stale now

    def _owner_key(self):
        if not hasattr(self, '_token'):          # models built without __init__ (bench / load flows)
            self._token = object()
        return (self._token, self._upload_serial)

    def _owns_device_fields(self, dev):
        return getattr(dev, 'fields_owner', None) == self._owner_key()

    def _ingest_on_device(self, data):
        """preprocess='device': upload the raw fields, drop their NaN columns and center them there, keep them resident.
        False (nothing changed) when the fields are not plain real float32/float64 arrays of one dtype, or when a
        field has no NaN-free column (the host path then raises the reference's errors)."""
        if len(data) == 0:
            return False
        dtypes = {np.dtype(f.dtype) for f in data.values()}
        if len(dtypes) != 1 or next(iter(dtypes)) not in (np.dtype(np.float32), np.dtype(np.float64)):
            return False
        dev = self._device()
        flat = {k: np.ascontiguousarray(f.reshape(f.shape[0], int(np.prod(f.shape[1:])))) for k, f in data.items()}
        stats, keep = {}, {}
        for side, k in enumerate(self._keys):
            dev.set_field(side, flat[k])
            keep[k], n_keep = dev.compact_field(side, flat[k].shape[1])      # array.py:191-197 on the device
            if n_keep == 0:
                dev.fields_owner = None
                return False
            stats[k] = dev.center_field(side, n_keep)
        self._set_field_meta(data)
        store = {}
        for k, f in flat.items():
            self._no_nan_index[k] = keep[k]
            self._field_means[k] = stats[k][0].astype(f.dtype, copy=False)
            self._field_stds[k] = stats[k][1].astype(f.dtype, copy=False)
            store[k] = _RawField(f, keep[k])
        self._fields_store = store              # stand-ins: only shape / dtype are read while `_store_is_raw`
        self._store_is_raw = True
        dev.fields_owner = self._owner_key()
        return True

    def _materialize_fields(self):
        """Host copy of the centered fields of a device-preprocessed model: downloaded while the device still holds them,
        otherwise recomputed from the raw input."""
        dev = self._device()
        if self._owns_device_fields(dev):
            store = {k: dev.get_field(side, self._fields_store[k].shape, self._fields_store[k].dtype)
                     for side, k in enumerate(self._keys)}
        else:
            store = {k: f.centered() for k, f in self._fields_store.items()}
        self._fields_store = store
        self._store_is_raw = False

    def _device(self):
        return self._handle_override or _hip.default_handle()

    # ------------------------------------------------------------------------------------------
    # constructor helpers (array.py:191-240)
    # ------------------------------------------------------------------------------------------
    def _ingest(self, data):
        """meta, reshape to 2-D, NaN mask, mean/std, centering - in the reference's order (array.py:110-117)."""
        self._set_field_meta(data)
        data = self._reshape_to_2d(data)
        self._set_no_nan_idx(data)
        data = self._remove_nan_cols(data)
        self._set_field_means(data)
        self._set_field_stds(data)
        self._fields = self._center(data)

    def _set_field_meta(self, data):
        for k, field in data.items():
            self._shape[k] = field.shape
            self._n_observations[k] = field.shape[0]
            self._fields_spatial_shape[k] = field.shape[1:]
            self._n_variables[k] = int(np.prod(field.shape[1:]))
            self._field_names[k] = k

    def _reshape_to_2d(self, data):
        return {k: f.reshape(f.shape[0], int(np.prod(f.shape[1:]))) for k, f in data.items()}

    def _set_no_nan_idx(self, data):
        for k, f in data.items():
            self._no_nan_index[k] = ~get_nan_cols(f)

    def _remove_nan_cols(self, data):
        return {k: remove_nan_cols(f) for k, f in data.items()}

    def _set_field_means(self, data):
        for k, f in data.items():
            self._field_means[k] = f.mean(axis=0)

    def _set_field_stds(self, data):
        for k, f in data.items():
            self._field_stds[k] = f.std(axis=0)

    def _center(self, data):
        # time-major contiguous rows for the device upload (boolean column selection leaves a column-major layout; the
        # means above were taken on it, exactly like the reference, so the values are bit-identical)
        return {k: np.ascontiguousarray(remove_mean(f)) for k, f in data.items()}

    def _get_method_id(self):
        return 'mca' if self._analysis['is_bivariate'] else 'pca'

    def _get_slice(self, input):
        """int n -> slice(0, n); slice (1-based, inclusive stop) -> 0-based slice.  array.py:145-173"""
        if input is None or np.issubdtype(type(input), np.integer):
            return slice(0, self._analysis['rank'] if input is None else input)
        if isinstance(input, slice):
            start = 0 if input.start is None else max(0, input.start - 1)
            stop = self._analysis['rank'] if input.stop is None else min(input.stop, self._analysis['rank'])
            return slice(start, stop, input.step)
        raise ValueError('Invalid type {:}. Must be either int or slice.'.format(type(input)))

    def set_field_names(self, left='left', right='right'):
        """Names used in plots and saved files."""
        self._field_names['left'] = left
        self._field_names['right'] = right

    # ------------------------------------------------------------------------------------------
    # scaling helpers (array.py:264-315)
    # ------------------------------------------------------------------------------------------
    def _scale_X(self, data_dict):
        scaled = data_dict.copy()
        field = None
        k = None
        for k, field in scaled.items():
            field -= self._field_means[k]
        # as in the reference the normalisation sits outside the loop: only the LAST field is divided (array.py:269-272)
        if self._analysis['is_normalized'] and field is not None:
            field /= self._field_stds[k]
        return scaled

    def _scale_X_inverse(self, data_dict):
        for k, field in data_dict.items():
            if self._analysis['is_normalized']:
                field *= self._field_stds[k]
            field += self._field_means[k]
        return data_dict

    def _get_X(self, original_scale=False, real=False):
        X = {k: f.copy() for k, f in self._fields.items()}
        if real:
            X = {k: x.real for k, x in X.items()}
        if original_scale:
            X = self._scale_X_inverse(X)
        return X

    def _with_nan_columns(self, key, values, lead_shape):
        """re-insert the masked grid points: values (..., N') -> (..., N) filled with NaN."""
        out = np.zeros(lead_shape + (self._n_variables[key],), dtype=values.dtype) * np.nan
        out[..., self._no_nan_index[key]] = values
        return out

    def _get_fields(self, original_scale=False):
        n_obs = self._n_observations['left']
        fields = {}
        for k, X in self._get_X(original_scale=original_scale).items():
            full = self._with_nan_columns(k, X, (n_obs,))
            fields[k] = full.reshape((n_obs,) + self._fields_spatial_shape[k])
        return fields

    # ------------------------------------------------------------------------------------------
    # pre-processing
    # ------------------------------------------------------------------------------------------
    def apply_weights(self, left=None, right=None):
        """Multiply the (centered) fields by weights broadcastable to (T, N').  array.py:317-349"""
        weights = {'left': 1 if left is None else left, 'right': 1 if right is None else right}
        if self._scale_on_device({k: weights[k] for k in self._keys}, divide=False):
            return
        self._fields = {k: f * weights[k] for k, f in self._fields.items()}

    def normalize(self):
        """Divide every grid point's series by its standard deviation.  array.py:351-365"""
        if not self._scale_on_device({k: self._field_stds[k] for k in self._keys}, divide=True):
            fields = self._fields
            self._fields = {k: fields[k] / self._field_stds[k] for k in self._keys}
        self._analysis['is_normalized'] = True
        self._analysis['is_coslat_corrected'] = False
        self._analysis['method'] = self._get_method_id()

    def _scale_on_device(self, factors, divide):
        """Device-preprocessed model whose fields are still resident: per-column factors (scalars, (N',) or (1, N')
        arrays that do not change the dtype) are applied there.  False when the host path has to do it."""
        if not self._store_is_raw:
            return False
        dev = self._device()
        if not self._owns_device_fields(dev):
            return False
        cols = {}
        for k, w in factors.items():
            f = self._fields_store[k]
            w = np.asarray(w)
            if w.ndim > 2 or (w.ndim == 2 and w.shape[0] != 1) or np.iscomplexobj(w):
                return False
            orig = factors[k]
            weak = isinstance(orig, (int, float)) and not isinstance(orig, np.generic)     # python scalars do not promote
            if np.result_type(f.dtype, orig if weak else w.dtype) != f.dtype:
                return False
            try:
                cols[k] = np.ascontiguousarray(np.broadcast_to(w.reshape(-1) if w.ndim else w, (f.shape[1],)), dtype=f.dtype)
            except ValueError:
                return False
        # a complexified solve on the general path leaves an imaginary plane next to the resident real one; the real
        # plane is untouched and the next solve re-complexifies (the reference simply rescales `_fields`)
        dev.decomplexify()
        for side, k in enumerate(self._keys):
            dev.scale_field(side, cols[k], divide)
            self._fields_store[k].ops.append((divide, cols[k]))
        return True

    # ------------------------------------------------------------------------------------------
    # complexification on the host (only needed for extend != False, and lazily for the getters)
    # ------------------------------------------------------------------------------------------
    def _theta_forecast(self, series):
        try:
            from statsmodels.tsa.forecasting.theta import ThetaModel
        except Exception as err:          # statsmodels is an optional dependency
            raise ImportError("extend='theta' needs statsmodels") from err
        steps = len(series)
        model = ThetaModel(series, period=self._analysis['theta_period'], deseasonalize=True, use_test=False).fit()
        return model.forecast(steps=steps, theta=20)

    def _get_reg_coefs(self, x, y):
        assert x.shape[0] == y.shape[0]
        n = x.shape[0]
        xmean, ymean = np.mean(x, axis=0), np.mean(y, axis=0)
        xstd = np.mean(x, axis=0)      # sic: the reference uses the mean here (array.py:384); kept for parity
        cov = np.sum((x - xmean) * (y - ymean), axis=0) / n
        slope = cov / (xstd ** 2)
        return ymean - xmean * slope, slope

    def _exp_forecast(self, field):
        n = field.shape[0]
        x = np.repeat(np.arange(n)[:, np.newaxis], field.shape[1], axis=1)
        intercept, slope = self._get_reg_coefs(x, field)
        linear_end = slope * x[-1, :] + intercept
        offset = field[-1, :] - linear_end
        theta = self._analysis['theta_period']
        return offset * np.exp(-(x + 1) / theta) + (slope * x) + linear_end

    def _extend(self, field):
        extend = self._analysis['extend']
        if extend == 'theta':
            return np.array([self._theta_forecast(col) for col in field.T]).T
        if extend == 'exp':
            return self._exp_forecast(field)
        raise ValueError('''{:} is not a valid extension. Choose either
            `exp` or `theta`.'''.format(extend))

    def _complexify(self, fields):
        """Analytic signal along time (array.py:429-472), with optional fore/back-cast extension."""
        from scipy.signal import hilbert
        n_obs = self._n_observations['left']
        out = {}
        for k in self._keys:
            f = fields[k].real
            if self._analysis['extend']:
                post = self._extend(f)
                pre = self._extend(f[::-1])[::-1]
                f = np.concatenate([pre, f, post])
            f = hilbert(f, axis=0)
            if self._analysis['extend']:
                f = remove_mean(f[n_obs:(2 * n_obs)])
            out[k] = f
        return out

    # ------------------------------------------------------------------------------------------
    # solve (array.py:509-603) - numerical core on the device
    # ------------------------------------------------------------------------------------------
    def solve(self, complexify=False, extend=False, period=1):
        """EOF analysis / MCA: singular value decomposition of the (cross-)covariance matrix.

        complexify : Hilbert-transform the fields first (complex EOF/MCA).
        extend     : False, 'exp' or 'theta' - fore/back-cast before the Hilbert transform.
        period     : season length (theta) / e-folding time (exp).
        """
        store = self._fields_store
        if len(store) == 0 or (not self._store_is_raw and any(np.isnan(f).all() for f in store.values())):
            raise RuntimeError('''
            Fields are empty. Did you forget to load data?
            ''')
        self._analysis['is_complex'] = complexify
        self._analysis['extend'] = extend
        self._analysis['theta_period'] = period
        if isinstance(getattr(self, '_V', None), _LazyVectors):
            del self._V                                              # vectors of an earlier solve still on the device: not wanted

        dev = self._device()
        if complexify and extend:
            self._fields = self._complexify(self._fields)           # host path (nonlinear extension)
            self._device_hilbert = False
        elif self._store_is_raw:
            self._device_hilbert = bool(complexify)                  # centered real fields are resident already
            self._pending_hilbert = bool(complexify)
        else:
            real = {k: self._fields[k].real if np.iscomplexobj(self._fields_store[k]) else self._fields_store[k]
                    for k in self._keys}
            self._device_hilbert = bool(complexify)                  # X_im = Ht X on the device
            if complexify:
                self._fields_store = real
                self._pending_hilbert = True                         # host copy of the analytic signal: on first use
            else:
                self._fields = real
        self._upload_fields(dev)

        try:
            rank = dev.solve(len(self._keys))
        except np.linalg.LinAlgError as err:                       # array.py:575-578 (the device message is the cause)
            raise np.linalg.LinAlgError('''SVD failed. NaN entries may be the problem.''') from err

        real_dtype = _real_dtype(next(iter(self._fields_store.values())).dtype)
        singular_values = dev.singular_values(rank).astype(real_dtype, copy=False)
        # the vectors stay on the device until something reads them (233 MB at C2; pcs / eofs / rotate of a few modes
        # fetch just those modes); anything that would invalidate them on the handle makes this model fetch them first
        self._V = _LazyVectors(dev, {k: (side, self._fields_store[k].shape[1]) for side, k in enumerate(self._keys)}, rank, real_dtype)
        dev.hold_result(self)

        self._singular_values = singular_values
        self._variance = singular_values
        self._var_idx = np.argsort(singular_values)[::-1]
        self._norm = {k: np.sqrt(singular_values) for k in self._keys}
        self._analysis['total_covariance'] = singular_values.sum()
        self._analysis['total_squared_covariance'] = (singular_values ** 2).sum()
        self._analysis['rank'] = len(singular_values)
        self._analysis['is_rotated'] = False
        self._analysis['n_rot'] = len(singular_values)
        self._analysis['power'] = 0
        # unrotated: both are the unit matrix (array.py:594-595) - built by rotation_matrix() / correlation_matrix() when
        # somebody asks (two rank x rank arrays are 400 MB at rank 5000, and rotate() would pay for freeing them)
        for name in ('_rotation_matrix', '_correlation_matrix'):
            if hasattr(self, name):
                delattr(self, name)
        self._analysis['is_truncated_at'] = len(singular_values)

    # ------------------------------------------------------------------------------------------
    # state accessors used by the getters (array.py:605-779)
    # ------------------------------------------------------------------------------------------
    def _get_svals(self, n=None):
        try:
            return self._singular_values[self._get_slice(n)]
        except AttributeError:
            raise RuntimeError('Cannot retrieve singular values. Please call the method `solve` first.')

    def _get_min_mode(self, n=None, rotated=False):
        cand = [self._analysis['rank']]
        if n is not None:
            cand.append(n)
        if rotated:
            cand.append(self._analysis['n_rot'])
        return np.min(cand)

    def _get_max_mode(self, n=None, rotated=False):
        cand = [self._analysis['rank'] if n is None else n]
        if rotated:
            cand.append(self._analysis['n_rot'])
        return np.max(cand)

    def _max_mode(self, n, rotated):
        if rotated:
            return self._analysis['n_rot']
        return n.stop if isinstance(n, slice) else n

    def _materialize_vectors(self):
        """Called by the handle before the device result this model still reads from is overwritten."""
        V = getattr(self, '_V', None)
        if isinstance(V, _LazyVectors):
            V.materialize()

    def _get_V(self, n=None, rotated=True):
        # an unrotated model: the reference multiplies all `rank` modes by sqrt(s) I / sqrt(s) and reorders them by the
        # (already descending) singular values - the identity; only the requested modes are touched here
        rotated = rotated and self._analysis['is_rotated']
        max_mode = self._max_mode(n, rotated)
        keep = self._get_slice(n)
        try:
            V = {k: (self._V.head(k, max_mode) if isinstance(self._V, _LazyVectors) else self._V[k][:, :max_mode]) for k in self._V}
        except AttributeError:
            raise RuntimeError('Cannot retrieve singular vectors. Please call the method `solve` first.')
        for k in self._keys:
            if rotated:
                sqrt_svals = np.sqrt(self._get_svals(max_mode))
                norm = self._get_norm(max_mode, sorted=False)
                V[k] = (V[k] * sqrt_svals @ self.rotation_matrix() / norm[k])[:, self._var_idx]
            V[k] = V[k][:, keep]
        return V

    def _upload_fields(self, dev):
        """Makes the fields solve() works on resident on the device and records this model as their owner."""
        if self._store_is_raw:
            if self._owns_device_fields(dev):                   # device-preprocessed and still resident: nothing to send
                if self._device_hilbert:
                    dev.complexify(self._n_observations['left'])
                else:
                    dev.decomplexify()
                return
            self._materialize_fields()                          # the handle was used elsewhere: recompute, then upload
        store = self._fields_store
        for side, k in enumerate(self._keys):
            dev.set_field(side, _device_ready(store[k]))
        if self._device_hilbert and not any(np.iscomplexobj(f) for f in store.values()):
            dev.complexify(self._n_observations['left'])        # (a materialised host analytic signal goes up as it is)
        dev.fields_owner = self._owner_key()

    def _project_on_device(self, V):
        """fields[k] @ V[k] of `_get_U` (array.py:391) as a device GEMM over the resident fields."""
        dev = self._device()
        if not self._owns_device_fields(dev):
            self._upload_serial += 1          # another model / rule_n used the handle in between: upload again
            self._upload_fields(dev)
        T = self._n_observations['left']
        if next(iter(V.values())).shape[1] == 0:          # pcs(0): nothing to project (bootstrapping's first iterative step)
            return {k: np.zeros((T, 0), dtype=V[k].dtype) for k in self._keys}
        return {k: dev.project(side, V[k], T) for side, k in enumerate(self._keys)}

    def _get_U(self, n=None, rotated=True):
        keep = self._get_slice(n)
        mix = rotated and self._analysis['is_rotated']
        if mix:
            max_mode = self._max_mode(n, rotated)
        else:
            # the reference multiplies by the identity rotation matrix over all `rank` modes here (array.py:393); only
            # the kept modes are projected instead - same numbers, no T x N x rank product for pcs(10), and the
            # null modes (sigma = 0 exactly on the device) cannot leak 0 * inf into the kept columns
            max_mode = keep.stop if keep.stop is not None else self._analysis['rank']
        V = self._get_V(max_mode, rotated=False)
        sqrt_svals = np.sqrt(self._get_svals(max_mode))
        XV = self._project_on_device(V)
        U = {}
        for k in self._keys:
            U[k] = XV[k].astype(np.result_type(V[k].dtype, self._fields_store[k].dtype), copy=False) / sqrt_svals
            if mix:
                R = self.rotation_matrix(inverse_transpose=True)
                U[k] = (U[k] @ R)[:, self._var_idx]
            U[k] = U[k][:, keep]
        return U

    def _get_norm(self, n=None, sorted=True):
        try:
            norm = self._norm
        except AttributeError:
            raise RuntimeError('Cannot retrieve field norms. Please call the method `solve` first.')
        if sorted:
            norm = {k: v[self._var_idx] for k, v in norm.items()}
        modes = self._get_slice(n)
        return {k: v[modes] for k, v in norm.items()}

    def _get_variance(self, n=None, sorted=True):
        norm = self._get_norm(n=n, sorted=sorted)
        if self._analysis['is_bivariate']:
            return norm['left'] * norm['right']
        return norm['left'] ** 2

    def _eofs_from_device(self, n, rotated):
        """(N' x q) array per field in its final memory layout, mixed on the device from the vectors still resident there
        (`xmca_get_eofs`), or None when they are not (then `_get_V`'s host path is taken).  Same numbers as `_get_V`:
        `(V sqrt(s)) @ R / norm`, columns ordered by explained variance, then the requested slice (array.py:615-646)."""
        Vl = getattr(self, '_V', None)
        if not (isinstance(Vl, _LazyVectors) and Vl._pending == set(self._keys)):
            return None                                   # (vectors already on the host - or injected by a test: no device needed)
        dev = self._device()
        if not dev.holds_result_of(self):
            return None
        rotated = rotated and self._analysis['is_rotated']
        max_mode = self._max_mode(n, rotated)
        max_mode = self._analysis['rank'] if max_mode is None else min(max_mode, self._analysis['rank'])
        keep = self._get_slice(n)
        if max_mode < 1 or len(range(max_mode)[keep]) < 1:
            return None
        out = {}
        for side, k in enumerate(self._keys):
            n_k = Vl._where[k][1]
            if rotated:
                sqrt_svals = np.sqrt(self._get_svals(max_mode))
                norm = self._get_norm(max_mode, sorted=False)
                W = ((sqrt_svals[:, None] * self.rotation_matrix()) / norm[k])[:, self._var_idx][:, keep]
                out[k] = dev.eofs(side, n_k, max_mode, W, np.float64)
            else:
                cols = range(max_mode)[keep]
                if cols.start == 0 and cols.step == 1:
                    out[k] = dev.eofs(side, n_k, len(cols), None, Vl._dtype)
                else:
                    W = np.eye(max_mode)[:, keep]
                    out[k] = dev.eofs(side, n_k, max_mode, W, Vl._dtype)
        return out

    def _get_eofs(self, n=None, scaling='None', phase_shift=0, rotated=True):
        V = self._eofs_from_device(n, rotated)
        if V is None:
            V = self._get_V(n, rotated=rotated)
        eofs = {}
        for k in self._keys:
            n_modes = V[k].shape[1]
            if self._n_variables[k] == V[k].shape[0] and V[k].flags['C_CONTIGUOUS']:
                full = V[k]                                                   # no masked points: the array is final as it is
            else:
                full = np.full((self._n_variables[k], n_modes), np.nan, dtype=V[k].dtype)
                full[self._no_nan_index[k]] = V[k]                            # (N, n_modes), NaN at masked points
            eofs[k] = full.reshape(self._fields_spatial_shape[k] + (n_modes,))
            if self._analysis['is_complex'] and phase_shift != 0:
                eofs[k] = eofs[k] * cmath.rect(1, phase_shift)
            space_axes = tuple(range(eofs[k].ndim - 1))
            if scaling == 'None':
                pass
            elif scaling == 'eigen':
                eofs[k] = eofs[k] * self._get_norm(V[self._keys[0]].shape[1], sorted=True)[k]
            elif scaling == 'max':
                eofs[k] = eofs[k] / np.nanmax(abs(eofs[k].real), axis=space_axes)
            elif scaling == 'std':
                eofs[k] = eofs[k] / np.nanstd(eofs[k].real, axis=space_axes)
            else:
                raise ValueError(_SCALINGS_MSG.format(scaling))
        return eofs

    def _get_pcs(self, n=None, scaling='None', phase_shift=0, rotated=True):
        U = self._get_U(n, rotated=rotated)
        for k in self._keys:
            if self._analysis['is_complex']:
                U[k] = U[k] * cmath.rect(1, phase_shift)
            if scaling == 'None':
                pass
            elif scaling == 'eigen':
                U[k] = U[k] * self._get_norm(n, sorted=True)[k]
            elif scaling == 'max':
                U[k] = U[k] / np.nanmax(abs(U[k].real), axis=0)
            elif scaling == 'std':
                U[k] = U[k] / np.nanstd(U[k].real, axis=0)
            else:
                raise ValueError(_SCALINGS_MSG.format(scaling))
        return U

    # ------------------------------------------------------------------------------------------
    # rotate (array.py:781-844) - Varimax/Promax loop on the device
    # ------------------------------------------------------------------------------------------
    def rotate(self, n_rot, power=1, tol=1e-8):
        """Promax rotation of the first `n_rot` modes (`power=1`: Varimax).

        Raises ValueError for `n_rot < 2` / `power < 1`, RuntimeError when Varimax does not converge
        within 1000 iterations.
        """
        if n_rot < 2:
            raise ValueError('`n_rot` must be > 1')
        if power < 1:
            raise ValueError('`power` must be >=1')
        dev = self._device()
        V = getattr(self, '_V', None)
        if (isinstance(V, _LazyVectors) and V._pending == set(self._keys) and dev.holds_result_of(self)
                and n_rot <= self._analysis['rank'] and (V._dtype == np.float64 or dev.vectors_are_f32(0))):
            # the vectors of solve() are still resident: the stacked loadings V sqrt(s) are built on the device.  float32
            # models: the reference rotates float32 loadings (float32 vectors x float32 sqrt(s)) - the device does the same
            # product when the vectors are resident in float32 (one real field, dual side); any other float32 model
            # takes the host path below
            out = dev.rotate_solved(n_rot, power=power, tol=tol, max_iter=1000)
        else:
            sqrt_svals = np.sqrt(self._get_svals(n_rot))
            V = self._get_V(n_rot, rotated=False)
            n_vars_left = V['left'].shape[0]
            # loadings of both fields stacked (Cheng and Dunkerton 1995)
            L = np.concatenate(list(V.values())) * sqrt_svals
            out = dev.rotate_loadings(L, n_left=n_vars_left, power=power, tol=tol, max_iter=1000)
        self._varimax_iterations = out['n_iter']

        norm = {'left': out['norm_left'], 'right': out['norm_right']}
        if not self._analysis['is_bivariate']:
            norm['right'] = norm['left']
        variance = norm['left'] * norm['right']
        self._norm = norm
        self._variance = variance
        self._var_idx = np.argsort(variance)[::-1]
        self._rotation_matrix = out['R']
        self._correlation_matrix = out['Phi']
        self._analysis['is_rotated'] = True
        self._analysis['n_rot'] = n_rot
        self._analysis['power'] = power

    def rotation_matrix(self, inverse_transpose=False):
        """Rotation matrix (unit matrix when not rotated); `inverse_transpose` matters for Promax only."""
        try:
            R = self._rotation_matrix
        except AttributeError:
            R = np.eye(len(self.singular_values()))
        if inverse_transpose and self._analysis['power'] > 1:
            R = np.linalg.pinv(R).conjugate().T
        return R

    def correlation_matrix(self):
        """Correlation matrix of the (rotated) PCs, ordered by variance."""
        try:
            idx = self._var_idx
            return self._correlation_matrix[idx, :][:, idx]
        except AttributeError:
            return np.eye(len(self.singular_values()))

    # ------------------------------------------------------------------------------------------
    # public getters (array.py:898-1297)
    # ------------------------------------------------------------------------------------------
    def fields(self, original_scale=False):
        """The (centered / normalised / complexified) input fields, optionally back in original units."""
        return self._get_fields(original_scale)

    def singular_values(self, n=None):
        return self._get_svals(n)

    def norm(self, n=None, sorted=True):
        return self._get_norm(n=n, sorted=sorted)

    def variance(self, n=None, sorted=True):
        return self._get_variance(n=n, sorted=sorted)

    def scf(self, n=None):
        """Squared covariance fraction in percent."""
        variance = self._variance[self._var_idx][:n]
        return variance ** 2 / self._analysis['total_squared_covariance'] * 100

    def explained_variance(self, n=None):
        """Covariance fraction in percent."""
        return self._get_variance(n=n, sorted=True) / self._analysis['total_covariance'] * 100

    def pcs(self, n=None, scaling='None', phase_shift=0, rotated=True):
        return self._get_pcs(n, scaling, phase_shift, rotated)

    def eofs(self, n=None, scaling='None', phase_shift=0, rotated=True):
        return self._get_eofs(n, scaling, phase_shift, rotated)

    def spatial_amplitude(self, n=None, scaling='None', rotated=True):
        out = {}
        for k, eof in self.eofs(n, scaling='None', rotated=rotated).items():
            out[k] = np.sqrt(eof * eof.conjugate()).real
            if scaling == 'max':
                out[k] /= np.nanmax(out[k], axis=tuple(range(out[k].ndim - 1)))
        return out

    def spatial_phase(self, n=None, phase_shift=0, rotated=True):
        return {k: np.arctan2(e.imag, e.real).real
                for k, e in self.eofs(n, phase_shift=phase_shift, rotated=rotated).items()}

    def temporal_amplitude(self, n=None, scaling='None', rotated=True):
        out = {}
        for k, pc in self.pcs(n, scaling='None', rotated=rotated).items():
            out[k] = np.sqrt(pc * pc.conjugate()).real
            if scaling == 'max':
                out[k] /= np.nanmax(out[k], axis=0)
        return out

    def temporal_phase(self, n=None, phase_shift=0, rotated=True):
        return {k: np.arctan2(p.imag, p.real).real
                for k, p in self.pcs(n, phase_shift=phase_shift, rotated=rotated).items()}

    def _correlation_maps(self, n, phase_shift, pair):
        """Pearson correlation of every grid point (real part of the field) with the PCs and its p-value
        (array.py:1188-1261, tools/array.py:76-88).  The correlations are one tall GEMM over the field resident on the
        device (`xmca_correlate`) instead of the reference's (N + m)^2 `np.corrcoef` matrix; p-values on the host."""
        import scipy.stats
        pcs = self._get_pcs(n=n, phase_shift=phase_shift)
        dev = self._device()
        if not self._owns_device_fields(dev):
            self._upload_serial += 1
            self._upload_fields(dev)
        n_obs = self._n_observations['left']
        rvals, pvals = {}, {}
        for side, k in enumerate(self._keys):
            try:
                y = pcs[pair[k]].real
            except KeyError:
                raise KeyError('Key not found. Two fields needed for heterogenous maps.')
            r = dev.correlate(side, y, self._fields_store[k].shape[1])
            r = r.astype(np.result_type(self._fields_store[k].real.dtype, y.dtype), copy=False)
            p = _two_sided_p(r, n_obs)
            for src, dst in ((r, rvals), (p, pvals)):
                full = self._with_nan_columns(k, src.T, (src.shape[1],)).T
                dst[k] = full.reshape(self._fields_spatial_shape[k] + (src.shape[1],))
        return rvals, pvals

    def homogeneous_patterns(self, n=None, phase_shift=0):
        return self._correlation_maps(n, phase_shift, {k: k for k in self._keys})

    def heterogeneous_patterns(self, n=None, phase_shift=0):
        other = dict(zip(['left', 'right'], ['right', 'left']))
        return self._correlation_maps(n, phase_shift, other)

    def _reconstructed_X(self, mode=None, original_scale=True):
        V = self._get_V(n=mode, rotated=True)
        U = self._get_pcs(n=mode, scaling='eigen', rotated=True)
        Xrec = {k: (U[k] @ V[k].conj().T).real for k in self._keys}
        if original_scale:
            Xrec = self._scale_X_inverse(Xrec)
        return Xrec

    def reconstructed_fields(self, mode=None, original_scale=True):
        n_obs = self._n_observations['left']
        out = {}
        for k, X in self._reconstructed_X(mode=mode, original_scale=original_scale).items():
            full = self._with_nan_columns(k, np.asarray(X, dtype=float), (n_obs,))
            out[k] = full.reshape((-1,) + self._fields_spatial_shape[k])
        return out

    _reconstructed_fields = reconstructed_fields

    # ------------------------------------------------------------------------------------------
    # predict (array.py:1299-1428)
    # ------------------------------------------------------------------------------------------
    def predict(self, left=None, right=None, n=None, scaling='None', phase_shift=0):
        """Project new data on the singular vectors (rotated if the model is)."""
        new = {k: d.copy() for k, d in zip(self._keys, [left, right]) if d is not None}
        V = self._get_V(rotated=False)
        sqrt_svals = np.sqrt(self._get_svals())
        R = self.rotation_matrix(inverse_transpose=True)
        n_rot = R.shape[0]
        if n is None:
            n = n_rot
        out = {}
        for k, x in new.items():
            try:
                x = x.reshape(x.shape[0], self._n_variables[k])[:, self._no_nan_index[k]]
            except ValueError as err:
                if len(x.shape) != len(self._shape[k]):
                    msg = ('Error in {:} field. Dimension of new data ({:}) and the original field ({:}) do not match. '
                           'Did you forget the time dimension?').format(k, len(x.shape), len(self._shape[k]))
                elif x.shape[1:] != self._field_means[k].shape:
                    msg = ('Error in {:} field. Spatial dimensions of new data {:} and the original field {:} '
                           'do not match.').format(k, x.shape[1:], self._shape[k][1:])
                else:
                    msg = 'Dimension mismatch in {:} field.'.format(k)
                raise ValueError(msg) from err
            try:
                x = self._scale_X({k: x})[k]
            except ValueError as err:
                msg = ('Error in {:} field. Spatial dimensions of new data {:} and the original field {:} '
                       'do not match.').format(k, x.shape[1:], self._field_means[k].shape)
                raise ValueError(msg) from err
            pcs = (x @ V[k][:, :n_rot] / sqrt_svals[:n_rot]) @ R
            pcs = pcs[:, self._var_idx][:, :n]
            if self._analysis['is_complex']:
                pcs = pcs * cmath.rect(1, phase_shift)
            if scaling == 'None':
                pass
            elif scaling == 'eigen':
                pcs = pcs * self._get_norm(n, sorted=True)[k]
            elif scaling == 'max':
                pcs = pcs / np.nanmax(abs(self._get_pcs(n, 'None', phase_shift)[k].real), axis=0)
            elif scaling == 'std':
                pcs = pcs / np.nanstd(self._get_pcs(n, 'None', phase_shift)[k].real, axis=0)
            else:
                raise ValueError(_SCALINGS_MSG.format(scaling))
            out[k] = pcs
        return out

    # ------------------------------------------------------------------------------------------
    # significance (array.py:1716-1952)
    # ------------------------------------------------------------------------------------------
    def rule_n(self, n_runs, n_modes=None, seed=None, dtype=np.float64):
        """Rule N (Overland & Preisendorfer 1982): spectra of `n_runs` Gaussian surrogates, scaled to the model's sum.

        The surrogate loop (array.py:1753-1765) runs on the device; when `torch.distributed` is initialised the
        runs are sharded over the ranks (contiguous blocks) and gathered with one collective.
        `seed` (extension): key of the counter-based device generator; default: drawn from numpy's global RNG, so
        `np.random.seed(s)` makes the result reproducible like the reference's use of the global stream.
        Returns an array (modes x kept runs); runs whose rotation does not converge are dropped (array.py:1762-1763).
        """
        from . import dist
        if seed is None:
            seed = int(np.random.randint(0, 2 ** 31 - 1)) * 2 ** 31 + int(np.random.randint(0, 2 ** 31 - 1))
        m = self._n_observations
        n = self._n_variables
        rotated = self._analysis['is_rotated']
        n_rot = self._analysis['n_rot']
        rank = min([m['left']] + [n[k] for k in self._keys])
        n_out = n_rot if rotated else rank
        spectra, kept = dist.sharded_rule_n(
            self._device(), n_runs, T=m['left'], Nx=n['left'], Ny=n.get('right', 0), n_fields=len(self._keys),
            complexify=self._analysis['is_complex'], rotated=rotated, p=n_rot, power=self._analysis['power'],
            tol=1e-8, seed=seed, dtype=dtype, n_out=n_out)
        svals = spectra[kept.astype(bool)].T          # modes x kept runs
        ref = self._get_variance()
        svals /= svals.sum(axis=0) / ref.sum()
        return svals[self._get_slice(n_modes)]

    def rule_north(self, n=None):
        """North's rule of thumb (x sqrt(2) for complex models, Horel 1984)."""
        err = self._get_svals(n) * np.sqrt(2. / self._n_observations['left'])
        if self._analysis['is_complex']:
            err *= np.sqrt(2)
        return err

    def bootstrapping(self, n_runs, n_modes=20, axis=0, on_left=True, on_right=False, block_size=1, replace=True,
                      strategy='standard', disable_progress=False):
        """Monte Carlo (moving-block) bootstrap / permutation of the model (array.py:1813-1952).

        The block indices are drawn on the host from numpy's global RNG exactly as the reference draws them
        (tools/array.py:91-138), the replicates themselves - cumulative row resampling, centering, solve, rotation,
        variance - run on the device (`xmca_bootstrap_run`).  Column resampling (`axis=1`) and models with a
        fore/back-cast extension keep the reference's host loop with one device solve per replicate.
        """
        complexify = self._analysis['is_complex']
        extend = self._analysis['extend']
        period = self._analysis['theta_period']
        is_rotated = self._analysis['is_rotated']
        n_rot = self._analysis['n_rot']
        power = self._analysis['power']
        n_modes_max = self._get_min_mode(n_modes, rotated=True)
        var_surr = np.zeros([n_modes_max, n_runs])
        on_device = axis == 0 and not extend and not getattr(self, '_bootstrap_on_host', False)
        dev = self._device()
        n_obs = self._n_observations['left']
        for mode in range(n_modes):
            X_surr = self._get_X(original_scale=False, real=True)
            if strategy == 'iterative':
                X_rec = self._reconstructed_X(mode=mode, original_scale=False)
                for k in X_surr:
                    X_surr[k] -= X_rec[k]
            if on_device:
                if on_right and 'right' not in X_surr:
                    raise ValueError('No bootstrapping possible. There is no right field. Set `on_right=False`.')
                if n_obs % block_size:
                    raise ValueError('Length of data array ({:}) must be a multiple of block size {:}'.format(n_obs, block_size))
                for side, k in enumerate(self._keys):
                    dev.set_field(side, _device_ready(np.ascontiguousarray(X_surr[k])))
                dev.bootstrap_begin(len(self._keys))
                n_blocks = n_obs // block_size
                rank = min([n_obs] + [X_surr[k].shape[1] for k in self._keys])
                n_out = n_rot if is_rotated else rank
                # the reference resamples cumulatively (X_surr is overwritten, array.py:1935-1943): replicate r sees the rows
                # c_r = c_{r-1}[idx_r] of the original field.  The draws are made here in the reference's order, composed, and
                # the device runs all replicates in one call, several at a time.
                cum = np.arange(n_obs)
                composed = np.empty((n_runs, n_obs), dtype=np.int64)
                for run in range(n_runs):
                    if on_left or on_right:
                        # one draw per replicate, like tools/array.py:136 (both sides share it when both are resampled)
                        pick = np.random.choice(n_blocks, size=n_blocks, replace=replace)
                        rows = (pick[:, None] * block_size + np.arange(block_size)[None, :]).reshape(-1)
                        cum = cum[rows]
                    composed[run] = cum
                # (replicates are independent once composed: sharded over the ranks of a torch.distributed job like the Rule-N runs)
                from . import dist
                spec, kept = dist.sharded_bootstrap(dev, n_runs, T=n_obs, complexify=complexify, idx_left=composed if on_left else None,
                                                    idx_right=composed if on_right else None, rotated=is_rotated, p=n_rot,
                                                    power=max(power, 1), tol=1e-8, n_out=n_out)
                for run in range(n_runs):
                    if kept[run]:
                        var_surr[mode:, run] = spec[run, :n_modes_max - mode]
                if strategy == 'standard':
                    break
                continue
            for run in range(n_runs):
                if on_left and not on_right:
                    X_surr['left'] = block_bootstrap(X_surr['left'], axis=axis, block_size=block_size, replace=replace)
                elif on_right and not on_left:
                    try:
                        X_surr['right'] = block_bootstrap(X_surr['right'], axis=axis, block_size=block_size,
                                                          replace=replace)
                    except KeyError as err:
                        raise ValueError('No bootstrapping possible. There is no right field. '
                                         'Set `on_right=False`.') from err
                elif on_left and on_right:
                    n_left = X_surr['left'].shape[1]
                    both = block_bootstrap(np.concatenate(list(X_surr.values()), axis=1), axis=axis,
                                           block_size=block_size, replace=replace)
                    X_surr['left'], X_surr['right'] = both[:, :n_left], both[:, n_left:]
                model = MCA(*list(X_surr.values()), handle=self._handle_override)
                model.solve(complexify=complexify, extend=extend, period=period)
                if is_rotated:
                    try:
                        model.rotate(n_rot, power)
                    except RuntimeError:
                        continue
                var_surr[mode:, run] = model._get_variance(n_modes_max - mode)
            if strategy == 'standard':
                break
        return var_surr

    # ------------------------------------------------------------------------------------------
    # truncation / persistence (array.py:1602-1714, :1954-2012)
    # ------------------------------------------------------------------------------------------
    def truncate(self, n):
        """Keep only the first `n` modes (must not cut into a rotated solution)."""
        if self._analysis['is_rotated'] and n < self._analysis['n_rot']:
            raise ValueError('Cannot truncte rotated solution. Please ensure `n` > `n_rot`')
        if n < self._singular_values.size:
            self._singular_values = self._singular_values[:n]
            for k in self._keys:
                self._V[k] = self._V[k][:, :n]
            self._analysis['is_truncated'] = True
            self._analysis['is_truncated_at'] = n

    def _get_analysis_path(self, path=None):
        if path is None:
            folder = secure_str('_'.join(self._field_names.values()))
            return os.path.join(os.getcwd(), 'xmca', folder)
        return path if os.path.isabs(path) else os.path.abspath(path)

    def _create_analysis_path(self, path):
        path = self._get_analysis_path(path)
        os.makedirs(path, exist_ok=True)

    def _create_info_file(self, path):
        """`info.xmca`: `key : value` lines in the reference's layout (array.py:1629-1659)."""
        sep = '\n#' + '-' * 79
        lines = [wrap_str('This file contains information neccessary to load stored analysis'
                          'data from xmca module.'),
                 '\n# To load this analysis use:', '\n# from xmca.xarray import xMCA', '\n# mca = xMCA()',
                 '\n# mca.load_analysis(PATH_TO_THIS_FILE)', '\n', sep, sep,
                 '\n{:<20} : {:<57}'.format('created', datetime.now().strftime("%Y-%m-%d %H:%M:%S")), sep]
        for key, name in self._field_names.items():
            lines.append('\n{:<20} : {:<57}'.format(key, str(name)))
        lines.append(sep)
        for key, info in self._analysis.items():
            if key in ['is_bivariate', 'is_complex', 'is_rotated', 'is_truncated']:
                lines.append(sep)
            lines.append('\n{:<20} : {:<57}'.format(key, str(info)))
        with open(os.path.join(path, 'info.xmca'), 'w+') as fh:
            fh.write(''.join(lines))

    def _get_file_names(self, format):
        fields, eofs = {}, {}
        for key, variable in self._field_names.items():
            variable = secure_str(variable)
            fields[key] = '.'.join([variable, format])
            eofs[key] = '.'.join(['_'.join([variable, 'eofs']), format])
        return {'fields': fields, 'eofs': eofs, 'pcs': {}, 'singular': '.'.join(['singular_values', format]), 'norm': {}}

    def _save_data(self, data_array, path, *args, **kwargs):
        raise NotImplementedError('only works for `xarray`')

    def _set_analysis(self, key, value):
        try:
            key_type = type(self._analysis[key])
        except KeyError:
            raise KeyError("Key `{}` not found in info file.".format(key))
        self._analysis[key] = (value == 'True') if key_type == bool else key_type(value)

    def _set_info_from_file(self, path):
        with open(path, 'r') as fh:
            for line in fh.readlines():
                if line[0] == '#':
                    continue
                key = line.split(':')[0].rstrip()
                if key in ['left', 'right']:
                    self._field_names[key] = line.split(':')[1].strip()
                if key in self._analysis.keys():
                    if key == 'version':
                        continue                      # the file's writer, not this package
                    self._set_analysis(key, line.split(':')[1].strip())

    def plot(self, *args, **kwargs):
        """Not part of the accelerated path (xmca/array.py:1430-1711 draws with matplotlib / cartopy): every number the
        reference's figure shows is available from `eofs()`, `pcs()`, `explained_variance()`."""
        raise NotImplementedError('xmca_amd does not draw: plot()/save_plot() of the reference need matplotlib and cartopy; '
                                  'use eofs(), pcs(), explained_variance() of this model with your own plotting code')

    def save_plot(self, *args, **kwargs):
        return self.plot(*args, **kwargs)

    def load_analysis(self, path, fields=None, eofs=None, singular_values=None):
        """Restore a model written by `save_analysis` (fields / eofs / singular values supplied by the caller)."""
        self._set_info_from_file(path)
        self._keys = ['left', 'right'] if self._analysis['is_bivariate'] else ['left']
        self._ingest(fields)
        if self._analysis['is_normalized']:
            self.normalize()
        if self._analysis['is_complex']:
            self._fields = self._complexify(self._fields)
        self._V = {}
        self._norm = {}
        self._singular_values = singular_values
        self._variance = singular_values
        self._var_idx = np.argsort(singular_values)[::-1]
        for key in self._keys:
            self._norm[key] = np.sqrt(singular_values)
            n_modes = eofs[key].shape[-1]
            flat = eofs[key].reshape(self._n_variables[key], n_modes)
            self._V[key] = remove_nan_cols(flat.T).T
        if self._analysis['is_rotated']:
            self.rotate(self._analysis['n_rot'], self._analysis['power'])

    def summary(self):
        """Print the analysis meta information."""
        import yaml
        print(yaml.dump({k: str(v) for k, v in self._analysis.items()}, sort_keys=False, default_flow_style=False))


def _gpu_visible():
    try:
        return _hip.load_library().xmca_device_count() > 0
    except Exception:
        return False


def secure_str(string):
    """file name form of a field name (info.xmca / netCDF names of save_analysis, array.py:1602-1627)"""
    return string.lower().replace(' ', '_')


def wrap_str(string):
    """'# '-prefixed comment block (header of info.xmca, array.py:1629-1640)"""
    import textwrap
    return textwrap.indent(textwrap.fill(string, width=80), '# ')


def _real_dtype(dt):
    dt = np.dtype(dt)
    if dt in (np.float32, np.complex64):
        return np.float32
    return np.float64


def _device_ready(field):
    """float32/float64 (or their complex pairs) pass through; anything else is promoted to float64."""
    dt = np.dtype(field.dtype)
    if dt in (np.float32, np.float64, np.complex64, np.complex128):
        return field
    return field.astype(np.complex128 if np.iscomplexobj(field) else np.float64)
