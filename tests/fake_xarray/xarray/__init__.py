"""Minimal stand-in for the `xarray` package (NOT installed in the build / GPU image) - test infrastructure only.

Implements the small part of `xarray.DataArray` that `xmca_amd.xarray.xMCA` (like the reference's `xmca.xarray.xMCA`)
touches: construction from values + dims + coords + name + attrs, `.values` / `.data` / `.dims` / `.coords` / `.shape`
/ `.name` / `.attrs` / `.real`, numpy ufuncs on a DataArray, and `*` between DataArrays with broadcasting BY DIMENSION
NAME.  `tests/test_gpu_xarray_facade.py` puts this directory on sys.path only when the real package is missing.
"""
import numpy as np

__version__ = "0.0-standin"


class DataArray:
    def __init__(self, data, dims=None, coords=None, name=None, attrs=None):
        self.values = np.asarray(data)
        self.dims = tuple(dims) if dims is not None else tuple("dim_%d" % i for i in range(self.values.ndim))
        assert len(self.dims) == self.values.ndim, (self.dims, self.values.shape)
        self.name = name
        self.attrs = dict(attrs or {})
        self.coords = {}
        for k, v in dict(coords or {}).items():
            if isinstance(v, DataArray):
                self.coords[k] = v
            else:
                v = np.asarray(v)
                self.coords[k] = DataArray(v, dims=(k,), name=k) if v.ndim == 1 else v
        for d, n in zip(self.dims, self.values.shape):
            if d in self.coords and isinstance(self.coords[d], DataArray):
                assert self.coords[d].values.shape == (n,), (d, n)

    data = property(lambda self: self.values)
    shape = property(lambda self: self.values.shape)
    dtype = property(lambda self: self.values.dtype)
    size = property(lambda self: self.values.size)
    ndim = property(lambda self: self.values.ndim)
    real = property(lambda self: DataArray(self.values.real, self.dims, self.coords, self.name, self.attrs))

    def __getitem__(self, key):
        if isinstance(key, str):
            return self.coords[key]
        return DataArray(self.values[key], dims=self.dims if np.ndim(self.values[key]) == self.values.ndim else None)

    def __array__(self, dtype=None, copy=None):
        return self.values if dtype is None else self.values.astype(dtype)

    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        if method != "__call__":
            return NotImplemented
        if len(inputs) == 2 and all(isinstance(x, DataArray) for x in inputs):
            a, b = _align(*inputs)
            return DataArray(ufunc(a[0], b[0], **kwargs), dims=a[1], coords={**inputs[1].coords, **inputs[0].coords})
        me = next(x for x in inputs if isinstance(x, DataArray))
        raw = [x.values if isinstance(x, DataArray) else x for x in inputs]
        return DataArray(ufunc(*raw, **kwargs), dims=me.dims, coords=me.coords, name=me.name)

    def __mul__(self, other):
        return np.multiply(self, other)

    __rmul__ = __mul__

    def __add__(self, other):
        return np.add(self, other)

    __radd__ = __add__

    def __repr__(self):
        return "<stand-in DataArray %s %s %r>" % (self.name, dict(zip(self.dims, self.shape)), self.values.dtype)


def _align(a, b):
    """broadcast two DataArrays by dimension name (result dims: a's, then b's new ones)"""
    dims = list(a.dims) + [d for d in b.dims if d not in a.dims]

    def expand(x):
        order = [x.dims.index(d) for d in dims if d in x.dims]
        v = np.transpose(x.values, order)
        shape = [x.values.shape[x.dims.index(d)] if d in x.dims else 1 for d in dims]
        return v.reshape(shape)
    return (expand(a), tuple(dims)), (expand(b), tuple(dims))
