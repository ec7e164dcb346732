"""Host-side array helpers (mirror of xmca/tools/array.py: same names, same behaviour)."""
import warnings

import numpy as np


def remove_mean(arr):
    """Subtract the column means (a column holding a NaN becomes all-NaN).  xmca/tools/array.py:14-24"""
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", category=RuntimeWarning)
        return arr - arr.mean(axis=0)


def get_nan_cols(arr):
    """Boolean index of columns containing at least one NaN.  xmca/tools/array.py:27-42"""
    return np.isnan(arr).any(axis=0)


def remove_nan_cols(arr):
    """Drop the columns flagged by `get_nan_cols`.  xmca/tools/array.py:45-62"""
    return arr[:, ~get_nan_cols(arr)]


def has_nan_time_steps(array):
    """True when some time step (axis 0) is NaN everywhere.  xmca/tools/array.py:65-73"""
    other_axes = tuple(range(1, array.ndim))
    return bool(np.isnan(array).all(axis=other_axes).any())


def pearsonr(x, y):
    """Column-wise Pearson correlation of x (T x n) with y (T x m) and two-sided p-values.  xmca/tools/array.py:76-88"""
    import scipy.stats
    if x.shape[0] != y.shape[0]:
        raise ValueError('Time dimensions are different.')
    n = x.shape[0]
    r = np.corrcoef(x, y, rowvar=False)[:x.shape[1], x.shape[1]:]
    dist = scipy.stats.beta(n / 2 - 1, n / 2 - 1, loc=-1, scale=2)
    return r, 2 * dist.cdf(-abs(r))


def block_bootstrap(arr, axis=0, block_size=1, replace=True):
    """(Moving-)block resampling of a 2-D array along `axis` using the global numpy RNG.  xmca/tools/array.py:91-138"""
    if axis not in (0, 1):
        raise ValueError('{:} not a valid axis. either 0 or 1.'.format(axis))
    work = arr.T if axis == 1 else arr
    n_obs = work.shape[0]
    try:
        blocks = work.reshape(-1, block_size, work.shape[1])
    except ValueError as err:
        msg = 'Length of data array ({:}) must be a multiple of block size {:}'.format(n_obs, block_size)
        raise ValueError(msg) from err
    pick = np.random.choice(blocks.shape[0], size=blocks.shape[0], replace=replace)
    out = blocks[pick].reshape(work.shape)
    return out.T if axis == 1 else out
