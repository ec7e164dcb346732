"""ctypes binding of libxmca_hip.so (C ABI: include/xmca_hip.h).

The product path has no CPU fallback: importing this module without the built
library, or creating a handle without a visible MI355X, raises.
"""
import ctypes
import weakref
import ctypes.util
import os

import numpy as np

from . import build as _build

XMCA_F32, XMCA_F64 = 0, 1
HOST, DEVICE = 0, 1

ERR_INVALID, ERR_HIP, ERR_NOT_CONVERGED, ERR_STATE, ERR_UNSUPPORTED, ERR_NUMERIC = -1, -2, -3, -4, -5, -6

_c_i64 = ctypes.c_int64
_c_int = ctypes.c_int
_c_dbl = ctypes.c_double
_vp = ctypes.c_void_p
_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int)

# name -> (restype, argtypes); the ABI test checks every one of these is exported
SIGNATURES = {
    "xmca_version": (ctypes.c_char_p, []),
    "xmca_abi_version": (_c_int, []),
    "xmca_device_count": (_c_int, []),
    "xmca_create": (_c_int, [_c_int, ctypes.POINTER(_vp)]),
    "xmca_destroy": (None, [_vp]),
    "xmca_last_error": (ctypes.c_char_p, [_vp]),
    "xmca_set_field": (_c_int, [_vp, _c_int, _vp, _vp, _c_i64, _c_i64, _c_int, _c_int]),
    "xmca_complexify": (_c_int, [_vp, _vp]),
    "xmca_solve": (_c_int, [_vp, _c_int, _c_i64, ctypes.POINTER(_c_i64)]),
    "xmca_get_singular_values": (_c_int, [_vp, _vp, _c_i64]),
    "xmca_get_vectors": (_c_int, [_vp, _c_int, _vp, _c_i64, _c_int]),
    "xmca_get_eofs": (_c_int, [_vp, _c_int, _vp, _c_i64, _c_i64, _c_int, _vp, _c_int]),
    "xmca_center_field": (_c_int, [_vp, _c_int, _vp, _vp, ctypes.POINTER(_c_i64)]),
    "xmca_compact_field": (_c_int, [_vp, _c_int, _vp, ctypes.POINTER(_c_i64)]),
    "xmca_scale_field": (_c_int, [_vp, _c_int, _vp, _c_int]),
    "xmca_get_field": (_c_int, [_vp, _c_int, _vp]),
    "xmca_bootstrap_begin": (_c_int, [_vp, _c_int]),
    "xmca_bootstrap_run": (_c_int, [_vp, _vp, _vp, _vp, _c_int, _c_int, _c_int, _c_dbl, _vp, ctypes.POINTER(_c_int), _c_i64]),
    "xmca_bootstrap_runs": (_c_int, [_vp, _vp, _vp, _vp, _c_i64, _c_int, _c_int, _c_int, _c_dbl, _vp, _vp, _c_i64]),
    "xmca_correlate": (_c_int, [_vp, _c_int, _vp, _c_i64, _c_i64, _vp]),
    "xmca_project": (_c_int, [_vp, _c_int, _vp, _c_i64, _c_i64, _c_int, _vp, ctypes.POINTER(_c_int)]),
    "xmca_is_complex": (_c_int, [_vp]),
    "xmca_vectors_are_f32": (_c_int, [_vp, _c_int]),
    "xmca_persistent_giveups": (ctypes.c_longlong, []),
    "xmca_get_solve_info": (_c_int, [_vp, _vp, _c_int]),
    "xmca_rotate_loadings": (_c_int, [_vp, _vp, _c_i64, _c_i64, _c_int, _c_int, _c_int, _c_dbl, _c_int, _c_int, _c_dbl,
                                      _vp, _vp, _vp, _vp, _vp, _ip]),
    "xmca_rotate_solved": (_c_int, [_vp, _c_int, _c_int, _c_dbl, _c_int, _vp, _vp, _vp, _vp, _ip]),
    "xmca_rule_n": (_c_int, [_vp, _c_i64, _c_i64, _c_i64, _c_int, _vp, _c_int, _c_int, _c_int, _c_dbl, _c_i64, _c_i64,
                             ctypes.c_uint64, _c_int, _vp, _vp, _c_i64]),
    "xmca_comm_unique_id": (_c_int, [_vp]),
    "xmca_comm_create": (_c_int, [_vp, _vp, _c_int, _c_int, ctypes.POINTER(_vp)]),
    "xmca_comm_destroy": (None, [_vp]),
    "xmca_comm_last_error": (ctypes.c_char_p, [_vp]),
    "xmca_comm_allgather": (_c_int, [_vp, _vp, _vp, _c_i64]),
    "xmca_comm_broadcast": (_c_int, [_vp, _vp, _c_i64, _c_int]),
    "xmca_comm_info": (_c_int, [_vp, _ip, _ip, ctypes.POINTER(_c_i64), ctypes.POINTER(_c_i64)]),
    "xmca_rule_n_sharded": (_c_int, [_vp, _vp, _c_i64, _c_i64, _c_i64, _c_i64, _c_int, _vp, _c_int, _c_int, _c_int, _c_dbl,
                                     ctypes.c_uint64, _c_int, _vp, _vp, _c_i64]),
    "xmca_surrogate": (_c_int, [_vp, _c_i64, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, _vp]),
    "xmca_get_timings": (_c_int, [_vp, ctypes.c_char_p, _c_int, _vp, _c_int]),
    "xmca_get_reduction_info": (_c_int, [_vp, ctypes.c_char_p, _c_int]),
    "xmca_reset_timings": (_c_int, [_vp]),
    "xmca_fft": (_c_int, [_vp, _vp, _vp, _c_int, _c_int, _c_int, _vp, _vp]),
    "xmca_pool_bytes": (_c_int, [_vp, ctypes.POINTER(ctypes.c_int64)]),
    "xmca_trim_pool": (_c_int, [_vp]),
    "xmca_gemm": (_c_int, [_vp, _vp, _c_i64, _c_int, _vp, _c_i64, _c_int, _vp, _c_int, _c_int, _c_int, _c_int, _c_dbl,
                           _c_int, _c_int, _c_int]),
    "xmca_eigh": (_c_int, [_vp, _vp, _c_int, _c_int, _vp, _vp, _vp]),
    "xmca_cholesky": (_c_int, [_vp, _vp, _c_int, _c_int, _c_dbl, _vp, ctypes.POINTER(_c_int)]),
    "xmca_bench_gram": (_c_int, [_vp, _c_int, _c_int, _dp, _dp, _dp]),
    "xmca_bench_gemm": (_c_int, [_vp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _dp]),
}

_lib = None


def library_path():
    return _build.LIB


ABI_VERSION = 9          # bumped whenever a signature of include/xmca_hip.h changes; checked against xmca_abi_version()


def load_library():
    """Loads libxmca_hip.so and binds every symbol of include/xmca_hip.h.  The library is never built implicitly
    (`python -m xmca_amd.build` / `__graft_entry__.build()` do that); a missing library, a missing symbol or a library
    built from an older header (ABI number) raises ImportError - there is no CPU fallback."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise ImportError(
            "xmca_amd: %s is missing. Build it with `python -m xmca_amd.build` (needs hipcc, "
            "--offload-arch=gfx950). There is no CPU fallback." % path)
    lib = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as err:
            raise ImportError("xmca_amd: %s does not export %s - rebuild it with `python -m xmca_amd.build --force`"
                              % (path, name)) from err
        fn.restype = res
        fn.argtypes = args
    have = lib.xmca_abi_version()
    if have != ABI_VERSION:
        raise ImportError("xmca_amd: %s was built for ABI %d, this package binds ABI %d - rebuild it with "
                          "`python -m xmca_amd.build --force`" % (path, have, ABI_VERSION))
    _lib = lib
    return lib


class HipError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(message)
        self.code = code


def _raise(code, message):
    if code == ERR_INVALID:
        raise ValueError(message)
    if code == ERR_NOT_CONVERGED:
        raise RuntimeError(message)
    if code == ERR_UNSUPPORTED:
        raise NotImplementedError(message)
    if code == ERR_NUMERIC:
        raise np.linalg.LinAlgError(message)
    raise HipError(code, message)


def _ptr(a):
    return None if a is None else a.ctypes.data_as(_vp)


def _np_dtype_code(dt):
    dt = np.dtype(dt)
    if dt in (np.float32, np.complex64):
        return XMCA_F32
    if dt in (np.float64, np.complex128):
        return XMCA_F64
    raise TypeError("unsupported dtype %s (float32 / float64 / complex64 / complex128 only)" % dt)


def hilbert_imag_column(T):
    """First column of the imaginary part of the analytic-signal operator of scipy.signal.hilbert (axis 0).

    hilbert(x) = ifft(fft(x) * h) with h = [1, 2, ..., 2, (1), 0, ...]; its imaginary part acts on a real x as the
    real circulant matrix Ht[t, s] = col[(t - s) mod T] with col = imag(ifft(h)).
    """
    h = np.zeros(T)
    if T % 2 == 0:
        h[0] = h[T // 2] = 1.0
        h[1:T // 2] = 2.0
    else:
        h[0] = 1.0
        h[1:(T + 1) // 2] = 2.0
    return np.ascontiguousarray(np.fft.ifft(h).imag)


def hilbert_imag_operator(T):
    """The full T x T operator (tests / documentation)."""
    col = hilbert_imag_column(T)
    idx = (np.arange(T)[:, None] - np.arange(T)[None, :]) % T
    return np.ascontiguousarray(col[idx])


class Handle:
    """One device + stream + workspace.  Not thread-safe."""

    def __init__(self, device=0):
        self._lib = load_library()
        self._h = _vp()
        n = self._lib.xmca_device_count()
        if n <= 0:
            raise HipError(ERR_HIP, "xmca_amd: no HIP device visible (MI355X / gfx950 required; there is no CPU fallback)")
        rc = self._lib.xmca_create(int(device), ctypes.byref(self._h))
        if rc != 0:
            raise HipError(rc, "xmca_create(device=%d) failed" % device)
        self.device = device
        self._keep = []      # host / device buffers that must outlive the handle's use of them

    def close(self):
        if getattr(self, "_h", None):
            self._lib.xmca_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            msg = self._lib.xmca_last_error(self._h)
            _raise(rc, msg.decode("utf-8", "replace") if msg else "xmca error %d" % rc)

    # ---- result ownership -------------------------------------------------------------------
    def hold_result(self, holder):
        """`holder` (anything with `_materialize_vectors()`) still reads the vectors of the last solve from the device
        on demand; it is asked to fetch them before anything invalidates that result."""
        self._result_holder = weakref.ref(holder)

    def release_result(self):
        ref, self._result_holder = getattr(self, "_result_holder", None), None
        holder = ref() if ref is not None else None
        if holder is not None:
            holder._materialize_vectors()

    # ---- fields -------------------------------------------------------------------------------
    def set_field(self, side, field):
        """field: T x N numpy array (real or complex, float32/float64 based)."""
        self.release_result()
        self.fields_owner = None            # whoever uploads claims the resident fields afterwards (MCA._upload_fields)
        field = np.asarray(field)
        if field.ndim != 2:
            raise ValueError("field must be 2-D (time x space)")
        code = _np_dtype_code(field.dtype)
        T, N = field.shape
        if np.iscomplexobj(field):
            re = np.ascontiguousarray(field.real)
            im = np.ascontiguousarray(field.imag)
        else:
            re = np.ascontiguousarray(field)
            im = None
        self._check(self._lib.xmca_set_field(self._h, side, _ptr(re), _ptr(im), T, N, code, HOST))

    def set_field_device(self, side, re_ptr, im_ptr, T, N, dtype):
        """Adopt device pointers (e.g. torch tensors' data_ptr()); the caller keeps them alive."""
        self.release_result()
        self.fields_owner = None
        self._check(self._lib.xmca_set_field(self._h, side, _vp(re_ptr), _vp(im_ptr) if im_ptr else None, T, N,
                                             _np_dtype_code(dtype), DEVICE))

    def complexify(self, T):
        self.release_result()
        ht = hilbert_imag_column(T)
        self._check(self._lib.xmca_complexify(self._h, _ptr(ht)))

    def decomplexify(self):
        """Back to the real resident fields (undoes `complexify` for the next solve)."""
        self.release_result()
        self._check(self._lib.xmca_complexify(self._h, None))

    # ---- solve --------------------------------------------------------------------------------
    def solve(self, n_fields, n_vec=-1):
        self.release_result()
        rank = _c_i64(0)
        self._check(self._lib.xmca_solve(self._h, n_fields, n_vec, ctypes.byref(rank)))
        return int(rank.value)

    def solve_info(self):
        info = np.zeros(12, dtype=np.int32)
        self._check(self._lib.xmca_get_solve_info(self._h, _ptr(info), 12))
        return [{"sweeps": int(info[3 * i]), "tile": int(info[3 * i + 1]), "slots": int(info[3 * i + 2]), "lr_step": int(info[9 + i]) & 1,
                 "tridiag": (int(info[9 + i]) >> 1) & 1} for i in range(3)]

    def singular_values(self, n):
        out = np.empty(n, dtype=np.float64)
        self._check(self._lib.xmca_get_singular_values(self._h, _ptr(out), n))
        return out

    def vectors(self, side, n_modes, N, dtype):
        """Returns Vt (n_modes x N); V = Vt.T."""
        cplx = bool(self._lib.xmca_is_complex(self._h))
        code = _np_dtype_code(dtype)
        base = np.float32 if code == XMCA_F32 else np.float64
        if cplx:
            out = np.empty((n_modes, N), dtype=np.complex64 if code == XMCA_F32 else np.complex128)
        else:
            out = np.empty((n_modes, N), dtype=base)
        self._check(self._lib.xmca_get_vectors(self._h, side, _ptr(out), n_modes, code))
        return out

    def eofs(self, side, N, m, W, dtype):
        """(N x q) EOFs of `side` in their final layout: V[:, :m] @ W mixed on the device (W: m x q float64 / complex128), or the
        first m vectors as they are (W None).  xmca_get_eofs: array.py:615-646 + :676-721 without an N x m pass on the host."""
        cplx = bool(self._lib.xmca_is_complex(self._h))
        code = _np_dtype_code(dtype)
        w_cplx = W is not None and np.iscomplexobj(W)
        if W is not None:
            W = np.ascontiguousarray(W, dtype=np.complex128 if w_cplx else np.float64)
            m, q = W.shape
        else:
            q = m
        if cplx or w_cplx:
            out = np.empty((N, q), dtype=np.complex64 if code == XMCA_F32 else np.complex128)
        else:
            out = np.empty((N, q), dtype=np.float32 if code == XMCA_F32 else np.float64)
        self._check(self._lib.xmca_get_eofs(self._h, side, _ptr(W), m, q, int(w_cplx), _ptr(out), code))
        return out

    def project(self, side, V, T):
        """U = X~ V (T x m) on the resident field of `side` (the analytic signal when the model is complex);
        float64 / complex128.  MCA._get_U's `fields[k] @ V[k]` (array.py:391)."""
        V = np.asarray(V)
        cplx = np.iscomplexobj(V)
        Vd = np.ascontiguousarray(V, dtype=np.complex128 if cplx else np.float64)
        N, m = Vd.shape
        out = np.empty((T, m), dtype=np.complex128)          # large enough for either result type
        out_cplx = _c_int(0)
        self._check(self._lib.xmca_project(self._h, side, _ptr(Vd), N, m, int(cplx), _ptr(out), ctypes.byref(out_cplx)))
        if out_cplx.value:
            return out
        return out.view(np.float64).reshape(-1)[:T * m].reshape(T, m).copy()

    def center_field(self, side, N):
        """Centers the resident (raw) field of `side` in place.  Returns (mean[N], std[N], number of NaN entries); when
        the last is not zero the field is unchanged."""
        self.release_result()
        mean = np.empty(N, dtype=np.float64)
        std = np.empty(N, dtype=np.float64)
        n_nan = _c_i64(0)
        self._check(self._lib.xmca_center_field(self._h, side, _ptr(mean), _ptr(std), ctypes.byref(n_nan)))
        return mean, std, int(n_nan.value)

    def compact_field(self, side, N):
        """Drops the NaN columns of the resident raw field of `side`.  Returns (keep mask[N], number of kept columns)."""
        self.release_result()
        keep = np.empty(N, dtype=np.int32)
        n_keep = _c_i64(0)
        self._check(self._lib.xmca_compact_field(self._h, side, _ptr(keep), ctypes.byref(n_keep)))
        return keep.astype(bool), int(n_keep.value)

    def scale_field(self, side, w, divide=False):
        """Multiplies (divides) column c of the resident real field of `side` by w[c]; `w` in the field's dtype."""
        self.release_result()
        w = np.ascontiguousarray(w)
        self._check(self._lib.xmca_scale_field(self._h, side, _ptr(w), int(bool(divide))))

    def get_field(self, side, shape, dtype):
        """Real plane of the resident field of `side` as a (T, N) array of `dtype` (the dtype it was set with)."""
        out = np.empty(shape, dtype=dtype)
        self._check(self._lib.xmca_get_field(self._h, side, _ptr(out)))
        return out

    def bootstrap_begin(self, n_fields):
        """Working copies of the resident fields for `bootstrap_run` (MCA.bootstrapping on the device)."""
        self.release_result()
        self._check(self._lib.xmca_bootstrap_begin(self._h, n_fields))

    def bootstrap_run(self, T, complexify, idx_left, idx_right, rotated, p, power, tol, n_out):
        """One replicate: resample rows (cumulatively), center, solve (+ rotate).  Returns (spectrum[n_out], kept)."""
        ht = hilbert_imag_column(T) if complexify else None
        il = None if idx_left is None else np.ascontiguousarray(idx_left, dtype=np.int64)
        ir = None if idx_right is None else np.ascontiguousarray(idx_right, dtype=np.int64)
        out = np.zeros(n_out, dtype=np.float64)
        kept = _c_int(0)
        self._check(self._lib.xmca_bootstrap_run(self._h, _ptr(ht), _ptr(il), _ptr(ir), int(rotated), int(p), int(power), float(tol),
                                                 _ptr(out), ctypes.byref(kept), n_out))
        return out, bool(kept.value)

    def bootstrap_runs(self, T, complexify, idx_left, idx_right, n_runs, rotated, p, power, tol, n_out):
        """All replicates in one call (several in flight on the device).  idx_*: (n_runs, T) COMPOSED row indices into the
        fields as they were at `bootstrap_begin`, or None.  Returns (spectra[n_runs, n_out], kept[n_runs])."""
        ht = hilbert_imag_column(T) if complexify else None
        il = None if idx_left is None else np.ascontiguousarray(idx_left, dtype=np.int64).reshape(n_runs, T)
        ir = None if idx_right is None else np.ascontiguousarray(idx_right, dtype=np.int64).reshape(n_runs, T)
        out = np.zeros((n_runs, n_out), dtype=np.float64)
        kept = np.zeros(n_runs, dtype=np.int32)
        self._check(self._lib.xmca_bootstrap_runs(self._h, _ptr(ht), _ptr(il), _ptr(ir), n_runs, int(rotated), int(p), int(power),
                                                  float(tol), _ptr(out), _ptr(kept), n_out))
        return out, kept.astype(bool)

    def correlate(self, side, Y, N):
        """r (N x m) = Pearson correlation of the real part of every column of the resident field `side` with the columns
        of Y (T x m).  tools/array.py:76-88 without the (N + m)^2 corrcoef matrix."""
        Yd = np.ascontiguousarray(np.asarray(Y).real, dtype=np.float64)
        T, m = Yd.shape
        r = np.empty((N, m), dtype=np.float64)
        self._check(self._lib.xmca_correlate(self._h, side, _ptr(Yd), T, m, _ptr(r)))
        return r

    # ---- rotation -----------------------------------------------------------------------------
    def rotate_loadings(self, L, n_left, power=1, tol=1e-8, max_iter=1000, varimax_only=False, want_B=False, gamma=1.0):
        L = np.asarray(L)
        cplx = np.iscomplexobj(L)
        Ld = np.ascontiguousarray(L, dtype=np.complex128 if cplx else np.float64)
        N, p = Ld.shape
        cdt = np.complex128 if cplx else np.float64
        R = np.empty((p, p), dtype=cdt)
        Phi = np.empty((p, p), dtype=cdt)
        nl = np.zeros(p)
        nr = np.zeros(p)
        B = np.empty((N, p), dtype=cdt) if want_B else None
        iters = _c_int(0)
        rc = self._lib.xmca_rotate_loadings(self._h, _ptr(Ld), N, int(n_left), p, int(cplx), int(power), float(tol),
                                            int(max_iter), int(varimax_only), float(gamma), _ptr(B), _ptr(R), _ptr(Phi), _ptr(nl),
                                            _ptr(nr), ctypes.byref(iters))
        self.last_iters = int(iters.value)
        self._check(rc)
        return {"B": B, "R": R, "Phi": Phi, "norm_left": nl, "norm_right": nr, "n_iter": int(iters.value)}

    def rotate_solved(self, p, power=1, tol=1e-8, max_iter=1000):
        """MCA.rotate on the resident result of the last solve (the loadings are built on the device)."""
        cplx = bool(self._lib.xmca_is_complex(self._h))
        cdt = np.complex128 if cplx else np.float64
        R = np.empty((p, p), dtype=cdt)
        Phi = np.empty((p, p), dtype=cdt)
        nl = np.zeros(p)
        nr = np.zeros(p)
        iters = _c_int(0)
        rc = self._lib.xmca_rotate_solved(self._h, int(p), int(power), float(tol), int(max_iter), _ptr(R), _ptr(Phi), _ptr(nl),
                                          _ptr(nr), ctypes.byref(iters))
        self.last_iters = int(iters.value)
        self._check(rc)
        return {"B": None, "R": R, "Phi": Phi, "norm_left": nl, "norm_right": nr, "n_iter": int(iters.value)}

    def vectors_are_f32(self, side=0):
        """the vectors of the last solve are resident in float32 (real float32 field, dual side): include/xmca_hip.h"""
        return bool(self._lib.xmca_vectors_are_f32(self._h, int(side)))

    def holds_result_of(self, holder):
        ref = getattr(self, "_result_holder", None)
        return ref is not None and ref() is holder

    # ---- rule N -------------------------------------------------------------------------------
    def rule_n(self, T, Nx, Ny, n_fields, complexify, rotated, p, power, tol, run_begin, run_end, seed, dtype, n_out):
        self.release_result()
        n = run_end - run_begin
        self.fields_owner = None            # the surrogates overwrite the resident fields
        spectra = np.zeros((max(n, 0), n_out), dtype=np.float64)
        kept = np.zeros(max(n, 0), dtype=np.int32)
        ht = hilbert_imag_column(T) if complexify else None
        if n > 0:
            self._check(self._lib.xmca_rule_n(self._h, T, Nx, Ny if n_fields == 2 else 0, n_fields, _ptr(ht), int(rotated),
                                              int(p), int(power), float(tol), run_begin, run_end, int(seed),
                                              _np_dtype_code(dtype), _ptr(spectra), _ptr(kept), n_out))
        return spectra, kept

    def rule_n_sharded(self, comm, n_runs, T, Nx, Ny, n_fields, complexify, rotated, p, power, tol, seed, dtype, n_out):
        """xmca_rule_n_sharded: this rank's block of the runs [0, n_runs) + ONE ncclAllGather (native RCCL communicator
        `comm`, see `Comm`); every rank gets all n_runs x n_out spectra and kept flags."""
        self.release_result()
        self.fields_owner = None
        spectra = np.zeros((max(n_runs, 0), n_out), dtype=np.float64)
        kept = np.zeros(max(n_runs, 0), dtype=np.int32)
        ht = hilbert_imag_column(T) if complexify else None
        self._check(self._lib.xmca_rule_n_sharded(self._h, comm._c, int(n_runs), T, Nx, Ny if n_fields == 2 else 0, n_fields,
                                                  _ptr(ht), int(rotated), int(p), int(power), float(tol), int(seed),
                                                  _np_dtype_code(dtype), _ptr(spectra), _ptr(kept), n_out))
        return spectra, kept

    def surrogate(self, n, seed, run, side):
        out = np.empty(n, dtype=np.float64)
        self._check(self._lib.xmca_surrogate(self._h, n, int(seed), int(run), int(side), _ptr(out)))
        return out

    # ---- instrumentation ----------------------------------------------------------------------
    def reduction_info(self):
        """names of the kernels of the last tridiagonal reduction (xmca_get_reduction_info)"""
        buf = ctypes.create_string_buffer(4096)
        n = self._lib.xmca_get_reduction_info(self._h, buf, 4096)
        if n < 0:
            self._check(n)
        return buf.value.decode()

    def timings(self):
        names = ctypes.create_string_buffer(4096)
        ms = np.zeros(64)
        n = self._lib.xmca_get_timings(self._h, names, 4096, _ptr(ms), 64)
        if n < 0:
            self._check(n)
        keys = names.value.decode().split(";") if n > 0 else []
        return {k: float(ms[i]) for i, k in enumerate(keys[:n])}

    def reset_timings(self):
        self._check(self._lib.xmca_reset_timings(self._h))

    def fft(self, x, sign=-1):
        """Batched DFT along the last axis of a 2-D array (rows), numpy.fft.fft convention for sign = -1 (csrc/fft.h)."""
        x = np.asarray(x)
        re = np.ascontiguousarray(x.real, dtype=np.float64)
        im = np.ascontiguousarray(x.imag, dtype=np.float64) if np.iscomplexobj(x) else None
        out_r, out_i = np.empty_like(re), np.empty_like(re)
        self._check(self._lib.xmca_fft(self._h, _ptr(re), _ptr(im), re.shape[0], re.shape[1], int(sign), _ptr(out_r), _ptr(out_i)))
        return out_r + 1j * out_i

    def pool_bytes(self):
        """device memory the handle keeps for re-use (solver temporaries; include/xmca_hip.h xmca_pool_bytes)"""
        n = ctypes.c_int64(0)
        self._check(self._lib.xmca_pool_bytes(self._h, ctypes.byref(n)))
        return int(n.value)

    def trim_pool(self):
        """give the kept device memory back to the driver"""
        self._check(self._lib.xmca_trim_pool(self._h))

    # ---- kernel-level entry points --------------------------------------------------------------
    def gemm(self, A, B, a_kfast=True, b_nfast=True, alpha=1.0, upper_only=False, mirror=0, splits=0):
        """C = alpha * op(A) op(B); A: (M,K) if a_kfast else (K,M); B: (K,N) if b_nfast else (N,K)."""
        A = np.ascontiguousarray(A)
        B = np.ascontiguousarray(B, dtype=A.dtype)
        code = _np_dtype_code(A.dtype)
        M, K = A.shape if a_kfast else A.shape[::-1]
        Kb, N = B.shape if b_nfast else B.shape[::-1]
        if K != Kb:
            raise ValueError("gemm: inner dimensions differ")
        C = np.zeros((M, N), dtype=np.float64)
        self._check(self._lib.xmca_gemm(self._h, _ptr(A), A.shape[1], int(a_kfast), _ptr(B), B.shape[1], int(b_nfast),
                                        _ptr(C), M, N, K, code, float(alpha), int(upper_only), int(mirror), int(splits)))
        return C

    def eigh(self, A, vectors=True):
        """Returns (lam descending, U) with A = U diag(lam) U^H; `vectors=False`: (lam, None), eigenvalues only."""
        A = np.asarray(A)
        cplx = np.iscomplexobj(A)
        Ad = np.ascontiguousarray(A, dtype=np.complex128 if cplx else np.float64)
        n = Ad.shape[0]
        lam = np.empty(n)
        Zh = np.empty((n, n), dtype=Ad.dtype) if vectors else None
        info = np.zeros(4, dtype=np.int32)
        self._check(self._lib.xmca_eigh(self._h, _ptr(Ad), n, int(cplx), _ptr(lam), _ptr(Zh) if vectors else None, _ptr(info)))
        self.last_eigh_info = {"sweeps": int(info[0]), "tile": int(info[1]), "slots": int(info[2]), "lr_step": int(info[3]) & 1,
                               "tridiag": (int(info[3]) >> 1) & 1}
        return lam, (Zh.conj().T if vectors else None)

    def cholesky(self, A, rel_shift=0.0):
        """Returns (R upper triangular with R^H R = A + rel_shift max(diag A) I, ok)."""
        A = np.asarray(A)
        cplx = np.iscomplexobj(A)
        Ad = np.ascontiguousarray(A, dtype=np.complex128 if cplx else np.float64)
        n = Ad.shape[0]
        R = np.empty((n, n), dtype=Ad.dtype)
        ok = _c_int(0)
        self._check(self._lib.xmca_cholesky(self._h, _ptr(Ad), n, int(cplx), float(rel_shift), _ptr(R), ctypes.byref(ok)))
        return R, bool(ok.value)

    def bench_gemm(self, M, N, K, dtype, a_kfast=True, b_nfast=True, upper_only=False, splits=0, reps=5):
        """ms per product C = op(A) op(B) on device-resident random operands."""
        ms = _c_dbl(0)
        self._check(self._lib.xmca_bench_gemm(self._h, M, N, K, _np_dtype_code(dtype), int(a_kfast), int(b_nfast), int(upper_only),
                                              int(splits), int(reps), ctypes.byref(ms)))
        return ms.value

    def bench_gram(self, side, reps):
        a, k, f = _c_dbl(0), _c_dbl(0), _c_dbl(0)
        self._check(self._lib.xmca_bench_gram(self._h, side, reps, ctypes.byref(a), ctypes.byref(k), ctypes.byref(f)))
        return {"avg_ms": a.value, "kernel_ms": k.value, "flops": f.value}


_default = {}


def default_handle(device=None):
    """Process-wide handle per device (LOCAL_RANK selects the device when not given)."""
    if device is None:
        device = int(os.environ.get("XMCA_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        n = load_library().xmca_device_count()
        if n > 0:
            device %= n
    if device not in _default:
        _default[device] = Handle(device)
    return _default[device]


COMM_ID_BYTES = 128


def comm_unique_id():
    """An ncclUniqueId (128 bytes) made by rank 0 for `Comm`; NotImplementedError when RCCL cannot be loaded."""
    lib = load_library()
    buf = ctypes.create_string_buffer(COMM_ID_BYTES)
    rc = lib.xmca_comm_unique_id(buf)
    if rc != 0:
        _raise(rc, "xmca_comm_unique_id failed (RCCL not available?)")
    return buf.raw


class Comm:
    """RCCL communicator of the C ABI (xmca_comm_*): one per process / GPU, created collectively by all `world` ranks from
    the unique id rank 0 made.  The only collective of the path is the all-gather of the rule_n spectra."""

    def __init__(self, handle, unique_id, rank, world):
        self._lib = load_library()
        self._c = None
        if len(unique_id) != COMM_ID_BYTES:
            raise ValueError("unique id must be %d bytes" % COMM_ID_BYTES)
        out = _vp()
        rc = self._lib.xmca_comm_create(handle._h, ctypes.c_char_p(bytes(unique_id)), int(rank), int(world), ctypes.byref(out))
        if rc != 0:
            handle._check(rc)
        self._c = out
        self._fin = weakref.finalize(self, self._lib.xmca_comm_destroy, out)

    def _check(self, rc):
        if rc != 0:
            _raise(rc, (self._lib.xmca_comm_last_error(self._c) or b"").decode())

    def allgather(self, local):
        local = np.ascontiguousarray(local, dtype=np.float64)
        rank, world, _, _ = self.info()
        out = np.empty((world,) + local.shape, dtype=np.float64)
        self._check(self._lib.xmca_comm_allgather(self._c, _ptr(local), _ptr(out), local.size))
        return out

    def broadcast(self, values, root=0):
        buf = np.ascontiguousarray(values, dtype=np.float64).copy()
        self._check(self._lib.xmca_comm_broadcast(self._c, _ptr(buf), buf.size, int(root)))
        return buf

    def info(self):
        r, w = _c_int(), _c_int()
        n, b = _c_i64(), _c_i64()
        self._check(self._lib.xmca_comm_info(self._c, ctypes.byref(r), ctypes.byref(w), ctypes.byref(n), ctypes.byref(b)))
        return r.value, w.value, n.value, b.value

    def close(self):
        if self._c is not None:
            self._fin()
            self._c = None
