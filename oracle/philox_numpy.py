"""TEST INFRASTRUCTURE (never imported by the product): numpy restatement of the device's surrogate generator.

`xmca_surrogate` / `philox_normal_kernel` (xmca_amd/csrc/kernels.h) replace the `np.random.standard_normal([m, n])` of the
reference's Rule N loop (xmca/array.py:1755-1756) by a counter-based stream so that the surrogate of run r does not depend on
the GPU that generates it:

    counter = (pair index lo, pair index hi, run, side), key = (seed lo, seed hi)      Philox4x32-10 (Salmon et al. 2011)
    a = r0:r1, b = r2:r3 (64 bits each);  u = ((x >> 11) + 0.5) 2^-53  in (0, 1)
    out[2i] = sqrt(-2 ln u1) cos(2 pi u2),  out[2i + 1] = sqrt(-2 ln u1) sin(2 pi u2)   (Box-Muller)

The integer part is bit-exact by construction; the normals agree with the device's to the last few ulps of libm
(`tests/test_gpu_rule_n.py::test_numpy_generator_equals_the_device_generator`: <= 4e-15 absolute).  With it a surrogate of
any size exists without a GPU, so the REAL reference can be run on exactly the numbers the device generates
(`oracle/make_config_goldens.py c4_run0`).
"""
import numpy as np

_M0 = np.uint64(0xD2511F53)
_M1 = np.uint64(0xCD9E8D57)
_W0 = 0x9E3779B9
_W1 = 0xBB67AE85
_MASK = np.uint64(0xFFFFFFFF)
_S32 = np.uint64(32)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """c*: uint64 arrays holding 32-bit counter words; k0, k1: python ints. Returns the four output words (uint64 arrays < 2^32)."""
    for _ in range(10):
        p0 = _M0 * c0                       # 32 x 32 -> 64 bits, exact in uint64
        p1 = _M1 * c2
        hi0, lo0 = p0 >> _S32, p0 & _MASK
        hi1, lo1 = p1 >> _S32, p1 & _MASK
        c0, c1, c2, c3 = hi1 ^ c1 ^ np.uint64(k0), lo1, hi0 ^ c3 ^ np.uint64(k1), lo0
        k0 = (k0 + _W0) & 0xFFFFFFFF
        k1 = (k1 + _W1) & 0xFFFFFFFF
    return c0, c1, c2, c3


def _sincospi(x):
    """sin(pi x), cos(pi x) for x in [0, 2) with an exact argument reduction (|r| <= 1/4 before the multiplication by pi)."""
    q = np.rint(2.0 * x)                    # nearest multiple of 1/2
    r = x - 0.5 * q                         # exact
    s, c = np.sin(np.pi * r), np.cos(np.pi * r)
    q = q.astype(np.int64) & 3
    sn = np.where(q == 0, s, np.where(q == 1, c, np.where(q == 2, -s, -c)))
    cs = np.where(q == 0, c, np.where(q == 1, -s, np.where(q == 2, -c, s)))
    return sn, cs


def surrogate(n, seed, run, side, dtype=np.float64, chunk=1 << 22):
    """The n normals of (seed, run, side), in the device's order."""
    n = int(n)
    pairs = (n + 1) // 2
    out = np.empty(2 * pairs, dtype=dtype)
    k0, k1 = int(seed) & 0xFFFFFFFF, (int(seed) >> 32) & 0xFFFFFFFF
    for lo in range(0, pairs, chunk):
        hi = min(pairs, lo + chunk)
        i = np.arange(lo, hi, dtype=np.uint64)
        r0, r1, r2, r3 = philox4x32_10(i & _MASK, i >> _S32, np.full(hi - lo, run, np.uint64), np.full(hi - lo, side, np.uint64), k0, k1)
        a = (r0 << _S32) | r1
        b = (r2 << _S32) | r3
        u1 = ((a >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)
        u2 = ((b >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)
        rad = np.sqrt(-2.0 * np.log(u1))
        sn, cs = _sincospi(2.0 * u2)
        out[2 * lo:2 * hi:2] = rad * cs
        out[2 * lo + 1:2 * hi:2] = rad * sn
    return out[:n]


def surrogate_fields(T, widths, seed, run, dtype=np.float64):
    """The T x N fields of surrogate `run` (side = position of the field), as `xmca_rule_n` generates them."""
    return [surrogate(T * w, seed, run, side, dtype).reshape(T, w) for side, w in enumerate(widths)]
