"""Seeded input generators shared by oracle/make_goldens.py and the tests.

The golden .npz files store OUTPUTS of the reference; inputs are regenerated
here from fixed seeds (legacy ``np.random.seed`` for the reference's unit-test
inputs, PCG64 otherwise), except the reference's own real-data fixture
(sst/prcp) which is stored in ``tests/golden/reference_fixtures.npz``.
"""
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _unit_fields():
    # reference tests/unit/test_array.py:8-14
    np.random.seed(7)
    left = np.random.rand(500, 20, 15)
    np.random.seed(8)
    right = np.random.rand(500, 15, 10)
    return left, right


def _signal_field(rng, T, N, k, amp0=10.0, decay=0.8, noise=0.5, pcs=None):
    """low-rank signal with geometric amplitudes (well separated modes) + noise."""
    if pcs is None:
        pcs = rng.standard_normal((T, k))
    amp = amp0 * decay ** np.arange(k)
    pat = np.zeros((k, N))
    w = max(N // k, 1)
    for j in range(k):
        pat[j, j * w:(j + 1) * w] = np.hanning(w + 2)[1:-1]
    pat = pat + 0.2 * rng.standard_normal((k, k)) @ pat
    return (pcs * amp) @ pat + noise * rng.standard_normal((T, N)), pcs


def _wide(dtype=np.float64):
    rng = np.random.default_rng(11)
    T = 64
    t = np.arange(T)[:, None]
    f = np.linspace(0.03, 0.3, 8)[None, :]
    pcs = np.cos(2 * np.pi * f * t + rng.uniform(0, 6.28, (1, 8)))
    a, _ = _signal_field(rng, T, 300, 8, pcs=pcs)
    b, _ = _signal_field(rng, T, 200, 8, pcs=pcs)
    return a.astype(dtype), b.astype(dtype)


def _small():
    rng = np.random.default_rng(21)
    T = 40
    pcs = rng.standard_normal((T, 4))
    a, _ = _signal_field(rng, T, 24, 4, pcs=pcs, noise=0.3)
    b, _ = _signal_field(rng, T, 18, 4, pcs=pcs, noise=0.3)
    return a.reshape(T, 4, 6), b.reshape(T, 3, 6)


def _mixed():
    # one field wider than T, the other narrower: exercises both Gram orientations
    rng = np.random.default_rng(31)
    T = 48
    pcs = rng.standard_normal((T, 5))
    a, _ = _signal_field(rng, T, 120, 5, pcs=pcs)
    b, _ = _signal_field(rng, T, 30, 5, pcs=pcs)
    return a, b


def _loadings(tag):
    spec = {"r4": (300, 4, False, 41), "r10": (300, 10, False, 42), "r10p4": (300, 10, False, 43),
            "c4": (300, 4, True, 44), "c10p4": (300, 10, True, 45), "c10p2": (300, 10, True, 46)}
    n, p, cplx, seed = spec[tag]
    rng = np.random.default_rng(seed)
    # "simple structure": each column loads mostly on its own row band, then mixed by
    # a random orthogonal/unitary matrix so Varimax has something to undo.
    L = 0.15 * rng.standard_normal((n, p))
    w = n // p
    for j in range(p):
        L[j * w:(j + 1) * w, j] += np.hanning(w) * (3.0 - 0.15 * j)
    if cplx:
        L = L * np.exp(1j * rng.uniform(0, 2 * np.pi, (n, 1)) * 0.3) + 0.05j * rng.standard_normal((n, p))
        M = rng.standard_normal((p, p)) + 1j * rng.standard_normal((p, p))
    else:
        M = rng.standard_normal((p, p))
    Q, _ = np.linalg.qr(M)
    return L @ Q


def gen_A(T=2920, N=10_000, k=20, seed=0):
    """SURVEY.md Appendix C generator A (BASELINE config C2): low-rank Gaussian signal + unit noise, float64."""
    rng = np.random.default_rng(seed)
    return (rng.standard_normal((T, k)) * np.linspace(10, 1, k)) @ rng.standard_normal((k, N)) \
        + rng.standard_normal((T, N))


def gen_B(T=5000, Nx=20_000, Ny=15_000, k=40, seed=3, geometric=False):
    """SURVEY.md Appendix C generator B (configs C3 / C4): Hann patterns mixed by a random k x k matrix, cosine PCs,
    float64.  `geometric=True`: amplitudes 10 * 0.85^j (leading gaps >= 10 %, for the 1e-5 loadings check)."""
    rng = np.random.default_rng(seed)

    def pat(N):
        P = np.zeros((k, N))
        w = N // k
        for j in range(k):
            P[j, j * w:(j + 1) * w] = np.hanning(w)
        return rng.standard_normal((k, k)) @ P * 0.3 + P
    t = np.arange(T)[:, None]
    f = np.linspace(0.01, 0.2, k)[None, :]
    amp = 10.0 * 0.85 ** np.arange(k) if geometric else np.linspace(10, 5, k)
    pcs = amp * np.cos(2 * np.pi * f * t + rng.uniform(0, 6.28, (1, k)))
    A = pcs @ pat(Nx) + 0.5 * rng.standard_normal((T, Nx))
    B = pcs @ pat(Ny) + 0.5 * rng.standard_normal((T, Ny))
    return A, B


def gen_C(T=1200, ny=720, nx=1440, k=30, seed=5):
    """SURVEY.md Appendix C generator C (config C5): float32 noise + k unmixed Hann patterns, 3-D input (T, ny, nx)."""
    rng = np.random.default_rng(seed)
    N = ny * nx
    X = rng.standard_normal((T, N), dtype=np.float32) * np.float32(0.5)
    P = np.zeros((k, N), dtype=np.float32)
    w = N // k
    for j in range(k):
        P[j, j * w:(j + 1) * w] = np.hanning(w)
    pcs = (rng.standard_normal((T, k)) * np.linspace(10, 5, k)).astype(np.float32)
    X += pcs @ P
    return X.reshape(T, ny, nx)


def gen_C1(T=2920, shapes=((25, 53), (25, 27)), k=12, seed=9):
    """Stand-in for BASELINE configs[0] (xr.tutorial air_temperature split west / east: T = 2920 six-hourly steps, 25 x 53
    and 25 x 27 grid points, float32; the dataset itself is not in the image): smooth standing patterns shared by both
    halves, an annual + a diurnal cycle and red-noise PCs, white noise on top.  Both fields are narrower than T."""
    rng = np.random.default_rng(seed)
    t = np.arange(T)
    pcs = np.zeros((T, k))
    pcs[:, 0] = np.cos(2 * np.pi * t / 1460.0)
    pcs[:, 1] = np.sin(2 * np.pi * t / 4.0)
    for j in range(2, k):
        e = rng.standard_normal(T)
        for i in range(1, T):
            e[i] += 0.9 * e[i - 1]
        pcs[:, j] = e / e.std()
    amp = 12.0 * 0.75 ** np.arange(k)
    out = []
    for ny, nx in shapes:
        y = np.linspace(0, 1, ny)[:, None]
        x = np.linspace(0, 1, nx)[None, :]
        pat = np.stack([np.cos(np.pi * (j % 4 + 0.5) * x + 0.3 * j) * np.sin(np.pi * (j // 4 + 1) * y + 0.2 * j) for j in range(k)])
        pat = pat + (0.1 * rng.standard_normal((k, k)) @ pat.reshape(k, -1)).reshape(k, ny, nx)
        f = np.einsum('tk,kyx->tyx', pcs * amp, pat) + 0.4 * rng.standard_normal((T, ny, nx))
        out.append((f + 280.0).astype(np.float32))
    return tuple(out)


def make_input(name):
    """Returns a tuple of 1 or 2 arrays (time first)."""
    if name == "c1_standin":         # BASELINE configs[0]: air_temperature-shaped stand-in, float32
        return gen_C1()
    if name == "c2_full":            # BASELINE configs[1] at full size
        return (gen_A(),)
    if name == "c3_reduced":         # BASELINE configs[2] at T = 1000 x (4000, 3000), geometric amplitudes
        return gen_B(1000, 4000, 3000, geometric=True)
    if name == "c3_real_full":       # the fields of c3_full, solved without complexify
        return gen_B(5000, 20_000, 15_000, geometric=True)
    if name == "c3_full":            # BASELINE configs[2] at FULL size: T = 5000 x (20 000, 15 000), geometric amplitudes
        return gen_B(5000, 20_000, 15_000, geometric=True)
    if name == "c5_scaled":          # BASELINE configs[4] at T = 1200 x (144 x 288 = 41 472), float32, 3-D
        return (gen_C(1200, 144, 288),)
    if name == "unit_left":
        return (_unit_fields()[0],)
    if name == "unit_both":
        return _unit_fields()
    if name == "wide_left":
        return (_wide()[0],)
    if name == "wide_both":
        return _wide()
    if name == "wide_both_f32":
        return _wide(np.float32)
    if name == "small_both":
        return _small()
    if name == "mixed_both":
        return _mixed()
    if name == "sst_prcp":
        fx = np.load(os.path.join(GOLDEN_DIR, "reference_fixtures.npz"))
        return fx["sst"], fx["prcp"]
    if name.startswith("loadings_"):
        tag = name[len("loadings_"):]
        if tag == "noconv":
            # complex white noise: the reference needs 2760 iterations (> maxIter=1000)
            rng = np.random.default_rng(1)
            return (rng.standard_normal((1000, 20)) + 1j * rng.standard_normal((1000, 20)),)
        return (_loadings(tag),)
    raise KeyError(name)
