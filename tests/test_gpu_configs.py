"""The BASELINE.json configurations through `xmca_amd.array.MCA` on the GPU against golden vectors produced by the REAL
reference at those sizes (oracle/make_config_goldens.py -> tests/golden/config_cases.npz):

  c2_full     configs[1] at FULL size: EOF T = 2920 x N = 10 000 float64, solve() + rotate(10, 1)       (530 iterations)
  c3_reduced  configs[2] at T = 1000 x (4000, 3000): complexify=True, rotate(20, 4)                       (413 iterations)
  c5_scaled   configs[4] at T = 1200 x 41 472 float32, 3-D input: EOF, solve() + rotate(10, 1)            (20 iterations)

plus the T x T eigenproblems of the full-size configurations (n = 2920 real: C2; n = 2501 complex: the analytic-signal
subspace of C3, T/2 + 1) against LAPACK on the host.

Tolerances: 1e-5 relative (north_star) on singular values, phase-aligned loadings, R, Phi, norms, variance, PCs and an
EQUAL Varimax iteration count for float64 input; float32 input at the tolerances of tests/test_gpu_solve.py (the
reference itself runs sgesdd there: 2e-5 on sigma, 1e-3 on vectors - its own test tolerance - and +-3 iterations).
"""
import os

import numpy as np
import pytest

from conftest import align_modes
from golden_inputs import GOLDEN_DIR, make_input
from xmca_amd.array import MCA

pytestmark = pytest.mark.gpu

CONFIGS = [("c2_full", False, 10, 1), ("c3_reduced", True, 20, 4), ("c5_scaled", False, 10, 1)]


def _rel(a, b):
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLDEN_DIR, "config_cases.npz"))


@pytest.mark.parametrize("preprocess", ["host", "device"])
@pytest.mark.parametrize("name,cplx,n_rot,power", CONFIGS)
def test_config_matches_reference(gold, name, cplx, n_rot, power, preprocess):
    _check_config(gold, name, cplx, n_rot, power, preprocess)


def test_c3_at_full_size_matches_reference():
    """configs[2] at FULL size - T = 5000 x (20 000, 15 000) float64, complexify=True, rotate(20, 4) - against the real
    reference's output (oracle/make_config_goldens.py c3_full: 275 s of CPU; loadings stored at every 8th grid point, PCs at
    every 5th time step): all 5000 singular values, the 20 leading loadings of both fields, R, Phi, norms, variance, PCs
    and the Varimax iteration count."""
    gold = np.load(os.path.join(GOLDEN_DIR, "config_c3_full.npz"))
    m = _check_config(gold, "c3_full", True, 20, 4, "device")
    # every non-null vector against the 20 leading ones and against a sample of the noise floor (2300 modes at 3e-5..1e-4
    # sigma_1): weak rows are formed by cancelling products and had |u_18^H u_2499| = 7e-4 before Solver::project_out_rows
    sel = np.r_[0:20, 20:2500:31, 2490:2500]
    for key in m._keys:
        V = np.asarray(m._V[key][:, :2500])
        G = V[:, sel].conj().T @ V
        assert np.max(np.abs(G - np.eye(2500)[sel])) < 1e-6, key
    info = m._device().solve_info()
    if os.environ.get("XMCA_CHOLESKY_FACTOR", "1") != "0":
        assert info[0]["sweeps"] == 0 and info[2]["sweeps"] > 0      # no eigen-decomposition of the first field (Cholesky factor)


def test_real_model_at_c3_size_matches_reference():
    """The fields of configs[2] WITHOUT complexify - the general two-field route (Cholesky factor of G_a in time space, weak
    block, per-mode consistency check, no second solve) at T = 5000 x (20 000, 15 000) against the real reference
    (oracle/make_config_goldens.py c3_real_full): all 4999 non-null singular values, leading loadings, rotate(20, 2)."""
    gold = np.load(os.path.join(GOLDEN_DIR, "config_c3_real_full.npz"))
    m = _check_config(gold, "c3_real_full", False, 20, 2, "device")
    sel = np.r_[0:20, 20:4999:61, 4989:4999]
    for key in m._keys:
        V = np.asarray(m._V[key][:, :4999])
        G = V[:, sel].T @ V
        assert np.max(np.abs(G - np.eye(4999)[sel])) < 1e-6, key
    if os.environ.get("XMCA_CHOLESKY_FACTOR", "1") != "0":
        assert "deflate" not in m._device().timings()             # every mode passed the consistency check: one solve


def _check_config(gold, name, cplx, n_rot, power, preprocess):
    g = {k[len(name) + 2:]: gold[k] for k in gold.files if k.startswith(name + "__")}
    fields = make_input(name)
    f32 = fields[0].dtype == np.float32
    m = MCA(*fields, preprocess=preprocess)
    m.solve(complexify=cplx)
    gs = g["singular_values"]
    s = m._singular_values.astype(np.float64)
    assert m._analysis["rank"] == int(g["rank"]) and s.shape == gs.shape
    # ---- singular values: every mode that is not numerically null ----
    if f32:
        assert np.max(np.abs(s - gs)) < 1e-5 * gs[0]
        keep = gs > 1e-2 * gs[0]
        assert _rel(s[keep], gs[keep]) < 2e-5 and np.max(np.abs(s[keep] - gs[keep]) / gs[keep]) < 2e-5
        assert abs(s.sum() - float(g["total_covariance"])) < 1e-4 * gs.sum()
    else:
        keep = gs > 1e-6 * gs[0]
        assert np.max(np.abs(s[keep] - gs[keep]) / gs[keep]) < 1e-5
        assert abs(s.sum() - float(g["total_covariance"])) < 1e-8 * gs.sum()
    # ---- leading loadings, phase aligned (stored every `stride`-th grid point) ----
    stride = int(g["stride"])
    phases = None
    for key in m._keys:
        gv = g["V_" + key]
        nv = gv.shape[1]
        mine, ph = align_modes(m._V.head(key, nv)[::stride], gv)
        if phases is None:
            phases = ph
        err = np.max(np.abs(mine - gv), axis=0) / np.max(np.abs(gv), axis=0)
        assert np.all(err < (1e-3 if f32 else 1e-5)), (key, err)
    # ---- rotation ----
    m.rotate(n_rot, power)
    if f32:
        assert abs(m._varimax_iterations - int(g["n_iter"])) <= 3
    else:
        assert m._varimax_iterations == int(g["n_iter"])
    t = 1e-3 if f32 else 1e-5
    D = np.diag(phases[:n_rot])
    assert _rel(D @ m._rotation_matrix @ D.conj().T, g["R"]) < t
    assert _rel(D @ m._correlation_matrix @ D.conj().T, g["Phi"]) < t
    assert _rel(m._variance, g["variance"]) < t
    assert np.array_equal(m._var_idx, g["var_idx"])
    assert _rel(m.explained_variance(), g["explained_variance"]) < t
    pcs = m.pcs(n_rot)
    for key in m._keys:
        assert _rel(m._norm[key], g["norm_" + key]) < t
        al, _ = align_modes(pcs[key][::int(g["pcs_stride"]) if "pcs_stride" in g else 1], g["pcs_" + key])
        assert _rel(al, g["pcs_" + key]) < (5e-3 if f32 else 1e-4)
    # ---- properties over ALL grid points (the goldens hold a subset for c5) ----
    # (two fields: sigma^2 = lambda(K K^H) - orthogonality of weak modes degrades like 5e-14 (sigma_1 / sigma_m)^2)
    V = m._V.head("left", n_rot)
    assert np.max(np.abs(V.conj().T @ V - np.eye(n_rot))) < (1e-4 if f32 else 1e-6 if len(fields) == 2 else 1e-9)
    return m


@pytest.mark.parametrize("n,cplx", [(2920, False), (2501, True)])
def test_eigh_at_config_size_matches_lapack(hip, n, cplx):
    """The T x T eigenproblems of C2 (n = 2920, real) and of C3's analytic-signal subspace (n = T/2 + 1 = 2501, complex
    Hermitian) at full size: all eigenvalues against numpy.linalg.eigvalsh, orthonormal vectors, residuals."""
    rng = np.random.default_rng(n)
    N = 3 * n
    X = (rng.standard_normal((n, 20)) * np.linspace(10, 1, 20)) @ rng.standard_normal((20, N)) + rng.standard_normal((n, N))
    if cplx:
        X = X + 1j * ((rng.standard_normal((n, 20)) * np.linspace(6, 1, 20)) @ rng.standard_normal((20, N)) + rng.standard_normal((n, N)))
    X -= X.mean(axis=0)
    G = X @ X.conj().T
    lam, U = hip.eigh(G)
    ref = np.linalg.eigvalsh(G)[::-1]
    assert np.max(np.abs(lam - ref)) < 1e-11 * ref[0], hip.last_eigh_info
    assert hip.last_eigh_info["sweeps"] <= 20 and hip.last_eigh_info["slots"] == -(-n // hip.last_eigh_info["tile"])
    sel = np.r_[0:40, n // 2:n // 2 + 40, n - 40:n]                   # leading, bulk and trailing (null: centered) vectors
    Us = U[:, sel]
    assert np.max(np.abs(Us.conj().T @ U - np.eye(n)[sel])) < 1e-11
    res = np.linalg.norm(G @ Us - Us * lam[sel], axis=0)
    assert np.max(res) < 1e-10 * ref[0]


# ----------------------------------------------------------------------------------------------
# bootstrapping against the REFERENCE's numbers (oracle/make_config_goldens.py, np.random.seed(5))
# ----------------------------------------------------------------------------------------------
BOOT = [("small_std", "small_both", False, False, None, dict(on_left=True, on_right=True, block_size=2)),
        ("wide_rot", "wide_both", False, False, (5, 2), dict(on_left=True, on_right=False, block_size=1)),
        ("wide_single_cplx", "wide_both", True, True, None, dict(on_left=True, on_right=False, block_size=4, replace=False)),
        ("wide_cplx_rot", "wide_both", False, True, (4, 1), dict(on_left=False, on_right=True, block_size=1)),
        ("sst_iterative", "sst_prcp", False, False, None, dict(on_left=True, on_right=True, block_size=3, strategy='iterative'))]


@pytest.mark.parametrize("tag,inp,single,cplx,rot,kw", BOOT)
def test_bootstrapping_matches_reference(tag, inp, single, cplx, rot, kw):
    """`xmca.array.MCA.bootstrapping(3, n_modes=4, ...)` of the real reference under np.random.seed(5) (xmca/array.py:
    1813-1952, tools/array.py:91-138): the device replicates draw the same blocks from the same global stream."""
    ref = np.load(os.path.join(GOLDEN_DIR, "bootstrap_cases.npz"))[tag]
    fields = make_input(inp)
    if single:
        fields = fields[:1]
    m = MCA(*fields)
    m.solve(complexify=cplx)
    if rot:
        m.rotate(*rot)
    np.random.seed(5)
    out = m.bootstrapping(3, n_modes=4, **kw)
    assert out.shape == ref.shape
    tol = 1e-4 if fields[0].dtype == np.float32 else 1e-5
    assert np.max(np.abs(out - ref) / np.abs(ref)) < tol
