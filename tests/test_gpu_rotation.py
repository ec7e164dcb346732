"""Varimax / Promax on the device vs. golden vectors generated from the reference's
xmca/tools/rotation.py (tests/golden/rotation_direct.npz).  Same input loadings -> no gauge freedom:
R, Phi, B and the ITERATION COUNT must match (tolerance 1e-7 relative, written below)."""
import os

import numpy as np
import pytest

from golden_inputs import GOLDEN_DIR, make_input

pytestmark = pytest.mark.gpu
TOL = 1e-7


def _rel(a, b):
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLDEN_DIR, "rotation_direct.npz"))


@pytest.mark.parametrize("tag", ["r4", "r10", "r10p4", "c4", "c10p4", "c10p2"])
def test_promax_matches_reference(hip, gold, tag):
    A = make_input("loadings_" + tag)[0]
    power = int(gold[tag + "_power"])
    out = hip.rotate_loadings(A, n_left=A.shape[0], power=power, tol=1e-8, want_B=True)
    assert out["n_iter"] == int(gold[tag + "_n_iter"])
    assert _rel(out["R"], gold[tag + "_R"]) < TOL
    assert _rel(out["Phi"], gold[tag + "_Phi"]) < TOL
    assert _rel(out["B"], gold[tag + "_B"]) < TOL
    # column norms of the rotated loadings (what MCA.rotate keeps, array.py:827)
    assert _rel(out["norm_left"], np.linalg.norm(gold[tag + "_B"], axis=0)) < TOL


@pytest.mark.parametrize("tag", ["r4", "r10", "c4"])
def test_varimax_matches_reference(hip, gold, tag):
    A = make_input("loadings_" + tag)[0]
    out = hip.rotate_loadings(A, n_left=A.shape[0], varimax_only=True, want_B=True)
    assert _rel(out["R"], gold[tag + "_Rv"]) < TOL
    assert _rel(out["B"], gold[tag + "_Bv"]) < TOL


def test_split_norms(hip, gold):
    A = make_input("loadings_r10p4")[0]
    out = hip.rotate_loadings(A, n_left=120, power=4)
    B = gold["r10p4_B"]
    assert _rel(out["norm_left"], np.linalg.norm(B[:120], axis=0)) < TOL
    assert _rel(out["norm_right"], np.linalg.norm(B[120:], axis=0)) < TOL


def test_not_converged_raises_runtime_error(hip):
    A = make_input("loadings_noconv")[0]
    with pytest.raises(RuntimeError):
        hip.rotate_loadings(A, n_left=A.shape[0], power=4)
    assert hip.last_iters == 1000


def test_argument_errors(hip):
    A = make_input("loadings_r4")[0]
    with pytest.raises(ValueError):
        hip.rotate_loadings(A, n_left=10, power=0)
    with pytest.raises(ValueError):
        hip.rotate_loadings(A[:, :1], n_left=10)


def test_zero_row_is_linalg_error(hip):
    A = make_input("loadings_r4")[0].copy()
    A[7] = 0.0           # Kaiser normalisation divides by the row norm -> NaN, as in the reference
    with pytest.raises(np.linalg.LinAlgError):
        hip.rotate_loadings(A, n_left=A.shape[0])


def _wide_loadings(n, p, cplx, seed):
    rng = np.random.default_rng(seed)
    L = 0.15 * rng.standard_normal((n, p))
    w = n // p
    for j in range(p):
        L[j * w:(j + 1) * w, j] += np.hanning(w) * (3.0 - 0.05 * j)
    if cplx:
        L = L * np.exp(1j * rng.uniform(0, 2 * np.pi, (n, 1)) * 0.3) + 0.05j * rng.standard_normal((n, p))
        M = rng.standard_normal((p, p)) + 1j * rng.standard_normal((p, p))
    else:
        M = rng.standard_normal((p, p))
    Q, _ = np.linalg.qr(M)
    return L @ Q


@pytest.mark.parametrize("n,p,cplx,seed", [(900, 17, False, 55), (600, 20, False, 51), (480, 24, True, 52),
                                           (640, 32, True, 54), (800, 40, False, 53), (20000, 12, True, 56),
                                           # every k-step count of the unrolled Newton-Schulz tiles (rotate.h ns_tile: 5 .. 16)
                                           (500, 28, False, 57), (500, 36, True, 58), (600, 44, True, 59), (600, 48, True, 60),
                                           (700, 52, False, 61), (700, 56, False, 62), (700, 60, False, 63), (800, 64, False, 64)])
def test_varimax_many_modes_matches_oracle(hip, n, p, cplx, seed):
    """More than 16 rotated modes take the multi-tile MFMA accumulation and the LDS Newton-Schulz path, many grid
    points take several tiles per workgroup: same R, B and iteration count as the numpy restatement of rotation.py."""
    from oracle import ref_numpy as O
    A = _wide_loadings(n, p, cplx, seed)
    B_ref, R_ref, n_iter = O.varimax(A)
    out = hip.rotate_loadings(A, n_left=n, varimax_only=True, want_B=True)
    assert out["n_iter"] == n_iter
    assert _rel(out["R"], R_ref) < TOL
    assert _rel(out["B"], B_ref) < TOL


@pytest.mark.parametrize("n,p,cplx,seed", [(20000, 20, True, 71), (30000, 24, False, 72), (12000, 32, True, 73)])
def test_varimax_wide_grid_and_two_stage_sum(hip, monkeypatch, n, p, cplx, seed):
    """16 < p <= 32 on many grid points: the wide grid (up to one workgroup per CU), the unrolled accumulation, the
    padded Newton-Schulz tiles and the two-stage sum of the partial G matrices (each entry added up by one workgroup, in a
    fixed order).  Against the numpy restatement of rotation.py, and the all-to-all sum must stop at the same iteration."""
    from oracle import ref_numpy as O
    A = _wide_loadings(n, p, cplx, seed)
    B_ref, R_ref, n_iter = O.varimax(A)
    outs = {}
    for form in ("1", "0"):
        monkeypatch.setenv("XMCA_ROT_TWO_STAGE", form)
        outs[form] = hip.rotate_loadings(A, n_left=n // 2, varimax_only=True, want_B=True)
        assert outs[form]["n_iter"] == n_iter
        assert _rel(outs[form]["R"], R_ref) < TOL
        assert _rel(outs[form]["B"], B_ref) < TOL
    assert _rel(outs["1"]["R"], outs["0"]["R"]) < 1e-12
    monkeypatch.delenv("XMCA_ROT_TWO_STAGE")
    again = hip.rotate_loadings(A, n_left=n // 2, varimax_only=True, want_B=True)
    assert np.array_equal(again["R"], outs["1"]["R"])            # (default = two stages here; fixed summation order: same bits)


@pytest.mark.parametrize("tag,gamma", [("r10", 0.0), ("r10", 0.5), ("c4", 0.0), ("c10p4", 0.3), ("r4", 1.0)])
def test_varimax_gamma_family_matches_oracle(hip, tag, gamma):
    """`varimax(A, gamma)` (rotation.py:15, :56-57): gamma = 1 Varimax, 0 Quartimax, anything in between - same
    trajectory, same stop iteration as the numpy restatement."""
    from oracle import ref_numpy as O
    from xmca_amd.tools.rotation import varimax
    A = make_input("loadings_" + tag)[0]
    Bo, Ro, n_iter = O.varimax(A, gamma=gamma)
    B, R = varimax(A, gamma=gamma, handle=hip)
    assert hip.last_iters == n_iter
    assert _rel(R, Ro) < TOL and _rel(B, Bo) < TOL


@pytest.mark.parametrize("n,p,cplx,power", [(900, 70, False, 1), (900, 70, False, 2), (700, 52, True, 1), (700, 50, True, 3), (1500, 130, False, 1)])
def test_more_modes_than_the_fused_kernels_hold(hip, n, p, cplx, power):
    """The reference has no limit on n_rot.  Beyond 64 real / 48 complex modes the fused single-workgroup kernels do not
    fit their LDS image; the GEMM-based path (Rotator::run_generic) takes over: same trajectory, same stop iteration,
    R / Phi / B against the numpy restatement of rotation.py:15-149."""
    from oracle import ref_numpy as O
    rng = np.random.default_rng(100 + p)
    L = 0.12 * rng.standard_normal((n, p))
    w = n // p
    for j in range(p):
        L[j * w:(j + 1) * w, j] += np.hanning(w) * (3.0 - 1.5 * j / p)
    if cplx:
        L = L * np.exp(1j * rng.uniform(0, 2 * np.pi, (n, 1)) * 0.3) + 0.04j * rng.standard_normal((n, p))
        M = rng.standard_normal((p, p)) + 1j * rng.standard_normal((p, p))
    else:
        M = rng.standard_normal((p, p))
    Q, _ = np.linalg.qr(M)
    A = L @ Q
    Bo, Ro, Phio, n_iter = O.promax(A, power)
    out = hip.rotate_loadings(A, n_left=n // 3, power=power, tol=1e-8, want_B=True)
    assert out["n_iter"] == n_iter
    assert _rel(out["R"], Ro) < 1e-6 and _rel(out["Phi"], Phio) < 1e-6 and _rel(out["B"], Bo) < 1e-6
    assert _rel(out["norm_left"], np.linalg.norm(Bo[:n // 3], axis=0)) < 1e-6
    assert _rel(out["norm_right"], np.linalg.norm(Bo[n // 3:], axis=0)) < 1e-6
