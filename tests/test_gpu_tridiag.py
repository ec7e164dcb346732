"""The second eigensolver of the library: Householder reduction to tridiagonal form + Sturm multisection (+ twisted
factorisation vectors, blocked back-transformation, Newton-Schulz clean-up) - csrc/tridiag.h, csrc/tridiag_vec.h - which
replaces LAPACK's *gesdd / eigh on the T x T stage of xmca/array.py:479 and :570 for eigenproblems of 192 (values only) /
768 (with vectors) and more.  Everything against numpy.linalg.eigh / eigvalsh on the same matrices:

* both forms of the reduction - one launch per column, and the persistent kernel with the matrix resident in registers
  (its column exchange by tagged values and by epoch flags) - real and complex, sizes around the chunk / row-slot boundaries of the kernels;
* spectra the twisted vectors cannot resolve (repeated eigenvalues, wide null spaces) come back through the Jacobi sweeps;
* bit-reproducibility (every sum has a fixed order) and NaN input -> LinAlgError like gesdd.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _gram(n, cplx, seed=None, spikes=20):
    rng = np.random.default_rng(n if seed is None else seed)
    N = 3 * n
    X = (rng.standard_normal((n, spikes)) * np.linspace(10, 1, spikes)) @ rng.standard_normal((spikes, N)) + rng.standard_normal((n, N))
    if cplx:
        X = X + 1j * rng.standard_normal((n, N))
    X -= X.mean(axis=0)
    return X @ X.conj().T


def _reduction_form(monkeypatch, form):
    """"0": one launch per column; "tagged" / "flags": the persistent kernel with either form of its column exchange"""
    monkeypatch.setenv("XMCA_TRD_RESIDENT", "0" if form == "0" else "1")
    if form != "0":
        monkeypatch.setenv("XMCA_TRD_TAGGED", "1" if form == "tagged" else "0")


@pytest.mark.parametrize("resident", ["tagged", "flags", "0"])
@pytest.mark.parametrize("n,cplx", [(2, False), (3, True), (64, False), (127, True), (129, False), (385, False), (512, True), (1000, False),
                                    (1025, True), (1100, False), (2049, False), (2100, True)])
def test_eigenvalues_only_match_lapack(hip, monkeypatch, n, cplx, resident):
    monkeypatch.setenv("XMCA_TRIDIAG_MIN_N", "2")
    _reduction_form(monkeypatch, resident)
    monkeypatch.setenv("XMCA_TRD_RESIDENT_MIN_N", "2")
    G = _gram(n, cplx)
    lam, U = hip.eigh(G, vectors=False)
    assert U is None and hip.last_eigh_info["tridiag"] == 1 and hip.last_eigh_info["sweeps"] == 0
    ref = np.linalg.eigvalsh(G)[::-1]
    assert np.max(np.abs(lam - ref)) < 1e-13 * ref[0]
    again, _ = hip.eigh(G, vectors=False)
    assert np.array_equal(lam, again)                       # fixed summation order: the same bits


@pytest.mark.parametrize("n,cplx", [(130, False), (193, True), (514, False), (577, True), (1000, False), (1025, True), (1100, False)])
def test_back_transformation_with_short_last_blocks(hip, monkeypatch, n, cplx):
    """Super-blocks of 512 reflectors whose compact-WY factors are built ahead of the loop (64 x 64 inverses, then the
    levels 64 -> 128 -> 256 -> 512 as block-sparse GEMM launches); the last super-block is shorter - a single reflector
    (n = 514), less than one block tile, exactly full (n = 1025) - and levels whose second half is empty: a full orthonormal
    decomposition all the same.  The second call of a workspace builds the factors on the eigensolver's second stream: the
    same bits."""
    monkeypatch.setenv("XMCA_TRIDIAG_VEC_MIN_N", "2")
    G = _gram(n, cplx)
    lam, U = hip.eigh(G)
    assert hip.last_eigh_info["tridiag"] == 1
    ref = np.linalg.eigvalsh(G)[::-1]
    assert np.max(np.abs(lam - ref)) < 1e-13 * ref[0]
    assert np.max(np.abs(U.conj().T @ U - np.eye(n))) < 1e-12
    assert np.max(np.linalg.norm(G @ U - U * lam, axis=0)) < 1e-12 * ref[0]
    for _ in range(2):
        lam2, U2 = hip.eigh(G)
        assert hip.last_eigh_info["tridiag"] == 1
        assert np.array_equal(lam, lam2) and np.array_equal(U, U2)


@pytest.mark.parametrize("scale", [1e6, 1.0, 1e-6])
@pytest.mark.parametrize("cplx", [False, True])
def test_indefinite_matrix_with_zero_diagonal(hip, monkeypatch, cplx, scale):
    """xmca_eigh is a general Hermitian solver: an indefinite matrix whose diagonal is exactly zero and whose off-diagonal
    entries are large or tiny.  The working copy is scaled by max |a_ij| (the diagonal alone gave f = 1 here, and the
    rescaling interval of the Sturm count and the pivot floors of the twisted factorisation assume entries below 1)."""
    monkeypatch.setenv("XMCA_TRIDIAG_VEC_MIN_N", "2")
    monkeypatch.setenv("XMCA_TRIDIAG_MIN_N", "2")
    rng = np.random.default_rng(77)
    n = 400
    B = rng.standard_normal((n, n))
    if cplx:
        B = B + 1j * rng.standard_normal((n, n))
    A = (B + B.conj().T) * scale
    np.fill_diagonal(A, 0.0)
    ref = np.linalg.eigvalsh(A)[::-1]
    nrm = np.max(np.abs(ref))
    lam, _ = hip.eigh(A, vectors=False)
    assert hip.last_eigh_info["tridiag"] == 1
    assert np.max(np.abs(lam - ref)) < 1e-13 * nrm
    lam, U = hip.eigh(A)
    assert np.max(np.abs(lam - ref)) < 1e-13 * nrm
    assert np.max(np.abs(U.conj().T @ U - np.eye(n))) < 1e-12
    assert np.max(np.linalg.norm(A @ U - U * lam, axis=0)) < 1e-12 * nrm


@pytest.mark.parametrize("resident", ["tagged", "flags", "0"])
@pytest.mark.parametrize("n,cplx", [(70, False), (200, True), (777, False), (1000, True), (1500, False)])
def test_eigenvectors_match_lapack(hip, monkeypatch, n, cplx, resident):
    monkeypatch.setenv("XMCA_TRIDIAG_VEC_MIN_N", "2")
    _reduction_form(monkeypatch, resident)
    monkeypatch.setenv("XMCA_TRD_RESIDENT_MIN_N", "2")
    G = _gram(n, cplx)
    lam, U = hip.eigh(G)
    assert hip.last_eigh_info["tridiag"] == 1
    ref, Uref = np.linalg.eigh(G)
    ref, Uref = ref[::-1], Uref[:, ::-1]
    assert np.max(np.abs(lam - ref)) < 1e-13 * ref[0]
    assert np.max(np.abs(U.conj().T @ U - np.eye(n))) < 1e-12
    assert np.max(np.linalg.norm(G @ U - U * lam, axis=0)) < 1e-12 * ref[0]
    # the well separated leading vectors up to phase
    ov = np.abs(np.sum(U[:, :10].conj() * Uref[:, :10], axis=0))
    assert np.all(np.abs(ov - 1) < 1e-9)


@pytest.mark.parametrize("kind", ["graded12", "lowrank", "identity", "diagonal", "two_clusters", "tridiagonal_input"])
def test_hard_spectra_with_vectors(hip, monkeypatch, kind):
    """graded / already diagonal / already tridiagonal matrices stay on the tridiagonal route; repeated eigenvalues and wide
    null spaces - where twisted-factorisation vectors of one cluster are not independent - must be detected
    (max |Z^H Z - I| > 0.3) and come back through the Jacobi sweeps.  Either way the result is a full decomposition."""
    monkeypatch.setenv("XMCA_TRIDIAG_VEC_MIN_N", "2")
    n = 600
    rng = np.random.default_rng(5)
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    if kind == "graded12":
        G = (Q * np.logspace(0, -12, n)) @ Q.T
    elif kind == "lowrank":
        Y = rng.standard_normal((n, 20))
        G = Y @ Y.T
    elif kind == "identity":
        G = np.eye(n)
    elif kind == "diagonal":
        G = np.diag(np.arange(1.0, n + 1))
    elif kind == "two_clusters":
        G = (Q * np.r_[np.full(n // 2, 2.0), np.full(n - n // 2, 1.0)]) @ Q.T
    else:
        G = np.diag(rng.standard_normal(n)) + np.diag(rng.standard_normal(n - 1), 1)
        G = G + G.T
    G = (G + G.T) / 2
    lam, U = hip.eigh(G)
    ref = np.linalg.eigvalsh(G)[::-1]
    sc = np.max(np.abs(ref))
    assert np.max(np.abs(lam - ref)) < 1e-12 * sc
    assert np.max(np.abs(U.T @ U - np.eye(n))) < 1e-10
    assert np.max(np.linalg.norm(G @ U - U * lam, axis=0)) < 1e-10 * sc
    if kind in ("lowrank", "identity", "two_clusters"):
        assert hip.last_eigh_info["tridiag"] == 0           # handed to the Jacobi solver
    else:
        assert hip.last_eigh_info["tridiag"] == 1


def _glued_wilkinson(blocks, glue):
    """`blocks` copies of the Wilkinson matrix W21+ (diagonal |i - 10|, off-diagonal 1) glued by off-diagonal entries
    `glue`: its large eigenvalues come in pairs that agree to ~glue^2 / gap or much closer (the classic stress test of
    inverse-iteration / MRRR eigenvectors; Dhillon, Parlett & Voemel 2005)."""
    n = 21 * blocks
    d = np.tile(np.abs(np.arange(21) - 10.0), blocks)
    e = np.ones(n - 1)
    e[20::21] = glue
    return np.diag(d) + np.diag(e, 1) + np.diag(e, -1)


@pytest.mark.parametrize("kind", ["wilkinson_1e-5", "wilkinson_1e-7", "pairs_1e-10", "pairs_1e-12", "pairs_1e-14", "mixed_gaps"])
def test_near_degenerate_spectra_with_vectors(hip, monkeypatch, kind):
    """Between `well separated` (twisted vectors + one Newton-Schulz step) and `repeated` (max |Z^H Z - I| > 0.3 -> Jacobi) lie
    eigenvalue pairs with gaps of 1e-10 ... 1e-14 ||T||: twisted vectors of such a pair are nearly parallel, the clean-up
    either repairs them or the solver must hand over.  Whichever route ends up handling the matrix (both are accepted), the
    result has to be a full orthonormal decomposition - checked here at 1e-10, with the eigenvalues at 1e-12 ||A||."""
    monkeypatch.setenv("XMCA_TRIDIAG_VEC_MIN_N", "2")
    rng = np.random.default_rng(11)
    if kind.startswith("wilkinson"):
        T = _glued_wilkinson(30, float(kind.split("_")[1]))                 # n = 630, pairs glued at 1e-5 / 1e-7
        n = T.shape[0]
        Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
        G = Q @ T @ Q.T                                                      # dense: the reduction has to find the structure again
    else:
        n = 640
        lam = np.sort(rng.uniform(1.0, 2.0, n // 2))[::-1]
        if kind == "mixed_gaps":
            gaps = 10.0 ** rng.uniform(-15, -6, n // 2)
        else:
            gaps = np.full(n // 2, float(kind.split("_")[1]))
        lam = np.r_[lam, lam * (1.0 - gaps)]                                 # every eigenvalue has a partner at relative distance `gap`
        Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
        G = (Q * lam) @ Q.T
    G = (G + G.T) / 2
    lam_d, U = hip.eigh(G)
    ref = np.linalg.eigvalsh(G)[::-1]
    sc = np.max(np.abs(ref))
    assert np.max(np.abs(lam_d - ref)) < 1e-12 * sc
    assert np.max(np.abs(U.T @ U - np.eye(n))) < 1e-10, hip.last_eigh_info
    assert np.max(np.linalg.norm(G @ U - U * lam_d, axis=0)) < 1e-10 * sc
    # values only: the same eigenvalues without the vector machinery
    monkeypatch.setenv("XMCA_TRIDIAG_MIN_N", "2")
    lam_v, _ = hip.eigh(G, vectors=False)
    assert np.max(np.abs(lam_v - ref)) < 1e-12 * sc


def test_nan_input_raises_like_gesdd(hip, monkeypatch):
    monkeypatch.setenv("XMCA_TRIDIAG_MIN_N", "2")
    G = _gram(300, False)
    G[17, 40] = G[40, 17] = np.nan
    with pytest.raises(np.linalg.LinAlgError):
        hip.eigh(G, vectors=False)
    with pytest.raises(np.linalg.LinAlgError):
        hip.eigh(G)


def test_values_of_a_model_do_not_depend_on_the_eigensolver(monkeypatch):
    """MCA.rule_n without rotation (xmca/array.py:1753-1765) takes eigenvalues only: the tridiagonal route and the Jacobi
    sweeps must give the same spectra to rounding."""
    from golden_inputs import make_input
    from xmca_amd.array import MCA
    m = MCA(*make_input("c1_standin"))
    m.solve()
    monkeypatch.setenv("XMCA_TRIDIAG", "1")
    a = m.rule_n(3, seed=11)
    monkeypatch.setenv("XMCA_TRIDIAG", "0")
    b = m.rule_n(3, seed=11)
    keep = b > 1e-9 * b[0]
    assert a.shape == b.shape and np.max(np.abs(a[keep] - b[keep]) / b[keep]) < 1e-9


@pytest.mark.parametrize("n,cplx", [(700, False), (1300, True), (1500, False), (2501, True), (2920, False)])
def test_chain_of_repacked_launches_gives_the_bits_of_one_launch(hip, monkeypatch, n, cplx):
    """Round 6: the persistent reduction is handed from instantiation to instantiation as the trailing block shrinks (tridiag.h
    `trd_plan`: cuts at multiples of 256 columns, so every sum keeps its order).  The tridiagonal matrix - hence every eigenvalue -
    must not depend on where the chain is cut: default cuts, one launch, and two hand-picked cuts give identical bits, with and
    without reflectors kept (the vectors' route); against LAPACK as before."""
    G = _gram(n, cplx, spikes=8)
    monkeypatch.setenv("XMCA_TRD_CHAIN", "1")
    lam, _ = hip.eigh(G, vectors=False)
    monkeypatch.setenv("XMCA_TRD_CHAIN", "0")
    one, _ = hip.eigh(G, vectors=False)
    monkeypatch.setenv("XMCA_TRD_CHAIN", "1")
    monkeypatch.setenv("XMCA_TRD_BREAKS", "256,768" if n > 1100 else "256")
    cut, _ = hip.eigh(G, vectors=False)
    assert np.array_equal(lam, one) and np.array_equal(lam, cut)
    ref = np.linalg.eigvalsh(G)[::-1]
    assert np.max(np.abs(lam - ref)) < 1e-13 * ref[0]
    if n <= 1500:
        monkeypatch.delenv("XMCA_TRD_BREAKS")
        monkeypatch.setenv("XMCA_TRIDIAG_VEC_MIN_N", "2")
        lam_v, U = hip.eigh(G)
        assert hip.last_eigh_info["tridiag"] == 1
        assert np.max(np.abs(lam_v - ref)) < 1e-13 * ref[0]
        assert np.max(np.abs(U.conj().T @ U - np.eye(n))) < 1e-12
        assert np.max(np.abs(G @ U - U * lam_v)) < 1e-11 * ref[0]


@pytest.mark.parametrize("n,cplx", [(513, False), (513, True), (1025, True), (1537, False), (2049, True), (2305, False), (2560, True),
                                    (2561, False), (3072, False)])
def test_chain_at_the_instantiation_boundaries(hip, monkeypatch, n, cplx):
    """Orders just above / at every capacity of the chain's instantiations (512, 1024, ... columns; complex 2560 and real 3072 are
    the largest resident orders): the number of links follows the order, the eigenvalues are those of one launch bit for bit and
    LAPACK's to 1e-13 (scripts/chain_boundary_sweep.py runs the full list)."""
    rng = np.random.default_rng(n + cplx)
    X = rng.standard_normal((n, n + 50))
    if cplx:
        X = X + 1j * rng.standard_normal((n, n + 50))
    G = X @ X.conj().T
    monkeypatch.setenv("XMCA_TRD_CHAIN", "1")
    lam, _ = hip.eigh(G, vectors=False)
    links = hip.reduction_info().count("trd_resident_kernel<")
    assert links == -(-n // 512), hip.reduction_info()
    monkeypatch.setenv("XMCA_TRD_CHAIN", "0")
    one, _ = hip.eigh(G, vectors=False)
    assert hip.reduction_info().count("trd_resident_kernel<") == 1
    assert np.array_equal(lam, one)
    ref = np.linalg.eigvalsh(G)[::-1]
    assert np.max(np.abs(lam - ref)) < 1e-13 * ref[0]
