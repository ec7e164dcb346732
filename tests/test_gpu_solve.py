"""solve() numerical core on the device (through the C ABI) vs golden vectors produced by the reference
(tests/golden/solve_cases.npz) and vs the numpy oracle on the same inputs.

Tolerances (north_star): 1e-5 relative on singular values and on sign/phase-aligned loadings of the
well-separated leading modes; float32 inputs are compared at the reference's own 1e-3 (test_integration_xarray.py:33-35)
for the vectors and 2e-5 for the singular values (the reference itself runs sgesdd there).
"""
import os

import numpy as np
import pytest

from conftest import align_modes
from golden_inputs import GOLDEN_DIR, make_input
from oracle import ref_numpy as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLDEN_DIR, "solve_cases.npz"))


def device_solve(hip, fields, complexify, n_vec=-1):
    prepared = [O.flatten_and_center(f)[0] for f in fields]
    for s, f in enumerate(prepared):
        hip.set_field(s, f)
    if complexify:
        hip.complexify(prepared[0].shape[0])
    rank = hip.solve(len(prepared), n_vec)
    sig = hip.singular_values(rank)
    m = rank if n_vec < 0 else min(rank, n_vec)
    V = [hip.vectors(s, m, f.shape[1], f.dtype).T for s, f in enumerate(prepared)]
    return rank, sig, V


CASES = ["unit_left", "unit_both", "wide_left", "wide_both", "mixed_both", "sst_prcp"]


@pytest.mark.parametrize("cplx", [False, True])
@pytest.mark.parametrize("name", CASES)
def test_solve_matches_reference_goldens(hip, gold, name, cplx):
    tag = name + ("_cplx" if cplx else "_std") + "__"
    fields = make_input(name)
    rank, sig, V = device_solve(hip, fields, cplx)
    gs = gold[tag + "singular_values"].astype(np.float64)
    assert rank == int(gold[tag + "rank"])
    f32 = fields[0].dtype == np.float32
    # singular values: every mode that is not numerically null (relative to the largest)
    if f32:
        # the golden itself comes from sgesdd: absolute accuracy ~1e-6 * sigma_1, so small modes are pinned absolutely
        assert np.max(np.abs(sig - gs)) < 1e-5 * gs[0]
        keep = gs > 1e-2 * gs[0]
        assert np.max(np.abs(sig[keep] - gs[keep]) / gs[keep]) < 2e-5
    else:
        keep = gs > 1e-6 * gs[0]
        assert np.max(np.abs(sig[keep] - gs[keep]) / gs[keep]) < 1e-5
    assert abs(sig.sum() - float(gold[tag + "total_covariance"])) < (1e-4 if f32 else 1e-8) * gs.sum()
    # leading vectors, phase aligned; only modes separated from their neighbours by > 2 % are pinned
    for s, key in enumerate(["left", "right"][:len(fields)]):
        gv = gold[tag + "V_" + key]
        m = min(10, gv.shape[1])
        gaps = np.minimum(np.abs(np.diff(gs[:m + 1], prepend=np.inf)), np.abs(np.diff(gs[:m + 1], append=0)))[:m] / gs[:m]
        sel = np.where(gaps > 0.02)[0]
        mine, _ = align_modes(V[s][:, :m], gv[:, :m])
        err = np.max(np.abs(mine[:, sel] - gv[:, sel]), axis=0) / np.max(np.abs(gv[:, sel]), axis=0)
        tol = 1e-3 if f32 else 1e-5 / np.minimum(gaps[sel], 1.0) * 0.1 + 1e-5
        assert np.all(err < tol), (key, err, gaps[sel])


@pytest.mark.parametrize("cplx", [False, True])
def test_f32_fields(hip, gold, cplx):
    tag = "wide_both_f32" + ("_cplx" if cplx else "_std") + "__"
    fields = make_input("wide_both_f32")
    rank, sig, V = device_solve(hip, fields, cplx)
    gs = gold[tag + "singular_values"].astype(np.float64)
    keep = gs > 1e-4 * gs[0]
    assert np.max(np.abs(sig[keep] - gs[keep]) / gs[keep]) < 2e-5
    assert V[0].dtype == (np.complex64 if cplx else np.float32)
    gv = gold[tag + "V_left"]
    mine, _ = align_modes(V[0][:, :5], gv[:, :5])
    assert np.max(np.abs(mine - gv[:, :5])) < 1e-3


def test_orthonormal_vectors_and_gauge(hip):
    """properties the reference tests (test_orthogonality / test_correlation): V^H V = I and U_l^H U_r diagonal, real, positive."""
    fields = make_input("wide_both")
    rank, sig, V = device_solve(hip, fields, True)
    X = [O.analytic_signal(O.flatten_and_center(f)[0]) for f in fields]
    k = 8          # the signal modes; orthogonality of weaker modes degrades like (sigma_1/sigma_m)^2 * 1e-13
    for v in V:
        assert np.max(np.abs(v[:, :k].conj().T @ v[:, :k] - np.eye(k))) < 1e-9
    T = X[0].shape[0]
    cov = (X[0] @ V[0][:, :k]).conj().T @ (X[1] @ V[1][:, :k]) / (T - 1)
    assert np.max(np.abs(cov - np.diag(sig[:k]))) < 1e-9 * sig[0]


def test_partial_backprojection(hip):
    fields = make_input("wide_both")
    rank, sig, V = device_solve(hip, fields, False, n_vec=6)
    rank2, sig2, V2 = device_solve(hip, fields, False)
    assert V[0].shape[1] == 6 and np.allclose(sig, sig2, rtol=1e-12)
    mine, _ = align_modes(V[0], V2[0][:, :6])
    assert np.max(np.abs(mine - V2[0][:, :6])) < 1e-10


def test_errors(hip):
    from xmca_amd import _hip
    h = _hip.Handle(0)
    with pytest.raises(RuntimeError):
        h.solve(1)                                    # no field
    h.set_field(0, np.zeros((10, 4)))
    with pytest.raises(ValueError):
        h.set_field(1, np.zeros((11, 4)))             # different T
    with pytest.raises(RuntimeError):
        h.singular_values(3)                          # before solve
    bad = np.random.default_rng(0).standard_normal((12, 30))
    bad[3, 4] = np.nan
    h.set_field(0, bad)
    with pytest.raises(np.linalg.LinAlgError):        # array.py:575-578
        h.solve(1)
    h.close()


def test_analytic_subspace_path_equals_general_path(hip, monkeypatch):
    """complexify on wide fields uses the T/2-dimensional Fourier subspace (no imaginary plane on the device); with
    XMCA_ANALYTIC=0 the same model goes through X_im = Ht X and T x T eigenproblems: identical sigma and vectors."""
    import subprocess, sys, json, os
    fields = make_input("wide_both")
    rank, sig, V = device_solve(hip, fields, True)
    code = ("import sys, json, numpy as np; sys.path.insert(0, %r); sys.path.insert(0, %r);"
            "from golden_inputs import make_input; from oracle import ref_numpy as O; from xmca_amd import _hip;"
            "h = _hip.Handle(0); f = [O.flatten_and_center(x)[0] for x in make_input('wide_both')];"
            "[h.set_field(s, x) for s, x in enumerate(f)]; h.complexify(f[0].shape[0]); r = h.solve(2);"
            "s = h.singular_values(r); v = h.vectors(0, 8, f[0].shape[1], f[0].dtype);"
            "print(json.dumps({'s': s.tolist(), 'vr': v.real.tolist(), 'vi': v.imag.tolist(), 'info': h.solve_info()}))"
            ) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, XMCA_ANALYTIC="0")
    out = json.loads(subprocess.run([sys.executable, "-c", code], env=env, check=True, capture_output=True, text=True).stdout)
    s_gen = np.array(out["s"])
    v_gen = (np.array(out["vr"]) + 1j * np.array(out["vi"])).T
    T = 64
    m = T // 2 + 1
    assert hip.solve_info()[0]["slots"] <= out["info"][0]["slots"]           # never a larger eigenproblem (33 vs 64 here)
    keep = s_gen > 1e-9 * s_gen[0]
    assert np.max(np.abs(sig[keep] - s_gen[keep]) / s_gen[keep]) < 1e-9
    assert np.all(sig[m:] == 0.0)                                            # null modes are exact zeros
    mine, _ = align_modes(V[0][:, :8], v_gen)
    assert np.max(np.abs(mine - v_gen)) < 1e-8


def test_values_only_solve(hip):
    fields = make_input("wide_left")
    rank, sig, V = device_solve(hip, fields, False, n_vec=0)
    rank2, sig2, _ = device_solve(hip, fields, False)
    assert V[0].shape[1] == 0 and np.allclose(sig, sig2, rtol=1e-12, atol=0)


@pytest.mark.parametrize("complexify", [False, True])
def test_values_only_cholesky_route_matches_decomposition_route(complexify):
    """rule_n without rotation needs singular values only: the Cholesky route (no field is diagonalised) must give the
    spectra of the eigen-decomposition route (XMCA_CHOLESKY=0) - separate processes, the switch is read once."""
    import json
    import subprocess
    import sys
    code = ("import json, numpy as np\n"
            "from xmca_amd import _hip\n"
            "h = _hip.Handle(0)\n"
            "sp, kept = h.rule_n(60, 300, 200, 2, %r, False, 0, 0, 1e-8, 0, 2, 11, np.float64, 60)\n"
            "print(json.dumps({'sp': sp.tolist(), 'stages': sorted(h.timings())}))\n" % complexify)
    REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = {}
    for flag in ("1", "0"):
        env = dict(os.environ, XMCA_CHOLESKY=flag)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, cwd=REPO, timeout=300)
        assert r.returncode == 0, r.stderr
        out[flag] = json.loads(r.stdout.strip().splitlines()[-1])
    a, b = np.array(out["1"]["sp"]), np.array(out["0"]["sp"])
    n_sig = 30 if complexify else 58                       # meaningful modes (analytic signal: m = T/2 + 1, minus the mean)
    assert np.max(np.abs(a[:, :n_sig] - b[:, :n_sig]) / b[:, :n_sig]) < 1e-9
    assert np.all(np.abs(a[:, n_sig:] - b[:, n_sig:]) < 1e-5 * b[:, :1])
    assert "cholesky" in out["1"]["stages"] and "eigh" not in out["1"]["stages"]             # no field decomposition at all
    assert "eigh" in out["0"]["stages"] and "cholesky" not in out["0"]["stages"]


@pytest.mark.parametrize("shape_a,shape_b,cplx,dtype,tol", [
    ((2, 5), None, False, np.float64, 1e-10), ((3, 1), None, False, np.float64, 1e-10), ((5, 3), None, False, np.float64, 1e-10),
    ((4, 4), None, False, np.float64, 1e-10), ((2, 2), (2, 3), False, np.float64, 1e-10), ((2, 1), (2, 1), False, np.float64, 1e-10),
    ((6, 2), (6, 9), False, np.float64, 1e-10), ((10, 3), (10, 3), True, np.float64, 1e-10), ((7, 20), None, True, np.float64, 1e-10),
    ((3, 10), (3, 12), True, np.float64, 1e-10), ((65, 64), None, False, np.float64, 1e-10),
    ((64, 65), (64, 130), False, np.float64, 1e-9), ((33, 200), (33, 40), True, np.float64, 1e-10),
    ((40, 7), (40, 300), False, np.float64, 1e-10), ((129, 500), (129, 128), False, np.float32, 1e-5)])
def test_small_and_ragged_shapes_match_oracle(shape_a, shape_b, cplx, dtype, tol):
    """tiny, square, tall, wide and mixed (one field narrower than T, the other wider) inputs through the whole class:
    every branch of the field reduction (dual / primal / unreduced) against the numpy restatement."""
    from oracle import ref_numpy as O
    from xmca_amd.array import MCA
    rng = np.random.default_rng(sum(shape_a) + 17)
    fields = [rng.standard_normal(shape_a).astype(dtype)]
    if shape_b is not None:
        fields.append(rng.standard_normal(shape_b).astype(dtype))
    m = MCA(*fields)
    m.solve(complexify=cplx)
    ref = O.OracleModel(*fields).solve(complexify=cplx)["singular_values"]
    s = m._singular_values
    assert len(s) == len(ref)
    keep = ref > 1e-8 * ref[0]
    assert np.max(np.abs(s[keep] - ref[keep]) / ref[keep]) < tol


@pytest.mark.parametrize("two_fields,cplx,N", [(False, False, 700), (True, False, 700), (True, True, 700), (True, False, 240),
                                               (True, True, 240)])
def test_graded_spectrum_fields_match_oracle(two_fields, cplx, N):
    """Smooth fields whose spectrum falls evenly over many decades: the eigensolver switches to its Cholesky LR
    step (jacobi.h); singular values, loadings and the orthonormality of ALL modes must not notice."""
    from oracle import ref_numpy as O
    from xmca_amd.array import MCA
    from conftest import align_modes
    rng = np.random.default_rng(11)
    T = 300          # N = 700: fields wider than T (one-sided / analytic routes); N = 240: narrower (explicit kernel K)

    def field(seed_shift):
        k = T
        x = np.linspace(0, 1, N)
        modes = np.cos(np.pi * (np.arange(k)[:, None] + seed_shift) * x[None, :])
        amp = np.logspace(0, -5, k)
        return (rng.standard_normal((T, k)) * amp) @ modes

    fields = [field(0.0)] + ([field(0.3)[:, :N - 50]] if two_fields else [])
    m = MCA(*fields)
    m.solve(complexify=cplx)
    info = m._device().solve_info()
    assert N < T or any(e["lr_step"] for e in info), info
    ref = O.OracleModel(*fields).solve(complexify=cplx)
    gs = ref["singular_values"]
    s = m._singular_values
    # one field: sigma = lambda / dof is linear in the eigenvalues; two fields: sigma^2 = lambda(K K^H), whose absolute
    # accuracy (~1e-13 sigma_1^2) bounds the relative accuracy of sigma by ~5e-14 (sigma_1 / sigma)^2 in ONE solve - the modes
    # below 1e-3 sigma_1 come from further solves on the deflated fields (Solver::refine_by_deflation, general path) or from
    # the weak block of the subspace problem (Solver::refine_weak_block, analytic-signal path) - DESIGN.md 1
    keep = gs > (1e-9 if two_fields else 1e-6) * gs[0]
    assert np.max(np.abs(s[keep] - gs[keep]) / gs[keep]) < 1e-5
    if two_fields:
        assert ("refine_weak" if (cplx and N > T) else "deflate") in m._device().timings()
    print("graded spectrum", two_fields, cplx, N, "max rel err of sigma down to %.0e sigma_1:" % (gs[keep][-1] / gs[0]),
          float(np.max(np.abs(s[keep] - gs[keep]) / gs[keep])))
    for side, key in enumerate(["left", "right"][:len(fields)]):
        V = m._V[key]
        nk = int(np.sum(keep))
        G = V[:, :nk].conj().T @ V[:, :nk]
        assert np.max(np.abs(G - np.eye(nk))) < (1e-5 if two_fields else 1e-9), key
        gv = ref["V"][side][:, :6]
        mine, _ = align_modes(V[:, :6], gv)
        assert np.max(np.abs(mine - gv)) < 1e-5 * np.max(np.abs(gv)), key
        if two_fields:      # ... and EVERY kept mode, down to 1e-9 sigma_1, phase aligned
            gv = ref["V"][side][:, :nk]
            mine, _ = align_modes(V[:, :nk], gv)
            err = np.max(np.abs(mine - gv), axis=0) / np.max(np.abs(gv), axis=0)
            assert np.max(err) < 1e-5, (key, int(np.argmax(err)), float(np.max(err)))


def test_randomised_shapes_and_spectra_match_oracle():
    """scripts/fuzz_solve.py: 150 random models around the route boundaries (N <> T, analytic / general, T with and without
    an FFT, one / two fields, f32 / f64, noise / signal / graded / low-rank / duplicated columns) - sigma of every non-null
    mode, well-separated leading vectors and the orthonormality of all non-null vectors against the numpy oracle."""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(repo, "scripts", "fuzz_solve.py"), "150", "7"], capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def test_randomised_models_at_multi_tile_size_match_oracle():
    """the same sweep at T = 300 / 640 / 1000 - eigenproblems of several pair tiles, where the route decisions of
    csrc/solver.h (which factor, weak block, per-mode consistency, deflation; DESIGN.md 2, decision table) apply."""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(repo, "scripts", "fuzz_solve.py"), "14", "11", "300,640,1000"], capture_output=True,
                       text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def test_blocking_sync_switch_changes_nothing_but_the_waiting():
    """XMCA_BLOCKING_SYNC=1 (several ranks per node under a CPU quota: host threads sleep while they wait for the GPU,
    hipDeviceScheduleBlockingSync at xmca_create) is read when a process creates its first handle - so in a fresh process:
    same singular values, bit for bit, as this one."""
    import subprocess
    import sys
    from xmca_amd.array import MCA
    left, right = make_input("wide_both")
    m = MCA(left, right)
    m.solve()
    here = m.singular_values()
    code = ("import sys; sys.path[:0] = %r; import numpy as np; from golden_inputs import make_input; "
            "from xmca_amd.array import MCA; l, r = make_input('wide_both'); m = MCA(l, r); m.solve(); "
            "np.save(sys.argv[1], m.singular_values())") % [p for p in sys.path if p]
    out = os.path.join(os.environ.get("TMPDIR", "/tmp"), "xmca_blocking_sync_sv_%d.npy" % os.getpid())
    env = dict(os.environ, XMCA_BLOCKING_SYNC="1")
    subprocess.run([sys.executable, "-c", code, out], check=True, env=env, timeout=600)
    there = np.load(out)
    os.remove(out)
    assert np.array_equal(here, there)
