"""rule_n (xmca/array.py:1716-1771) on the GPU:

* EXACT per-run parity: the device generator's normals of run r (`xmca_surrogate(seed, r, side)`) are fed to the numpy
  oracle (MCA() -> solve -> rotate -> variance, oracle/ref_numpy.py) and compared with row r of `xmca_rule_n` - unrotated,
  rotated real, complex, and a configuration where the reference drops runs (Varimax does not converge in 1000
  iterations, array.py:1759-1763);
* distribution: median and 1 % / 99 % quantiles of every leading mode over 400 device runs against 400 oracle runs fed
  numpy normals, plus a two-sample Kolmogorov-Smirnov test;
* two ranks on ONE device (gloo gather) give bit-for-bit the spectra of a single rank.
"""
import os
import socket
import sys

import numpy as np
import pytest

from golden_inputs import make_input
from oracle import ref_numpy as O
from xmca_amd.array import MCA

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _oracle_run(hip, T, widths, seed, run, cplx, rot):
    data = [hip.surrogate(T * w, seed, run, side).reshape(T, w) for side, w in enumerate(widths)]
    om = O.OracleModel(*data)
    om.solve(complexify=cplx)
    if rot:
        om.rotate(*rot)                   # RuntimeError when Varimax does not converge: the reference drops the run
    return om.variance()


@pytest.mark.parametrize("T,widths,cplx,rot", [
    (40, (24, 18), False, None),            # both fields narrower than T (primal route)
    (60, (300, 200), False, None),          # both wider (values-only Cholesky route)
    (60, (300, 200), True, None),           # analytic-signal subspace (T = 2^2 3 5: Fourier reduction by FFT)
    (62, (300, 200), True, None),           # ... T = 2 x 31: by products with the explicit Fourier vectors
    (60, (300,), False, None),              # EOF
    (48, (120, 30), True, None),            # mixed widths, complex
    (60, (300, 200), False, (6, 1)),        # rotated: Varimax
    (60, (300, 200), False, (5, 3)),        # rotated: Promax
    (60, (200, 150), True, (4, 2)),         # rotated complex
    (40, (24,), False, (4, 1)),             # rotated EOF
    # eigenproblems of 192 rows and more: the tridiagonal values-only route that every C4 surrogate takes
    # (values_by_cholesky -> trd_step_kernel / trd_resident_kernel -> trd_bisect_kernel)
    (500, (1200, 900), True, None),         # m = 251 complex: one launch per column
    (900, (2200, 1700), True, None),        # m = 451 complex: the persistent reduction, tagged exchange
    (800, (2000,), False, None),            # EOF, n = 800 real
    (640, (1500, 1100), False, None),       # real two-field model, n = 640
])
def test_rule_n_runs_equal_the_oracle_on_the_same_normals(hip, T, widths, cplx, rot):
    seed, n_runs = 20240607, (5 if T < 200 else 2)
    n_fields = len(widths)
    rank = min((T,) + widths)
    p, power = rot if rot else (0, 0)
    n_out = p if rot else rank
    trd_before = hip.timings().get("trd_reduce_calls", 0)
    spectra, kept = hip.rule_n(T, widths[0], widths[1] if n_fields == 2 else 0, n_fields, cplx, bool(rot), p, max(power, 1), 1e-8,
                               0, n_runs, seed, np.float64, n_out)
    assert spectra.shape == (n_runs, n_out)
    for r in range(n_runs):
        try:
            ref = _oracle_run(hip, T, widths, seed, r, cplx, rot)
        except RuntimeError:
            assert kept[r] == 0
            continue
        assert kept[r] == 1
        assert ref.shape == (n_out,)
        keep = ref > 1e-7 * ref[0]          # null modes (centering; the analytic signal keeps T/2 of them) carry rounding only
        assert np.max(np.abs(spectra[r][keep] - ref[keep]) / ref[keep]) < 1e-5, (r, widths, cplx, rot)
        assert np.all(np.abs(spectra[r][~keep]) < 1e-5 * ref[0])
    if T >= 200:
        assert hip.timings().get("trd_reduce_calls", 0) >= trd_before + n_runs        # ... and it is the route that ran
    # a later block of runs is keyed by the run index only
    again, _ = hip.rule_n(T, widths[0], widths[1] if n_fields == 2 else 0, n_fields, cplx, bool(rot), p, max(power, 1), 1e-8,
                          n_runs - 2, n_runs, seed, np.float64, n_out)
    assert np.array_equal(again, spectra[n_runs - 2:n_runs])


def test_numpy_generator_equals_the_device_generator(hip):
    """oracle/philox_numpy.py restates philox_normal_kernel (Philox4x32-10 keyed (seed, run, side) + Box-Muller): the
    integers are the same by construction, the normals to the last ulps of the two libms.  With it a surrogate exists
    without a GPU - tests/golden/rule_n_c4_run0.npz is the REAL reference's spectrum of one."""
    from oracle.philox_numpy import surrogate
    for n, seed, run, side in [(1, 1, 0, 0), (1001, 1, 0, 1), (65536, 20240607, 3, 0), (300000, (1 << 40) + 12345, 7, 1)]:
        dev = hip.surrogate(n, seed, run, side)
        ref = surrogate(n, seed, run, side)
        assert dev.shape == ref.shape
        assert np.max(np.abs(dev - ref)) < 4e-15, (n, seed, run, side)
    g = np.load(os.path.join(REPO, "tests", "golden", "rule_n_c4_run0.npz"))
    T, (Nx, Ny), seed = int(g["T"]), [int(x) for x in g["widths"]], int(g["seed"])
    assert np.max(np.abs(hip.surrogate(64, seed, 0, 0) - g["first_normals_left"])) < 4e-15         # the fields the reference saw
    assert np.max(np.abs(hip.surrogate(64, seed, 0, 1) - g["first_normals_right"])) < 4e-15
    assert abs(hip.surrogate(T * Nx, seed, 0, 0).sum() - float(g["checksum_left"])) < 1e-6


def test_c4_surrogate_at_full_size_equals_the_reference(hip):
    """BASELINE configs[3] at FULL size: surrogate (seed 1, run 0) of rule_n on T = 5000 x (20 000, 15 000), complexify -
    the normals the device generates, restated in numpy (oracle/philox_numpy.py), went through the REAL reference
    (xmca/array.py:1753-1765: MCA(*data).solve(complexify=True), 392 s on 8 cores; oracle/make_config_goldens.py c4_run0).
    Row 0 of xmca_rule_n must be that spectrum: all 2500 non-null modes at 1e-5; the reference drops the rotated run
    (Varimax does not converge on complex white noise, array.py:1762-1763) and so must the device."""
    g = np.load(os.path.join(REPO, "tests", "golden", "rule_n_c4_run0.npz"))
    T, (Nx, Ny), seed = int(g["T"]), [int(x) for x in g["widths"]], int(g["seed"])
    ref = g["variance"]
    trd_before = hip.timings().get("trd_reduce_calls", 0)
    spectra, kept = hip.rule_n(T, Nx, Ny, 2, True, False, 0, 1, 1e-8, 0, 1, seed, np.float64, ref.size)
    assert kept[0] == 1 and hip.timings().get("trd_reduce_calls", 0) == trd_before + 1
    keep = ref > 1e-7 * ref[0]
    assert keep.sum() == T // 2
    assert np.max(np.abs(spectra[0][keep] - ref[keep]) / ref[keep]) < 1e-5
    assert np.all(np.abs(spectra[0][~keep]) < 1e-5 * ref[0])
    assert int(g["rotated_dropped"]) == 1
    _, kept_rot = hip.rule_n(T, Nx, Ny, 2, True, True, 20, 4, 1e-8, 0, 1, seed, np.float64, 20)
    assert kept_rot[0] == 0


_DROP_CASE = dict(T=150, widths=(400, 300), seed=99, n_runs=8, rot=(30, 4))


def _drop_case_reference_kept(hip, spectra=None):
    c = _DROP_CASE
    ref_kept = []
    for r in range(c["n_runs"]):
        try:
            ref = _oracle_run(hip, c["T"], c["widths"], c["seed"], r, True, c["rot"])
            ref_kept.append(1)
            if spectra is not None:
                assert np.max(np.abs(spectra[r] - ref) / ref) < 1e-4      # (hundreds of iterations on noise: 1e-5 per iteration adds up)
        except RuntimeError:
            ref_kept.append(0)
    return ref_kept


def test_rule_n_drops_the_runs_the_reference_drops(hip):
    """complex white noise, n_rot = 30, power = 4: Varimax needs more than 1000 iterations for most surrogates (SURVEY.md
    6: 4/4 seeds at T = 1000) - the reference catches the RuntimeError and drops the run (array.py:1762-1763).
    Kernel level: `xmca_rule_n` reports exactly the runs the oracle keeps on the same normals."""
    c = _DROP_CASE
    spectra, kept = hip.rule_n(c["T"], c["widths"][0], c["widths"][1], 2, True, True, c["rot"][0], c["rot"][1], 1e-8, 0, c["n_runs"],
                               c["seed"], np.float64, c["rot"][0])
    ref_kept = _drop_case_reference_kept(hip, spectra)
    assert list(kept) == ref_kept
    assert 0 in ref_kept                                                  # the case is only useful if something is dropped


def test_rule_n_dropped_runs_shrink_the_run_axis_of_the_class(hip):
    """... and through the class (array.py:1762-1769): the model's own rotation converges (Hann patterns + cosine PCs,
    generator B at this size: 134 iterations in the reference), its white-noise surrogates mostly do not, and the
    dropped runs shrink the run axis of `MCA.rule_n`."""
    from golden_inputs import gen_B
    c = _DROP_CASE
    ref_kept = _drop_case_reference_kept(hip)
    assert 0 in ref_kept
    A, B = gen_B(c["T"], c["widths"][0], c["widths"][1], k=30, seed=3)
    m = MCA(A, B)
    m.solve(complexify=True)
    m.rotate(*c["rot"])                                                   # converges (no RuntimeError)
    out = m.rule_n(c["n_runs"], seed=c["seed"])
    assert out.shape == (c["rot"][0], int(np.sum(ref_kept)))
    assert np.all(np.isfinite(out))


@pytest.mark.parametrize("cplx,rot", [(False, None), (True, None), (False, (4, 1))])
def test_rule_n_distribution_quantiles_match_oracle(cplx, rot):
    """per-mode median and 1 % / 99 % quantiles over >= 200 runs within Monte-Carlo error of the oracle's
    (fed numpy normals), here 400 + 400 runs; the quantile errors use the normal approximation of the order statistics:
    se(q_p) = sigma sqrt(p (1 - p) / n) / phi(z_p)."""
    from scipy import stats
    fields = make_input("small_both")
    m = MCA(*fields)
    m.solve(complexify=cplx)
    om = O.OracleModel(*fields)
    om.solve(complexify=cplx)
    if rot:
        m.rotate(*rot)
        om.rotate(*rot)
    n = 400
    mine = m.rule_n(n, seed=7)
    rng = np.random.default_rng(11)
    ref = O.rule_n(om, n, normal=lambda shape: rng.standard_normal(shape))
    assert mine.shape[0] == ref.shape[0] and mine.shape[1] >= 0.9 * n and ref.shape[1] >= 0.9 * n
    k = min(6, ref.shape[0])
    for mode in range(k):
        a, b = mine[mode], ref[mode]
        sd = np.sqrt(0.5 * (a.var() + b.var()))
        for prob in (0.5, 0.01, 0.99):
            z = stats.norm.ppf(prob)
            se = sd * np.sqrt(prob * (1 - prob)) / stats.norm.pdf(z) * np.sqrt(1.0 / len(a) + 1.0 / len(b))
            assert abs(np.quantile(a, prob) - np.quantile(b, prob)) < 5 * se, (mode, prob)
        assert stats.ks_2samp(a, b).pvalue > 1e-4, mode


# ----------------------------------------------------------------------------------------------
# two ranks, one device: sharded rule_n + gather (gloo here; the same code path uses RCCL under backend "nccl")
# ----------------------------------------------------------------------------------------------
def _worker(rank, world, port, n_runs, q):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import torch.distributed as td
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    td.init_process_group("gloo", rank=rank, world_size=world)
    from golden_inputs import make_input as mk
    from xmca_amd import _hip
    from xmca_amd.array import MCA as M
    m = M(*mk("wide_both"), handle=_hip.Handle(0))
    m.solve(complexify=True)
    out = m.rule_n(n_runs, seed=1000 + rank)        # ranks disagree on purpose: rank 0's seed is broadcast
    m.rotate(5, 2)
    out_rot = m.rule_n(n_runs, seed=77)
    q.put((rank, out, out_rot))
    td.destroy_process_group()


def test_two_ranks_on_one_device_equal_a_single_rank():
    import torch.multiprocessing as mp
    from xmca_amd import _hip
    n_runs, world = 7, 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_runs, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    m = MCA(*make_input("wide_both"), handle=_hip.Handle(0))
    m.solve(complexify=True)
    single = m.rule_n(n_runs, seed=1000)
    m.rotate(5, 2)
    single_rot = m.rule_n(n_runs, seed=77)
    for _, out, out_rot in results:
        assert np.array_equal(out, single)                   # bit for bit: the generator is keyed by (seed, run, side)
        assert np.array_equal(out_rot, single_rot)


def _nccl_worker(port, n_runs, q):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import torch
    import torch.distributed as td
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    td.init_process_group("nccl", rank=0, world_size=1)
    from golden_inputs import make_input as mk
    from xmca_amd import _hip, dist
    from xmca_amd.array import MCA as M
    m = M(*mk("wide_both"), handle=_hip.Handle(0))
    m.solve(complexify=True)
    assert td.get_backend() == "nccl" and dist._comm_device(td, m._device()).type == "cuda"
    assert dist.broadcast_seed(123456789, m._device()) == 123456789          # RCCL broadcast of the seed
    out = m.rule_n(n_runs, seed=1000)                                        # spectra through RCCL all_gather
    m.rotate(5, 2)
    out_rot = m.rule_n(n_runs, seed=77)
    q.put((out, out_rot))
    td.destroy_process_group()


def test_rule_n_through_an_rccl_group_of_one_rank():
    """The `nccl` (= RCCL) branch of xmca_amd/dist.py - seed broadcast and spectra all_gather on CUDA tensors - in a
    one-rank process group on the one GPU of the test box (array.py:1753-1769 sharded): bit for bit the spectra of a
    process without torch.distributed."""
    import torch.multiprocessing as mp
    from xmca_amd import _hip
    n_runs = 5
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_nccl_worker, args=(port, n_runs, q))
    p.start()
    out, out_rot = q.get(timeout=600)
    p.join(timeout=120)
    assert p.exitcode == 0
    m = MCA(*make_input("wide_both"), handle=_hip.Handle(0))
    m.solve(complexify=True)
    assert np.array_equal(out, m.rule_n(n_runs, seed=1000))
    m.rotate(5, 2)
    assert np.array_equal(out_rot, m.rule_n(n_runs, seed=77))


def test_values_only_route_never_falls_back_inside_four_lanes():
    """Every surrogate of an unrotated two-field rule_n takes the values-only Cholesky route (`values_by_cholesky`): ONE values-only
    eigenproblem per run, no `eigh` stage.  Inside several lanes the launches of different streams interleave - in round 5 the
    panel kernel of the factorisation overwrote its diagonal block while late workgroups of the same launch still had to read
    it, the factorisation then failed now and then, and the solver fell back to its eigen-decomposition route: the spectra
    stayed right (so no parity test noticed), 5 of 9 surrogates took twice the time.  Counted here: reductions == runs."""
    import json
    import subprocess
    code = ("import sys, json, numpy as np; sys.path.insert(0, %r); from xmca_amd import _hip; h = _hip.Handle(0);"
            "args = (2000, 5000, 4000, 2, True, False, 0, 1, 1e-8);"
            "h.rule_n(*args, 0, 4, 3, np.float64, 2000); h.reset_timings();"
            "sp, kept = h.rule_n(*args, 0, 24, 5, np.float64, 2000); t = h.timings();"
            "print(json.dumps({'keys': sorted(t), 'reductions': t.get('trd_reduce_calls', 0), 'kept': int(kept.sum())}))" % REPO)
    for lanes in ("4", "3"):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, XMCA_RULE_N_LANES=lanes), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        out = json.loads(r.stdout.splitlines()[-1])
        assert out["kept"] == 24 and out["reductions"] == 24, out
        assert "eigh" not in out["keys"] and "cholesky" in out["keys"], out


def test_native_rccl_communicator_of_one_rank():
    """The library's own RCCL communicator (C ABI xmca_comm_*, csrc/comm.h; SURVEY 8(b) `mca_comm_*`) in a subprocess WITHOUT
    torch: unique id -> ncclCommInitRank (world 1: all the one-GPU test box allows, RCCL refuses two ranks on a device) ->
    ncclAllGather / ncclBroadcast round trips, and xmca_rule_n_sharded = xmca_rule_n bit for bit (array.py:1753-1769 sharded);
    dist.init_native's file rendezvous supplies the id."""
    import subprocess
    import tempfile
    code = ("import sys, os, numpy as np; sys.path.insert(0, %r);"
            "from xmca_amd import _hip, dist;"
            "assert 'torch' not in sys.modules;"
            "h = _hip.Handle(0); c = dist.init_native(h, rank=0, world=1, id_file=sys.argv[1] + '.id');"
            "x = np.arange(12.0).reshape(3, 4); g = c.allgather(x); assert g.shape == (1, 3, 4) and np.array_equal(g[0], x);"
            "b = c.broadcast([3.0, 2.0 ** 40 + 1]); assert np.array_equal(b, [3.0, 2.0 ** 40 + 1]);"
            "args = (150, 400, 300, 2, True, True, 6, 2, 1e-8);"
            "a, ka = h.rule_n(*args, 0, 5, 2 ** 40 + 77, np.float64, 6);"
            "s, ks = dist.sharded_rule_n(h, 5, T=150, Nx=400, Ny=300, n_fields=2, complexify=True, rotated=True, p=6, power=2,"
            " tol=1e-8, seed=2 ** 40 + 77, dtype=np.float64, n_out=6, comm=c);"
            "assert 'torch' not in sys.modules;"
            "r, w, n, nb = c.info(); c.close();"
            "np.savez(sys.argv[1], a=a, ka=ka, s=s, ks=ks, info=np.array([r, w, n, nb]))" % REPO)
    with tempfile.TemporaryDirectory() as tmp:
        dst = os.path.join(tmp, "o.npz")
        r = subprocess.run([sys.executable, "-c", code, dst], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        o = dict(np.load(dst))
    assert np.array_equal(o["a"], o["s"]) and np.array_equal(o["ka"], o["ks"]) and o["ka"].sum() >= 1
    rank, world, n_coll, n_bytes = [int(v) for v in o["info"]]
    assert (rank, world) == (0, 1) and n_coll == 4                      # allgather, broadcast, seed broadcast, spectra all-gather
    assert n_bytes == 8 * (12 + 2 + 2 + (5 + 1) * 7)                    # (5 runs + the status row) x (6 values + kept)


def test_native_sharded_rule_n_reports_a_failed_shard_after_the_collective():
    """VERDICT r05 weak #9 / advisor: xmca_rule_n_sharded used to return BEFORE the all-gather when its own runs failed, leaving the
    other ranks in ncclAllGather for good.  Now the status travels through the collective (a status row per rank) and every rank
    returns the error afterwards.  One rank is all a one-GPU box allows: the shard is made to fail (XMCA_TEST_FAIL_RANK), the
    communicator's counters show that the all-gather WAS carried out, and the error names the rank."""
    import subprocess
    import tempfile
    code = ("import sys, os, numpy as np; sys.path.insert(0, %r);"
            "from xmca_amd import _hip, dist;"
            "h = _hip.Handle(0); c = dist.init_native(h, rank=0, world=1, id_file=sys.argv[1] + '.id');"
            "r0, w0, n0, b0 = c.info();"
            "os.environ['XMCA_TEST_FAIL_RANK'] = '0';"
            "msg = '';\n"
            "try:\n"
            "    dist.sharded_rule_n(h, 5, T=150, Nx=400, Ny=300, n_fields=2, complexify=True, rotated=False, p=0, power=1,"
            " tol=1e-8, seed=3, dtype=np.float64, n_out=150, comm=c)\n"
            "except Exception as err:\n"
            "    msg = str(err)\n"
            "r, w, n, nb = c.info(); c.close();"
            "print('COLL', n - n0, 'MSG', msg)" % REPO)
    with tempfile.TemporaryDirectory() as tmp:
        r = subprocess.run([sys.executable, "-c", code, os.path.join(tmp, "o")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("COLL")][-1]
    assert line.split()[1] == "2", line                                # seed broadcast + the all-gather: both entered
    assert "shard of rank 0 failed" in line and "injected failure" in line, line


def test_host_budget_script_on_a_stand_in():
    """scripts/host_budget.py (the host-CPU budget of 8 ranks x 3 lanes under the 16-CPU quota of a GPU box; the full-size
    record is profiles/r05_host_budget.json) on its T = 1000 stand-in with 3 ranks sharing the GPU: the legs run, nothing gives
    up in the single-rank legs, CPU-seconds per surrogate are reported, and two CPUs do not halve the rate."""
    import json
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(REPO, "scripts", "host_budget.py"), "--small", "--ranks", "3"], cwd=REPO,
                       capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout[r.stdout.index("{"):])
    for leg in ("unpinned", "pinned_2_cpus", "spinning_sync_unpinned"):
        assert out[leg]["giveups"] == 0 and out[leg]["cpu_ms_per_surrogate"] > 0 and out[leg]["surrogates_per_s"] > 1
    assert out["pinned_2_cpus"]["n_affinity"] == 2
    assert out["unpinned"]["checksum"] == out["pinned_2_cpus"]["checksum"] == out["spinning_sync_unpinned"]["checksum"]
    assert out["loss_at_2_cpus"] < 0.5
    assert out["ranks_sharing_gpu0"]["surrogates"] == 6 and out["ranks_sharing_gpu0"]["cpu_ms_per_surrogate"] > 0


def test_bench_py_launches_its_own_ranks():
    """`python bench.py --gpus 2` outside a launcher starts two ranks itself (here both on GPU 0, gloo gather) and reports
    n_gpus = 2 with the run-sharded rule_n - the flow the driver's 2/4/8-GPU scaling runs use (RCCL there)."""
    import json
    import subprocess
    env = dict(os.environ, XMCA_BENCH_SHARE_GPU="1")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--T", "400",
                        "--N", "1500", "--no-cpu-baseline", "--no-e2e", "--rule-n-runs", "1", "--rule-n-rotated-runs", "0"], env=env, cwd=REPO,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                  # rank 0 only
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["rule_n"]["runs"] == 2 and line["rule_n"]["shape"] == [5000, 2]
    # the gathered spectra are those of the runs 0, 1: run 0 is the surrogate the REAL reference solved (rule_n_c4_run0.npz)
    assert line["rule_n"]["run0_vs_reference"]["max_rel_err_nonnull_modes"] < 1e-5 and line["rule_n"]["run0_vs_reference"]["modes"] == 2500
    assert line["roofline"]["kernel"].startswith("jacobi_fused_round_kernel")      # (T = 400: below the tridiagonal route's 768 rows)
    # a launcher that started a different number of ranks than --gpus is an error, not a silent n_gpus = 1
    bad = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "4", "--steps", "1"],
                         env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), cwd=REPO, capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0 and "WORLD_SIZE" in bad.stderr


def test_bench_py_reports_the_tridiagonal_reduction_as_its_dominant_kernel():
    """At a workload whose T x T eigenproblem takes the tridiagonal route with vectors (T >= 768) the `roofline` block of the
    bench line must be the reduction kernel's - the branch the driver's full-size run takes - with the flop count of a
    Householder tridiagonalisation, (4/3) T^3, and the bound it really has (one exchange per column: latency)."""
    import json
    import subprocess
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    T = 800
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--T", str(T),
                        "--N", "2400", "--no-cpu-baseline", "--no-e2e", "--no-rule-n"], env=env, cwd=REPO,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    rf = line["roofline"]
    # (the name is the library's own: xmca_get_reduction_info - one persistent launch or the chain of re-packed ones, round 6)
    assert ("trd_resident_kernel<real,NC=" in rf["kernel"] or rf["kernel"].startswith("trd_step_kernel")) and rf["bound"] == "latency"
    assert abs(rf["flops_per_launch"] - 4.0 / 3.0 * T ** 3) < 1.0
    assert 0 < rf["frac"] < 1 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert rf["exchange_us_per_column"] > 0 and rf["launches_per_step"] >= 1
    g = line["roofline_gemm"]
    assert g["bound"] == "mfma" and 0 < g["frac"] < 1 and g["kernel"].startswith("gemm_kernel")


def test_persistent_kernels_of_four_lanes_pass_one_gate():
    """Rotated rule_n with four surrogates in flight, at a size where every surrogate runs BOTH persistent kernels of the
    library - the register-resident tridiagonal reduction (n = 800 with vectors: every CU) and the one-launch Varimax loop (a
    grid of its own per lane).  Round 3 serialised only the reductions; a Varimax grid of another lane could interleave with
    one, and the two partly resident grids then waited for each other until the bounded spins gave up (~0.2 s) and the work
    was repeated launch by launch.  All persistent launches of a device now pass one gate (csrc/common.h PersistGate): no
    give-up, and the call takes what the kernels take."""
    import subprocess
    code = ("import sys, time, json, numpy as np; sys.path.insert(0, %r); from xmca_amd import _hip; h = _hip.Handle(0);"
            "args = (800, 2000, 0, 1, False, True, 6, 1, 1e-8);"
            "h.rule_n(*args, 0, 4, 5, np.float64, 6);"                                     # warm-up: lanes, workspaces
            "g0 = _hip.load_library().xmca_persistent_giveups(); h.reset_timings(); t0 = time.perf_counter();"
            "sp, kept = h.rule_n(*args, 0, 16, 5, np.float64, 6); dt = time.perf_counter() - t0;"
            "tm = h.timings();"
            "print(json.dumps({'seconds': dt, 'giveups': _hip.load_library().xmca_persistent_giveups() - g0, 'kept': int(kept.sum()),"
            " 'resident': tm.get('trd_resident_calls', 0), 'reductions': tm.get('trd_reduce_calls', 0)}))" % REPO)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, XMCA_RULE_N_LANES="4", XMCA_TRACE="giveup"), capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    import json
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["giveups"] == 0, r.stderr[-1000:]
    assert out["kept"] >= 12                                           # (a white-noise surrogate may not converge in 1000 iterations: dropped)
    assert out["reductions"] == 16 and out["resident"] == 16          # every surrogate: one reduction, by the persistent kernel
    assert out["seconds"] < 10.0                                       # sanity only (~15 ms of kernels per surrogate): the counters above are the property; a tight wall-clock bound flakes on a cold or shared GPU (advisor, round 4)


def test_dead_workgroups_of_the_tagged_reduction_do_not_stall_two_lanes():
    """Two surrogates in flight at a size whose persistent reduction has many more workgroups than live rows towards its end
    (m = 451 complex, 113 workgroups; m = 900 real).  In the tagged exchange a workgroup without live rows publishes nothing,
    so nobody waits for it; as long as it kept listening it could fall two columns behind (the other lane's kernels on its CU),
    find its slots overwritten with the tag of the column after next, run out of its spins and force the reduction to be
    repeated launch by launch: a give-up in nearly every call at this size, 0.2 s each, and - through the different summation
    order of the fallback - different bits from call to call.  Dead workgroups leave now, and the writer of d / e / tau is the
    owner of the last row: no give-up, the same bits in every call."""
    import subprocess
    code = ("import sys, json, numpy as np; sys.path.insert(0, %r); from xmca_amd import _hip; h = _hip.Handle(0); lib = _hip.load_library();"
            "out = {};\n"
            "for T, Nx, Ny, c in ((900, 2200, 1700, True), (900, 2200, 1700, False)):\n"
            "    args = (T, Nx, Ny, 2, c, False, 0, 0, 1e-8)\n"
            "    ref, _ = h.rule_n(*args, 0, 8, 3, np.float64, T)\n"
            "    g0 = lib.xmca_persistent_giveups(); same = True\n"
            "    for r in range(5):\n"
            "        sp, _ = h.rule_n(*args, 0, 8, 3, np.float64, T); same = same and bool(np.array_equal(sp, ref))\n"
            "    out[str(c)] = [int(lib.xmca_persistent_giveups() - g0), same, h.timings().get('trd_resident_calls', 0)]\n"
            "print(json.dumps(out))" % REPO)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, XMCA_RULE_N_LANES="2", XMCA_TRACE="giveup"), capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    import json
    out = json.loads(r.stdout.strip().splitlines()[-1])
    for key in ("True", "False"):
        assert out[key][0] == 0 and out[key][1] is True and out[key][2] > 0, (out, r.stderr[-600:])


def test_rule_n_spectra_do_not_depend_on_the_number_of_lanes():
    """xmca_rule_n keeps several surrogates in flight (one stream + workspaces + host thread per lane, xmca_hip.cpp
    rule_n_impl); the generator is keyed by (seed, run, side), so 1, 2, 3 and 5 lanes must give the same bits - unrotated,
    rotated with dropped runs, and with the device memory pool switched off."""
    import subprocess
    code = ("import sys, numpy as np; sys.path.insert(0, %r); from xmca_amd import _hip; h = _hip.Handle(0);"
            "a, ka = h.rule_n(60, 300, 200, 2, True, False, 0, 1, 1e-8, 0, 7, 5, np.float64, 60);"
            "b, kb = h.rule_n(150, 400, 300, 2, True, True, 30, 4, 1e-8, 0, 6, 99, np.float64, 30);"
            "c, kc = h.rule_n(48, 120, 0, 1, False, True, 4, 1, 1e-8, 2, 9, 3, np.float32, 4);"
            "np.savez(sys.argv[1], a=a, ka=ka, b=b, kb=kb, c=c, kc=kc)" % REPO)
    import tempfile
    outs = []
    with tempfile.TemporaryDirectory() as tmp:
        for lanes, pool in [("1", "1"), ("2", "1"), ("3", "1"), ("5", "0")]:
            dst = os.path.join(tmp, "l%s.npz" % lanes)
            env = dict(os.environ, XMCA_RULE_N_LANES=lanes, XMCA_POOL=pool)
            r = subprocess.run([sys.executable, "-c", code, dst], env=env, capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, r.stderr[-2000:]
            outs.append(dict(np.load(dst)))
    assert 0 in outs[0]["kb"] and 1 in outs[0]["kb"]                 # the rotated case drops some runs and keeps others
    for o in outs[1:]:
        for k in outs[0]:
            assert np.array_equal(o[k], outs[0][k]), k


@pytest.mark.parametrize("tagged", ["1", "0"])
def test_persistent_reduction_gives_the_same_bits_for_one_and_two_lanes_in_both_exchange_forms(tagged):
    """The persistent tridiagonal reduction (eigenproblems of 451 rows: T = 900 complex) inside 1 and 2 surrogate lanes, with
    the tagged exchange and with the epoch flags (XMCA_TRD_TAGGED=0).  Inside lanes the tagged form deals CONTIGUOUS rows to
    the workgroups (they leave early); the flags form sums per-workgroup partials of p^H v, so its row ownership must not
    depend on the lane count (advisor, round 4: it did) - a rank that holds a single run gets one lane, its neighbour three."""
    import subprocess
    import tempfile
    code = ("import sys, numpy as np; sys.path.insert(0, %r); from xmca_amd import _hip; h = _hip.Handle(0);"
            "a, ka = h.rule_n(900, 2200, 1700, 2, True, False, 0, 1, 1e-8, 0, 4, 11, np.float64, 900);"
            "t = h.timings(); lib = _hip.load_library();"
            "np.savez(sys.argv[1], a=a, ka=ka, g=np.int64(lib.xmca_persistent_giveups()))" % REPO)
    outs = []
    with tempfile.TemporaryDirectory() as tmp:
        for lanes in ("1", "2"):
            dst = os.path.join(tmp, "l%s.npz" % lanes)
            env = dict(os.environ, XMCA_RULE_N_LANES=lanes, XMCA_TRD_TAGGED=tagged)
            r = subprocess.run([sys.executable, "-c", code, dst], env=env, capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, r.stderr[-2000:]
            outs.append(dict(np.load(dst)))
    assert int(outs[0]["g"]) == 0 and int(outs[1]["g"]) == 0          # (a give-up takes the launch-per-column path: other bits)
    assert np.array_equal(outs[0]["a"], outs[1]["a"]) and np.array_equal(outs[0]["ka"], outs[1]["ka"])


def test_an_error_in_a_lane_is_reported(hip):
    """n_out that does not match the rank is rejected inside every lane (first error wins, the others are joined)."""
    with pytest.raises(Exception, match="n_out"):
        hip.rule_n(40, 24, 18, 2, False, False, 0, 1, 1e-8, 0, 6, 1, np.float64, 17)
    sp, kept = hip.rule_n(40, 24, 18, 2, False, False, 0, 1, 1e-8, 0, 6, 1, np.float64, 18)      # the handle is still usable
    assert sp.shape == (6, 18) and kept.sum() == 6
