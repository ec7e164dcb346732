#!/usr/bin/env python3
"""Headline benchmark: MCA.solve() + MCA.rotate() on the BASELINE.json configuration C2
(synthetic EOF, T = 2920 x N = 10 000 grid points, float64, rotate(n_rot=10, power=1)), one MI355X per rank, and the
run-sharded rule_n of configuration C4 (surrogates/s over all ranks).

    python bench.py --gpus N --steps K --warmup W

* a "step" = one pass of the hot path: device solve (Gram GEMM, eigensolver, back-projection of all `rank` modes) +
  Varimax/Promax rotation, with the centered field ALREADY RESIDENT in HBM when the timed region starts (the
  PCIe-inclusive time through the MCA class is reported separately as `e2e_ms`, never as `value`).
* --gpus N > 1 without a torch.distributed environment: bench.py launches itself as N ranks (torch.distributed.run,
  127.0.0.1), one GPU each; under a launcher (WORLD_SIZE set) it is one of the ranks and WORLD_SIZE must equal --gpus.
  Every rank processes its own replica of the workload (the path has no exchange step) -> weak scaling;
  `value` = steps of all ranks / max-over-ranks wall time.  The sharded rule_n (the only collective of the path: one
  all_gather of the spectra) runs after the timed region on the C4 configuration and is reported as
  `rule_n.surrogates_per_s` (all ranks together).  XMCA_BENCH_SHARE_GPU=1 puts every rank on GPU 0 with the gloo
  backend (RCCL refuses two ranks on one device) - the way the multi-rank flow is exercised on a 1-GPU box.
* prints ONE JSON line on rank 0 with the driver's contract keys plus
    `roofline`       the DOMINANT kernel of the step - since round 3 the Householder tridiagonalisation of the T x T Gram
                     matrix (trd_resident_kernel, csrc/tridiag.h; the block-Jacobi round kernel when XMCA_TRIDIAG=0):
                     algorithmic flops / average launch duration, hipEvents on the library's stream in the timed region;
    `roofline_gemm`  the covariance (Gram) GEMM, the kernel north_star quotes an MFMA utilisation for;
    `roofline_c5`    the same product at BASELINE configs[4] (T = 1200 x N = 1 036 800 float32, resident field);
    `cpu_baseline`   the numpy oracle on this host (N = 1 only): one C2 solve()+rotate() and one reduced C4 surrogate.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time


def _cap_host_threads():
    """BLAS/OpenMP threads = the CPUs this container may actually use.  The GPU boxes show 256 cores but run under a cgroup
    quota of 16: a BLAS call with one spinning thread per core exhausts the quota and the kernel then stalls every thread of
    the process - including the one waiting for the device - for 50-80 ms at a time (measured: nr_throttled in cpu.stat)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, int(parts[0]) // int(parts[1])))
            else:
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                    per = int(f.read())
                if int(parts[0]) > 0:
                    n = min(n, max(1, int(parts[0]) // per))
            break
        except (OSError, ValueError, IndexError):
            continue
    try:                                        # several ranks on one node share the quota
        n = max(1, n // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1"))))
    except ValueError:
        pass
    for v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ.setdefault(v, str(n))
    return n


HOST_THREADS = _cap_host_threads()          # before numpy / torch load their thread pools
if os.environ.get("LOCAL_WORLD_SIZE", "1") not in ("", "1"):
    # several ranks on one node: the library's waits sleep on an interrupt instead of spinning (1.4 host cores per rank while
    # the GPU works, more with the surrogate lanes of rule_n - enough to exhaust a container's CPU quota: csrc/common.h)
    os.environ.setdefault("XMCA_BLOCKING_SYNC", "1")

import numpy as np   # noqa: E402

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

F64_MFMA_PEAK_TF = 78.6     # MI355X FP64 matrix (= FP64 vector) peak, AMD CDNA4 datasheet; 256 CU * 4 SIMD * 32 FLOP/clk * 2.4 GHz
F32_MFMA_PEAK_TF = 157.3    # MI355X_MICROARCH.md, "Peak FP32 (matrix)"
BF16_MFMA_PEAK_TF = 2500.0  # dense bf16 MFMA peak (the task's figure; AMD's headline number includes 2:1 sparsity)


def gen_A(T=2920, N=10_000, k=20, seed=0):
    """SURVEY.md Appendix C generator A (config 2): low-rank Gaussian signal + unit noise, float64."""
    rng = np.random.default_rng(seed)
    return (rng.standard_normal((T, k)) * np.linspace(10, 1, k)) @ rng.standard_normal((k, N)) \
        + rng.standard_normal((T, N))


def device_step(h, T, N, n_rot, power, dtype):
    """solve + rotate on resident data; returns (sigma, R, n_iter)."""
    rank = h.solve(1)
    sig = h.singular_values(rank)
    Vt = h.vectors(0, n_rot, N, dtype)                    # n_rot x N (0.8 MB): the loadings of array.py:821-822
    L = Vt.T * np.sqrt(sig[:n_rot])
    out = h.rotate_loadings(L, n_left=N, power=power, tol=1e-8)
    return sig, out


def csrc_hash():
    """sha256 (16 hex digits) over the kernel sources: the PMC file below is only valid for the build it was taken from."""
    import hashlib
    hh = hashlib.sha256()
    base = os.path.join(REPO, "xmca_amd", "csrc")
    for name in sorted(os.listdir(base)):
        with open(os.path.join(base, name), "rb") as f:
            hh.update(name.encode() + b"\0" + f.read())
    return hh.hexdigest()[:16]


PMC_FILE = os.path.join("profiles", "r06_pmc_c2.json")                 # counter passes of the C2 step (scripts/r06_profiles.sh)
PMC_GRAM_C2 = os.path.join("profiles", "r06_pmc_gram_c2.json")         # ... of the Gram product alone (scripts/gram_only.py c2)
PMC_GRAM_C5 = os.path.join("profiles", "r06_pmc_gram_c5.json")         # ... and at C5


def pmc_traffic(kernel, same_workload, pmc_file=None):
    """HBM-side bytes per launch of `kernel` from the committed PMC passes (separate rocprofv3 --pmc runs, FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for gfx950).  A file carries the hash of xmca_amd/csrc it was measured on: None (never a
    stale number) when the sources have changed since, when the file is absent or when the workload is not the one that was
    profiled."""
    path = os.path.join(REPO, pmc_file or PMC_FILE)
    if not same_workload or not os.path.exists(path):
        return None
    try:
        with open(path) as f:
            d = json.load(f)
        if d.get("csrc_sha16") != csrc_hash():
            return None
        return float(d[kernel]["hbm_side_bytes_per_launch_gfx950_corrected"])
    except (KeyError, ValueError, OSError):
        return None


def launch_ranks(args):
    """--gpus N > 1 outside a launcher: re-run this script as N ranks (one per GPU) and pass rank 0's line through."""
    if os.environ.get("XMCA_BENCH_SHARE_GPU") != "1":
        import torch
        if torch.cuda.device_count() < args.gpus:
            sys.exit("bench.py: --gpus %d but only %d GPU(s) visible (XMCA_BENCH_SHARE_GPU=1 puts every rank on GPU 0)"
                     % (args.gpus, torch.cuda.device_count()))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.run(cmd, env=env).returncode


def rule_n_model(MCA, h, Tn, Nxn, Nyn):
    """only the meta data rule_n reads is needed (array.py:1744-1749)"""
    model = MCA.__new__(MCA)
    MCA.__init__(model)
    model._keys = ['left', 'right']
    model._n_observations = {'left': Tn, 'right': Tn}
    model._n_variables = {'left': Nxn, 'right': Nyn}
    model._analysis.update({'is_bivariate': True, 'is_complex': True, 'rank': Tn, 'n_rot': Tn})
    model._norm = {'left': np.ones(Tn), 'right': np.ones(Tn)}
    model._var_idx = np.arange(Tn)
    model._handle_override = h
    return model


def native_rccl_check(h, td, rank, world, kw, n_runs, ref, timeout=120.0):
    """The same sharded rule_n once more through the LIBRARY's own RCCL communicator (C ABI xmca_comm_* / xmca_rule_n_sharded,
    csrc/comm.h) instead of torch.distributed: the unique id of rank 0 travels through the torch group (or stays local at one
    rank), every rank calls ncclCommInitRank, and the spectra come back by ONE ncclAllGather.  Reported, never fatal: the
    attempt runs in a daemon thread with a time limit (a communicator that cannot form must not hang the bench line)."""
    import threading
    res = {}

    def work():
        try:
            from xmca_amd import _hip, dist
            uid = _hip.comm_unique_id() if rank == 0 else None
            if td is not None:
                import torch
                if td.get_backend() == "nccl":
                    torch.cuda.set_device(h.device)       # (a new thread starts on device 0: the object broadcast must use this rank's GPU)
                obj = [uid]
                td.broadcast_object_list(obj, src=0, device=torch.device("cuda", h.device) if td.get_backend() == "nccl" else None)
                uid = obj[0]
            c = _hip.Comm(h, uid, rank, world)
            t0 = time.perf_counter()
            sp, kept = dist.sharded_rule_n(h, n_runs, comm=c, **kw)
            dt = time.perf_counter() - t0
            r, w, n_coll, n_bytes = c.info()
            c.close()
            res.update(ok=True, world=int(w), collectives=int(n_coll), bytes_received=int(n_bytes), seconds=dt,
                       equals_torch_path=bool(np.array_equal(sp, ref[0]) and np.array_equal(kept, ref[1])))
        except Exception as e:                                    # noqa: BLE001
            res.update(ok=False, error=repr(e)[:300])

    th = threading.Thread(target=work, daemon=True)
    th.start()
    th.join(timeout)
    if th.is_alive():
        return {"ok": False, "error": "no result after %.0f s" % timeout, "hung": True}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--T", type=int, default=2920)
    ap.add_argument("--N", type=int, default=10_000)
    ap.add_argument("--n-rot", type=int, default=10)
    ap.add_argument("--power", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-rule-n", action="store_true")
    ap.add_argument("--no-e2e", action="store_true")
    ap.add_argument("--rule-n-runs", type=int, default=25,
                    help="timed rule_n surrogates per GPU (C4 configuration; BASELINE configs[3]: 25 runs per GPU)")
    ap.add_argument("--rule-n-rotated-runs", type=int, default=4,
                    help="runs per GPU of the ROTATED C4 variant (n_rot=20, power=4) with its dropped-run count; 0 = skip")
    ap.add_argument("--no-c5", action="store_true", help="skip the C5 covariance-GEMM roofline leg (5 GB float32 field)")
    ap.add_argument("--no-native-rccl", action="store_true",
                    help="skip the cross-check of the rule_n gather through the library's own RCCL communicator (xmca_comm_*)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    # stdout carries ONE JSON line and nothing else: RCCL prints a version banner to the C stdout of rank 0 when a communicator
    # forms (measured: five lines after the JSON line).  File descriptor 1 therefore points at stderr for the whole run and the
    # line is written to the saved descriptor at the end.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))
    import torch
    td = None
    share = os.environ.get("XMCA_BENCH_SHARE_GPU") == "1"
    backend = os.environ.get("XMCA_BENCH_BACKEND", "gloo" if share else "nccl")
    if share:
        local_rank = 0
    elif world > torch.cuda.device_count():
        sys.exit("bench.py: %d ranks but %d GPUs visible (XMCA_BENCH_SHARE_GPU=1 shares GPU 0)" % (world, torch.cuda.device_count()))
    if world > 1:
        import torch.distributed as td
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            td.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            td.init_process_group(backend=backend)
    comm_dev = "cuda" if backend == "nccl" else "cpu"

    from xmca_amd import _hip
    from xmca_amd.array import MCA
    h = _hip.Handle(local_rank)
    hung_thread = False

    T, N = args.T, args.N
    X = gen_A(T, N)
    X = X - X.mean(axis=0)                 # what MCA.__init__ hands to solve (array.py:117)
    t0 = time.perf_counter()
    h.set_field(0, X)                      # H2D outside the timed region
    upload_s = time.perf_counter() - t0

    def barrier():
        torch.cuda.synchronize()
        if td is not None:
            td.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        sig, out = device_step(h, T, N, args.n_rot, args.power, X.dtype)
    h.reset_timings()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        sig, out = device_step(h, T, N, args.n_rot, args.power, X.dtype)
    barrier()
    elapsed = time.perf_counter() - t0
    timings = h.timings()
    round_ms = timings.pop("jacobi_round_kernel_ms", 0.0)
    round_launches = timings.pop("jacobi_round_kernel_launches", 0.0)
    trd_ms = timings.pop("trd_reduce_kernel_ms", 0.0)
    trd_calls = timings.pop("trd_reduce_calls", 0.0)
    trd_resident = timings.pop("trd_resident_calls", 0.0)
    stages = {k: v / args.steps for k, v in timings.items()}
    stages["eigh_info"] = h.solve_info()[0]
    if td is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=comm_dev)
        td.all_reduce(tt, op=td.ReduceOp.MAX)
        elapsed = float(tt.item())
    ms_per_step = 1e3 * elapsed / args.steps
    value = world * args.steps / elapsed

    # ---- roofline of the dominant kernel: one launch of jacobi_fused_round_kernel per round of the block-Jacobi sweeps ----
    # algorithmic work of a round with S pair slots of NT x NT tiles: every upper off-diagonal tile G[P,Q] <- J_P^H G J_Q
    # (two NT^3 products = 4 NT^3 flop) and every eigenvector tile Z[P,c] <- J_P^H Z (one product, 2 NT^3 flop); every
    # tile is read once and written once.  Duration: hipEvents on the library's stream around the rounds of
    # every sweep of the timed region (jacobi_impl.inc), divided by the number of launches.
    info = stages["eigh_info"]
    roofline = None
    if round_launches > 0 and info.get("slots", 0) > 1:
        S, nt = info["slots"], info["tile"]
        flops_round = (S * (S - 1) // 2) * 4.0 * nt ** 3 + S * S * 2.0 * nt ** 3
        bytes_round = 2.0 * 8.0 * nt * nt * (S * (S + 1) // 2 + S * S)
        us_round = 1e3 * round_ms / round_launches
        tf = flops_round / us_round / 1e6
        roofline = {"kernel": "jacobi_fused_round_kernel<%d,real> (v_mfma_f64_16x16x4_f64; one launch per round, %d rounds per sweep)"
                              % (nt, 2 * S - 1),
                    "bound": "mfma", "achieved": tf, "peak": F64_MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": tf / F64_MFMA_PEAK_TF,
                    "traffic": None,
                    "traffic_source": "profiles/r02_pmc_c2.json holds the round-2 PMC passes of this kernel (240.6 MB per launch); not "
                                      "re-measured for this build, hence null",
                    "flops_per_launch": flops_round, "avg_launch_us": us_round, "launches_per_step": round_launches / args.steps,
                    "share_of_step": round_ms / args.steps / ms_per_step,
                    "algorithmic_bytes": bytes_round, "algorithmic_GBs": bytes_round / us_round / 1e3,
                    "frac_of_hbm_peak": bytes_round / us_round / 1e3 / 8000.0}

    # ---- ... or, on the tridiagonal route (csrc/tridiag.h, the default since round 3), the Householder reduction of the T x T
    # Gram matrix.  Useful work (one triangle of every symmetric operand): (4/3) T^3 flop of float64 VECTOR
    # FMAs (78.6 TF; the kernel has no MFMA).  It is neither flop- nor HBM-bound: the matrix lives in registers and every
    # column costs one all-to-all exchange between the 256 workgroups (`exchange_us_per_column`) - `bound` says so.
    if trd_calls > 0 and trd_ms / args.steps > (round_ms / args.steps if round_launches else 0.0):
        flops_trd = 4.0 / 3.0 * float(T) ** 3
        ms_trd = trd_ms / trd_calls
        tf = flops_trd / ms_trd / 1e9
        resident = trd_resident >= trd_calls
        # (the name comes from the library - xmca_get_reduction_info: instantiation and column range of every launch of the chain)
        roofline = {"kernel": h.reduction_info() + " (Householder tridiagonalisation of the T x T Gram matrix, matrix resident in "
                              "registers, one grid exchange per column; `avg_launch_ms` = the whole chain, hipEvents around it)",
                    "bound": "latency", "achieved": tf, "peak": F64_MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": tf / F64_MFMA_PEAK_TF,
                    "note": "useful flops (4/3) T^3 of float64 vector FMAs against the 78.6 TF float64 peak (vector = matrix "
                            "peak on MI355X); latency-bound by design: T dependent columns, each one exchange across the chip",
                    # (a reduction is a chain of launches: per-launch mean of the PMC passes x the links of the chain)
                    "traffic": (lambda t, k: None if t is None else t * k)(pmc_traffic("trd_resident_kernel", (T, N) == (2920, 10000) and resident),
                                                                          max(1, h.reduction_info().count("trd_resident_kernel<"))),
                    "launches_in_chain": max(1, h.reduction_info().count("trd_resident_kernel<")),
                    "traffic_source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command at the default workload, "
                                      "gfx950-corrected, bytes per launch: %s (scripts/r06_profiles.sh), stamped with the hash "
                                      "of xmca_amd/csrc; null when the sources differ or for any other workload" % PMC_FILE,
                    "flops_per_launch": flops_trd, "avg_launch_ms": ms_trd, "launches_per_step": trd_calls / args.steps,
                    "share_of_step": trd_ms / args.steps / ms_per_step, "exchange_us_per_column": 1e3 * ms_trd / T,
                    "algorithmic_bytes": 8.0 * T * T if resident else 16.0 * T ** 3 / 3.0,
                    "algorithmic_GBs": (8.0 * T * T if resident else 16.0 * T ** 3 / 3.0) / ms_trd / 1e6}

    # ---- the covariance (Gram) GEMM, measured live with hipEvents on the library's stream ----
    g = h.bench_gram(0, 5)
    gram_tf = g["flops"] / (g["kernel_ms"] * 1e-3) / 1e12
    roofline_gemm = {"kernel": "gemm_kernel<f64> (Gram X X^T, v_mfma_f64_16x16x4_f64, upper block triangle)", "bound": "mfma",
                     "achieved": gram_tf, "peak": F64_MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": gram_tf / F64_MFMA_PEAK_TF,
                     "traffic": pmc_traffic("gemm_kernel", (T, N) == (2920, 10000), PMC_GRAM_C2),
                     "traffic_source": "rocprofv3 --pmc passes of scripts/gram_only.py c2 (the Gram launch alone: field read through "
                                       "the Infinity Cache + split-K slabs written and read once), bytes per launch: %s" % PMC_GRAM_C2,
                     "flops_per_launch": g["flops"], "avg_launch_ms": g["kernel_ms"],
                     "product_ms_incl_reduction": g["avg_ms"], "algorithmic_bytes": 8.0 * (T * N + T * T)}

    extra = {}
    if stages.get("varimax") and out.get("n_iter"):
        extra["varimax_us_per_iteration"] = 1e3 * stages["varimax"] / out["n_iter"]
        # the second kernel of the step (since the chained reduction of round 6 the LARGEST single launch: the rocprof summary lists it
        # first): one persistent launch for the whole Varimax loop (rotation.py:52-64), per iteration one pass over the loadings, one
        # all-to-all exchange of the workgroups' p x p partials and the polar factor of their sum - latency-bound like the reduction
        p_, it_ = args.n_rot, int(out["n_iter"])
        fl_it = 4.0 * N * p_ * p_ + 10.0 * N * p_
        extra["roofline_varimax"] = {
            "kernel": "varimax_persistent_kernel<real> (whole loop in ONE launch; loadings resident in LDS, partial G exchanged write-through + "
                      "epoch flags, polar factor by scaled Newton-Schulz in one wave on MFMA accumulator-layout registers)",
            "bound": "latency", "achieved": fl_it * it_ / (stages["varimax"] * 1e-3) / 1e12, "peak": F64_MFMA_PEAK_TF, "unit": "TFLOP/s",
            "frac": fl_it * it_ / (stages["varimax"] * 1e-3) / 1e12 / F64_MFMA_PEAK_TF,
            "note": "useful flops 4 N p^2 + 10 N p per iteration (SURVEY 8d); bound by one chip-wide exchange + a p x p polar decomposition per "
                    "iteration, not by flops or bytes: see us_per_iteration",
            "iterations": it_, "us_per_iteration": 1e3 * stages["varimax"] / it_, "stage_ms": stages["varimax"],
            "share_of_step": stages["varimax"] / ms_per_step, "flops_per_iteration": fl_it, "traffic": None}
    # cheap self-check of the timed result (full parity against the oracle is in cpu_baseline/parity and tests/)
    Vt = h.vectors(0, args.n_rot, N, X.dtype)
    proj = X @ Vt.T
    extra["self_check"] = {
        "trace_identity_rel_err": float(abs(sig.sum() - (X * X).sum() / (T - 1)) / sig.sum()),
        "rayleigh_rel_err_first_modes": float(np.max(np.abs((proj * proj).sum(axis=0) / (T - 1) - sig[:args.n_rot]) / sig[:args.n_rot])),
        "orthonormality_first_modes": float(np.max(np.abs(Vt @ Vt.T - np.eye(args.n_rot)))),
    }
    # ---- PCIe-inclusive end-to-end through the drop-in class (reported, never `value`) ----
    if rank == 0 and not args.no_e2e:
        Xraw = X + 3.0                                    # an uncentered input, like a user's

        def through_class(preprocess):
            t0 = time.perf_counter()
            m = MCA(Xraw, handle=h, preprocess=preprocess)
            t1 = time.perf_counter()
            m.solve()
            t2 = time.perf_counter()
            m.rotate(args.n_rot, args.power)
            t3 = time.perf_counter()
            return m, {"ctor": 1e3 * (t1 - t0), "solve": 1e3 * (t2 - t1), "rotate": 1e3 * (t3 - t2), "total": 1e3 * (t3 - t0)}
        through_class('device')                           # (first use of the preprocessing kernels)
        m, extra["e2e_ms"] = through_class('device')      # the default path of MCA(X): raw field up, centered on the GPU
        extra["e2e_ms"]["path"] = "MCA(X) -> solve() -> rotate(): upload of the raw field + device centering + device solve/rotate"
        t3 = time.perf_counter()
        pcs = m.pcs(args.n_rot)                     # X V on the resident field (device GEMM)
        t3b = time.perf_counter()
        import scipy.stats                            # (the p-values come from scipy: keep its import out of the timing)
        t6 = time.perf_counter()
        maps = m.homogeneous_patterns(args.n_rot)     # correlation maps (device GEMM + host p-values)
        t7 = time.perf_counter()
        # what a user pays after rotate() for the arrays the reference holds in host memory after solve(): `_V` is lazy here
        # (VERDICT r05 weak #11).  eofs(n_rot): rotated EOFs mixed on the device into their final layout (xmca_get_eofs);
        # eofs(rotated=False): all `rank` modes (234 MB at C2); then the plain download of all vectors (`m._V[key]`).
        t8 = time.perf_counter()
        e1 = m.eofs(args.n_rot)
        t9 = time.perf_counter()
        e2 = m.eofs(rotated=False)
        t10 = time.perf_counter()
        vall = m._V['left']
        t11 = time.perf_counter()
        extra["e2e_ms"].update({"upload_only": 1e3 * upload_s, "pcs": 1e3 * (t3b - t3), "homogeneous_patterns": 1e3 * (t7 - t6),
                                "eofs_n_rot": 1e3 * (t9 - t8), "eofs_all_modes": 1e3 * (t10 - t9), "vectors_d2h": 1e3 * (t11 - t10),
                                "eofs_shapes": [list(e1['left'].shape), list(e2['left'].shape), list(vall.shape)]})
        del e1, e2, vall
        extra["varimax_iterations"] = m._varimax_iterations
        del pcs, maps, m
        _, extra["e2e_host_preprocess_ms"] = through_class('host')   # bit-compatible numpy constructor (preprocess='host')

    # ---- sharded rule_n (one all_gather of the spectra): the C4 configuration, a few runs per rank ----
    if not args.no_rule_n:
        Tn, Nxn, Nyn = 5000, 20000, 15000               # BASELINE.json configs[3]: rule_n on the synthetic MCA config
        model = rule_n_model(MCA, h, Tn, Nxn, Nyn)
        model.rule_n(3 * world, seed=7)               # three untimed surrogates per rank (workspaces of every lane, tile maps)
        n_runs = args.rule_n_runs * world
        h.reset_timings()
        barrier()
        t0 = time.perf_counter()
        sp = model.rule_n(n_runs, seed=1)
        barrier()
        dt = time.perf_counter() - t0
        if td is not None:
            tt = torch.tensor([dt], dtype=torch.float64, device=comm_dev)
            td.all_reduce(tt, op=td.ReduceOp.MAX)
            dt = float(tt.item())
        lanes = int(os.environ.get("XMCA_RULE_N_LANES", "2"))
        tim_rn = h.timings()
        for k in ("jacobi_round_kernel_ms", "jacobi_round_kernel_launches", "trd_reduce_kernel_ms", "trd_reduce_calls", "trd_resident_calls"):
            tim_rn.pop(k, None)
        # roofline of the rule_n shard (north_star: "... for the Rule-N shard as absolute numbers and as fraction of the HBM/MFMA
        # roofline"): the only dense contraction of a surrogate is the pair of real Gram products X X^T of its two fields,
        # T (T + 1) (Nx + Ny) useful float64 flops (one triangle each) - a surrogate cannot take less than that at
        # the float64 MFMA peak.  `frac` = that ideal time / measured seconds per surrogate and GPU.
        gram_flops = float(Tn) * (Tn + 1) * (Nxn + Nyn)
        s_per_surr = dt / args.rule_n_runs
        rn_roofline = {"bound": "mfma", "unit": "TFLOP/s", "peak": F64_MFMA_PEAK_TF,
                       "achieved": gram_flops / s_per_surr / 1e12, "frac": gram_flops / s_per_surr / 1e12 / F64_MFMA_PEAK_TF,
                       "useful_gram_flops_per_surrogate": gram_flops, "ms_per_surrogate_per_gpu": 1e3 * s_per_surr,
                       "ideal_ms_per_surrogate": 1e3 * gram_flops / (F64_MFMA_PEAK_TF * 1e12),
                       "stage_ms_per_surrogate_summed_over_lanes": {k: v / args.rule_n_runs for k, v in tim_rn.items()},
                       "note": "stage times are hipEvent sums over the %d lanes of this rank, which overlap in time: their sum "
                               "exceeds ms_per_surrogate_per_gpu by what the lanes hide" % lanes}
        n_out_rn = Tn
        cap = -(-n_runs // world)
        collective = {"backend": ("%s (torch.distributed%s)" % (backend, ": RCCL over xGMI" if backend == "nccl" else "")) if td is not None
                      else "none (single rank: no process group)", "world": world, "collectives_per_call": 2 if td is not None else 0,
                      "bytes_gathered_per_rank": 8 * cap * (n_out_rn + 1) * world if td is not None else 0}
        extra["rule_n"] = {"config": "C4: MCA T=%d x (%d, %d) f64 surrogates, complexify=True, unrotated; %d runs per GPU, "
                                     "run-sharded, one all_gather (%s); %d surrogates in flight per GPU (XMCA_RULE_N_LANES)" % (
                                         Tn, Nxn, Nyn, args.rule_n_runs, backend if world > 1 else "single rank", lanes),
                           "runs": n_runs, "runs_per_gpu": args.rule_n_runs, "lanes_per_gpu": lanes, "seconds": dt,
                           "surrogates_per_s": n_runs / dt, "shape": list(sp.shape), "roofline": rn_roofline,
                           "collective": collective}
        if not args.no_native_rccl and (backend == "nccl" or td is None):
            # two more surrogates per rank through the torch path and through the native communicator: same bits expected
            from xmca_amd import dist as _dist
            kw = dict(T=Tn, Nx=Nxn, Ny=Nyn, n_fields=2, complexify=True, rotated=False, p=Tn, power=0, tol=1e-8, seed=11,
                      dtype=np.float64, n_out=Tn)
            ref = _dist.sharded_rule_n(h, 2 * world, **kw)
            collective["native_rccl"] = native_rccl_check(h, td, rank, world, kw, 2 * world, ref)
            if collective["native_rccl"].get("hung"):
                hung_thread = True
        # parity of the timed runs: surrogate (seed 1, run 0) went through the REAL reference once (oracle/make_config_goldens.py
        # c4_run0 on the numpy restatement of the device generator, tests/golden/rule_n_c4_run0.npz).  `sp` is normalised
        # per run (array.py:1767-1769: each column sums to the model's total), so the reference column is normalised alike.
        gpath = os.path.join(REPO, "tests", "golden", "rule_n_c4_run0.npz")
        if rank == 0 and os.path.exists(gpath):
            g0 = np.load(gpath)
            if (int(g0["T"]), int(g0["widths"][0]), int(g0["widths"][1]), int(g0["seed"])) == (Tn, Nxn, Nyn, 1):
                ref0 = g0["variance"] / g0["variance"].sum() * model._get_variance().sum()
                keep = g0["variance"] > 1e-7 * g0["variance"][0]
                extra["rule_n"]["run0_vs_reference"] = {
                    "max_rel_err_nonnull_modes": float(np.max(np.abs(sp[keep, 0] - ref0[keep]) / ref0[keep])),
                    "modes": int(keep.sum()), "reference": "xmca.array.MCA on the same normals (392 s on 8 cores)"}
        if args.rule_n_rotated_runs > 0:
            n_rot_runs = args.rule_n_rotated_runs * world
            model._analysis.update({'is_rotated': True, 'n_rot': 20, 'power': 4})
            model._norm = {'left': np.ones(20), 'right': np.ones(20)}
            model._var_idx = np.arange(20)
            barrier()
            t0 = time.perf_counter()
            spr = model.rule_n(n_rot_runs, seed=1)
            barrier()
            dtr = time.perf_counter() - t0
            extra["rule_n_rotated"] = {"config": "C4 rotated: n_rot=20, power=4 (complex white noise: Varimax rarely converges in "
                                                 "1000 iterations; the reference drops those runs, array.py:1759-1763)",
                                       "runs": n_rot_runs, "kept": int(spr.shape[1]), "dropped": int(n_rot_runs - spr.shape[1]),
                                       "surrogates_per_s": n_rot_runs / dtr}

    # ---- BASELINE configs[4]: the covariance GEMM at the large-grid size, T = 1200 x N = 1 036 800 float32 (4.98 GB resident) --
    # The Gram matrix X X^T of the dual formulation (xmca/array.py:479 on the float32 field), T (T + 1) N useful flops (upper
    # block triangle), f32 MFMA, slices added in float64.  hipEvents on the library's stream; 3 products.
    if rank == 0 and world == 1 and not args.no_c5:
        try:
            T5, N5 = 1200, 1_036_800
            rng5 = np.random.default_rng(5)
            X5 = rng5.standard_normal((T5, N5), dtype=np.float32)
            h5 = _hip.Handle(local_rank)
            h5.set_field(0, X5)
            del X5
            h5.bench_gram(0, 1)
            g5 = h5.bench_gram(0, 3)
            tf5 = g5["flops"] / (g5["kernel_ms"] * 1e-3) / 1e12
            x3 = os.environ.get("XMCA_GEMM_BF16X3", "1") != "0"
            extra["roofline_c5"] = {"kernel": "gemm_kernel<f32> (Gram X X^T of the resident T=1200 x N=1036800 float32 field, "
                                              + ("each float32 split into three bfloat16 pieces in registers, six "
                                                 "v_mfma_f32_32x32x16_bf16 terms per product" if x3 else "v_mfma_f32_16x16x4_f32")
                                              + ", k-slices of <= 16384 products added in float64 inside the launch; achieved / "
                                              "peak count the USEFUL float32 flops against the float32 MFMA peak)",
                                    "bound": "mfma", "bf16x3": x3,
                                    "mfma_flops_issued_per_launch": g5["flops"] * (6.0 if x3 else 1.0),
                                    "achieved": tf5, "peak": F32_MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": tf5 / F32_MFMA_PEAK_TF,
                                    # the same launch against the peak of the instruction it issues: six bf16
                                    # products per useful float32 product against the dense bf16 MFMA peak
                                    "frac_issued_vs_bf16_peak": (tf5 * 6.0 / BF16_MFMA_PEAK_TF) if x3 else None,
                                    "peak_bf16": BF16_MFMA_PEAK_TF,
                                    "traffic": pmc_traffic("gemm_kernel", True, PMC_GRAM_C5),
                                    "traffic_source": "rocprofv3 --pmc passes of scripts/gram_only.py c5, bytes per launch: %s" % PMC_GRAM_C5,
                                    "flops_per_launch": g5["flops"], "avg_launch_ms": g5["kernel_ms"],
                                    "product_ms_incl_reduction": g5["avg_ms"], "algorithmic_bytes": 4.0 * T5 * N5 + 8.0 * T5 * T5}
            # ---- ... and the rotation at that grid size: Varimax / Promax of 10 modes on N = 1 036 800 points.  A
            # Varimax iteration is ONE pass over the p x N float64 planes (83 MB: HBM / Infinity-Cache bound, rotation.py:54-60).
            try:
                p5 = args.n_rot
                L5 = (0.2 * rng5.standard_normal((N5, p5), dtype=np.float32)).astype(np.float64)
                w5 = N5 // p5
                for j in range(p5):
                    L5[j * w5:(j + 1) * w5, j] += np.hanning(w5) * (3 - 0.1 * j)
                Q5, _ = np.linalg.qr(rng5.standard_normal((p5, p5)))
                L5 = L5 @ Q5
                h5.rotate_loadings(L5, n_left=N5, power=args.power)        # (first use: workspaces)
                h5.reset_timings()
                o5 = h5.rotate_loadings(L5, n_left=N5, power=args.power)
                t5 = h5.timings()
                us_it = 1e3 * t5.get("varimax", 0.0) / max(o5["n_iter"], 1)
                gbs = 8.0 * p5 * N5 / (us_it * 1e-6) / 1e9 if us_it > 0 else 0.0
                extra["roofline_rotate_c5"] = {
                    "kernel": "varimax_persistent_kernel<real> (whole Varimax loop in one launch; per iteration one pass over the %d x %d "
                              "float64 loading planes, prefetched 64-point tiles, MFMA accumulation, two-stage sum of 256 partials)" % (p5, N5),
                    "bound": "hbm", "achieved": gbs, "peak": 6300.0, "unit": "GB/s", "frac": gbs / 6300.0,
                    "note": "peak = achievable HBM rate (MI355X_MICROARCH.md); the 83 MB of planes also fit the Infinity Cache; `varimax_ms` "
                            "includes the Gram pass and the set-up of the loop",
                    "varimax_ms": t5.get("varimax"), "promax_ms": t5.get("promax"), "iterations": int(o5["n_iter"]),
                    "us_per_iteration": us_it, "algorithmic_bytes_per_iteration": 8.0 * p5 * N5}
                del L5
            except Exception as e:                                # noqa: BLE001
                extra["roofline_rotate_c5"] = {"error": repr(e)[:300]}
            del h5
        except Exception as e:                                    # noqa: BLE001  (reported, never fatal for the headline)
            extra["roofline_c5"] = {"error": repr(e)[:300]}

    # ---- CPU baseline: the numpy oracle (formula-identical to the reference) on this host, N = 1 only ----
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import ref_numpy as O
        t0 = time.perf_counter()
        o = O.solve([X])
        t1 = time.perf_counter()
        r = O.rotate(o["V"], o["singular_values"], args.n_rot, args.power)
        t2 = time.perf_counter()
        cpu = {"value": 1.0 / (t2 - t0), "unit": "solve+rotate/s", "cores": HOST_THREADS, "kind": "port",
               "sample": "1 full solve()+rotate() of the same C2 input (numpy oracle: gesdd per field, kernel, gesdd, "
                         "back-projection, Varimax loop)",
               "solve_ms": 1e3 * (t1 - t0), "rotate_ms": 1e3 * (t2 - t1), "varimax_iterations": int(r["n_iter"])}
        # parity of this very run (sign-aligned leading loadings, singular values)
        ks = 20
        V10 = o["V"][0][:, :args.n_rot]
        ph = np.sign(np.sum(V10 * Vt.T, axis=0))
        extra["parity"] = {"sigma_rel_err_first20": float(np.max(np.abs(sig[:ks] - o["singular_values"][:ks])
                                                                 / o["singular_values"][:ks])),
                           "sigma_rel_err_all_nonnull": float(np.max(np.abs(sig[:T - 1] - o["singular_values"][:T - 1])
                                                                     / o["singular_values"][:T - 1])),
                           "loadings_max_err_first_modes": float(np.max(np.abs(Vt.T * ph - V10)) / np.max(np.abs(V10))),
                           "iterations_equal": bool(out["n_iter"] == r["n_iter"])}
        del o, r
        if not args.no_rule_n:
            # rule_n CPU baseline (time n_runs = 2, extrapolate linearly): two surrogates of the C4 body at a
            # quarter of every dimension (T = 1250 x (5000, 3750), complexify) - normals, constructor, solve: the reference's
            # per-run work (array.py:1755-1764) - because two at full size take ~13 minutes on this host (392 s each in the
            # build container, tests/golden/rule_n_c4_run0.npz).  Linear in the number of runs; one run scales ~T^2 N = x64.
            Tq, Nxq, Nyq = 1250, 5000, 3750
            per_run = []
            for _ in range(2):
                t0 = time.perf_counter()
                data = [np.random.standard_normal([Tq, Nxq]), np.random.standard_normal([Tq, Nyq])]
                om = O.OracleModel(*data)
                om.solve(complexify=True)
                om.variance()
                per_run.append(time.perf_counter() - t0)
                del om, data
            dtq = float(np.mean(per_run))
            cpu["rule_n"] = {"sample": "n_runs = 2 at T=%d x (%d, %d), complexify (1/4 of every C4 dimension)" % (Tq, Nxq, Nyq),
                             "seconds_per_run": per_run, "extrapolation": "linear in n_runs; x64 per run to full size (gesdd ~ T^2 N)",
                             "rule_n_200_seconds_at_sample_size": 200.0 * dtq,
                             "extrapolated_full_size_seconds_per_run": 64.0 * dtq,
                             "measured_full_size_seconds_per_run_build_container_8_cores": 392.1,
                             "surrogates_per_s_full_size": 1.0 / (64.0 * dtq)}

    if rank == 0:
        try:        # persistent launches of this process that ran out of their bounded spins and were repeated (0 on a GPU of its own)
            extra["persistent_giveups"] = int(_hip.load_library().xmca_persistent_giveups())
        except Exception:                                         # noqa: BLE001
            pass
        line = {
            "metric": "MCA solve+rotate throughput (ms_per_step = solve+rotate wall-clock, ms); rule_n surrogates/s in `rule_n`",
            "value": value, "unit": "solve+rotate/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "C2: synthetic EOF T=%d x N=%d float64 (generator A), solve() + rotate(n_rot=%d, power=%d)"
                                   % (T, N, args.n_rot, args.power),
                       "parallelism": "replicas x%d (solve/rotate do not shard); rule_n run-sharded x%d" % (world, world)},
            "rule_n_surrogates_per_s": extra.get("rule_n", {}).get("surrogates_per_s"),
            "stages_ms": stages, "roofline": roofline, "roofline_gemm": roofline_gemm, "cpu_baseline": cpu,
        }
        line.update(extra)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    if hung_thread:
        os._exit(0)            # (a native communicator that never formed holds a thread inside RCCL: leave without joining it)
    if td is not None:
        td.destroy_process_group()


if __name__ == "__main__":
    main()
