#!/usr/bin/env python3
"""Headline benchmark: MCA.solve() + MCA.rotate() on the BASELINE.json configuration C2
(synthetic EOF, T = 2920 x N = 10 000 grid points, float64, rotate(n_rot=10, power=1)), one MI355X per rank.

    python bench.py --gpus N --steps K --warmup W

* a "step" = one pass of the hot path: device solve (Gram GEMM, eigensolver, back-projection of all `rank`
  modes) + Varimax/Promax rotation, with the centered field ALREADY RESIDENT in HBM when the timed region starts
  (the PCIe-inclusive time through the MCA class is reported separately as `e2e_ms`, never as `value`).
* N > 1: every rank processes its own replica of the workload on its own GPU (the path has no exchange step;
  SURVEY.md 8e) -> weak scaling; `value` = steps of all ranks / wall time.  The sharded rule_n (the only
  collective of the path: one all_gather of the spectra) runs after the timed region on the C4 configuration and
  is reported as `rule_n.surrogates_per_s` (all ranks together).
* prints ONE JSON line on rank 0 with the driver's contract keys plus `roofline` (Gram GEMM, measured with
  hipEvents on the library's stream) and `cpu_baseline` (the numpy oracle timed on this host, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

F64_MFMA_PEAK_TF = 78.6     # MI355X FP64 matrix (= FP64 vector) peak, AMD CDNA4 datasheet; 256 CU * 4 SIMD * 32 FLOP/clk * 2.4 GHz
F32_MFMA_PEAK_TF = 157.3    # MI355X_MICROARCH.md, "Peak FP32 (matrix)"


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
    ap.add_argument("--rule-n-runs", type=int, default=3, help="timed rule_n surrogates per GPU (C4 configuration)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    td = None
    # XMCA_BENCH_BACKEND=gloo (+ XMCA_BENCH_SHARE_GPU=1: every rank on GPU 0) exercises the multi-rank flow on a 1-GPU box
    backend = os.environ.get("XMCA_BENCH_BACKEND", "nccl")
    if os.environ.get("XMCA_BENCH_SHARE_GPU") == "1":
        local_rank = 0
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
    stages = {k: v / args.steps for k, v in h.timings().items()}
    stages["eigh_info"] = h.solve_info()[0]
    if td is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=comm_dev)
        td.all_reduce(tt, op=td.ReduceOp.MAX)
        elapsed = float(tt.item())
    ms_per_step = 1e3 * elapsed / args.steps
    value = world * args.steps / elapsed

    # ---- roofline of the covariance (Gram) GEMM, measured live with hipEvents on the library's stream ----
    # achieved = algorithmic flops T(T+1)N of one Gram product / average duration of the MFMA kernel launch
    # (hipEvents recorded around that launch on the library's stream); the product including its split-K reduction
    # is reported next to it.  traffic: HBM-side bytes per launch of the same kernel on the same workload from
    # rocprofv3 PMC passes (profiles/r01_pmc_gram_c2.json: (2*FETCH_SIZE + WRITE_SIZE) * 1024, gfx950 correction).
    g = h.bench_gram(0, 5)
    gram_tf = g["flops"] / (g["kernel_ms"] * 1e-3) / 1e12
    default_workload = (T, N) == (2920, 10_000)
    roofline = {"kernel": "gemm_kernel<f64> (Gram X X^T, v_mfma_f64_16x16x4_f64, upper block triangle)", "bound": "mfma",
                "achieved": gram_tf, "peak": F64_MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": gram_tf / F64_MFMA_PEAK_TF,
                "traffic": 2.456e9 if default_workload else None, "flops_per_launch": g["flops"],
                "avg_launch_ms": g["kernel_ms"], "product_ms_incl_splitk_reduce": g["avg_ms"],
                "algorithmic_bytes": 8.0 * (T * N + T * T)}

    extra = {}
    # ---- the T x T eigensolver ("latency bound; report ms and achieved GF/s informally") ----
    # one launch of jacobi_fused_round_kernel per round; per round every upper tile of G (two 64^3 products) and every
    # eigenvector tile (one product) is read and written once
    info = stages["eigh_info"]
    if info.get("slots", 0) > 1 and info.get("sweeps", 0) > 0:
        S, nt = info["slots"], info["tile"]
        rounds = info["sweeps"] * (2 * S - 1)
        flops_round = (S * (S - 1) // 2) * 2 * 2.0 * nt ** 3 + S * S * 2.0 * nt ** 3
        bytes_round = 2.0 * 8.0 * nt * nt * (S * (S + 1) // 2 + S * S)
        us_round = 1e3 * stages["eigh"] / rounds
        extra["eigensolver"] = {"kernel": "jacobi_fused_round_kernel<64,real> (one launch per round)", "rounds": rounds,
                                "us_per_round_incl_init_and_gather": us_round,
                                "algorithmic_flops_per_round": flops_round, "achieved_TFLOPs": flops_round / us_round / 1e6,
                                "frac_of_f64_mfma_peak": flops_round / us_round / 1e6 / F64_MFMA_PEAK_TF,
                                "algorithmic_bytes_per_round": bytes_round, "achieved_GBs": bytes_round / us_round / 1e3,
                                "frac_of_hbm_peak": bytes_round / us_round / 1e3 / 8000.0}
    if stages.get("varimax") and out.get("n_iter"):
        extra["varimax_us_per_iteration"] = 1e3 * stages["varimax"] / out["n_iter"]
    # cheap self-check of the timed result (full parity against the oracle is in cpu_baseline/parity and tests/)
    Vt = h.vectors(0, args.n_rot, N, X.dtype)
    proj = X @ Vt.T
    extra["self_check"] = {
        "trace_identity_rel_err": float(abs(sig.sum() - (X * X).sum() / (T - 1)) / sig.sum()),
        "rayleigh_rel_err_first_modes": float(np.max(np.abs((proj * proj).sum(axis=0) / (T - 1) - sig[:args.n_rot]) / sig[:args.n_rot])),
        "orthonormality_first_modes": float(np.max(np.abs(Vt @ Vt.T - np.eye(args.n_rot)))),
    }
    # ---- PCIe-inclusive end-to-end through the drop-in class (reported, never `value`) ----
    if rank == 0:
        t0 = time.perf_counter()
        m = MCA(X, handle=h)
        t1 = time.perf_counter()
        m.solve()
        t2 = time.perf_counter()
        m.rotate(args.n_rot, args.power)
        t3 = time.perf_counter()
        pcs = m.pcs(args.n_rot)                     # X V on the resident field (device GEMM)
        t3b = time.perf_counter()
        Vh = m._V.head('left', args.n_rot)            # (leading modes only: the vectors stay on the device until read)
        t4 = time.perf_counter()
        host_pcs = X @ Vh                             # the reference's host product, for scale
        t5 = time.perf_counter()
        import scipy.stats                            # (the p-values come from scipy: keep its import out of the timing)
        t6 = time.perf_counter()
        maps = m.homogeneous_patterns(args.n_rot)     # correlation maps (device GEMM + host p-values)
        t7 = time.perf_counter()
        extra["e2e_ms"] = {"ctor": 1e3 * (t1 - t0), "solve": 1e3 * (t2 - t1), "rotate": 1e3 * (t3 - t2),
                           "upload_only": 1e3 * upload_s, "pcs_device": 1e3 * (t3b - t3), "pcs_host_product_only": 1e3 * (t5 - t4),
                           "homogeneous_patterns": 1e3 * (t7 - t6)}
        del pcs, host_pcs, maps
        # the same through MCA(..., preprocess='device') (centering / mean / std on the GPU)
        t0 = time.perf_counter()
        md = MCA(X, handle=h, preprocess='device')
        t1 = time.perf_counter()
        md.solve()
        t2 = time.perf_counter()
        md.rotate(args.n_rot, args.power)
        t3 = time.perf_counter()
        extra["e2e_device_preprocess_ms"] = {"ctor": 1e3 * (t1 - t0), "solve": 1e3 * (t2 - t1), "rotate": 1e3 * (t3 - t2)}
        del md
        extra["varimax_iterations"] = m._varimax_iterations

    # ---- sharded rule_n (one all_gather of the spectra): the C4 configuration, a few runs per rank ----
    if not args.no_rule_n:
        Tn, Nxn, Nyn = 5000, 20000, 15000               # BASELINE.json configs[3]: rule_n on the synthetic MCA config
        model = MCA.__new__(MCA)                      # only the meta data rule_n reads is needed
        MCA.__init__(model)
        model._keys = ['left', 'right']
        model._n_observations = {'left': Tn, 'right': Tn}
        model._n_variables = {'left': Nxn, 'right': Nyn}
        model._analysis.update({'is_bivariate': True, 'is_complex': True, 'rank': Tn, 'n_rot': Tn})
        model._norm = {'left': np.ones(Tn), 'right': np.ones(Tn)}
        model._var_idx = np.arange(Tn)
        model._handle_override = h
        model.rule_n(world, seed=7)                   # one untimed surrogate per rank (workspaces, tile maps)
        n_runs = args.rule_n_runs * world
        barrier()
        t0 = time.perf_counter()
        sp = model.rule_n(n_runs, seed=1)
        barrier()
        dt = time.perf_counter() - t0
        extra["rule_n"] = {"config": "C4: MCA T=5000 x (20000, 15000) f64 surrogates, complexify=True, unrotated; %d runs per GPU" % args.rule_n_runs,
                           "runs": n_runs, "surrogates_per_s": n_runs / dt, "shape": list(sp.shape)}

    # ---- CPU baseline: the numpy oracle (formula-identical to the reference) on this host, N = 1 only ----
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import ref_numpy as O
        t0 = time.perf_counter()
        o = O.solve([X])
        t1 = time.perf_counter()
        r = O.rotate(o["V"], o["singular_values"], args.n_rot, args.power)
        t2 = time.perf_counter()
        cpu = {"value": 1.0 / (t2 - t0), "unit": "solve+rotate/s", "cores": os.cpu_count(), "kind": "port",
               "sample": "1 full solve()+rotate() of the same C2 input (numpy oracle: gesdd per field, kernel, gesdd, "
                         "back-projection, Varimax loop)",
               "solve_ms": 1e3 * (t1 - t0), "rotate_ms": 1e3 * (t2 - t1), "varimax_iterations": int(r["n_iter"])}
        # parity of this very run (sign-aligned leading loadings, singular values)
        ks = 20
        extra["parity"] = {"sigma_rel_err_first20": float(np.max(np.abs(sig[:ks] - o["singular_values"][:ks])
                                                                 / o["singular_values"][:ks])),
                           "iterations_equal": bool(out["n_iter"] == r["n_iter"])}

    if rank == 0:
        line = {
            "metric": "MCA solve+rotate throughput (ms_per_step = solve+rotate wall-clock, ms)",
            "value": value, "unit": "solve+rotate/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "C2: synthetic EOF T=%d x N=%d float64 (generator A), solve() + rotate(n_rot=%d, power=%d)"
                                   % (T, N, args.n_rot, args.power),
                       "parallelism": "replicas x%d (rule_n run-sharding exercised separately)" % world},
            "stages_ms": stages, "roofline": roofline, "cpu_baseline": cpu,
        }
        line.update(extra)
        print(json.dumps(line))
    if td is not None:
        td.destroy_process_group()


if __name__ == "__main__":
    main()
