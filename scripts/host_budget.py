#!/usr/bin/env python3
"""Host-CPU budget of the run-sharded rule_n (VERDICT r04 weak #7 / next #3b): the only evidence for the 8-GPU target
(array.py:1753-1771 sharded, >= 6x at 8 vs 1 GPU) that one GPU can give.

The GPU boxes grant 16 host CPUs (cgroup quota); an 8-GPU node runs 8 ranks x 3 surrogate lanes + the launcher.  Measured here:

  1. `pinned`:  one rank, C4 rule_n (8 timed surrogates after 3 warm-ups), the process pinned to 2 CPUs
                (os.sched_setaffinity) with XMCA_BLOCKING_SYNC=1 -> surrogates/s and getrusage CPU-seconds per surrogate,
                next to the same run unpinned (all granted CPUs) -> the loss at 2 CPUs;
  2. `ranks8`:  8 ranks on GPU 0 (XMCA_BENCH_SHARE_GPU-style: gloo gather, every rank its own process and handle), 2 surrogates
                each at C4 size -> aggregate host CPU-seconds per surrogate with 8 processes' worth of threads alive (the GPU
                is shared 8 ways, so surrogates/s of this leg says nothing - CPU-seconds per surrogate is the figure).

    python scripts/host_budget.py [--out profiles/r05_host_budget.json] [--small]

Acceptance (VERDICT): <= 40 ms host CPU per surrogate (8 ranks x 21/s x 0.04 = 6.7 of the 16 granted cores) and < 5 % loss at
2 CPUs.  `--small` runs a T = 1000 x (4000, 3000) stand-in (the -m gpu test uses it: seconds instead of a minute).
"""
import argparse
import json
import os
import resource
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import json, os, resource, sys, time
cpus = os.environ.get("HB_CPUS")
if cpus:
    os.sched_setaffinity(0, set(int(c) for c in cpus.split(",")))
for v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(v, "2")
import numpy as np
sys.path.insert(0, %(repo)r)
from xmca_amd import _hip
T, Nx, Ny, warm, runs = [int(v) for v in sys.argv[1:6]]
h = _hip.Handle(0)
m = T // 2 + 1
args = (T, Nx, Ny, 2, True, False, 0, 1, 1e-8)
h.rule_n(*args, 0, warm, 7, np.float64, T)
r0 = resource.getrusage(resource.RUSAGE_SELF)
t0 = time.perf_counter()
sp, kept = h.rule_n(*args, 0, runs, 1, np.float64, T)
dt = time.perf_counter() - t0
r1 = resource.getrusage(resource.RUSAGE_SELF)
cpu = (r1.ru_utime - r0.ru_utime) + (r1.ru_stime - r0.ru_stime)
print(json.dumps({"runs": runs, "seconds": dt, "surrogates_per_s": runs / dt, "cpu_seconds": cpu,
                  "cpu_ms_per_surrogate": 1e3 * cpu / runs, "user_s": r1.ru_utime - r0.ru_utime, "sys_s": r1.ru_stime - r0.ru_stime,
                  "host_cores_busy": cpu / dt, "affinity": sorted(os.sched_getaffinity(0))[:8], "n_affinity": len(os.sched_getaffinity(0)),
                  "giveups": int(_hip.load_library().xmca_persistent_giveups()), "checksum": float(sp.sum())}))
"""


def run_worker(T, Nx, Ny, warm, runs, env_extra):
    env = dict(os.environ, XMCA_BLOCKING_SYNC="1")
    env.update(env_extra)
    r = subprocess.run([sys.executable, "-c", WORKER % {"repo": REPO}, str(T), str(Nx), str(Ny), str(warm), str(runs)], env=env,
                       capture_output=True, text=True, timeout=1800)
    if r.returncode != 0:
        raise RuntimeError(r.stderr[-2000:])
    return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--small", action="store_true")
    ap.add_argument("--ranks", type=int, default=8)
    args = ap.parse_args()
    T, Nx, Ny = (1000, 4000, 3000) if args.small else (5000, 20000, 15000)
    avail = sorted(os.sched_getaffinity(0))
    two = ",".join(str(c) for c in avail[:2])
    out = {"config": "rule_n surrogates T=%d x (%d, %d) float64, complexify, unrotated (C4%s), XMCA_BLOCKING_SYNC=1"
                     % (T, Nx, Ny, " stand-in" if args.small else ""),
           "host": {"cpu_count": os.cpu_count(), "affinity": len(avail)}}
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            out["host"]["cgroup_cpu_max"] = f.read().strip()
    except OSError:
        pass
    out["unpinned"] = run_worker(T, Nx, Ny, 3, 8, {})
    out["pinned_2_cpus"] = run_worker(T, Nx, Ny, 3, 8, {"HB_CPUS": two})
    out["spinning_sync_unpinned"] = run_worker(T, Nx, Ny, 3, 8, {"XMCA_BLOCKING_SYNC": "0"})
    out["loss_at_2_cpus"] = 1.0 - out["pinned_2_cpus"]["surrogates_per_s"] / out["unpinned"]["surrogates_per_s"]
    # ---- 8 processes sharing GPU 0: aggregate CPU-seconds per surrogate ----
    t0 = time.perf_counter()
    procs = []
    for r in range(args.ranks):
        env = dict(os.environ, XMCA_BLOCKING_SYNC="1")
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER % {"repo": REPO}, str(T), str(Nx), str(Ny), "1", "2"], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    rows = []
    for p in procs:
        so, se = p.communicate(timeout=3600)
        if p.returncode != 0:
            raise RuntimeError(se[-2000:])
        rows.append(json.loads([ln for ln in so.splitlines() if ln.startswith("{")][-1]))
    wall = time.perf_counter() - t0
    n = sum(r["runs"] for r in rows)
    cpu = sum(r["cpu_seconds"] for r in rows)
    out["ranks_sharing_gpu0"] = {"ranks": args.ranks, "surrogates": n, "wall_seconds_incl_process_start": wall,
                                 "cpu_seconds_timed_regions": cpu, "cpu_ms_per_surrogate": 1e3 * cpu / n,
                                 "giveups": sum(r["giveups"] for r in rows),
                                 "per_rank_cpu_ms_per_surrogate": [r["cpu_ms_per_surrogate"] for r in rows],
                                 "note": "the GPU is shared %d ways (persistent kernels of different processes can make each other "
                                         "give up and repeat launch by launch): CPU-seconds per surrogate is the figure, not the rate" % args.ranks}
    per = out["pinned_2_cpus"]["cpu_ms_per_surrogate"]
    rate1 = out["unpinned"]["surrogates_per_s"]
    out["budget_8_gpus"] = {"cpu_ms_per_surrogate_one_rank": out["unpinned"]["cpu_ms_per_surrogate"],
                            "cores_needed_for_8_ranks_at_single_gpu_rate": 8 * rate1 * out["unpinned"]["cpu_ms_per_surrogate"] * 1e-3,
                            "cores_granted": 16, "cpu_ms_per_surrogate_pinned": per}
    txt = json.dumps(out, indent=1)
    print(txt)
    if args.out:
        with open(args.out, "w") as f:
            f.write(txt + "\n")


if __name__ == "__main__":
    main()
