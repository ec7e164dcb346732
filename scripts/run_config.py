#!/usr/bin/env python3
"""Runs one of the BASELINE.json configurations through the drop-in class and prints timings + size-independent
self checks (trace identity, Rayleigh quotients, orthonormality, SVD consistency  X_l^H X_r v = sigma u (T-1))."""
import argparse
import json
import sys
import time
import os

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def gen_B(T=5000, Nx=20_000, Ny=15_000, k=40, seed=3, geometric=True):
    rng = np.random.default_rng(seed)

    def pat(N):
        P = np.zeros((k, N)); w = N // k
        for j in range(k): P[j, j * w:(j + 1) * w] = np.hanning(w)
        return rng.standard_normal((k, k)) @ P * 0.3 + P
    t = np.arange(T)[:, None]; f = np.linspace(0.01, 0.2, k)[None, :]
    amp = 10 * 0.85 ** np.arange(k) if geometric else np.linspace(10, 5, k)
    pcs = amp * np.cos(2 * np.pi * f * t + rng.uniform(0, 6.28, (1, k)))
    A = pcs @ pat(Nx) + 0.5 * rng.standard_normal((T, Nx))
    B = pcs @ pat(Ny) + 0.5 * rng.standard_normal((T, Ny))
    return A, B


def gen_C(T=1200, ny=720, nx=1440, k=30, seed=5):
    rng = np.random.default_rng(seed); N = ny * nx
    X = rng.standard_normal((T, N), dtype=np.float32) * np.float32(0.5)
    P = np.zeros((k, N), dtype=np.float32); w = N // k
    for j in range(k): P[j, j * w:(j + 1) * w] = np.hanning(w)
    pcs = (rng.standard_normal((T, k)) * np.linspace(10, 5, k)).astype(np.float32)
    X += pcs @ P
    return X.reshape(T, ny, nx)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("config", choices=["C3", "C5", "C3small"])
    ap.add_argument("--scale", type=float, default=1.0)
    a = ap.parse_args()
    from xmca_amd import _hip
    from xmca_amd.array import MCA
    h = _hip.Handle(0)
    out = {"config": a.config}
    if a.config in ("C3", "C3small"):
        T, Nx, Ny = (5000, 20000, 15000) if a.config == "C3" else (1000, 4000, 3000)
        t0 = time.perf_counter(); A, B = gen_B(T, Nx, Ny); out["gen_s"] = time.perf_counter() - t0
        t0 = time.perf_counter(); m = MCA(A, B, handle=h); out["ctor_s"] = time.perf_counter() - t0
        # the first solve of a process pays for first launches and its device buffers (~10 ms at this size); later ones run out
        # of the handle's pool - first and steady state are both reported
        h.reset_timings()
        t0 = time.perf_counter(); m.solve(complexify=True); out["first_solve_s"] = time.perf_counter() - t0
        out["first_solve_stages_ms"] = h.timings()
        s_first = m._singular_values.copy()
        # (the second solve of a handle creates the eigensolver's second stream - ~10 ms once, csrc/tridiag_vec.h
        #  trd_wy_prepare; the third is the steady state of a process that solves repeatedly)
        t0 = time.perf_counter(); m.solve(complexify=True); out["second_solve_s"] = time.perf_counter() - t0
        h.reset_timings()
        t0 = time.perf_counter(); m.solve(complexify=True); out["solve_s"] = time.perf_counter() - t0
        out["stages_ms"] = h.timings(); out["evd"] = h.solve_info()
        out["second_solve_equals_first"] = bool(np.array_equal(s_first, m._singular_values))
        h.reset_timings()
        t0 = time.perf_counter(); m.rotate(20, 4); out["rotate_s"] = time.perf_counter() - t0
        out["rotate_stages_ms"] = h.timings()
        out["varimax_iterations"] = m._varimax_iterations
        k = 10
        t0 = time.perf_counter()
        X = m._fields                      # host analytic signal (lazy)
        out["host_hilbert_s"] = time.perf_counter() - t0
        Vl, Vr, s = m._V['left'][:, :k], m._V['right'][:, :k], m._singular_values[:k]
        Ul, Ur = X['left'] @ Vl, X['right'] @ Vr
        cov = Ul.conj().T @ Ur / (T - 1)
        out["check"] = {"diag_cov_rel_err": float(np.max(np.abs(np.diag(cov) - s) / s)),
                        "offdiag_cov_rel": float(np.max(np.abs(cov - np.diag(np.diag(cov)))) / s[0]),
                        "orth_left": float(np.max(np.abs(Vl.conj().T @ Vl - np.eye(k)))),
                        "orth_right": float(np.max(np.abs(Vr.conj().T @ Vr - np.eye(k)))),
                        "sigma_head": [float(x) for x in s[:5]], "rank": int(m._analysis['rank'])}
    else:
        T = 1200
        ny, nx = (720, 1440) if a.scale >= 1 else (int(720 * a.scale), int(1440 * a.scale))
        t0 = time.perf_counter(); X = gen_C(T, ny, nx); out["gen_s"] = time.perf_counter() - t0
        t0 = time.perf_counter(); m = MCA(X, handle=h); out["ctor_s"] = time.perf_counter() - t0
        del X
        h.reset_timings()
        t0 = time.perf_counter(); m.solve(); out["solve_s"] = time.perf_counter() - t0
        out["stages_ms"] = h.timings(); out["evd"] = h.solve_info()
        t0 = time.perf_counter(); m.rotate(10, 1); out["rotate_s"] = time.perf_counter() - t0
        out["varimax_iterations"] = m._varimax_iterations
        k = 10
        F = m._fields['left']
        V, s = m._V['left'][:, :k], m._singular_values[:k].astype(np.float64)
        U = F.astype(np.float64) @ V.astype(np.float64)
        lam = (U * U).sum(axis=0) / (T - 1)
        out["check"] = {"rayleigh_rel_err": float(np.max(np.abs(lam - s) / s)),
                        "orth": float(np.max(np.abs(V.T.astype(np.float64) @ V.astype(np.float64) - np.eye(k)))),
                        "trace_rel_err": float(abs(m._singular_values.astype(np.float64).sum() - (F.astype(np.float64) ** 2).sum() / (T - 1))
                                               / m._singular_values.astype(np.float64).sum()),
                        "dtype": str(m._V['left'].dtype), "rank": int(m._analysis['rank'])}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
