#!/usr/bin/env python3
"""Randomised parity sweep of MCA.solve against the numpy oracle: random shapes around the route boundaries (N <> T, analytic
/ general, FFT-able T or not), rank-deficient and duplicated columns, graded amplitudes, f32 / f64, one or two fields.
Prints one line per failure and a summary; exit code 1 when anything failed.
usage: fuzz_solve.py [n_cases] [seed] [T,T,...]   (default sizes: single-tile eigenproblems; e.g. 300,640,1000 for the
multi-tile route heuristics of csrc/solver.h)"""
import os, sys, json
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from conftest import align_modes
from oracle import ref_numpy as O
from xmca_amd.array import MCA

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
T_CHOICES = [int(t) for t in sys.argv[3].split(",")] if len(sys.argv) > 3 else [8, 12, 30, 31, 48, 60, 62, 97, 100, 120]
fails = 0
routes = {}
for case in range(n_cases):
    T = int(rng.choice(T_CHOICES))
    two = rng.random() < 0.7
    cplx = rng.random() < 0.5
    f32 = rng.random() < 0.25
    kind = rng.choice(["noise", "signal", "graded", "lowrank", "dupcols"])
    def field(N):
        if kind == "noise":
            x = rng.standard_normal((T, N))
        elif kind == "signal":
            k = min(5, N, T)
            x = (rng.standard_normal((T, k)) * np.linspace(10, 3, k)) @ rng.standard_normal((k, N)) + rng.standard_normal((T, N))
        elif kind == "graded":
            k = min(T, N)
            x = (rng.standard_normal((T, k)) * np.logspace(0, -3, k)) @ rng.standard_normal((k, N))
        elif kind == "lowrank":
            k = max(2, min(T, N) // 3)
            x = rng.standard_normal((T, k)) @ rng.standard_normal((k, N))
        else:
            x = rng.standard_normal((T, N))
            x[:, N // 2:] = x[:, :N - N // 2]
        return x.astype(np.float32 if f32 else np.float64)
    Ns = [int(rng.choice([3, T // 2, T - 1, T, T + 1, 2 * T, 5 * T])) for _ in range(2 if two else 1)]
    Ns = [max(2, n) for n in Ns]
    fields = [field(n) for n in Ns]
    tag = dict(case=case, T=T, Ns=Ns, cplx=cplx, f32=f32, kind=str(kind))
    try:
        m = MCA(*fields)
        m.solve(complexify=cplx)
        ref = O.OracleModel(*fields).solve(complexify=cplx)
    except Exception as e:                                    # noqa: BLE001
        fails += 1
        print("EXCEPTION", json.dumps(tag), repr(e)[:200])
        continue
    gs = ref["singular_values"]; s = m._singular_values.astype(np.float64)
    stol, vtol, floor = (3e-4, 5e-3, 1e-2) if f32 else (1e-5, 1e-5, 1e-7)
    if len(s) != len(gs):
        fails += 1; print("LEN", json.dumps(tag), len(s), len(gs)); continue
    keep = gs > floor * gs[0]
    serr = float(np.max(np.abs(s[keep] - gs[keep]) / gs[keep])) if keep.any() else 0.0
    # vectors of the well-separated leading modes
    nk = int(min(keep.sum(), 6))
    gaps_ok = nk > 0
    verr = orth = 0.0
    if nk > 0:
        for side, key in enumerate(m._keys):
            gv = ref["V"][side][:, :nk]
            rel_gap = np.abs(np.diff(gs[:nk + 1])) / gs[:nk] if len(gs) > nk else np.r_[np.abs(np.diff(gs[:nk])) / gs[:nk - 1], 1.0]
            sel = np.nonzero(rel_gap > 0.05)[0]
            sel = np.array([i for i in sel if i == 0 or rel_gap[i - 1] > 0.05], dtype=int)
            V = np.asarray(m._V[key][:, :nk])
            if len(sel):
                mine, _ = align_modes(V[:, sel], gv[:, sel])
                verr = max(verr, float(np.max(np.abs(mine - gv[:, sel])) / np.max(np.abs(gv[:, sel]))))
            nn = int(keep.sum())
            Vall = np.asarray(m._V[key][:, :nn])
            orth = max(orth, float(np.abs(Vall.conj().T @ Vall - np.eye(nn)).max()))
    bad = serr > stol or verr > vtol or orth > (2e-3 if f32 else 1e-5)
    if bad:
        fails += 1
        print("FAIL", json.dumps(tag), "sigma %.2e vec %.2e orth %.2e" % (serr, verr, orth), sorted(m._device().timings()))
print("cases %d, failures %d" % (n_cases, fails))
sys.exit(1 if fails else 0)
