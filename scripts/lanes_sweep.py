#!/usr/bin/env python3
"""surrogates/s of xmca_rule_n by number of lanes (XMCA_RULE_N_LANES is read once per process: one process per setting).
    python scripts/lanes_sweep.py            # C4, C4 rotated, C2-shaped EOF, each with 1 / 2 / 3 / 4 lanes"""
import json, os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r"""
import sys, time, json, numpy as np
sys.path.insert(0, %r)
from xmca_amd import _hip
h = _hip.Handle(0)
args, n_out, runs = json.loads(sys.argv[1])
h.rule_n(*args, 0, 4, 7, np.float64, n_out)
t0 = time.perf_counter(); sp, kept = h.rule_n(*args, 0, runs, 1, np.float64, n_out); dt = time.perf_counter() - t0
print(json.dumps({"per_s": runs / dt, "kept": int(kept.sum())}))
""" % REPO
CASES = {"C4": ([5000, 20000, 15000, 2, True, False, 0, 1, 1e-8], 5000, 18),
         "C4 rotated": ([5000, 20000, 15000, 2, True, True, 20, 4, 1e-8], 20, 8),
         "C2-shaped EOF": ([2920, 10000, 0, 1, False, False, 0, 1, 1e-8], 2920, 18)}
for name, (args, n_out, runs) in CASES.items():
    row = []
    for lanes in (1, 2, 3, 4):
        r = subprocess.run([sys.executable, "-c", CODE, json.dumps([args, n_out, runs])], env=dict(os.environ, XMCA_RULE_N_LANES=str(lanes)),
                           capture_output=True, text=True, timeout=1200)
        row.append(json.loads(r.stdout.splitlines()[-1])["per_s"] if r.returncode == 0 else float("nan"))
    print("%-16s lanes 1/2/3/4: %s" % (name, "  ".join("%6.2f" % v for v in row)), flush=True)
