"""C5 (EOF of a 0.25-degree global grid, T = 1200 x N = 1 036 800, float32) through the class with the constructor
preprocessing on the device."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scripts.run_config import gen_C
from xmca_amd import _hip
from xmca_amd.array import MCA
X = gen_C()
h = _hip.Handle(0)
out = {}
t0 = time.perf_counter(); m = MCA(X, handle=h, preprocess='device'); out["ctor_s"] = time.perf_counter() - t0
t0 = time.perf_counter(); m.solve(); out["first_solve_s"] = time.perf_counter() - t0      # (pays the hipMalloc of the 5 GB float32 result)
out["first_solve_stages_ms"] = h.timings()
h.reset_timings()
t0 = time.perf_counter(); m.solve(); out["second_solve_s"] = time.perf_counter() - t0     # (creates the eigensolver's second stream: ~10 ms once)
h.reset_timings()
t0 = time.perf_counter(); m.solve(); out["solve_s"] = time.perf_counter() - t0
out["stages_ms"] = h.timings()
out["vectors_are_f32"] = bool(h.vectors_are_f32(0))
h.reset_timings()
t0 = time.perf_counter(); m.rotate(10, 1); out["rotate_s"] = time.perf_counter() - t0
out["rotate_stages_ms"] = h.timings(); out["varimax_iterations"] = m._varimax_iterations
t0 = time.perf_counter(); e = m.eofs(10); out["eofs10_s"] = time.perf_counter() - t0
out["sigma_head"] = [float(x) for x in m._singular_values[:3]]
print(json.dumps(out))
