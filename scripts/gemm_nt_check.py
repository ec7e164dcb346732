"""GPU box: correctness + speed of the NT stream-K GEMM (csrc/gemm_nt.h) against numpy and the general kernel."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xmca_amd import _hip
h = _hip.Handle(0)
rng = np.random.default_rng(0)
ok = True
for (M, N, K, dt, upper) in [(128, 128, 32, np.float64, False), (300, 200, 1000, np.float64, False), (257, 257, 5000, np.float64, True),
                             (1200, 1200, 20000, np.float32, True), (64, 640, 77, np.float32, False), (2920, 2920, 1000, np.float64, True),
                             (130, 130, 200001, np.float32, True)]:
    A = rng.standard_normal((M, K)).astype(dt)
    B = A if upper else rng.standard_normal((N, K)).astype(dt)
    C = h.gemm(A, B, a_kfast=True, b_nfast=False, alpha=0.5, upper_only=upper, mirror=1 if upper else 0)
    ref = 0.5 * (A.astype(np.float64) @ B.astype(np.float64).T)
    err = np.max(np.abs(C - ref)) / np.max(np.abs(ref))
    tol = 1e-13 if dt == np.float64 else 2e-6
    print("NT %5d x %5d x %7d %s upper=%d  rel err %.2e %s" % (M, N, K, np.dtype(dt).name, upper, err, "ok" if err < tol else "FAIL"))
    ok &= err < tol
print("all ok" if ok else "FAILED")
for (M, N, K, dt, name) in [(2920, 2920, 10000, np.float64, "C2 Gram f64"), (1200, 1200, 1036800, np.float32, "C5 Gram f32"),
                            (5000, 5000, 20000, np.float32, "C3 Gram f32"), (5000, 5000, 20000, np.float64, "C3-like f64"),
                            (4096, 4096, 4096, np.float64, "dense 4096 f64"), (4096, 4096, 4096, np.float32, "dense 4096 f32")]:
    for up in (True,):
        ms = h.bench_gemm(M, N, K, dt, a_kfast=True, b_nfast=False, upper_only=up, reps=5)
        fl = (M * (M + 1.0) if up else 2.0 * M * N) * K
        peak = 78.6 if dt == np.float64 else 157.3
        print("%-16s NT=%s  %.3f ms  %.1f TF  %.1f %% of peak" % (name, os.environ.get("XMCA_GEMM_NT", "1"), ms, fl / ms / 1e9, 100 * fl / ms / 1e9 / peak))
