"""More two-/four-lane rule_n shapes: rotated (both persistent kernels), one field, float32 surrogates - give-ups and bit-equality."""
import sys, numpy as np
sys.path.insert(0, ".")
from xmca_amd import _hip
h = _hip.Handle(0); lib = _hip.load_library()
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 4
cases = [  # T, Nx, Ny, n_fields, cplx, rotated, p, power, dtype
    (800, 2000, 0, 1, False, True, 6, 1, np.float64),
    (1000, 2400, 1800, 2, True, True, 8, 2, np.float64),
    (1000, 2400, 1800, 2, False, True, 8, 1, np.float64),
    (1200, 3000, 0, 1, True, False, 0, 0, np.float64),
    (900, 2200, 1700, 2, True, False, 0, 0, np.float32),
    (2920, 10000, 0, 1, False, True, 10, 1, np.float64),
]
for T, Nx, Ny, nf, cplx, rot, p, power, dt in cases:
    args = (T, Nx, Ny, nf, cplx, rot, p, power, 1e-8)
    n_out = p if rot else T
    ref, k0 = h.rule_n(*args, 0, 8, 3, dt, n_out)
    g0 = lib.xmca_persistent_giveups(); same = True
    for r in range(calls):
        sp, k = h.rule_n(*args, 0, 8, 3, dt, n_out)
        same = same and np.array_equal(sp, ref) and np.array_equal(k, k0)
    print(T, Nx, Ny, nf, cplx, rot, dt.__name__, "giveups:", lib.xmca_persistent_giveups() - g0, "same bits:", same, "kept", int(k0.sum()), flush=True)
