#!/usr/bin/env python3
"""Golden vectors at the BASELINE.json configurations, from the REAL reference (build container only).

Run:  PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python oracle/make_config_goldens.py

Cases (inputs are regenerated from seeds by tests/golden_inputs.py - SURVEY.md Appendix C generators):
  c1_standin  configs[0]: T = 2920 x (25 x 53, 25 x 27) float32 stand-in of the tutorial split, solve() + rotate(10, 1)
  c2_full     configs[1] at FULL size: EOF T = 2920 x N = 10 000 float64, solve() + rotate(10, 1)
  c3_reduced  configs[2] at T = 1000 x (4000, 3000): MCA, complexify=True, rotate(20, 4) (geometric amplitudes)
  c5_scaled   configs[4] at T = 1200 x 41 472 float32 (3-D input): EOF, solve() + rotate(10, 1)
For each case the real `xmca.array.MCA` is run, `oracle/ref_numpy.py` is pinned against it (sigma, R, Phi, variance,
Varimax iteration count) and the outputs are written to tests/golden/config_cases.npz: all singular values, the
leading unrotated vectors (float32 / complex64 storage: 6e-8 relative, far below the 1e-5 they are compared at), R,
Phi, norms, variance, mode order, explained variance, rotated PCs, iteration count.

  c4_run0     configs[3]: surrogate (seed 1, run 0) of rule_n at C4 size, T = 5000 x (20 000, 15 000), complexify, through the
              real reference on the normals of oracle/philox_numpy.py (= the device generator) -> rule_n_c4_run0.npz

Also: bootstrapping goldens (reference `MCA.bootstrapping(3, ...)` under `np.random.seed(5)`) for the five
parameterisations of tests/test_gpu_mca.py -> tests/golden/bootstrap_cases.npz.
"""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "oracle"))
sys.path.insert(0, os.path.join(REPO, "tests"))

from make_goldens import OUT, SvdCounter, import_reference, rel  # noqa: E402
from golden_inputs import make_input  # noqa: E402

CONFIGS = [  # name, complexify, n_rot, power, modes of V stored, row stride of stored vectors
    ("c1_standin", False, 10, 1, 10, 1),      # configs[0]: air_temperature-shaped stand-in, T = 2920 x (25 x 53, 25 x 27) float32
    ("c2_full", False, 10, 1, 10, 1),
    ("c3_reduced", True, 20, 4, 20, 1),
    ("c5_scaled", False, 10, 1, 10, 2),
    ("c3_full", True, 20, 4, 20, 8),          # ~6 minutes and ~15 GB on 8 cores; written to config_c3_full.npz
    ("c3_real_full", False, 20, 2, 20, 8),    # the same fields without complexify (general two-field route); config_c3_real_full.npz
]

BOOT = [  # tag, input, single field, complexify, rotation, kwargs  (tests/test_gpu_mca.py bootstrapping cases)
    ("small_std", "small_both", False, False, None, dict(on_left=True, on_right=True, block_size=2)),
    ("wide_rot", "wide_both", False, False, (5, 2), dict(on_left=True, on_right=False, block_size=1)),
    ("wide_single_cplx", "wide_both", True, True, None, dict(on_left=True, on_right=False, block_size=4, replace=False)),
    ("wide_cplx_rot", "wide_both", False, True, (4, 1), dict(on_left=False, on_right=True, block_size=1)),
    ("sst_iterative", "sst_prcp", False, False, None, dict(on_left=True, on_right=True, block_size=3, strategy='iterative')),
]


def config_case(MCA, name, cplx, n_rot, power, n_vec, stride, pin_oracle=True, pcs_stride=1):
    from oracle import ref_numpy as O
    fields = make_input(name)
    t0 = time.perf_counter()
    m = MCA(*fields)
    t1 = time.perf_counter()
    m.solve(complexify=cplx)
    t2 = time.perf_counter()
    keys = list(m._V.keys())
    out = {
        "singular_values": np.asarray(m._singular_values, dtype=np.float64),
        "total_covariance": np.asarray(m._analysis["total_covariance"], dtype=np.float64),
        "rank": np.asarray(m._analysis["rank"]),
        "stride": np.asarray(stride),
    }
    store = np.complex64 if cplx else np.float32
    for k in keys:
        out["V_" + k] = np.ascontiguousarray(m._V[k][::stride, :n_vec]).astype(store)
    with SvdCounter() as cnt:
        m.rotate(n_rot, power)
    t3 = time.perf_counter()
    out.update({"n_iter": np.asarray(cnt.n), "R": m._rotation_matrix, "Phi": m._correlation_matrix,
                "variance": np.asarray(m._variance, dtype=np.float64), "var_idx": m._var_idx,
                "explained_variance": np.asarray(m.explained_variance(), dtype=np.float64)})
    pcs = m.pcs(n_rot)
    for k in keys:
        out["norm_" + k] = np.asarray(m._norm[k], dtype=np.float64)
        out["pcs_" + k] = pcs[k][::pcs_stride].astype(store)
    out["pcs_stride"] = np.asarray(pcs_stride)
    out["cpu_seconds"] = np.asarray([t1 - t0, t2 - t1, t3 - t2])
    print("%-11s reference: ctor %.1f s, solve %.1f s, rotate %.2f s, %d Varimax iterations" %
          (name, t1 - t0, t2 - t1, t3 - t2, cnt.n), flush=True)

    if not pin_oracle:       # (the full-size C3 case: the restatement is the same numpy calls once more - 5 more minutes)
        return out
    # pin the oracle at this size
    om = O.OracleModel(*fields)
    om.solve(complexify=cplx)
    oo = om.rotate(n_rot, power)
    f32 = np.asarray(fields[0]).dtype == np.float32
    tol = 1e-5 if f32 else 1e-10
    assert rel(om.singular_values, m._singular_values) < tol, name
    assert oo["n_iter"] == cnt.n, (name, oo["n_iter"], cnt.n)
    assert rel(oo["R"], m._rotation_matrix) < (1e-4 if f32 else 1e-9), name
    assert rel(oo["variance"], m._variance) < (1e-4 if f32 else 1e-9), name
    return out


def bootstrap_case(MCA, inp, single, cplx, rot, kw):
    fields = make_input(inp)
    if single:
        fields = fields[:1]
    m = MCA(*fields)
    m.solve(complexify=cplx)
    if rot:
        m.rotate(*rot)
    np.random.seed(5)
    return m.bootstrapping(3, n_modes=4, **kw)


def main():
    MCA, _, _ = import_reference()
    only = sys.argv[1:]
    if not only or "configs" in only:
        out = {}
        for name, cplx, n_rot, power, n_vec, stride in CONFIGS:
            if name in ("c3_full", "c3_real_full"):
                continue
            for k, v in config_case(MCA, name, cplx, n_rot, power, n_vec, stride).items():
                out[name + "__" + k] = v
        dst = os.path.join(OUT, "config_cases.npz")
        np.savez_compressed(dst, **out)
        print("wrote %s (%.2f MB)" % (dst, os.path.getsize(dst) / 1e6))
    if "c1_standin" in only:          # add / refresh this one case inside the existing file (the others take minutes)
        dst = os.path.join(OUT, "config_cases.npz")
        out = dict(np.load(dst))
        name, cplx, n_rot, power, n_vec, stride = [c for c in CONFIGS if c[0] == "c1_standin"][0]
        out = {k: v for k, v in out.items() if not k.startswith(name + "__")}
        for k, v in config_case(MCA, name, cplx, n_rot, power, n_vec, stride).items():
            out[name + "__" + k] = v
        np.savez_compressed(dst, **out)
        print("wrote %s (%.2f MB)" % (dst, os.path.getsize(dst) / 1e6))
    for big in ("c3_full", "c3_real_full"):
        if big not in only:
            continue
        name, cplx, n_rot, power, n_vec, stride = [c for c in CONFIGS if c[0] == big][0]
        out = {name + "__" + k: v for k, v in config_case(MCA, name, cplx, n_rot, power, n_vec, stride, pin_oracle=False, pcs_stride=5).items()}
        dst = os.path.join(OUT, "config_%s.npz" % name)
        np.savez_compressed(dst, **out)
        print("wrote %s (%.2f MB)" % (dst, os.path.getsize(dst) / 1e6))
    if "c4_run0" in only:
        # BASELINE configs[3]: ONE surrogate of rule_n on the C4 configuration, exactly the normals the device generates for
        # (seed 1, run 0) - oracle/philox_numpy.py restates the generator - through the REAL reference (array.py:1753-1765).
        # ~5 minutes and ~15 GB on 8 cores.  bench.py and tests/test_gpu_configs.py compare row 0 of xmca_rule_n with it.
        from oracle.philox_numpy import surrogate_fields
        Tn, widths, seed = 5000, (20000, 15000), 1
        t0 = time.perf_counter()
        fields = surrogate_fields(Tn, widths, seed, 0)
        t1 = time.perf_counter()
        m = MCA(*fields)
        m.solve(complexify=True)
        t2 = time.perf_counter()
        out = {"T": np.asarray(Tn), "widths": np.asarray(widths), "seed": np.asarray(seed), "run": np.asarray(0),
               "variance": np.asarray(m._get_variance(), dtype=np.float64),
               "singular_values": np.asarray(m._singular_values, dtype=np.float64),
               "first_normals_left": fields[0].ravel()[:64].copy(), "first_normals_right": fields[1].ravel()[:64].copy(),
               "checksum_left": np.asarray(fields[0].sum()), "checksum_right": np.asarray(fields[1].sum())}
        try:
            m.rotate(20, 4)
            out["rotated_dropped"] = np.asarray(0)
            out["rotated_variance"] = np.asarray(m._get_variance(), dtype=np.float64)
        except RuntimeError:
            out["rotated_dropped"] = np.asarray(1)      # Varimax does not converge on complex white noise (array.py:1762-1763)
        t3 = time.perf_counter()
        out["cpu_seconds"] = np.asarray([t1 - t0, t2 - t1, t3 - t2])
        print("c4_run0 reference: normals %.1f s, ctor + solve %.1f s, rotate %.1f s (dropped: %d)" %
              (t1 - t0, t2 - t1, t3 - t2, int(out["rotated_dropped"])), flush=True)
        dst = os.path.join(OUT, "rule_n_c4_run0.npz")
        np.savez_compressed(dst, **out)
        print("wrote %s (%.3f MB)" % (dst, os.path.getsize(dst) / 1e6))
    if not only or "bootstrap" in only:
        out = {}
        for tag, inp, single, cplx, rot, kw in BOOT:
            out[tag] = bootstrap_case(MCA, inp, single, cplx, rot, kw)
            print("bootstrap", tag, out[tag].shape)
        dst = os.path.join(OUT, "bootstrap_cases.npz")
        np.savez_compressed(dst, **out)
        print("wrote %s (%.3f MB)" % (dst, os.path.getsize(dst) / 1e6))


if __name__ == "__main__":
    main()
