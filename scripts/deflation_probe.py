"""GPU box: does a second solve on DEFLATED fields restore the weak modes of a two-field model?  (prototype through the
class: the strong singular subspaces of the first solve are projected out of the centered fields on the host)"""
import json, os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from oracle import ref_numpy as O
from xmca_amd.array import MCA
from conftest import align_modes

def fields(T=300, N=700):
    rng = np.random.default_rng(11)
    def field(shift, n):
        x = np.linspace(0, 1, n)
        modes = np.cos(np.pi * (np.arange(T)[:, None] + shift) * x[None, :])
        return (rng.standard_normal((T, T)) * np.logspace(0, -5, T)) @ modes
    return [field(0.0, N), field(0.3, N - 50)]

f = fields()
ref = O.OracleModel(*f).solve()
gs = ref["singular_values"]
m = MCA(*f, preprocess='host')
m.solve()
s1 = m._singular_values.copy()
Va, Vb = m._V['left'].copy(), m._V['right'].copy()
levels = []
sig, VA, VB = s1.copy(), Va.copy(), Vb.copy()
Xa, Xb = m._fields['left'].copy(), m._fields['right'].copy()
done = 0
for level in range(3):
    thr = 1e-3 * sig[done]
    ns = done + int(np.sum(sig[done:] >= thr))
    if ns >= len(sig) - 1: break
    Ua, Ub = VA[:, :ns], VB[:, :ns]
    Ua, _ = np.linalg.qr(Ua); Ub, _ = np.linalg.qr(Ub)          # (orthonormal bases of the strong subspaces)
    Xa2 = Xa - (Xa @ Ua) @ Ua.T
    Xb2 = Xb - (Xb @ Ub) @ Ub.T
    m2 = MCA(Xa2, Xb2, preprocess='host')
    m2.solve()
    k = len(sig) - ns
    sig[ns:] = m2._singular_values[:k]
    VA[:, ns:] = m2._V['left'][:, :k]
    VB[:, ns:] = m2._V['right'][:, :k]
    levels.append(ns)
    done = ns
out = {"levels_split_at": levels}
for name, s, V in (("first_solve", s1, Va), ("deflated", sig, VA)):
    for lo in (1e-3, 1e-4, 1e-5, 1e-6, 1e-7, 1e-8, 1e-9):
        keep = gs > lo * gs[0]
        nk = int(keep.sum())
        al, _ = align_modes(V[:, :nk], ref["V"][0][:, :nk])
        verr = np.max(np.abs(al - ref["V"][0][:, :nk]), axis=0) / np.max(np.abs(ref["V"][0][:, :nk]), axis=0)
        out["%s above_%g" % (name, lo)] = {"n": nk, "sigma_rel_err": float(np.max(np.abs(s[keep] - gs[keep]) / gs[keep])),
                                          "orth": float(np.max(np.abs(V[:, :nk].T @ V[:, :nk] - np.eye(nk)))), "vec_err_max": float(verr.max())}
print(json.dumps(out, indent=1))
