"""Summarises gpurun_out/jac_prof.txt (phase stamps written by a -DXMCA_JAC_PROF build of the fused Jacobi round)."""
import sys
import numpy as np

a = np.loadtxt(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/jac_prof.txt", dtype=np.int64).reshape(512, 12, 12)
S = int(sys.argv[2]) if len(sys.argv) > 2 else 46
st = a[:, :, 2:]
asm = [st[w, 0, 0] - st[w, 0, 8] for w in range(S)]
evd = [st[w, 0, 1] - st[w, 0, 0] for w in range(S)]
print("solver WGs: assemble median %d  sweep median %d  total median %d max %d cycles" % (
    np.median(asm), np.median(evd), np.median(np.add(asm, evd)), np.max(np.add(asm, evd))))
durs = {"G": [], "Z": [], "D": []}
spans, nit, first = [], [], []
for w in range(S, 512):
    n, last = 0, None
    for it in range(12):
        s = st[w, it]
        if s[0] < st[w, 0, 8] or s[7] < s[0]:
            break
        n += 1
        last = s[7]
        if s[9] == 1:
            durs["G"].append([s[1] - s[0], s[2] - s[1], s[3] - s[2], s[4] - s[3], s[5] - s[4], s[6] - s[5], s[7] - s[6], s[7] - s[0]])
        elif s[9] == 2:
            durs["Z"].append([s[1] - s[0], s[2] - s[1], s[3] - s[2], s[6] - s[3], s[7] - s[6], s[7] - s[0]])
        else:
            durs["D"].append([s[7] - s[0]])
    nit.append(n)
    if n:
        spans.append(last - st[w, 0, 8])
        first.append(st[w, 0, 0] - st[w, 0, 8])
print("update WGs: items/WG hist", np.bincount(nit), " span median %d max %d  first-item delay median %d" % (
    np.median(spans), np.max(spans), np.median(first)))
for k, v in durs.items():
    if v:
        v = np.array(v)
        print(k, len(v), "median phase cycles", np.median(v, axis=0).astype(int))
