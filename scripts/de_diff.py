"""Where two (d, e) outputs of the tridiagonal reduction for the SAME input differ (files written by XMCA_TRACE=trdsum with
XMCA_TRD_DUMP_DIR=dir): de_diff.py [dir]"""
import sys
import glob, numpy as np, collections
groups = collections.defaultdict(list)
for f in glob.glob((sys.argv[1] if len(sys.argv) > 1 else "gpurun_out") + "/de_*.bin"):
    ci = f.split("_")[-2]
    groups[ci].append(f)
for ci, fs in groups.items():
    if len(fs) < 2: continue
    a = [np.fromfile(f) for f in fs]
    n = a[0].size // 2
    for b in a[1:]:
        dd = np.where(a[0][:n] != b[:n])[0]; de = np.where(a[0][n:] != b[n:])[0]
        print(ci, "n", n, "first d diff", dd[:3], "first e diff", de[:3], "count", dd.size, de.size, "max abs d", np.abs(a[0][:n]-b[:n]).max())
        for i in list(de[:4]):
            print("   e[%d] = %.17g vs %.17g  rel %.3g   d[%d] %.17g vs %.17g" % (i, a[0][n+i], b[n+i], abs(a[0][n+i]-b[n+i])/abs(a[0][n+i]), i+1, a[0][i+1], b[i+1]))
