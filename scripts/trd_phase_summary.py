import sys, numpy as np
for f in sys.argv[1:]:
    a = np.loadtxt(f)
    n = a.shape[0]
    # columns: 0 = gap from previous column's stamp0, 1..6 = stamps relative to stamp 0 (100 MHz ticks -> 10 ns)
    d = np.diff(np.concatenate([np.zeros((n,1)), a[:,1:]], axis=1), axis=1)
    print(f, "n", n, "per-column total us: %.2f" % (a[1:,0].mean()/100))
    names = ["gather+w", "x,norm", "v", "pass", "tail5", "tail6"]
    for lo, hi in ((0, n//4), (n//4, n//2), (n//2, 3*n//4), (3*n//4, n-1)):
        print("  cols %4d-%4d:" % (lo, hi), "  ".join("%s %.2f" % (nm, max(0, d[lo:hi, q].mean())/100) for q, nm in enumerate(names)), " total %.2f" % (a[lo+1:hi+1,0].mean()/100))
