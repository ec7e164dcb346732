"""Where the wall-clock of MCA(...).solve() goes at C2 (host side included)."""
import cProfile, pstats, sys, time
import numpy as np
sys.path.insert(0, ".")
from bench import gen_A
from xmca_amd.array import MCA

X = gen_A()
m = MCA(X); m.solve()          # warm-up
t0 = time.perf_counter(); m = MCA(X); t1 = time.perf_counter()
pr = cProfile.Profile(); pr.enable(); m.solve(); pr.disable()
t2 = time.perf_counter()
print("ctor %.1f ms, solve %.1f ms" % (1e3 * (t1 - t0), 1e3 * (t2 - t1)))
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
pr = cProfile.Profile(); pr.enable(); m = MCA(X); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(10)
