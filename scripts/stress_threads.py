"""Four host threads, a handle each (externally synchronised per handle, as the boundary says; ctypes releases the GIL inside the calls), planning, optimising,
growing and re-planning at the same time on ONE GPU — the results must be bit for bit what each handle gives alone.  Looks for shared state inside the
library (launch-attribute caches, the last-error slot, the plan workspace, host thread pools).  usage: python scripts/stress_threads.py [rounds]"""
import importlib, os, sys, threading
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
pkg = importlib.import_module("opendlv-logic-cfsd18-sensation-slam_amd")
pkg.binding.DEFAULT_DEBUG["grow_min_poses"] = 0
from conftest import append_tail, split_for_growth
from oracle import pyoracle as po
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
fe = po.OracleFrontend()
sizes = [(240, 200), (1000, 200), (3000, 600), (10000, 2000)]
graphs = []
for N, M in sizes:
    t = pkg.track.generate(N, M); g = pkg.track.bench_graph(t, fe); graphs.append((g, split_for_growth(g, 4)))
def work(i, out):
    g, (base, tail, full) = graphs[i]
    res = []
    for r in range(rounds):
        G = pkg.Graph(); G.load_bench_graph(base); G.optimize(3)
        append_tail(G, tail, (0, 2)); G.optimize(2); append_tail(G, tail, (2, 4)); G.optimize(2)
        G.set_fixed_pose(5, True); G.optimize(2)                      # a change that needs a new plan on the same handle (the workspace is reused)
        res.append((G.poses().copy(), G.landmarks().copy(), G.chi2())); G.close()
    out[i] = res
alone = {}
for i in range(len(sizes)): work(i, alone)
together = {}
th = [threading.Thread(target=work, args=(i, together)) for i in range(len(sizes))]
for x in th: x.start()
for x in th: x.join()
bad = 0
for i in range(len(sizes)):
    for r in range(rounds):
        a, b = alone[i][r], together[i][r]
        if not (np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2]): bad += 1; print("BAD handle", i, "round", r, float(np.abs(a[0] - b[0]).max()))
        if r and not np.array_equal(alone[i][0][0], a[0]): bad += 1; print("BAD not reproducible alone", i, r)
print("%d handles x %d rounds on 4 threads: %d BAD" % (len(sizes), rounds, bad))
sys.exit(1 if bad else 0)
