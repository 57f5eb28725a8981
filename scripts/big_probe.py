"""Phase timestamps of ONE big front (64-159 scalars) inside the table-driven factor launch.  usage: big_probe.py cfg4 K LEVEL [POS]"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from plan_exec import Plan
pkg = importlib.import_module("opendlv-logic-cfsd18-sensation-slam_amd")
name, K, lev = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]); posin = int(sys.argv[4]) if len(sys.argv) > 4 else 0
N, M = pkg.track.CONFIGS[name]
t = pkg.track.generate(N, M, K); fe = pkg.Graph(); g = pkg.track.bench_graph(t, fe)
H = pkg.Graph(device=-2); H.load_bench_graph(g); H.plan_build_host(); P = Plan(H.plan_export()); H.close()
pos = int(P.level_start[lev]) + posin
fr = int(P.level_fronts[pos])
os.environ["GS_DBG"] = str(16 | (pos << 8))
G = pkg.Graph(); G.load_bench_graph(g); G.initialize_optimization()
for _ in range(4):
    G.iterate()
G.synchronize()
ts = [int(v) for v in G.debug_timestamps()]
us = lambda a, b: (ts[b] - ts[a]) / 100.0
npan = (int(P.npiv[fr]) + 3) // 4
print("front %d at level %d: npiv %d nbnd %d children %d" % (fr, lev, P.npiv[fr], P.nbnd[fr], P.child_cnt[fr]))
print("   zero image + originals %.2f us | children %.2f | accumulators %.2f | %d panels %.2f (%s) | update matrix out + flag %.2f | total %.2f"
      % (us(0, 1), us(1, 2), us(2, 3), npan, us(3, 20), " ".join("%.2f" % ((ts[4 + b] - (ts[3 + b] if b else ts[3])) / 100.0) for b in range(min(npan, 16))), us(20, 21), us(0, 21)))
