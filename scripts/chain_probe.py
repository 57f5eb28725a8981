"""One front's phase timestamps against its children's completion times (library built with -DF3_DONE_TS=1: GS_LIB=...).
usage: python scripts/chain_probe.py cfg4 LEVEL [POS]"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from plan_exec import Plan
pkg = importlib.import_module("opendlv-logic-cfsd18-sensation-slam_amd")
name, lev = sys.argv[1], int(sys.argv[2]); posin = int(sys.argv[3]) if len(sys.argv) > 3 else 0
N, M = pkg.track.CONFIGS[name]
t = pkg.track.generate(N, M); fe = pkg.Graph(); g = pkg.track.bench_graph(t, fe)
H = pkg.Graph(device=-2); H.load_bench_graph(g); H.plan_build_host(); P = Plan(H.plan_export()); H.close()
pos = int(P.level_start[lev]) + posin
os.environ["GS_DBG"] = str(16 | (pos << 8))
G = pkg.Graph(); G.load_bench_graph(g); G.initialize_optimization()
for _ in range(5):
    G.iterate()
G.synchronize()
ts = G.debug_timestamps(); done = G.debug_front_times()[0]
fr = int(P.level_fronts[pos])
kids = [int(c) for c in P.children[P.child_off[fr]:P.child_off[fr] + P.child_cnt[fr]]]
t_k = [int(done[c]) for c in kids]; last = max(t_k)
us = lambda v: (int(v) - last) / 100.0
print("front %d (level %d, npiv %d, nbnd %d), children done at %s us (last = 0)" % (fr, lev, P.npiv[fr], P.nbnd[fr], [round(us(v), 2) for v in t_k]))
print("   start %.2f | originals assembled %.2f | children gathered %.2f | accumulators %.2f | panels start %.2f | panels done %.2f | done stamp %.2f | stores drained (probe only) %.2f"
      % (us(ts[0]), us(ts[5]), us(ts[9]), us(ts[10]), us(ts[6]), us(ts[7]), us(done[fr]), us(ts[8])))
b = G.debug_front_times()[1]; par = int(P.parent[fr])
if par >= 0 and ts[32]:
    ub = lambda v: (int(v) - int(b[par])) / 100.0
    print("backward solve: parent's done stamp = 0 | start %.2f | L in LDS %.2f | columns in registers, boundary values seen %.2f | mat-vec %.2f | substitution %.2f | done stamp %.2f | store drained (probe only) %.2f"
          % (ub(ts[32]), ub(ts[34]), ub(ts[35]), ub(ts[36]), ub(ts[37]), ub(b[fr]), ub(ts[38])))
