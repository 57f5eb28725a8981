"""Random fault injection (gs_debug_fail_at_iteration: a zero pivot or a front-flag timeout reported by the k-th iteration) on laps of random size and
random launch modes, against the oracle: a zero pivot stops the call with the last good iterate (g2o: optimize() returns 0), a timeout is repaired by
the per-level launches within the call, and the handle works on afterwards (and returns to the whole-tree launches at its 4th later call).
usage: python scripts/stress_failures.py [first_seed] [count]"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import make_oracle_graph
pkg = importlib.import_module("opendlv-logic-cfsd18-sensation-slam_amd")
from oracle import pyoracle as po
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0; count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
fe = po.OracleFrontend(); bad = 0
def rel(a, b): return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))
for seed in range(first, first + count):
    rng = np.random.default_rng(4000 + seed)
    N = int(rng.choice([50, 240, 1000, 3000])); M = max(30, N // 5)
    t = pkg.track.generate(N, M); g = pkg.track.bench_graph(t, fe)
    dbg = dict(tickets=int(rng.integers(0, 2)), leaf_kernel=int(rng.choice([-1, 0, 2])), block_fronts=int(rng.choice([0, 16, 512])))
    n = int(rng.integers(2, 9)); k = int(rng.integers(1, n + 1)); code = int(rng.integers(1, 3))
    og = make_oracle_graph(po, g)
    G = pkg.Graph(debug=dbg); G.load_bench_graph(g); G.initialize_optimization(); G.debug_fail_at_iteration(k, code)
    done, st = G.optimize(n)
    if code == 1:
        og.optimize(k - 1, ordering=1)
        ok = done == 0 and st.iterations == k - 1 and st.numeric_failure == 1 and rel(G.poses(), og.poses()) < 1e-8 and rel(G.landmarks(), og.landmarks()) < 1e-8
        applied = k - 1
    else:
        og.optimize(n, ordering=1)
        ok = done == n and st.numeric_failure == 0 and st.first_failure == 2 and st.fell_back == 1 and rel(G.poses(), og.poses()) < 1e-8
        applied = n
    why = ("first call", done, st.iterations, st.numeric_failure, st.first_failure, st.fell_back)
    if ok:
        for c in range(5):                                            # the handle works on; a fallen-back one retries the whole-tree launches at its 4th call
            m = int(rng.integers(1, 4)); d2, s2 = G.optimize(m); og.optimize(m, ordering=1)
            if d2 != m or s2.numeric_failure != 0 or rel(G.poses(), og.poses()) > 1e-8: ok = False; why = ("later call", c, d2, s2.numeric_failure); break
        if ok and code == 2 and G.stats().fell_back != 0: ok = False; why = ("did not return to the whole-tree launches",)
    if not ok: bad += 1; print("BAD seed", seed, dict(N=N, n=n, k=k, code=code, **dbg), why, flush=True)
    G.close()
print("seeds %d..%d: %d BAD" % (first, first + count - 1, bad))
sys.exit(1 if bad else 0)
