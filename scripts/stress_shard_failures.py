"""Random fault injection into SHARDED handles (not part of the suite; tests/ holds one zero-pivot and one timeout case): laps of 200-3 000 poses as 2-4 rank handles on
this GPU (rank-local ingestion or the whole graph on every rank, by the coin), a zero pivot or a flag timeout injected on a random rank at a random iteration of six,
the exchange summed in-process.  Checks: every rank reports at the next wait — the origin its own code, the others "another rank's" —, only the origin of a timeout falls
back, nobody applied the failed iteration or a later one of that batch, and after three more iterations the merged estimates are the oracle's after (k - 1) + 3.
usage: python scripts/stress_shard_failures.py [first_seed] [count]"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import make_oracle_graph
pkg = importlib.import_module("opendlv-logic-cfsd18-sensation-slam_amd")
from oracle import pyoracle as po
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0; count = int(sys.argv[2]) if len(sys.argv) > 2 else 30
fe = pkg.Graph(); bad = 0
rel = lambda a, b: np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)
for seed in range(first, first + count):
    rng = np.random.default_rng(9100 + seed)
    N = int(rng.integers(200, 3000)); M = max(40, N // int(rng.integers(4, 8))); world = int(rng.integers(2, 5)); local = bool(rng.integers(0, 2))
    try: t = pkg.track.generate(N, M)
    except ValueError: continue
    g = pkg.track.bench_graph(t, fe)
    code = int(rng.integers(1, 3)); k = int(rng.integers(1, 7)); who = int(rng.integers(0, world))
    ranks = []
    masks = pkg.binding.landmark_windows(g, world) if local else None
    for r in range(world):
        G = pkg.Graph()
        if local: G.load_bench_graph_shard(g, r, world, masks)
        else: G.load_bench_graph(g); G.dist_configure(r, world)
        G.initialize_optimization(); ranks.append(G)
    def iterate(n):
        for _ in range(n):
            for G in ranks: G.dist_iterate_local()
            total = sum(G.dist_read_exchange() for G in ranks)
            for G in ranks: G.dist_write_exchange(total); G.dist_iterate_finish()
    ranks[who].debug_fail_at_iteration(k, code)
    iterate(6)
    try:
        codes = []
        for G in ranks:
            try: G.synchronize(); codes.append(0)
            except pkg.GsError as e: codes.append(e.code)
        want = [(-8 if code == 1 else -10)] * world
        assert codes == want, ("codes", codes, want)
        fb = [G.stats().fell_back for G in ranks]
        assert fb == [1 if (code == 2 and r == who) else 0 for r in range(world)], ("fell back", fb)
        iterate(3)
        P = np.zeros((N, 3)); L = np.zeros((len(g["lm_est"]), 2))
        for G in ranks:
            G.sync_estimates(); pk, lk, pp, lp = G.dist_known(); P += G.poses() * pp[:, None]; L += G.landmarks() * lp[:, None]
        og = make_oracle_graph(po, g); og.optimize(k - 1 + 3, ordering=1)
        assert rel(P, og.poses()) < 1e-7 and rel(L, og.landmarks()) < 1e-7, ("estimates", rel(P, og.poses()), rel(L, og.landmarks()))
    except AssertionError as ex:
        bad += 1; print("BAD seed", seed, "N", N, "world", world, "local", local, "code", code, "iteration", k, "rank", who, ex, flush=True)
    for G in ranks: G.close()
print("seeds %d..%d: %d BAD" % (first, first + count - 1, bad))
sys.exit(1 if bad else 0)
