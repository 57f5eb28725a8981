"""Random-graph stress of the HIP path against the oracle (not part of the suite; tests/ holds fixed seeds of the same shapes): one Gauss-Newton step
of a single handle, then the same graph as 2 - 5 rank handles sharing this GPU (exchange summed in-process), irregular graphs of random shape —
loop-closure odometry edges, parallel edges, fixed poses in the middle of the chain, few cones seen from everywhere (fat separators: workgroup
fronts, the block VALU fallback beyond 159 scalars).  usage: python scripts/stress_gpu.py [first_seed] [count]"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import make_oracle_graph, random_graph
pkg = importlib.import_module("opendlv-logic-cfsd18-sensation-slam_amd")
pkg.binding.DEFAULT_DEBUG["grow_min_poses"] = 0
from oracle import pyoracle as po
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0; count = int(sys.argv[2]) if len(sys.argv) > 2 else 60
bad = skipped = single = sharded = local_steps = not_by_window = 0
variants = {}
for seed in range(first, first + count):
    rng = np.random.default_rng(7000 + seed)
    kw = dict(n_poses=int(rng.integers(8, 600)), n_lms=int(rng.integers(4, 70)), extra_pp=int(rng.integers(0, 10)) if rng.random() < 0.6 else 0,
              obs_per_pose=int(rng.integers(1, 6)), dup_edges=int(rng.integers(0, 4)) if rng.random() < 0.5 else 0)
    if len(sys.argv) > 3 and sys.argv[3] == "fat":               # few cones, each seen from everywhere: separators of 60 - 200 scalars
        kw.update(n_lms=int(rng.integers(20, 95)), obs_per_pose=int(rng.integers(4, 12)), n_poses=int(rng.integers(100, 1500)))
    kw["obs_per_pose"] = min(kw["obs_per_pose"], kw["n_lms"])
    g = random_graph(seed, **kw)
    g["fixed_poses"] = np.array(sorted(set([0] + list(rng.choice(kw["n_poses"], int(rng.integers(0, 3)), replace=False)))), dtype=np.int32)
    og = make_oracle_graph(po, g); og.build_system()
    try: x = og.solve_ldlt(0)
    except Exception: skipped += 1; continue                     # a cone nobody sees: singular, the oracle's LDL^T stops (so does the HIP path: tests/)
    og.apply_update(x); dp_o, dl_o = og.delta(); scale = max(np.abs(dp_o).max(), np.abs(dl_o).max())
    G = pkg.Graph(); G.load_bench_graph(g); done, st = G.optimize(1); dp, dl = G.export_delta(); G.close()
    variants[st.factor_variant] = variants.get(st.factor_variant, 0) + 1; single += 1
    err = max(np.abs(dp - dp_o).max(), np.abs(dl - dl_o).max()) / scale
    if done != 1 or err > 1e-8: bad += 1; print("BAD single seed", seed, kw, "done", done, "err %.2e" % err, "max front", st.max_front, "variant", st.factor_variant, flush=True)
    for world, local in ((2, 0), (3, 0), (5, 0), (2, 1), (4, 1)):
        if kw["n_poses"] < 8 * world: continue
        ranks = []
        if local:                                                    # rank-local ingestion: only graphs that can be planned by windows (edges grouped by pose, odometry along the chain)
            try:
                masks = pkg.binding.landmark_windows(g, world)
                for r in range(world):
                    H = pkg.Graph(); ranks.append(H); H.load_bench_graph_shard(g, r, world, masks); H.initialize_optimization()
            except pkg.GsError as ex:
                [H.close() for H in ranks]
                if "landmark windows" in str(ex): not_by_window += 1; continue
                raise
            local_steps += 1
        else:
          for r in range(world):
            H = pkg.Graph(); H.load_bench_graph(g); H.dist_configure(r, world); H.initialize_optimization(); ranks.append(H)
        lens = {H.dist_exchange_doubles() for H in ranks}
        if len(lens) != 1: bad += 1; print("BAD exchange sizes seed", seed, kw, world, sorted(lens), flush=True); [H.close() for H in ranks]; continue
        for H in ranks: H.dist_iterate_local()
        total = sum(H.dist_read_exchange() for H in ranks)
        for H in ranks: H.dist_write_exchange(total); H.dist_iterate_finish()
        P = np.zeros_like(g["pose_est"]); L = np.zeros_like(g["lm_est"]); cp = np.zeros(len(P)); cl = np.zeros(len(L))
        for H in ranks:
            H.sync_estimates(); pk, lk, pprim, lprim = H.dist_known()
            P += H.poses() * pprim[:, None]; L += H.landmarks() * lprim[:, None]; cp += pprim; cl += lprim; H.close()
        e2 = max(np.abs(P - og.poses()).max(), np.abs(L - og.landmarks()).max()) / max(scale, 1e-300)
        sharded += 1
        if not (np.all(cp == 1) and np.all(cl == 1)) or e2 > 1e-7: bad += 1; print("BAD sharded seed", seed, kw, "world", world, "err %.2e" % e2, flush=True)
print("seeds %d..%d: %d single-handle steps (factor variants %s), %d sharded steps (%d of them with rank-local ingestion; %d graphs refused it: not plannable by windows), %d singular graphs skipped, %d BAD" % (first, first + count - 1, single, variants, sharded, local_steps, not_by_window, skipped, bad))
sys.exit(1 if bad else 0)
