import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("opendlv-logic-cfsd18-sensation-slam_amd")
world = int(sys.argv[1]) if len(sys.argv) > 1 else 4
Nw, Mw = pkg.track.CONFIGS[sys.argv[2] if len(sys.argv) > 2 else "cfg3"]
t = pkg.track.generate(Nw * world, Mw * world); fe = pkg.Graph(); g = pkg.track.bench_graph(t, fe); fe.close()
print("fixed poses", g["fixed_poses"], "fixed lms", g["fixed_landmarks"])
for local in (1, 0):
    ranks = []
    masks = pkg.binding.landmark_windows(g, world) if local else None
    for r in range(world):
        G = pkg.Graph()
        if local: keep = G.load_bench_graph_shard(g, r, world, masks)
        else: G.load_bench_graph(g); G.dist_configure(r, world)
        G.initialize_optimization(); ranks.append(G)
    ok = True
    for it in range(3):
        for G in ranks: G.dist_iterate_local()
        total = sum(G.dist_read_exchange() for G in ranks)
        for G in ranks:
            G.dist_write_exchange(total)
            try: G.dist_iterate_finish()
            except Exception as e: print("local" if local else "full", "iteration", it, "finish:", e); ok = False
        if not ok: break
    for G in ranks:
        try: G.synchronize()
        except Exception as e: print("local" if local else "full", "sync:", e); ok = False; break
    st = ranks[0].stats()
    print("local" if local else "full", "world", world, "ok" if ok else "FAILED", "fronts", st.n_fronts, "shared", st.n_shared_fronts, "exchange", ranks[0].dist_exchange_doubles(), [G.dist_exchange_doubles() for G in ranks])
    for G in ranks: G.close()
