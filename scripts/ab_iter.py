"""A/B of full Gauss-Newton iterations inside ONE session (boxes differ by ~15 % in latency-bound kernels: compare only
within a call).  usage: python scripts/ab_iter.py cfg4 "GS_CLUSTER_WAYS=2" "GS_CLUSTER_WAYS=8" "GS_LEAF_POSES=5" ...
Each argument after the workload is a space-separated list of KEY=VALUE environment settings ("" = defaults)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("opendlv-logic-cfsd18-sensation-slam_amd")
name = sys.argv[1]
variants = sys.argv[2:] or [""]
N, M = pkg.track.CONFIGS[name]
t = pkg.track.generate(N, M); fe = pkg.Graph(); g = pkg.track.bench_graph(t, fe); fe.close()
for rep in range(2):
    for v in variants:
        kv = dict(x.split("=", 1) for x in v.split()) if v else {}
        old = {k: os.environ.get(k) for k in kv}
        os.environ.update(kv)
        G = pkg.Graph(); G.load_bench_graph(g); G.initialize_optimization(); st = G.stats()
        s = G.time_iterations(30)
        print("%-40s fronts %6d levels %2d maxf %2d | lin %.4f factor %.4f back %.4f upd %.4f total %.4f ms -> %.0f it/s  (fail %d)"
              % (v or "(defaults)", st.n_fronts, st.n_levels, st.max_front, s.ms_linearize, s.ms_factor, s.ms_backsolve, s.ms_update, s.ms_total, 1e3 / s.ms_total, s.numeric_failure), flush=True)
        G.close()
        for k, o in old.items():
            if o is None: os.environ.pop(k, None)
            else: os.environ[k] = o
