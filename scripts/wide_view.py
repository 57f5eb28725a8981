"""Phase times of full Gauss-Newton iterations on tracks with K = 8 / 16 / 24 cones in view (100k poses / 10k cones by default):
K = 8 is SURVEY 8d's track (every front fits a wave); 16 / 24 are frames of the reference's coneMappingThreshold of 50 m
(src/slam.cpp:608): fronts of 64-159 scalars get a workgroup each (k_factor3_tab / k_backsolve3_tab)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("opendlv-logic-cfsd18-sensation-slam_amd")
name = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
Ks = [int(k) for k in sys.argv[2:]] or [8, 16, 24]
N, M = pkg.track.CONFIGS[name]
base = None
for K in Ks:
    t = pkg.track.generate(N, M, K); fe = pkg.Graph(); g = pkg.track.bench_graph(t, fe); fe.close()
    G = pkg.Graph(); G.load_bench_graph(g); G.initialize_optimization(); st = G.stats()
    s = G.time_iterations(20)
    rate = 1e3 / s.ms_total * len(g["pl_p"])
    if base is None: base = rate
    print("K=%2d edges %8d fronts %6d (big %6d) levels %2d maxf %3d | lin %.4f factor %.4f back %.4f upd %.4f total %.4f ms -> %7.0f it/s, per-edge rate vs first %.2f (fail %d, structure %.1f ms)"
          % (K, len(g["pl_p"]), st.n_fronts, st.n_big_fronts, st.n_levels, st.max_front, s.ms_linearize, s.ms_factor, s.ms_backsolve, s.ms_update, s.ms_total,
             1e3 / s.ms_total, rate / base, s.numeric_failure, st.ms_structure), flush=True)
    G.close()
