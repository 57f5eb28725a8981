"""Ad-hoc GPU probe: per-phase timings and convergence for the BASELINE configs."""
import importlib, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
pkg = importlib.import_module("opendlv-logic-cfsd18-sensation-slam_amd")
names = sys.argv[1:] or ["cfg2", "cfg3", "cfg4"]
fe = pkg.Graph()
for name in names:
    N, M = pkg.track.CONFIGS[name]
    t0 = time.time(); t = pkg.track.generate(N, M); g = pkg.track.bench_graph(t, fe); t1 = time.time()
    G = pkg.Graph(); G.load_bench_graph(g); t2 = time.time()
    G.initialize_optimization(); t3 = time.time()
    st = G.stats()
    print(f"{name}: gen+graph {t1-t0:.2f}s load {t2-t1:.2f}s init {t3-t2:.3f}s fronts {st.n_fronts} levels {st.n_levels} maxf {st.max_front} "
          f"flops {st.factor_flops/1e9:.3f}G bytes {st.factor_bytes/1e6:.1f}MB", flush=True)
    ms = G.time_linearize(20); B = G.linearize_bytes()
    print(f"  linearize {ms*1e3:.1f} us  alg bytes {B/1e6:.2f} MB -> {B/ms/1e6:.1f} GB/s ({B/ms/1e6/8000*100:.2f}% of 8 TB/s)")
    s = G.time_iterations(5)
    print(f"  per-iter ms: lin {s.ms_linearize:.3f} factor {s.ms_factor:.3f} back {s.ms_backsolve:.3f} upd {s.ms_update:.3f} total {s.ms_total:.3f}  -> {1e3/s.ms_total:.1f} it/s")
    chis = []
    for it in range(16):
        done, so = G.optimize(1); dp, dl = G.export_delta()
        chis.append((so.chi2_initial, np.abs(dp).max()))
    print("  chi2/maxdx:", " ".join(f"{c:.4g}/{d:.2g}" for c, d in chis))
    tp = t["truth_poses"]; P = G.poses()
    print("  rmse vs truth", np.sqrt(((P[:, :2]-tp[:, :2])**2).sum(1).mean()))
    G.close()
