"""Append-only growth at the headline size: a cfg4 track whose last pose arrives after the structure phase.  Times the full structure phase,
the growth step (gs_initialize_optimization after one more pose + its odometry edge + its K observation edges, reference
src/slam.cpp:433-459, 537-550), the iteration before / after, and checks the grown handle against a fresh full build.
usage: growth_time.py [cfg4] [poses_held_back=1] [batches=1] [keep=0: whole lap; else only the first `keep` poses, whose last ones discover cones]"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import append_tail, split_for_growth
pkg = importlib.import_module("opendlv-logic-cfsd18-sensation-slam_amd")
name = sys.argv[1] if len(sys.argv) > 1 else "cfg4"; h = int(sys.argv[2]) if len(sys.argv) > 2 else 1; batches = int(sys.argv[3]) if len(sys.argv) > 3 else 1; keep = int(sys.argv[4]) if len(sys.argv) > 4 and int(sys.argv[4]) > 0 else None
N, M = pkg.track.CONFIGS[name]
t = pkg.track.generate(N, M); fe = pkg.Graph(); g = pkg.track.bench_graph(t, fe)
base, tail, full = split_for_growth(g, h, keep)
G = pkg.Graph(); G.load_bench_graph(base); G.initialize_optimization(); G.initialize_optimization()
s_full = G.stats(); it0 = G.time_iterations(20)
per = h // batches; grow_ms = []; new_lms = 0
for k in range(batches):
    new_lms += append_tail(G, tail, (k * per, h if k == batches - 1 else (k + 1) * per))
    t0 = time.perf_counter(); G.initialize_optimization(); grow_ms.append(1e3 * (time.perf_counter() - t0))
    assert G.plan_growths() == k + 1, G.growth_refusal()
it1 = G.time_iterations(20); s_grown = G.stats()
done, _ = G.optimize(10)
F = pkg.Graph(); F.load_bench_graph(full); done_f, _ = F.optimize(10)
rms = np.sqrt((F.poses()[:, :2] ** 2).sum(1).mean())
d = np.sqrt(((G.poses()[:, :2] - F.poses()[:, :2]) ** 2).sum(1).mean()) / rms
print("%s%s, last %d pose(s) (+ %d new cones) appended in %d batch(es): full structure phase %.2f ms | growth step(s) %s ms (wall; stats.ms_structure %.3f) | iteration %.4f -> %.4f ms "
      "(lin %.4f -> %.4f, factor %.4f -> %.4f, back %.4f -> %.4f) | fronts %d, max front %d -> %d | 10 iterations: pose RMSE vs a fresh full build, relative %.2e (%d / %d applied)"
      % (name, "" if keep is None else " (first %d poses)" % keep, h, new_lms, batches, s_full.ms_structure, " ".join("%.3f" % v for v in grow_ms), s_grown.ms_structure, it0.ms_total, it1.ms_total, it0.ms_linearize, it1.ms_linearize,
         it0.ms_factor, it1.ms_factor, it0.ms_backsolve, it1.ms_backsolve, s_grown.n_fronts, s_full.max_front, s_grown.max_front, d, done, done_f))
