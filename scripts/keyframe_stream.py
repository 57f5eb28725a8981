"""A stream of keyframes on a mapped stretch: the first `keep` poses of a track are in the graph and optimised; then one keyframe at a time arrives
(a pose, its odometry edge, its observation edges, the cones it is the first to see — reference src/slam.cpp:433-459, 525-550) and the whole graph is
optimised again (the call the reference has commented out at :594 / :620-621).  Per keyframe: wall time of gs_optimize(10) — structure phase or growth
step, 10 iterations, estimates back.  Run once as it is and once with GS_GROW=0; the estimates of the two runs must agree.
usage: keyframe_stream.py [cfg3 | N:M] [keep=6000] [keyframes=24]"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import append_tail, split_for_growth
pkg = importlib.import_module("opendlv-logic-cfsd18-sensation-slam_amd")
name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"; N, M = pkg.track.CONFIGS[name] if name in pkg.track.CONFIGS else tuple(int(v) for v in name.split(":"))
keep = int(sys.argv[2]) if len(sys.argv) > 2 else int(0.6 * N); K = int(sys.argv[3]) if len(sys.argv) > 3 else 24
t = pkg.track.generate(N, M); fe = pkg.Graph(); g = pkg.track.bench_graph(t, fe)
base, tail, full = split_for_growth(g, K, keep + K)
out = {}
for mode in ("grow", "rebuild"):
    if mode == "rebuild": os.environ["GS_GROW"] = "0"
    G = pkg.Graph(); G.load_bench_graph(base); G.optimize(10)
    ms, struct, grew, cones = [], [], 0, 0
    for k in range(K):
        cones += append_tail(G, tail, (k, k + 1))
        t0 = time.perf_counter(); done, st = G.optimize(10); ms.append(1e3 * (time.perf_counter() - t0))
        assert done == 10
        struct.append(st.ms_structure); grew += st.n_growths > 0
    out[mode] = (G.poses().copy(), G.landmarks().copy())
    print("%-8s %s, %d poses mapped, then %d keyframes (+%d cones), gs_optimize(10) after each: per keyframe median %.2f ms, mean %.2f ms, total %.1f ms | structure part: "
          "median %.3f ms, mean %.2f ms | %d of %d keyframes absorbed by the plan | per keyframe [ms]: %s"
          % (mode, name, keep, K, cones, np.median(ms), np.mean(ms), np.sum(ms), np.median(struct), np.mean(struct), grew, K, " ".join("%.1f" % v for v in ms)))
    G.close()
os.environ.pop("GS_GROW", None)
rms = np.sqrt((out["rebuild"][0][:, :2] ** 2).sum(1).mean())
print("grown vs rebuilt after the stream: pose RMSE rel %.2e, landmark RMSE rel %.2e"
      % (np.sqrt(((out["grow"][0][:, :2] - out["rebuild"][0][:, :2]) ** 2).sum(1).mean()) / rms, np.sqrt(((out["grow"][1] - out["rebuild"][1]) ** 2).sum(1).mean()) / rms))
