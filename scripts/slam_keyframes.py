"""Per-keyframe latency of gs_slam_perform over one lap of a track, as the reference runs it (the graph is optimised once, at loop closure) and with
cfg.optimize_every_keyframe = 1 (optimizeGraph + updateMap at the end of every keyframe: the calls the reference carries commented out at
src/slam.cpp:403, 594, 620-621), with and without append-only growth.  usage: slam_keyframes.py [N=1000] [M=200]"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("opendlv-logic-cfsd18-sensation-slam_amd")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000; M = int(sys.argv[2]) if len(sys.argv) > 2 else 200
t = pkg.track.generate(N, M)
for label, every, env in (("reference behaviour (optimise at loop closure only)", 0, None), ("an optimisation per keyframe", 1, None), ("an optimisation per keyframe, GS_GROW=0", 1, "0")):
    if env is not None: os.environ["GS_GROW"] = env
    S = pkg.Slam(same_cone_threshold=1.2, cone_mapping_threshold=67.0, reference_quirks=0, optimize_every_keyframe=every)
    ms = []; grown = 0
    for k in list(range(N)) + list(range(10)):
        t0 = time.perf_counter(); S.perform_slam(t["odom_poses"][k], t["obs"][k]); ms.append(1e3 * (time.perf_counter() - t0))
        grown += S.graph.plan_growths() > 0
    ms = np.array(ms)
    print("%-52s %d poses / %d cones, %d keyframes: per keyframe median %.3f ms, mean %.3f, 99th percentile %.2f, max %.2f ms | keyframes whose structure phase was a growth step: %d | loop closed: %s"
          % (label + ":", N, M, len(ms), np.median(ms), ms.mean(), np.percentile(ms, 99), ms.max(), grown, S.loop_closed))
    S.close(); os.environ.pop("GS_GROW", None)
