"""Randomised laps through the Slam mirror (csrc/gs_slam.cpp over the HIP C-ABI) against tests/ref_slam.py (the reference's performSLAM /
addConesToMap / localizer restated over the CPU oracle), frame by frame — the fixed-seed version of this is
tests/test_gpu_parity.py::test_slam_mirror_frame_by_frame_matches_reference_logic.  Random lap size, thresholds, observation noise, dropped
cones (frames of 1 .. K cones), yaw rates and sample-time gaps.  usage: python scripts/stress_slam.py [first_seed] [count]"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
pkg = importlib.import_module("opendlv-logic-cfsd18-sensation-slam_amd")
pkg.binding.DEFAULT_DEBUG["grow_min_poses"] = 0
from ref_slam import RefSlam
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0; count = int(sys.argv[2]) if len(sys.argv) > 2 else 30
bad = 0; closed = 0; frames_total = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(9000 + seed)
    N = int(rng.integers(60, 320)); M = int(rng.integers(max(30, N // 4), max(40, N // 2))); quirks = int(rng.integers(0, 2))
    same = float(rng.uniform(0.9, 1.4)); mapping = float(rng.uniform(25.0, 80.0))
    try: t = pkg.track.generate(N, M)
    except ValueError: continue
    S = pkg.Slam(same_cone_threshold=same, cone_mapping_threshold=mapping, reference_quirks=quirks)
    R = RefSlam(same_cone_threshold=same, cone_mapping_threshold=mapping, quirks=bool(quirks))
    frames = list(range(N)) + list(range(int(rng.integers(2, 10))))
    ok = True; where = None
    for n, k in enumerate(frames):
        obs = np.array(t["obs"][k], dtype=float).copy()
        obs[:, 2] *= 1.0 + rng.normal(0, 0.002, len(obs))                 # range noise
        keep = rng.random(len(obs)) > 0.15
        if keep.sum() == 0: keep[int(rng.integers(len(obs)))] = True
        obs = obs[keep]
        wz = float(np.float32(rng.normal(0, 0.3))); dt_us = int(rng.choice([0, 40000, 250000, 1500000]))
        for X in (S, R):
            X.next_yaw_rate(wz); X.set_sample_times(10_000_000 + 100_000 * n + dt_us, 10_000_000 + 100_000 * n)
        S.perform_slam(t["odom_poses"][k], obs); R.perform(t["odom_poses"][k], obs)
        frames_total += 1
        same_state = (S.map_size == len(R.map) and S.loop_closed == R.loop_closing_complete and S.current_cone_index == R.current_cone_index
                      and S.graph.n_pl == R.g.n_pl and S.graph.n_pp == R.g.n_pp and np.abs(S.send_pose() - R.send_pose).max() < 1e-6)
        if not same_state: ok = False; where = (n, S.map_size, len(R.map), S.loop_closed, R.loop_closing_complete, S.graph.n_pl, R.g.n_pl, float(np.abs(S.send_pose() - R.send_pose).max())); break
    if ok:
        xy, ty = S.map(); Rm = np.array([[c[0], c[1]] for c in R.map]); Rt = np.array([c[2] for c in R.map])
        ok = np.array_equal(ty, Rt) and np.abs(xy - Rm).max() < 1e-6 and np.abs(S.graph.poses() - R.g.poses()).max() < 1e-6
        if not ok: where = ("final", float(np.abs(xy - Rm).max()), float(np.abs(S.graph.poses() - R.g.poses()).max()))
    closed += bool(S.loop_closed)
    if not ok: bad += 1; print("BAD seed", seed, dict(N=N, M=M, quirks=quirks, same=round(same, 3), mapping=round(mapping, 1)), where, flush=True)
    S.close()
print("seeds %d..%d: %d frames, loop closed in %d laps, %d BAD" % (first, first + count - 1, frames_total, closed, bad))
sys.exit(1 if bad else 0)
