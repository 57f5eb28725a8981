"""Random maps / observations through the batched association (host-pointer entry point: brute force, grid, automatic; and the resident entry
point with its device-built hashed grid) against the oracle's insertion-order scan (reference src/slam.cpp:570-607) — indices bit for bit.
Clustered cones (several map entries inside one threshold ball: first match must win), wrong-colour twins, duplicates, queries on cell
borders, far queries, azimuth-0 (NaN) queries, thresholds from 0.05 to 30 m, maps of 0 .. 6 000 cones, coordinates up to 1e5 m.
usage: python scripts/stress_assoc.py [first_seed] [count]"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
pkg = importlib.import_module("opendlv-logic-cfsd18-sensation-slam_amd")
from oracle import pyoracle as po
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0; count = int(sys.argv[2]) if len(sys.argv) > 2 else 60
fe = po.OracleFrontend(); DA = pkg.binding.DeviceArray
bad = 0; queries = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(5000 + seed)
    n_map = int(rng.choice([0, 1, 2, 7, 60, 500, 3000, 6000])); n_pose = int(rng.integers(1, 40)); K = int(rng.integers(1, 30)); n = n_pose * K
    span = float(rng.choice([5.0, 60.0, 2000.0, 1e5])); thr = float(rng.choice([0.05, 0.5, 1.2, 3.0, 30.0]))
    centres = rng.uniform(-span, span, (max(n_map // 3, 1), 2))
    map_xy = (centres[rng.integers(0, len(centres), n_map)] + rng.normal(0, thr * 0.6, (n_map, 2))) if n_map else np.zeros((0, 2))
    map_type = rng.integers(1, 5, n_map).astype(np.int32)
    poses = np.concatenate([rng.uniform(-span, span, (n_pose, 2)), rng.uniform(-np.pi, np.pi, (n_pose, 1))], axis=1)
    po_ = np.repeat(np.arange(n_pose, dtype=np.int32), K)
    # observations aimed at map cones (with noise around the threshold) or at nothing; {azimuth deg, zenith deg, distance, type}
    obs = np.zeros((n, 4))
    for i in range(n):
        p = poses[po_[i]]
        if n_map and rng.random() < 0.8:
            j = int(rng.integers(n_map)); tgt = map_xy[j] + rng.normal(0, thr * 0.7, 2); ty = map_type[j] if rng.random() < 0.8 else int(rng.integers(1, 5))
        else: tgt = p[:2] + rng.uniform(-50, 50, 2); ty = int(rng.integers(1, 5))
        d = tgt - p[:2]; c, s_ = np.cos(p[2]), np.sin(p[2]); lx, ly = c * d[0] + s_ * d[1], -s_ * d[0] + c * d[1]
        obs[i] = [np.degrees(np.arctan2(ly, lx)), 0.0, np.hypot(lx, ly), ty]
    if n > 3: obs[1, 0] = 0.0; obs[2, 2] = 5e6
    ref = fe.associate(poses, po_, obs, map_xy, map_type, thr)
    G = pkg.Graph(); outs = {}
    for mode in (0, 1, -1):
        G.set_debug(assoc_grid=mode); outs[mode] = G.associate(poses, po_, obs, map_xy, map_type, thr)
    G.set_debug(assoc_grid=-1)
    if n_map:
        d_p, d_po, d_ob, d_out = DA(poses), DA(po_), DA(obs), DA(nbytes=4 * n)
        G.map_append(map_xy, map_type); G.associate_resident(d_p, n_pose, d_po, d_ob, n, thr, d_out); G.synchronize()
        outs["resident"] = d_out.to_host(np.int32, n)
        for a in (d_p, d_po, d_ob, d_out): a.free()
    G.close(); queries += n
    for k, v in outs.items():
        if not np.array_equal(v, ref):
            bad += 1; w = np.where(v != ref)[0]
            print("BAD seed", seed, "mode", k, dict(n_map=n_map, n=n, span=span, thr=thr), "first mismatch", int(w[0]), int(v[w[0]]), int(ref[w[0]]), "of", len(w), flush=True)
print("seeds %d..%d: %d queries, matched fraction of the last %.2f, %d BAD" % (first, first + count - 1, queries, float((ref >= 0).mean()) if len(ref) else 0.0, bad))
sys.exit(1 if bad else 0)
