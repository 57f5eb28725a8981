#!/usr/bin/env python3
"""Generates tests/golden/cfg1_oracle.npz: BASELINE config 1 (50 poses / 30 cones) inputs and the CPU
oracle's outputs (H blocks, b, increment of iteration 0, chi2 history, estimates after 10 iterations).

The reference repository holds no golden vectors, fixtures or recorded data for this path (SURVEY.md §4,
§8c), so this file is produced by OUR oracle and pins regressions of oracle + generator + GPU path; it is
not reference output.  Run from the repo root:  python tests/golden/make_golden.py
"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import make_oracle_graph  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

pkg = importlib.import_module("opendlv-logic-cfsd18-sensation-slam_amd")
po.build()
t = pkg.track.generate(50, 30)
g = pkg.track.bench_graph(t, po.OracleFrontend())
out = {"in_" + k: v for k, v in g.items()}
out["track_obs"] = t["obs"]; out["track_odom"] = t["odom_poses"]; out["track_truth"] = t["truth_poses"]
og = make_oracle_graph(po, g)
for k, v in og.linearize_blocks().items():
    out["out_" + k] = v
og.build_system(); og.apply_update(og.solve_ldlt(1))
out["out_dpose_it0"], out["out_dlm_it0"] = og.delta()
og2 = make_oracle_graph(po, g)
done, chi, _ = og2.optimize(10, ordering=1)
out["out_chi2"] = chi; out["out_poses_it10"] = og2.poses(); out["out_lms_it10"] = og2.landmarks()
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cfg1_oracle.npz")
np.savez_compressed(path, **out)
print("wrote", path, os.path.getsize(path), "bytes")
