"""Batched association with everything resident (gs_associate_resident): kernel time at a config (bench.py's `association` entry alone)."""
import importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
pkg = importlib.import_module("opendlv-logic-cfsd18-sensation-slam_amd")
name = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
N, M = pkg.track.CONFIGS[name]
t = pkg.track.generate(N, M); fe = pkg.Graph(); g = pkg.track.bench_graph(t, fe); fe.close()
print(json.dumps(bench.association_roofline(pkg, np, t, g, 0)))
