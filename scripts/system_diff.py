"""Where do the GPU's and the oracle's linearised systems differ most?  (cfg4, initial point; GS_HOST_TRIG=1 to take the device trig out)"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import make_oracle_graph
from oracle import pyoracle as po
pkg = importlib.import_module("opendlv-logic-cfsd18-sensation-slam_amd")
name = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
N, M = pkg.track.CONFIGS[name]
t = pkg.track.generate(N, M); g = pkg.track.bench_graph(t, po.OracleFrontend())
G = pkg.Graph(); G.load_bench_graph(g); G.linearize(); sg = G.export_system(); G.close()
og = make_oracle_graph(po, g); so = og.linearize_blocks()
for k in ("b_pose", "Hpp_diag", "Hpl", "b_lm", "Hpp_off"):
    a, b = sg[k], so[k]; m = (np.abs(a) > 0) & (np.abs(b) > 0); d = np.where(m, np.abs(a - b), 0.0)
    top = np.argsort(d.max(axis=1))[-6:][::-1]
    print(k, "scale", np.abs(b).max())
    for i in top: print("   row", i, "abs diff", d[i], "| oracle", b[i], "| pose theta" if k in ("b_pose", "Hpp_diag") else "", g["pose_est"][i, 2] if k in ("b_pose", "Hpp_diag") else "")
    comp = d.max(axis=0); print("   max abs diff per component", comp)
