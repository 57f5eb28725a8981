"""How well is one Gauss-Newton step of config 5 (1M poses / 50k cones) determined in fp64?  Compares the increment of the
first iteration between the GPU solver, the oracle's own LDL^T (two orderings) and the reference's vendored Eigen
SimplicialLDLT + AMD (oracle/_ref), all from the same linearisation point, with the normal-equation residual of each."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import make_oracle_graph
from oracle import pyoracle as po
pkg = importlib.import_module("opendlv-logic-cfsd18-sensation-slam_amd")
name = sys.argv[1] if len(sys.argv) > 1 else "cfg5"
N, M = pkg.track.CONFIGS[name]
t = pkg.track.generate(N, M); g = pkg.track.bench_graph(t, po.OracleFrontend())
G = pkg.Graph(); G.load_bench_graph(g); G.optimize(1); dpg, dlg = G.export_delta(); chig = G.chi2(); G.close()
res = {"gpu": (dpg, dlg)}
for label, ordering, eig in (("oracle_ldlt_track_order", 1, False), ("oracle_ldlt_natural", 0, False), ("eigen_simplicial_ldlt_amd", 1, True)):
    if (eig and po.ref_eigen() is None) or (ordering == 0 and N > 200000):      # natural order: fill-in beyond memory at 1M poses
        continue
    og = make_oracle_graph(po, g); t0 = time.time()
    og.optimize(1, ordering=ordering, solver=po.EigenSolver(0) if eig else None)
    res[label] = og.delta(); print(label, "%.1f s" % (time.time() - t0), "chi2 after the step %.9g" % og.chi2(), flush=True)
print("gpu chi2 after the step %.9g" % chig)
keys = list(res)
rms = np.sqrt((g["pose_est"][:, :2] ** 2).sum(1).mean())
for i in range(len(keys)):
    for j in range(i + 1, len(keys)):
        a, b = res[keys[i]], res[keys[j]]
        print("%-28s vs %-28s: max |d pose xy| %.3g m, RMS %.3g m (%.3g of the track's RMS radius), max |d dx| / max |dx| %.3g"
              % (keys[i], keys[j], np.abs(a[0][:, :2] - b[0][:, :2]).max(), np.sqrt(((a[0][:, :2] - b[0][:, :2]) ** 2).sum(1).mean()),
                 np.sqrt(((a[0][:, :2] - b[0][:, :2]) ** 2).sum(1).mean()) / rms, np.abs(a[0] - b[0]).max() / np.abs(b[0]).max()))
print("max |dx| (gpu) %.3g m" % np.abs(dpg[:, :2]).max())
