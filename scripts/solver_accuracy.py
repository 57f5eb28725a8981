"""Where does the first-iteration difference between the GPU increment and the CPU increments come from: the linearised
system (H, b differ in the last bits: device vs host transcendental functions, summation order) or the solve?
Prints block differences and the residual of each increment against EACH system (cfg4 by default)."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import make_oracle_graph
from test_gpu_parity import normal_equation_residual
from oracle import pyoracle as po
pkg = importlib.import_module("opendlv-logic-cfsd18-sensation-slam_amd")
name = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
N, M = pkg.track.CONFIGS[name]
t = pkg.track.generate(N, M); g = pkg.track.bench_graph(t, po.OracleFrontend())
G = pkg.Graph(); G.load_bench_graph(g); G.linearize(); sg = G.export_system(); G.optimize(1); dpg, dlg = G.export_delta(); G.close()
og = make_oracle_graph(po, g); so = og.linearize_blocks()
for k in ("Hpp_diag", "Hll_diag", "Hpp_off", "Hpl", "b_pose", "b_lm"):
    a, b = sg[k], so[k]
    # the oracle leaves blocks of fixed vertices in place, the export zeroes them: compare where both are non-zero
    m = (np.abs(a) > 0) & (np.abs(b) > 0)
    print("%-9s max |gpu - oracle| / max |oracle| = %.3g   (median relative difference of the entries %.3g)" % (k, np.abs(a - b)[m].max() / np.abs(b).max(), np.median(np.abs(a - b)[m] / np.abs(b[m]))))
og.optimize(1, ordering=1, solver=po.EigenSolver(0) if po.ref_eigen() is not None else None); dpo, dlo = og.delta()
# oracle system in export form: zero the blocks of fixed vertices like the export does
so2 = {k: v.copy() for k, v in so.items()}
fp, fl = g["fixed_poses"], g["fixed_landmarks"]
so2["Hpp_diag"][fp] = 0; so2["b_pose"][fp] = 0; so2["Hll_diag"][fl] = 0; so2["b_lm"][fl] = 0
so2["Hpp_off"][np.isin(g["pp_i"], fp) | np.isin(g["pp_j"], fp)] = 0; so2["Hpl"][np.isin(g["pl_p"], fp) | np.isin(g["pl_l"], fl)] = 0
print("residual |H dx - b| / |b|:")
print("   gpu increment    against the gpu system %.3g, against the oracle system %.3g" % (normal_equation_residual(g, sg, dpg, dlg), normal_equation_residual(g, so2, dpg, dlg)))
print("   oracle increment against the gpu system %.3g, against the oracle system %.3g" % (normal_equation_residual(g, sg, dpo, dlo), normal_equation_residual(g, so2, dpo, dlo)))
print("increments: max |gpu - oracle| %.3g m of max |dx| %.3g m" % (np.abs(dpg[:, :2] - dpo[:, :2]).max(), np.abs(dpo[:, :2]).max()))
