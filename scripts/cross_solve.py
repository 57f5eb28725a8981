"""Where does the distance between the GPU's first Gauss-Newton increment and the CPU paths' come from — the linearised SYSTEM
(H, b differ in their last bits) or the SOLVE (another exact factorisation, other rounding)?  Both systems (the one the GPU
exports, the one the oracle builds) are solved by the SAME third solver (scipy SuperLU, fp64) and the four increments are
compared pairwise.  cfg4 by default."""
import importlib, os, sys, time
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import make_oracle_graph
from oracle import pyoracle as po
pkg = importlib.import_module("opendlv-logic-cfsd18-sensation-slam_amd")
name = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
N, M = pkg.track.CONFIGS[name]
t = pkg.track.generate(N, M); g = pkg.track.bench_graph(t, po.OracleFrontend())
G = pkg.Graph(); G.load_bench_graph(g); G.linearize(); sg = G.export_system(); G.optimize(1); dpg, dlg = G.export_delta(); G.close()
og = make_oracle_graph(po, g); so = og.linearize_blocks(); og.optimize(1, ordering=1); dpo, dlo = og.delta()
fp = np.zeros(N, bool); fp[g["fixed_poses"]] = True; fl = np.zeros(M, bool); fl[g["fixed_landmarks"]] = True
poff = np.cumsum(np.where(fp, 0, 3)) - np.where(fp, 0, 3); loff = 3 * int((~fp).sum()) + np.cumsum(np.where(fl, 0, 2)) - np.where(fl, 0, 2)
n = 3 * int((~fp).sum()) + 2 * int((~fl).sum())
def assemble(s):
    rows, cols, vals = [], [], []
    def add(r0, c0, blk, ok):                         # blk [K, nr, nc], r0 / c0 [K]
        K, nr, nc = blk.shape
        r = (r0[:, None, None] + np.arange(nr)[None, :, None]) * np.ones((1, 1, nc), int); c = (c0[:, None, None] + np.arange(nc)[None, None, :]) * np.ones((1, nr, 1), int)
        rows.append(r[ok].ravel()); cols.append(c[ok].ravel()); vals.append(blk[ok].ravel())
    add(poff, poff, s["Hpp_diag"].reshape(N, 3, 3), ~fp); add(loff, loff, s["Hll_diag"].reshape(M, 2, 2), ~fl)
    i, j = g["pp_i"], g["pp_j"]; ok = ~fp[i] & ~fp[j]; B = s["Hpp_off"].reshape(-1, 3, 3)
    add(poff[i], poff[j], B, ok); add(poff[j], poff[i], B.transpose(0, 2, 1), ok)
    p, l = g["pl_p"], g["pl_l"]; ok = ~fp[p] & ~fl[l]; B = s["Hpl"].reshape(-1, 3, 2)
    add(poff[p], loff[l], B, ok); add(loff[l], poff[p], B.transpose(0, 2, 1), ok)
    H = sp.csc_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n, n))
    b = np.zeros(n); b[(poff[~fp][:, None] + np.arange(3)).ravel()] = s["b_pose"][~fp].ravel(); b[(loff[~fl][:, None] + np.arange(2)).ravel()] = s["b_lm"][~fl].ravel()
    return H, b
def unpack(x):
    dp = np.zeros((N, 3)); dl = np.zeros((M, 2)); dp[~fp] = x[(poff[~fp][:, None] + np.arange(3))]; dl[~fl] = x[(loff[~fl][:, None] + np.arange(2))]; return dp, dl
res = {"gpu_increment": (dpg, dlg), "oracle_increment": (dpo, dlo)}
for label, s in (("superlu_on_the_gpu_system", sg), ("superlu_on_the_oracle_system", so)):
    H, b = assemble(s); t0 = time.time(); x = spla.splu(H).solve(b)
    r = np.abs(H @ x - b).max() / np.abs(b).max()
    print("%s: %.1f s, residual %.2g" % (label, time.time() - t0, r), flush=True); res[label] = unpack(x)
keys = list(res)
print("max |dx| %.3g m" % np.abs(dpo[:, :2]).max())
for a in range(len(keys)):
    for b_ in range(a + 1, len(keys)):
        d = res[keys[a]][0][:, :2] - res[keys[b_]][0][:, :2]
        print("%-30s vs %-30s: max %.3g m, rms %.3g m" % (keys[a], keys[b_], np.abs(d).max(), np.sqrt((d ** 2).sum(1).mean())))
