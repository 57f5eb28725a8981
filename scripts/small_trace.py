"""One lap-sized graph (default 240 poses / 200 cones), gs_optimize(10) repeated: the workload for a kernel trace of the
launch-bound regime (rocprofv3 --kernel-trace --stats -- python3 scripts/small_trace.py).  usage: small_trace.py [N M [calls]]"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("opendlv-logic-cfsd18-sensation-slam_amd")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 240; M = int(sys.argv[2]) if len(sys.argv) > 2 else 200
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 20
fe = pkg.Graph(); t = pkg.track.generate(N, M); g = pkg.track.bench_graph(t, fe)
G = pkg.Graph(); G.load_bench_graph(g); G.optimize(10)
t0 = time.perf_counter()
for _ in range(calls):
    G.optimize(10)
dt = time.perf_counter() - t0
st = G.time_iterations(20)
print("%d:%d  gs_optimize(10): %.3f ms per call (%.1f us per iteration); event-timed phases per iteration [ms]: lin %.4f factor %.4f back %.4f upd %.4f total %.4f"
      % (N, M, 1e3 * dt / calls, 1e5 * dt / calls, st.ms_linearize, st.ms_factor, st.ms_backsolve, st.ms_update, st.ms_total))
