import importlib, os, sys, time
sys.path.insert(0, "/root/repo")
pkg = importlib.import_module("opendlv-logic-cfsd18-sensation-slam_amd")
N, M = 100000, 10000
fe = pkg.Graph(); t = pkg.track.generate(N, M); g = pkg.track.bench_graph(t, fe)
G = pkg.Graph(); G.load_bench_graph(g)
ts = []
for rep in range(8):
    t0 = time.perf_counter(); G.initialize_optimization(); ts.append(1e3 * (time.perf_counter() - t0))
print("GS_THREADS=%s: structure phase of one handle, 8 times in a row: %s  (plan %.1f)" % (os.environ.get("GS_THREADS", "default"), " ".join("%.1f" % x for x in ts), G.stats().ms_plan_host))
