"""Is a lap-sized iteration bound by the host's enqueue or by the GPU?  Host time to ENQUEUE n iterations against the time until they are DONE.
usage: python scripts/enqueue_probe.py [poses cones]"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("opendlv-logic-cfsd18-sensation-slam_amd")
N, M = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (240, 200)
t = pkg.track.generate(N, M); fe = pkg.Graph(); g = pkg.track.bench_graph(t, fe); fe.close()
G = pkg.Graph(); G.load_bench_graph(g); G.initialize_optimization()
for _ in range(50): G.iterate()
G.synchronize()
for n in (10, 100, 1000):
    best = None
    for rep in range(5):
        t0 = time.perf_counter()
        for _ in range(n): G.iterate()
        t1 = time.perf_counter(); G.synchronize(); t2 = time.perf_counter()
        if best is None or t2 - t0 < best[1]: best = (t1 - t0, t2 - t0)
    print("%d:%d  %4d iterations: enqueued in %.1f us each, done in %.1f us each" % (N, M, n, best[0] / n * 1e6, best[1] / n * 1e6))
done, st = G.optimize(10)
t0 = time.perf_counter(); done, st = G.optimize(10); t1 = time.perf_counter()
print("gs_optimize(10): %.1f us" % ((t1 - t0) * 1e6))
