"""Phase timestamps of one front inside the variant-3 factor / backsolve kernels.
usage: GS_FACTOR_VARIANT=3 GS_DBG=$((8 | (COUNT << 8))) python scripts/ts_probe.py [cfg]   (COUNT = fronts in the probed level)"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("opendlv-logic-cfsd18-sensation-slam_amd")
name = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
N, M = pkg.track.CONFIGS[name]
t = pkg.track.generate(N, M); fe = pkg.Graph(); g = pkg.track.bench_graph(t, fe)
G = pkg.Graph(); G.load_bench_graph(g)
if len(sys.argv) > 2:                                            # probe the first front of tree level argv[2] (whole-tree launch): GS_DBG is set here
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    from plan_exec import Plan
    H = pkg.Graph(device=-2); H.load_bench_graph(g); H.plan_build_host(); P = Plan(H.plan_export()); H.close()
    os.environ["GS_DBG"] = str(16 | ((int(P.level_start[int(sys.argv[2])]) + (int(sys.argv[3]) if len(sys.argv) > 3 else 0)) << 8))   # argv[3]: position inside the level
G.initialize_optimization()
for _ in range(4):
    G.iterate()
G.synchronize()
ts = G.debug_timestamps()
f = ts[0:9]; b = ts[32:39]
print("factor  phases (us):", [round((int(f[i + 1]) - int(f[i])) / 100.0, 2) for i in range(8)], "total", (int(f[8]) - int(f[0])) / 100.0)
if ts[9] and ts[10]:
    print("   inside phase 5 (us): children 0/1 gathered %.2f | accumulators built + summed %.2f | further children %.2f" % ((int(ts[9]) - int(f[5])) / 100.0, (int(ts[10]) - int(ts[9])) / 100.0, (int(f[6]) - int(ts[10])) / 100.0))
print("   0 desc | 1 issue rec+pinv | 2 wait | 3 zero LDS | 4 gather issue+wait | 5 originals | 6 acc sum | 7 panels | 8 U out")
print("backsolve phases (us):", [round((int(b[i + 1]) - int(b[i])) / 100.0, 2) for i in range(6)], "total", (int(b[6]) - int(b[0])) / 100.0)
print("   32 desc | 33 L->LDS | 34 xe gather | 35 boundary mat-vec | 36 substitution | 37 store")
for k, name in enumerate(("first", "middle", "last")):
    l = ts[40 + 8 * k: 40 + 8 * k + 7]
    if l[0]:
        print("linearise wave tile %-6s (us, LIN_TS build):" % name, [round((int(l[i + 1]) - int(l[i])) / 100.0, 2) for i in range(6)], "total", (int(l[6]) - int(l[0])) / 100.0)
print("   0 prologue | 1 edge slots 0-1 | 2 edge slots 2-3 | 3 odometry incidences | 4 pose sums | 5 landmark groups")
t0 = min(int(ts[40 + 8 * k]) for k in range(3) if ts[40 + 8 * k])  if any(ts[40 + 8 * k] for k in range(3)) else 0
for k, name in enumerate(("first", "middle", "last")):
    if ts[40 + 8 * k]:
        print("   wave tile %-6s starts at %+.2f us, ends at %+.2f us (relative to the earliest start of the three)" % (name, (int(ts[40 + 8 * k]) - t0) / 100.0, (int(ts[46 + 8 * k]) - t0) / 100.0))
