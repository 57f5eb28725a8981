"""Random keyframe streams through the append-only growth (gs::grow_plan + upload_growth) against a fresh full build of the same graph, on the GPU:
laps of random size cut at a random pose, the tail appended in random batches with an optimisation between them, the grown handle against a fresh
handle on the final graph (and the oracle for the smaller ones).  A refusal (a front would outgrow a wave, the 17th pose ...) is a full rebuild —
allowed, counted; a wrong answer is not.  The fixed-size version: tests/test_gpu_parity.py::test_appended_poses_are_absorbed_by_the_plan...
usage: python scripts/stress_growth.py [first_seed] [count]"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import append_tail, make_oracle_graph, split_for_growth
pkg = importlib.import_module("opendlv-logic-cfsd18-sensation-slam_amd")
pkg.binding.DEFAULT_DEBUG["grow_min_poses"] = 0
from oracle import pyoracle as po
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0; count = int(sys.argv[2]) if len(sys.argv) > 2 else 30
fe = po.OracleFrontend(); bad = grown_steps = rebuilt_steps = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(3000 + seed)
    N = int(rng.choice([60, 150, 400, 1000, 3000, 10000])); M = max(30, N // int(rng.integers(4, 7)))
    try: t = pkg.track.generate(N, M)
    except ValueError: continue
    g = pkg.track.bench_graph(t, fe)
    h = int(rng.integers(1, 25)); keep = None if rng.random() < 0.5 else int(rng.integers(max(h + 20, N // 2), N))
    try: base, tail, full = split_for_growth(g, h, keep)
    except AssertionError: continue
    G = pkg.Graph(); G.load_bench_graph(base); G.optimize(int(rng.integers(0, 4)))
    a = 0
    while a < h:
        b = min(h, a + int(rng.integers(1, 6)))
        append_tail(G, tail, (a, b)); before = G.plan_growths()
        its = int(rng.integers(0, 3)); G.optimize(its) if its else G.initialize_optimization()
        if G.plan_growths() == before + 1: grown_steps += 1
        else: rebuilt_steps += 1
        a = b
    F = pkg.Graph(); F.load_bench_graph(full)
    # the two handles hold different iterates now (the grown one optimised on the way): give both the same start, then the same 6 iterations
    ok = True; why = ""
    if True:
        Pg, Lg = G.poses(), G.landmarks()
        for i in range(len(Pg)): F.set_pose_estimate(i, Pg[i])
        for l in range(len(Lg)): F.set_landmark_estimate(l, Lg[l])
        dG, sG = G.optimize(6); dF, sF = F.optimize(6)
        rms = float(np.sqrt((F.poses()[:, :2] ** 2).sum(1).mean()))
        e = max(np.sqrt(((G.poses()[:, :2] - F.poses()[:, :2]) ** 2).sum(1).mean()), np.sqrt(((G.landmarks() - F.landmarks()) ** 2).sum(1).mean())) / rms
        ok = dG == dF == 6 and e < 1e-6 and abs(sG.chi2_final - sF.chi2_final) <= 1e-5 * max(sF.chi2_final, 1e-9)
        why = "err %.2e chi2 %.6g vs %.6g done %d %d" % (e, sG.chi2_final, sF.chi2_final, dG, dF)
    if not ok: bad += 1; print("BAD seed", seed, dict(N=N, M=M, h=h, keep=keep), why, "refusal:", G.growth_refusal(), flush=True)
    G.close(); F.close()
print("seeds %d..%d: %d growth steps absorbed, %d rebuilt, %d BAD" % (first, first + count - 1, grown_steps, rebuilt_steps, bad))
sys.exit(1 if bad else 0)
