"""CPU tests of the host-side structure phase (csrc/gs_plan.cpp) through the C-ABI's host-only handle.

The plan (nested-dissection order, symbolic factorisation, assembly records, extend-add maps) is checked
(a) for the invariants any valid multifrontal plan satisfies and (b) numerically: tests/plan_exec.py
replays the plan in numpy on the oracle's H blocks and must reproduce the oracle's joint-system increment.
No HIP code runs here."""
import os

import numpy as np
import pytest

from conftest import append_tail, make_oracle_graph, random_graph, split_for_growth
from plan_exec import Plan


def host_graph(pkg, g, **kw):
    G = pkg.Graph(device=-2, **kw)
    G.load_bench_graph(g)
    return G


def oracle_increment(po, g, ordering=1):
    og = make_oracle_graph(po, g)
    blocks = og.linearize_blocks()
    og.build_system(); og.apply_update(og.solve_ldlt(ordering))
    return blocks, og.delta()


@pytest.mark.parametrize("N,M,leaf", [(50, 30, 0), (50, 30, 1), (1000, 200, 0), (1000, 200, 3), (1000, 200, 64)])
def test_track_plan_invariants_and_numeric_replay(pkg, po, bench_graphs, N, M, leaf):
    _, g = bench_graphs(N, M)
    G = host_graph(pkg, g, leaf_poses=leaf)
    info = G.plan_build_host()
    assert info.n_scalar == 3 * (N - 2) + 2 * (len(g["lm_est"]) - 2)
    P = Plan(G.plan_export()); P.check_invariants()
    assert P.n_fronts == info.n_fronts and P.max_front == info.max_front
    blocks, (dp_o, dl_o) = oracle_increment(po, g)
    dp, dl, ok = P.solve(blocks)
    scale = max(np.abs(dp_o).max(), np.abs(dl_o).max())
    assert ok and np.abs(dp - dp_o).max() / scale < 1e-8 and np.abs(dl - dl_o).max() / scale < 1e-8
    G.close()


def test_track_plan_is_a_shallow_tree_of_small_fronts(pkg, bench_graphs):
    """The design premise: separators of the joint pose+cone graph are tiny, so fronts stay
    ~50 scalars wide and the tree has ~log2(N/leaf) levels."""
    _, g = bench_graphs(10000, 2000)
    G = host_graph(pkg, g)
    info = G.plan_build_host()
    assert info.max_front <= 96 and info.n_levels <= 16
    P = Plan(G.plan_export()); P.check_invariants()
    assert (P.npiv + P.nbnd).mean() < 64
    G.close()


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_irregular_graph_plan_numeric_replay(pkg, po, seed):
    """Not a track: random observations, extra pose-pose edges spanning the split, duplicate parallel edges,
    anisotropic information.  Ordering is a heuristic, the symbolic phase must still be exact."""
    g = random_graph(seed)
    for leaf in (0, 2):
        G = host_graph(pkg, g, leaf_poses=leaf)
        G.plan_build_host()
        P = Plan(G.plan_export()); P.check_invariants()
        assert P.asm_dup.sum() >= 1                      # the parallel duplicate edges were detected
        blocks, (dp_o, dl_o) = oracle_increment(po, g, ordering=0)
        dp, dl, ok = P.solve(blocks)
        scale = max(np.abs(dp_o).max(), np.abs(dl_o).max())
        assert ok and np.abs(dp - dp_o).max() / scale < 1e-9 and np.abs(dl - dl_o).max() / scale < 1e-9
        G.close()


def test_pose_only_chain_and_isolated_landmark(pkg, po):
    rng = np.random.default_rng(0)
    n = 30
    g = dict(pose_est=np.cumsum(rng.normal(0.5, 0.1, (n, 3)), axis=0), lm_est=np.array([[1.0, 2.0], [4.0, 1.0]]),
             pp_i=np.arange(n - 1, dtype=np.int32), pp_j=np.arange(1, n, dtype=np.int32), pp_z=rng.normal(0.5, 0.1, (n - 1, 3)),
             pp_info=np.tile(np.eye(3).reshape(1, 9), (n - 1, 1)),
             # landmark 0 is seen only from the FIXED pose: it couples to no free vertex; landmark 1 from pose 7
             pl_p=np.array([0, 7], dtype=np.int32), pl_l=np.array([0, 1], dtype=np.int32), pl_z=np.array([[1.0, 2.1], [0.3, 0.2]]),
             pl_info=np.tile(np.eye(2).reshape(1, 4), (2, 1)), fixed_poses=np.array([0], dtype=np.int32),
             fixed_landmarks=np.array([], dtype=np.int32))
    G = host_graph(pkg, g, leaf_poses=2); G.plan_build_host()
    P = Plan(G.plan_export()); P.check_invariants()
    blocks, (dp_o, dl_o) = oracle_increment(po, g, ordering=0)
    dp, dl, ok = P.solve(blocks)
    assert ok and np.abs(dp - dp_o).max() < 1e-10 and np.abs(dl - dl_o).max() < 1e-10
    G.close()


def test_no_free_vertex_is_an_error_not_a_crash(pkg):
    G = pkg.Graph(device=-2)
    G.add_pose(0, [0, 0, 0]); G.set_fixed_pose(0)
    with pytest.raises(pkg.GsError) as e:
        G.plan_build_host()
    assert e.value.code == -7                           # GS_ERR_EMPTY
    G.close()


def test_plan_edge_orders_are_permutations(pkg, bench_graphs):
    _, g = bench_graphs(1000, 200)
    G = host_graph(pkg, g); G.plan_build_host(); P = Plan(G.plan_export())
    assert sorted(P.pl_order.tolist()) == list(range(len(g["pl_p"])))
    assert np.all(np.diff(g["pl_p"][P.pl_order]) >= 0)   # device order: grouped by pose
    assert sorted(P.pp_order.tolist()) == list(range(len(g["pp_i"])))
    G.close()


def test_plan_is_the_same_for_any_number_of_host_threads(pkg):
    """The structure phase runs on several host threads (csrc/gs_parallel.hpp: adjacency, the top levels of the nested
    dissection, wave tiles, fronts; round 4: grouping, window masks, numbering, fills); the plan — of the whole graph and of a rank of 8 —
    must not depend on how many: same bytes with GS_THREADS = 1, 3 and 8."""
    import hashlib
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import importlib, sys, hashlib; sys.path.insert(0, %r); from oracle import pyoracle as po; "
            "pkg = importlib.import_module('opendlv-logic-cfsd18-sensation-slam_amd'); "
            "t = pkg.track.generate(10000, 2000); g = pkg.track.bench_graph(t, po.OracleFrontend()); "
            "H = pkg.Graph(device=-2); H.load_bench_graph(g); H.plan_build_host(); d = hashlib.md5(H.plan_export().tobytes()); "
            "R = pkg.Graph(device=-2); R.load_bench_graph(g); R.dist_configure(3, 8); R.plan_build_host(); d.update(R.plan_export().tobytes()); "       # + a rank of 8 (planned by windows)
            "print(d.hexdigest())" % root)
    digests = set()
    for n in ("1", "3", "8"):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, GS_THREADS=n), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-1000:]
        digests.add(r.stdout.strip().splitlines()[-1])
    assert len(digests) == 1, digests


def test_a_rank_plans_its_own_window_and_the_shared_top_only(pkg, bench_graphs):
    """Pose-window shards: the other ranks' subtrees stay single (opaque) supernodes in a rank's plan — vertices and boundary,
    no records, no storage — and the observation edges are laid out for the poses the rank sweeps only.  Fronts, factor
    storage and edge layout of a rank are ~1 / world of the single-GPU plan; the exchange layout (slots of the shared
    fronts, in elimination order) is the same on every rank."""
    _, g = bench_graphs(10000, 2000)
    G = host_graph(pkg, g); one = G.plan_build_host(); P1 = Plan(G.plan_export()); G.close()
    world = 8
    slots = None
    for r in range(world):
        G = host_graph(pkg, g); G.dist_configure(r, world); info = G.plan_build_host(); P = Plan(G.plan_export()); G.close()
        P.check_invariants()
        own = int((P.owner == r).sum()); shared = int((P.owner < 0).sum()); foreign = int((P.owner >= 0).sum()) - own
        assert own <= 1.3 * one.n_fronts / world + 8 and shared <= 3 * world and foreign <= 3 * world        # each other window: one or two opaque supernodes
        assert info.l_doubles <= 1.3 * one.l_doubles / world + 10000 and info.max_front <= 63                  # the opaque ones do not count
        assert P.ell_len <= 1.3 * P1.ell_len / world + 4096                                                   # edges laid out for the swept poses only
        mine = (P.pl_rank == r)
        laid = np.zeros(len(mine), dtype=bool); laid[P.ell_ins[P.ell_ins >= 0]] = True
        assert np.all(laid[mine])                                                                             # every edge the rank evaluates has a slot
        sl = [(int(P.npiv[s]), int(P.nbnd[s]), int(P.x_off[s])) for s in range(P.n_fronts) if P.owner[s] < 0]
        assert slots is None or sl == slots                                                                   # same shared fronts, same slots, on every rank
        slots = sl
    assert len(slots) >= world - 1


@pytest.mark.parametrize("world", [2, 4, 8])
def test_a_rank_builds_the_top_of_the_tree_from_window_masks(pkg, po, world):
    """Round 4: a rank of a sharded graph builds the shared top from two bit masks per landmark (which windows see it) and otherwise
    touches only the edges of its own window and of the windows' first poses (gs_plan.cpp, nd_top) — instead of grouping, listing and
    walking every edge of every window.  Where the general recursion's middle pose is a window's first one (windows of equal size, a power of
    two of them) the two constructions must give the SAME plan, field by field; the per-edge owner arrays differ only in that another
    window's interior edges are marked "not mine" (-1) instead of carrying their owner."""
    t = pkg.track.generate(1602, 320); g = pkg.track.bench_graph(t, po.OracleFrontend())      # 1 600 free poses: divisible by 8
    for rank in range(world):
        P = []
        for by_window in (0, 1):
            H = pkg.Graph(device=-2, debug=dict(shard_by_window=by_window)); H.load_bench_graph(g); H.dist_configure(rank, world); H.plan_build_host()
            P.append(Plan(H.plan_export())); H.close()
        A, B = P
        for k, v in A.__dict__.items():
            w = B.__dict__[k]
            if k in ("pl_rank", "pp_rank"):
                assert np.array_equal(v == rank, w == rank), (k, rank)          # the edges this rank evaluates: the same set
                assert ((w == -1) | (w == v)).all(), (k, rank)                  # every other entry: the owner, or "not mine"
            elif isinstance(v, np.ndarray): assert v.shape == w.shape and np.array_equal(v, w), (k, rank)
            else: assert v == w, (k, rank)
        assert (B.pl_rank == -1).any()


@pytest.mark.parametrize("N,M,h,steps,keep", [(50, 30, 1, 1, None), (1000, 200, 3, 1, None), (1000, 200, 4, 2, None), (1000, 200, 6, 3, 600), (10000, 2000, 4, 2, 4007)])
def test_appended_poses_grow_the_plan_instead_of_rebuilding_it(pkg, po, bench_graphs, N, M, h, steps, keep):
    """Append-only growth (reference src/slam.cpp:433-459, 525-550; gs::grow_plan): the last h poses of the track — with `keep`, of an
    open stretch of it, where they are the first to see some cones — arrive after the plan was built, in `steps` batches.  The plan
    must absorb them (same fronts, same tree, only root-path fronts larger), stay a valid multifrontal plan, and its numeric replay
    must reproduce the oracle's joint solve of the WHOLE graph."""
    _, g = bench_graphs(N, M)
    base, tail, full = split_for_growth(g, h, keep)
    Nf, Mf = len(full["pose_est"]), len(full["lm_est"])
    G = host_graph(pkg, base)
    info0 = G.plan_build_host(); P0 = Plan(G.plan_export())
    assert G.plan_growths() == 0
    per = h // steps; new_lms = 0
    for k in range(steps):
        new_lms += append_tail(G, tail, (k * per, h if k == steps - 1 else (k + 1) * per))
        info = G.plan_build_host()
        assert G.plan_growths() == k + 1, G.growth_refusal()
    assert new_lms == Mf - len(base["lm_est"]) and (keep is None or new_lms > 0)
    P = Plan(G.plan_export()); P.check_invariants()
    assert P.n_fronts == P0.n_fronts and np.array_equal(P.parent, P0.parent) and np.array_equal(P.level, P0.level)
    assert info.n_scalar == info0.n_scalar + 3 * h + 2 * new_lms and P.n_poses == Nf and P.n_lms == Mf
    changed = np.flatnonzero((P.npiv != P0.npiv) | (P.nbnd != P0.nbnd))
    assert 0 < len(changed) <= 16 * P.n_levels and changed[-1] == P.n_fronts - 1            # a few root paths, the root among them
    assert np.array_equal(P.pose_gidx[:Nf - h], P0.pose_gidx) and np.array_equal(P.lm_gidx[:Mf - new_lms], P0.lm_gidx)      # nothing older moved
    blocks, (dp_o, dl_o) = oracle_increment(po, full)
    dp, dl, ok = P.solve(blocks)
    scale = max(np.abs(dp_o).max(), np.abs(dl_o).max())
    tol = 1e-8 if keep is None else 1e-5          # (an open stretch anchored at one end is worse conditioned than the closed lap: 1e-6 at 4 000 poses, for a full build's replay as well)
    assert ok and np.abs(dp - dp_o).max() / scale < tol and np.abs(dl - dl_o).max() / scale < tol
    if keep is not None:                          # ... so there the grown plan is also held against the replay of a FULL build of the same graph
        Ff = host_graph(pkg, full); Ff.plan_build_host(); dpf, dlf, okf = Plan(Ff.plan_export()).solve(blocks); Ff.close()
        assert okf and np.abs(dp - dpf).max() / scale < tol and np.abs(dl - dlf).max() / scale < tol
    # what growth cannot absorb is refused with a reason, and the full build takes over
    G.set_fixed_pose(Nf - 1, True)
    G.plan_build_host()
    assert G.plan_growths() == 0 and "fixed" in G.growth_refusal()
    G.close()


@pytest.mark.parametrize("seed", range(8))
def test_growth_of_irregular_graphs_replays_like_a_full_build(pkg, po, seed):
    """Random graphs (loop-closure style odometry edges, anisotropic information, duplicate edges) grown by two keyframes that bring
    everything a keyframe can: an odometry edge to an arbitrary old pose, one between the two new poses, observations of old cones,
    a cone seen first by the first new pose and again by the second, an observation of a FIXED cone.  The grown plan must stay valid
    and replay to the oracle's joint solve — or refuse with a reason (a small random graph's fronts can be full), never be wrong."""
    rng = np.random.default_rng(100 + seed)
    g = random_graph(seed, n_poses=30 + 3 * seed, n_lms=20 + seed, extra_pp=4, obs_per_pose=3, dup_edges=1)
    N, M = len(g["pose_est"]), len(g["lm_est"])
    G = host_graph(pkg, g); G.plan_build_host(); P0 = Plan(G.plan_export())
    spd = lambda n: (lambda A: (A @ A.T + n * np.eye(n)))(rng.normal(size=(n, n)))
    full = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in g.items()}
    def add(kind, a, b, z, info):
        if kind == "pp":
            G.add_odometry_edge(a, b, z, info)
            full["pp_i"] = np.append(full["pp_i"], np.int32(a)); full["pp_j"] = np.append(full["pp_j"], np.int32(b))
            full["pp_z"] = np.vstack([full["pp_z"], z]); full["pp_info"] = np.vstack([full["pp_info"], info.reshape(1, 9)])
        else:
            G.add_observation_edge(a, b, z, info)
            full["pl_p"] = np.append(full["pl_p"], np.int32(a)); full["pl_l"] = np.append(full["pl_l"], np.int32(b))
            full["pl_z"] = np.vstack([full["pl_z"], z]); full["pl_info"] = np.vstack([full["pl_info"], info.reshape(1, 4)])
    steps = 0; rebuilt = False
    for batch in range(2):
        p = N + batch; est = g["pose_est"][-1] + rng.normal(0.5, 0.2, 3) * [1 + batch, 0.3, 0.05]
        G.add_pose(p, est); full["pose_est"] = np.vstack([full["pose_est"], est])
        if batch == 0:
            G.add_landmark(M, [3.0, 4.0]); full["lm_est"] = np.vstack([full["lm_est"], [3.0, 4.0]])
            add("pp", int(rng.integers(2, N)), p, rng.normal(0, 0.5, 3), spd(3))            # from an arbitrary old pose
        else:
            add("pp", p - 1, p, rng.normal(0, 0.5, 3), spd(3))                               # between the two new poses
            add("pp", p, int(rng.integers(2, N)), rng.normal(0, 0.5, 3), spd(3))             # new pose as the i end, old pose as j
        for l in rng.choice(np.arange(2, M), 2, replace=False):
            add("pl", p, int(l), rng.normal(0, 3, 2), spd(2))
        add("pl", p, M, rng.normal(0, 3, 2), spd(2))                                          # the new cone (first seen in batch 0)
        add("pl", p, int(g["fixed_landmarks"][0]), rng.normal(0, 3, 2), spd(2))             # a fixed cone: feeds the pose's block only
        G.plan_build_host()
        if G.plan_growths() != steps + 1:
            assert G.growth_refusal() in ("a front would exceed 63 scalars", "a front would exceed 159 scalars", "plan outside the matrix-core forms", "forest: more than one root",
                                          "landmark without a partial-sum slot"), G.growth_refusal()     # (the last: an old cone no edge of the base graph observes)
            rebuilt = True; break
        steps += 1
    P = Plan(G.plan_export()); P.check_invariants()
    if steps and not rebuilt:
        assert P.n_fronts == P0.n_fronts and np.array_equal(P.parent, P0.parent)
    blocks, (dp_o, dl_o) = oracle_increment(po, full)
    dp, dl, ok = P.solve(blocks)
    scale = max(np.abs(dp_o).max(), np.abs(dl_o).max())
    assert ok and np.abs(dp - dp_o).max() / scale < 1e-8 and np.abs(dl - dl_o).max() / scale < 1e-8
    G.close()


@pytest.mark.parametrize("N,M,K,h", [(1000, 200, 16, 3), (1000, 200, 24, 2)])
def test_plans_with_workgroup_fronts_grow_up_to_159_scalars(pkg, po, frontend, N, M, K, h):
    """Wide views (16 / 24 cones per frame): their plans hold fronts of 64-159 scalars; appended keyframes are absorbed up to 159 per
    front, and the grown plan replays to the oracle's joint solve."""
    t = pkg.track.generate(N, M, K) if K else pkg.track.generate(N, M)
    g = pkg.track.bench_graph(t, frontend)
    base, tail, full = split_for_growth(g, h)
    G = host_graph(pkg, base); i0 = G.plan_build_host()
    assert i0.max_front > 63
    for k in range(h):
        append_tail(G, tail, (k, k + 1)); i = G.plan_build_host()
        assert G.plan_growths() == k + 1, G.growth_refusal()
    assert 63 < i.max_front <= 159
    P = Plan(G.plan_export()); P.check_invariants()
    blocks, (dp_o, dl_o) = oracle_increment(po, full)
    dp, dl, ok = P.solve(blocks)
    scale = max(np.abs(dp_o).max(), np.abs(dl_o).max())
    assert ok and np.abs(dp - dp_o).max() / scale < 1e-8 and np.abs(dl - dl_o).max() / scale < 1e-8
    G.close()


def test_the_host_worker_pool_survives_a_fork(pkg, po):
    """The structure phase runs on a pool of host threads (csrc/gs_parallel.hpp, round 4) that lives as long as the process.  A forked child has none of
    the parent's threads: it must get a pool of its own on first use (and the parent must go on with its own) instead of waiting for workers that do not
    exist.  subprocess: a fork of the pytest process itself would duplicate its state."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import importlib, os, sys; sys.path.insert(0, %r); from oracle import pyoracle as po; "
            "pkg = importlib.import_module('opendlv-logic-cfsd18-sensation-slam_amd'); "
            "t = pkg.track.generate(3000, 600); g = pkg.track.bench_graph(t, po.OracleFrontend())\n"
            "def plan():\n"
            "    H = pkg.Graph(device=-2); H.load_bench_graph(g); H.plan_build_host(); n = H.stats().n_fronts; H.close(); return n\n"
            "a = plan(); pid = os.fork()\n"
            "if pid == 0:\n"
            "    b = plan(); c = plan(); os._exit(0 if (b == a and c == a) else 3)\n"
            "_, status = os.waitpid(pid, 0); d = plan(); print('ok' if (d == a and os.WEXITSTATUS(status) == 0) else 'bad')" % root)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip().splitlines()[-1] == "ok", (r.stdout[-500:], r.stderr[-1000:])
