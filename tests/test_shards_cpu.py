"""Pose-window shards on the CPU: the host-side partition of the plan (front ownership, which rank
evaluates which edge, exchange-buffer layout) is exercised by real multi-process runs, torch.distributed with
the gloo backend and world_size 2 / 4: every rank replays ITS part of the plan in numpy (tests/plan_exec.py)
on the oracle's H blocks of ITS edges, the exchange buffer is all-reduced (sum) over gloo exactly where the HIP
path all-reduces it over RCCL, each rank finishes the shared top redundantly, and the merged increment must
equal the oracle's joint-system increment."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, N, M, out_dir, local=False):
    import importlib
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from conftest import make_oracle_graph
    from oracle import pyoracle as po
    from plan_exec import Plan
    pkg = importlib.import_module("opendlv-logic-cfsd18-sensation-slam_amd")
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t = pkg.track.generate(N, M)
    g = pkg.track.bench_graph(t, po.OracleFrontend())
    G = pkg.Graph(device=-2)                          # host-only handle: plan logic, no arithmetic
    if local:
        # rank-local ingestion the way separate processes do it: every rank holds its own window's observation edges (+ the windows' first poses, the
        # fixed poses), computes its OWN bits of the landmark windows from them, the ranks SUM the two uint64 arrays (disjoint bits: the sum is the union)
        M_all = len(g["lm_est"]); zero = np.zeros(M_all, np.uint64)
        keep = G.load_bench_graph_shard(g, rank, world, (zero, zero))
        a, b = G.dist_local_landmark_windows(M_all)
        ta = torch.from_numpy(a.view(np.int64).copy()); tb = torch.from_numpy(b.view(np.int64).copy())
        dist.all_reduce(ta, op=dist.ReduceOp.SUM); dist.all_reduce(tb, op=dist.ReduceOp.SUM)
        G.dist_set_landmark_windows(ta.numpy().view(np.uint64), tb.numpy().view(np.uint64))
        g = dict(g)                                   # from here on "the graph" is what this rank holds
        for k in ("pl_p", "pl_l", "pl_z", "pl_info"):
            g[k] = g[k][keep]
    else:
        G.load_bench_graph(g)
        G.dist_configure(rank, world)
    G.plan_build_host()
    P = Plan(G.plan_export()); P.check_invariants()
    assert P.world == world and P.rank == rank and P.n_shared >= 1
    # H blocks of the edges THIS rank evaluates: an oracle graph holding only those edges
    sub = dict(g)
    kp = P.pp_rank == rank; kl = P.pl_rank == rank
    for k in ("pp_i", "pp_j", "pp_z", "pp_info"):
        sub[k] = g[k][kp]
    for k in ("pl_p", "pl_l", "pl_z", "pl_info"):
        sub[k] = g[k][kl]
    blk_sub = make_oracle_graph(po, sub).linearize_blocks()
    blocks = dict(blk_sub)                           # scatter the per-edge blocks back to global edge numbering
    blocks["Hpp_off"] = np.zeros((len(g["pp_i"]), 9)); blocks["Hpp_off"][kp] = blk_sub["Hpp_off"]
    blocks["Hpl"] = np.zeros((len(g["pl_p"]), 6)); blocks["Hpl"][kl] = blk_sub["Hpl"]
    X, ok1 = P.shard_local(blocks)
    xt = torch.from_numpy(X)
    dist.all_reduce(xt, op=dist.ReduceOp.SUM)        # <- the one exchange step of an iteration
    dp, dl, ok2 = P.shard_finish(xt.numpy())
    # merge: every vertex is "primary" on exactly one rank
    pk, lk, pprim, lprim = G.dist_known()
    assert np.array_equal(pk, P.pose_known) and np.array_equal(lk, P.lm_known)
    mp = torch.from_numpy(dp * pprim[:, None]); ml = torch.from_numpy(dl * lprim[:, None])
    cnt = torch.from_numpy(np.concatenate([pprim, lprim]).astype(np.float64))
    dist.all_reduce(mp); dist.all_reduce(ml); dist.all_reduce(cnt)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), dp=mp.numpy(), dl=ml.numpy(), cnt=cnt.numpy(), ok=ok1 and ok2,
             n_shared=P.n_shared, exchange=P.exchange_doubles, owned=int((P.owner == rank).sum()),
             my_pl=int(kl.sum()), my_pp=int(kp.sum()))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,N,M,local", [(2, 50, 30, False), (2, 1000, 200, False), (4, 1000, 200, False), (4, 1000, 200, True), (2, 1000, 200, True)])
def test_sharded_increment_over_gloo_equals_joint_solve(po, bench_graphs, tmp_path, world, N, M, local):
    """(local: rank-local ingestion — the ranks hold their own windows' observation edges only and agree on the landmark windows by an all-reduce)"""
    import torch.multiprocessing as mp
    port = _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker, args=(r, world, port, N, M, str(tmp_path), local)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    from conftest import make_oracle_graph
    _, g = bench_graphs(N, M)
    og = make_oracle_graph(po, g); og.build_system(); og.apply_update(og.solve_ldlt(1)); dp_o, dl_o = og.delta()
    scale = max(np.abs(dp_o).max(), np.abs(dl_o).max())
    outs = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]
    for o in outs:
        assert bool(o["ok"])
        assert np.array_equal(o["cnt"], np.ones_like(o["cnt"]))            # every vertex primary exactly once
        assert np.abs(o["dp"] - dp_o).max() / scale < 1e-8 and np.abs(o["dl"] - dl_o).max() / scale < 1e-8
    # the edges are partitioned, the work is spread, the exchange is small
    assert sum(int(o["my_pl"]) for o in outs) == len(g["pl_p"]) and sum(int(o["my_pp"]) for o in outs) == len(g["pp_i"])      # (rank-local ingestion too: every edge of the whole graph is evaluated by exactly one rank)
    assert all(int(o["owned"]) > 0 for o in outs)
    assert int(outs[0]["exchange"]) * 8 < 2_000_000


@pytest.mark.parametrize("world", [3, 5, 7, 8])
@pytest.mark.parametrize("by_window", [1, 0])
def test_odd_worlds_and_fixed_poses_in_one_process(pkg, po, bench_graphs, world, by_window):
    """The per-rank plans of worlds that are no power of two, on a lap with fixed poses sprinkled through it (windows count FREE poses): every
    rank's part replayed in numpy, the exchange buffers summed as the all-reduce would, the shared top finished on every rank, the merged
    increment against the oracle's joint solve — for the plan built by windows (round 4: per-landmark window masks, a rank's lists from its own
    window's edges) and for the general recursion.  Every edge has exactly one evaluator, every vertex one primary rank."""
    from conftest import make_oracle_graph
    from plan_exec import Plan
    _, g0 = bench_graphs(1000, 200)
    g = dict(g0); g["fixed_poses"] = np.array(sorted(set([0, 1] + list(range(90, 1000, 97)))), dtype=np.int32)
    og = make_oracle_graph(po, g); og.build_system(); og.apply_update(og.solve_ldlt(1)); dp_o, dl_o = og.delta()
    scale = max(np.abs(dp_o).max(), np.abs(dl_o).max())
    plans, locals_, prim = [], [], []
    n_pl, n_pp = len(g["pl_p"]), len(g["pp_i"])
    seen_pl, seen_pp = np.zeros(n_pl, int), np.zeros(n_pp, int)
    for rank in range(world):
        G = pkg.Graph(device=-2, debug=dict(shard_by_window=by_window)); G.load_bench_graph(g); G.dist_configure(rank, world); G.plan_build_host()
        P = Plan(G.plan_export()); P.check_invariants()
        kp = P.pp_rank == rank; kl = P.pl_rank == rank
        seen_pl += kl; seen_pp += kp
        sub = dict(g)
        for k in ("pp_i", "pp_j", "pp_z", "pp_info"): sub[k] = g[k][kp]
        for k in ("pl_p", "pl_l", "pl_z", "pl_info"): sub[k] = g[k][kl]
        blk_sub = make_oracle_graph(po, sub).linearize_blocks()
        blocks = dict(blk_sub)
        blocks["Hpp_off"] = np.zeros((n_pp, 9)); blocks["Hpp_off"][kp] = blk_sub["Hpp_off"]
        blocks["Hpl"] = np.zeros((n_pl, 6)); blocks["Hpl"][kl] = blk_sub["Hpl"]
        X, ok = P.shard_local(blocks); assert ok
        plans.append(P); locals_.append(X); prim.append(G.dist_known()); G.close()
    assert (seen_pl == 1).all() and (seen_pp == 1).all()          # every edge: exactly one evaluator
    assert len({P.exchange_doubles for P in plans}) == 1 and len({P.n_shared for P in plans}) == 1
    Xsum = np.sum(locals_, axis=0)                                 # the all-reduce
    dp = np.zeros_like(dp_o); dl = np.zeros_like(dl_o); cnt_p = np.zeros(len(dp_o)); cnt_l = np.zeros(len(dl_o))
    for P, (pk, lk, pprim, lprim) in zip(plans, prim):
        a, b, ok = P.shard_finish(Xsum.copy()); assert ok
        dp += a * pprim[:, None]; dl += b * lprim[:, None]; cnt_p += pprim; cnt_l += lprim
    free_p = np.ones(len(dp_o), bool); free_p[g["fixed_poses"]] = False
    free_l = np.ones(len(dl_o), bool); free_l[g["fixed_landmarks"]] = False
    assert (cnt_p[free_p] == 1).all() and (cnt_l[free_l] == 1).all()
    assert np.abs(dp - dp_o).max() / scale < 1e-8 and np.abs(dl - dl_o).max() / scale < 1e-8


@pytest.mark.parametrize("world", [2, 5, 8])
def test_rank_local_ingestion_gives_the_plan_of_the_whole_graph(pkg, po, bench_graphs, world):
    """gs_dist_set_landmark_windows (round 4): a rank that is told which windows see which landmark holds the observation edges of its own window,
    of the windows' first poses and of the fixed poses only — a fraction 1 / world + a few of them.  Its plan must be the plan it builds from the
    WHOLE graph: same fronts (pivots, boundaries, parents, levels, owners), same boundary rows, same exchange slots, same scalar numbering; and the
    numbers must come out: every rank's part replayed from ITS edges only, exchange buffers summed, the merged increment against the oracle's
    joint solve.  The masks: numpy over the whole graph here, and (checked equal) the OR of the ranks' own bits from gs_dist_local_landmark_windows."""
    from conftest import make_oracle_graph
    from plan_exec import Plan
    _, g0 = bench_graphs(1000, 200)
    g = dict(g0); g["fixed_poses"] = np.array(sorted(set([0, 1] + list(range(90, 1000, 97)))), dtype=np.int32)
    og = make_oracle_graph(po, g); og.build_system(); og.apply_update(og.solve_ldlt(1)); dp_o, dl_o = og.delta()
    scale = max(np.abs(dp_o).max(), np.abs(dl_o).max())
    masks = pkg.binding.landmark_windows(g, world)
    n_pl, n_pp, M = len(g["pl_p"]), len(g["pp_i"]), len(g["lm_est"])
    seen_pl, seen_pp = np.zeros(n_pl, int), np.zeros(n_pp, int)
    own_a, own_b = np.zeros(M, np.uint64), np.zeros(M, np.uint64)
    plans, locals_, prim, kept = [], [], [], []
    for rank in range(world):
        F = pkg.Graph(device=-2); F.load_bench_graph(g); F.dist_configure(rank, world); F.plan_build_host(); PF = Plan(F.plan_export()); F.close()
        G = pkg.Graph(device=-2); keep = G.load_bench_graph_shard(g, rank, world, masks); G.plan_build_host()
        P = Plan(G.plan_export()); P.check_invariants()
        a, b = G.dist_local_landmark_windows(M); own_a |= a; own_b |= b
        kept.append(keep.mean())
        for name in ("npiv", "nbnd", "parent", "level", "owner", "piv0", "bnd_rows", "pose_gidx", "lm_gidx", "x_off", "level_start", "pp_rank"):
            assert np.array_equal(getattr(P, name), getattr(PF, name)), (rank, name)
        assert P.exchange_doubles == PF.exchange_doubles and P.n_shared == PF.n_shared and P.n_scalar == PF.n_scalar
        idx = np.flatnonzero(keep)                                   # the handle's observation edge k is g's edge idx[k]
        assert np.array_equal(P.pl_rank == rank, (PF.pl_rank == rank)[idx]) and not (PF.pl_rank == rank)[~keep].any()      # everything it evaluates is among the edges it holds
        kp = P.pp_rank == rank; kl = np.zeros(n_pl, bool); kl[idx[P.pl_rank == rank]] = True
        seen_pl += kl; seen_pp += kp
        sub = dict(g)
        for k in ("pp_i", "pp_j", "pp_z", "pp_info"): sub[k] = g[k][kp]
        for k in ("pl_p", "pl_l", "pl_z", "pl_info"): sub[k] = g[k][kl]
        blk_sub = make_oracle_graph(po, sub).linearize_blocks()
        blocks = dict(blk_sub)
        blocks["Hpp_off"] = np.zeros((n_pp, 9)); blocks["Hpp_off"][kp] = blk_sub["Hpp_off"]
        blocks["Hpl"] = np.zeros((len(idx), 6)); blocks["Hpl"][(P.pl_rank == rank)] = blk_sub["Hpl"]     # indexed by the handle's OWN edge numbers
        X, ok = P.shard_local(blocks); assert ok
        plans.append(P); locals_.append(X); prim.append(G.dist_known()); G.close()
    assert np.array_equal(own_a, masks[0]) and np.array_equal(own_b, masks[1])       # the ranks' own bits add up to the whole graph's masks
    assert (seen_pl == 1).all() and (seen_pp == 1).all() and max(kept) < 1.0 / world + 0.08
    Xsum = np.sum(locals_, axis=0)
    dp = np.zeros_like(dp_o); dl = np.zeros_like(dl_o)
    for P, (pk, lk, pprim, lprim) in zip(plans, prim):
        a, b, ok = P.shard_finish(Xsum.copy()); assert ok
        dp += a * pprim[:, None]; dl += b * lprim[:, None]
    assert np.abs(dp - dp_o).max() / scale < 1e-8 and np.abs(dl - dl_o).max() / scale < 1e-8
    # masks that leave an edge out are refused, and so is a graph that cannot be planned by windows
    G = pkg.Graph(device=-2); G.load_bench_graph_shard(g, 0, world, (np.zeros(M, np.uint64), np.zeros(M, np.uint64)))
    with pytest.raises(pkg.GsError, match="missing from the landmark windows"):
        G.plan_build_host()
    G.close()


@pytest.mark.parametrize("seed", [31, 33, 36, 37, 5, 12])
def test_windows_that_fall_apart_still_give_every_rank_the_same_shared_top(pkg, po, seed):
    """Irregular graphs (random observations: every cone is seen from everywhere and sits in a separator) with poses fixed in the middle of the
    chain: a rank's window then falls into several subtrees with smaller boundaries, hanging under different shared fronts — while the other
    ranks see the window as ONE opaque supernode.  The owner hands the shared top the union boundary the others compute (round 4; before, ranks
    disagreed on the rows of shared fronts — seeds 31 / 33 / 36 / 37 here — and the exchange buffers did not even have the same length).
    Worlds 2-8, by windows and by the general recursion: same exchange layout on every rank, every edge one evaluator, the merged increment
    against the oracle's joint solve."""
    from conftest import make_oracle_graph, random_graph
    from plan_exec import Plan
    rng = np.random.default_rng(1000 + seed)
    n_poses = int(rng.integers(60, 400)); n_lms = int(rng.integers(10, 80)); opp = int(rng.integers(1, 4))
    g = random_graph(seed, n_poses=n_poses, n_lms=n_lms, extra_pp=0, obs_per_pose=opp, dup_edges=0)
    g["fixed_poses"] = np.array(sorted(set([0] + list(rng.choice(n_poses, int(rng.integers(0, 4)), replace=False)))), dtype=np.int32)
    og = make_oracle_graph(po, g); og.build_system(); og.apply_update(og.solve_ldlt(1)); dp_o, dl_o = og.delta()
    scale = max(np.abs(dp_o).max(), np.abs(dl_o).max())
    n_pl, n_pp = len(g["pl_p"]), len(g["pp_i"])
    for world in (2, 3, 4, 6, 8):
        if n_poses < 6 * world: continue
        for by_window in (1, 0):
            plans, locals_, prim = [], [], []
            seen_pl, seen_pp = np.zeros(n_pl, int), np.zeros(n_pp, int)
            for rank in range(world):
                G = pkg.Graph(device=-2, debug=dict(shard_by_window=by_window)); G.load_bench_graph(g); G.dist_configure(rank, world); G.plan_build_host()
                P = Plan(G.plan_export()); P.check_invariants()
                kp = P.pp_rank == rank; kl = P.pl_rank == rank; seen_pl += kl; seen_pp += kp
                sub = dict(g)
                for k in ("pp_i", "pp_j", "pp_z", "pp_info"): sub[k] = g[k][kp]
                for k in ("pl_p", "pl_l", "pl_z", "pl_info"): sub[k] = g[k][kl]
                bs = make_oracle_graph(po, sub).linearize_blocks(); blocks = dict(bs)
                blocks["Hpp_off"] = np.zeros((n_pp, 9)); blocks["Hpp_off"][kp] = bs["Hpp_off"]
                blocks["Hpl"] = np.zeros((n_pl, 6)); blocks["Hpl"][kl] = bs["Hpl"]
                X, ok = P.shard_local(blocks); assert ok
                plans.append(P); locals_.append(X); prim.append(G.dist_known()); G.close()
            assert (seen_pl == 1).all() and (seen_pp == 1).all(), (world, by_window)
            assert len({len(x) for x in locals_}) == 1 and len({P.exchange_doubles for P in plans}) == 1, (world, by_window, [P.exchange_doubles for P in plans])
            Xsum = np.sum(locals_, axis=0)
            dp = np.zeros_like(dp_o); dl = np.zeros_like(dl_o)
            for P, (pk, lk, pprim, lprim) in zip(plans, prim):
                a, b, ok = P.shard_finish(Xsum.copy()); assert ok
                dp += a * pprim[:, None]; dl += b * lprim[:, None]
            assert max(np.abs(dp - dp_o).max(), np.abs(dl - dl_o).max()) / scale < 1e-7, (world, by_window)
