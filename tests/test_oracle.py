"""CPU tests of the oracle itself (no GPU).  The reference holds no golden vectors for this path
(parity unpinned for the g2o arithmetic, see oracle/graphslam_oracle.h), so the oracle is pinned by:
finite-difference Jacobians, a hand-computed micro-graph, independent numpy geometry for A0, dense
numpy solves, and — for the linear solve A8 — the reference's own vendored Eigen 3.3.4 built from
/root/reference/thirdparty (oracle/_ref)."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import make_oracle_graph, random_graph

_dp = C.POINTER(C.c_double)


def d(a):
    return a.ctypes.data_as(_dp)


def edge_se2(L, xi, xj, z, jac=True):
    xi, xj, z = [np.ascontiguousarray(v, dtype=np.float64) for v in (xi, xj, z)]
    e = np.zeros(3); A = np.zeros(9); B = np.zeros(9)
    L.orc_edge_se2(d(xi), d(xj), d(z), d(e), d(A) if jac else None, d(B) if jac else None)
    return e, A.reshape(3, 3), B.reshape(3, 3)


def edge_pl(L, xp, l, z, jac=True):
    xp, l, z = [np.ascontiguousarray(v, dtype=np.float64) for v in (xp, l, z)]
    e = np.zeros(2); A = np.zeros(6); B = np.zeros(4)
    L.orc_edge_se2_pointxy(d(xp), d(l), d(z), d(e), d(A) if jac else None, d(B) if jac else None)
    return e, A.reshape(2, 3), B.reshape(2, 2)


def test_normalize_theta(po):
    L = po.lib()
    for th, want in [(0.0, 0.0), (np.pi, -np.pi), (-np.pi, -np.pi), (3 * np.pi + 0.1, -np.pi + 0.1),
                     (-7.0, -7.0 + 2 * np.pi), (100.0, 100.0 - 16 * 2 * np.pi)]:
        assert abs(L.orc_normalize_theta(th) - want) < 1e-12


def test_se2_group_laws(po):
    L = po.lib(); rng = np.random.default_rng(0)
    for _ in range(20):
        a = rng.normal(size=3) * [5, 5, 2]
        inv = np.zeros(3); out = np.zeros(3)
        L.orc_se2_inverse(d(a), d(inv)); L.orc_se2_compose(d(a), d(inv), d(out))
        assert np.abs(out).max() < 1e-12


def test_edge_se2_jacobians_finite_difference(po):
    """analytic A, B of EdgeSE2 vs central differences (h = 1e-6)."""
    L = po.lib(); rng = np.random.default_rng(1); h = 1e-6
    for _ in range(10):
        xi = rng.normal(size=3) * [3, 3, 1]; xj = xi + rng.normal(size=3) * [1, 1, 0.3]; z = rng.normal(size=3) * [1, 1, 0.3]
        _, A, B = edge_se2(L, xi, xj, z)
        for k in range(3):
            dv = np.zeros(3); dv[k] = h
            fa = (edge_se2(L, xi + dv, xj, z, False)[0] - edge_se2(L, xi - dv, xj, z, False)[0]) / (2 * h)
            fb = (edge_se2(L, xi, xj + dv, z, False)[0] - edge_se2(L, xi, xj - dv, z, False)[0]) / (2 * h)
            assert np.abs(fa - A[:, k]).max() < 1e-8 and np.abs(fb - B[:, k]).max() < 1e-8


def test_edge_pointxy_jacobians_finite_difference(po):
    """analytic A (2x3), B (2x2) of EdgeSE2PointXY vs central differences."""
    L = po.lib(); rng = np.random.default_rng(2); h = 1e-6
    for _ in range(10):
        xp = rng.normal(size=3) * [3, 3, 1]; l = rng.normal(size=2) * 5; z = rng.normal(size=2) * 5
        _, A, B = edge_pl(L, xp, l, z)
        for k in range(3):
            dv = np.zeros(3); dv[k] = h
            fa = (edge_pl(L, xp + dv, l, z, False)[0] - edge_pl(L, xp - dv, l, z, False)[0]) / (2 * h)
            assert np.abs(fa - A[:, k]).max() < 1e-8
        for k in range(2):
            dv = np.zeros(2); dv[k] = h
            fb = (edge_pl(L, xp, l + dv, z, False)[0] - edge_pl(L, xp, l - dv, z, False)[0]) / (2 * h)
            assert np.abs(fb - B[:, k]).max() < 1e-8


def test_odometry_measurement_gives_zero_error_like_reference(po):
    """reference src/slam.cpp:451-456: z = prev^-1 * current => e == 0 on a fresh odometry edge."""
    L = po.lib(); rng = np.random.default_rng(3)
    a = rng.normal(size=3); b = rng.normal(size=3)
    ai = np.zeros(3); z = np.zeros(3)
    L.orc_se2_inverse(d(a), d(ai)); L.orc_se2_compose(d(ai), d(b), d(z))
    e, _, _ = edge_se2(L, a, b, z)
    assert np.abs(e).max() < 1e-14


def test_closed_form_micro_graph(po):
    """Hand-computed: fixed pose at the origin sees a free cone at (2,1) as (2,1.2), Omega = I  =>
    e = (0,-0.2), B = I, H_ll = I, b = (0,0.2), dl = (0,0.2); a free second pose with a consistent odometry
    edge has b = 0 and does not move."""
    og = po.OracleGraph()
    og.add_poses([[0, 0, 0], [1, 0, 0]]); og.add_landmarks([[2.0, 1.0]])
    og.add_odometry_edges([0], [1], [[1, 0, 0]], np.eye(3).reshape(1, 9))
    og.add_observation_edges([0], [0], [[2.0, 1.2]], np.eye(2).reshape(1, 4))
    og.set_fixed_pose(0)
    blk = og.linearize_blocks()
    assert np.allclose(blk["Hll_diag"][0], [1, 0, 0, 1]) and np.allclose(blk["b_lm"][0], [0, 0.2])
    assert np.allclose(blk["b_pose"][1], 0) and np.allclose(blk["Hpp_diag"][0], 0)
    assert abs(og.chi2() - 0.04) < 1e-15
    done, chi, _ = og.optimize(1, ordering=0)
    assert done == 1 and np.allclose(og.landmarks()[0], [2.0, 1.2]) and np.allclose(og.poses()[1], [1, 0, 0])


def test_polar_to_xy_against_independent_geometry(po, frontend):
    """A0: LiDAR 1.5 m ahead of the CoG; for cones ahead of the CoG the reference's law-of-cosines
    construction must equal plain vector geometry (up to its float PI literal, ~9e-8 rad)."""
    rng = np.random.default_rng(4)
    az = rng.uniform(-120, 120, 500); az[az == 0] = 3.0; dist = rng.uniform(2.0, 40, 500)
    lx = dist * np.cos(np.radians(az)) + 1.5; ly = dist * np.sin(np.radians(az))
    keep = lx > 0.2
    got = frontend.polar_to_xy(az, np.zeros_like(az), dist)
    assert np.abs(got[keep, 0] - lx[keep]).max() < 1e-5 and np.abs(got[keep, 1] - ly[keep]).max() < 1e-5
    assert np.isnan(frontend.polar_to_xy([0.0], [0.0], [5.0])).all()


def test_association_first_match_in_map_order(frontend):
    poses = np.zeros((1, 3)); obs = np.array([[5.0, 0.0, 6.0, 1.0]])
    gxy = frontend.cone_to_global(poses, [0], obs)[0]
    map_xy = np.array([gxy + [5.0, 0], gxy + [0.9, 0.0], gxy + [0.1, 0.0], gxy + [0.0, 0.05]])
    assert frontend.associate(poses, [0], obs, map_xy, np.array([1, 1, 1, 1], np.int32), 1.2)[0] == 1
    assert frontend.associate(poses, [0], obs, map_xy, np.array([1, 2, 2, 1], np.int32), 1.2)[0] == 3     # colour gate
    assert frontend.associate(poses, [0], obs, map_xy, np.array([2, 2, 2, 2], np.int32), 1.2)[0] == -1


def dense_from_ccs(n, colptr, rowind, values):
    H = np.zeros((n, n))
    for c in range(n):
        for p in range(colptr[c], colptr[c + 1]):
            H[rowind[p], c] = values[p]
    return H + np.triu(H, 1).T


@pytest.mark.parametrize("seed", [5, 6])
def test_ccs_assembly_equals_block_assembly(po, seed):
    """Two independent code paths of the oracle (per-block export vs scalar CCS) agree entry by entry."""
    g = random_graph(seed)
    og = make_oracle_graph(po, g)
    blk = og.linearize_blocks(); n, colptr, rowind, values, b = og.build_system()
    H = dense_from_ccs(n, colptr, rowind, values)
    # rebuild from blocks with the oracle's index map: free landmarks first, then free poses
    M, N = len(g["lm_est"]), len(g["pose_est"])
    off = {}; o = 0
    for l in range(M):
        if l not in g["fixed_landmarks"]:
            off[("l", l)] = o; o += 2
    for p in range(N):
        if p not in g["fixed_poses"]:
            off[("p", p)] = o; o += 3
    H2 = np.zeros((n, n)); b2 = np.zeros(n)
    for (kind, idx), o in off.items():
        dim = 2 if kind == "l" else 3
        src = blk["Hll_diag"][idx] if kind == "l" else blk["Hpp_diag"][idx]
        H2[o:o + dim, o:o + dim] += src.reshape(dim, dim)
        b2[o:o + dim] = (blk["b_lm"] if kind == "l" else blk["b_pose"])[idx]
    for k, (i, j) in enumerate(zip(g["pp_i"], g["pp_j"])):
        if ("p", i) in off and ("p", j) in off:
            B = blk["Hpp_off"][k].reshape(3, 3); oi, oj = off[("p", i)], off[("p", j)]
            H2[oi:oi + 3, oj:oj + 3] += B; H2[oj:oj + 3, oi:oi + 3] += B.T
    for k, (p, l) in enumerate(zip(g["pl_p"], g["pl_l"])):
        if ("p", p) in off and ("l", l) in off:
            B = blk["Hpl"][k].reshape(3, 2); op, ol = off[("p", p)], off[("l", l)]
            H2[op:op + 3, ol:ol + 2] += B; H2[ol:ol + 2, op:op + 3] += B.T
    assert np.abs(H - H2).max() <= 1e-12 * np.abs(H).max() and np.abs(b - b2).max() <= 1e-12 * np.abs(b).max()


@pytest.mark.parametrize("N,M", [(50, 30), (1000, 200)])
def test_ldlt_orderings_dense_and_reference_eigen_agree(po, bench_graphs, N, M):
    """A8: natural vs track-interleave ordering vs numpy dense vs the reference's vendored Eigen (AMD):
    same x to rounding, normal-equation residual ~ machine precision."""
    _, g = bench_graphs(N, M)
    og = make_oracle_graph(po, g)
    n, colptr, rowind, values, b = og.build_system()
    x0 = og.solve_ldlt(0); x1 = og.solve_ldlt(1)
    H = dense_from_ccs(n, colptr, rowind, values)
    xd = np.linalg.solve(H, b)
    scale = np.abs(xd).max()
    assert np.abs(x0 - x1).max() / scale < 1e-9 and np.abs(x1 - xd).max() / scale < 1e-8
    assert np.abs(H @ x1 - b).max() / np.abs(b).max() < 1e-11
    if po.ref_eigen() is not None:
        for kind in (0, 1):                                   # SimplicialLDLT and SimplicialLLT
            xe = po.EigenSolver(kind).solve(n, colptr, rowind, values, b)
            assert np.abs(xe - x1).max() / scale < 1e-9
            assert np.abs(H @ xe - b).max() / np.abs(b).max() < 1e-11


def test_reference_eigen_is_the_vendored_3_3_4(po):
    R = po.ref_eigen()
    if R is None:
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    assert R.ref_eigen_version() == b"Eigen 3.3.4"           # reference thirdparty/Eigen/src/Core/util/Macros.h:14-16
    out = np.zeros(4); R.ref_eigen_rotation2d(0.3, d(out))
    assert np.allclose(out, [np.cos(0.3), -np.sin(0.3), np.sin(0.3), np.cos(0.3)], atol=1e-16)


def test_optimize_with_reference_eigen_matches_own_ldlt(po, bench_graphs):
    if po.ref_eigen() is None:
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    _, g = bench_graphs(1000, 200)
    a = make_oracle_graph(po, g); b = make_oracle_graph(po, g)
    da, chia, _ = a.optimize(10, ordering=1)
    db, chib, _ = b.optimize(10, solver=po.EigenSolver(0))
    assert da == db == 10
    assert np.abs(a.poses() - b.poses()).max() < 1e-9 and np.abs(a.landmarks() - b.landmarks()).max() < 1e-9
    assert np.allclose(chia, chib, rtol=1e-9)


@pytest.mark.parametrize("N,M", [(50, 30), (1000, 200)])
def test_gauss_newton_converges_on_the_synthetic_loop(po, bench_graphs, N, M):
    _, g = bench_graphs(N, M)
    og = make_oracle_graph(po, g)
    done, chi, _ = og.optimize(10, ordering=1)
    assert done == 10 and chi[-1] < chi[0] and abs(chi[-1] - chi[-2]) < 1e-6 * chi[-1]
    dp, dl = og.delta()
    assert np.abs(dp).max() < 1e-6 and np.abs(dl).max() < 1e-6
    assert np.array_equal(og.poses()[:2], g["pose_est"][:2]) and np.array_equal(og.landmarks()[:2], g["lm_est"][:2])


def test_golden_fixture_config1(po, pkg):
    """tests/golden/cfg1_oracle.npz (made by tests/golden/make_golden.py from THIS oracle — a regression pin,
    not reference output: the reference has none): inputs + H blocks, b, dx after 1 iteration, chi2 history and
    estimates after the reference's 10 iterations."""
    path = os.path.join(os.path.dirname(__file__), "golden", "cfg1_oracle.npz")
    z = np.load(path, allow_pickle=False)
    g = {k[3:]: z[k] for k in z.files if k.startswith("in_")}
    og = make_oracle_graph(po, g)
    blk = og.linearize_blocks()
    for k in ("Hpp_diag", "Hll_diag", "Hpp_off", "Hpl", "b_pose", "b_lm"):
        assert np.abs(blk[k] - z["out_" + k]).max() <= 1e-13 * max(np.abs(z["out_" + k]).max(), 1), k
    og.build_system(); og.apply_update(og.solve_ldlt(1)); dp, dl = og.delta()
    assert np.abs(dp - z["out_dpose_it0"]).max() < 1e-12 and np.abs(dl - z["out_dlm_it0"]).max() < 1e-12
    og2 = make_oracle_graph(po, g)
    done, chi, _ = og2.optimize(10, ordering=1)
    assert np.allclose(chi, z["out_chi2"], rtol=1e-10)
    assert np.abs(og2.poses() - z["out_poses_it10"]).max() < 1e-11 and np.abs(og2.landmarks() - z["out_lms_it10"]).max() < 1e-11
    # the generator is deterministic: the committed inputs are what the track source produces today
    t = pkg.track.generate(50, 30)
    assert np.array_equal(t["obs"], z["track_obs"]) and np.array_equal(t["odom_poses"], z["track_odom"])


def test_optimize_until_stop_rule_and_failure_semantics(po, pkg, frontend):
    """The build-defined stop rule (the reference has none) and g2o's failure rule in the oracle."""
    t = pkg.track.generate(50, 30); g = pkg.track.bench_graph(t, frontend)
    og = make_oracle_graph(po, g)
    done, chi, failed = og.optimize_until(40, 1e-9)
    assert not failed and 2 <= done < 40
    assert abs(chi[done - 2] - chi[done - 1]) <= 1e-9 * chi[done - 1]           # the rule fired at the last iteration ...
    assert all(abs(chi[k - 1] - chi[k]) > 1e-9 * chi[k] for k in range(1, done - 1))   # ... and not before
    og2 = make_oracle_graph(po, g); d2, _, _ = og2.optimize_until(done, -1.0)   # no rule: same iterate after the same count
    assert d2 == done and np.array_equal(og.poses(), og2.poses())
    # singular system: the first solve fails, no update is applied (g2o returns 0, vertices untouched)
    s = po.OracleGraph(); P0 = np.array([[0.0, 0, 0], [1.1, 0.2, 0.1], [2.3, -0.1, 0.2]]); s.add_poses(P0)
    info = np.tile(np.diag([1.0, 1.0, 0.0]).reshape(1, 9), (2, 1))
    s.add_odometry_edges([0, 1], [1, 2], np.array([[1.0, 0, 0], [1.0, 0, 0]]), info)
    done, _, failed = s.optimize_until(3, -1.0, ordering=0)
    assert done == 0 and failed and np.array_equal(s.poses(), P0)


def test_caller_supplied_elimination_order_is_another_exact_order(pkg, po, bench_graphs):
    """ordering 2 = the oracle's LDL^T in the elimination order of ANOTHER solver — here the nested-dissection order of the
    product's host-side plan (no GPU involved).  Any exact order gives the same increment up to rounding; a sequence that is
    not a permutation is refused.  (scripts/parity_spread.py uses this to separate "other order" from "other arithmetic" when
    the GPU increment is compared with the CPU increments.)"""
    from plan_exec import Plan
    _, g = bench_graphs(1000, 200)
    H = pkg.Graph(device=-2); H.load_bench_graph(g); H.plan_build_host(); P = Plan(H.plan_export()); H.close()
    og = make_oracle_graph(po, g); og.build_system(); x1 = og.solve_ldlt(1)
    og.set_elimination_order_like(P.pose_gidx, P.lm_gidx); x2 = og.solve_ldlt(2)
    assert np.abs(x1 - x2).max() <= 1e-8 * np.abs(x1).max()
    assert not np.array_equal(x1, x2)                          # ... and it really is a different order (different rounding)
    bad = np.zeros(len(x1), dtype=np.int32)
    assert po.lib().orc_set_elimination_order(og.g, bad.ctypes.data_as(po._ip), len(bad)) == -2
