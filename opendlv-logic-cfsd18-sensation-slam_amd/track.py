"""Synthetic cone-track source (ctypes over csrc/gs_track.c) and bench-graph construction.

The generator emits what the reference microservice would have latched per keyframe: an odometry
pose (reference src/slam.cpp:173-182,207-209) and a 4 x K cone collector matrix of (azimuth deg,
zenith deg, distance m, type) columns (src/slam.cpp:83-84,108,136), plus ground truth.
`bench_graph` turns one lap into the graph the reference's Slam would have built frame by frame
(addPoseToGraph / addOdometryMeasurement / addConeToGraph / addConeMeasurement,
src/slam.cpp:433-459,525-550) with the generator's ground-truth association, SURVEY §8-B items 1-2
switched off (no duplicated first edge), gauge = first two poses + first two cones (src/slam.cpp:464-474).
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libgstrack.so")

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_lib = None

ODOMETRY_INFORMATION = 5.0     # reference src/slam.cpp:456
CONE_INFORMATION = 0.01        # reference src/slam.cpp:546


def build(force=False):
    src = os.path.join(CSRC, "gs_track.c")
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-std=c11", "-O2", "-fPIC", "-shared", "-o", LIB, src, "-lm"])


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        _lib = C.CDLL(LIB)
        _lib.gs_track_generate.argtypes = [C.c_int32, C.c_int32, _dp, _dp, _dp, _ip, _dp, _ip]
        _lib.gs_track_generate_k.argtypes = [C.c_int32, C.c_int32, C.c_int32, _dp, _dp, _dp, _ip, _dp, _ip]
    return _lib


def generate(n_poses, n_cones, obs_per_pose=None):
    """One lap of the synthetic track.  Returns a dict of numpy arrays:
    truth_poses [N,3], odom_poses [N,3], cone_xy [M,2], cone_type [M], obs [N,K,4], obs_cone [N,K].
    obs_per_pose: K (default 8, SURVEY §8d: the 4 pairs within 20 m); 16 / 24 = the pairs within 40 / 60 m."""
    L = _load()
    K = L.gs_track_obs_per_pose() if obs_per_pose is None else int(obs_per_pose)
    N, M = int(n_poses), int(n_cones)
    t = dict(truth_poses=np.zeros((N, 3)), odom_poses=np.zeros((N, 3)), cone_xy=np.zeros((M, 2)),
             cone_type=np.zeros(M, dtype=np.int32), obs=np.zeros((N, K, 4)),
             obs_cone=np.zeros((N, K), dtype=np.int32))
    a = (t["truth_poses"].ctypes.data_as(_dp), t["odom_poses"].ctypes.data_as(_dp), t["cone_xy"].ctypes.data_as(_dp),
         t["cone_type"].ctypes.data_as(_ip), t["obs"].ctypes.data_as(_dp), t["obs_cone"].ctypes.data_as(_ip))
    rc = L.gs_track_generate(N, M, *a) if obs_per_pose is None else L.gs_track_generate_k(N, M, K, *a)
    if rc != 0:
        raise ValueError("gs_track_generate(%d, %d) rejected its arguments" % (N, M))
    t["K"] = K
    return t


def se2_inverse(a):
    """Vectorised SE2 inverse (g2o convention)."""
    th = normalize_theta(-a[:, 2])
    c, s = np.cos(th), np.sin(th)
    tx, ty = -a[:, 0], -a[:, 1]
    return np.stack([c * tx - s * ty, s * tx + c * ty, th], axis=1)


def se2_compose(a, b):
    c, s = np.cos(a[:, 2]), np.sin(a[:, 2])
    return np.stack([a[:, 0] + (c * b[:, 0] - s * b[:, 1]), a[:, 1] + (s * b[:, 0] + c * b[:, 1]),
                     normalize_theta(a[:, 2] + b[:, 2])], axis=1)


def normalize_theta(th):
    th = np.asarray(th, dtype=np.float64).copy()
    bad = ~((th >= -np.pi) & (th < np.pi))
    if np.any(bad):
        t = th[bad]
        t = t - np.floor(t / (2 * np.pi)) * 2 * np.pi
        t = np.where(t >= np.pi, t - 2 * np.pi, t)
        t = np.where(t < -np.pi, t + 2 * np.pi, t)
        th[bad] = t
    return th


def bench_graph(track, frontend):
    """Arrays of the graph the reference would have built over this lap.

    `frontend` provides the A0 arithmetic: polar_to_xy(az, zen, dist) -> [n,2] and
    cone_to_global(poses, pose_of_obs, obs[n,4]) -> [n,2]  (the product's HIP kernels or the oracle's).
    Landmarks are numbered in map order = order of first observation (m_map.size() at insertion,
    reference src/slam.cpp:556,610)."""
    N = len(track["odom_poses"]); K = track["K"]
    obs = track["obs"].reshape(N * K, 4)
    cone = track["obs_cone"].reshape(N * K)
    pose_of_obs = np.repeat(np.arange(N, dtype=np.int32), K)
    valid = cone >= 0
    obs, cone, pose_of_obs = obs[valid], cone[valid], pose_of_obs[valid]
    # map order: first appearance in frame order
    uniq, first = np.unique(cone, return_index=True)
    order = np.argsort(first, kind="stable")
    map_true_id = uniq[order].astype(np.int32)            # map index -> ground-truth cone id
    first_obs = first[order]
    true_to_map = -np.ones(int(track["cone_type"].shape[0]), dtype=np.int32)
    true_to_map[map_true_id] = np.arange(len(map_true_id), dtype=np.int32)
    lm_of_obs = true_to_map[cone]

    odom = track["odom_poses"]
    # A2 landmark initial estimate = coneToGlobal(pose of first observer, observation)
    lm_est = frontend.cone_to_global(odom, pose_of_obs[first_obs], obs[first_obs])
    # A2 observation measurement = CoG-frame XY of the polar observation
    z_pl = frontend.polar_to_xy(obs[:, 0], obs[:, 1], obs[:, 2])
    # A2 odometry measurement z = est_{k-1}^-1 * pose_k (reference src/slam.cpp:451-454)
    z_pp = se2_compose(se2_inverse(odom[:-1]), odom[1:])
    return dict(
        pose_est=odom.copy(), lm_est=np.ascontiguousarray(lm_est), lm_type=track["cone_type"][map_true_id].copy(),
        map_true_id=map_true_id,
        pp_i=np.arange(0, N - 1, dtype=np.int32), pp_j=np.arange(1, N, dtype=np.int32), pp_z=z_pp,
        pp_info=np.tile((ODOMETRY_INFORMATION * np.eye(3)).reshape(1, 9), (N - 1, 1)),
        pl_p=pose_of_obs.astype(np.int32), pl_l=lm_of_obs.astype(np.int32), pl_z=np.ascontiguousarray(z_pl),
        pl_info=np.tile((CONE_INFORMATION * np.eye(2)).reshape(1, 4), (len(z_pl), 1)),
        fixed_poses=np.array([0, 1], dtype=np.int32), fixed_landmarks=np.array([0, 1], dtype=np.int32),
    )


CONFIGS = {  # BASELINE.json configs (N poses, M cones)
    "cfg1": (50, 30), "cfg2": (1000, 200), "cfg3": (10000, 2000), "cfg4": (100000, 10000),
    "cfg5": (1000000, 50000),
}
