"""ctypes binding of the CPU oracle (TEST INFRASTRUCTURE — see oracle/graphslam_oracle.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "liboracle.so")
REF = os.path.join(HERE, "_ref", "libref_eigen.so")

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
SOLVER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, _ip, _ip, _dp, _dp, _dp)


def build(force=False):
    """Compile liboracle.so (and oracle/_ref when /root/reference is present)."""
    if force or not os.path.exists(LIB) or \
            os.path.getmtime(LIB) < os.path.getmtime(os.path.join(HERE, "graphslam_oracle.c")):
        subprocess.check_call(["make", "-C", HERE, os.path.join(HERE, "liboracle.so")])
    subprocess.check_call(["make", "-C", HERE, "ref"], stdout=subprocess.DEVNULL)


def _d(a):
    return a.ctypes.data_as(_dp)


def _i(a):
    return a.ctypes.data_as(_ip)


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        L = C.CDLL(LIB)
        L.orc_create.restype = C.c_void_p
        L.orc_chi2.restype = C.c_double
        L.orc_normalize_theta.restype = C.c_double
        L.orc_normalize_theta.argtypes = [C.c_double]
        L.orc_system_colptr.restype = _ip
        L.orc_system_rowind.restype = _ip
        L.orc_system_values.restype = _dp
        L.orc_system_b.restype = _dp
        L.orc_transform_cone_to_cog.argtypes = [C.c_double, C.c_double, C.c_double, _dp]
        L.orc_spherical_to_cartesian.argtypes = [C.c_double, C.c_double, C.c_double, C.c_double, _dp]
        L.orc_cone_to_global.argtypes = [_dp, _dp, C.c_double, _dp]
        L.orc_polar_to_xy_batch.argtypes = [C.c_int, _dp, _dp, _dp, C.c_double, _dp]
        L.orc_cone_to_global_batch.argtypes = [C.c_int, _dp, _ip, _dp, C.c_double, _dp]
        L.orc_associate_fixed_map.argtypes =[C.c_int, _dp, _ip, _dp, C.c_int, _dp, _ip,
                                              C.c_double, C.c_double, C.c_double, _ip]
        L.orc_optimize.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, _dp, _dp]
        L.orc_optimize_until.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_void_p, C.c_void_p, _dp, _dp, _ip]
        for f in ("orc_destroy", "orc_add_poses", "orc_add_landmarks", "orc_add_odometry_edges",
                  "orc_add_observation_edges", "orc_set_fixed_pose", "orc_set_fixed_landmark",
                  "orc_num_poses", "orc_num_landmarks", "orc_num_odometry_edges",
                  "orc_num_observation_edges", "orc_get_poses", "orc_get_landmarks", "orc_set_poses",
                  "orc_set_landmarks", "orc_chi2", "orc_linearize_blocks", "orc_build_system",
                  "orc_system_n", "orc_system_nnz", "orc_system_colptr", "orc_system_rowind",
                  "orc_system_values", "orc_system_b", "orc_solve_ldlt", "orc_apply_update", "orc_set_elimination_order", "orc_vertex_offsets",
                  "orc_get_delta"):
            fn = getattr(L, f)
            if fn.argtypes is None:
                pass
        _lib = L
    return _lib


_ref = None


def ref_eigen():
    """The reference's vendored Eigen SimplicialLDLT/LLT (oracle/_ref), or None if not built."""
    global _ref
    if _ref is None:
        if not os.path.exists(REF):
            return None
        R = C.CDLL(REF)
        R.ref_eigen_create.restype = C.c_void_p
        R.ref_eigen_create.argtypes = [C.c_int]
        R.ref_eigen_destroy.argtypes = [C.c_void_p]
        R.ref_eigen_solve.argtypes = [C.c_void_p, C.c_int, C.c_int, _ip, _ip, _dp, _dp, _dp]
        R.ref_eigen_timings.argtypes = [C.c_void_p, _dp]
        R.ref_eigen_reset_timings.argtypes = [C.c_void_p]
        R.ref_eigen_version.restype = C.c_char_p
        R.ref_eigen_rotation2d.argtypes = [C.c_double, _dp]
        _ref = R
    return _ref


class EigenSolver:
    """Handle on the reference's Eigen solver; pass .fn/.ctx to OracleGraph.optimize."""

    def __init__(self, kind=0):
        R = ref_eigen()
        if R is None:
            raise RuntimeError("oracle/_ref/libref_eigen.so not built (needs /root/reference)")
        self.R = R
        self.ctx = C.c_void_p(R.ref_eigen_create(kind))
        self.fn = C.cast(R.ref_eigen_solve, C.c_void_p)

    def solve(self, n, colptr, rowind, values, b, analyze=True):
        x = np.zeros(n)
        rc = self.R.ref_eigen_solve(self.ctx, int(analyze), n, _i(colptr), _i(rowind), _d(values), _d(b), _d(x))
        if rc != 0:
            raise RuntimeError("Eigen factorisation failed")
        return x

    def timings(self):
        t = np.zeros(3)
        self.R.ref_eigen_timings(self.ctx, _d(t))
        return t

    def reset_timings(self):
        self.R.ref_eigen_reset_timings(self.ctx)

    def __del__(self):
        try:
            self.R.ref_eigen_destroy(self.ctx)
        except Exception:
            pass


class OracleGraph:
    def __init__(self):
        self.L = lib()
        self.g = C.c_void_p(self.L.orc_create())

    def __del__(self):
        try:
            self.L.orc_destroy(self.g)
        except Exception:
            pass

    # ---- construction (indices, not ids) ----
    def add_poses(self, est):
        est = np.ascontiguousarray(est, dtype=np.float64).reshape(-1, 3)
        self.L.orc_add_poses(self.g, len(est), _d(est))

    def add_landmarks(self, est):
        est = np.ascontiguousarray(est, dtype=np.float64).reshape(-1, 2)
        self.L.orc_add_landmarks(self.g, len(est), _d(est))

    def add_odometry_edges(self, i, j, z, info):
        i = np.ascontiguousarray(i, dtype=np.int32); j = np.ascontiguousarray(j, dtype=np.int32)
        z = np.ascontiguousarray(z, dtype=np.float64).reshape(-1, 3)
        info = np.ascontiguousarray(info, dtype=np.float64).reshape(-1, 9)
        rc = self.L.orc_add_odometry_edges(self.g, len(i), _i(i), _i(j), _d(z), _d(info))
        assert rc >= 0

    def add_observation_edges(self, p, l, z, info):
        p = np.ascontiguousarray(p, dtype=np.int32); l = np.ascontiguousarray(l, dtype=np.int32)
        z = np.ascontiguousarray(z, dtype=np.float64).reshape(-1, 2)
        info = np.ascontiguousarray(info, dtype=np.float64).reshape(-1, 4)
        rc = self.L.orc_add_observation_edges(self.g, len(p), _i(p), _i(l), _d(z), _d(info))
        assert rc >= 0

    def set_fixed_pose(self, i, fixed=True):
        self.L.orc_set_fixed_pose(self.g, int(i), int(fixed))

    def set_fixed_landmark(self, l, fixed=True):
        self.L.orc_set_fixed_landmark(self.g, int(l), int(fixed))

    # ---- sizes / state ----
    @property
    def n_poses(self): return self.L.orc_num_poses(self.g)
    @property
    def n_landmarks(self): return self.L.orc_num_landmarks(self.g)
    @property
    def n_pp(self): return self.L.orc_num_odometry_edges(self.g)
    @property
    def n_pl(self): return self.L.orc_num_observation_edges(self.g)

    def poses(self):
        out = np.zeros((self.n_poses, 3)); self.L.orc_get_poses(self.g, _d(out)); return out

    def landmarks(self):
        out = np.zeros((self.n_landmarks, 2)); self.L.orc_get_landmarks(self.g, _d(out)); return out

    def set_poses(self, a):
        a = np.ascontiguousarray(a, dtype=np.float64); self.L.orc_set_poses(self.g, _d(a))

    def set_landmarks(self, a):
        a = np.ascontiguousarray(a, dtype=np.float64); self.L.orc_set_landmarks(self.g, _d(a))

    # ---- arithmetic ----
    def chi2(self):
        return float(self.L.orc_chi2(self.g))

    def linearize_blocks(self):
        N, M, Epp, Epl = self.n_poses, self.n_landmarks, self.n_pp, self.n_pl
        out = dict(Hpp_diag=np.zeros((N, 9)), Hll_diag=np.zeros((M, 4)), Hpp_off=np.zeros((Epp, 9)),
                   Hpl=np.zeros((Epl, 6)), b_pose=np.zeros((N, 3)), b_lm=np.zeros((M, 2)))
        self.L.orc_linearize_blocks(self.g, _d(out["Hpp_diag"]), _d(out["Hll_diag"]), _d(out["Hpp_off"]),
                                    _d(out["Hpl"]), _d(out["b_pose"]), _d(out["b_lm"]))
        return out

    def build_system(self):
        """Returns (n, colptr, rowind, values, b) of the scalar upper CCS (copies)."""
        n = self.L.orc_build_system(self.g)
        nnz = self.L.orc_system_nnz(self.g)
        colptr = np.ctypeslib.as_array(self.L.orc_system_colptr(self.g), (n + 1,)).copy()
        rowind = np.ctypeslib.as_array(self.L.orc_system_rowind(self.g), (max(nnz, 1),))[:nnz].copy()
        values = np.ctypeslib.as_array(self.L.orc_system_values(self.g), (max(nnz, 1),))[:nnz].copy()
        b = np.ctypeslib.as_array(self.L.orc_system_b(self.g), (max(n, 1),))[:n].copy()
        return n, colptr, rowind, values, b

    def set_elimination_order_like(self, pose_gidx, lm_gidx):
        """ordering=2: eliminate in the order of another exact solver, given as the first scalar of every vertex in ITS
        elimination order (-1 fixed) — e.g. pose_gidx / lm_gidx of the HIP back-end's nested-dissection plan (gs_plan_export)."""
        n = self.L.orc_build_system(self.g)
        po_ = np.zeros(self.n_poses, dtype=np.int32); lo_ = np.zeros(self.n_landmarks, dtype=np.int32)
        assert self.L.orc_vertex_offsets(self.g, _i(po_), _i(lo_)) == n
        perm = -np.ones(n, dtype=np.int32)
        pg = np.asarray(pose_gidx, dtype=np.int64); lg = np.asarray(lm_gidx, dtype=np.int64)
        assert np.array_equal(pg >= 0, po_ >= 0) and np.array_equal(lg >= 0, lo_ >= 0), "the two solvers disagree on the fixed vertices"
        for t in range(3): perm[pg[pg >= 0] + t] = po_[po_ >= 0] + t
        for t in range(2): perm[lg[lg >= 0] + t] = lo_[lo_ >= 0] + t
        rc = self.L.orc_set_elimination_order(self.g, _i(np.ascontiguousarray(perm)), n)
        if rc != 0: raise ValueError("not a permutation (%d)" % rc)

    def solve_ldlt(self, ordering=1):
        n = self.L.orc_system_n(self.g)
        x = np.zeros(n)
        rc = self.L.orc_solve_ldlt(self.g, ordering, _d(x))
        if rc != 0:
            raise RuntimeError("oracle LDLT failed rc=%d" % rc)
        return x

    def apply_update(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64); self.L.orc_apply_update(self.g, _d(x))

    def delta(self):
        dp = np.zeros((self.n_poses, 3)); dl = np.zeros((self.n_landmarks, 2))
        self.L.orc_get_delta(self.g, _d(dp), _d(dl)); return dp, dl

    def optimize(self, iterations, ordering=1, solver=None):
        """Returns (iterations_done, chi2[it], timings_ms[5])."""
        chi = np.zeros(max(iterations, 1)); tm = np.zeros(5)
        fn = solver.fn if solver is not None else None
        ctx = solver.ctx if solver is not None else None
        done = self.L.orc_optimize(self.g, iterations, ordering, fn, ctx, _d(chi), _d(tm))
        return done, chi[:iterations], tm

    def optimize_until(self, max_iterations, rel_chi2_tol, ordering=1, solver=None):
        """The stop rule of BASELINE config 2 (rel_chi2_tol < 0: none).  Returns (updates_applied, chi2[it], failed)."""
        chi = np.zeros(max(max_iterations, 1)); tm = np.zeros(5); failed = C.c_int(0)
        fn = solver.fn if solver is not None else None
        ctx = solver.ctx if solver is not None else None
        done = self.L.orc_optimize_until(self.g, max_iterations, float(rel_chi2_tol), ordering, fn, ctx, _d(chi), _d(tm),
                                         C.byref(failed))
        return done, chi[:max_iterations], bool(failed.value)


class OracleFrontend:
    """A0/A1 on the CPU oracle (same call shapes as the product's GPU front end)."""

    def __init__(self, lidar_to_cog=1.5):
        self.L = lib(); self.lidar = float(lidar_to_cog)

    def polar_to_xy(self, az, zen, dist):
        az = np.ascontiguousarray(az, dtype=np.float64); zen = np.ascontiguousarray(zen, dtype=np.float64)
        dist = np.ascontiguousarray(dist, dtype=np.float64)
        out = np.zeros((len(az), 2))
        self.L.orc_polar_to_xy_batch(len(az), _d(az), _d(zen), _d(dist), self.lidar, _d(out))
        return out

    def cone_to_global(self, poses, pose_of_obs, obs):
        poses = np.ascontiguousarray(poses, dtype=np.float64).reshape(-1, 3)
        obs = np.ascontiguousarray(obs, dtype=np.float64).reshape(-1, 4)
        pose_of_obs = np.ascontiguousarray(pose_of_obs, dtype=np.int32)
        out = np.zeros((len(obs), 2))
        self.L.orc_cone_to_global_batch(len(obs), _d(poses), _i(pose_of_obs), _d(obs), self.lidar, _d(out))
        return out

    def associate(self, poses, pose_of_obs, obs, map_xy, map_type, thr, type_tol=1e-4):
        poses = np.ascontiguousarray(poses, dtype=np.float64).reshape(-1, 3)
        obs = np.ascontiguousarray(obs, dtype=np.float64).reshape(-1, 4)
        pose_of_obs = np.ascontiguousarray(pose_of_obs, dtype=np.int32)
        map_xy = np.ascontiguousarray(map_xy, dtype=np.float64).reshape(-1, 2)
        map_type = np.ascontiguousarray(map_type, dtype=np.int32)
        out = np.zeros(len(obs), dtype=np.int32)
        self.L.orc_associate_fixed_map(len(obs), _d(poses), _i(pose_of_obs), _d(obs), len(map_xy),
                                       _d(map_xy), _i(map_type), float(thr), float(type_tol), self.lidar, _i(out))
        return out


_ref_wgs84 = None


def ref_wgs84():
    """The reference's own WGS84 <-> Cartesian header, compiled into oracle/_ref/libref_wgs84.so (None if not built)."""
    global _ref_wgs84
    if _ref_wgs84 is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "libref_wgs84.so")
        if not os.path.exists(path):
            return None
        R = C.CDLL(path)
        dp = C.POINTER(C.c_double)
        R.ref_wgs84_to_cartesian.argtypes = [dp, dp, dp]; R.ref_wgs84_from_cartesian.argtypes = [dp, dp, dp]
        _ref_wgs84 = R
    return _ref_wgs84


def ref_to_cartesian(ref_latlon, pos_latlon):
    R = ref_wgs84(); o = np.zeros(2)
    R.ref_wgs84_to_cartesian(_d(np.ascontiguousarray(ref_latlon, dtype=np.float64)), _d(np.ascontiguousarray(pos_latlon, dtype=np.float64)), _d(o)); return o


def ref_from_cartesian(ref_latlon, xy):
    R = ref_wgs84(); o = np.zeros(2)
    R.ref_wgs84_from_cartesian(_d(np.ascontiguousarray(ref_latlon, dtype=np.float64)), _d(np.ascontiguousarray(xy, dtype=np.float64)), _d(o)); return o


_ref_cone = None


def ref_cone():
    """The reference's own Cone class (src/cone.cpp), compiled into oracle/_ref/libref_cone.so (None if not built)."""
    global _ref_cone
    if _ref_cone is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "libref_cone.so")
        if not os.path.exists(path):
            return None
        R = C.CDLL(path)
        dp, fp = C.POINTER(C.c_double), C.POINTER(C.c_float)
        R.ref_cone_encode.argtypes = [C.c_double, C.c_double, C.c_int, C.c_int, dp, fp, fp, fp]
        R.ref_cone_record.argtypes = [C.c_double, C.c_double, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, dp]
        _ref_cone = R
    return _ref_cone


def ref_cone_encode(x, y, ctype, cid, pose):
    """Cone(x, y, type, id).getDirection(pose), .getDistance(pose) of the reference: (azimuthAngle, zenithAngle, distance) as float32."""
    R = ref_cone(); az, zen, di = C.c_float(), C.c_float(), C.c_float()
    R.ref_cone_encode(float(x), float(y), int(ctype), int(cid), _d(np.ascontiguousarray(pose, dtype=np.float64)), C.byref(az), C.byref(zen), C.byref(di))
    return np.float32(az.value), np.float32(zen.value), np.float32(di.value)
