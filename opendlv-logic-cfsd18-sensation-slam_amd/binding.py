"""ctypes binding of include/graphslam.h (libgraphslam_hip.so).

This is plumbing above the C-ABI: the product is the HIP library.  There is no CPU fallback: if the
shared library is missing or no gfx950 device is usable, calls raise.
"""
import ctypes as C
import os
import re
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_PATH = os.environ.get("GS_LIB") or os.path.join(CSRC, "libgraphslam_hip.so")    # GS_LIB: A/B builds of the same library (tuning)
HEADER = os.path.join(os.path.dirname(HERE), "include", "graphslam.h")
DEBUG_HEADER = os.path.join(os.path.dirname(HERE), "include", "graphslam_debug.h")     # tuning / fault injection / timestamps: not the drop-in boundary

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)

GS_OK = 0
ERRORS = {-1: "GS_ERR_INVALID", -2: "GS_ERR_DUPLICATE_ID", -3: "GS_ERR_UNKNOWN_ID", -4: "GS_ERR_NO_DEVICE",
          -5: "GS_ERR_HIP", -6: "GS_ERR_NOT_INITIALIZED", -7: "GS_ERR_EMPTY", -8: "GS_ERR_NUMERIC",
          -9: "GS_ERR_CAPACITY", -10: "GS_ERR_TIMEOUT"}


class GsError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("%s (%d): %s" % (ERRORS.get(code, "GS_ERR"), code, msg))
        self.code = code


class Config(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("device", C.c_int32), ("verbose", C.c_int32),
                ("leaf_poses", C.c_int32), ("factor_variant", C.c_int32), ("linearize_gather", C.c_int32),
                ("odometry_information", C.c_double), ("cone_information", C.c_double),
                ("same_cone_threshold", C.c_double), ("cone_mapping_threshold", C.c_double),
                ("lidar_to_cog", C.c_double), ("loop_closing_radius", C.c_double),
                ("loop_closing_min_index", C.c_int32), ("optimize_iterations", C.c_int32),
                ("reference_quirks", C.c_int32), ("optimize_every_keyframe", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("iterations", C.c_int32),
                ("n_free_poses", C.c_int32), ("n_free_landmarks", C.c_int32),
                ("n_odometry_edges", C.c_int32), ("n_observation_edges", C.c_int32),
                ("n_fronts", C.c_int32), ("n_levels", C.c_int32), ("max_front", C.c_int32),
                ("numeric_failure", C.c_int32),
                ("chi2_initial", C.c_double), ("chi2_final", C.c_double),
                ("ms_structure", C.c_double), ("ms_linearize", C.c_double), ("ms_factor", C.c_double),
                ("ms_backsolve", C.c_double), ("ms_update", C.c_double), ("ms_total", C.c_double),
                ("factor_flops", C.c_int64), ("factor_bytes", C.c_int64), ("ms_event_overhead", C.c_double),
                ("fell_back", C.c_int32), ("first_failure", C.c_int32), ("factor_variant", C.c_int32), ("n_big_fronts", C.c_int32),
                ("device_bytes", C.c_int64), ("n_own_fronts", C.c_int32), ("n_shared_fronts", C.c_int32), ("ms_plan_host", C.c_double),
                ("ms_linearize_kernel", C.c_double), ("n_growths", C.c_int32), ("n_subtrees", C.c_int32)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class ShellMsg(C.Structure):
    _fields_ = [("data_type", C.c_int32), ("sender_stamp", C.c_uint32), ("sample_time_us", C.c_int64),
                ("object_id", C.c_uint32), ("reserved", C.c_uint32), ("v", C.c_double * 3)]


class DebugOptions(C.Structure):
    """gs_debug_options (include/graphslam_debug.h): every tuning switch of a handle."""
    _fields_ = [("struct_size", C.c_int32)] + [(k, C.c_int32) for k in (
        "tree", "block_fronts", "leaf_kernel", "leaf_min", "bs_wide", "leaf_nt3", "f3_lds_kb", "reserved0",
        "leaf_poses", "cluster_ways", "ell_lanes", "big_cluster", "grow_headroom", "factor_variant",
        "grow", "grow_min_poses", "assoc_grid", "force_shared_top", "host_trig", "pool_poison", "plan_timing", "dbg", "subtree", "tickets", "shard_by_window")] + [("reserved", C.c_int32 * 5)]


# switches applied to every handle this process creates through the binding (tests: conftest sets grow_min_poses = 0)
DEFAULT_DEBUG = {}


class PlanInfo(C.Structure):
    _fields_ = [("n_scalar", C.c_int32), ("n_fronts", C.c_int32), ("n_levels", C.c_int32),
                ("max_front", C.c_int32), ("l_doubles", C.c_int64), ("u_doubles", C.c_int64),
                ("n_asm_blocks", C.c_int64), ("n_child_map", C.c_int64)]


def declared_symbols(debug=True):
    """Every function name include/graphslam.h (and, debug=True, include/graphslam_debug.h) declares."""
    names = set()
    for h in (HEADER, DEBUG_HEADER) if debug else (HEADER,):
        txt = open(h).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        names |= set(re.findall(r"\b(gs_[a-z0-9_]+)\s*\(", txt))
    return sorted(names)


def build(force=False):
    """Compile libgraphslam_hip.so for gfx950 (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC, "-j4"]
    if force:
        subprocess.check_call(["make", "-C", CSRC, "clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(cmd, stdout=subprocess.DEVNULL)


_lib = None


def lib():
    """Load the HIP library.  Raises if it has not been built: there is no fallback path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("libgraphslam_hip.so is missing (run __graft_entry__.build()); "
                           "the GraphSLAM back-end has no CPU fallback")
    L = C.CDLL(LIB_PATH)
    L.gs_last_error.restype = C.c_char_p
    L.gs_linearize_bytes.restype = C.c_int64
    L.gs_dist_exchange_doubles.restype = C.c_int64
    L.gs_slam_graph.restype = C.c_void_p
    vp = C.c_void_p
    L.gs_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    L.gs_slam_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    for name in ("gs_destroy", "gs_clear", "gs_initialize_optimization", "gs_iterate", "gs_sync_estimates",
                 "gs_stream_synchronize", "gs_linearize", "gs_num_poses", "gs_num_landmarks",
                 "gs_num_odometry_edges", "gs_num_observation_edges", "gs_dist_iterate_local",
                 "gs_dist_iterate_finish", "gs_slam_destroy", "gs_slam_map_size", "gs_slam_loop_closed",
                 "gs_slam_current_cone_index"):
        getattr(L, name).argtypes = [vp]
    L.gs_linearize_bytes.argtypes = [vp]
    L.gs_debug_timestamps.argtypes = [vp, C.POINTER(C.c_int64)]
    L.gs_debug_front_times.argtypes = [vp, C.POINTER(C.c_int64), C.c_int64]
    L.gs_dist_exchange_doubles.argtypes = [vp]
    L.gs_slam_graph.argtypes = [vp]
    L.gs_set_stream.argtypes = [vp, vp]
    L.gs_dist_set_exchange_buffer.argtypes = [vp, vp]
    L.gs_dist_configure.argtypes = [vp, C.c_int32, C.c_int32]
    if hasattr(L, "gs_dist_window_starts"):               # (a tuning build of an older tree loaded through GS_LIB may predate rank-local ingestion)
        L.gs_dist_window_starts.argtypes = [vp, C.POINTER(C.c_int32), C.c_int32]
        L.gs_dist_local_landmark_windows.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_int32]
        L.gs_dist_set_landmark_windows.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_int32]
        L.gs_dist_share_landmark_windows.argtypes = [vp]
    L.gs_add_pose.argtypes = [vp, C.c_int32, _dp]
    L.gs_add_landmark.argtypes = [vp, C.c_int32, _dp]
    L.gs_add_odometry_edge.argtypes = [vp, C.c_int32, C.c_int32, _dp, _dp]
    L.gs_add_observation_edge.argtypes = [vp, C.c_int32, C.c_int32, _dp, _dp]
    L.gs_add_poses.argtypes = [vp, C.c_int32, _ip, _dp]
    L.gs_add_landmarks.argtypes = [vp, C.c_int32, _ip, _dp]
    L.gs_add_odometry_edges.argtypes = [vp, C.c_int32, _ip, _ip, _dp, _dp]
    L.gs_add_observation_edges.argtypes = [vp, C.c_int32, _ip, _ip, _dp, _dp]
    L.gs_set_fixed_pose.argtypes = [vp, C.c_int32, C.c_int32]
    L.gs_set_fixed_landmark.argtypes = [vp, C.c_int32, C.c_int32]
    L.gs_set_pose_estimate.argtypes = [vp, C.c_int32, _dp]
    L.gs_set_landmark_estimate.argtypes = [vp, C.c_int32, _dp]
    L.gs_get_pose.argtypes = [vp, C.c_int32, _dp]
    L.gs_get_landmark.argtypes = [vp, C.c_int32, _dp]
    L.gs_get_poses.argtypes = [vp, C.c_int32, _ip, _dp]
    L.gs_get_landmarks.argtypes = [vp, C.c_int32, _ip, _dp]
    L.gs_optimize.argtypes = [vp, C.c_int32, C.POINTER(Stats)]
    L.gs_optimize_until.argtypes = [vp, C.c_int32, C.c_double, C.POINTER(Stats)]
    L.gs_debug_fail_at_iteration.argtypes = [vp, C.c_int32, C.c_int32]
    L.gs_debug_select_factor_variant.argtypes = [C.c_int32, C.c_int32, C.c_int64]
    L.gs_debug_options_default.argtypes = [C.POINTER(DebugOptions)]
    L.gs_debug_get_options.argtypes = [vp, C.POINTER(DebugOptions)]
    L.gs_debug_set_options.argtypes = [vp, C.POINTER(DebugOptions)]
    L.gs_chi2.argtypes = [vp, _dp]
    L.gs_get_stats.argtypes = [vp, C.POINTER(Stats)]
    L.gs_time_linearize.argtypes = [vp, C.c_int32, _dp]
    L.gs_export_system.argtypes = [vp, _dp, _dp, _dp, _dp, _dp, _dp, _ip, _ip]
    L.gs_export_delta.argtypes = [vp, _dp, _dp]
    L.gs_time_iterations.argtypes = [vp, C.c_int32, C.POINTER(Stats)]
    L.gs_plan_build_host.argtypes = [vp, C.POINTER(PlanInfo)]
    L.gs_plan_export.argtypes = [vp, _ip, C.POINTER(C.c_int64)]
    L.gs_reserve_device.argtypes = [vp, C.c_int64]
    L.gs_plan_growths.argtypes = [vp]; L.gs_growth_refusal.argtypes = [vp]; L.gs_growth_refusal.restype = C.c_char_p
    L.gs_polar_to_xy_batch.argtypes = [vp, C.c_int32, _dp, _dp, _dp, _dp]
    L.gs_cone_to_global_batch.argtypes = [vp, C.c_int32, _dp, C.c_int32, _ip, _dp, _dp]
    L.gs_associate_batch.argtypes = [vp, C.c_int32, _dp, C.c_int32, _ip, _dp, C.c_int32, _dp, _ip,
                                     C.c_double, C.c_double, _ip]
    L.gs_associate_resident.argtypes = [vp, C.c_int32, vp, C.c_int32, vp, vp, C.c_double, C.c_double, vp]
    L.gs_debug_time_associate_resident.argtypes = [vp, C.c_int32, vp, C.c_int32, vp, vp, C.c_double, C.c_double, vp, C.c_int32, _dp]
    L.gs_map_clear.argtypes = [vp]; L.gs_map_size.argtypes = [vp]
    L.gs_map_append.argtypes = [vp, C.c_int32, _dp, _ip]
    L.gs_map_set_xy.argtypes = [vp, C.c_int32, C.c_int32, _dp]
    L.gs_frame_frontend.argtypes = [vp, _dp, _dp, C.c_int32, C.c_double, C.c_double, C.c_int32, _dp, _dp, _ip]
    L.gs_dist_unique_id.argtypes = [C.c_char_p]
    L.gs_dist_comm_init.argtypes = [vp, C.c_char_p, C.c_int32, C.c_int32]
    L.gs_dist_set_communicator.argtypes = [vp, vp]
    L.gs_dist_iterate.argtypes = [vp]
    L.gs_dist_optimize.argtypes = [vp, C.c_int32, C.POINTER(Stats)]
    L.gs_debug_time_exchange.argtypes = [vp, C.c_int32, _dp]
    L.gs_dist_read_exchange.argtypes = [vp, _dp]
    L.gs_dist_write_exchange.argtypes = [vp, _dp]
    u8 = C.POINTER(C.c_uint8)
    L.gs_dist_known.argtypes = [vp, u8, u8, u8, u8]
    L.gs_slam_collect_extract.argtypes = [vp, C.POINTER(C.c_int32), _dp]
    L.gs_shell_create.argtypes = [C.c_int32, C.POINTER(C.c_char_p), C.c_int32, C.POINTER(vp)]
    L.gs_shell_destroy.argtypes = [vp]
    L.gs_shell_on_message.argtypes = [vp, C.POINTER(ShellMsg), C.c_int64]
    L.gs_shell_poll.argtypes = [vp, C.c_int64]
    L.gs_shell_pending_output.argtypes = [vp]
    L.gs_shell_take_output.argtypes = [vp, C.c_int32, C.POINTER(ShellMsg)]
    L.gs_shell_counters.argtypes = [vp, C.POINTER(C.c_int64)]
    L.gs_shell_cid.argtypes = [vp]
    L.gs_shell_slam.argtypes = [vp]; L.gs_shell_slam.restype = vp
    L.gs_slam_perform.argtypes = [vp, _dp, _dp, C.c_int32]
    L.gs_slam_get_map.argtypes = [vp, C.c_int32, _dp, _ip]
    L.gs_slam_get_send_pose.argtypes = [vp, _dp]
    L.gs_slam_collect_direction.argtypes = [vp, C.c_uint32, C.c_double, C.c_double]
    L.gs_slam_collect_distance.argtypes = [vp, C.c_uint32, C.c_double]
    L.gs_slam_collect_type.argtypes = [vp, C.c_uint32, C.c_uint32]
    L.gs_slam_collect_flush.argtypes = [vp, _dp, C.POINTER(C.c_int32), _dp]
    L.gs_slam_encode_cones.argtypes = [vp, C.c_int32, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_int32)]
    L.gs_cone_encode.argtypes = [_dp, _dp, C.c_int32, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.gs_wgs84_to_cartesian.argtypes = [_dp, _dp, _dp]
    L.gs_wgs84_from_cartesian.argtypes = [_dp, _dp, _dp]
    L.gs_slam_set_gps_reference.argtypes = [vp, C.c_double, C.c_double]
    L.gs_slam_next_wgs84.argtypes = [vp, C.c_double, C.c_double]
    L.gs_slam_next_heading.argtypes = [vp, C.c_double]
    L.gs_slam_next_geolocation.argtypes = [vp, C.c_double, C.c_double, C.c_double]
    L.gs_slam_next_yaw_rate.argtypes = [vp, C.c_double]
    L.gs_slam_get_odometry.argtypes = [vp, _dp]
    L.gs_slam_set_sample_times.argtypes = [vp, C.c_int64, C.c_int64]
    L.gs_slam_encode_pose.argtypes = [vp, C.POINTER(C.c_float)]
    _lib = L
    return L


def _f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a if shape is None else a.reshape(shape)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _d(a):
    return a.ctypes.data_as(_dp)


def _i(a):
    return a.ctypes.data_as(_ip)


def default_config(**kw):
    cfg = Config()
    lib().gs_config_default(C.byref(cfg))
    for k, v in kw.items():
        setattr(cfg, k, v)
    return cfg


_hip = None


def hip_runtime():
    """libamdhip64 (the runtime the library itself uses), for callers that keep their OWN buffers in device memory (gs_associate_resident,
    gs_dist_set_exchange_buffer): tests and bench.py — a C++ consumer calls hipMalloc / hipMemcpy itself."""
    global _hip
    if _hip is None:
        lib()                                               # the library has loaded the runtime already
        H = C.CDLL("libamdhip64.so")
        H.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]; H.hipFree.argtypes = [C.c_void_p]
        H.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        _hip = H
    return _hip


class DeviceArray:
    """A caller-owned array in device memory (hipMalloc + hipMemcpy), e.g. the observations of gs_associate_resident."""

    def __init__(self, host=None, nbytes=None):
        H = hip_runtime()
        self.host = None if host is None else np.ascontiguousarray(host)
        self.nbytes = int(nbytes if nbytes is not None else self.host.nbytes)
        p = C.c_void_p()
        if H.hipMalloc(C.byref(p), max(self.nbytes, 8)) != 0:
            raise RuntimeError("hipMalloc failed")
        self.ptr = p
        if self.host is not None and self.nbytes:
            if H.hipMemcpy(self.ptr, self.host.ctypes.data_as(C.c_void_p), self.nbytes, 1) != 0:
                raise RuntimeError("hipMemcpy (host to device) failed")

    def to_host(self, dtype, count):
        out = np.zeros(count, dtype=dtype)
        if hip_runtime().hipMemcpy(out.ctypes.data_as(C.c_void_p), self.ptr, out.nbytes, 2) != 0:
            raise RuntimeError("hipMemcpy (device to host) failed")
        return out

    def free(self):
        if getattr(self, "ptr", None):
            hip_runtime().hipFree(self.ptr); self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def dist_unique_id():
    """ncclGetUniqueId through the library (128 bytes): made by one rank, handed to the others by whatever channel they share."""
    buf = C.create_string_buffer(128)
    rc = lib().gs_dist_unique_id(buf)
    if rc < 0:
        raise GsError(rc, (lib().gs_last_error() or b"").decode())
    return buf.raw


def device_count():
    return int(lib().gs_device_count())


class Graph:
    """Mirror of the calls Slam makes on g2o::SparseOptimizer (reference src/slam.cpp:53-65,433-484,525-550)."""

    def __init__(self, cfg=None, _handle=None, debug=None, **kw):
        self.L = lib()
        self._owned = _handle is None
        if _handle is not None:
            self.h = C.c_void_p(_handle)
            return
        if cfg is None:
            cfg = default_config(**kw)
        h = C.c_void_p()
        self._check(self.L.gs_create(C.byref(cfg), C.byref(h)))
        self.h = h
        if DEFAULT_DEBUG or debug:
            self.set_debug(**dict(DEFAULT_DEBUG, **(debug or {})))

    # ---- tuning switches (include/graphslam_debug.h)
    def debug_options(self):
        o = DebugOptions(); self._check(self.L.gs_debug_get_options(self.h, C.byref(o))); return o

    def set_debug(self, **kw):
        """gs_debug_set_options: change the named switches of this handle, keep the others."""
        o = self.debug_options()
        for k, v in kw.items():
            if not hasattr(o, k):
                raise AttributeError("gs_debug_options has no field %r" % k)
            setattr(o, k, int(v))
        self._check(self.L.gs_debug_set_options(self.h, C.byref(o)))

    def _check(self, rc):
        if rc < 0:
            raise GsError(rc, (self.L.gs_last_error() or b"").decode())
        return rc

    def close(self):
        if getattr(self, "h", None) and self._owned:
            self.L.gs_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- construction
    def add_pose(self, pid, est):
        e = _f64(est); self._check(self.L.gs_add_pose(self.h, int(pid), _d(e)))

    def add_landmark(self, lid, est):
        e = _f64(est); self._check(self.L.gs_add_landmark(self.h, int(lid), _d(e)))

    def add_odometry_edge(self, i, j, z, info):
        z = _f64(z); info = _f64(info); self._check(self.L.gs_add_odometry_edge(self.h, int(i), int(j), _d(z), _d(info)))

    def add_observation_edge(self, p, l, z, info):
        z = _f64(z); info = _f64(info); self._check(self.L.gs_add_observation_edge(self.h, int(p), int(l), _d(z), _d(info)))

    def add_poses(self, ids, est):
        ids = _i32(ids); est = _f64(est); self._check(self.L.gs_add_poses(self.h, len(ids), _i(ids), _d(est)))

    def add_landmarks(self, ids, est):
        ids = _i32(ids); est = _f64(est); self._check(self.L.gs_add_landmarks(self.h, len(ids), _i(ids), _d(est)))

    def add_odometry_edges(self, i, j, z, info=None):
        i = _i32(i); j = _i32(j); z = _f64(z)
        ip = _d(_f64(info)) if info is not None else None
        if info is not None:
            info = _f64(info); ip = _d(info)
        self._check(self.L.gs_add_odometry_edges(self.h, len(i), _i(i), _i(j), _d(z), ip))

    def add_observation_edges(self, p, l, z, info=None):
        p = _i32(p); l = _i32(l); z = _f64(z); ip = None
        if info is not None:
            info = _f64(info); ip = _d(info)
        self._check(self.L.gs_add_observation_edges(self.h, len(p), _i(p), _i(l), _d(z), ip))

    def set_fixed_pose(self, pid, fixed=True):
        self._check(self.L.gs_set_fixed_pose(self.h, int(pid), int(fixed)))

    def set_fixed_landmark(self, lid, fixed=True):
        self._check(self.L.gs_set_fixed_landmark(self.h, int(lid), int(fixed)))

    def set_pose_estimate(self, pid, est):
        e = _f64(est); self._check(self.L.gs_set_pose_estimate(self.h, int(pid), _d(e)))

    def set_landmark_estimate(self, lid, est):
        e = _f64(est); self._check(self.L.gs_set_landmark_estimate(self.h, int(lid), _d(e)))

    def clear(self):
        self._check(self.L.gs_clear(self.h))

    # ---- read-back
    @property
    def n_poses(self): return self._check(self.L.gs_num_poses(self.h))
    @property
    def n_landmarks(self): return self._check(self.L.gs_num_landmarks(self.h))
    @property
    def n_pp(self): return self._check(self.L.gs_num_odometry_edges(self.h))
    @property
    def n_pl(self): return self._check(self.L.gs_num_observation_edges(self.h))

    def get_pose(self, pid):
        o = np.zeros(3); self._check(self.L.gs_get_pose(self.h, int(pid), _d(o))); return o

    def get_landmark(self, lid):
        o = np.zeros(2); self._check(self.L.gs_get_landmark(self.h, int(lid), _d(o))); return o

    def poses(self):
        n = self.n_poses; o = np.zeros((n, 3)); self._check(self.L.gs_get_poses(self.h, n, None, _d(o))); return o

    def landmarks(self):
        n = self.n_landmarks; o = np.zeros((n, 2)); self._check(self.L.gs_get_landmarks(self.h, n, None, _d(o))); return o

    # ---- optimisation
    def initialize_optimization(self):
        self._check(self.L.gs_initialize_optimization(self.h))

    def optimize(self, iterations=10, stats=True):
        if not stats:                                       # the way Slam calls it (csrc/gs_slam.cpp: no statistics asked for, no chi2 pass behind the iterations)
            return self._check(self.L.gs_optimize(self.h, int(iterations), None)), None
        st = Stats(); st.struct_size = C.sizeof(Stats)
        done = self._check(self.L.gs_optimize(self.h, int(iterations), C.byref(st)))
        return done, st

    def optimize_until(self, max_iterations, rel_chi2_tol):
        st = Stats(); st.struct_size = C.sizeof(Stats)
        done = self._check(self.L.gs_optimize_until(self.h, int(max_iterations), float(rel_chi2_tol), C.byref(st)))
        return done, st

    def debug_fail_at_iteration(self, k, code=1):
        self._check(self.L.gs_debug_fail_at_iteration(self.h, int(k), int(code)))

    def iterate(self):
        return self._check(self.L.gs_iterate(self.h))

    def synchronize(self):
        self._check(self.L.gs_stream_synchronize(self.h))

    def sync_estimates(self):
        self._check(self.L.gs_sync_estimates(self.h))

    def chi2(self):
        o = C.c_double(); self._check(self.L.gs_chi2(self.h, C.byref(o))); return o.value

    def stats(self):
        st = Stats(); self._check(self.L.gs_get_stats(self.h, C.byref(st))); return st

    def set_stream(self, stream_ptr):
        self._check(self.L.gs_set_stream(self.h, C.c_void_p(stream_ptr)))

    # ---- measurement / parity hooks
    def linearize(self):
        self._check(self.L.gs_linearize(self.h))

    def time_linearize(self, reps):
        o = C.c_double(); self._check(self.L.gs_time_linearize(self.h, int(reps), C.byref(o))); return o.value

    def linearize_bytes(self):
        return int(self.L.gs_linearize_bytes(self.h))

    def debug_front_times(self):
        """[2, n_fronts] completion times (100 MHz ticks) of the last factor / backward-solve launches (F3_DONE_TS builds)."""
        n = self.stats().n_fronts; buf = (C.c_int64 * (2 * n))()
        self._check(self.L.gs_debug_front_times(self.h, buf, 2 * n)); return np.array(buf[:], dtype=np.int64).reshape(2, n)

    def debug_timestamps(self):
        """100 MHz phase timestamps of one front (tuning aid, see include/graphslam.h gs_debug_timestamps)."""
        buf = (C.c_int64 * 64)(); self._check(self.L.gs_debug_timestamps(self.h, buf)); return np.array(buf[:], dtype=np.int64)

    def time_iterations(self, reps):
        st = Stats(); self._check(self.L.gs_time_iterations(self.h, int(reps), C.byref(st))); return st

    def export_system(self):
        """Block-sparse H and b of the last linearisation, edge arrays re-ordered to INSERTION order."""
        N, M, Epp, Epl = self.n_poses, self.n_landmarks, self.n_pp, self.n_pl
        o = dict(Hpp_diag=np.zeros((N, 9)), Hll_diag=np.zeros((M, 4)), Hpp_off=np.zeros((Epp, 9)),
                 Hpl=np.zeros((Epl, 6)), b_pose=np.zeros((N, 3)), b_lm=np.zeros((M, 2)))
        ppo = np.zeros(Epp, dtype=np.int32); plo = np.zeros(Epl, dtype=np.int32)
        self._check(self.L.gs_export_system(self.h, _d(o["Hpp_diag"]), _d(o["Hll_diag"]), _d(o["Hpp_off"]),
                                            _d(o["Hpl"]), _d(o["b_pose"]), _d(o["b_lm"]), _i(ppo), _i(plo)))
        hpp = np.zeros_like(o["Hpp_off"]); hpp[ppo] = o["Hpp_off"]; o["Hpp_off"] = hpp
        hpl = np.zeros_like(o["Hpl"]); hpl[plo] = o["Hpl"]; o["Hpl"] = hpl
        return o

    def export_delta(self):
        dp = np.zeros((self.n_poses, 3)); dl = np.zeros((self.n_landmarks, 2))
        self._check(self.L.gs_export_delta(self.h, _d(dp), _d(dl))); return dp, dl

    # ---- host-only plan (no device work)
    def plan_build_host(self):
        info = PlanInfo(); self._check(self.L.gs_plan_build_host(self.h, C.byref(info))); return info

    def reserve_device(self, nbytes):
        """takes device memory for the first structure phase now (start-up) instead of inside the first optimize()"""
        self._check(self.L.gs_reserve_device(self.h, int(nbytes)))

    def plan_growths(self):
        """append-only growth steps the current plan has absorbed (0: the plan is a full build)"""
        return self._check(self.L.gs_plan_growths(self.h))

    def growth_refusal(self):
        """why the last structure change was not absorbed by growing the plan ('' if it was)"""
        return self.L.gs_growth_refusal(self.h).decode()

    def plan_export(self):
        n = C.c_int64(0)
        self._check(self.L.gs_plan_export(self.h, None, C.byref(n)))
        out = np.zeros(n.value, dtype=np.int32)
        self._check(self.L.gs_plan_export(self.h, _i(out), C.byref(n)))
        return out

    # ---- front end
    def polar_to_xy(self, az, zen, dist):
        az = _f64(az); zen = _f64(zen); dist = _f64(dist); out = np.zeros((len(az), 2))
        self._check(self.L.gs_polar_to_xy_batch(self.h, len(az), _d(az), _d(zen), _d(dist), _d(out))); return out

    def cone_to_global(self, poses, pose_of_obs, obs):
        poses = _f64(poses, (-1, 3)); obs = _f64(obs, (-1, 4)); po = _i32(pose_of_obs); out = np.zeros((len(obs), 2))
        self._check(self.L.gs_cone_to_global_batch(self.h, len(obs), _d(poses), len(poses), _i(po), _d(obs), _d(out)))
        return out

    def associate(self, poses, pose_of_obs, obs, map_xy, map_type, thr, type_tol=1e-4):
        poses = _f64(poses, (-1, 3)); obs = _f64(obs, (-1, 4)); po = _i32(pose_of_obs)
        map_xy = _f64(map_xy, (-1, 2)); map_type = _i32(map_type); out = np.zeros(len(obs), dtype=np.int32)
        self._check(self.L.gs_associate_batch(self.h, len(obs), _d(poses), len(poses), _i(po), _d(obs), len(map_xy),
                                              _d(map_xy), _i(map_type), float(thr), float(type_tol), _i(out)))
        return out

    # ---- per-keyframe front end against the resident map
    def associate_resident(self, d_poses, n_poses, d_pose_of_obs, d_obs, n, thr, d_out, type_tol=1e-4):
        """gs_associate_resident: the resident map (map_append), everything else DeviceArray; asynchronous (synchronize() before reading d_out)."""
        self._check(self.L.gs_associate_resident(self.h, int(n), d_poses.ptr, int(n_poses), d_pose_of_obs.ptr, d_obs.ptr, float(thr), float(type_tol), d_out.ptr))

    def time_associate_resident(self, d_poses, n_poses, d_pose_of_obs, d_obs, n, thr, d_out, reps, type_tol=1e-4):
        ms = C.c_double()
        self._check(self.L.gs_debug_time_associate_resident(self.h, int(n), d_poses.ptr, int(n_poses), d_pose_of_obs.ptr, d_obs.ptr, float(thr), float(type_tol), d_out.ptr,
                                                             int(reps), C.byref(ms)))
        return ms.value

    def map_clear(self): self._check(self.L.gs_map_clear(self.h))
    def map_size(self): return self._check(self.L.gs_map_size(self.h))

    def map_append(self, xy, types):
        xy = _f64(xy, (-1, 2)); ty = _i32(types); self._check(self.L.gs_map_append(self.h, len(ty), _d(xy), _i(ty)))

    def map_set_xy(self, first, xy):
        xy = _f64(xy, (-1, 2)); self._check(self.L.gs_map_set_xy(self.h, int(first), len(xy), _d(xy)))

    def frame_frontend(self, pose, obs, thr, type_tol=1e-4, signed_type=0):
        """(zxy [k,2], gxy [k,2], idx [k]) of one frame's observations [k,4] against the resident map."""
        pose = _f64(pose); obs = _f64(obs, (-1, 4)); k = len(obs)
        z = np.zeros((k, 2)); gx = np.zeros((k, 2)); idx = np.zeros(k, dtype=np.int32)
        self._check(self.L.gs_frame_frontend(self.h, _d(pose), _d(obs), k, float(thr), float(type_tol), int(signed_type), _d(z), _d(gx), _i(idx)))
        return z, gx, idx

    # ---- pose-window shards (one handle per rank / GPU)
    def dist_configure(self, rank, world):
        self._check(self.L.gs_dist_configure(self.h, int(rank), int(world)))

    # rank-local ingestion (include/graphslam.h, gs_dist_set_landmark_windows)
    def dist_window_starts(self, world):
        out = np.zeros(world + 1, dtype=np.int32)
        self._check(self.L.gs_dist_window_starts(self.h, out.ctypes.data_as(C.POINTER(C.c_int32)), len(out)))
        return out

    def dist_local_landmark_windows(self, n_landmarks):
        a = np.zeros(n_landmarks, dtype=np.uint64); b = np.zeros(n_landmarks, dtype=np.uint64); u = C.POINTER(C.c_uint64)
        self._check(self.L.gs_dist_local_landmark_windows(self.h, a.ctypes.data_as(u), b.ctypes.data_as(u), n_landmarks))
        return a, b

    def dist_share_landmark_windows(self):
        self._check(self.L.gs_dist_share_landmark_windows(self.h))

    def dist_set_landmark_windows(self, seen_interior, seen_first):
        a = np.ascontiguousarray(seen_interior, dtype=np.uint64); b = np.ascontiguousarray(seen_first, dtype=np.uint64); u = C.POINTER(C.c_uint64)
        assert len(a) == len(b)
        self._check(self.L.gs_dist_set_landmark_windows(self.h, a.ctypes.data_as(u), b.ctypes.data_as(u), len(a)))

    def load_bench_graph_shard(self, g, rank, world, masks=None):
        """Rank-local ingestion of the arrays bench_graph makes: every vertex and odometry edge, the observation edges of this rank's window, of the
        windows' first poses and of the fixed poses only, and the whole-graph landmark windows (`masks`, default: landmark_windows(g, world)).
        Returns the boolean selection of g's observation edges that went in (the handle's edge k is g's edge flatnonzero(keep)[k])."""
        N, M = len(g["pose_est"]), len(g["lm_est"])
        self.add_poses(np.arange(N), g["pose_est"]); self.add_landmarks(np.arange(M), g["lm_est"])
        self.add_odometry_edges(g["pp_i"], g["pp_j"], g["pp_z"], g["pp_info"])
        fixed = np.zeros(N, dtype=bool)
        for i in g["fixed_poses"]:
            self.set_fixed_pose(int(i)); fixed[int(i)] = True
        for l in g["fixed_landmarks"]:
            self.set_fixed_landmark(int(l))
        self.dist_configure(rank, world)
        first = self.dist_window_starts(world)
        p = np.asarray(g["pl_p"])
        keep = ((p >= first[rank]) & (p < first[rank + 1])) | np.isin(p, first[1:world]) | fixed[p]
        self.add_observation_edges(p[keep], np.asarray(g["pl_l"])[keep], np.asarray(g["pl_z"])[keep], np.asarray(g["pl_info"])[keep])
        if masks is None:
            masks = landmark_windows(g, world)
        self.dist_set_landmark_windows(*masks)
        return keep

    def dist_exchange_doubles(self):
        return int(self.L.gs_dist_exchange_doubles(self.h))

    def dist_set_exchange_buffer(self, device_ptr):
        self._check(self.L.gs_dist_set_exchange_buffer(self.h, C.c_void_p(device_ptr)))

    def dist_comm_init(self, unique_id, rank, world):
        """ncclCommInitRank inside the library (collective: every rank calls it with the id rank 0 made, dist_unique_id())."""
        self._check(self.L.gs_dist_comm_init(self.h, bytes(unique_id), int(rank), int(world)))

    def dist_set_communicator(self, comm_ptr):
        self._check(self.L.gs_dist_set_communicator(self.h, C.c_void_p(comm_ptr)))

    def dist_iterate(self):
        """one sharded iteration, all from C++: local half -> ncclAllReduce of the exchange buffer -> finish"""
        return self._check(self.L.gs_dist_iterate(self.h))

    def dist_optimize(self, iterations=10):
        st = Stats(); st.struct_size = C.sizeof(Stats)
        done = self._check(self.L.gs_dist_optimize(self.h, int(iterations), C.byref(st)))
        return done, st

    def time_exchange(self, reps):
        ms = C.c_double(); self._check(self.L.gs_debug_time_exchange(self.h, int(reps), C.byref(ms))); return ms.value

    def dist_iterate_local(self):
        self._check(self.L.gs_dist_iterate_local(self.h))

    def dist_iterate_finish(self):
        self._check(self.L.gs_dist_iterate_finish(self.h))

    def dist_read_exchange(self):
        out = np.zeros(max(self.dist_exchange_doubles(), 1))
        self._check(self.L.gs_dist_read_exchange(self.h, _d(out)))
        return out[:self.dist_exchange_doubles()]

    def dist_write_exchange(self, buf):
        buf = _f64(buf)
        if len(buf) == 0:
            buf = np.zeros(1)
        self._check(self.L.gs_dist_write_exchange(self.h, _d(buf)))

    def dist_known(self):
        """(pose_known, lm_known, pose_primary, lm_primary) boolean arrays, insertion order."""
        N, M = self.n_poses, self.n_landmarks
        a = [np.zeros(max(N, 1), dtype=np.uint8), np.zeros(max(M, 1), dtype=np.uint8),
             np.zeros(max(N, 1), dtype=np.uint8), np.zeros(max(M, 1), dtype=np.uint8)]
        u8 = C.POINTER(C.c_uint8)
        self._check(self.L.gs_dist_known(self.h, *[x.ctypes.data_as(u8) for x in a]))
        return a[0][:N].astype(bool), a[1][:M].astype(bool), a[2][:N].astype(bool), a[3][:M].astype(bool)

    # ---- convenience: load the arrays of track.bench_graph (ids = indices)
    def load_bench_graph(self, g):
        N, M = len(g["pose_est"]), len(g["lm_est"])
        self.add_poses(np.arange(N), g["pose_est"]); self.add_landmarks(np.arange(M), g["lm_est"])
        self.add_odometry_edges(g["pp_i"], g["pp_j"], g["pp_z"], g["pp_info"])
        self.add_observation_edges(g["pl_p"], g["pl_l"], g["pl_z"], g["pl_info"])
        for i in g["fixed_poses"]:
            self.set_fixed_pose(int(i))
        for l in g["fixed_landmarks"]:
            self.set_fixed_landmark(int(l))


def landmark_windows(g, world):
    """The whole-graph landmark windows of gs_dist_set_landmark_windows, from bench_graph arrays in numpy: (seen_interior, seen_first), uint64 per landmark,
    bit w = an interior pose / the first pose of window w observes it (windows = contiguous runs of the FREE poses, ceil(w * n_free / world) first)."""
    N, M = len(g["pose_est"]), len(g["lm_est"])
    fixed_p = np.zeros(N, dtype=bool); fixed_p[np.asarray(g["fixed_poses"], dtype=np.int64)] = True
    fixed_l = np.zeros(M, dtype=bool); fixed_l[np.asarray(g["fixed_landmarks"], dtype=np.int64)] = True
    fp = np.cumsum(~fixed_p) - 1; nfree = int((~fixed_p).sum())
    wf = np.array([(w * nfree + world - 1) // world for w in range(world + 1)], dtype=np.int64)
    p = np.asarray(g["pl_p"]); l = np.asarray(g["pl_l"]); ok = ~fixed_p[p] & ~fixed_l[l]
    fpe = fp[p[ok]]; le = l[ok]
    w = np.minimum(np.searchsorted(wf, fpe, side="right") - 1, world - 1)
    first = (w >= 1) & (fpe == wf[w])
    seen_interior = np.zeros(M, dtype=np.uint64); seen_first = np.zeros(M, dtype=np.uint64)
    for x in range(world):
        sel = w == x
        seen_interior[np.unique(le[sel & ~first])] |= np.uint64(1 << x)
        seen_first[np.unique(le[sel & first])] |= np.uint64(1 << x)
    return seen_interior, seen_first


class Slam:
    """Mirror of the graph side of class Slam (reference src/slam.cpp:298-338 performSLAM and what it calls)."""

    def __init__(self, cfg=None, _handle=None, debug=None, **kw):
        self.L = lib()
        self._owned = _handle is None
        if _handle is not None:
            self.h = C.c_void_p(_handle)
        else:
            if cfg is None:
                cfg = default_config(**kw)
            h = C.c_void_p()
            rc = self.L.gs_slam_create(C.byref(cfg), C.byref(h))
            if rc < 0:
                raise GsError(rc, (self.L.gs_last_error() or b"").decode())
            self.h = h
        self.graph = Graph(_handle=self.L.gs_slam_graph(self.h))
        if _handle is None and (DEFAULT_DEBUG or debug):
            self.graph.set_debug(**dict(DEFAULT_DEBUG, **(debug or {})))

    def _check(self, rc):
        if rc < 0:
            raise GsError(rc, (self.L.gs_last_error() or b"").decode())
        return rc

    def close(self):
        if getattr(self, "h", None) and self._owned:
            self.L.gs_slam_destroy(self.h)
        self.h = None
        if getattr(self, "graph", None) is not None:
            self.graph.h = None                    # borrowed from the slam handle: gone with it (a NULL handle is refused by the C-ABI)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def perform_slam(self, pose, cones_4xk):
        """cones: [K,4] rows (az deg, zen deg, dist m, type) = columns of the reference's collector matrix."""
        pose = _f64(pose); cones = _f64(cones_4xk, (-1, 4))
        self._check(self.L.gs_slam_perform(self.h, _d(pose), _d(cones), len(cones)))

    @property
    def map_size(self): return self._check(self.L.gs_slam_map_size(self.h))
    @property
    def loop_closed(self): return bool(self._check(self.L.gs_slam_loop_closed(self.h)))
    @property
    def current_cone_index(self): return self._check(self.L.gs_slam_current_cone_index(self.h))

    def map(self):
        n = self.map_size; xy = np.zeros((n, 2)); ty = np.zeros(n, dtype=np.int32)
        self._check(self.L.gs_slam_get_map(self.h, n, _d(xy), _i(ty))); return xy, ty

    def send_pose(self):
        o = np.zeros(3); self._check(self.L.gs_slam_get_send_pose(self.h, _d(o))); return o

    # ---- frame collector and output encoders (reference Slam::nextCone / initializeCollection / sendCones)
    def collect_direction(self, object_id, azimuth_deg, zenith_deg):
        return self._check(self.L.gs_slam_collect_direction(self.h, int(object_id), float(azimuth_deg), float(zenith_deg)))

    def collect_distance(self, object_id, distance):
        return self._check(self.L.gs_slam_collect_distance(self.h, int(object_id), float(distance)))

    def collect_type(self, object_id, cone_type):
        return self._check(self.L.gs_slam_collect_type(self.h, int(object_id), int(cone_type)))

    def collect_flush(self, pose):
        """Extracts the collected frame ([K,4] rows az, zen, dist, type), resets the collector, runs perform_slam on it."""
        pose = _f64(pose); k = C.c_int32(0); buf = np.zeros((1000, 4))
        self._check(self.L.gs_slam_collect_flush(self.h, _d(pose), C.byref(k), _d(buf)))
        return buf[:k.value].copy()

    # ---- odometry intake and pose output (reference Slam::nextSplitPose / nextPose / nextYawRate / sendPose)
    def set_gps_reference(self, lat, lon): self._check(self.L.gs_slam_set_gps_reference(self.h, float(lat), float(lon)))
    def next_wgs84(self, lat, lon): self._check(self.L.gs_slam_next_wgs84(self.h, float(lat), float(lon)))
    def next_heading(self, north_heading): self._check(self.L.gs_slam_next_heading(self.h, float(north_heading)))
    def next_geolocation(self, lat, lon, heading): self._check(self.L.gs_slam_next_geolocation(self.h, float(lat), float(lon), float(heading)))
    def next_yaw_rate(self, wz): self._check(self.L.gs_slam_next_yaw_rate(self.h, float(wz)))

    def set_sample_times(self, yaw_received_us, last_cone_us):
        self._check(self.L.gs_slam_set_sample_times(self.h, int(yaw_received_us), int(last_cone_us)))

    def odometry(self):
        o = np.zeros(4); self._check(self.L.gs_slam_get_odometry(self.h, _d(o))); return o

    def encode_pose(self):
        o = np.zeros(3, dtype=np.float32); self._check(self.L.gs_slam_encode_pose(self.h, o.ctypes.data_as(C.POINTER(C.c_float)))); return o

    def encode_cones(self, cones_per_packet):
        n = int(cones_per_packet); az = np.zeros(n, dtype=np.float32); di = np.zeros(n, dtype=np.float32); ty = np.zeros(n, dtype=np.int32)
        self._check(self.L.gs_slam_encode_cones(self.h, n, az.ctypes.data_as(C.POINTER(C.c_float)), di.ctypes.data_as(C.POINTER(C.c_float)), _i(ty)))
        return az, di, ty


def cone_encode(cone_xy, pose, reference_quirks=0):
    """Product implementation of Cone::getDirection / getDistance (reference src/cone.cpp:34-53): (azimuth deg, distance) as float32."""
    L = lib(); az, di = C.c_float(), C.c_float(); c = _f64(cone_xy); p = _f64(pose)
    rc = L.gs_cone_encode(_d(c), _d(p), int(reference_quirks), C.byref(az), C.byref(di))
    if rc != 0: raise GsError(rc, (L.gs_last_error() or b"").decode())
    return np.float32(az.value), np.float32(di.value)


def wgs84_to_cartesian(ref_latlon, pos_latlon):
    """Product implementation of the reference's wgs84::toCartesian (host side, no device needed)."""
    L = lib(); o = np.zeros(2); r = _f64(ref_latlon); p = _f64(pos_latlon)
    if L.gs_wgs84_to_cartesian(_d(r), _d(p), _d(o)) != 0: raise GsError("gs_wgs84_to_cartesian failed")
    return o


def wgs84_from_cartesian(ref_latlon, xy):
    L = lib(); o = np.zeros(2); r = _f64(ref_latlon); p = _f64(xy)
    if L.gs_wgs84_from_cartesian(_d(r), _d(p), _d(o)) != 0: raise GsError("gs_wgs84_from_cartesian failed")
    return o



class Shell:
    """The microservice shell (reference src/opendlv-logic-cfsd18-sensation-slam.cpp:49-119), decoded-message boundary."""
    WGS84, ANGULAR_VELOCITY, HEADING, GEOLOCATION, OBJECT_TYPE, OBJECT_DIRECTION, OBJECT_DISTANCE = 19, 1031, 1051, 1116, 1131, 1133, 1134

    def __init__(self, argv, device=-1):
        self.L = lib()
        arr = (C.c_char_p * len(argv))(*[a.encode() for a in argv])
        h = C.c_void_p()
        rc = self.L.gs_shell_create(len(argv), arr, int(device), C.byref(h))
        if rc < 0:
            raise GsError(rc, (self.L.gs_last_error() or b"").decode())
        self.h = h
        self.slam = Slam(_handle=self.L.gs_shell_slam(self.h))

    def _check(self, rc):
        if rc < 0:
            raise GsError(rc, (self.L.gs_last_error() or b"").decode())
        return rc

    def close(self):
        if getattr(self, "h", None):
            self.L.gs_shell_destroy(self.h)
            self.slam.h = None                     # the shell owned it: the borrowed handles must not outlive it
            self.slam.graph.h = None
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def on_message(self, data_type, sender_stamp, sample_time_us, now_us, object_id=0, v=(0.0, 0.0, 0.0)):
        m = ShellMsg(); m.data_type = int(data_type); m.sender_stamp = int(sender_stamp); m.sample_time_us = int(sample_time_us)
        m.object_id = int(object_id)
        for i, x in enumerate(v):
            m.v[i] = float(x)
        return self._check(self.L.gs_shell_on_message(self.h, C.byref(m), int(now_us)))

    def poll(self, now_us):
        return self._check(self.L.gs_shell_poll(self.h, int(now_us)))

    def take_output(self):
        n = self._check(self.L.gs_shell_pending_output(self.h))
        buf = (ShellMsg * max(n, 1))()
        n = self._check(self.L.gs_shell_take_output(self.h, n, buf))
        return [(m.data_type, m.sender_stamp, m.sample_time_us, m.object_id, (m.v[0], m.v[1], m.v[2])) for m in buf[:n]]

    def counters(self):
        a = (C.c_int64 * 2)(); self._check(self.L.gs_shell_counters(self.h, a)); return int(a[0]), int(a[1])

    @property
    def cid(self): return self._check(self.L.gs_shell_cid(self.h))
