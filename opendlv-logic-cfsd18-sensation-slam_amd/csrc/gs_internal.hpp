// gs_internal.hpp — the opaque handle behind gs_graph (shared by gs_api.cpp and gs_slam.cpp).
#pragma once
#include <hip/hip_runtime.h>

#include <chrono>
#include <string>
#include <vector>

#include "../../include/graphslam.h"
#include "../../include/graphslam_debug.h"
#include "gs_device.hpp"
#include "gs_host.hpp"

struct gs_graph {
    // launch parameters of one list of fronts grouped by level
    struct LevelSet { std::vector<int32_t> start; std::vector<int> max_f, max_npiv, max_nbnd; };

    gs_config cfg{};
    gs_debug_options opt{};                 // every tuning switch (graphslam_debug.h): filled once at gs_create, replaced by gs_debug_set_options
    int device = 0;
    bool host_only = false;                 // cfg.device == -2: no HIP calls, no arithmetic
    gs::HostGraph h;
    gs::Plan plan;
    std::vector<uint64_t> lm_seen_interior, lm_seen_first;        // rank-local ingestion: which windows see a landmark (gs_dist_set_landmark_windows); empty: the plan build walks all edges
    int pp_records_late = 0;                                       // odometry edges whose records a shard had to send after the plan (expected: 0)
    std::shared_ptr<void> plan_ws;          // scratch of gs::build_plan, kept between the plan builds of this handle
    uint64_t plan_version = ~0ull;          // h.structure_version the plan was built for
    gs::DevGraph d;
    // device memory of this handle: chunks the plan's small arrays are carved from, and one allocation per big array.  They SURVIVE a
    // structure phase (dev_release keeps them: on some boxes hipFree + hipMalloc of a plan's memory costs more than building the plan);
    // what the next plan does not take again is returned afterwards (dev_trim).
    struct DevChunk { void *p = nullptr; size_t size = 0; bool big = false, in_use = false; };
    std::vector<DevChunk> allocs;
    char *pool_base = nullptr; size_t pool_size = 0, pool_off = 0, pool_next = 0, pool_total = 0;     // pool_total: bytes of the chunks in use
    bool dev_valid = false;                 // device mirrors the host graph + plan
    bool dev_estimates_newer = false;       // estimates in HBM are ahead of the host copy
    uint64_t dev_estimate_version = 0;
    hipStream_t stream = nullptr; bool own_stream = false;
    hipEvent_t ev[8]{};
    hipEvent_t ev_lin[2]{};                 // timed iterations: start / stop attached to the linearisation dispatch (null: none)
    LevelSet own, shared;                   // this rank's fronts / the shared top (pose-window shards)
    int shared_base = 0;                    // offset of the shared list inside d.level_fronts
    int leaf_max_f = 0;                     // largest leaf front (<= 47: the three-tile-row leaf instance)
    int block_n = -1;                       // trailing level positions of the whole-tree factor launch that get a workgroup each (-1: not decided yet)
    int leaf_n = -1, leaf_slot = 0;             // level-0 fronts handled by the leaf instance of the factor kernel, its LDS slot (doubles per wave)
    int sub_n = 0, sub_first = 0, sub_free = 0; // bottom subtrees (k_factor3_sub): level-1 fronts [sub_first, sub_first + sub_n) each take the leaves below them; the leaf launch covers [0, sub_free)
    int32_t *d_posof = nullptr;                 // device: front -> level position
    double ms_structure = 0;
    int rank = 0, world = 1;
    double *exchange = nullptr; bool exchange_external = false;   // caller-provided exchange buffer (e.g. a torch tensor)
    void *comm = nullptr; bool own_comm = false; int comm_world = 0;   // RCCL communicator of gs_dist_iterate / gs_dist_optimize (ncclComm_t; own: created by gs_dist_comm_init)
    bool force_gather = false;              // cfg.linearize_gather
    // front end (A0 / A1): nothing is allocated or freed per call
    struct FrontEnd {
        char *arena = nullptr; size_t arena_bytes = 0;          // grow-only device scratch of the batch calls
        double *pin_in = nullptr; char *pin_out = nullptr;      // per-frame path: pinned staging, device in / out, capacity in observations
        double *dev_in = nullptr; char *dev_out = nullptr; int cap_obs = 0;
        double *map_xy = nullptr; int32_t *map_type = nullptr;  // the resident association map (mirror of Slam::m_map), grow-only
        int map_n = 0, map_cap = 0;
        char *pin_map = nullptr; size_t pin_map_bytes = 0; bool pin_map_busy = false;   // pinned staging of map appends (busy: a copy out of it may be in flight)
        char *grid_mem = nullptr; size_t grid_bytes = 0;        // the resident map's uniform grid, built on the device (gs_associate_resident): parameters, cell starts, items
        bool grid_valid = false; int grid_map_n = 0; double grid_thr = 0.0; long long grid_max_cells = 0;
        double *pcs = nullptr; size_t pcs_bytes = 0;            // cos / sin of the poses of a batched association (scratch, grow-only)
    } fe;
    int default_factor_variant = 0;         // see upload_graph
    // plans that hold a front of more than 63 scalars: table-driven whole-tree launches (workgroup -> {level position, kind | count << 8}),
    // built once per plan; *_level[l] = first table entry of level l (one launch per level after a fallback)
    struct WgSeg { int first, count, level; size_t lds; int cls; };  // a run of table entries of one level with the same LDS need and kernel class
    std::vector<WgSeg> seg_f, seg_b;
    std::vector<int32_t> wg_f, wg_b;
    int2 *d_wg_f = nullptr, *d_wg_b = nullptr; int small_max_npiv = 0, small_max_f = 0;
    std::vector<WgSeg> segs_c, segs_t, segs_b; std::vector<int32_t> wgs_c, wgs_t, wgs_b;     // the SHARED top of a sharded plan with workgroup fronts: contributions, top, backward solve
    int2 *d_wgs_c = nullptr, *d_wgs_t = nullptr, *d_wgs_b = nullptr;
    int enqueue_rc = 0; std::string enqueue_err;     // a failure of the enqueue itself (the lazily built launch tables): returned by the entry point that enqueued
    bool tree_proven = false;               // a whole-tree launch sequence of the CURRENT plan has completed without a flag timeout (gs_optimize then sends all iterations of a call at once)
    bool fell_back = false;                 // a whole-tree launch gave up on a flag: this handle uses one launch per level — until the next plan, or until a retry (below) comes back clean
    int fallback_calls = 0, fallback_retry_after = 4; bool fallback_retrying = false;   // gs_optimize calls on the slow path since the fallback; the call that tries the whole-tree launches again
    // append-only growth (gs::grow_plan): the full structure phase leaves room behind the plan's arrays; a growth step re-writes the
    // changed fronts' runs there and rebuilds those fronts' device tables.  used_* = entries taken so far, cap_* = allocated.
    struct GrowRoom { int64_t cap_bnd = 0, cap_map = 0, cap_asm = 0, cap_sc = 0, used_sc = 0, cap_L = 0, cap_U = 0, used_U = 0, cap_xe = 0; bool ok = false; } room;
    int32_t *d_bf = nullptr, *d_xrow = nullptr, *d_patch = nullptr, *d_list = nullptr;      // per-front counts / children-table offsets (kept for growth), patch + list staging
    std::vector<int32_t> bf_host, u3_off_host, u3_size_host, pos_of_front;                  // host mirrors the patch is computed from
    gs::Sc3Args sc3_args{};
    std::string no_growth_reason;           // why the last structure change was not absorbed by growing (empty: it was, or nothing tried)
};

namespace gs {
extern thread_local std::string g_last_error;
int fail(int code, const std::string &msg);
}
