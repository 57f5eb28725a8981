// gs_api.cpp — C-ABI of the GraphSLAM back-end (include/graphslam.h) over the HIP kernels.
// Host side of the drop-in boundary: everything Slam calls on g2o::SparseOptimizer
// (reference src/slam.cpp:53-65, 433-484, 525-550, 713-732) lands here.
#include "../../include/graphslam.h"
#include "../../include/graphslam_debug.h"
#include "gs_device.hpp"
#include "gs_host.hpp"
#include "gs_internal.hpp"
#include "gs_parallel.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>          // types and prototypes only: the library is resolved at run time (rccl_api), never linked

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

using namespace gs;

namespace gs {
thread_local std::string g_last_error;
int fail(int code, const std::string &msg) { g_last_error = msg; return code; }
}  // namespace gs

#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) \
    return fail(GS_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); } while (0)

// ------------------------------------------------------------------ helpers
void gs_frontend_release(gs_graph *g);       // front-end buffers of the handle (defined with the front end below)
void gs_dist_comm_release(gs_graph *g);      // the handle's own RCCL communicator (defined with the multi-GPU entry points)
static int usable_devices() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// Device memory of a plan comes out of a few large chunks (8 MB, then doubling): the ~70 arrays of one structure phase
// cost a dozen hipMalloc calls instead of 70 (each is 50-100 us of the structure phase), and dev_free_all returns them together.
// Arrays of 32 MB and more get an allocation of their own.
// gs_debug_options.pool_poison (tests): every chunk is filled with 0xFF bytes (NaN doubles, negative indices) when it is allocated and whenever a plan
// releases it — an array that is read before this code writes it cannot pass for zero-initialised
static bool pool_poison(const gs_graph *g) { return g->opt.pool_poison > 0; }
template <class T> static int dev_alloc(gs_graph *g, T **ptr, size_t count) {
    *ptr = nullptr;
    const size_t bytes = (std::max<size_t>(count, 1) * sizeof(T) + 255) & ~(size_t)255;
    if (bytes >= ((size_t)32 << 20)) {                                // a big array: its own allocation, exactly its size (a chunk rounded up to a
        int best = -1;                                                // power of two for it, or the abandoned rest of the current chunk, were 270 MB of
        for (size_t i = 0; i < g->allocs.size(); ++i) { const auto &c = g->allocs[i];      // an 800 MB footprint at 100k poses); one kept from the last plan
            if (c.big && !c.in_use && c.size >= bytes && c.size <= bytes + bytes / 4 && (best < 0 || c.size < g->allocs[best].size)) best = (int)i; }   // serves if it fits within 25 %
        if (best < 0) { void *p = nullptr;
            HIP_TRY(hipMalloc(&p, bytes));
            if (pool_poison(g)) HIP_TRY(hipMemsetAsync(p, 0xFF, bytes, g->stream));
            gs_graph::DevChunk c; c.p = p; c.size = bytes; c.big = true; g->allocs.push_back(c); best = (int)g->allocs.size() - 1; }
        g->allocs[best].in_use = true; g->pool_total += g->allocs[best].size; *ptr = (T *)g->allocs[best].p;
        return GS_OK; }
    if (g->pool_off + bytes > g->pool_size) {
        size_t want = std::max<size_t>(g->pool_next, (size_t)8 << 20);
        while (want < bytes) want <<= 1;
        int pick = -1;
        for (size_t i = 0; i < g->allocs.size() && pick < 0; ++i) { const auto &c = g->allocs[i]; if (!c.big && !c.in_use && c.size >= want) pick = (int)i; }   // a chunk of the last plan
        if (pick < 0) { void *p = nullptr;
            HIP_TRY(hipMalloc(&p, want));
            if (pool_poison(g)) HIP_TRY(hipMemsetAsync(p, 0xFF, want, g->stream));
            gs_graph::DevChunk c; c.p = p; c.size = want; g->allocs.push_back(c); pick = (int)g->allocs.size() - 1; }
        auto &c = g->allocs[pick]; c.in_use = true; g->pool_total += c.size;
        g->pool_base = (char *)c.p; g->pool_size = c.size; g->pool_off = 0;
        g->pool_next = std::min<size_t>(std::max(want, c.size) << 1, (size_t)128 << 20);      // chunks of at most 128 MB: little slack in the footprint
    }
    *ptr = (T *)(g->pool_base + g->pool_off);
    g->pool_off += bytes;
    return GS_OK;
}
template <class T, class A> static int dev_upload(gs_graph *g, T **ptr, const std::vector<T, A> &v) {
    int rc = dev_alloc(g, ptr, v.size());
    if (rc != GS_OK) return rc;
    if (!v.empty()) HIP_TRY(hipMemcpyAsync(*ptr, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, g->stream));
    return GS_OK;
}
// the device side of a plan goes away; keep = the memory stays with the handle for the next plan
static void dev_release(gs_graph *g, bool keep) {
    if (!keep) { for (auto &c : g->allocs) hipFree(c.p); g->allocs.clear(); }
    else for (auto &c : g->allocs) { c.in_use = false; if (pool_poison(g) && g->stream) hipMemsetAsync(c.p, 0xFF, c.size, g->stream); }
    g->pool_base = nullptr; g->pool_size = g->pool_off = 0; g->pool_next = 0; g->pool_total = 0;
    g->d = DevGraph();
    g->dev_valid = false;
    g->room = gs_graph::GrowRoom(); g->d_bf = g->d_xrow = g->d_patch = g->d_list = nullptr;
}
static void dev_free_all(gs_graph *g) { dev_release(g, false); }
// after a structure phase: what the new plan did not take again goes back to the device
static void dev_trim(gs_graph *g) {
    size_t idle = 0;
    for (const auto &c : g->allocs) if (!c.in_use) idle += c.size;
    if (idle <= std::max<size_t>((size_t)64 << 20, g->pool_total / 2)) return;      // a modest reserve stays (hipFree is not free either: 0.3-0.5 ms for a 32 MB chunk)
    size_t w = 0;
    for (size_t i = 0; i < g->allocs.size(); ++i) { if (g->allocs[i].in_use) g->allocs[w++] = g->allocs[i]; else hipFree(g->allocs[i].p); }
    g->allocs.resize(w);
}

static int ensure_device(gs_graph *g) {
    if (g->host_only) return fail(GS_ERR_NO_DEVICE, "host-only handle (device = -2): no compute without a gfx950 device");
    HIP_TRY(hipSetDevice(g->device));
    return GS_OK;
}

// ------------------------------------------------------------------ misc
extern "C" int gs_version(void) { return GS_VERSION_MAJOR * 100 + GS_VERSION_MINOR; }
extern "C" const char *gs_last_error(void) { return g_last_error.c_str(); }
extern "C" int gs_device_count(void) { return usable_devices(); }

extern "C" int gs_config_default(gs_config *c) {
    if (!c) return fail(GS_ERR_INVALID, "null config");
    std::memset(c, 0, sizeof(*c));
    c->struct_size = (int32_t)sizeof(gs_config);
    c->device = -1; c->verbose = 0; c->leaf_poses = 0; c->factor_variant = 0; c->linearize_gather = 0;
    c->odometry_information = 5.0;       // reference src/slam.cpp:456
    c->cone_information = 0.01;          // reference src/slam.cpp:546
    c->same_cone_threshold = 1.0;        // m_newConeThreshold default, reference src/slam.hpp:114
    c->cone_mapping_threshold = 67.0;    // reference src/slam.hpp:117
    c->lidar_to_cog = 1.5;               // reference src/slam.cpp:514
    c->loop_closing_radius = 1.0;        // reference src/slam.cpp:702
    c->loop_closing_min_index = 20;      // reference src/slam.cpp:702
    c->optimize_iterations = 10;         // reference src/slam.cpp:481
    c->reference_quirks = 0;
    return GS_OK;
}


// ------------------------------------------------------------------ tuning switches (include/graphslam_debug.h)
extern "C" int gs_debug_options_default(gs_debug_options *o) {
    if (!o) return fail(GS_ERR_INVALID, "null options");
    std::memset(o, 0, sizeof(*o));
    o->struct_size = (int32_t)sizeof(*o);
    o->subtree = 0; o->tickets = 0; o->shard_by_window = 1;
    o->tree = 1; o->block_fronts = 512; o->leaf_kernel = -1; o->leaf_min = 2048; o->bs_wide = 2048; o->leaf_nt3 = 1; o->f3_lds_kb = 0;
    o->leaf_poses = 0; o->cluster_ways = 0; o->ell_lanes = 0; o->big_cluster = -1; o->grow_headroom = -1; o->factor_variant = 0;
    o->grow = 1; o->grow_min_poses = 128;
    o->assoc_grid = -1;
    o->force_shared_top = 0;
    o->host_trig = 0; o->pool_poison = 0; o->plan_timing = 0; o->dbg = 0;
    return GS_OK;
}
// The ONE place the environment is read: once per gs_create (graphslam_debug.h names the variable of every field).
static void options_from_environment(gs_debug_options &o) {
    gs_debug_options_default(&o);
    auto env = [](const char *name, int32_t &field) { if (const char *e = std::getenv(name)) field = (int32_t)std::atoi(e); };
    env("GS_TREE", o.tree); env("GS_BLOCK_FRONTS", o.block_fronts); env("GS_LEAF_KERNEL", o.leaf_kernel); env("GS_LEAF_MIN", o.leaf_min);
    env("GS_SUBTREE", o.subtree); env("GS_TICKETS", o.tickets); env("GS_SHARD_BY_WINDOW", o.shard_by_window); env("GS_BS_WIDE", o.bs_wide); env("GS_LEAF_NT3", o.leaf_nt3); env("GS_F3_LDS_KB", o.f3_lds_kb);
    env("GS_LEAF_POSES", o.leaf_poses); env("GS_CLUSTER_WAYS", o.cluster_ways); env("GS_ELL_LANES", o.ell_lanes); env("GS_BIG_CLUSTER", o.big_cluster);
    env("GS_GROW_HEADROOM", o.grow_headroom); env("GS_FACTOR_VARIANT", o.factor_variant);
    env("GS_GROW", o.grow); env("GS_GROW_MIN_POSES", o.grow_min_poses); env("GS_ASSOC_GRID", o.assoc_grid); env("GS_FORCE_SHARED_TOP", o.force_shared_top);
    env("GS_HOST_TRIG", o.host_trig); env("GS_POOL_POISON", o.pool_poison); env("GS_DBG", o.dbg);
    if (std::getenv("GS_PLAN_TIMING")) o.plan_timing = 1;
}
extern "C" int gs_debug_get_options(gs_graph *g, gs_debug_options *o) {
    if (!g || !o) return fail(GS_ERR_INVALID, "null argument");
    *o = g->opt; return GS_OK;
}
extern "C" int gs_debug_set_options(gs_graph *g, const gs_debug_options *o) {
    if (!g || !o) return fail(GS_ERR_INVALID, "null argument");
    gs_debug_options n; gs_debug_options_default(&n);
    std::memcpy(&n, o, std::min<size_t>(sizeof(n), (size_t)std::max(o->struct_size, 0))); n.struct_size = (int32_t)sizeof(n);
    const gs_debug_options &c = g->opt;
    // a "plan" field changed: the next structure phase is a full one (a grown plan keeps the launch shapes it was built with)
    const bool plan_changed = n.tree != c.tree || n.block_fronts != c.block_fronts || n.leaf_kernel != c.leaf_kernel || n.leaf_min != c.leaf_min ||
        n.bs_wide != c.bs_wide || n.subtree != c.subtree || n.tickets != c.tickets || n.shard_by_window != c.shard_by_window || n.leaf_nt3 != c.leaf_nt3 || n.f3_lds_kb != c.f3_lds_kb || n.leaf_poses != c.leaf_poses ||
        n.cluster_ways != c.cluster_ways || n.ell_lanes != c.ell_lanes || n.big_cluster != c.big_cluster || n.grow_headroom != c.grow_headroom ||
        n.factor_variant != c.factor_variant || n.force_shared_top != c.force_shared_top || n.host_trig != c.host_trig || n.pool_poison != c.pool_poison ||
        n.dbg != c.dbg;
    g->opt = n;
    if (plan_changed) { ++g->h.structure_version; ++g->h.reshape_version; }
    return GS_OK;
}

extern "C" int gs_create(const gs_config *cfg, gs_graph **out) {
    if (!out) return fail(GS_ERR_INVALID, "null out");
    *out = nullptr;
    gs_config c;
    gs_config_default(&c);
    if (cfg) { size_t n = std::min<size_t>(sizeof(c), (size_t)std::max(cfg->struct_size, 0)); std::memcpy(&c, cfg, n); c.struct_size = (int32_t)sizeof(c); }
    if (c.device == -2) {   // host-only handle: graph container + plan inspection, never any arithmetic
        gs_graph *g = new gs_graph(); g->cfg = c; g->device = -2; g->host_only = true; options_from_environment(g->opt); *out = g; return GS_OK; }
    int ndev = usable_devices();
    if (ndev <= 0) return fail(GS_ERR_NO_DEVICE, "no HIP device: this back-end has no CPU fallback");
    int dev = c.device;
    if (dev < 0) { if (hipGetDevice(&dev) != hipSuccess) dev = 0; }
    if (dev >= ndev) return fail(GS_ERR_NO_DEVICE, "device ordinal out of range");
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, dev));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(GS_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", kernels are built for gfx950 only");
    gs_graph *g = new gs_graph();
    g->cfg = c; g->device = dev; g->force_gather = c.linearize_gather != 0; g->default_factor_variant = c.factor_variant;
    options_from_environment(g->opt);
    HIP_TRY(hipSetDevice(dev));
    if (hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking) != hipSuccess) { delete g; return fail(GS_ERR_HIP, "hipStreamCreate failed"); }
    g->own_stream = true;
    for (auto &e : g->ev) hipEventCreate(&e);
    *out = g;
    return GS_OK;
}

extern "C" int gs_destroy(gs_graph *g) {
    if (!g) return GS_OK;
    if (g->host_only) { delete g; return GS_OK; }
    hipSetDevice(g->device);
    hipStreamSynchronize(g->stream);
    dev_free_all(g);
    gs_dist_comm_release(g);
    gs_frontend_release(g);
    for (auto &e : g->ev) hipEventDestroy(e);
    if (g->own_stream) hipStreamDestroy(g->stream);
    delete g;
    return GS_OK;
}

extern "C" int gs_clear(gs_graph *g) {
    if (!g) return fail(GS_ERR_INVALID, "null graph");
    if (!g->host_only) { hipSetDevice(g->device); hipStreamSynchronize(g->stream); dev_free_all(g); }
    g->h.clear(); g->plan = Plan(); g->plan_version = ~0ull;
    return GS_OK;
}

extern "C" int gs_reserve_device(gs_graph *g, int64_t bytes) {
    if (!g) return fail(GS_ERR_INVALID, "null graph");
    int rc = ensure_device(g); if (rc != GS_OK) return rc;
    int64_t have = 0;
    for (const auto &c : g->allocs) if (!c.big && !c.in_use) have += (int64_t)c.size;
    for (size_t want = (size_t)8 << 20; have < bytes; want = std::min<size_t>(want << 1, (size_t)128 << 20)) {     // the sizes dev_alloc asks for, in its order
        bool held = false;
        for (const auto &c : g->allocs) held = held || (!c.big && !c.in_use && c.size == want);
        if (held && want < ((size_t)128 << 20)) continue;
        void *p = nullptr;
        HIP_TRY(hipMalloc(&p, want));
        HIP_TRY(hipMemsetAsync(p, 0, want, g->stream));              // touch it now: the mapping work of a fresh allocation otherwise lands on the first launch that follows
        gs_graph::DevChunk c; c.p = p; c.size = want; g->allocs.push_back(c); have += (int64_t)want; }
    HIP_TRY(hipStreamSynchronize(g->stream));
    return GS_OK;
}
extern "C" int gs_set_stream(gs_graph *g, void *s) {
    if (!g) return fail(GS_ERR_INVALID, "null graph");
    { int rc = ensure_device(g); if (rc != GS_OK) return rc; }
    hipStreamSynchronize(g->stream);
    if (g->own_stream) { hipStreamDestroy(g->stream); g->own_stream = false; }
    if (s) g->stream = (hipStream_t)s;
    else { HIP_TRY(hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking)); g->own_stream = true; }
    return GS_OK;
}

// ------------------------------------------------------------------ construction (A2)
static int pull_estimates_if_needed(gs_graph *g);

extern "C" int gs_add_pose(gs_graph *g, int32_t id, const double est[3]) {
    if (!g || !est) return fail(GS_ERR_INVALID, "null argument");
    if (g->h.pose_index.count(id)) return fail(GS_ERR_DUPLICATE_ID, "pose id already present");
    int rc = pull_estimates_if_needed(g); if (rc != GS_OK) return rc;
    g->h.pose_index[id] = g->h.n_poses();
    g->h.pose_id.push_back(id); g->h.pose_est.insert(g->h.pose_est.end(), est, est + 3); g->h.pose_fixed.push_back(0);
    ++g->h.structure_version;
    return GS_OK;
}
extern "C" int gs_add_landmark(gs_graph *g, int32_t id, const double est[2]) {
    if (!g || !est) return fail(GS_ERR_INVALID, "null argument");
    if (g->h.lm_index.count(id)) return fail(GS_ERR_DUPLICATE_ID, "landmark id already present");
    int rc = pull_estimates_if_needed(g); if (rc != GS_OK) return rc;
    g->h.lm_index[id] = g->h.n_lms();
    g->h.lm_id.push_back(id); g->h.lm_est.insert(g->h.lm_est.end(), est, est + 2); g->h.lm_fixed.push_back(0);
    ++g->h.structure_version;
    return GS_OK;
}
static bool sym_ok(const double *m, int n) {
    for (int r = 0; r < n; ++r) for (int c = 0; c < r; ++c) {
        double a = m[r * n + c], b = m[c * n + r];
        if (!(std::fabs(a - b) <= 1e-12 * (std::fabs(a) + std::fabs(b)) + 1e-300)) return false;
    }
    return true;
}
extern "C" int gs_add_odometry_edge(gs_graph *g, int32_t idi, int32_t idj, const double z[3], const double info[9]) {
    if (!g || !z || !info) return fail(GS_ERR_INVALID, "null argument");
    auto a = g->h.pose_index.find(idi), b = g->h.pose_index.find(idj);
    if (a == g->h.pose_index.end() || b == g->h.pose_index.end()) return fail(GS_ERR_UNKNOWN_ID, "odometry edge references an unknown pose");
    if (a->second == b->second) return fail(GS_ERR_INVALID, "odometry edge joins a pose to itself");
    if (!sym_ok(info, 3)) return fail(GS_ERR_INVALID, "information matrix not symmetric");
    g->h.pp_i.push_back(a->second); g->h.pp_j.push_back(b->second);
    g->h.pp_z.insert(g->h.pp_z.end(), z, z + 3);
    const double s[6] = {info[0], info[1], info[2], info[4], info[5], info[8]};
    g->h.pp_info.insert(g->h.pp_info.end(), s, s + 6);
    ++g->h.structure_version;
    return GS_OK;
}
extern "C" int gs_add_observation_edge(gs_graph *g, int32_t idp, int32_t idl, const double z[2], const double info[4]) {
    if (!g || !z || !info) return fail(GS_ERR_INVALID, "null argument");
    auto a = g->h.pose_index.find(idp); auto b = g->h.lm_index.find(idl);
    if (a == g->h.pose_index.end() || b == g->h.lm_index.end()) return fail(GS_ERR_UNKNOWN_ID, "observation edge references an unknown vertex");
    if (!sym_ok(info, 2)) return fail(GS_ERR_INVALID, "information matrix not symmetric");
    g->h.pl_p.push_back(a->second); g->h.pl_l.push_back(b->second);
    g->h.pl_z.insert(g->h.pl_z.end(), z, z + 2);
    const double s[3] = {info[0], info[1], info[3]};
    g->h.pl_info.insert(g->h.pl_info.end(), s, s + 3);
    ++g->h.structure_version;
    return GS_OK;
}
extern "C" int gs_add_poses(gs_graph *g, int32_t n, const int32_t *ids, const double *est) {
    if (!g || (n > 0 && (!ids || !est))) return fail(GS_ERR_INVALID, "null argument");
    for (int k = 0; k < n; ++k) { int rc = gs_add_pose(g, ids[k], est + 3 * (size_t)k); if (rc != GS_OK) return rc; }
    return GS_OK;
}
extern "C" int gs_add_landmarks(gs_graph *g, int32_t n, const int32_t *ids, const double *est) {
    if (!g || (n > 0 && (!ids || !est))) return fail(GS_ERR_INVALID, "null argument");
    for (int k = 0; k < n; ++k) { int rc = gs_add_landmark(g, ids[k], est + 2 * (size_t)k); if (rc != GS_OK) return rc; }
    return GS_OK;
}
extern "C" int gs_add_odometry_edges(gs_graph *g, int32_t n, const int32_t *idi, const int32_t *idj, const double *z, const double *info) {
    if (!g || (n > 0 && (!idi || !idj || !z))) return fail(GS_ERR_INVALID, "null argument");
    const double w = g->cfg.odometry_information;
    const double def[9] = {w, 0, 0, 0, w, 0, 0, 0, w};
    for (int k = 0; k < n; ++k) { int rc = gs_add_odometry_edge(g, idi[k], idj[k], z + 3 * (size_t)k, info ? info + 9 * (size_t)k : def); if (rc != GS_OK) return rc; }
    return GS_OK;
}
extern "C" int gs_add_observation_edges(gs_graph *g, int32_t n, const int32_t *idp, const int32_t *idl, const double *z, const double *info) {
    if (!g || (n > 0 && (!idp || !idl || !z))) return fail(GS_ERR_INVALID, "null argument");
    const double w = g->cfg.cone_information;
    const double def[4] = {w, 0, 0, w};
    for (int k = 0; k < n; ++k) { int rc = gs_add_observation_edge(g, idp[k], idl[k], z + 2 * (size_t)k, info ? info + 4 * (size_t)k : def); if (rc != GS_OK) return rc; }
    return GS_OK;
}
extern "C" int gs_set_fixed_pose(gs_graph *g, int32_t id, int32_t fixed) {
    if (!g) return fail(GS_ERR_INVALID, "null graph");
    auto a = g->h.pose_index.find(id);
    if (a == g->h.pose_index.end()) return fail(GS_ERR_UNKNOWN_ID, "unknown pose id");
    uint8_t f = fixed != 0;
    if (g->h.pose_fixed[a->second] != f) { g->h.pose_fixed[a->second] = f; ++g->h.structure_version; ++g->h.reshape_version; }
    return GS_OK;
}
extern "C" int gs_set_fixed_landmark(gs_graph *g, int32_t id, int32_t fixed) {
    if (!g) return fail(GS_ERR_INVALID, "null graph");
    auto a = g->h.lm_index.find(id);
    if (a == g->h.lm_index.end()) return fail(GS_ERR_UNKNOWN_ID, "unknown landmark id");
    uint8_t f = fixed != 0;
    if (g->h.lm_fixed[a->second] != f) { g->h.lm_fixed[a->second] = f; ++g->h.structure_version; ++g->h.reshape_version; }
    return GS_OK;
}
extern "C" int gs_set_pose_estimate(gs_graph *g, int32_t id, const double est[3]) {
    if (!g || !est) return fail(GS_ERR_INVALID, "null argument");
    auto a = g->h.pose_index.find(id);
    if (a == g->h.pose_index.end()) return fail(GS_ERR_UNKNOWN_ID, "unknown pose id");
    int rc = pull_estimates_if_needed(g); if (rc != GS_OK) return rc;
    std::memcpy(&g->h.pose_est[3 * (size_t)a->second], est, 3 * sizeof(double));
    ++g->h.estimate_version;
    return GS_OK;
}
extern "C" int gs_set_landmark_estimate(gs_graph *g, int32_t id, const double est[2]) {
    if (!g || !est) return fail(GS_ERR_INVALID, "null argument");
    auto a = g->h.lm_index.find(id);
    if (a == g->h.lm_index.end()) return fail(GS_ERR_UNKNOWN_ID, "unknown landmark id");
    int rc = pull_estimates_if_needed(g); if (rc != GS_OK) return rc;
    std::memcpy(&g->h.lm_est[2 * (size_t)a->second], est, 2 * sizeof(double));
    ++g->h.estimate_version;
    return GS_OK;
}

// ------------------------------------------------------------------ read-back (A11)
static int pull_estimates_if_needed(gs_graph *g) {
    if (!g->dev_valid || !g->dev_estimates_newer) return GS_OK;
    int rc = ensure_device(g); if (rc != GS_OK) return rc;
    const int N = g->d.N + g->d.tN, M = g->d.M + g->d.tM;          // (tail vertices of a grown plan follow the base ones in the same arrays)
    if (N > 0) HIP_TRY(hipMemcpyAsync(g->h.pose_est.data(), g->d.pose_est, (size_t)N * 3 * sizeof(double), hipMemcpyDeviceToHost, g->stream));
    if (M > 0) HIP_TRY(hipMemcpyAsync(g->h.lm_est.data(), g->d.lm_est, (size_t)M * 2 * sizeof(double), hipMemcpyDeviceToHost, g->stream));
    HIP_TRY(hipStreamSynchronize(g->stream));
    g->dev_estimates_newer = false;
    return GS_OK;
}
// The device-side failure state: fail[0] = code (1 zero pivot, 2 a whole-tree launch gave up on a front's flag, 3 a
// zero pivot another rank reported, 4 a flag timeout another rank reported), fail[1] = updates applied since the last reset.  k_update applies nothing once the
// code is non-zero, so the estimates in HBM are those of the last good iterate, as in g2o after a failed solve.
static int read_failure(gs_graph *g, int32_t out[2]) {
    out[0] = out[1] = 0;
    if (!g->dev_valid || !g->d.fail) return GS_OK;
    HIP_TRY(hipMemcpyAsync(out, g->d.fail, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, g->stream));
    HIP_TRY(hipStreamSynchronize(g->stream));
    return GS_OK;
}
static int reset_failure(gs_graph *g) {
    g->d.inject_iter = 0; g->d.inject_code = 0;
    HIP_TRY(hipMemsetAsync(g->d.fail, 0, 4 * sizeof(int32_t), g->stream));
    // the ticket counter and its host-side running sum start again together (a launch that failed to ENQUEUE was counted on the host only)
    if (g->d.tickets) { HIP_TRY(hipMemsetAsync(g->d.tickets, 0, 2 * sizeof(uint32_t), g->stream)); g->d.ticket_base = 0; }
    return GS_OK;
}
// After gs_iterate / gs_dist_iterate_*: report a failure of the iterations run since the last report (once), apply the
// one-launch-per-level fallback after a flag timeout.  The estimates stay at the last good iterate either way.
static int surface_failure(gs_graph *g) {
    int32_t st[2]; int rc = read_failure(g, st); if (rc != GS_OK) return rc;
    if (st[0] == 0) return GS_OK;
    rc = reset_failure(g); if (rc != GS_OK) return rc;
    if (st[0] == 2) { g->d.tree = 0; g->fell_back = true;
        return fail(GS_ERR_TIMEOUT, "whole-tree launch: a front's completion flag did not arrive in time; no update was applied from that "
                                    "iteration on (estimates = last good iterate); the handle now uses one launch per level"); }
    if (st[0] == 4) return fail(GS_ERR_TIMEOUT, "a whole-tree launch of ANOTHER rank gave up on a front's flag (that rank now uses one launch per level); no update was applied from that "
                                                "iteration on (estimates = last good iterate): the iteration can be run again");
    return fail(GS_ERR_NUMERIC, st[0] == 3 ? "another rank met a zero pivot: no update applied from that iteration on (estimates = last good iterate)"
                                           : "zero pivot: H is singular; no update applied from that iteration on (estimates = last good iterate)");
}
extern "C" int gs_sync_estimates(gs_graph *g) {
    if (!g) return fail(GS_ERR_INVALID, "null graph");
    int rc = pull_estimates_if_needed(g); if (rc != GS_OK) return rc;
    if (g->host_only || !g->dev_valid) return GS_OK;
    return surface_failure(g);
}
extern "C" int gs_stream_synchronize(gs_graph *g) {
    if (!g) return fail(GS_ERR_INVALID, "null graph");
    int rc = ensure_device(g); if (rc != GS_OK) return rc;
    HIP_TRY(hipStreamSynchronize(g->stream));
    return surface_failure(g);
}
extern "C" int gs_debug_fail_at_iteration(gs_graph *g, int32_t k, int32_t code) {
    if (!g || k < 0 || (code != 1 && code != 2)) return fail(GS_ERR_INVALID, "bad argument");
    if (!g->dev_valid) return fail(GS_ERR_NOT_INITIALIZED, "call gs_initialize_optimization first");
    g->d.inject_iter = k > 0 ? g->d.iter + k : 0; g->d.inject_code = code;
    return GS_OK;
}
extern "C" int gs_get_pose(gs_graph *g, int32_t id, double out[3]) {
    if (!g || !out) return fail(GS_ERR_INVALID, "null argument");
    auto a = g->h.pose_index.find(id);
    if (a == g->h.pose_index.end()) return fail(GS_ERR_UNKNOWN_ID, "unknown pose id");
    int rc = pull_estimates_if_needed(g); if (rc != GS_OK) return rc;
    std::memcpy(out, &g->h.pose_est[3 * (size_t)a->second], 3 * sizeof(double));
    return GS_OK;
}
extern "C" int gs_get_landmark(gs_graph *g, int32_t id, double out[2]) {
    if (!g || !out) return fail(GS_ERR_INVALID, "null argument");
    auto a = g->h.lm_index.find(id);
    if (a == g->h.lm_index.end()) return fail(GS_ERR_UNKNOWN_ID, "unknown landmark id");
    int rc = pull_estimates_if_needed(g); if (rc != GS_OK) return rc;
    std::memcpy(out, &g->h.lm_est[2 * (size_t)a->second], 2 * sizeof(double));
    return GS_OK;
}
extern "C" int gs_num_poses(gs_graph *g) { return g ? g->h.n_poses() : fail(GS_ERR_INVALID, "null graph"); }
extern "C" int gs_num_landmarks(gs_graph *g) { return g ? g->h.n_lms() : fail(GS_ERR_INVALID, "null graph"); }
extern "C" int gs_num_odometry_edges(gs_graph *g) { return g ? g->h.n_pp() : fail(GS_ERR_INVALID, "null graph"); }
extern "C" int gs_num_observation_edges(gs_graph *g) { return g ? g->h.n_pl() : fail(GS_ERR_INVALID, "null graph"); }
extern "C" int gs_get_poses(gs_graph *g, int32_t cap, int32_t *ids, double *out) {
    if (!g || !out) return fail(GS_ERR_INVALID, "null argument");
    if (cap < g->h.n_poses()) return fail(GS_ERR_CAPACITY, "buffer too small");
    int rc = pull_estimates_if_needed(g); if (rc != GS_OK) return rc;
    if (ids) std::memcpy(ids, g->h.pose_id.data(), g->h.pose_id.size() * sizeof(int32_t));
    std::memcpy(out, g->h.pose_est.data(), g->h.pose_est.size() * sizeof(double));
    return g->h.n_poses();
}
extern "C" int gs_get_landmarks(gs_graph *g, int32_t cap, int32_t *ids, double *out) {
    if (!g || !out) return fail(GS_ERR_INVALID, "null argument");
    if (cap < g->h.n_lms()) return fail(GS_ERR_CAPACITY, "buffer too small");
    int rc = pull_estimates_if_needed(g); if (rc != GS_OK) return rc;
    if (ids) std::memcpy(ids, g->h.lm_id.data(), g->h.lm_id.size() * sizeof(int32_t));
    std::memcpy(out, g->h.lm_est.data(), g->h.lm_est.size() * sizeof(double));
    return g->h.n_lms();
}

// ------------------------------------------------------------------ structure phase (A3/A4) + upload
static void se2_inverse_host(const double *a, double *out) {
    auto norm = [](double th) { if (th >= -M_PI && th < M_PI) return th; double m = std::floor(th / (2 * M_PI)); th -= m * 2 * M_PI;
                                if (th >= M_PI) th -= 2 * M_PI; if (th < -M_PI) th += 2 * M_PI; return th; };
    double th = norm(-a[2]); double c = std::cos(th), s = std::sin(th);
    out[0] = c * (-a[0]) - s * (-a[1]); out[1] = s * (-a[0]) + c * (-a[1]); out[2] = th;
}

// Everything that does not depend on the plan goes to HBM on a helper thread WHILE the host builds the plan: estimates,
// fixed flags, odometry measurements (inverted, with their cos/sin: g2o keeps _inverseMeasurement) and information,
// and the observation edges as inserted (permuted into the ELL layout on the device afterwards).
struct RawUpload {
    std::thread th; int rc = GS_OK; std::string err;
    ~RawUpload() { if (th.joinable()) th.join(); }                  // an exception (bad_alloc in the plan build) must not meet a joinable thread: std::terminate
    int32_t *pl_l = nullptr; double *pl_z = nullptr, *pl_info = nullptr;
    std::vector<double> zinv; size_t pp_lo = 0, pp_hi = 0;         // the odometry edges whose records went up: [pp_lo, pp_hi) (all of them on a single GPU)
    uvec<int32_t> ell_l; uvec<double> ell_z, ell_w;               // pose-window shards: the ELL streams, filled on the host (they must outlive the copies: upload_graph ends with a sync)
};
static int upload_raw_begin(gs_graph *g, RawUpload &R) {
    const HostGraph &h = g->h; DevGraph &d = g->d;
    const size_t N = h.n_poses(), M = h.n_lms(), Epp = h.n_pp(), Epl = h.n_pl();
    int rc;
    // room for the tail of a grown plan (gs::grow_plan) behind the per-pose and per-odometry-edge arrays
    const size_t TP = TAIL_POSES, TPP = TAIL_PP;
    const size_t TL = TAIL_LMS;
    if ((rc = dev_alloc(g, &d.pose_est, (N + TP) * 3)) != GS_OK || (rc = dev_alloc(g, &d.lm_est, (M + TL) * 2)) != GS_OK ||
        (rc = dev_alloc(g, &d.pose_fixed, N + TP)) != GS_OK || (rc = dev_alloc(g, &d.lm_fixed, M + TL)) != GS_OK ||
        (rc = dev_alloc(g, &d.pose_cs, (N + TP) * 2)) != GS_OK || (rc = dev_alloc(g, &d.pp_zinv, (Epp + TPP) * 5)) != GS_OK ||
        (rc = dev_alloc(g, &d.pp_info, (Epp + TPP) * 6)) != GS_OK) return rc;
    HIP_TRY(hipMemsetAsync(d.pose_fixed + N, 0, TP, g->stream)); HIP_TRY(hipMemsetAsync(d.lm_fixed + M, 0, TL, g->stream));
    // the observation edges as inserted travel now only on a single GPU; a pose-window shard uploads the ones it evaluates, in
    // device layout, once the plan says which they are (upload_graph)
    const bool raw_pl = g->world <= 1;
    if (raw_pl && ((rc = dev_alloc(g, &R.pl_l, Epl)) != GS_OK || (rc = dev_alloc(g, &R.pl_z, Epl * 2)) != GS_OK || (rc = dev_alloc(g, &R.pl_info, Epl * 3)) != GS_OK)) return rc;
    R.th = std::thread([g, &R, N, M, Epp, Epl, raw_pl] {
        const HostGraph &h = g->h; DevGraph &d = g->d;
        auto cp = [&](void *dst, const void *src, size_t bytes) {
            if (R.rc != GS_OK || bytes == 0) return;
            hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, g->stream);
            if (e != hipSuccess) { R.rc = GS_ERR_HIP; R.err = std::string("raw upload: ") + hipGetErrorString(e); } };
        if (hipSetDevice(g->device) != hipSuccess) { R.rc = GS_ERR_HIP; R.err = "hipSetDevice failed on the upload thread"; return; }
        cp(d.pose_est, h.pose_est.data(), N * 3 * sizeof(double)); cp(d.lm_est, h.lm_est.data(), M * 2 * sizeof(double));
        cp(d.pose_fixed, h.pose_fixed.data(), N); cp(d.lm_fixed, h.lm_fixed.data(), M);
        if (raw_pl) { cp(R.pl_l, h.pl_l.data(), Epl * sizeof(int32_t)); cp(R.pl_z, h.pl_z.data(), Epl * 2 * sizeof(double));
            cp(R.pl_info, h.pl_info.data(), Epl * 3 * sizeof(double)); }
        // odometry edges keep their insertion order on the device.  A pose-window shard evaluates an odometry edge only if one of its poses lies in
        // the shard's window (gs_plan.cpp, rank_of_pp: the owner of an interior endpoint, else the window of the pose; an edge between two fixed
        // poses is rank 0's): the records of the first to the last such edge go up — an eighth of 0.8 M inverses, cosines, sines and of 70 MB at
        // world 8 (this thread took longer than the plan build).  upload_graph checks the plan's assignment against the range and sends what is missing.
        size_t k0 = 0, k1 = Epp;
        if (g->world > 1 && Epp > 0) {
            size_t nfree = 0; for (size_t p = 0; p < N; ++p) nfree += !h.pose_fixed[p];
            const size_t W = (size_t)g->world, r = (size_t)g->rank, f_lo = (r * nfree + W - 1) / W, f_hi = ((r + 1) * nfree + W - 1) / W;
            size_t p_lo = N, p_hi = N, f = 0;                        // insertion indices of the window's first free pose and of the next window's
            for (size_t p = 0; p < N; ++p) if (!h.pose_fixed[p]) { if (f == f_lo) p_lo = p; if (f == f_hi) { p_hi = p; break; } ++f; }
            auto in = [&](int32_t p) { return (size_t)p >= p_lo && (size_t)p < p_hi; };
            k0 = Epp; k1 = 0;
            for (size_t k = 0; k < Epp; ++k) { const int32_t i = h.pp_i[k], j = h.pp_j[k];
                if (in(i) || in(j) || (r == 0 && h.pose_fixed[i] && h.pose_fixed[j])) { k0 = std::min(k0, k); k1 = std::max(k1, k + 1); } }
            if (k1 <= k0) k0 = k1 = 0; }
        R.pp_lo = k0; R.pp_hi = k1;
        cp(d.pp_info + 6 * k0, h.pp_info.data() + 6 * k0, (k1 - k0) * 6 * sizeof(double));
        R.zinv.resize((k1 - k0) * 5);
        for (size_t k = k0; k < k1; ++k) { double inv[3]; se2_inverse_host(&h.pp_z[3 * k], inv);
            double *o = &R.zinv[5 * (k - k0)]; o[0] = inv[0]; o[1] = inv[1]; o[2] = inv[2]; o[3] = std::cos(inv[2]); o[4] = std::sin(inv[2]); }
        cp(d.pp_zinv + 5 * k0, R.zinv.data(), (k1 - k0) * 5 * sizeof(double));
    });
    return GS_OK;
}

static int upload_graph(gs_graph *g, RawUpload &raw) {
    const HostGraph &h = g->h; const Plan &P = g->plan; DevGraph &d = g->d;
    const bool ut_on = g->opt.plan_timing > 0; auto ut_prev = std::chrono::steady_clock::now();
#define GS_UT(name) do { if (ut_on) { auto n_ = std::chrono::steady_clock::now(); std::fprintf(stderr, "upload %-18s %.2f ms\n", (name), std::chrono::duration<double, std::milli>(n_ - ut_prev).count()); ut_prev = n_; } } while (0)
    const int N = h.n_poses(), M = h.n_lms(), Epp = h.n_pp(), Epl = h.n_pl();
    d.N = N; d.M = M; d.Epp = Epp; d.Epl = Epl; d.n_scalar = P.n_scalar;
    g->leaf_n = -1; g->block_n = -1;
    int rc;
#define UP(dst, vec) if ((rc = dev_upload(g, &d.dst, vec)) != GS_OK) return rc
    // estimates, fixed flags, odometry edges and the insertion-order observation arrays are in HBM already (RawUpload)
    launch_pose_trig(d, g->stream);
    // gs_debug_options.host_trig (an experiment, scripts/parity_spread.py): the cos / sin of the INITIAL pose angles from the host's libm
    // instead of the device's — what the CPU oracle linearises with — to tell how much of the first increment's distance
    // to the CPU paths is the last bit of two transcendental functions
    if (g->opt.host_trig > 0 && N > 0) {
        std::vector<double> cs(2 * (size_t)N);
        for (int p = 0; p < N; ++p) { cs[2 * (size_t)p] = std::cos(h.pose_est[3 * (size_t)p + 2]); cs[2 * (size_t)p + 1] = std::sin(h.pose_est[3 * (size_t)p + 2]); }
        HIP_TRY(hipMemcpyAsync(d.pose_cs, cs.data(), cs.size() * sizeof(double), hipMemcpyHostToDevice, g->stream));
        HIP_TRY(hipStreamSynchronize(g->stream)); }
    g->room = gs_graph::GrowRoom(); d.tN = d.tM = d.tEpp = d.tEpl = d.tLt = 0; d.tcapN = TAIL_POSES; d.tcapM = TAIL_LMS; d.tcapEpp = TAIL_PP; d.tcapEpl = TAIL_PL;
    if ((rc = dev_alloc(g, &d.pose_gidx, (size_t)N + TAIL_POSES)) != GS_OK) return rc;              // (room for a grown plan's tail poses)
    if (N > 0) HIP_TRY(hipMemcpyAsync(d.pose_gidx, P.pose_gidx.data(), (size_t)N * sizeof(int32_t), hipMemcpyHostToDevice, g->stream));
    if ((rc = dev_alloc(g, &d.lm_gidx, (size_t)M + TAIL_LMS)) != GS_OK) return rc;
    if (M > 0) HIP_TRY(hipMemcpyAsync(d.lm_gidx, P.lm_gidx.data(), (size_t)M * sizeof(int32_t), hipMemcpyHostToDevice, g->stream));
    d.ell_T = P.ell_T; d.ell_R = P.ell_R; d.ell_len = P.ell_len; d.ell_p0 = P.ell_p0; d.ell_np = P.ell_np;
    { const size_t L = (size_t)P.ell_len;
      if ((rc = dev_alloc(g, &d.ell_l, L)) != GS_OK || (rc = dev_alloc(g, &d.ell_z, 2 * L)) != GS_OK || (rc = dev_alloc(g, &d.ell_w, 3 * L)) != GS_OK) return rc;
      if (P.world <= 1) {                                            // ELL streams: permuted on the device out of the arrays that travelled during the plan build (k_build_ell)
          int32_t *ins = nullptr;
          if ((rc = dev_upload(g, &ins, P.ell_ins)) != GS_OK) return rc;
          launch_build_ell((int64_t)L, ins, raw.pl_l, raw.pl_z, raw.pl_info, nullptr, P.rank, d.ell_l, d.ell_z, d.ell_w, g->stream);
      } else {                                                       // pose-window shard: only the poses it sweeps are laid out; the streams are filled on the host
          // ... on a thread of its own, beside the rest of this function (nothing here reads the streams; joined before the final wait): the fill
          // and three copies out of pageable memory were 2.5-5 of a rank's ~8 ms of upload at 8 x 100k poses
          raw.ell_l.resize(L); raw.ell_z.resize(2 * L); raw.ell_w.resize(3 * L);        // (threads) with the edges this rank evaluates, the others stay empty (l = -1)
          raw.th = std::thread([g, &raw, L] { const HostGraph &h = g->h; const Plan &P = g->plan; DevGraph &d = g->d;
              if (hipSetDevice(g->device) != hipSuccess) { raw.rc = GS_ERR_HIP; raw.err = "hipSetDevice failed on the upload thread"; return; }
              parallel_chunks((int64_t)L, 16384, [&](int64_t b, int64_t e2, int) {
                  for (int64_t e = b; e < e2; ++e) { int k = P.ell_ins[(size_t)e]; if (k >= 0 && P.pl_rank[k] != P.rank) k = -1;
                      raw.ell_l[e] = k >= 0 ? h.pl_l[k] : -1;
                      raw.ell_z[e] = k >= 0 ? h.pl_z[2 * (size_t)k] : 0.0; raw.ell_z[L + e] = k >= 0 ? h.pl_z[2 * (size_t)k + 1] : 0.0;
                      raw.ell_w[e] = k >= 0 ? h.pl_info[3 * (size_t)k] : 0.0; raw.ell_w[L + e] = k >= 0 ? h.pl_info[3 * (size_t)k + 1] : 0.0;
                      raw.ell_w[2 * L + e] = k >= 0 ? h.pl_info[3 * (size_t)k + 2] : 0.0; } });
              hipError_t e1 = hipMemcpyAsync(d.ell_l, raw.ell_l.data(), L * sizeof(int32_t), hipMemcpyHostToDevice, g->stream);
              hipError_t e2 = hipMemcpyAsync(d.ell_z, raw.ell_z.data(), 2 * L * sizeof(double), hipMemcpyHostToDevice, g->stream);
              hipError_t e3 = hipMemcpyAsync(d.ell_w, raw.ell_w.data(), 3 * L * sizeof(double), hipMemcpyHostToDevice, g->stream);
              if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) { raw.rc = GS_ERR_HIP; raw.err = "edge streams: copy to the device failed"; } }); } }
    GS_UT("estimates+edges");
    UP(lm_start, P.lm_start); UP(lm_edges, P.lm_edges); UP(ppadj_start, P.ppadj_start);
    { const size_t Q = P.ppinc.size() / 4;                                // device records are 8 bytes: {edge, other endpoint | role << 31}; the pose that
      std::vector<int32_t> inc(2 * Q);                                    // holds the record is known to the kernel; an edge another rank evaluates: edge = -1
      for (size_t q = 0; q < Q; ++q) { const int32_t k = P.ppinc[4 * q], role = P.ppinc[4 * q + 1], other = role ? P.ppinc[4 * q + 2] : P.ppinc[4 * q + 3];
          inc[2 * q] = (P.world > 1 && P.pp_rank[k] != P.rank) ? -1 : k; inc[2 * q + 1] = (int32_t)((uint32_t)other | ((uint32_t)role << 31)); }
      UP(ppinc, inc);
      // a shard's helper thread sent the records of the odometry edges [pp_lo, pp_hi) — the ones that touch its window; an edge the plan gives
      // this rank outside that range (none, by the assignment rule: kept as a check that cannot go wrong silently) is sent now
      std::vector<int32_t> miss;
      for (size_t q = 0; q < Q; ++q) { const int32_t k = inc[2 * q]; if (k >= 0 && ((size_t)k < raw.pp_lo || (size_t)k >= raw.pp_hi)) miss.push_back(k); }
      std::sort(miss.begin(), miss.end()); miss.erase(std::unique(miss.begin(), miss.end()), miss.end());
      for (int32_t k : miss) { double inv[3], o[5]; se2_inverse_host(&h.pp_z[3 * (size_t)k], inv);
          o[0] = inv[0]; o[1] = inv[1]; o[2] = inv[2]; o[3] = std::cos(inv[2]); o[4] = std::sin(inv[2]);
          HIP_TRY(hipMemcpy(d.pp_zinv + 5 * (size_t)k, o, sizeof(o), hipMemcpyHostToDevice));
          HIP_TRY(hipMemcpy(d.pp_info + 6 * (size_t)k, &h.pp_info[6 * (size_t)k], 6 * sizeof(double), hipMemcpyHostToDevice)); }
      g->pp_records_late = (int)miss.size();
      if (ut_on && !miss.empty()) std::fprintf(stderr, "upload: %d odometry edge records sent after the plan\n", (int)miss.size()); }
#define AL(dst, cnt) if ((rc = dev_alloc(g, &d.dst, (size_t)(cnt))) != GS_OK) return rc
#define ZERO(dst, cnt) HIP_TRY(hipMemsetAsync(d.dst, 0, std::max<size_t>((size_t)(cnt), 1) * sizeof(*d.dst), g->stream))
    d.n_wtiles = 0; d.n_groups = 0; d.wt_lo = 0; d.wt_hi = 0; d.rank = P.rank;
    // the fused kernel addresses the ELL planes with 32-bit byte offsets: 8 B * ell_len must stay below 4 GiB
    if (P.lin_ell_ok && !g->force_gather && P.ell_len < ((int64_t)1 << 29)) {
        d.n_wtiles = P.n_wtiles; d.n_groups = (int32_t)P.grp_lm.size();
        UP(wt_desc, P.wt_desc); UP(lm_grp_start, P.lm_grp_start);
        { const size_t Gn = P.grp_slot.size(); std::vector<int32_t> gt(2 * Gn + 2, 0);   // per group {first | end << 16 of its tile-local positions, partial-sum slot}
          for (int w = P.wt_lo; w < P.wt_hi; ++w) { const int ga = P.wt_desc[4 * (size_t)w], gn = P.wt_desc[4 * (size_t)w + 1], pos_off = P.wt_desc[4 * (size_t)w + 2];
              for (int q = ga; q < ga + gn; ++q) { gt[2 * (size_t)q] = (P.grp_pos_start[q] - pos_off) | ((P.grp_pos_start[q + 1] - pos_off) << 16); gt[2 * (size_t)q + 1] = P.grp_slot[q]; } }
          UP(grp_tab, gt); }
        UP(ell_dst, P.ell_dst);
        d.wt_lo = P.wt_lo; d.wt_hi = P.wt_hi;                             // the wave tiles this shard has any edge in (gs_plan.cpp)
    } else if (P.world > 1) return fail(GS_ERR_INVALID, "pose-window shards need the fused linearisation layout (<= 32 observations per pose)");
    // block-sparse H and b live in ONE arena (the variant-3 front assembly addresses every scalar by its offset in it)
    int64_t arena_off[14], arena_doubles = 0;
    { // (the last six parts: the blocks of a grown plan's tail — diagonal blocks and rhs of tail poses / landmarks, off-diagonal blocks of tail edges)
      const int64_t sizes[13] = {(int64_t)N * 6, (int64_t)N * 3, (int64_t)Epp * 9, (int64_t)P.ell_len * 6, (int64_t)d.n_groups * 8, (int64_t)M * 3, (int64_t)M * 2,
                                 (int64_t)TAIL_POSES * 6, (int64_t)TAIL_POSES * 3, (int64_t)TAIL_PP * 9, (int64_t)TAIL_PL * 6, (int64_t)TAIL_LMS * 3, (int64_t)TAIL_LMS * 2};
      arena_off[0] = 0;
      for (int k = 0; k < 13; ++k) { arena_off[k + 1] = arena_off[k] + ((sizes[k] + 1) & ~(int64_t)1);       // 16-byte aligned parts
          if (k + 1 == 4) arena_off[4] = (arena_off[4] + 7) & ~(int64_t)7; }                                   // (the partial-sum records: one 64-byte line each)
      if (arena_off[13] >= ((int64_t)1 << 31)) return fail(GS_ERR_INVALID, "graph too large for 32-bit arena offsets");
      arena_doubles = arena_off[13];
      AL(H_arena, (size_t)arena_off[13] + 2);
      // blocks of edges / tiles this rank never evaluates must read as zero
      ZERO(H_arena, (size_t)arena_off[13] + 2);
      d.t_Hpp_diag = d.H_arena + arena_off[7]; d.t_b_pose = d.H_arena + arena_off[8]; d.t_Hpp_off = d.H_arena + arena_off[9]; d.t_Hpl = d.H_arena + arena_off[10];
      d.t_Hll_diag = d.H_arena + arena_off[11]; d.t_b_lm = d.H_arena + arena_off[12];
      AL(t_pp_ij, (size_t)TAIL_PP * 2); AL(t_pl, (size_t)TAIL_PL * 2); AL(t_pl_z, (size_t)TAIL_PL * 2); AL(t_pl_w, (size_t)TAIL_PL * 3);
      AL(t_pose_start, (size_t)TAIL_POSES + 1); AL(t_pose_edges, (size_t)TAIL_PL); AL(t_lt_id, (size_t)TAIL_PL); AL(t_lt_start, (size_t)TAIL_PL + 1); AL(t_lt_edges, (size_t)TAIL_PL);
      // (the fused linearisation kernel stores Hpp_diag's 6 planes and b_pose's 3 as 9 contiguous planes: 6N is even, no padding between)
      d.Hpp_diag = d.H_arena + arena_off[0]; d.b_pose = d.H_arena + arena_off[1]; d.Hpp_off = d.H_arena + arena_off[2];
      d.Hpl = d.H_arena + arena_off[3]; d.lm_part = d.H_arena + arena_off[4]; d.Hll_diag = d.H_arena + arena_off[5]; d.b_lm = d.H_arena + arena_off[6]; }
    d.n_chi2_partial = std::max((N + 255) / 256, d.n_wtiles);
    AL(chi2_partial, d.n_chi2_partial + 1); AL(chi2, 80); ZERO(chi2_partial, d.n_chi2_partial + 1);     // (+1: the partial of a grown plan's tail)
    UP(pose_known, P.pose_known); UP(lm_known, P.lm_known);
    GS_UT("tiles+arena");
    // plan
    { std::vector<DevFront> df(P.fronts.size());
      for (size_t s = 0; s < P.fronts.size(); ++s) { const Front &F = P.fronts[s]; DevFront &o = df[s];
          o.npiv = F.npiv; o.nbnd = F.nbnd; o.piv0 = F.piv0; o.parent = F.parent; o.asm_off = F.asm_off; o.asm_cnt = F.asm_cnt;
          o.asm_dup = F.asm_dup; o.child_off = F.child_off; o.child_cnt = F.child_cnt; o.owner = F.owner; o.level = F.level; o.pad0 = 0;
          o.bnd_off = F.bnd_off; o.map_off = F.map_off; o.L_off = F.L_off; o.U_off = F.U_off; }
      UP(fronts, df); d.n_fronts = (int32_t)df.size(); }
    // boundary rows, child maps and assembly records with room behind them: a growth step re-writes the runs of the fronts it changes there
    // (sized with the plan, within bounds: a lap-sized graph must not pay for a 100k-pose graph's room with extra device chunks)
    auto room_of = [](size_t n, size_t lo, size_t hi) { return std::min(hi, std::max(lo, n / 2)); };
    const size_t ROOM_ROWS = room_of(P.bnd_rows.size(), 8 * 1024, 64 * 1024), ROOM_RECS = room_of(P.asm_recs.size(), 12 * 1024, 96 * 1024);
    { auto up_room = [&](int32_t **dst, const int32_t *src, size_t n, size_t room) -> int {
          int r2 = dev_alloc(g, dst, n + room); if (r2 != GS_OK) return r2;
          if (n) { hipError_t e = hipMemcpyAsync(*dst, src, n * sizeof(int32_t), hipMemcpyHostToDevice, g->stream); if (e != hipSuccess) return fail(GS_ERR_HIP, hipGetErrorString(e)); }
          return GS_OK; };
      if ((rc = up_room(&d.bnd_rows, P.bnd_rows.data(), P.bnd_rows.size(), ROOM_ROWS)) != GS_OK) return rc;
      if ((rc = up_room(&d.child_map, P.child_map.data(), P.child_map.size(), ROOM_ROWS)) != GS_OK) return rc;
      g->room.cap_bnd = (int64_t)(P.bnd_rows.size() + ROOM_ROWS); g->room.cap_map = (int64_t)(P.child_map.size() + ROOM_ROWS); }
    UP(children, P.children);
    { std::vector<int32_t> cd(P.children.size() * 4);
      for (size_t q = 0; q < P.children.size(); ++q) { const Front &C = P.fronts[P.children[q]];
          cd[4 * q] = P.children[q]; cd[4 * q + 1] = C.npiv | (C.nbnd << 16); cd[4 * q + 2] = C.owner; cd[4 * q + 3] = (int32_t)C.map_off; }
      UP(child_desc, cd); }
    // level lists on the device: this rank's own fronts, then the shared top (empty when world == 1)
    { std::vector<int32_t> lf = P.level_fronts_owned; g->shared_base = (int)lf.size();
      lf.insert(lf.end(), P.level_fronts_shared.begin(), P.level_fronts_shared.end());
      UP(level_fronts, lf); }
    d.xfail_off = -1; d.iter = 0; d.inject_iter = 0; d.inject_code = 0;
    if (P.dist) { UP(x_off, P.x_off);
        d.xfail_off = P.exchange_doubles - 2;                             // the ranks' failure flags ride at the tail of the exchange buffer
        if (!g->exchange_external) { AL(exchange, P.exchange_doubles); ZERO(exchange, P.exchange_doubles); }
        else d.exchange = g->exchange; }
    { static_assert(sizeof(AsmRec) == 16, "AsmRec is uploaded as 4 int32");
      if ((rc = dev_alloc(g, &d.asm_recs, (P.asm_recs.size() + ROOM_RECS) * 4)) != GS_OK) return rc;
      if (!P.asm_recs.empty()) HIP_TRY(hipMemcpyAsync(d.asm_recs, P.asm_recs.data(), P.asm_recs.size() * sizeof(AsmRec), hipMemcpyHostToDevice, g->stream));
      g->room.cap_asm = (int64_t)(P.asm_recs.size() + ROOM_RECS); }
    GS_UT("plan arrays");
    // factor kernel variant (gs_config.factor_variant; gs_debug_options.factor_variant overrides): 0 = default = 3 when every
    // front fits 159 scalars, else 4.  3 = LDL^T on the fp64 matrix cores, a wave or a workgroup per front; 4 = block-per-front
    // VALU Cholesky (any front size).
    { int v = g->default_factor_variant;
      if (g->opt.factor_variant > 0) v = g->opt.factor_variant;
      v = gs_debug_select_factor_variant(v, P.max_front, arena_doubles);
      if (v == 4) v = 0;                                              // device-side code for the block-per-front kernel
      g->wg_f.clear(); g->wg_b.clear(); g->d_wg_f = g->d_wg_b = g->d_wgs_c = g->d_wgs_t = g->d_wgs_b = nullptr;
      d.factor_variant = v;
      d.dbg = g->opt.dbg; d.leaf_nt3 = g->opt.leaf_nt3 != 0 ? 1 : 0; d.f3_lds_kb = std::max(g->opt.f3_lds_kb, 0);
      if (v == 3) {
          std::vector<int32_t> lf = P.level_fronts_owned;
          lf.insert(lf.end(), P.level_fronts_shared.begin(), P.level_fronts_shared.end());
          constexpr int F3W = 224;                           // 32 descriptor ints + the row tables of the first two children + the front's own store table
          // update matrices, packed: row r' (0 .. nbnd, the last = rhs) of the boundary block holds columns 0 .. min(r', nbnd - 1)
          // at r'(r'+1)/2; then one double that stays zero (clamped gathers land on it) and one that collects clamped stores
          std::vector<int32_t> u3_off(P.fronts.size()), u3_size(P.fronts.size());
          { int64_t tot = 0;
            for (size_t f0 = 0; f0 < P.fronts.size(); ++f0) { const int nb = P.fronts[f0].nbnd;
                u3_off[f0] = (int32_t)tot; u3_size[f0] = (nb * (nb + 1)) / 2 + nb; tot += ((u3_size[f0] + 2 + 1) & ~1);
                if (tot >= ((int64_t)1 << 31)) return fail(GS_ERR_INVALID, "update-matrix arena beyond 32-bit offsets"); }
            const int64_t ROOM_U = (int64_t)room_of((size_t)tot, (size_t)128 << 10, (size_t)(P.max_front > 63 ? 4 : 1) << 20);      // doubles: the update matrices of fronts a growth step enlarges move here
            if (tot + ROOM_U >= ((int64_t)1 << 31)) return fail(GS_ERR_INVALID, "update-matrix arena beyond 32-bit offsets");
            AL(Uimg, (size_t)(tot + ROOM_U) + 2); ZERO(Uimg, (size_t)(tot + ROOM_U) + 2);
            g->room.used_U = tot; g->room.cap_U = tot + ROOM_U; }
          UP(u3_off, u3_off); UP(u3_size, u3_size);
          g->u3_off_host = u3_off; g->u3_size_host = u3_size;
          AL(done_f, P.fronts.size()); ZERO(done_f, P.fronts.size());
          d.tickets = nullptr; d.ticket_base = 0;
          if (g->opt.tickets != 0) { AL(tickets, 2); ZERO(tickets, 2); }    // workgroups of the whole-tree launches take their number from this counter (gs_kernels.hip, "tickets")
          d.epoch = 0; d.tree = g->opt.tree != 0 ? 1 : 0; g->fell_back = false; g->fallback_calls = 0; g->fallback_retry_after = 4; g->fallback_retrying = false;   // whole-tree launches for this rank's own subtrees (gs_debug_options.tree = 0: one launch per level)
          // ---- everything below is expanded ON THE DEVICE from the compact plan arrays
          const bool fused = P.lin_ell_ok && d.n_wtiles > 0;
          // block assembly records: the plan's, as they are (AsmRec = 4 ints); landmark-diagonal records of the fused
          // linearisation get their partial-slot range patched in by a kernel
          if ((rc = dev_alloc(g, &d.asm3, (P.asm_recs.size() + ROOM_RECS) * 4)) != GS_OK) return rc;
          if (!P.asm_recs.empty()) HIP_TRY(hipMemcpyAsync(d.asm3, P.asm_recs.data(), P.asm_recs.size() * sizeof(AsmRec), hipMemcpyHostToDevice, g->stream));
          if (fused) { for (int l = 0; l < M; ++l) if (P.lm_grp_start[l + 1] - P.lm_grp_start[l] >= (1 << 22)) return fail(GS_ERR_INVALID, "landmark seen from too many wave tiles");
              launch_patch_asm3((int64_t)P.asm_recs.size(), d.asm3, d.lm_grp_start, g->stream); }
          GS_UT("asm3");
          // scalar assembly records {offset in H_arena, offset in the staging image}, padded per front to a multiple of
          // 64 with (0 -> image offset 1, a don't-care upper-triangle slot); fused landmark diagonals go to lm3.
          // The host only counts them per front.
          { const size_t S = P.fronts.size();
            std::vector<int32_t> bf(8 * S, 0);
            parallel_chunks((int64_t)S, 2048, [&](int64_t b, int64_t e, int) {
                for (int64_t sidx = b; sidx < e; ++sidx) { const Front &F = P.fronts[sidx]; int ns = 0, nl = 0;
                    for (int t = F.asm_off; t < F.asm_off + F.asm_cnt - F.asm_dup; ++t) { const int k = P.asm_recs[t].kind;
                        if (k == 0) ns += 9; else if (k == 1) { if (fused) ++nl; else ns += 5; } else if (k <= 3) ns += 9; else if (k == 6) ns += 5; else ns += 6; }
                    bf[8 * sidx + 4] = (ns + 63) & ~63; bf[8 * sidx + 6] = nl; } });
            int64_t so = 0, lo = 0;
            for (size_t sidx = 0; sidx < S; ++sidx) { const Front &F = P.fronts[sidx]; int32_t *r = &bf[8 * sidx];
                if (so >= ((int64_t)1 << 31) - 64) return fail(GS_ERR_INVALID, "too many assembly scalars");
                r[0] = F.asm_off; r[1] = F.asm_cnt - F.asm_dup; r[2] = F.npiv + F.nbnd; r[3] = (int32_t)so; r[5] = (int32_t)lo; so += r[4]; lo += r[6]; }
            const int64_t ROOM_SC = (int64_t)room_of((size_t)so, (size_t)64 << 10, (size_t)(P.max_front > 63 ? 4 : 1) << 19);         // scalar records of the fronts a growth step rebuilds
            if (so + ROOM_SC >= ((int64_t)1 << 31) - 64) return fail(GS_ERR_INVALID, "too many assembly scalars");
            AL(sc3, 2 * (size_t)(so + ROOM_SC) + 2); AL(lm3, 4 * (size_t)lo + 4);
            g->room.used_sc = so; g->room.cap_sc = so + ROOM_SC;
            int32_t *bf_dev = nullptr; if ((rc = dev_upload(g, &bf_dev, bf)) != GS_OK) return rc;
            g->d_bf = bf_dev; g->bf_host = bf;
            Sc3Args A; for (int k = 0; k < 8; ++k) A.off[k] = arena_off[k];
            A.L = P.ell_len; A.N = N; A.M = M; A.Epp = Epp; A.fused = fused ? 1 : 0;
            for (int k = 0; k < 6; ++k) A.toff[k] = arena_off[7 + k];
            A.tcapN = TAIL_POSES; A.tcapEpp = TAIL_PP; A.tcapEpl = TAIL_PL; A.tcapM = TAIL_LMS;
            g->sc3_args = A;
            launch_build_sc3(bf_dev, d.asm3, d.sc3, d.lm3, (int)S, A, g->stream);
            GS_UT("sc3 build");
            // descriptors + children tables: one wave per level position (k_build_f3)
            d.f3x_stride = P.max_front > 63 ? 168 : 72;                 // a child's row table: 64 entries, or 160 when the plan holds a big front
            std::vector<int32_t> xrow(lf.size() + 1, 0);
            for (size_t q = 0; q < lf.size(); ++q) { xrow[q + 1] = xrow[q] + d.f3x_stride * P.fronts[lf[q]].child_cnt;
                if (xrow[q + 1] >= (1 << 30)) return fail(GS_ERR_INVALID, "children table too large"); }
            int32_t *xrow_dev = nullptr; if ((rc = dev_upload(g, &xrow_dev, xrow)) != GS_OK) return rc;
            g->d_xrow = xrow_dev;
            g->pos_of_front.assign(P.fronts.size(), -1);
            for (size_t q = 0; q < lf.size(); ++q) g->pos_of_front[lf[q]] = (int32_t)q;
            if ((rc = dev_upload(g, &g->d_posof, g->pos_of_front)) != GS_OK) return rc;      // front -> level position: into the children's headers (k_factor3_sub finds a leaf's descriptor through it)
            if ((rc = dev_alloc(g, &g->d_patch, (size_t)1024 * 32)) != GS_OK || (rc = dev_alloc(g, &g->d_list, (size_t)2048)) != GS_OK) return rc;
            AL(f3_desc, lf.size() * (size_t)F3W); AL(f3_x, (size_t)xrow[lf.size()] + 168);
            launch_build_f3((int)lf.size(), d.level_fronts, d.fronts, d.children, d.child_map, d.u3_off, d.u3_size, bf_dev, xrow_dev,
                            P.dist ? d.x_off : nullptr, d.f3_desc, d.f3_x, d.f3x_stride, g->stream, nullptr, g->d_posof);
            // a growth step needs all of the above: variant 3, one GPU, the fused linearisation layout
            g->room.ok = !P.dist && fused;
            GS_UT("f3 tables"); }
      } }
    GS_UT("f3 x+desc upload");
    AL(dbg_ts, 64); ZERO(dbg_ts, 64);
    AL(done_ts, 2 * P.fronts.size() + 2); ZERO(done_ts, 2 * P.fronts.size() + 2);
    { const int64_t room_L = g->room.ok ? (int64_t)room_of((size_t)P.l_doubles, (size_t)256 << 10, (size_t)(P.max_front > 63 ? 8 : 2) << 20) : 0;          // doubles: the L panels of fronts a growth step enlarges move here
      AL(Lbuf, P.l_doubles + room_L); g->room.cap_L = P.l_doubles + room_L; }
    AL(Ubuf, d.factor_variant == 0 ? P.u_doubles : 1);              // variant 3 keeps its update matrices in Uimg
    AL(xe, P.n_scalar + 3 * TAIL_POSES + 2 * TAIL_LMS); g->room.cap_xe = P.n_scalar + 3 * TAIL_POSES + 2 * TAIL_LMS;
    AL(dpose, ((size_t)N + TAIL_POSES) * 3); AL(dlm, ((size_t)M + TAIL_LMS) * 2); AL(fail, 4);
    HIP_TRY(hipMemsetAsync(d.fail, 0, 4 * sizeof(int32_t), g->stream));
    HIP_TRY(hipMemsetAsync(d.chi2, 0, 80 * sizeof(double), g->stream));
    HIP_TRY(hipMemsetAsync(d.dpose, 0, ((size_t)N + TAIL_POSES) * 3 * sizeof(double), g->stream));
    HIP_TRY(hipMemsetAsync(d.dlm, 0, ((size_t)M + TAIL_LMS) * 2 * sizeof(double), g->stream));
    // per-level launch parameters and the global workspace for fronts beyond the LDS limit
    const int nlev = (int)P.level_start.size() - 1;
    const int lim = factor_lds_limit_f();
    int64_t max_blocks_oversize = 0;
    auto level_params = [&](const std::vector<int32_t> &start, const std::vector<int32_t> &list, gs_graph::LevelSet &ls) {
        ls.start = start; ls.max_f.assign(nlev, 0); ls.max_npiv.assign(nlev, 0); ls.max_nbnd.assign(nlev, 0);
        for (int l = 0; l < nlev; ++l) {
            for (int q = start[l]; q < start[l + 1]; ++q) { const Front &F = P.fronts[list[q]];
                ls.max_f[l] = std::max(ls.max_f[l], F.npiv + F.nbnd);
                ls.max_npiv[l] = std::max(ls.max_npiv[l], F.npiv); ls.max_nbnd[l] = std::max(ls.max_nbnd[l], F.nbnd); }
            if (ls.max_f[l] > lim) { const int64_t f = ls.max_f[l];
                d.front_ws_stride = std::max(d.front_ws_stride, ((f + 1) | 1) * f);
                max_blocks_oversize = std::max<int64_t>(max_blocks_oversize, start[l + 1] - start[l]); }
        }
    };
    level_params(P.level_start_owned, P.level_fronts_owned, g->own);
    level_params(P.level_start_shared, P.level_fronts_shared, g->shared);
    if (max_blocks_oversize > 0) {      // fronts beyond the LDS limit use a global workspace, one slice per block
        max_blocks_oversize = std::max<int64_t>(max_blocks_oversize, (int64_t)P.level_fronts_shared.size());
        AL(front_ws, d.front_ws_stride * max_blocks_oversize);
    }
#undef UP
#undef AL
#undef ZERO
    GS_UT("arenas+levels");
    if (raw.th.joinable()) { raw.th.join(); if (raw.rc != GS_OK) return fail(raw.rc, raw.err); }      // a shard's edge streams (above)
    HIP_TRY(hipStreamSynchronize(g->stream));
    GS_UT("final sync");
    g->dev_valid = true; g->dev_estimates_newer = false; g->tree_proven = false;
    g->dev_estimate_version = h.estimate_version;
    return GS_OK;
}

// ---- append-only growth on the device (after gs::grow_plan changed the host plan): the new poses' and edges' data into the tail
// arrays, the re-written runs of the changed fronts behind the plan arrays, those fronts' rows of the compact tables through one
// patch buffer, then the device-side expansion (k_build_sc3, k_build_f3) for those fronts only.  Everything older stays where it
// is.  Returns GS_ERR_CAPACITY when the room left by the full structure phase is used up (the caller rebuilds).
static int upload_growth(gs_graph *g, const Growth &gr) {
    const HostGraph &h = g->h; const Plan &P = g->plan; DevGraph &d = g->d;
    if (!g->room.ok || d.factor_variant != 3) return fail(GS_ERR_CAPACITY, "growth: this plan was not uploaded with room to grow");
    const int nf = (int)gr.fronts.size();
    if (nf > 1024 || (int64_t)P.bnd_rows.size() > g->room.cap_bnd || (int64_t)P.child_map.size() > g->room.cap_map ||
        (int64_t)P.asm_recs.size() > g->room.cap_asm || P.l_doubles > g->room.cap_L || P.n_scalar > g->room.cap_xe)
        return fail(GS_ERR_CAPACITY, "growth: room behind the plan arrays used up");
    const bool fused = g->sc3_args.fused != 0;
    // update-matrix slots and scalar-record runs of the changed fronts
    std::vector<int32_t> patch((size_t)nf * 32, 0), poslist(nf);
    int64_t used_U = g->room.used_U, used_sc = g->room.used_sc;
    for (int i = 0; i < nf; ++i) { const int s = gr.fronts[i]; const Front &F = P.fronts[s]; int32_t *r = &patch[(size_t)i * 32];
        const int nb = F.nbnd; const int32_t usz = (nb * (nb + 1)) / 2 + nb;
        int ns = 0;
        for (int t = F.asm_off; t < F.asm_off + F.asm_cnt - F.asm_dup; ++t) { const int k = P.asm_recs[t].kind;
            if (k == 0) ns += 9; else if (k == 1) { if (!fused) ns += 5; } else if (k <= 3) ns += 9; else if (k == 6) ns += 5; else ns += 6; }
        const int32_t sc_cnt = (ns + 63) & ~63;
        if (used_U + usz + 4 > g->room.cap_U || used_sc + sc_cnt > g->room.cap_sc) return fail(GS_ERR_CAPACITY, "growth: room behind the update matrices / scalar records used up");
        DevFront o; o.npiv = F.npiv; o.nbnd = F.nbnd; o.piv0 = F.piv0; o.parent = F.parent; o.asm_off = F.asm_off; o.asm_cnt = F.asm_cnt;
        o.asm_dup = F.asm_dup; o.child_off = F.child_off; o.child_cnt = F.child_cnt; o.owner = F.owner; o.level = F.level; o.pad0 = 0;
        o.bnd_off = F.bnd_off; o.map_off = F.map_off; o.L_off = F.L_off; o.U_off = F.U_off;
        r[0] = s; std::memcpy(r + 1, &o, sizeof(o));
        r[21] = (int32_t)used_U; r[22] = usz; used_U += (usz + 2 + 1) & ~1;
        const int32_t *b0 = &g->bf_host[8 * (size_t)s];
        r[23] = F.asm_off; r[24] = F.asm_cnt - F.asm_dup; r[25] = F.npiv + F.nbnd; r[26] = (int32_t)used_sc; r[27] = sc_cnt; r[28] = b0[5]; r[29] = b0[6]; r[30] = 0;
        used_sc += sc_cnt;
        poslist[i] = g->pos_of_front[s];
        if (poslist[i] < 0) return fail(GS_ERR_INVALID, "growth: front without a level position"); }
    // ---- from here on the device changes
    const int N0 = gr.first_pose, N1 = P.planned_N, E0 = gr.first_pp, E1 = P.planned_Epp, K0 = gr.first_pl, K1 = P.planned_Epl;
    std::vector<double> zinv((size_t)(E1 - E0) * 5), plz((size_t)(K1 - K0) * 2), plw((size_t)(K1 - K0) * 3);
    std::vector<int32_t> ppij((size_t)(E1 - E0) * 2), plpl((size_t)(K1 - K0) * 2);
    auto H2D = [&](void *dst, const void *src, size_t bytes) -> int {
        if (!bytes) return GS_OK;
        hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, g->stream);
        return e == hipSuccess ? GS_OK : fail(GS_ERR_HIP, std::string("growth upload: ") + hipGetErrorString(e)); };
    int rc;
    if ((rc = H2D(d.pose_est + 3 * (size_t)N0, &h.pose_est[3 * (size_t)N0], (size_t)(N1 - N0) * 3 * sizeof(double))) != GS_OK) return rc;
    if ((rc = H2D(d.pose_gidx + N0, &P.pose_gidx[N0], (size_t)(N1 - N0) * sizeof(int32_t))) != GS_OK) return rc;
    { const int M0 = gr.first_lm, M1 = P.planned_M;                  // landmarks first seen by the new poses
      if (M1 > M0) { if ((rc = H2D(d.lm_est + 2 * (size_t)M0, &h.lm_est[2 * (size_t)M0], (size_t)(M1 - M0) * 2 * sizeof(double))) != GS_OK) return rc;
          if ((rc = H2D(d.lm_gidx + M0, &P.lm_gidx[M0], (size_t)(M1 - M0) * sizeof(int32_t))) != GS_OK) return rc; } }
    if (g->dev_estimate_version != h.estimate_version) {          // a host-side setEstimate on an OLDER vertex since the last upload (g2o: setEstimate, then
        const int M1 = P.planned_M;                                 // optimize() uses the new value): the whole estimate arrays go up again, not only the tail's
        if ((rc = H2D(d.pose_est, h.pose_est.data(), (size_t)N1 * 3 * sizeof(double))) != GS_OK) return rc;
        if ((rc = H2D(d.lm_est, h.lm_est.data(), (size_t)M1 * 2 * sizeof(double))) != GS_OK) return rc;
        launch_pose_trig_range(d, 0, N1, g->stream);
    } else launch_pose_trig_range(d, N0, N1 - N0, g->stream);
    for (int k = E0; k < E1; ++k) { double inv[3]; se2_inverse_host(&h.pp_z[3 * (size_t)k], inv);
        double *o = &zinv[5 * (size_t)(k - E0)]; o[0] = inv[0]; o[1] = inv[1]; o[2] = inv[2]; o[3] = std::cos(inv[2]); o[4] = std::sin(inv[2]);
        ppij[2 * (size_t)(k - E0)] = h.pp_i[k]; ppij[2 * (size_t)(k - E0) + 1] = h.pp_j[k]; }
    if ((rc = H2D(d.pp_zinv + 5 * (size_t)E0, zinv.data(), zinv.size() * sizeof(double))) != GS_OK) return rc;
    if ((rc = H2D(d.pp_info + 6 * (size_t)E0, &h.pp_info[6 * (size_t)E0], (size_t)(E1 - E0) * 6 * sizeof(double))) != GS_OK) return rc;
    if ((rc = H2D(d.t_pp_ij + 2 * (size_t)(E0 - P.base_Epp), ppij.data(), ppij.size() * sizeof(int32_t))) != GS_OK) return rc;
    for (int k = K0; k < K1; ++k) { const size_t q = (size_t)(k - K0);
        plpl[2 * q] = h.pl_p[k]; plpl[2 * q + 1] = h.pl_l[k]; plz[2 * q] = h.pl_z[2 * (size_t)k]; plz[2 * q + 1] = h.pl_z[2 * (size_t)k + 1];
        for (int c = 0; c < 3; ++c) plw[3 * q + c] = h.pl_info[3 * (size_t)k + c]; }
    { const size_t s0 = (size_t)(K0 - P.base_Epl);
      if ((rc = H2D(d.t_pl + 2 * s0, plpl.data(), plpl.size() * sizeof(int32_t))) != GS_OK) return rc;
      if ((rc = H2D(d.t_pl_z + 2 * s0, plz.data(), plz.size() * sizeof(double))) != GS_OK) return rc;
      if ((rc = H2D(d.t_pl_w + 3 * s0, plw.data(), plw.size() * sizeof(double))) != GS_OK) return rc; }
    // the tail's edges grouped by pose and by touched landmark (edge order inside a group: the order of the sums in k_linearize_tail);
    // the whole tail, not only this step's part
    std::vector<int32_t> tps, tpe, ltid, lts, lte;
    { const int tN = N1 - P.base_N, tE = K1 - P.base_Epl;
      tps.assign((size_t)tN + 1, 0); tpe.resize((size_t)tE);
      for (int e = 0; e < tE; ++e) tps[(size_t)(h.pl_p[P.base_Epl + e] - P.base_N) + 1]++;
      for (int t = 0; t < tN; ++t) tps[(size_t)t + 1] += tps[(size_t)t];
      { std::vector<int32_t> fill(tps.begin(), tps.end() - 1); for (int e = 0; e < tE; ++e) tpe[(size_t)fill[(size_t)(h.pl_p[P.base_Epl + e] - P.base_N)]++] = e; }
      std::vector<std::pair<int32_t, int32_t>> le; le.reserve((size_t)tE);        // (landmark, edge), fixed cones left out: nothing is summed for them
      for (int e = 0; e < tE; ++e) { const int l = h.pl_l[P.base_Epl + e]; if (!h.lm_fixed[l]) le.emplace_back(l, e); }
      std::sort(le.begin(), le.end());
      lts.push_back(0);
      for (size_t q = 0; q < le.size(); ++q) { if (q == 0 || le[q].first != le[q - 1].first) { if (q) lts.push_back((int32_t)q); ltid.push_back(le[q].first); } lte.push_back(le[q].second); }
      if (!le.empty()) lts.push_back((int32_t)le.size());
      d.tLt = (int32_t)ltid.size();
      if ((rc = H2D(d.t_pose_start, tps.data(), tps.size() * sizeof(int32_t))) != GS_OK || (rc = H2D(d.t_pose_edges, tpe.data(), tpe.size() * sizeof(int32_t))) != GS_OK ||
          (rc = H2D(d.t_lt_id, ltid.data(), ltid.size() * sizeof(int32_t))) != GS_OK || (rc = H2D(d.t_lt_start, lts.data(), lts.size() * sizeof(int32_t))) != GS_OK ||
          (rc = H2D(d.t_lt_edges, lte.data(), lte.size() * sizeof(int32_t))) != GS_OK) return rc; }
    // the re-written runs
    if ((rc = H2D(d.bnd_rows + gr.bnd_from, &P.bnd_rows[(size_t)gr.bnd_from], (P.bnd_rows.size() - (size_t)gr.bnd_from) * sizeof(int32_t))) != GS_OK) return rc;
    if ((rc = H2D(d.child_map + gr.map_from, &P.child_map[(size_t)gr.map_from], (P.child_map.size() - (size_t)gr.map_from) * sizeof(int32_t))) != GS_OK) return rc;
    { const size_t na = P.asm_recs.size() - (size_t)gr.asm_from;
      if (na) { if ((rc = H2D(d.asm_recs + 4 * gr.asm_from, &P.asm_recs[(size_t)gr.asm_from], na * sizeof(AsmRec))) != GS_OK) return rc;
          if ((rc = H2D(d.asm3 + 4 * gr.asm_from, &P.asm_recs[(size_t)gr.asm_from], na * sizeof(AsmRec))) != GS_OK) return rc;
          if (fused) launch_patch_asm3((int64_t)na, d.asm3 + 4 * gr.asm_from, d.lm_grp_start, g->stream); } }
    // compact tables: the changed fronts' rows
    if ((rc = H2D(g->d_patch, patch.data(), patch.size() * sizeof(int32_t))) != GS_OK) return rc;
    launch_apply_front_patch(nf, g->d_patch, d.fronts, d.u3_off, d.u3_size, g->d_bf, g->stream);
    if ((rc = H2D(g->d_list, gr.fronts.data(), (size_t)nf * sizeof(int32_t))) != GS_OK) return rc;
    if ((rc = H2D(g->d_list + 1024, poslist.data(), (size_t)nf * sizeof(int32_t))) != GS_OK) return rc;
    // device-side expansion for those fronts: scalar records, then descriptors + children tables (a changed front's parent is a
    // changed front too: its copy of the child's row table is rebuilt with it)
    launch_build_sc3(g->d_bf, d.asm3, d.sc3, d.lm3, nf, g->sc3_args, g->stream, g->d_list);
    launch_build_f3(nf, d.level_fronts, d.fronts, d.children, d.child_map, d.u3_off, d.u3_size, g->d_bf, g->d_xrow, nullptr, d.f3_desc, d.f3_x, d.f3x_stride, g->stream, g->d_list + 1024, g->d_posof);
    d.n_scalar = P.n_scalar; d.tN = N1 - P.base_N; d.tM = P.planned_M - P.base_M; d.tEpp = E1 - P.base_Epp; d.tEpl = K1 - P.base_Epl;
    HIP_TRY(hipStreamSynchronize(g->stream));                       // the staging vectors above go out of scope
    { hipError_t e = hipGetLastError(); if (e != hipSuccess) return fail(GS_ERR_HIP, std::string("growth: ") + hipGetErrorString(e)); }
    // host mirrors and launch parameters
    for (int i = 0; i < nf; ++i) { const int s = gr.fronts[i]; const int32_t *r = &patch[(size_t)i * 32];
        g->u3_off_host[s] = r[21]; g->u3_size_host[s] = r[22]; for (int c = 0; c < 8; ++c) g->bf_host[8 * (size_t)s + c] = r[23 + c]; }
    g->room.used_U = used_U; g->room.used_sc = used_sc;
    { gs_graph::LevelSet &ls = g->own; const int nlev = (int)ls.start.size() - 1;
      for (int l = 0; l < nlev; ++l) { ls.max_f[l] = ls.max_npiv[l] = ls.max_nbnd[l] = 0;
          for (int q = ls.start[l]; q < ls.start[l + 1]; ++q) { const Front &F = P.fronts[P.level_fronts_owned[q]];
              ls.max_f[l] = std::max(ls.max_f[l], F.npiv + F.nbnd); ls.max_npiv[l] = std::max(ls.max_npiv[l], F.npiv); ls.max_nbnd[l] = std::max(ls.max_nbnd[l], F.nbnd); } } }
    g->leaf_n = -1; g->block_n = -1; g->tree_proven = false;        // the leaf instance and its LDS slot are chosen again from the grown fronts
    g->wg_f.clear(); g->wg_b.clear(); g->d_wg_f = g->d_wg_b = g->d_wgs_c = g->d_wgs_t = g->d_wgs_b = nullptr;   // ... and so are the workgroup tables of a plan with workgroup fronts (a grown front may change its size class)
    g->dev_estimate_version = h.estimate_version;
    return GS_OK;
}

// The factor kernel a plan gets (gs_config.factor_variant; gs_debug_options.factor_variant overrides): 0 = default = 3.
//   3 = LDL^T on the fp64 matrix cores, latency-shaped: a front of up to 63 scalars a wave, one of 64 .. 159 a workgroup, chosen per
//   front; 4 = block-per-front VALU Cholesky (any front size, 64-bit addressing throughout).  (Rounds 1-3 also kept a wave-per-front
//   VALU kernel and a first matrix-core Cholesky as variants 1 and 2; nothing but tests ran them: removed in round 4, requests for
//   them get variant 3.)
// Variant 3 names every scalar of the linearised system by a 32-bit BYTE offset into H_arena ((uint32_t)record * 8 in the front
// kernels): beyond 2^29 doubles (4 GiB) those would wrap and assemble the wrong entries silently, so such a graph gets variant 4.
extern "C" int gs_debug_select_factor_variant(int32_t requested, int32_t max_front, int64_t arena_doubles) {
    int v = requested;
    if (v != 4) v = 3;
    if (v == 3 && max_front > 159) v = 4;                           // variant 3: a wave up to 63 scalars, a workgroup up to 159 (ten tile rows)
    if (v == 3 && arena_doubles >= ((int64_t)1 << 29)) v = 4;
    return v;
}

static int build_plan_host(gs_graph *g) {
    PlanOptions o; o.leaf_poses = g->cfg.leaf_poses; o.world = g->world; o.rank = g->rank;
    const gs_debug_options &t = g->opt;                             // tuning overrides (graphslam_debug.h)
    if (t.leaf_poses > 0) o.leaf_poses = t.leaf_poses;
    if (t.cluster_ways > 0) o.cluster_ways = t.cluster_ways;             // 2 = binary dissection down to the leaves
    if (t.ell_lanes > 0) o.ell_lanes = t.ell_lanes;                      // lanes per pose of the ELL layout
    if (t.big_cluster >= 0) o.big_cluster_front = t.big_cluster;         // 0 = clusters only where they fit a wave
    if (t.grow_headroom >= 0) { o.grow_headroom = t.grow_headroom; o.grow_spine_headroom = std::min(o.grow_spine_headroom, 3 * o.grow_headroom); }   // 0 = cluster fronts up to the full 63 scalars
    o.timing = t.plan_timing > 0;
    o.by_window = t.shard_by_window != 0;
    o.force_shared_top = g->world <= 1 ? std::max(t.force_shared_top, 0) : 0;
    if (g->world > 1 && !g->lm_seen_interior.empty()) {
        if ((int)g->lm_seen_interior.size() != g->h.n_lms()) return fail(GS_ERR_INVALID, "gs_dist_set_landmark_windows: the masks cover another number of landmarks than the graph holds");
        o.lm_seen_interior = g->lm_seen_interior.data(); o.lm_seen_first = g->lm_seen_first.data(); }
    std::string err;
    if (!build_plan(g->h, o, g->plan, err, &g->plan_ws)) { g->plan_version = ~0ull; return fail(GS_ERR_EMPTY, "plan: " + err); }
    g->plan_version = g->h.structure_version;
    return GS_OK;
}

extern "C" int gs_plan_build_host(gs_graph *g, gs_plan_info *info) {
    if (!g) return fail(GS_ERR_INVALID, "null graph");
    int rc = GS_OK;
    // a host-only handle absorbs appended poses / edges the way a device handle does (gs::grow_plan), so that the grown plan can be
    // inspected and replayed without a GPU; GS_GROW=0 or any other change: full build
    bool grown = false;
    if (g->host_only && g->plan.valid && g->plan_version != ~0ull && g->plan_version != g->h.structure_version) {
        const bool on = g->opt.grow != 0;
        Growth gr; std::string why;
        if (on && grow_plan(g->h, g->plan, gr, why)) { grown = true; g->plan_version = g->h.structure_version; g->no_growth_reason.clear(); }
        else g->no_growth_reason = on ? why : "growth switched off (gs_debug_options.grow = 0 / GS_GROW=0)";
    }
    if (!grown) { rc = build_plan_host(g); if (rc != GS_OK) return rc; }
    if (info) { const Plan &P = g->plan; info->n_scalar = P.n_scalar; info->n_fronts = (int32_t)P.fronts.size();
        info->n_levels = (int32_t)P.level_start.size() - 1; info->max_front = P.max_front; info->l_doubles = P.l_doubles;
        info->u_doubles = P.u_doubles; info->n_asm_blocks = (int64_t)P.asm_recs.size(); info->n_child_map = (int64_t)P.child_map.size(); }
    // a host-only plan must not be mistaken for an uploaded one
    if (g->dev_valid && !g->host_only) { hipSetDevice(g->device); hipStreamSynchronize(g->stream); pull_estimates_if_needed(g); dev_free_all(g); }
    return GS_OK;
}
extern "C" int gs_plan_growths(gs_graph *g) { return g ? g->plan.n_growths : fail(GS_ERR_INVALID, "null graph"); }
extern "C" const char *gs_growth_refusal(gs_graph *g) { return g ? g->no_growth_reason.c_str() : ""; }
extern "C" int gs_plan_export(gs_graph *g, int32_t *out, int64_t *out_len) {
    if (!g || !out_len) return fail(GS_ERR_INVALID, "null argument");
    if (!g->plan.valid) return fail(GS_ERR_NOT_INITIALIZED, "no plan built");
    std::vector<int32_t> v; export_plan(g->plan, v);
    if (!out) { *out_len = (int64_t)v.size(); return GS_OK; }
    if (*out_len < (int64_t)v.size()) return fail(GS_ERR_CAPACITY, "buffer too small");
    std::memcpy(out, v.data(), v.size() * sizeof(int32_t)); *out_len = (int64_t)v.size();
    return GS_OK;
}

extern "C" int gs_initialize_optimization(gs_graph *g) {
    if (!g) return fail(GS_ERR_INVALID, "null graph");
    int rc = ensure_device(g); if (rc != GS_OK) return rc;
    auto t0 = std::chrono::steady_clock::now();
    rc = pull_estimates_if_needed(g); if (rc != GS_OK) return rc;
    HIP_TRY(hipStreamSynchronize(g->stream));
    // append-only growth: poses / edges added since the plan was built enter the existing plan and device tables (gs::grow_plan,
    // upload_growth); anything else — or GS_GROW=0 — rebuilds
    g->no_growth_reason.clear();
    if (g->dev_valid && g->plan.valid && g->plan_version != ~0ull && g->plan_version != g->h.structure_version) {
        const bool on = g->opt.grow != 0;
        Growth gr; std::string why;
        // below ~a hundred poses the full phase costs 0.25 ms, the tail kernel of ten iterations 0.08: nothing to gain (a lap, 200 poses
        // mapped: 1.03 ms per optimize(10) grown against 1.16 rebuilt, scripts/keyframe_stream.py)
        const int min_poses = g->opt.grow_min_poses;
        if (!on) g->no_growth_reason = "growth switched off (gs_debug_options.grow = 0 / GS_GROW=0)";
        else if (g->plan.base_N < min_poses) g->no_growth_reason = "graph below the size at which growing pays (GS_GROW_MIN_POSES)";
        else if (!g->room.ok) g->no_growth_reason = "plan uploaded without room to grow";
        else if (grow_plan(g->h, g->plan, gr, why)) {
            rc = upload_growth(g, gr);
            if (rc == GS_OK) { g->plan_version = g->h.structure_version;
                g->ms_structure = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
                return GS_OK; }
            g->no_growth_reason = g_last_error;                     // (the plan object is rebuilt from scratch below)
        } else g->no_growth_reason = why;
    }
    const bool st_on = g->opt.plan_timing != 0; auto st_prev = std::chrono::steady_clock::now();      // gs_debug_options.plan_timing: the steps of this call on stderr
    auto ST = [&](const char *what) { if (st_on) { auto n_ = std::chrono::steady_clock::now();
        std::fprintf(stderr, "structure %-24s %.2f ms\n", what, std::chrono::duration<double, std::milli>(n_ - st_prev).count()); st_prev = n_; } };
    dev_release(g, true);                                            // the handle keeps its device memory for the new plan
    RawUpload raw;
    rc = upload_raw_begin(g, raw); if (rc != GS_OK) { if (raw.th.joinable()) raw.th.join(); dev_free_all(g); return rc; }
    ST("release + raw begin");
    rc = build_plan_host(g);                                        // the host threads build the plan while the raw arrays travel
    ST("plan (host)");
    raw.th.join();
    ST("wait for the raw upload");
    if (rc == GS_OK && raw.rc != GS_OK) rc = fail(raw.rc, raw.err);
    if (rc != GS_OK) { dev_free_all(g); return rc; }
    rc = upload_graph(g, raw); if (rc != GS_OK) { dev_free_all(g); return rc; }
    ST("upload_graph");
    dev_trim(g);
    ST("trim");
    g->ms_structure = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return GS_OK;
}

static int ensure_ready(gs_graph *g) {
    int rc = ensure_device(g); if (rc != GS_OK) return rc;
    if (!g->dev_valid || g->plan_version != g->h.structure_version) return gs_initialize_optimization(g);
    if (g->dev_estimate_version != g->h.estimate_version) {      // host-side setEstimate since the upload
        const int N = g->d.N + g->d.tN, M = g->d.M + g->d.tM;
        if (N > 0) HIP_TRY(hipMemcpyAsync(g->d.pose_est, g->h.pose_est.data(), (size_t)N * 3 * sizeof(double), hipMemcpyHostToDevice, g->stream));
        if (M > 0) HIP_TRY(hipMemcpyAsync(g->d.lm_est, g->h.lm_est.data(), (size_t)M * 2 * sizeof(double), hipMemcpyHostToDevice, g->stream));
        launch_pose_trig(g->d, g->stream);
        HIP_TRY(hipStreamSynchronize(g->stream));
        g->dev_estimate_version = g->h.estimate_version; g->dev_estimates_newer = false;
    }
    return GS_OK;
}

// ------------------------------------------------------------------ one Gauss-Newton iteration (A5-A9)
// own fronts bottom-up (mode 0), shared top bottom-up from the all-reduced exchange buffer (mode 2)
// ---- plans that hold a front of more than 63 scalars (single GPU): the workgroup tables of the table-driven launches, built on
// first use.  Factor: level positions upwards from the end of the leaf launch — a big front a workgroup (NT = 7 or 10 tile
// rows), a small front of the upper levels (the last block_n positions) four waves, other small fronts a wave each in groups of
// up to four that do not straddle a level.  Backward solve: every position from the root downwards.
static int build_big_tables(gs_graph *g, const gs_graph::LevelSet &ls) {
    const Plan &P = g->plan; const int nlev = (int)ls.start.size() - 1, total = ls.start[nlev];
    auto f_of = [&](int q) { const Front &F = P.fronts[P.level_fronts_owned[q]]; return F.npiv + F.nbnd; };
    auto big_kind = [&](int f) { return f <= 79 ? 4 : (f <= 111 ? 2 : 3); };        // 5, 7 or 10 tile rows
    g->wg_f.clear(); g->wg_b.clear(); g->seg_f.clear(); g->seg_b.clear();
    g->wgs_c.clear(); g->wgs_t.clear(); g->wgs_b.clear(); g->segs_c.clear(); g->segs_t.clear(); g->segs_b.clear();
    g->small_max_npiv = 1; g->small_max_f = 1;
    for (const Front &F : P.fronts) if (!F.opaque && F.npiv + F.nbnd <= 63) { g->small_max_npiv = std::max(g->small_max_npiv, (int)F.npiv); g->small_max_f = std::max(g->small_max_f, F.npiv + F.nbnd); }
    auto push = [&](std::vector<int32_t> &tab, std::vector<gs_graph::WgSeg> &segs, int pos, int kind_cnt, int level, size_t lds, int cls) {
        const int e = (int)tab.size() / 2; tab.push_back(pos); tab.push_back(kind_cnt);
        if (!segs.empty() && segs.back().level == level && segs.back().lds == lds && segs.back().cls == cls) ++segs.back().count; else segs.push_back({e, 1, level, lds, cls}); };
    auto fcls = [](int kind) { return kind == 4 ? 0 : (kind == 3 ? 2 : 1); };       // factor kernel class: fronts of 64-79 | small fronts and 80-111 | 112-159
    const int first = std::max(g->leaf_n, 0), first_block = total - std::max(g->block_n, 0);
    for (int l = 0; l < nlev; ++l)
        for (int q = std::max(ls.start[l], first); q < ls.start[l + 1]; ) {
            const int f = f_of(q);
            if (f > 63) { const int k = big_kind(f); push(g->wg_f, g->seg_f, q, k, l, factor_tab_lds_bytes(k), fcls(k)); ++q; }
            else if (q >= first_block) { push(g->wg_f, g->seg_f, q, 1 | (1 << 8), l, factor_tab_lds_bytes(1), 1); ++q; }
            else { int cnt = 1; while (cnt < 4 && q + cnt < ls.start[l + 1] && q + cnt < first_block && f_of(q + cnt) <= 63) ++cnt;
                push(g->wg_f, g->seg_f, q, 0 | (cnt << 8), l, factor_tab_lds_bytes(0), 1); q += cnt; } }
    for (int l = nlev - 1; l >= 0; --l)
        for (int q = ls.start[l + 1] - 1; q >= ls.start[l]; ) {
            const int f = f_of(q);
            if (f > 63) { // LDS by the size class of the front (the largest front of the class), so that runs of one class share a launch
                const int k = big_kind(f), fc = k == 4 ? 79 : (k == 2 ? 111 : 159);
                push(g->wg_b, g->seg_b, q, k, l, backsolve_tab_lds_bytes(k, fc, 0), 1); --q; }
            else { int cnt = 1; while (cnt < 4 && q - cnt >= ls.start[l] && f_of(q - cnt) <= 63) ++cnt;
                push(g->wg_b, g->seg_b, q, 0 | (cnt << 8), l, backsolve_tab_lds_bytes(0, g->small_max_f, g->small_max_npiv), 0); q -= cnt; } }
    // ---- pose-window shards: the SHARED top of a plan with workgroup fronts (round 4).  Three tables over the shared level positions
    // (shared_base + q): contributions (mode CONTRIB: no dependencies among them; a small front a wave — the four-wave form has no such
    // mode —, a big one a workgroup), the top itself (mode TOP, children first: a small front four waves, a big one a workgroup), and
    // the backward solve (root first).
    { const gs_graph::LevelSet &sh = g->shared; const int nls = (int)sh.start.size() - 1, B0 = g->shared_base;
      auto fs = [&](int q) { const Front &F = P.fronts[P.level_fronts_shared[q]]; return F.npiv + F.nbnd; };
      for (int l = 0; l < nls; ++l)
          for (int q = sh.start[l]; q < sh.start[l + 1]; ) { const int f = fs(q);
              if (f > 63) { const int k = big_kind(f);
                  push(g->wgs_c, g->segs_c, B0 + q, k, l, factor_tab_lds_bytes(k), fcls(k)); push(g->wgs_t, g->segs_t, B0 + q, k, l, factor_tab_lds_bytes(k), fcls(k)); ++q; }
              else { push(g->wgs_t, g->segs_t, B0 + q, 1 | (1 << 8), l, factor_tab_lds_bytes(1), 1);
                  int cnt = 1; while (cnt < 4 && q + cnt < sh.start[l + 1] && fs(q + cnt) <= 63) ++cnt;
                  push(g->wgs_c, g->segs_c, B0 + q, 0 | (cnt << 8), l, factor_tab_lds_bytes(0), 1);
                  for (int k2 = 1; k2 < cnt; ++k2) push(g->wgs_t, g->segs_t, B0 + q + k2, 1 | (1 << 8), l, factor_tab_lds_bytes(1), 1);
                  q += cnt; } }
      for (int l = nls - 1; l >= 0; --l)
          for (int q = sh.start[l + 1] - 1; q >= sh.start[l]; ) { const int f = fs(q);
              if (f > 63) { const int k = big_kind(f), fc = k == 4 ? 79 : (k == 2 ? 111 : 159); push(g->wgs_b, g->segs_b, B0 + q, k, l, backsolve_tab_lds_bytes(k, fc, 0), 1); --q; }
              else { int cnt = 1; while (cnt < 4 && q - cnt >= sh.start[l] && fs(q - cnt) <= 63) ++cnt;
                  push(g->wgs_b, g->segs_b, B0 + q, 0 | (cnt << 8), l, backsolve_tab_lds_bytes(0, g->small_max_f, g->small_max_npiv), 0); q -= cnt; } } }
    int rc;
    auto up = [&](int2 **dst, const std::vector<int32_t> &v) -> int {
        int r2 = dev_alloc(g, (int32_t **)dst, v.size()); if (r2 != GS_OK) return r2;
        if (!v.empty()) { hipError_t e = hipMemcpyAsync(*dst, v.data(), v.size() * sizeof(int32_t), hipMemcpyHostToDevice, g->stream); if (e != hipSuccess) return fail(GS_ERR_HIP, hipGetErrorString(e)); }   // the host vectors live on the handle
        return GS_OK; };
    if ((rc = up(&g->d_wg_f, g->wg_f)) != GS_OK || (rc = up(&g->d_wg_b, g->wg_b)) != GS_OK || (rc = up(&g->d_wgs_c, g->wgs_c)) != GS_OK ||
        (rc = up(&g->d_wgs_t, g->wgs_t)) != GS_OK || (rc = up(&g->d_wgs_b, g->wgs_b)) != GS_OK) return rc;
    return GS_OK;
}
// launches = maximal runs of table entries with the same LDS need (whole-tree mode: across levels; after a flag timeout: never
// across a level, so that no workgroup waits for one of its own launch)
template <class Launch> static void for_each_run(const std::vector<gs_graph::WgSeg> &segs, bool across_levels, Launch &&fn) {
    for (size_t i = 0; i < segs.size(); ) {
        size_t j = i + 1; int n = segs[i].count;
        while (j < segs.size() && segs[j].lds == segs[i].lds && segs[j].cls == segs[i].cls && (across_levels || segs[j].level == segs[i].level)) { n += segs[j].count; ++j; }
        fn(segs[i].first, n, segs[i].lds, segs[i].cls);
        i = j; }
}
static bool ensure_big_tables(gs_graph *g) {
    // the workgroup tables of a plan with fronts beyond a wave are built on first use; if that fails (device memory), NO solver launch of
    // this iteration may run and its update must not be applied: the failure is kept on the handle (enqueue_rc: every entry point that
    // enqueues iterations returns it) and raised on the device like a failed solve, so that k_update applies nothing
    if (g->d_wg_f) return true;
    const int rc = build_big_tables(g, g->own);
    if (rc == GS_OK) return true;
    g->enqueue_rc = rc; g->enqueue_err = g_last_error; g->d_wg_f = g->d_wg_b = g->d_wgs_c = g->d_wgs_t = g->d_wgs_b = nullptr;
    const int32_t one = 1; hipMemcpyAsync(g->d.fail, &one, sizeof(int32_t), hipMemcpyHostToDevice, g->stream); hipStreamSynchronize(g->stream);
    return false;
}
static void enqueue_factor_big(gs_graph *g, const gs_graph::LevelSet &ls, bool tree) {
    if (!ensure_big_tables(g)) return;
    if (g->leaf_n > 0) launch_factor_tree(g->d, g->leaf_n, g->leaf_slot, g->leaf_max_f, g->leaf_n, 0, 0, 0, g->stream);       // the leaf instance alone
    // "no flags to wait for at level 1" holds only if EVERY leaf went through the leaf launch (big leaves share the table launch with their parents)
    const int leaf_pre = (g->leaf_n > 0 && g->leaf_n == ls.start[1]) ? 1 : 0;
    for_each_run(g->seg_f, tree, [&](int first, int n, size_t lds, int cls) { launch_factor_tab(g->d, g->d_wg_f + first, n, leaf_pre, lds, cls, g->stream); });
}
static void enqueue_backsolve_big(gs_graph *g, const gs_graph::LevelSet &, bool tree) {
    if (!g->d_wg_b) return;
    for_each_run(g->seg_b, tree, [&](int first, int n, size_t lds, int cls) { launch_backsolve_tab(g->d, g->d_wg_b + first, n, g->small_max_npiv, g->small_max_f, lds, cls, g->stream); });
}
// the shared top of a sharded plan with workgroup fronts: contributions (mode 1), the top from the exchange (mode 2), its backward solve
static void enqueue_shared_big(gs_graph *g, int what) {
    if (!ensure_big_tables(g)) return;
    const bool tree = g->d.tree != 0;
    if (what == 1) for_each_run(g->segs_c, true, [&](int first, int n, size_t lds, int cls) { launch_factor_tab(g->d, g->d_wgs_c + first, n, 0, lds, cls, g->stream, 1); });
    else if (what == 2) for_each_run(g->segs_t, tree, [&](int first, int n, size_t lds, int cls) { launch_factor_tab(g->d, g->d_wgs_t + first, n, 0, lds, cls, g->stream, 2); });
    else for_each_run(g->segs_b, tree, [&](int first, int n, size_t lds, int cls) { launch_backsolve_tab(g->d, g->d_wgs_b + first, n, g->small_max_npiv, g->small_max_f, lds, cls, g->stream); });
}
static void enqueue_factor_levels(gs_graph *g, const gs_graph::LevelSet &ls, int base, int mode) {
    const int nlev = (int)ls.start.size() - 1;
    if (g->d.factor_variant == 3 && g->d.tree && mode == 0 && base == 0 && nlev > 0) {     // every own level in one launch
        ++g->d.epoch;
        // leaf instance: level 0 only if its fronts really have no children (always true for an elimination tree's level 0)
        if (g->leaf_n < 0) {                                          // once per plan
            int n_leaf = ls.start[1], F_leaf_all = 0, slot = 256; g->leaf_max_f = 0;
            // leaves beyond a wave (the fronts of a level are sorted by size class: the small ones first) go to the table-driven launch
            for (int q = 0; q < n_leaf; ++q) { const Front &F = g->plan.fronts[g->plan.level_fronts_owned[q]]; if (F.npiv + F.nbnd > 63) { n_leaf = q; break; } }
            for (int q = 0; q < n_leaf; ++q) { const Front &F = g->plan.fronts[g->plan.level_fronts_owned[q]];
                g->leaf_max_f = std::max(g->leaf_max_f, F.npiv + F.nbnd);
                if (F.child_cnt != 0) { n_leaf = 0; break; }
                slot = std::max(slot, (((F.npiv + F.nbnd + 1) | 1) * F.npiv + 1) & ~1); }
            F_leaf_all = n_leaf;                                        // GS_LEAF_KERNEL=2: leaf launches whatever their number
            // few leaves (all resident at once anyway: <= GS_LEAF_MIN, default 2048): no separate leaf launches, the whole-tree
            // launches take level 0 as well — two kernel boundaries less per iteration (cfg1-cfg3: 6-11 % of it)
            if (n_leaf <= g->opt.leaf_min) n_leaf = 0;
            if (g->opt.leaf_kernel == 0) n_leaf = 0; else if (g->opt.leaf_kernel == 2) n_leaf = F_leaf_all;
            g->leaf_n = n_leaf; g->leaf_slot = slot;
            // the bottom subtrees (k_factor3_sub): every level-1 front of this rank with the leaves below it in one workgroup — their update
            // matrices never leave the chip.  Taken when the leaf instance is in use, the plan put the leaves under
            // level-1 fronts behind the others (gs_plan.cpp) and the workgroup's LDS fits; the leaf launch then covers positions [0, sub_free).
            g->sub_n = 0; g->sub_first = 0; g->sub_free = n_leaf;
            if (g->opt.subtree != 0 && n_leaf > 0 && n_leaf == ls.start[1] && g->plan.max_front <= 63 && nlev >= 2 &&
                factor_sub_lds_bytes(slot) <= (size_t)160 * 1024) {
                const Plan &P = g->plan; const auto &lfo = P.level_fronts_owned;
                auto under = [&](int s) { const int pa = P.fronts[s].parent; return pa >= 0 && P.fronts[pa].level == 1 && g->pos_of_front[pa] >= ls.start[1] && g->pos_of_front[pa] < ls.start[2]; };
                int nfree = 0; while (nfree < n_leaf && !under(lfo[nfree])) ++nfree;
                bool ok = true; int64_t kids = 0;
                for (int q = nfree; q < n_leaf && ok; ++q) ok = under(lfo[q]);
                for (int q = ls.start[1]; q < ls.start[2] && ok; ++q) { const Front &F = P.fronts[lfo[q]]; kids += F.child_cnt;
                    for (int c = 0; c < F.child_cnt && ok; ++c) { const int cp = g->pos_of_front[P.children[F.child_off + c]]; ok = cp >= nfree && cp < n_leaf; } }
                if (ok && kids == n_leaf - nfree && ls.start[2] > ls.start[1]) { g->sub_first = ls.start[1]; g->sub_n = ls.start[2] - ls.start[1]; g->sub_free = nfree; } } }
        // the upper levels — few fronts, all of them in the dependent chain — get four waves per front: whole levels from the
        // top down while a level has at most GS_BLOCK_FRONTS (512) fronts (those workgroups are all resident at once)
        if (g->block_n < 0) { const int thr = g->opt.block_fronts;
            int nb = 0;
            const int lowest = g->sub_n > 0 ? 2 : (g->leaf_n > 0 ? 1 : 0);      // the first level of the flagged launch
            for (int l = nlev - 1; l >= lowest; --l) { const int nl = ls.start[l + 1] - ls.start[l];
                if (nl > thr) break;
                nb += nl; }
            g->block_n = std::min(nb, ls.start[nlev] - (g->sub_n > 0 ? g->sub_first + g->sub_n : std::max(g->leaf_n, 0))); }
        if (g->plan.max_front > 63) { enqueue_factor_big(g, ls, true); return; }
        launch_factor_tree(g->d, g->sub_n > 0 ? g->sub_free : g->leaf_n, g->leaf_slot, g->leaf_max_f, ls.start[nlev], g->block_n, g->sub_first, g->sub_n, g->stream); return; }
    if (g->d.factor_variant == 3 && !g->d.tree && mode == 0 && base == 0 && nlev > 0 && g->plan.max_front > 63) { ++g->d.epoch; enqueue_factor_big(g, ls, false); return; }
    if (g->d.factor_variant == 3 && mode == 2 && nlev > 0 && ls.start[nlev] > 0 && g->plan.max_front > 63) { enqueue_shared_big(g, 2); return; }     // ... of a plan with workgroup fronts: table-driven
    if (g->d.factor_variant == 3 && g->d.tree && mode == 2 && nlev > 0 && ls.start[nlev] > 0) {     // the shared top of a sharded graph, one flagged launch
        launch_factor_tree_top(g->d, base, ls.start[nlev], g->stream); return; }
    for (int l = 0; l < nlev; ++l)
        launch_factor_level(g->d, base + ls.start[l], ls.start[l + 1] - ls.start[l], ls.max_f[l], mode, g->stream);
}
static void enqueue_backsolve_levels(gs_graph *g, const gs_graph::LevelSet &ls, int base) {
    const int nlev = (int)ls.start.size() - 1;
    if (g->d.factor_variant == 3 && base == 0 && nlev > 0 && g->plan.max_front > 63) { enqueue_backsolve_big(g, ls, g->d.tree != 0); return; }
    if (g->d.factor_variant == 3 && base != 0 && nlev > 0 && ls.start[nlev] > 0 && g->plan.max_front > 63) { enqueue_shared_big(g, 3); return; }
    if (g->d.factor_variant == 3 && g->d.tree && base == 0 && nlev > 0) {
        // levels >= 1 in one launch (fronts wait for their parent's flag), then the leaf level on its own: by then every
        // parent is done, so it needs no flags, and its LDS slot is sized for the leaves alone (more resident waves)
        // The flagged launch is register-heavy (each lane preloads its L columns: 2 waves per SIMD) — right for the chain
        // of the upper levels (2.2 us per level), wrong for the wide levels at the bottom, which are bound by resident
        // waves x bytes: levels of more than GS_BS_WIDE (2048) fronts run one light launch each, like the leaves
        // (measured per-level completion times: scripts/level_times.py).
        const int wide = g->opt.bs_wide;
        int l0 = 0;
        if (g->leaf_n != 0) while (l0 + 1 < nlev && ls.start[l0 + 1] - ls.start[l0] > wide) ++l0;
        if (l0 == 0 && nlev > 1 && g->leaf_n != 0) l0 = 1;
        int mn = 0, mf = 0; for (int l = l0; l < nlev; ++l) { mn = std::max(mn, ls.max_npiv[l]); mf = std::max(mf, ls.max_f[l]); }
        launch_backsolve_tree(g->d, ls.start[l0], ls.start[nlev] - ls.start[l0], mn, mf, g->stream);
        for (int l = l0 - 1; l >= 0; --l) launch_backsolve_level(g->d, ls.start[l], ls.start[l + 1] - ls.start[l], ls.max_npiv[l], ls.max_nbnd[l], g->stream);
        return; }
    if (g->d.factor_variant == 3 && g->d.tree && base != 0 && nlev > 0 && ls.start[nlev] > 0) {      // shared top: one flagged launch, root first
        int mn = 0, mf = 0; for (int l = 0; l < nlev; ++l) { mn = std::max(mn, ls.max_npiv[l]); mf = std::max(mf, ls.max_f[l]); }
        launch_backsolve_tree(g->d, base, ls.start[nlev], mn, mf, g->stream); return; }
    for (int l = nlev - 1; l >= 0; --l)
        launch_backsolve_level(g->d, base + ls.start[l], ls.start[l + 1] - ls.start[l], ls.max_npiv[l], ls.max_nbnd[l], g->stream);
}
// pose-window shards, first half: linearise this shard's edges, factorise its own subtrees, write its contribution
// to every shared front into the exchange buffer (the caller all-reduces that buffer: RCCL sum, fp64)
static void enqueue_local(gs_graph *g, bool timed) {
    ++g->d.iter;                                                     // kernels see the iteration they belong to (fault injection, gs_debug_fail_at_iteration)
    if (timed) hipEventRecord(g->ev[0], g->stream);
    launch_linearize(g->d, g->stream, g->ev_lin[0], g->ev_lin[1]);   // (null outside gs_time_iterations' second pass)
    launch_linearize_tail(g->d, g->stream);                          // a grown plan's tail (no launch without one)
    if (timed) hipEventRecord(g->ev[1], g->stream);
    enqueue_factor_levels(g, g->own, 0, 0);
    const int nshared = (int)g->plan.level_fronts_shared.size();
    if (nshared > 0 && g->d.factor_variant == 3 && g->plan.max_front > 63) enqueue_shared_big(g, 1);      // a plan with workgroup fronts: table-driven
    else if (nshared > 0) { int mf = 0; for (int v : g->shared.max_f) mf = std::max(mf, v);
        launch_factor_level(g->d, g->shared_base, nshared, mf, 1, g->stream); }
}
// second half: the shared top (redundantly on every rank), backward solve top-down, update
static void enqueue_finish(gs_graph *g, bool timed) {
    enqueue_factor_levels(g, g->shared, g->shared_base, 2);
    if (timed) hipEventRecord(g->ev[2], g->stream);
    enqueue_backsolve_levels(g, g->shared, g->shared_base);
    enqueue_backsolve_levels(g, g->own, 0);
    if (timed) hipEventRecord(g->ev[3], g->stream);
    launch_update(g->d, g->stream);
    if (timed) hipEventRecord(g->ev[4], g->stream);
    g->dev_estimates_newer = true;
}
static void enqueue_iteration(gs_graph *g, bool timed) { enqueue_local(g, timed); enqueue_finish(g, timed); }
// a failure of the enqueue itself (not of the arithmetic): reported once by the entry point that enqueued
static int take_enqueue_error(gs_graph *g) {
    if (g->enqueue_rc == GS_OK) return GS_OK;
    const int rc = g->enqueue_rc; g->enqueue_rc = GS_OK;
    return fail(rc, "solver launch tables: " + g->enqueue_err + " (no update applied)");
}

extern "C" int gs_iterate(gs_graph *g) {
    if (!g) return fail(GS_ERR_INVALID, "null graph");
    if (!g->dev_valid || g->plan_version != g->h.structure_version) return fail(GS_ERR_NOT_INITIALIZED, "call gs_initialize_optimization first");
    if (g->plan.dist) return fail(GS_ERR_INVALID, "sharded graph: use gs_dist_iterate (RCCL inside the library) or gs_dist_iterate_local / all-reduce / gs_dist_iterate_finish");
    int rc = ensure_device(g); if (rc != GS_OK) return rc;
    enqueue_iteration(g, false);
    if (g->enqueue_rc != GS_OK) return take_enqueue_error(g);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(GS_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
    return 1;
}

static void fill_plan_stats(gs_graph *g, gs_stats *s) {
    const Plan &P = g->plan;
    s->n_free_poses = 0; s->n_free_landmarks = 0;
    for (auto v : P.pose_gidx) s->n_free_poses += v >= 0;
    for (auto v : P.lm_gidx) s->n_free_landmarks += v >= 0;
    s->n_odometry_edges = g->h.n_pp(); s->n_observation_edges = g->h.n_pl();
    s->n_fronts = (int32_t)P.fronts.size(); s->n_levels = (int32_t)P.level_start.size() - 1; s->max_front = P.max_front;
    s->factor_flops = P.factor_flops; s->factor_bytes = (P.l_doubles + P.u_doubles) * 8; s->ms_structure = g->ms_structure;
    s->fell_back = g->fell_back ? 1 : 0;
    s->factor_variant = g->dev_valid ? (g->d.factor_variant == 0 ? 4 : g->d.factor_variant) : 0;
    s->n_big_fronts = 0;
    for (const Front &F : P.fronts) s->n_big_fronts += (!F.opaque && F.npiv + F.nbnd > 63);
    s->device_bytes = (int64_t)g->pool_total; s->ms_plan_host = P.ms_build; s->n_growths = P.n_growths;
    s->n_own_fronts = (int32_t)P.level_fronts_owned.size(); s->n_shared_fronts = (int32_t)P.level_fronts_shared.size();
    s->n_subtrees = g->leaf_n >= 0 ? g->sub_n : 0;
}

extern "C" int gs_get_stats(gs_graph *g, gs_stats *s) {
    if (!g || !s) return fail(GS_ERR_INVALID, "null argument");
    if (!g->plan.valid) return fail(GS_ERR_NOT_INITIALIZED, "no plan built");
    std::memset(s, 0, sizeof(*s)); s->struct_size = (int32_t)sizeof(*s);
    fill_plan_stats(g, s);
    return GS_OK;
}

// gs_optimize (rel_tol < 0: the reference's fixed iteration count) and gs_optimize_until (rel_tol >= 0: the stop rule)
static int optimize_impl(gs_graph *g, int32_t iterations, double rel_tol, gs_stats *stats) {
    if (!g) return fail(GS_ERR_INVALID, "null graph");
    if (iterations < 0) return fail(GS_ERR_INVALID, "negative iteration count");
    if (g->world > 1 || g->opt.force_shared_top > 0) return fail(GS_ERR_INVALID, "sharded graph: use gs_dist_optimize (RCCL inside the library), or drive gs_dist_iterate_local / all-reduce / gs_dist_iterate_finish");
    // g2o: optimize() is always preceded by initializeOptimization() (reference src/slam.cpp:480-481);
    // the plan is rebuilt only when the structure changed since the last call.
    int rc = ensure_ready(g); if (rc != GS_OK) return rc;
    // A handle that fell back to one launch per level (a whole-tree launch gave up on a front's flag) does not stay there until the next plan:
    // what makes a flag late — the chip shared with another process, a debugger, a profiler replaying kernels — passes.  After 4 calls on the
    // slow path the whole-tree launches are tried again (first iteration on its own, like a new plan's); another timeout quadruples the wait
    // (16, 64, ... 1024 calls), a clean launch ends the episode.
    if (g->fell_back && iterations > 0 && g->opt.tree != 0 && !g->d.tree && ++g->fallback_calls >= g->fallback_retry_after) {
        g->d.tree = 1; g->tree_proven = false; g->fallback_calls = 0; g->fallback_retrying = true; }
    { const int ii = g->d.inject_iter, ic = g->d.inject_code;       // an armed fault injection survives the reset below
      HIP_TRY(hipMemsetAsync(g->d.fail, 0, 4 * sizeof(int32_t), g->stream)); g->d.inject_iter = ii; g->d.inject_code = ic; }
    const bool until = rel_tol >= 0.0;
    g->d.conv_tol = until ? rel_tol : -1.0;
    if (until) { const double none = -1.0; HIP_TRY(hipMemcpyAsync(g->d.chi2 + 70, &none, sizeof(double), hipMemcpyHostToDevice, g->stream)); }
    hipEventRecord(g->ev[5], g->stream);
    const int nh = std::min(iterations, 64);
    // All iterations are enqueued up front (no host round trip between them).  g2o leaves its loop at the first failed
    // solve and keeps the previous iterate: k_update applies nothing once the failure flag is up, and fail[1] says how
    // many updates went in.  A flag timeout of a whole-tree launch (code 2) is not a property of H: the handle falls
    // back to one launch per level and runs the remaining iterations again from the last good iterate.
    // Stop rule (gs_optimize_until): k_update compares the chi2 of consecutive linearisation points on the device and
    // raises fail[2]; later updates are skipped like after a failure.  The host enqueues chunks of 4 iterations and
    // looks at the flags in between, so at most 3 enqueued iterations run as no-ops after convergence.
    // Chunks: the FIRST iteration on its own, then groups of 8 — a remainder of up to 12 in one — (stop rule: 4).  A whole-tree launch whose flag hand-off fails
    // (its pollers are bounded and leave at once when any front has reported a failure, so such a launch drains in one poll
    // budget, ~30 ms) would otherwise have every remaining iteration queued up behind it, each paying the same again: with
    // chunks a timeout costs one chunk before the per-level fallback takes over.  One host round trip per chunk.
    int applied = 0, enq = 0, first_failure = 0; int32_t ff[4] = {0, 0, 0, 0}; bool fell_back = false;
    while (enq < iterations) {
        // (a remainder of up to 12 goes out as one chunk: the reference's optimize(10) is 1 + 9, two host round trips instead of three)
        // (the FIRST iteration goes out alone only until a whole-tree launch of THIS plan has come back clean once: the flag hand-off
        // depends on the launch geometry, not on the numbers — a repeated optimize(10), the reference's quirk path, is one host round trip)
        const bool alone = enq == 0 && !(g->tree_proven && g->d.tree);
        const int upto = std::min(iterations, alone ? 1 : (until ? enq + 4 : (iterations - enq <= 12 ? iterations : enq + 8)));
        for (int it = enq; it < upto; ++it) {
            g->d.hist_slot = it < nh ? it : -1;                      // k_update files the chi2 of this iteration's linearisation point itself
            enqueue_iteration(g, false);
        }
        g->d.hist_slot = -1;
        if (g->enqueue_rc != GS_OK) { g->d.conv_tol = -1.0; hipStreamSynchronize(g->stream); reset_failure(g); return take_enqueue_error(g); }
        enq = upto;
        HIP_TRY(hipMemcpyAsync(ff, g->d.fail, sizeof(ff), hipMemcpyDeviceToHost, g->stream));
        HIP_TRY(hipStreamSynchronize(g->stream));
        applied = ff[1];
        if (ff[0] == 0 && g->d.tree) { g->tree_proven = true;
            if (g->fallback_retrying) { g->fallback_retrying = false; g->fell_back = false; g->fallback_retry_after = 4; } }     // back on the whole-tree launches
        if (ff[0] != 0 && first_failure == 0) first_failure = ff[0];
        if (ff[0] == 2 && g->d.tree && !fell_back) {
            if (g->fallback_retrying) { g->fallback_retrying = false; g->fallback_retry_after = std::min(g->fallback_retry_after * 4, 1024); }
            g->fallback_calls = 0;
            g->d.tree = 0; fell_back = true; g->fell_back = true; g->d.inject_iter = 0;
            HIP_TRY(hipMemsetAsync(g->d.fail, 0, sizeof(int32_t), g->stream));      // the code only: the update count goes on
            enq = applied; ff[0] = 0; continue; }
        if (ff[0] != 0 || ff[2] != 0) break;
    }
    g->d.conv_tol = -1.0;
    if (until) HIP_TRY(hipMemsetAsync(g->d.fail + 2, 0, sizeof(int32_t), g->stream));     // the stop flag must not gate later gs_iterate calls
    const int nshow = std::min(applied, nh);
    if (g->cfg.verbose || stats) { launch_chi2_only(g->d, g->stream);
        hipMemcpyAsync(g->d.chi2 + 1 + nh, g->d.chi2, sizeof(double), hipMemcpyDeviceToDevice, g->stream); }
    hipEventRecord(g->ev[6], g->stream);
    double hist[80];
    HIP_TRY(hipMemcpyAsync(hist, g->d.chi2, sizeof(hist), hipMemcpyDeviceToHost, g->stream));
    // the estimates come back with the same wait (on failure: the last good iterate, what g2o's vertices hold)
    const bool pull = g->dev_valid && g->dev_estimates_newer;
    if (pull) { const size_t Np = (size_t)(g->d.N + g->d.tN), Mp = (size_t)(g->d.M + g->d.tM);
        if (Np) HIP_TRY(hipMemcpyAsync(g->h.pose_est.data(), g->d.pose_est, Np * 3 * sizeof(double), hipMemcpyDeviceToHost, g->stream));
        if (Mp) HIP_TRY(hipMemcpyAsync(g->h.lm_est.data(), g->d.lm_est, Mp * 2 * sizeof(double), hipMemcpyDeviceToHost, g->stream)); }
    HIP_TRY(hipStreamSynchronize(g->stream));
    if (pull) g->dev_estimates_newer = false;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { reset_failure(g); return fail(GS_ERR_HIP, std::string("iteration: ") + hipGetErrorString(e)); }      // (a launch that never ran: the ticket counter and its running sum start again)
    float ms = 0; hipEventElapsedTime(&ms, g->ev[5], g->ev[6]);
    if (g->cfg.verbose) for (int it = 0; it < nshow; ++it)  // g2o prints the chi2 AFTER the update of iteration it
        std::fprintf(stderr, "iteration= %d\t chi2= %.6f\t edges= %d\t schur= 0\n", it, it + 1 < applied ? hist[2 + it] : hist[1 + nh], g->h.n_pp() + g->h.n_pl());
    if (stats) { std::memset(stats, 0, sizeof(*stats)); stats->struct_size = (int32_t)sizeof(*stats);
        fill_plan_stats(g, stats); stats->iterations = applied; stats->numeric_failure = ff[0]; stats->first_failure = first_failure;
        stats->chi2_initial = iterations > 0 ? hist[1] : hist[1 + nh]; stats->chi2_final = hist[1 + nh]; stats->ms_total = ms; }
    if (ff[0]) { rc = reset_failure(g); if (rc != GS_OK) return rc; }
    if (ff[0] == 2) { g_last_error = "a front's completion flag did not arrive in time, with one launch per level as well"; return 0; }
    if (ff[0]) { g_last_error = "zero pivot: H is singular (g2o: optimize() returns 0, the vertices keep the last good iterate)"; return 0; }
    return applied;
}
extern "C" int gs_optimize(gs_graph *g, int32_t iterations, gs_stats *stats) { return optimize_impl(g, iterations, -1.0, stats); }
extern "C" int gs_optimize_until(gs_graph *g, int32_t max_iterations, double rel_chi2_tol, gs_stats *stats) {
    if (!(rel_chi2_tol >= 0.0)) return fail(GS_ERR_INVALID, "rel_chi2_tol must be >= 0");
    return optimize_impl(g, max_iterations, rel_chi2_tol, stats);
}

extern "C" int gs_chi2(gs_graph *g, double *out) {
    if (!g || !out) return fail(GS_ERR_INVALID, "null argument");
    int rc = ensure_ready(g); if (rc != GS_OK) return rc;
    launch_chi2_only(g->d, g->stream);
    HIP_TRY(hipMemcpyAsync(out, g->d.chi2, sizeof(double), hipMemcpyDeviceToHost, g->stream));
    HIP_TRY(hipStreamSynchronize(g->stream));
    return GS_OK;
}

// ------------------------------------------------------------------ measurement / parity hooks
extern "C" int gs_linearize(gs_graph *g) {
    if (!g) return fail(GS_ERR_INVALID, "null graph");
    int rc = ensure_ready(g); if (rc != GS_OK) return rc;
    launch_linearize(g->d, g->stream);
    launch_linearize_tail(g->d, g->stream);
    launch_linearize_finalize(g->d, g->stream);              // stand-alone pass: materialise H_ll, b_l, chi2 for export
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(GS_ERR_HIP, std::string("linearize: ") + hipGetErrorString(e));
    return GS_OK;
}
extern "C" int gs_time_linearize(gs_graph *g, int32_t reps, double *out_ms) {
    if (!g || !out_ms || reps <= 0) return fail(GS_ERR_INVALID, "bad argument");
    int rc = ensure_ready(g); if (rc != GS_OK) return rc;
    launch_linearize(g->d, g->stream);                       // warm
    hipEventRecord(g->ev[0], g->stream);
    for (int r = 0; r < reps; ++r) launch_linearize(g->d, g->stream);
    hipEventRecord(g->ev[1], g->stream);
    HIP_TRY(hipEventSynchronize(g->ev[1]));
    float ms = 0; HIP_TRY(hipEventElapsedTime(&ms, g->ev[0], g->ev[1]));
    *out_ms = (double)ms / reps;
    return GS_OK;
}
extern "C" int gs_debug_front_times(gs_graph *g, int64_t *out, int64_t capacity) {
    if (!g || !out) return fail(GS_ERR_INVALID, "null argument");
    if (!g->dev_valid) return fail(GS_ERR_NOT_INITIALIZED, "nothing on the device yet");
    const int64_t n = 2 * (int64_t)g->plan.fronts.size();
    if (capacity < n) return fail(GS_ERR_CAPACITY, "buffer too small");
    HIP_TRY(hipMemcpyAsync(out, g->d.done_ts, (size_t)n * sizeof(int64_t), hipMemcpyDeviceToHost, g->stream));
    HIP_TRY(hipStreamSynchronize(g->stream));
    return (int)(n / 2);
}
extern "C" int gs_debug_timestamps(gs_graph *g, int64_t *out64) {
    if (!g || !out64) return fail(GS_ERR_INVALID, "null argument");
    if (!g->dev_valid) return fail(GS_ERR_NOT_INITIALIZED, "nothing on the device yet");
    HIP_TRY(hipMemcpyAsync(out64, g->d.dbg_ts, 64 * sizeof(int64_t), hipMemcpyDeviceToHost, g->stream));
    HIP_TRY(hipStreamSynchronize(g->stream));
    return GS_OK;
}
extern "C" int64_t gs_linearize_bytes(gs_graph *g) {
    if (!g) return 0;
    return (int64_t)g->h.n_pp() * 152 + (int64_t)g->h.n_pl() * 96 + (int64_t)g->h.n_poses() * 120 + (int64_t)g->h.n_lms() * 64;
}
extern "C" int gs_export_system(gs_graph *g, double *Hpp_diag, double *Hll_diag, double *Hpp_off, double *Hpl,
                                double *b_pose, double *b_lm, int32_t *pp_order, int32_t *pl_order) {
    if (!g) return fail(GS_ERR_INVALID, "null graph");
    if (!g->dev_valid) return fail(GS_ERR_NOT_INITIALIZED, "nothing linearised yet");
    if (g->d.tN > 0) return fail(GS_ERR_INVALID, "the plan has grown by appended poses: their blocks live in the tail arenas, which this export does not read (gs_initialize_optimization with GS_GROW=0 rebuilds)");
    int rc = ensure_device(g); if (rc != GS_OK) return rc;
    const DevGraph &d = g->d;
    // the device keeps these arrays structure-of-arrays (and the diagonal blocks packed symmetric); the
    // export format is array-of-blocks, full and row-major
    const size_t N = (size_t)d.N, M = (size_t)d.M, Epp = (size_t)d.Epp, Epl = (size_t)d.Epl;
    const size_t L = (size_t)d.ell_len;
    std::vector<double> t0(N * 6), t1(M * 3), t2(Epp * 9), t3(L * 6), t4(N * 3), t5(M * 2);
    auto dl = [&](std::vector<double> &dst, const double *src) -> hipError_t {
        return dst.empty() ? hipSuccess : hipMemcpyAsync(dst.data(), src, dst.size() * sizeof(double), hipMemcpyDeviceToHost, g->stream); };
    HIP_TRY(dl(t0, d.Hpp_diag)); HIP_TRY(dl(t1, d.Hll_diag)); HIP_TRY(dl(t2, d.Hpp_off)); HIP_TRY(dl(t3, d.Hpl));
    HIP_TRY(dl(t4, d.b_pose)); HIP_TRY(dl(t5, d.b_lm));
    HIP_TRY(hipStreamSynchronize(g->stream));
    static const int sym3[3][3] = {{0, 1, 2}, {1, 3, 4}, {2, 4, 5}}, sym2[2][2] = {{0, 1}, {1, 2}};
    if (Hpp_diag) for (size_t p = 0; p < N; ++p) for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Hpp_diag[9 * p + 3 * r + c] = t0[sym3[r][c] * N + p];
    if (Hll_diag) for (size_t l = 0; l < M; ++l) for (int r = 0; r < 2; ++r) for (int c = 0; c < 2; ++c) Hll_diag[4 * l + 2 * r + c] = t1[sym2[r][c] * M + l];
    if (Hpp_off) for (size_t k = 0; k < Epp; ++k) for (int c = 0; c < 9; ++c) Hpp_off[9 * k + c] = t2[c * Epp + k];
    if (Hpl) for (size_t k = 0; k < Epl; ++k) { const int32_t e = g->plan.ell_of_ins[k];                 // insertion order; an edge outside this rank's layout: zeros
        for (int c = 0; c < 6; ++c) Hpl[6 * k + c] = e >= 0 ? t3[c * L + (size_t)e] : 0.0; }
    if (b_pose) for (size_t p = 0; p < N; ++p) for (int c = 0; c < 3; ++c) b_pose[3 * p + c] = t4[c * N + p];
    if (b_lm) for (size_t l = 0; l < M; ++l) for (int c = 0; c < 2; ++c) b_lm[2 * l + c] = t5[c * M + l];
    if (pp_order) std::memcpy(pp_order, g->plan.pp_order.data(), g->plan.pp_order.size() * sizeof(int32_t));
    if (pl_order) for (size_t k = 0; k < Epl; ++k) pl_order[k] = (int32_t)k;                 // exported in insertion order
    return GS_OK;
}
extern "C" int gs_export_delta(gs_graph *g, double *dpose, double *dlm) {
    if (!g) return fail(GS_ERR_INVALID, "null graph");
    if (!g->dev_valid) return fail(GS_ERR_NOT_INITIALIZED, "no iteration run yet");
    int rc = ensure_device(g); if (rc != GS_OK) return rc;
    if (dpose && g->d.N) HIP_TRY(hipMemcpyAsync(dpose, g->d.dpose, (size_t)(g->d.N + g->d.tN) * 3 * sizeof(double), hipMemcpyDeviceToHost, g->stream));
    if (dlm && g->d.M) HIP_TRY(hipMemcpyAsync(dlm, g->d.dlm, (size_t)(g->d.M + g->d.tM) * 2 * sizeof(double), hipMemcpyDeviceToHost, g->stream));
    HIP_TRY(hipStreamSynchronize(g->stream));
    return GS_OK;
}
extern "C" int gs_time_iterations(gs_graph *g, int32_t reps, gs_stats *s) {
    if (!g || !s || reps <= 0) return fail(GS_ERR_INVALID, "bad argument");
    if (g->world > 1 || g->opt.force_shared_top > 0) return fail(GS_ERR_INVALID, "sharded graph: time the two halves from the caller");
    int rc = ensure_ready(g); if (rc != GS_OK) return rc;
    const int N = g->d.N + g->d.tN, M = g->d.M + g->d.tM;
    double *sp = nullptr, *sl = nullptr;                        // save estimates
    HIP_TRY(hipMalloc((void **)&sp, std::max<size_t>((size_t)N * 3, 1) * sizeof(double)));
    HIP_TRY(hipMalloc((void **)&sl, std::max<size_t>((size_t)M * 2, 1) * sizeof(double)));
    hipMemcpyAsync(sp, g->d.pose_est, (size_t)N * 3 * sizeof(double), hipMemcpyDeviceToDevice, g->stream);
    hipMemcpyAsync(sl, g->d.lm_est, (size_t)M * 2 * sizeof(double), hipMemcpyDeviceToDevice, g->stream);
    const bool newer = g->dev_estimates_newer;
    std::memset(s, 0, sizeof(*s)); s->struct_size = (int32_t)sizeof(*s); fill_plan_stats(g, s);
    enqueue_iteration(g, false);                                 // warm
    // all repetitions are enqueued back to back like the iterations of gs_optimize (no host round trip in between);
    // every repetition has its own five phase events plus a sixth right behind the fifth: that empty interval is what
    // one event boundary costs on this stream (ms_event_overhead), i.e. how much of each phase time is the measurement
    // ... then `reps` more iterations with a start / stop pair attached to the linearisation kernel's own dispatch (hipExtLaunchKernelGGL):
    // its begin -> end as a kernel trace reports it, without the hand-over from k_update that the event-to-event interval also holds
    std::vector<hipEvent_t> evs((size_t)reps * 8);
    for (auto &e : evs) HIP_TRY(hipEventCreate(&e));
    hipEvent_t saved[5]; for (int k = 0; k < 5; ++k) saved[k] = g->ev[k];
    for (int r = 0; r < reps; ++r) {
        for (int k = 0; k < 5; ++k) g->ev[k] = evs[(size_t)r * 8 + k];
        enqueue_iteration(g, true);
        hipEventRecord(evs[(size_t)r * 8 + 5], g->stream);
    }
    for (int k = 0; k < 5; ++k) g->ev[k] = saved[k];
    // second pass, nothing recorded between the phases (a dispatch with events attached lengthens the event-to-event interval
    // around it by ~10 us: the two measurements do not share iterations)
    for (int r = 0; r < reps; ++r) {
        g->ev_lin[0] = evs[(size_t)r * 8 + 6]; g->ev_lin[1] = evs[(size_t)r * 8 + 7];
        enqueue_iteration(g, false);
    }
    g->ev_lin[0] = g->ev_lin[1] = nullptr;
    HIP_TRY(hipStreamSynchronize(g->stream));
    double ovh = 0.0, link = 0.0; int nlink = 0;
    for (int r = 0; r < reps; ++r) { const hipEvent_t *e = &evs[(size_t)r * 8];
        float a = 0, b = 0, c = 0, dd = 0, o = 0, lk = 0;
        hipEventElapsedTime(&a, e[0], e[1]); hipEventElapsedTime(&b, e[1], e[2]);
        hipEventElapsedTime(&c, e[2], e[3]); hipEventElapsedTime(&dd, e[3], e[4]); hipEventElapsedTime(&o, e[4], e[5]);
        if (hipEventElapsedTime(&lk, e[6], e[7]) == hipSuccess && lk > 0) { link += lk; ++nlink; }     // (the gather path launches several kernels: no pair)
        s->ms_linearize += a; s->ms_factor += b; s->ms_backsolve += c; s->ms_update += dd; ovh += o; }
    (void)hipGetLastError();
    for (auto &e : evs) hipEventDestroy(e);
    s->ms_linearize /= reps; s->ms_factor /= reps; s->ms_backsolve /= reps; s->ms_update /= reps; s->ms_event_overhead = ovh / reps;
    s->ms_linearize_kernel = nlink > 0 ? link / nlink : 0.0;
    s->ms_total = s->ms_linearize + s->ms_factor + s->ms_backsolve + s->ms_update; s->iterations = reps;
    hipMemcpyAsync(g->d.pose_est, sp, (size_t)N * 3 * sizeof(double), hipMemcpyDeviceToDevice, g->stream);
    launch_pose_trig(g->d, g->stream);
    hipMemcpyAsync(g->d.lm_est, sl, (size_t)M * 2 * sizeof(double), hipMemcpyDeviceToDevice, g->stream);
    int32_t failflag = 0;
    hipMemcpyAsync(&failflag, g->d.fail, sizeof(int32_t), hipMemcpyDeviceToHost, g->stream);
    hipMemsetAsync(g->d.fail, 0, sizeof(int32_t), g->stream);
    HIP_TRY(hipStreamSynchronize(g->stream));
    hipFree(sp); hipFree(sl);
    g->dev_estimates_newer = newer; s->numeric_failure = failflag;
    return GS_OK;
}

// ------------------------------------------------------------------ front end (A0, A1)
// Device memory of the front end lives on the handle and only ever grows: the batch calls carve a scratch arena, the
// per-frame path has pinned staging buffers and a device-resident copy of the map.  Nothing is allocated, freed or
// synchronised beyond the one wait for the results per call.
namespace {
struct Carver {   // carves the handle's grow-only arena (256-byte aligned pieces), valid until the next front-end call
    gs_graph *g; size_t off = 0;
    template <class T> T *get(size_t n) { T *p = (T *)(g->fe.arena + off); off += (std::max<size_t>(n, 1) * sizeof(T) + 255) & ~(size_t)255; return p; }
};
}
static int arena_reserve(gs_graph *g, size_t bytes) {
    if (bytes <= g->fe.arena_bytes) return GS_OK;
    HIP_TRY(hipStreamSynchronize(g->stream));
    if (g->fe.arena) { hipFree(g->fe.arena); g->fe.arena = nullptr; g->fe.arena_bytes = 0; }
    const size_t want = bytes + bytes / 2 + 4096;
    HIP_TRY(hipMalloc((void **)&g->fe.arena, want));
    g->fe.arena_bytes = want;
    return GS_OK;
}
static size_t padded(size_t n, size_t elem) { return (std::max<size_t>(n, 1) * elem + 255) & ~(size_t)255; }
void gs_frontend_release(gs_graph *g) {      // gs_destroy
    if (g->fe.arena) hipFree(g->fe.arena);
    if (g->fe.pin_in) hipHostFree(g->fe.pin_in);
    if (g->fe.pin_out) hipHostFree(g->fe.pin_out);
    if (g->fe.dev_in) hipFree(g->fe.dev_in);
    if (g->fe.dev_out) hipFree(g->fe.dev_out);
    if (g->fe.map_xy) hipFree(g->fe.map_xy);
    if (g->fe.map_type) hipFree(g->fe.map_type);
    if (g->fe.pin_map) hipHostFree(g->fe.pin_map);
    if (g->fe.grid_mem) hipFree(g->fe.grid_mem);
    if (g->fe.pcs) hipFree(g->fe.pcs);
    g->fe = gs_graph::FrontEnd();
}

extern "C" int gs_polar_to_xy_batch(gs_graph *g, int32_t n, const double *az, const double *zen, const double *dist, double *out) {
    if (!g || n < 0 || (n > 0 && (!az || !zen || !dist || !out))) return fail(GS_ERR_INVALID, "bad argument");
    int rc = ensure_device(g); if (rc != GS_OK) return rc;
    if (n == 0) return GS_OK;
    if ((rc = arena_reserve(g, 3 * padded(n, 8) + padded(2 * (size_t)n, 8))) != GS_OK) return rc;
    Carver c{g}; double *a = c.get<double>(n), *z = c.get<double>(n), *d = c.get<double>(n), *o = c.get<double>(2 * (size_t)n);
    HIP_TRY(hipMemcpyAsync(a, az, (size_t)n * 8, hipMemcpyHostToDevice, g->stream));
    HIP_TRY(hipMemcpyAsync(z, zen, (size_t)n * 8, hipMemcpyHostToDevice, g->stream));
    HIP_TRY(hipMemcpyAsync(d, dist, (size_t)n * 8, hipMemcpyHostToDevice, g->stream));
    launch_polar_to_xy(n, a, z, d, g->cfg.lidar_to_cog, o, g->stream);
    HIP_TRY(hipMemcpyAsync(out, o, 2 * (size_t)n * sizeof(double), hipMemcpyDeviceToHost, g->stream));
    HIP_TRY(hipStreamSynchronize(g->stream));
    return GS_OK;
}
extern "C" int gs_cone_to_global_batch(gs_graph *g, int32_t n, const double *poses, int32_t npose, const int32_t *pose_of_obs,
                                       const double *obs, double *out) {
    if (!g || n < 0 || npose < 0 || (n > 0 && (!poses || !pose_of_obs || !obs || !out))) return fail(GS_ERR_INVALID, "bad argument");
    for (int i = 0; i < n; ++i) if (pose_of_obs[i] < 0 || pose_of_obs[i] >= npose) return fail(GS_ERR_INVALID, "pose_of_obs out of range");
    int rc = ensure_device(g); if (rc != GS_OK) return rc;
    if (n == 0) return GS_OK;
    if ((rc = arena_reserve(g, padded(3 * (size_t)npose, 8) + padded(n, 4) + padded(4 * (size_t)n, 8) + padded(2 * (size_t)n, 8))) != GS_OK) return rc;
    Carver c{g}; double *p = c.get<double>(3 * (size_t)npose); int32_t *po = c.get<int32_t>(n);
    double *ob = c.get<double>(4 * (size_t)n), *o = c.get<double>(2 * (size_t)n);
    HIP_TRY(hipMemcpyAsync(p, poses, 3 * (size_t)npose * 8, hipMemcpyHostToDevice, g->stream));
    HIP_TRY(hipMemcpyAsync(po, pose_of_obs, (size_t)n * 4, hipMemcpyHostToDevice, g->stream));
    HIP_TRY(hipMemcpyAsync(ob, obs, 4 * (size_t)n * 8, hipMemcpyHostToDevice, g->stream));
    launch_cone_to_global(n, p, po, ob, g->cfg.lidar_to_cog, o, g->stream);
    HIP_TRY(hipMemcpyAsync(out, o, 2 * (size_t)n * sizeof(double), hipMemcpyDeviceToHost, g->stream));
    HIP_TRY(hipStreamSynchronize(g->stream));
    return GS_OK;
}
// the hashed grid of a map that is in device memory: built on the device (launch_grid_build), nothing crosses PCIe, nothing waits
struct GridBufs { int32_t *count, *start, *cursor, *items; long long buckets; };
static size_t grid_bytes(int n_map, long long &buckets) {
    buckets = 4096; while (buckets < 4 * (long long)n_map) buckets <<= 1;      // a power of two >= 4 n_map: mostly empty buckets, L2-resident
    return 3 * padded((size_t)buckets + 1, 4) + padded((size_t)n_map, 4);
}
static GridBufs grid_carve(char *base, int n_map, long long buckets) {
    GridBufs b; size_t off = 0; auto take = [&](size_t bytes) { char *p = base + off; off += (std::max<size_t>(bytes, 1) + 255) & ~(size_t)255; return p; };
    b.count = (int32_t *)take(((size_t)buckets + 1) * 4); b.start = (int32_t *)take(((size_t)buckets + 1) * 4);
    b.cursor = (int32_t *)take(((size_t)buckets + 1) * 4); b.items = (int32_t *)take((size_t)n_map * 4); b.buckets = buckets;
    return b;
}
extern "C" int gs_associate_batch(gs_graph *g, int32_t n, const double *poses, int32_t npose, const int32_t *pose_of_obs, const double *obs,
                                  int32_t n_map, const double *map_xy, const int32_t *map_type, double thr, double type_tol, int32_t *out) {
    if (!g || n < 0 || npose < 0 || n_map < 0 || (n > 0 && (!poses || !pose_of_obs || !obs || !out)) || (n_map > 0 && (!map_xy || !map_type)))
        return fail(GS_ERR_INVALID, "bad argument");
    for (int i = 0; i < n; ++i) if (pose_of_obs[i] < 0 || pose_of_obs[i] >= npose) return fail(GS_ERR_INVALID, "pose_of_obs out of range");
    int rc = ensure_device(g); if (rc != GS_OK) return rc;
    if (n == 0) return GS_OK;
    // maps beyond a few LDS tiles go through a hashed uniform grid (cell edge a hair above the threshold, so that every cone
    // within the threshold sits in the 3 x 3 cells around the query) that is BUILT ON THE DEVICE from the uploaded map; the brute-force
    // kernel stays for small maps and non-positive thresholds.  gs_debug_options.assoc_grid = 0 / 1 forces either (A/B, tests).
    bool grid = n_map >= 2048 && thr > 0.0;
    if (g->opt.assoc_grid >= 0) grid = g->opt.assoc_grid != 0 && n_map > 0 && thr > 0.0;
    long long buckets = 0; const size_t gbytes = grid ? grid_bytes(n_map, buckets) + 5 * 256 : 0;
    if ((rc = arena_reserve(g, padded(3 * (size_t)npose, 8) + padded(n, 4) + padded(4 * (size_t)n, 8) + padded(2 * (size_t)n_map, 8) +
                               padded(n_map, 4) + padded(n, 4) + gbytes + padded(2 * (size_t)npose, 8))) != GS_OK) return rc;
    Carver c{g}; double *p = c.get<double>(3 * (size_t)npose); int32_t *po = c.get<int32_t>(n); double *pcs = c.get<double>(2 * (size_t)npose);
    double *ob = c.get<double>(4 * (size_t)n), *mx = c.get<double>(2 * (size_t)n_map);
    int32_t *mt = c.get<int32_t>(n_map), *o = c.get<int32_t>(n);
    HIP_TRY(hipMemcpyAsync(p, poses, 3 * (size_t)npose * 8, hipMemcpyHostToDevice, g->stream));
    HIP_TRY(hipMemcpyAsync(po, pose_of_obs, (size_t)n * 4, hipMemcpyHostToDevice, g->stream));
    HIP_TRY(hipMemcpyAsync(ob, obs, 4 * (size_t)n * 8, hipMemcpyHostToDevice, g->stream));
    if (n_map > 0) { HIP_TRY(hipMemcpyAsync(mx, map_xy, 2 * (size_t)n_map * 8, hipMemcpyHostToDevice, g->stream));
                     HIP_TRY(hipMemcpyAsync(mt, map_type, (size_t)n_map * 4, hipMemcpyHostToDevice, g->stream)); }
    if (grid) { const GridBufs gb = grid_carve(c.get<char>(gbytes), n_map, buckets);
        launch_grid_build(n_map, mx, thr, buckets, gb.count, gb.start, gb.cursor, gb.items, g->stream);
        launch_associate_grid_dev(n, p, po, ob, g->cfg.lidar_to_cog, mx, mt, thr, type_tol, buckets, gb.start, gb.items, o, npose, pcs, g->stream);
    } else launch_associate(n, p, po, ob, g->cfg.lidar_to_cog, n_map, mx, mt, thr, type_tol, o, g->stream);
    HIP_TRY(hipMemcpyAsync(out, o, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, g->stream));
    HIP_TRY(hipStreamSynchronize(g->stream));        // the one wait of the call: the result copy
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(GS_ERR_HIP, std::string("association: ") + hipGetErrorString(e));
    return GS_OK;
}
// A1 batched with EVERYTHING resident: the map of gs_map_append (its grid is built on the device, once per map change or threshold),
// poses / observations / result in device memory, asynchronous on the handle's stream (the caller waits: gs_stream_synchronize).
static int resident_grid(gs_graph *g, double thr) {
    auto &fe = g->fe;
    if (fe.grid_valid && fe.grid_map_n == fe.map_n && fe.grid_thr == thr) return GS_OK;
    long long buckets = 0; const size_t bytes = grid_bytes(fe.map_n, buckets) + 5 * 256;
    if (bytes > fe.grid_bytes) { HIP_TRY(hipStreamSynchronize(g->stream));
        if (fe.grid_mem) hipFree(fe.grid_mem);
        fe.grid_mem = nullptr; fe.grid_bytes = 0;
        HIP_TRY(hipMalloc((void **)&fe.grid_mem, bytes + bytes / 2)); fe.grid_bytes = bytes + bytes / 2; }
    const GridBufs gb = grid_carve(fe.grid_mem, fe.map_n, buckets);
    launch_grid_build(fe.map_n, fe.map_xy, thr, buckets, gb.count, gb.start, gb.cursor, gb.items, g->stream);
    fe.grid_valid = true; fe.grid_map_n = fe.map_n; fe.grid_thr = thr; fe.grid_max_cells = buckets;
    return GS_OK;
}
extern "C" int gs_associate_resident(gs_graph *g, int32_t n, const double *dev_poses, int32_t npose, const int32_t *dev_pose_of_obs, const double *dev_obs,
                                     double thr, double type_tol, int32_t *dev_out) {
    if (!g || n < 0 || npose < 0 || (n > 0 && (!dev_poses || !dev_pose_of_obs || !dev_obs || !dev_out))) return fail(GS_ERR_INVALID, "bad argument");
    int rc = ensure_device(g); if (rc != GS_OK) return rc;
    if (n == 0) return GS_OK;
    auto &fe = g->fe;
    const bool grid = thr > 0.0 && fe.map_n > 0 && g->opt.assoc_grid != 0;
    if (grid) { if ((rc = resident_grid(g, thr)) != GS_OK) return rc;
        if ((size_t)npose * 2 * sizeof(double) > fe.pcs_bytes) { HIP_TRY(hipStreamSynchronize(g->stream));      // scratch for the poses' cos / sin, grow-only
            if (fe.pcs) hipFree(fe.pcs);
            fe.pcs = nullptr; fe.pcs_bytes = 0;
            const size_t want = (size_t)npose * 2 * sizeof(double) * 3 / 2 + 4096;
            HIP_TRY(hipMalloc((void **)&fe.pcs, want)); fe.pcs_bytes = want; }
        const GridBufs gb = grid_carve(fe.grid_mem, fe.map_n, fe.grid_max_cells);
        launch_associate_grid_dev(n, dev_poses, dev_pose_of_obs, dev_obs, g->cfg.lidar_to_cog, fe.map_xy, fe.map_type, thr, type_tol, gb.buckets, gb.start, gb.items, dev_out,
                                  npose, fe.pcs, g->stream, g->ev_lin[0], g->ev_lin[1]);
    } else launch_associate(n, dev_poses, dev_pose_of_obs, dev_obs, g->cfg.lidar_to_cog, fe.map_n, fe.map_xy, fe.map_type, thr, type_tol, dev_out, g->stream);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(GS_ERR_HIP, std::string("association: ") + hipGetErrorString(e));
    return GS_OK;
}
// bench / profiling hook (graphslam_debug.h): `reps` launches of the resident association, each with a start / stop event pair attached to
// its dispatch (the kernel's own begin -> end, as for the linearisation kernel); mean milliseconds per launch; the grid is built before
extern "C" int gs_debug_time_associate_resident(gs_graph *g, int32_t n, const double *dev_poses, int32_t npose, const int32_t *dev_pose_of_obs,
                                                const double *dev_obs, double thr, double type_tol, int32_t *dev_out, int32_t reps, double *out_ms) {
    if (!g || !out_ms || reps <= 0) return fail(GS_ERR_INVALID, "bad argument");
    int rc = gs_associate_resident(g, n, dev_poses, npose, dev_pose_of_obs, dev_obs, thr, type_tol, dev_out); if (rc != GS_OK) return rc;     // warm (and the grid)
    std::vector<hipEvent_t> ev(2 * (size_t)reps); for (auto &e : ev) HIP_TRY(hipEventCreate(&e));
    for (int r = 0; r < reps && rc == GS_OK; ++r) { g->ev_lin[0] = ev[2 * (size_t)r]; g->ev_lin[1] = ev[2 * (size_t)r + 1];
        rc = gs_associate_resident(g, n, dev_poses, npose, dev_pose_of_obs, dev_obs, thr, type_tol, dev_out); }
    g->ev_lin[0] = g->ev_lin[1] = nullptr;
    HIP_TRY(hipStreamSynchronize(g->stream));
    double tot = 0; int cnt = 0;
    for (int r = 0; r < reps; ++r) { float ms = 0; if (hipEventElapsedTime(&ms, ev[2 * (size_t)r], ev[2 * (size_t)r + 1]) == hipSuccess && ms > 0) { tot += ms; ++cnt; } }
    (void)hipGetLastError();
    for (auto &e : ev) hipEventDestroy(e);
    *out_ms = cnt ? tot / cnt : 0.0;
    return rc;
}

// ---- the per-keyframe path: resident map + one fused launch -------------------------------------------------------
static int map_reserve(gs_graph *g, int want) {
    if (want <= g->fe.map_cap) return GS_OK;
    const int cap = std::max(want + want / 2, 1024);
    double *xy = nullptr; int32_t *ty = nullptr;
    HIP_TRY(hipMalloc((void **)&xy, (size_t)cap * 2 * sizeof(double)));
    HIP_TRY(hipMalloc((void **)&ty, (size_t)cap * sizeof(int32_t)));
    if (g->fe.map_n > 0) { HIP_TRY(hipMemcpyAsync(xy, g->fe.map_xy, (size_t)g->fe.map_n * 2 * sizeof(double), hipMemcpyDeviceToDevice, g->stream));
                           HIP_TRY(hipMemcpyAsync(ty, g->fe.map_type, (size_t)g->fe.map_n * sizeof(int32_t), hipMemcpyDeviceToDevice, g->stream)); }
    HIP_TRY(hipStreamSynchronize(g->stream));
    if (g->fe.map_xy) hipFree(g->fe.map_xy);
    if (g->fe.map_type) hipFree(g->fe.map_type);
    g->fe.map_xy = xy; g->fe.map_type = ty; g->fe.map_cap = cap;
    return GS_OK;
}
static int pin_map_reserve(gs_graph *g, size_t bytes) {
    if (bytes <= g->fe.pin_map_bytes) return GS_OK;
    HIP_TRY(hipStreamSynchronize(g->stream));                       // a previous staged copy may still be in flight
    if (g->fe.pin_map) hipHostFree(g->fe.pin_map);
    g->fe.pin_map = nullptr; g->fe.pin_map_bytes = 0;
    const size_t want = std::max<size_t>(bytes + bytes / 2, 1 << 16);
    HIP_TRY(hipHostMalloc((void **)&g->fe.pin_map, want, hipHostMallocDefault));
    g->fe.pin_map_bytes = want;
    return GS_OK;
}
extern "C" int gs_map_size(gs_graph *g) { return g ? g->fe.map_n : fail(GS_ERR_INVALID, "null graph"); }
extern "C" int gs_map_clear(gs_graph *g) { if (!g) return fail(GS_ERR_INVALID, "null graph"); g->fe.map_n = 0; g->fe.grid_valid = false; return GS_OK; }
extern "C" int gs_map_append(gs_graph *g, int32_t n, const double *xy, const int32_t *type) {
    if (!g || n < 0 || (n > 0 && (!xy || !type))) return fail(GS_ERR_INVALID, "bad argument");
    int rc = ensure_device(g); if (rc != GS_OK) return rc;
    if (n == 0) return GS_OK;
    if ((rc = map_reserve(g, g->fe.map_n + n)) != GS_OK) return rc;
    // staged through pinned memory so that the copy is asynchronous; the staging buffer is reused once the stream has passed
    // it — every gs_frame_frontend call waits for the stream, and two appends without one in between wait here
    const size_t bx = (size_t)n * 2 * sizeof(double), bt = (size_t)n * sizeof(int32_t);
    if (g->fe.pin_map_busy) { HIP_TRY(hipStreamSynchronize(g->stream)); g->fe.pin_map_busy = false; }
    if ((rc = pin_map_reserve(g, bx + bt)) != GS_OK) return rc;
    std::memcpy(g->fe.pin_map, xy, bx); std::memcpy(g->fe.pin_map + bx, type, bt);
    HIP_TRY(hipMemcpyAsync(g->fe.map_xy + 2 * (size_t)g->fe.map_n, g->fe.pin_map, bx, hipMemcpyHostToDevice, g->stream));
    HIP_TRY(hipMemcpyAsync(g->fe.map_type + g->fe.map_n, g->fe.pin_map + bx, bt, hipMemcpyHostToDevice, g->stream));
    g->fe.pin_map_busy = true;                                      // no wait here: the next frame's launch is ordered behind the copies
    g->fe.map_n += n; g->fe.grid_valid = false;
    return GS_OK;
}
extern "C" int gs_map_set_xy(gs_graph *g, int32_t first, int32_t n, const double *xy) {
    if (!g || first < 0 || n < 0 || (n > 0 && !xy)) return fail(GS_ERR_INVALID, "bad argument");
    if (first + n > g->fe.map_n) return fail(GS_ERR_INVALID, "beyond the end of the map");
    int rc = ensure_device(g); if (rc != GS_OK) return rc;
    if (n == 0) return GS_OK;
    HIP_TRY(hipMemcpyAsync(g->fe.map_xy + 2 * (size_t)first, xy, (size_t)n * 2 * sizeof(double), hipMemcpyHostToDevice, g->stream));
    HIP_TRY(hipStreamSynchronize(g->stream));                       // pageable source: the caller's buffer is free on return
    g->fe.pin_map_busy = false; g->fe.grid_valid = false;
    return GS_OK;
}
extern "C" int gs_frame_frontend(gs_graph *g, const double pose[3], const double *obs, int32_t k, double thr, double type_tol,
                                 int32_t signed_type, double *out_zxy, double *out_gxy, int32_t *out_idx) {
    if (!g || !pose || k < 0 || (k > 0 && (!obs || !out_zxy || !out_gxy || !out_idx))) return fail(GS_ERR_INVALID, "bad argument");
    int rc = ensure_device(g); if (rc != GS_OK) return rc;
    if (k == 0) return GS_OK;
    if (k > g->fe.cap_obs) {                                        // grow-only: staging and device buffers for k observations
        HIP_TRY(hipStreamSynchronize(g->stream));
        if (g->fe.pin_in) hipHostFree(g->fe.pin_in);
        if (g->fe.pin_out) hipHostFree(g->fe.pin_out);
        if (g->fe.dev_in) hipFree(g->fe.dev_in);
        if (g->fe.dev_out) hipFree(g->fe.dev_out);
        g->fe.pin_in = nullptr; g->fe.pin_out = nullptr; g->fe.dev_in = nullptr; g->fe.dev_out = nullptr; g->fe.cap_obs = 0;
        const int cap = std::max(64, k + k / 2);
        const size_t bin = (3 + 4 * (size_t)cap) * sizeof(double), bout = (size_t)cap * (4 * sizeof(double) + sizeof(int32_t));
        HIP_TRY(hipHostMalloc((void **)&g->fe.pin_in, bin, hipHostMallocDefault)); HIP_TRY(hipHostMalloc((void **)&g->fe.pin_out, bout, hipHostMallocDefault));
        HIP_TRY(hipMalloc((void **)&g->fe.dev_in, bin)); HIP_TRY(hipMalloc((void **)&g->fe.dev_out, bout));
        g->fe.cap_obs = cap;
    }
    const size_t bin = (3 + 4 * (size_t)k) * sizeof(double), bz = (size_t)k * 2 * sizeof(double), bi = (size_t)k * sizeof(int32_t);
    std::memcpy(g->fe.pin_in, pose, 3 * sizeof(double)); std::memcpy(g->fe.pin_in + 3, obs, 4 * (size_t)k * sizeof(double));
    double *dz = (double *)g->fe.dev_out, *dg = dz + 2 * (size_t)k; int32_t *di = (int32_t *)(dg + 2 * (size_t)k);
    HIP_TRY(hipMemcpyAsync(g->fe.dev_in, g->fe.pin_in, bin, hipMemcpyHostToDevice, g->stream));
    launch_frame_frontend(k, g->fe.dev_in, g->cfg.lidar_to_cog, g->fe.map_n, g->fe.map_xy, g->fe.map_type, thr, type_tol, signed_type, dz, dg, di, g->stream);
    HIP_TRY(hipMemcpyAsync(g->fe.pin_out, g->fe.dev_out, 2 * bz + bi, hipMemcpyDeviceToHost, g->stream));
    HIP_TRY(hipStreamSynchronize(g->stream));
    g->fe.pin_map_busy = false;
    std::memcpy(out_zxy, g->fe.pin_out, bz); std::memcpy(out_gxy, g->fe.pin_out + bz, bz); std::memcpy(out_idx, g->fe.pin_out + 2 * bz, bi);
    return GS_OK;
}

// ------------------------------------------------------------------ multi-GPU
extern "C" int gs_dist_configure(gs_graph *g, int32_t rank, int32_t world) {
    if (!g || world < 1 || rank < 0 || rank >= world) return fail(GS_ERR_INVALID, "bad rank/world");
    g->rank = rank; g->world = world; ++g->h.structure_version; ++g->h.reshape_version;
    return GS_OK;
}
// ---- rank-local ingestion (round 4): a rank need not hold the observation edges of the other windows' interiors
static void window_starts_of(const gs_graph *g, std::vector<int32_t> &first_pose, std::vector<int32_t> *fp_of_pose = nullptr) {
    const HostGraph &h = g->h; const int N = h.n_poses(), W = std::max(1, g->world);
    int nfree = 0; for (int p = 0; p < N; ++p) nfree += !h.pose_fixed[p];
    first_pose.assign((size_t)W + 1, N);
    if (fp_of_pose) fp_of_pose->assign((size_t)N, -1);
    int f = 0, w = 0;
    for (int p = 0; p < N; ++p) if (!h.pose_fixed[p]) {
        while (w <= W && (int)(((int64_t)w * nfree + W - 1) / W) == f) first_pose[(size_t)w++] = p;     // (empty windows share a start)
        if (fp_of_pose) (*fp_of_pose)[(size_t)p] = f;
        ++f; }
}
extern "C" int gs_dist_window_starts(gs_graph *g, int32_t *out_first_pose, int32_t capacity) {
    if (!g || !out_first_pose) return fail(GS_ERR_INVALID, "null argument");
    if (capacity < g->world + 1) return fail(GS_ERR_CAPACITY, "gs_dist_window_starts: world + 1 entries are written");
    std::vector<int32_t> fp; window_starts_of(g, fp);
    std::memcpy(out_first_pose, fp.data(), fp.size() * sizeof(int32_t));
    return GS_OK;
}
extern "C" int gs_dist_local_landmark_windows(gs_graph *g, uint64_t *seen_interior, uint64_t *seen_first, int32_t n_landmarks) {
    if (!g || !seen_interior || !seen_first) return fail(GS_ERR_INVALID, "null argument");
    const HostGraph &h = g->h;
    if (n_landmarks != h.n_lms()) return fail(GS_ERR_INVALID, "gs_dist_local_landmark_windows: one entry per landmark of the graph");
    if (g->world > 64) return fail(GS_ERR_INVALID, "landmark windows are 64-bit masks: at most 64 ranks");
    std::vector<int32_t> first; window_starts_of(g, first);
    const int r = g->rank; const uint64_t bit = 1ull << r;
    std::fill(seen_interior, seen_interior + n_landmarks, 0ull); std::fill(seen_first, seen_first + n_landmarks, 0ull);
    for (size_t k = 0; k < h.pl_p.size(); ++k) { const int p = h.pl_p[k], l = h.pl_l[k];
        if (h.pose_fixed[p] || h.lm_fixed[l] || p < first[(size_t)r] || p >= first[(size_t)r + 1]) continue;      // this rank's own window only: the ranks' bits are disjoint, their sum is the union
        if (r >= 1 && p == first[(size_t)r]) seen_first[l] |= bit; else seen_interior[l] |= bit; }
    return GS_OK;
}
extern "C" int gs_dist_set_landmark_windows(gs_graph *g, const uint64_t *seen_interior, const uint64_t *seen_first, int32_t n_landmarks) {
    if (!g || n_landmarks < 0 || (n_landmarks > 0 && (!seen_interior || !seen_first))) return fail(GS_ERR_INVALID, "bad argument");
    g->lm_seen_interior.assign(seen_interior, seen_interior + n_landmarks); g->lm_seen_first.assign(seen_first, seen_first + n_landmarks);
    ++g->h.structure_version; ++g->h.reshape_version;
    return GS_OK;
}
extern "C" int64_t gs_dist_exchange_doubles(gs_graph *g) { return (g && g->plan.valid) ? g->plan.exchange_doubles : 0; }
extern "C" int gs_dist_set_exchange_buffer(gs_graph *g, void *p) {
    if (!g) return fail(GS_ERR_INVALID, "null graph");
    g->exchange = (double *)p; g->exchange_external = p != nullptr;
    if (g->dev_valid && p) g->d.exchange = (double *)p;        // the previous (own) buffer stays allocated until the next upload
    return GS_OK;
}
static int dist_ready(gs_graph *g) {
    if (!g) return fail(GS_ERR_INVALID, "null graph");
    if (!g->dev_valid || g->plan_version != g->h.structure_version) return fail(GS_ERR_NOT_INITIALIZED, "call gs_initialize_optimization first");
    if (g->plan.dist && !g->d.exchange) return fail(GS_ERR_NOT_INITIALIZED, "no exchange buffer");
    return ensure_device(g);
}
extern "C" int gs_dist_iterate_local(gs_graph *g) {
    int rc = dist_ready(g); if (rc != GS_OK) return rc;
    enqueue_local(g, false);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(GS_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
    return GS_OK;
}
extern "C" int gs_dist_iterate_finish(gs_graph *g) {
    int rc = dist_ready(g); if (rc != GS_OK) return rc;
    enqueue_finish(g, false);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(GS_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
    return 1;
}
// host copies of the exchange buffer (tests; all-reduce over a CPU backend when ranks share one GPU)
extern "C" int gs_dist_read_exchange(gs_graph *g, double *host) {
    int rc = dist_ready(g); if (rc != GS_OK) return rc;
    if (!host) return fail(GS_ERR_INVALID, "null buffer");
    if (g->plan.exchange_doubles > 0) HIP_TRY(hipMemcpyAsync(host, g->d.exchange, (size_t)g->plan.exchange_doubles * sizeof(double), hipMemcpyDeviceToHost, g->stream));
    HIP_TRY(hipStreamSynchronize(g->stream));
    return GS_OK;
}
extern "C" int gs_dist_write_exchange(gs_graph *g, const double *host) {
    int rc = dist_ready(g); if (rc != GS_OK) return rc;
    if (!host) return fail(GS_ERR_INVALID, "null buffer");
    if (g->plan.exchange_doubles > 0) HIP_TRY(hipMemcpyAsync(g->d.exchange, host, (size_t)g->plan.exchange_doubles * sizeof(double), hipMemcpyHostToDevice, g->stream));
    HIP_TRY(hipStreamSynchronize(g->stream));
    return GS_OK;
}

// ---- RCCL inside the library: the host side of the sharded iteration stays C++ (north_star: "Host stays C++ ... RCCL all-reduce over
// xGMI on the shared-landmark rows").  The RCCL library is resolved at run time — first the copy the process has loaded already (under
// bench.py: torch's), then the system's — so libgraphslam_hip.so has no link-time dependency on it and a single-GPU consumer never loads it.
namespace {
struct RcclApi {
    void *lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr; decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr; decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr; decltype(&ncclCommCount) CommCount = nullptr;
};
RcclApi *rccl_api(std::string &err) {
    static RcclApi api; static bool tried = false; static std::string why;
    if (!tried) { tried = true;
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char *n : names) if (!api.lib) api.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);     // a copy the process has loaded already
        for (const char *n : names) if (!api.lib) api.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (!api.lib) why = std::string("librccl.so not found: ") + (dlerror() ? dlerror() : "");
        else {
            api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(api.lib, "ncclGetUniqueId"); api.CommInitRank = (decltype(api.CommInitRank))dlsym(api.lib, "ncclCommInitRank");
            api.AllReduce = (decltype(api.AllReduce))dlsym(api.lib, "ncclAllReduce"); api.CommDestroy = (decltype(api.CommDestroy))dlsym(api.lib, "ncclCommDestroy");
            api.GetErrorString = (decltype(api.GetErrorString))dlsym(api.lib, "ncclGetErrorString"); api.CommCount = (decltype(api.CommCount))dlsym(api.lib, "ncclCommCount");
            if (!api.GetUniqueId || !api.CommInitRank || !api.AllReduce || !api.CommDestroy) { why = "librccl.so lacks ncclGetUniqueId / ncclCommInitRank / ncclAllReduce / ncclCommDestroy"; api.lib = nullptr; } } }
    if (!api.lib) { err = why; return nullptr; }
    return &api;
}
int rccl_fail(RcclApi *R, ncclResult_t rc, const char *what) {
    return fail(GS_ERR_HIP, std::string(what) + ": " + (R && R->GetErrorString ? R->GetErrorString(rc) : "RCCL error") + " (" + std::to_string((int)rc) + ")");
}
}  // namespace
extern "C" int gs_dist_unique_id(void *out128) {
    if (!out128) return fail(GS_ERR_INVALID, "null buffer");
    std::string err; RcclApi *R = rccl_api(err); if (!R) return fail(GS_ERR_NO_DEVICE, err);
    static_assert(sizeof(ncclUniqueId) == 128, "gs_dist_unique_id hands out NCCL_UNIQUE_ID_BYTES = 128 bytes");
    ncclUniqueId id; ncclResult_t rc = R->GetUniqueId(&id); if (rc != ncclSuccess) return rccl_fail(R, rc, "ncclGetUniqueId");
    std::memcpy(out128, &id, sizeof(id)); return GS_OK;
}
extern "C" int gs_dist_comm_init(gs_graph *g, const void *unique_id_128, int32_t rank, int32_t world) {
    if (!g || !unique_id_128 || world < 1 || rank < 0 || rank >= world) return fail(GS_ERR_INVALID, "bad argument");
    int rc = ensure_device(g); if (rc != GS_OK) return rc;
    std::string err; RcclApi *R = rccl_api(err); if (!R) return fail(GS_ERR_NO_DEVICE, err);
    if (g->comm && g->own_comm) { R->CommDestroy((ncclComm_t)g->comm); g->comm = nullptr; }
    ncclUniqueId id; std::memcpy(&id, unique_id_128, sizeof(id));
    ncclComm_t c = nullptr; ncclResult_t nr = R->CommInitRank(&c, world, id, rank);      // (collective: every rank of the group calls it; the current device is the handle's)
    if (nr != ncclSuccess) return rccl_fail(R, nr, "ncclCommInitRank");
    g->comm = c; g->own_comm = true; g->comm_world = world;
    return GS_OK;
}
extern "C" int gs_dist_set_communicator(gs_graph *g, void *nccl_comm) {
    if (!g) return fail(GS_ERR_INVALID, "null graph");
    std::string err; RcclApi *R = rccl_api(err); if (!R) return fail(GS_ERR_NO_DEVICE, err);
    if (g->comm && g->own_comm) R->CommDestroy((ncclComm_t)g->comm);
    g->comm = nccl_comm; g->own_comm = false; g->comm_world = 0;
    if (nccl_comm && R->CommCount) { int n = 0; if (R->CommCount((ncclComm_t)nccl_comm, &n) == ncclSuccess) g->comm_world = n; }
    return GS_OK;
}
void gs_dist_comm_release(gs_graph *g) {      // gs_destroy
    if (!g->comm || !g->own_comm) { g->comm = nullptr; return; }
    std::string err; if (RcclApi *R = rccl_api(err)) R->CommDestroy((ncclComm_t)g->comm);
    g->comm = nullptr;
}
// the all-reduce of the shared fronts' slots (and of the ranks' failure flags at the buffer's tail), enqueued on the handle's stream
static int enqueue_allreduce(gs_graph *g) {
    if (!g->comm) return fail(GS_ERR_NOT_INITIALIZED, "no RCCL communicator: gs_dist_comm_init or gs_dist_set_communicator first");
    if (g->comm_world > 0 && g->comm_world != g->world) return fail(GS_ERR_INVALID, "the communicator's size differs from gs_dist_configure's world");
    std::string err; RcclApi *R = rccl_api(err); if (!R) return fail(GS_ERR_NO_DEVICE, err);
    const int64_t n = g->plan.exchange_doubles;
    if (n <= 0) return GS_OK;
    ncclResult_t nr = R->AllReduce(g->d.exchange, g->d.exchange, (size_t)n, ncclDouble, ncclSum, (ncclComm_t)g->comm, g->stream);
    return nr == ncclSuccess ? GS_OK : rccl_fail(R, nr, "ncclAllReduce");
}
// rank-local ingestion without any other channel between the replicas than the library's own communicator: this rank's bits of the landmark windows
// (from the edges it holds), ncclAllReduce(uint64, sum) — the ranks' bits are disjoint, the sum is the union —, the result handed to the handle
extern "C" int gs_dist_share_landmark_windows(gs_graph *g) {
    if (!g) return fail(GS_ERR_INVALID, "null graph");
    if (!g->comm) return fail(GS_ERR_NOT_INITIALIZED, "no RCCL communicator: gs_dist_comm_init or gs_dist_set_communicator first");
    if (g->comm_world > 0 && g->comm_world != g->world) return fail(GS_ERR_INVALID, "the communicator's size differs from gs_dist_configure's world");
    int rc = ensure_device(g); if (rc != GS_OK) return rc;
    std::string err; RcclApi *R = rccl_api(err); if (!R) return fail(GS_ERR_NO_DEVICE, err);
    const int M = g->h.n_lms();
    std::vector<uint64_t> m(2 * (size_t)M);
    if ((rc = gs_dist_local_landmark_windows(g, m.data(), m.data() + M, M)) != GS_OK) return rc;
    if (M == 0) return gs_dist_set_landmark_windows(g, nullptr, nullptr, 0);
    uint64_t *dev = nullptr;
    HIP_TRY(hipMalloc(&dev, m.size() * sizeof(uint64_t)));
    hipError_t e = hipMemcpyAsync(dev, m.data(), m.size() * sizeof(uint64_t), hipMemcpyHostToDevice, g->stream);
    ncclResult_t nr = e == hipSuccess ? R->AllReduce(dev, dev, m.size(), ncclUint64, ncclSum, (ncclComm_t)g->comm, g->stream) : ncclSuccess;
    if (e == hipSuccess && nr == ncclSuccess) e = hipMemcpyAsync(m.data(), dev, m.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, g->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(g->stream);
    hipFree(dev);
    if (nr != ncclSuccess) return rccl_fail(R, nr, "ncclAllReduce (landmark windows)");
    if (e != hipSuccess) return fail(GS_ERR_HIP, std::string("landmark windows: ") + hipGetErrorString(e));
    return gs_dist_set_landmark_windows(g, m.data(), m.data() + M, M);
}
extern "C" int gs_dist_iterate(gs_graph *g) {
    int rc = dist_ready(g); if (rc != GS_OK) return rc;
    if (!g->plan.dist) return fail(GS_ERR_INVALID, "not a sharded graph: gs_iterate");
    enqueue_local(g, false);
    if ((rc = enqueue_allreduce(g)) != GS_OK) return rc;
    enqueue_finish(g, false);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(GS_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
    return 1;
}
// measurement hook (graphslam_debug.h): `reps` all-reduces of the exchange buffer back to back on the handle's stream, HIP events around
// them; mean milliseconds per all-reduce.  Collective: every rank of the communicator calls it.
extern "C" int gs_debug_time_exchange(gs_graph *g, int32_t reps, double *out_ms) {
    if (!out_ms || reps <= 0) return fail(GS_ERR_INVALID, "bad argument");
    int rc = dist_ready(g); if (rc != GS_OK) return rc;
    if ((rc = enqueue_allreduce(g)) != GS_OK) return rc;           // warm
    hipEventRecord(g->ev[0], g->stream);
    for (int r = 0; r < reps && rc == GS_OK; ++r) rc = enqueue_allreduce(g);
    hipEventRecord(g->ev[1], g->stream);
    HIP_TRY(hipEventSynchronize(g->ev[1]));
    float ms = 0; HIP_TRY(hipEventElapsedTime(&ms, g->ev[0], g->ev[1]));
    *out_ms = (double)ms / reps;
    return rc;
}
// Slam's optimize(10) (reference src/slam.cpp:481) on a sharded graph: every rank makes the same call; g2o's failure rule holds across
// ranks (a rank's failure flag rides through the all-reduce: no rank applies the update of that iteration or any later one).  Returns the
// iterations whose update was applied, 0 when any rank's factorisation failed.  The estimates this rank tracks (gs_dist_known) come back.
extern "C" int gs_dist_optimize(gs_graph *g, int32_t iterations, gs_stats *stats) {
    if (!g) return fail(GS_ERR_INVALID, "null graph");
    if (iterations < 0) return fail(GS_ERR_INVALID, "negative iteration count");
    int rc = ensure_ready(g); if (rc != GS_OK) return rc;
    if (!g->plan.dist) return fail(GS_ERR_INVALID, "not a sharded graph: gs_optimize");
    if ((rc = dist_ready(g)) != GS_OK) return rc;
    HIP_TRY(hipMemsetAsync(g->d.fail, 0, 4 * sizeof(int32_t), g->stream));
    g->d.conv_tol = -1.0;
    hipEventRecord(g->ev[5], g->stream);
    const int nh = std::min(iterations, 64);
    // A flag timeout on ANY rank (code 2 where it happened, 4 on the others: the same all-reduce tells everybody) is not a property of H: the rank it
    // happened on switches to one launch per level, every rank runs the iterations that were not applied again — the repair gs_optimize makes on one GPU,
    // decided identically on every rank (the count of applied updates is the same everywhere).  At most twice per call.
    int32_t ff[4] = {0, 0, 0, 0}; double hist[80]; int from = 0, first_failure = 0;
    for (int repair = 0; ; ++repair) {
        for (int it = from; it < iterations; ++it) {
            g->d.hist_slot = it < nh ? it : -1;
            enqueue_local(g, false);
            if ((rc = enqueue_allreduce(g)) != GS_OK) { g->d.hist_slot = -1; return rc; }
            enqueue_finish(g, false);
        }
        g->d.hist_slot = -1;
        hipEventRecord(g->ev[6], g->stream);
        HIP_TRY(hipMemcpyAsync(ff, g->d.fail, sizeof(ff), hipMemcpyDeviceToHost, g->stream));
        HIP_TRY(hipMemcpyAsync(hist, g->d.chi2, sizeof(hist), hipMemcpyDeviceToHost, g->stream));
        HIP_TRY(hipStreamSynchronize(g->stream));
        if (ff[0] != 0 && first_failure == 0) first_failure = ff[0];
        if ((ff[0] != 2 && ff[0] != 4) || repair >= 2) break;
        if (ff[0] == 2) { g->d.tree = 0; g->fell_back = true; g->fallback_calls = 0; }
        // Every rank is here (the code came with the same all-reduce).  A launch that gave up BEHIND the exchange — the shared top, a backward solve — is only
        // heard of with the NEXT contribution: by then the other ranks have applied an update the rank it happened on has not.  The ranks compare their counts
        // (one more all-reduce, only in this branch); if they differ the estimates have parted and no re-run can mend that: every rank says so, nobody goes on.
        { std::string err; RcclApi *R = rccl_api(err); if (!R) return fail(GS_ERR_NO_DEVICE, err);
          double v[2] = {(double)ff[1], -(double)ff[1]}, *dv = nullptr;
          HIP_TRY(hipMalloc(&dv, sizeof(v)));
          hipError_t e2 = hipMemcpyAsync(dv, v, sizeof(v), hipMemcpyHostToDevice, g->stream);
          ncclResult_t nr = e2 == hipSuccess ? R->AllReduce(dv, dv, 2, ncclDouble, ncclMax, (ncclComm_t)g->comm, g->stream) : ncclSuccess;
          if (e2 == hipSuccess && nr == ncclSuccess) e2 = hipMemcpyAsync(v, dv, sizeof(v), hipMemcpyDeviceToHost, g->stream);
          if (e2 == hipSuccess) e2 = hipStreamSynchronize(g->stream);
          hipFree(dv);
          if (nr != ncclSuccess) return rccl_fail(R, nr, "ncclAllReduce (applied updates)");
          if (e2 != hipSuccess) return fail(GS_ERR_HIP, std::string("applied updates: ") + hipGetErrorString(e2));
          if (v[0] != -v[1]) { reset_failure(g);
              return fail(GS_ERR_TIMEOUT, "a launch behind the exchange gave up on one rank after the others had applied that iteration's update: the ranks' estimates have parted "
                                          "(updates applied: " + std::to_string((long long)-v[1]) + " .. " + std::to_string((long long)v[0]) + "); set the estimates again on every rank"); } }
        g->d.inject_iter = 0;
        HIP_TRY(hipMemsetAsync(g->d.fail, 0, sizeof(int32_t), g->stream));      // the code only: the update count goes on
        if (g->d.tickets) { HIP_TRY(hipMemsetAsync(g->d.tickets, 0, 2 * sizeof(uint32_t), g->stream)); g->d.ticket_base = 0; }
        from = ff[1]; ff[0] = 0;
    }
    rc = pull_estimates_if_needed(g); if (rc != GS_OK) return rc;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { reset_failure(g); return fail(GS_ERR_HIP, std::string("iteration: ") + hipGetErrorString(e)); }      // (a launch that never ran: the ticket counter and its running sum start again)
    float ms = 0; hipEventElapsedTime(&ms, g->ev[5], g->ev[6]);
    if (stats) { std::memset(stats, 0, sizeof(*stats)); stats->struct_size = (int32_t)sizeof(*stats);
        fill_plan_stats(g, stats); stats->iterations = ff[1]; stats->numeric_failure = ff[0]; stats->first_failure = first_failure;
        stats->chi2_initial = iterations > 0 ? hist[1] : 0.0; stats->chi2_final = iterations > 0 ? hist[std::min(iterations, nh)] : 0.0;     // THIS rank's edges only (the ranks' sums add up to the graph's)
        stats->ms_total = ms; }
    if (ff[0]) { rc = reset_failure(g); if (rc != GS_OK) return rc;
        if (ff[0] == 2) { g->d.tree = 0; g->fell_back = true; g_last_error = "a front's completion flag did not arrive in time: the handle now uses one launch per level"; }
        else if (ff[0] == 4) g_last_error = "another rank's whole-tree launch gave up on a front's flag, twice in this call";
        else g_last_error = ff[0] == 3 ? "another rank met a zero pivot (g2o: optimize() returns 0, the vertices keep the last good iterate)" : "zero pivot: H is singular (g2o: optimize() returns 0, the vertices keep the last good iterate)";
        return 0; }
    return ff[1];
}
// which vertex estimates this rank tracks (its own subtrees + the shared top), insertion order; a vertex is
// `primary` on exactly one rank (shared vertices: rank 0), so summing primary-masked estimates over ranks merges them
extern "C" int gs_dist_known(gs_graph *g, uint8_t *pose_known, uint8_t *lm_known, uint8_t *pose_primary, uint8_t *lm_primary) {
    if (!g) return fail(GS_ERR_INVALID, "null graph");
    if (!g->plan.valid) return fail(GS_ERR_NOT_INITIALIZED, "no plan built");
    const Plan &P = g->plan;
    auto primary = [&](int gidx, uint8_t known, uint8_t fixed) -> uint8_t {
        if (fixed || gidx < 0) return P.rank == 0;                      // fixed vertices never move: take them from rank 0
        if (!known) return 0;
        // shared <=> known on every rank
        return 1; };
    // a shared vertex is known everywhere; make rank 0 its primary holder
    std::vector<int32_t> front_of_scalar;                              // scalar -> front owner lookup via pivots
    front_of_scalar.assign(P.n_scalar, 0);
    for (size_t s = 0; s < P.fronts.size(); ++s) for (int k = 0; k < P.fronts[s].npiv; ++k) front_of_scalar[P.fronts[s].piv0 + k] = P.fronts[s].owner;
    for (int p = 0; p < g->h.n_poses(); ++p) { const int gi = P.pose_gidx[p]; uint8_t kn = P.pose_known[p], pr = primary(gi, kn, g->h.pose_fixed[p]);
        if (gi >= 0 && kn && front_of_scalar[gi] < 0) pr = P.rank == 0;
        if (pose_known) pose_known[p] = kn; if (pose_primary) pose_primary[p] = pr; }
    for (int l = 0; l < g->h.n_lms(); ++l) { const int gi = P.lm_gidx[l]; uint8_t kn = P.lm_known[l], pr = primary(gi, kn, g->h.lm_fixed[l]);
        if (gi >= 0 && kn && front_of_scalar[gi] < 0) pr = P.rank == 0;
        if (lm_known) lm_known[l] = kn; if (lm_primary) lm_primary[l] = pr; }
    return GS_OK;
}
