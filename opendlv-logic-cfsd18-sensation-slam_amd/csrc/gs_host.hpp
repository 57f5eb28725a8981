// gs_host.hpp — host-side graph container and multifrontal plan of the GraphSLAM back-end.
//
// HostGraph is the SoA replacement of what g2o keeps as heap-allocated vertex/edge objects behind
// Slam::m_optimizer (reference src/slam.hpp:98; insertions src/slam.cpp:433-459, 525-550).
// Plan is the product of the structure phase that g2o runs in initializeOptimization() +
// BlockSolver::buildStructure() + Eigen analyzePattern (reference src/slam.cpp:480-481,
// SURVEY.md §8 row A4): index maps, elimination order, symbolic factorisation — here a
// nested-dissection multifrontal plan instead of AMD + simplicial column counts.
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <type_traits>
#include <utility>
#include <unordered_map>
#include <vector>

namespace gs {

// std::vector whose resize() leaves new elements uninitialised (arrays that are fully overwritten right after, in
// parallel: zero-filling 20 MB first is 2-3 ms of the structure phase and serial page faults)
template <class T> struct DefaultInitAlloc : std::allocator<T> {
    template <class U> struct rebind { using other = DefaultInitAlloc<U>; };
    using std::allocator<T>::allocator;
    template <class U> void construct(U *p) noexcept(std::is_nothrow_default_constructible<U>::value) { ::new (static_cast<void *>(p)) U; }
    template <class U, class... A> void construct(U *p, A &&...a) { ::new (static_cast<void *>(p)) U(std::forward<A>(a)...); }
};
template <class T> using uvec = std::vector<T, DefaultInitAlloc<T>>;

struct HostGraph {
    // vertices, insertion order
    std::vector<int32_t> pose_id;  std::vector<double> pose_est;  std::vector<uint8_t> pose_fixed;   // [N*3]
    std::vector<int32_t> lm_id;    std::vector<double> lm_est;    std::vector<uint8_t> lm_fixed;     // [M*2]
    std::unordered_map<int32_t, int32_t> pose_index, lm_index;                                      // id -> index
    // edges, insertion order; information stored packed symmetric
    std::vector<int32_t> pp_i, pp_j; std::vector<double> pp_z /*[E*3]*/, pp_info /*[E*6] xx xy xt yy yt tt*/;
    std::vector<int32_t> pl_p, pl_l; std::vector<double> pl_z /*[E*2]*/, pl_info /*[E*3] xx xy yy*/;
    uint64_t structure_version = 0;     // bumped by every change that invalidates the plan
    uint64_t reshape_version = 0;       // bumped by the changes a plan cannot absorb by growing: a fixed flag flipped, clear(), a new shard layout
    uint64_t estimate_version = 0;      // bumped by host-side estimate writes

    int n_poses() const { return (int)pose_id.size(); }
    int n_lms() const { return (int)lm_id.size(); }
    int n_pp() const { return (int)pp_i.size(); }
    int n_pl() const { return (int)pl_p.size(); }
    void clear();
};

// One assembly record = one original H block (or diagonal block + rhs) landing in a front.
enum AsmKind : int32_t {
    ASM_POSE_DIAG = 0,   // src = pose index      : 3x3 at (r0,r0) lower part, rhs row gets b_pose
    ASM_LM_DIAG = 1,     // src = landmark index  : 2x2 at (r0,r0) lower part, rhs row gets b_lm
    ASM_PP = 2,          // src = sorted pp edge  : F[r0+a][c0+b] = Hpp_off[a][b]   (i-vertex rows later)
    ASM_PP_T = 3,        //                         F[r0+a][c0+b] = Hpp_off[b][a]   (j-vertex rows later)
    ASM_PL = 4,          // src = sorted pl edge  : F[r0+a][c0+b] = Hpl[a][b]  3x2  (pose rows later)
    ASM_PL_T = 5,        //                         F[r0+a][c0+b] = Hpl[b][a]  2x3  (landmark rows later)
    ASM_LM_DIAG_TAIL = 6,// src = landmark index >= base_M (a landmark appended after the plan was built, grow_plan): like ASM_LM_DIAG, its block in the tail arena
};
struct AsmRec { int32_t kind, src, r0, c0; };

struct Front {
    int32_t npiv = 0, nbnd = 0;     // pivot / boundary scalars, f = npiv + nbnd
    int32_t piv0 = 0;               // first pivot's index in the elimination order (pivots contiguous)
    int32_t parent = -1, level = 0;
    int32_t owner = 0;              // rank that factorises it (multi-GPU); -1 = shared top of the tree
    int32_t opaque = 0;             // pose-window shards: a whole subtree of ANOTHER rank, kept as one supernode (vertices + boundary only: no records, no storage)
    int64_t bnd_off = 0;            // into Plan::bnd_rows  (elimination indices of boundary rows, ascending)
    int64_t map_off = 0;            // into Plan::child_map (this front's boundary rows -> rows of parent's front)
    int64_t L_off = 0;              // doubles: (f+1) x npiv column-major, ld = f+1 (last row = forward-solved rhs)
    int64_t U_off = 0;              // doubles: (nbnd+1) x nbnd column-major, ld = nbnd+1 (last row = rhs update)
    int32_t asm_off = 0, asm_cnt = 0, asm_dup = 0;   // records [asm_off, +asm_cnt): first asm_cnt-asm_dup unique, rest duplicates
    int32_t child_off = 0, child_cnt = 0;            // into Plan::children
};

struct Plan {
    bool valid = false;
    int32_t n_scalar = 0;                   // free scalar unknowns
    // vertex -> first scalar in elimination order (-1 fixed)
    std::vector<int32_t> pose_gidx, lm_gidx;
    // sorted edge orders (position -> insertion index) used on the device
    std::vector<int32_t> pl_order, pp_order;
    // CSR: pose -> its pl edges are [pl_start[p], pl_start[p+1]) in sorted order
    std::vector<int32_t> pl_start;
    // CSR: landmark -> ELL indices of its edges
    std::vector<int32_t> lm_start, lm_edges;
    // CSR: pose -> incident pp edges (sorted position * 2 + role; role 0 = i endpoint)
    std::vector<int32_t> ppadj_start, ppadj;
    // ELL layout of the observation edges on the device: T lanes per pose, R slots per lane
    int32_t ell_T = 1, ell_R = 1; int64_t ell_len = 0;   // ell_len = R * T * ell_np + 1 (the last entry: a permanent empty slot)
    int32_t ell_p0 = 0, ell_np = 0;                       // the poses the layout covers: [ell_p0, ell_p0 + ell_np) — all of them (world 1) or the ones this rank sweeps
    std::vector<int32_t> ell_ins;                         // [ell_len] ELL index -> insertion index (-1 = empty slot)
    std::vector<int32_t> ell_of_ins;                      // insertion index -> ELL index (-1: the edge's pose is outside the layout)
    std::vector<int32_t> ppinc;                           // [Q][4] edge, role, i, j per (pose, odometry edge) incidence
    // wave tiles of the fused A5-A7 kernel (valid when lin_ell_ok): one wave = 64/T consecutive poses
    bool lin_ell_ok = false; int32_t n_wtiles = 0;
    int32_t wt_lo = 0, wt_hi = 0;                         // the wave tiles this rank sweeps (world 1: all)
    std::vector<int32_t> wt_grp_start;                    // [WT+1] landmark groups of a wave tile
    std::vector<int32_t> wt_desc;                         // [WT][4] first group, #groups, first position, #positions
    std::vector<int32_t> grp_lm, grp_pos_start, grp_pos;  // group -> landmark, wave-local positions (slot*64 + lane)
    std::vector<uint16_t> ell_dst;                        // per ELL entry: its index in its wave tile's group-sorted position list (0xFFFF = padding)
    std::vector<int32_t> lm_grp_start;                    // landmark -> its run of partial-sum slots
    std::vector<int32_t> grp_slot;                        // group -> partial-sum slot, slots ordered by (landmark, wave tile)
    // fronts in elimination (post)order
    std::vector<Front> fronts;
    uvec<int32_t> bnd_rows, child_map; std::vector<int32_t> children;
    uvec<AsmRec> asm_recs;
    // levels: fronts of level l are level_fronts[level_start[l] .. level_start[l+1])
    std::vector<int32_t> level_start, level_fronts;
    int32_t max_front = 0;
    int64_t l_doubles = 0, u_doubles = 0, factor_flops = 0;
    // multi-GPU
    int32_t world = 1, rank = 0;
    int32_t n_shared_fronts = 0;            // fronts with owner == -1
    bool dist = false;                      // the iteration runs in two halves around an all-reduce of the exchange buffer: world > 1, or world 1 with a forced shared top
    int64_t exchange_doubles = 0;           // dense (f+1) x f slots of all shared fronts: the all-reduced buffer
    std::vector<int32_t> pl_rank, pp_rank;  // insertion order: the rank that evaluates the edge
    std::vector<uint8_t> pose_known, lm_known;   // this rank tracks the vertex's estimate (own subtree or shared top)
    std::vector<int32_t> level_start_owned, level_fronts_owned;    // this rank's fronts by level
    std::vector<int32_t> level_start_shared, level_fronts_shared;  // shared top by level
    std::vector<int64_t> x_off;             // front -> offset of its slot in the exchange buffer (-1 not shared)
    double ms_build = 0;
    // ---- append-only growth (grow_plan): what the LINEARISATION LAYOUT above covers are the base counts; poses / edges beyond them
    // form the tail (their blocks live in the tail arenas of the device, gs_device.hpp)
    int32_t base_N = 0, base_M = 0, base_Epp = 0, base_Epl = 0;      // counts build_plan saw
    int32_t planned_N = 0, planned_M = 0, planned_Epp = 0, planned_Epl = 0;   // counts the plan covers now (base + tail)
    int32_t n_growths = 0;                               // grow_plan calls since build_plan
    int32_t root_f0 = 0;                                 // size of the root front as build_plan left it
    int32_t front_limit = 63;                            // what a front may grow to: 63 (every front a wave: the plan was built without workgroup fronts) or 159 (it holds some: table-driven launches, 160-entry row tables)
    uint64_t reshape_version = 0;                        // HostGraph::reshape_version the plan was built at
};

// Tail capacities (device arenas are sized for them at every full structure phase)
constexpr int TAIL_POSES = 16, TAIL_LMS = 16, TAIL_PL = 512, TAIL_PP = 64;

// What one grow_plan call changed (the device patch works from this)
struct Growth {
    std::vector<int32_t> fronts;            // fronts whose size, storage, records or maps changed (ascending = elimination order; the root last)
    int64_t bnd_from = 0, map_from = 0, asm_from = 0;     // Plan::bnd_rows / child_map / asm_recs entries from these on are new
    int32_t first_pose = 0, first_lm = 0, first_pp = 0, first_pl = 0;   // the vertices / edges this call took in: [first, planned)
};

// Append-only growth of a built plan (reference src/slam.cpp:433-459, 537-550: one more pose vertex, its odometry edge, its
// observation edges): the new poses become extra pivots of the ROOT front — eliminated last — and every front on the path from
// a neighbour's front to the root gains them as boundary rows.  Only those fronts change; the elimination tree, the levels, the
// order and the linearisation layout of everything older stay.  Returns false (with the reason) when the change is not of
// that kind or does not fit (new landmark, fixed flag flipped, a front would exceed 63 scalars, tail capacity, sharded plan ...):
// the caller rebuilds.
bool grow_plan(const HostGraph &g, Plan &plan, Growth &out, std::string &why_not);

struct PlanOptions { int leaf_poses = 8; int world = 1; int rank = 0; int ell_lanes = 0;
                     bool by_window = true;        // pose-window shards: the top of the tree from per-landmark window masks (Builder::nd_top) where the graph allows; false: the general recursion
                     int cluster_ways = 0;         // fan-out of the multi-way split above the leaves (0 = default 8, <= 2 = binary all the way down)
                     int big_cluster_front = -1;       // second-pass bound of a cluster front when 63 scalars cannot be met (0 .. 63: off); -1 = by the view: 111 with more than 10
                                                       // cones per frame, off below (there binary splits of wave fronts beat a workgroup front: lap-sized graphs 1.1 vs 1.9-2.6 ms per optimize(10))
                     int grow_headroom = 6;            // scalars a cluster front stays below the 63 of a wave: room for the boundary rows of two appended poses (grow_plan)
                     int grow_spine_headroom = 18;     // the same for the cluster front that holds the LAST pose: appended keyframes continue the track there (six poses)
                     // rank-local ingestion (gs_dist_set_landmark_windows): per landmark (insertion index), the windows whose INTERIOR poses / whose FIRST pose see it,
                     // handed over by the caller — the graph then needs the observation edges of this rank's own window, of the windows' first poses and of the
                     // fixed poses only; nullptr: computed here in one pass over all observation edges (every rank holds the whole graph)
                     const uint64_t *lm_seen_interior = nullptr, *lm_seen_first = nullptr;
                     bool timing = false;              // per-phase wall times on stderr (gs_debug_options.plan_timing)
                     int force_shared_top = 0; };      // world 1: treat the top k levels as the shared top of a sharded graph (gs_debug_options.force_shared_top)

constexpr int LIN_R = 4;               // observation slots per lane handled by the fused linearisation kernel

// Builds the plan on the host (no device work).  Returns false (and sets err) on failure.
// workspace (optional): where the scratch of the build lives between two builds of one handle (opaque; freed with the last reference)
bool build_plan(const HostGraph &g, const PlanOptions &opt, Plan &plan, std::string &err, std::shared_ptr<void> *workspace = nullptr);

// flat int32 dump for tests (layout documented in gs_plan.cpp)
void export_plan(const Plan &plan, std::vector<int32_t> &out);

}  // namespace gs
