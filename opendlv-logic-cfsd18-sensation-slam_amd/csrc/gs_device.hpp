// gs_device.hpp — device-resident state of one graph (all pointers are HBM) and kernel launchers.
//
// HBM layout (fp64 values, int32 indices).  Streams that the edge kernels sweep are structure-of-arrays so
// that consecutive lanes touch consecutive addresses:
//   observation edges, ELL with T lanes per pose and R slots per lane, L = R*T*N entries, the s-th edge of
//   pose p at idx = (s / T)*(T*N) + T*p + (s % T):
//                                       ell_l[L] (landmark, -1 = empty) | ell_z [2][L] | ell_w [3][L] (xx xy yy)
//   odometry edges, insertion order  :  pp_zinv [E][5] (x y theta cos sin of z^-1) | pp_info [E][6]
//                                       ppinc [Q][2] {edge, other endpoint | role << 31} per (pose, edge) incidence, grouped by pose
//   estimates                        :  pose_est [N][3], lm_est [M][2]           (AoS, gathered)
//   block-sparse H and b (A6/A7 out) :  Hpl [6][L] | Hpp_off [9][Epp] | Hpp_diag [6][N] (xx xy xt yy yt tt)
//                                       b_pose [3][N] | Hll_diag [3][M] (00 01 11) | b_lm [2][M]
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace gs {

// Mirror of Front for the device (POD, 80 bytes)
struct DevFront {
    int32_t npiv, nbnd, piv0, parent;
    int32_t asm_off, asm_cnt, asm_dup, child_off;
    int32_t child_cnt, owner, level, pad0;
    int64_t bnd_off, map_off, L_off, U_off;
};

struct DevGraph {
    int32_t N = 0, M = 0, Epp = 0, Epl = 0, n_scalar = 0;
    // state
    double *pose_est = nullptr, *lm_est = nullptr;             // [N*3], [M*2]
    double *pose_cs = nullptr;                                  // [N*2] cos, sin of every pose's theta (kept by k_pose_trig / k_update)
    uint8_t *pose_fixed = nullptr, *lm_fixed = nullptr;
    int32_t *pose_gidx = nullptr, *lm_gidx = nullptr;          // first scalar in elimination order, -1 fixed
    // observation edges, ELL
    int32_t ell_T = 1, ell_R = 1; int64_t ell_len = 0;
    int32_t ell_p0 = 0, ell_np = 0;                             // the layout covers poses [ell_p0, ell_p0 + ell_np): all (single GPU) or the ones this rank sweeps; plane stride T * ell_np
    int32_t *ell_l = nullptr; double *ell_z = nullptr, *ell_w = nullptr;
    // odometry edges; the measurement is stored inverted (g2o keeps _inverseMeasurement) with its cos/sin
    double *pp_zinv = nullptr, *pp_info = nullptr;              // zinv [E][5], info [E][6]
    int32_t *ppinc = nullptr, *ppadj_start = nullptr;           // incidence records [Q][2] {edge, other endpoint | role << 31}, pose -> incidence range
    // landmark -> ELL indices of its edges (gather kernels)
    int32_t *lm_start = nullptr, *lm_edges = nullptr;
    // fused-kernel wave tiles
    int32_t n_wtiles = 0, n_groups = 0;
    int32_t *wt_desc = nullptr;                                 // [WT][4] first group, #groups, first position, #positions
    int32_t *grp_tab = nullptr, *lm_grp_start = nullptr;        // [groups][2] {first | end << 16 of the group's tile-local positions, partial-sum slot}; landmark -> its run of slots
    uint16_t *ell_dst = nullptr;        // per ELL entry: LDS position (group-sorted order inside its wave tile), 0xFFFF = padding
    double *lm_part = nullptr;                                  // [n_groups][8] per-(wave tile, landmark) partial sums {H00 H01 H11 b0 b1 - - -}: one 64-byte line per record,
                                                                // written by one wave instruction, read as one line by the front that sums the landmark's run of slots
    // block-sparse H and b (A6/A7 output), SoA
    double *Hpp_diag = nullptr, *Hll_diag = nullptr, *Hpp_off = nullptr, *Hpl = nullptr, *b_pose = nullptr, *b_lm = nullptr;
    double *chi2_partial = nullptr; int32_t n_chi2_partial = 0; double *chi2 = nullptr;     // chi2[0] = last value
    // multifrontal plan
    DevFront *fronts = nullptr; int32_t n_fronts = 0;
    int32_t *bnd_rows = nullptr, *child_map = nullptr, *children = nullptr, *asm_recs = nullptr /* [n*4] */;
    int32_t *level_fronts = nullptr;
    int32_t *child_desc = nullptr;                              // [children][4] child front, npiv | nbnd << 16, owner, map offset
    double *Lbuf = nullptr, *Ubuf = nullptr;                   // factor and update-matrix arenas
    double *xe = nullptr;                                       // solution in elimination order [n_scalar]
    double *dpose = nullptr, *dlm = nullptr;                    // last increment per vertex
    int32_t *fail = nullptr;                                    // [0] failure code: 0 none, 1 zero pivot (H singular), 2 a whole-tree launch gave up on a
                                                                // front's flag, 3 another rank reported a failure; [1] updates applied since the last reset
    int32_t iter = 0, inject_iter = 0, inject_code = 0;         // iteration counter of this handle; gs_debug_fail_at_iteration (fault injection)
    int64_t xfail_off = -1;                                     // pose-window shards: offset of the failure slot in the exchange buffer (-1: none)
    int32_t hist_slot = -1;                                     // gs_optimize: k_update also stores the chi2 total of this iteration's linearisation point in chi2[1 + hist_slot] (< 0: no)
    double conv_tol = -1.0;                                     // gs_optimize_until: relative chi2 change that stops the iterations (< 0: no stop rule); fail[2] = the iteration (iter) in which the rule fired, 0 = not yet; chi2[70] = previous chi2
    double *front_ws = nullptr; int64_t front_ws_stride = 0;    // global workspace for fronts too big for LDS
    long long *done_ts = nullptr;                               // [2][n_fronts] F3_DONE_TS tuning builds: 100 MHz completion time of every front (factor, backsolve)
    long long *dbg_ts = nullptr;                                // [64] phase timestamps (100 MHz) of one front, GS_DBG = 8 | level_count << 8
    int32_t dbg = 0;                                            // gs_debug_options.dbg: in-kernel timestamp probes (timing experiments only)
    int32_t leaf_nt3 = 1, f3_lds_kb = 0;                        // gs_debug_options: three-tile-row leaf instance; LDS per workgroup of the per-level factor launches
    int32_t factor_variant = 0;                                 // 0 block-per-front VALU Cholesky (the C-ABI's variant 4), 3 matrix-core LDL^T fronts (default)
    double *Uimg = nullptr;                                     // variant 3: update matrices as packed lower triangles (u3_off, u3_size)
    // variant 3 (latency-shaped MFMA/LDL^T kernels): flat per-level descriptors, value-ready assembly records,
    // per-front inverse row maps into the parent (formats: gs_kernels.hip, "variant 3")
    int32_t *f3_x = nullptr;                                    // row tables + headers of the children of a front (f3x_stride ints per child)
    int32_t f3x_stride = 72;                                    // 72 = a 64-entry table + 8 header ints; 168 = a 160-entry table (plans with a front of more than 63 scalars)
    int32_t *f3_desc = nullptr, *asm3 = nullptr, *pinv = nullptr, *sc3 = nullptr, *lm3 = nullptr, *u3_off = nullptr, *u3_size = nullptr;
    uint32_t *tickets = nullptr; mutable uint32_t ticket_base = 0; // whole-tree launches: a workgroup's work is named by the ticket it takes, not by blockIdx (gs_kernels.hip, "tickets"); ticket_base = the counter's value at the launch's first workgroup (host-side running sum of the ticketed grids; nullptr: blockIdx)
    int32_t *done_f = nullptr; int32_t epoch = 0, tree = 0;     // whole-tree factor launches: per-front completion flags (= epoch when done); the backward solve polls xe itself
    double *H_arena = nullptr;                                  // Hpp_diag | b_pose | Hpp_off | Hpl | lm_part | Hll_diag | b_lm, one allocation
    // append-only growth (grow_plan): the TAIL — poses / edges added after the plan was built.  N, Epp, Epl and ell_len above stay the
    // BASE counts (they are the plane strides of the H blocks and the extent of the linearisation layout); a tail pose p >= N uses
    // slot p - N, a tail odometry edge k >= Epp slot k - Epp, a tail observation edge the virtual layout index ell_len + slot.
    // pose_est / pose_cs / pose_fixed / pose_gidx / dpose / pp_zinv / pp_info / xe are allocated with room for the tail.
    int32_t tN = 0, tM = 0, tEpp = 0, tEpl = 0;                 // tail counts (0: no tail, nothing below is touched); tail landmark l >= M: slot l - M
    int32_t tcapN = 0, tcapM = 0, tcapEpp = 0, tcapEpl = 0;     // plane strides of the tail blocks
    int32_t *t_pp_ij = nullptr;                                 // [tcapEpp][2] endpoints of the tail odometry edges
    int32_t *t_pl = nullptr; double *t_pl_z = nullptr, *t_pl_w = nullptr;    // [tcapEpl][2] {pose, landmark}, [tcapEpl][2], [tcapEpl][3]
    int32_t tLt = 0;                                            // landmarks the tail's observation edges touch
    int32_t *t_pose_start = nullptr, *t_pose_edges = nullptr;   // [tcapN+1], [tcapEpl]: tail pose -> its tail observation edges, edge order
    int32_t *t_lt_id = nullptr, *t_lt_start = nullptr, *t_lt_edges = nullptr;   // [tcapEpl], [tcapEpl+1], [tcapEpl]: touched landmark -> its tail edges, edge order
    double *t_Hpp_diag = nullptr, *t_b_pose = nullptr, *t_Hpp_off = nullptr, *t_Hpl = nullptr;   // [6][tcapN] [3][tcapN] [9][tcapEpp] [6][tcapEpl], inside H_arena
    double *t_Hll_diag = nullptr, *t_b_lm = nullptr;            // [3][tcapM] [2][tcapM], inside H_arena (lm_est / lm_fixed / lm_gidx / dlm have room for tcapM more landmarks)
    // pose-window shards (world == 1: everything is "own", no exchange)
    int32_t rank = 0, wt_lo = 0, wt_hi = 0;                     // this shard sweeps wave tiles [wt_lo, wt_hi)
    uint8_t *pose_known = nullptr, *lm_known = nullptr;         // vertex estimates tracked by this rank
    int64_t *x_off = nullptr;                                   // front -> slot offset in the exchange buffer
    double *exchange = nullptr;                                 // dense slots of the shared fronts (all-reduced)
};

// launchers (gs_kernels.hip); all asynchronous on `st`
void launch_linearize(const DevGraph &d, hipStream_t st, hipEvent_t start = nullptr, hipEvent_t stop = nullptr);   // start / stop: events attached to the dispatch itself       // fused ELL kernel when wave tiles exist, else gather kernels
void launch_linearize_gather(const DevGraph &d, hipStream_t st);
void launch_linearize_finalize(const DevGraph &d, hipStream_t st);   // H_ll, b_l, chi2 total from the fused kernel's partials
void launch_chi2_only(const DevGraph &d, hipStream_t st);
// mode: 0 own front, 1 contribution of this rank to a shared front (-> exchange), 2 shared front from the exchange
void launch_factor_level(const DevGraph &d, int level_off, int count, int max_f, int mode, hipStream_t st);
void launch_factor_tree(const DevGraph &d, int n_leaf, int leaf_slot, int leaf_max_f, int count, int n_block, int sub_first, int n_sub, hipStream_t st);   // variant 3: leaf level (+ the bottom subtrees: a level-1 front with its leaves per workgroup) + every level above
size_t factor_sub_lds_bytes(int leaf_slot);                                  // LDS of one workgroup of k_factor3_sub
void launch_factor_tree_top(const DevGraph &d, int first, int count, hipStream_t st);           // shared top of a sharded graph (mode TOP)
void launch_backsolve_tree(const DevGraph &d, int first, int count, int max_npiv, int max_f, hipStream_t st);
void launch_backsolve_level(const DevGraph &d, int level_off, int count, int max_npiv, int max_nbnd, hipStream_t st);
void launch_update(const DevGraph &d, hipStream_t st);
void launch_pose_trig(const DevGraph &d, hipStream_t st);       // pose_cs from pose_est (after every host -> device estimate copy)
void launch_polar_to_xy(int n, const double *az, const double *zen, const double *dist, double lidar, double *out, hipStream_t st);
void launch_cone_to_global(int n, const double *poses, const int32_t *pose_of_obs, const double *obs, double lidar,
                           double *out, hipStream_t st);
void launch_associate(int n, const double *poses, const int32_t *pose_of_obs, const double *obs, double lidar,
                      int n_map, const double *map_xy, const int32_t *map_type, double thr, double type_tol,
                      int32_t *out, hipStream_t st);
// batched A1 with a hashed uniform grid built on the device (no host pass over the map, no host round trip): buckets = a power of two,
// count / start / cursor: buckets + 1 ints each, items: n_map ints; start / stop: events attached to the first / last dispatch of the query
void launch_grid_build(int n_map, const double *map_xy, double thr, long long buckets, int32_t *count, int32_t *start, int32_t *cursor, int32_t *items, hipStream_t st);
void launch_associate_grid_dev(int n, const double *poses, const int32_t *pose_of_obs, const double *obs, double lidar, const double *map_xy,
                               const int32_t *map_type, double thr, double type_tol, long long buckets, const int32_t *cell_start, const int32_t *cell_items,
                               int32_t *out, int n_poses, double *pose_cs_scratch /* [n_poses][2] */, hipStream_t st, hipEvent_t start = nullptr, hipEvent_t stop = nullptr);
// structure phase on the device: expand the block assembly records into scalar / landmark records (k_build_sc3)
struct Sc3Args { int64_t off[8]; int64_t L; int32_t N, M, Epp, fused;
                 int64_t toff[6]; int32_t tcapN, tcapEpp, tcapEpl, tcapM; };   // tail blocks (grow_plan): arena offsets of t_Hpp_diag, t_b_pose, t_Hpp_off, t_Hpl, t_Hll_diag, t_b_lm; plane strides
// list != nullptr: only the fronts list[0 .. n_fronts) (growth)
void launch_build_sc3(const int32_t *bf, const int32_t *asm3, int32_t *sc3, int32_t *lm3, int n_fronts, const Sc3Args &A, hipStream_t st, const int32_t *list = nullptr);
void launch_linearize_tail(const DevGraph &d, hipStream_t st);  // the tail's edges: their blocks into the tail arenas, their shares of old vertices' diagonal blocks added in place
void launch_pose_trig_range(const DevGraph &d, int first, int count, hipStream_t st);
// growth: per-front patch records {front, DevFront (20 ints), u3_off, u3_size, bf[8]} = 32 ints each -> fronts / u3_off / u3_size / bf
void launch_apply_front_patch(int n, const int32_t *patch, DevFront *fronts, int32_t *u3_off, int32_t *u3_size, int32_t *bf, hipStream_t st);
void launch_build_ell(int64_t L, const int32_t *ell_ins, const int32_t *raw_l, const double *raw_z, const double *raw_info,
                      const int32_t *pl_rank, int rank, int32_t *ell_l, double *ell_z, double *ell_w, hipStream_t st);
void launch_frame_frontend(int k, const double *in, double lidar, int n_map, const double *map_xy, const int32_t *map_type,
                           double thr, double type_tol, int signed_type, double *out_z, double *out_g, int32_t *out_idx, hipStream_t st);
// list != nullptr: only the level positions list[0 .. nq) (growth)
void launch_build_f3(int nq, const int32_t *lf, const DevFront *fronts, const int32_t *children, const int32_t *child_map,
                     const int32_t *u3_off, const int32_t *u3_size, const int32_t *bf, const int32_t *xrow_off, const int64_t *x_off,
                     int32_t *f3_desc, int32_t *f3_x, int x_stride, hipStream_t st, const int32_t *list = nullptr, const int32_t *pos_of = nullptr);   // pos_of: front -> level position (into the children's headers)
// plans that hold a front of more than 63 scalars: table-driven whole-tree launches (workgroup -> {level position, kind | count << 8})
void launch_factor_tab(const DevGraph &d, const int2 *wgt, int n_wg, int leaf_launch_preceded, size_t lds_bytes, int cls, hipStream_t st, int mode = 0);      // cls: 0 fronts of 64-79, 1 small fronts + 80-111, 2 fronts of 112-159; mode: 0 own, 1 contribution to a shared front, 2 shared front from the exchange
void launch_backsolve_tab(const DevGraph &d, const int2 *wgt, int n_wg, int max_npiv_small, int max_f_small, size_t lds_bytes, int cls, hipStream_t st);   // cls: 0 small fronts (a wave each), 1 big fronts
size_t factor_tab_lds_bytes(int kind);                                       // LDS of one workgroup of k_factor3_tab by kind (0 waves, 1 four waves, 4 / 2 / 3 = 5 / 7 / 10 tile rows)
size_t backsolve_tab_lds_bytes(int kind, int f_or_slot_f, int npiv_small);
void launch_patch_asm3(int64_t n, int32_t *asm3, const int32_t *lm_grp_start, hipStream_t st);
int  factor_lds_limit_f();      // largest front dimension that fits the LDS variant

}  // namespace gs
