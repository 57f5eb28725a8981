/*
 * graphslam.h — C-ABI of the MI355X-native GraphSLAM back-end.
 *
 * Drop-in boundary for the ONE hot path of cfsd/opendlv-logic-cfsd18-sensation-slam:
 * every call `Slam` makes on its private `g2o::SparseOptimizer m_optimizer`
 * (reference src/slam.hpp:98) plus the batched front-end arithmetic that feeds
 * it (polar->XY, cone->map association).  The reference has no FFI of its own;
 * each entry point below cites the reference call site it replaces.
 *
 * Conventions
 *  - plain C, opaque handles, caller-owned buffers, all inputs copied (g2o takes
 *    ownership of new-ed vertices/edges, we copy instead: src/slam.cpp:434-438).
 *  - every function returns an int status (GS_OK == 0, negative == error) except
 *    gs_optimize / gs_iterate which follow g2o's convention of returning the
 *    number of iterations performed (0 == factorisation failed) or a negative
 *    error code.  Nothing throws, nothing aborts.
 *  - all arithmetic is IEEE fp64, all indices int32.
 *  - pose ids and landmark ids live in SEPARATE id spaces (the reference's
 *    single namespace, cones 0.., poses 1000.., collides beyond 1000 cones:
 *    src/slam.hpp:118, src/slam.cpp:556,610).
 *  - a handle is externally synchronised (the reference holds m_optimizerMutex
 *    around every optimiser call: src/slam.cpp:324,372,402,560,586,614,627).
 *  - the compute path is HIP on gfx950 only.  There is no CPU fallback: without
 *    a usable device gs_create fails with GS_ERR_NO_DEVICE.
 */
#ifndef GRAPHSLAM_H
#define GRAPHSLAM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GS_VERSION_MAJOR 0
#define GS_VERSION_MINOR 1

/* ---- status codes ------------------------------------------------------- */
#define GS_OK                    0
#define GS_ERR_INVALID          -1  /* null pointer / bad argument              */
#define GS_ERR_DUPLICATE_ID     -2  /* g2o addVertex returns false on dup id    */
#define GS_ERR_UNKNOWN_ID       -3  /* g2o vertex(id) returns nullptr           */
#define GS_ERR_NO_DEVICE        -4  /* no gfx950 device / HIP runtime unusable  */
#define GS_ERR_HIP              -5  /* a HIP call failed (see gs_last_error)    */
#define GS_ERR_NOT_INITIALIZED  -6  /* gs_iterate before gs_initialize_optimization */
#define GS_ERR_EMPTY            -7  /* nothing to optimise (no free vertex)     */
#define GS_ERR_NUMERIC          -8  /* zero pivot: H singular (see gs_optimize) */
#define GS_ERR_CAPACITY         -9  /* output buffer too small                  */
#define GS_ERR_TIMEOUT         -10  /* a whole-tree solver launch gave up waiting for a front (see gs_stream_synchronize) */

typedef struct gs_graph gs_graph;   /* replaces g2o::SparseOptimizer (src/slam.hpp:98) */
typedef struct gs_slam  gs_slam;    /* replaces the graph-side state of class Slam     */

/* ---- configuration ------------------------------------------------------ */
typedef struct gs_config {
    int32_t struct_size;        /* sizeof(gs_config), for ABI evolution                     */
    int32_t device;             /* HIP device ordinal; -1 = current device; -2 = host-only
                                   handle (graph container + plan inspection, every compute
                                   entry point returns GS_ERR_NO_DEVICE)                    */
    int32_t verbose;            /* 1: print g2o-style "iteration= i chi2= ..." to stderr
                                   (reference: setVerbose(true), src/slam.cpp:63)          */
    int32_t leaf_poses;         /* nested-dissection leaf size in poses; 0 = default        */
    int32_t factor_variant;     /* front factorisation kernel: 0 = default (3 when every front has <= 159
                                   scalars and the H arena fits 32-bit byte offsets, else 4); 3 = LDL^T on the fp64
                                   matrix cores (v_mfma_f64_16x16x4_f64), chosen PER FRONT: a wave for a front of <= 63
                                   scalars, a workgroup for one of 64 .. 159; update matrices moved in storage order;
                                   4 = block-per-front VALU Cholesky, any front size                 */
    int32_t linearize_gather;   /* 1: force the general gather kernels instead of the fused tiled
                                   linearisation kernel (both are HIP; for tests and A/B timing)   */
    /* Slam-level constants, defaults are the reference's hard-coded values */
    double  odometry_information;   /* 5.0   (src/slam.cpp:456)                             */
    double  cone_information;       /* 0.01  (src/slam.cpp:546)                             */
    double  same_cone_threshold;    /* --sameConeThreshold, m_newConeThreshold (slam.cpp:740) */
    double  cone_mapping_threshold; /* --coneMappingThreshold (slam.cpp:744)                */
    double  lidar_to_cog;           /* 1.5 m (src/slam.cpp:514)                             */
    double  loop_closing_radius;    /* 1.0 m (src/slam.cpp:702)                             */
    int32_t loop_closing_min_index; /* 20    (src/slam.cpp:702)                             */
    int32_t optimize_iterations;    /* 10    (src/slam.cpp:481)                             */
    int32_t reference_quirks;       /* 1: keep SURVEY §8-B quirks 1-2 (duplicate first edge,
                                       re-optimise per remaining observation)              */
    int32_t optimize_every_keyframe; /* 0 (the reference): the graph is optimised once, at loop closure.  1 (NOT the reference's behaviour):
                                        gs_slam_perform also runs optimizeGraph + updateMap at the end of every keyframe that did not run
                                        them already.  The reference carries optimizeGraph() calls commented out at src/slam.cpp:594 and
                                        :620-621 (with updateMap) and at :403 (localizer: optimizeGraph ONLY, no updateMap); with this flag
                                        localizer-mode keyframes also run updateMap — the map and the resident association map are
                                        rewritten after loop closure, which restoring the reference's comments would not do.  One more
                                        keyframe does not rebuild the structure here (append-only growth), so the call costs about a
                                        millisecond at lap size; in localizer mode the published pose is then the optimised one. */
} gs_config;

/* per-call statistics of gs_optimize / gs_iterate (all times from HIP events on
 * the handle's stream, milliseconds, summed over the iterations of the call) */
typedef struct gs_stats {
    int32_t struct_size;
    int32_t iterations;         /* iterations whose update was applied (== requested unless the solve failed) */
    int32_t n_free_poses, n_free_landmarks;
    int32_t n_odometry_edges, n_observation_edges;
    int32_t n_fronts, n_levels, max_front;   /* multifrontal plan                          */
    int32_t numeric_failure;    /* 0 ok; 1 zero pivot; 2 front-flag timeout; 3 failure reported by another rank */
    double  chi2_initial;       /* chi2 at the linearisation point of the first iteration   */
    double  chi2_final;         /* chi2 at the linearisation point of the last iteration    */
    double  ms_structure;       /* host: ordering + symbolic + upload (iteration-0 work)   */
    double  ms_linearize;       /* A5+A6+A7 kernel(s)                                       */
    double  ms_factor;          /* multifrontal numeric factorisation + forward solve       */
    double  ms_backsolve;       /* backward solve                                           */
    double  ms_update;          /* A9                                                       */
    double  ms_total;           /* whole call, events                                       */
    int64_t factor_flops;       /* model flops of one factorisation                         */
    int64_t factor_bytes;       /* L + update-matrix storage, bytes                         */
    double  ms_event_overhead;  /* gs_time_iterations: an empty event-to-event interval on the
                                   stream, i.e. the share of every phase time that is measurement */
    int32_t fell_back;          /* 1: the handle runs one launch per level (the slow path) because a whole-tree launch gave
                                   up on a front's completion flag — in this call or an earlier one; 0: whole-tree launches.
                                   Not for good: the 4th gs_optimize call after a fallback tries the whole-tree launches again
                                   (then the 16th, 64th ... after each further timeout), and a new plan starts afresh */
    int32_t first_failure;      /* the first failure code this call met (0 none): a flag timeout (2) that the per-level
                                   fallback then repaired leaves numeric_failure 0 and first_failure 2 */
    int32_t factor_variant;     /* the front kernels this plan runs on (gs_config.factor_variant after the per-plan rules): 3 = LDL^T on
                                   the fp64 matrix cores (a wave per front up to 63 scalars, a workgroup per front up to 159), 4 = block VALU */
    int32_t n_big_fronts;       /* fronts of more than 63 scalars (variant 3: the ones that get a workgroup) */
    int64_t device_bytes;       /* HBM this handle's plan and graph occupy (the chunks its arrays are carved from) */
    int32_t n_own_fronts, n_shared_fronts;   /* pose-window shards: fronts this rank factorises / the shared top (world 1: all, 0) */
    double  ms_plan_host;       /* share of ms_structure spent building the plan on the host */
    double  ms_linearize_kernel; /* gs_time_iterations: the A5-A7 kernel's own begin -> end per launch, from events attached to its dispatch (ms_linearize
                                    is event to event on the stream: it also holds the hand-over from the previous kernel) */
    int32_t n_growths;          /* append-only growth steps the current plan has absorbed since the last full structure phase (0: none) */
    int32_t n_subtrees;         /* level-1 fronts that run with the leaves below them in ONE workgroup (k_factor3_sub; known after the first iteration of a plan) */
} gs_stats;

int  gs_version(void);                               /* major*100+minor */
const char *gs_last_error(void);                      /* thread-local message of the last failure */
int  gs_config_default(gs_config *cfg);
int  gs_device_count(void);                           /* number of usable gfx950 devices, 0 if none */

/* ---- lifetime: replaces Slam::setupOptimizer (src/slam.cpp:53-65) -------- */
int  gs_create(const gs_config *cfg, gs_graph **out);
int  gs_destroy(gs_graph *g);
int  gs_clear(gs_graph *g);                           /* drop all vertices and edges */
/* use an externally owned hipStream_t (e.g. torch's current stream); NULL = own stream */
int  gs_set_stream(gs_graph *g, void *hip_stream);
/* Device memory for the first structure phase, taken (and touched) now — e.g. at start-up, where the reference constructs its
 * optimizer (src/slam.cpp:36-65) — instead of inside the first gs_optimize: chunks of 8, 16, 32 ... MB until `bytes` are held.  A handle
 * keeps its device memory across structure phases anyway (a re-plan allocates nothing); this only moves the FIRST allocations out of
 * the first optimize(): 0.05 ms for a lap-sized graph, up to ~12 ms at 10k poses on machines where fresh allocations are slow.
 * gs_slam_create reserves 24 MB (a lap-sized graph needs 3-8 MB). */
int  gs_reserve_device(gs_graph *g, int64_t bytes);

/* ---- graph construction (A2) --------------------------------------------
 * gs_add_pose             <- new VertexSE2; setId; setEstimate; addVertex        (src/slam.cpp:434-438)
 * gs_add_landmark         <- new VertexPointXY; setId; setEstimate; addVertex    (src/slam.cpp:527-531)
 * gs_add_odometry_edge    <- new EdgeSE2; vertices; setMeasurement; setInformation; addEdge (src/slam.cpp:447-457)
 * gs_add_observation_edge <- new EdgeSE2PointXY; ...                              (src/slam.cpp:538-547)
 * information matrices are row-major full (9 resp. 4 doubles) and must be symmetric. */
int  gs_add_pose(gs_graph *g, int32_t pose_id, const double est_xytheta[3]);
int  gs_add_landmark(gs_graph *g, int32_t lm_id, const double est_xy[2]);
int  gs_add_odometry_edge(gs_graph *g, int32_t pose_id_i, int32_t pose_id_j,
                          const double z_xytheta[3], const double information[9]);
int  gs_add_observation_edge(gs_graph *g, int32_t pose_id, int32_t lm_id,
                             const double z_xy[2], const double information[4]);
/* bulk SoA/AoS variants (count records, arrays tightly packed, row-major per record) */
int  gs_add_poses(gs_graph *g, int32_t count, const int32_t *pose_ids, const double *est_xytheta);
int  gs_add_landmarks(gs_graph *g, int32_t count, const int32_t *lm_ids, const double *est_xy);
int  gs_add_odometry_edges(gs_graph *g, int32_t count, const int32_t *pose_ids_i,
                           const int32_t *pose_ids_j, const double *z_xytheta,
                           const double *information /* count*9, or NULL -> cfg.odometry_information*I */);
int  gs_add_observation_edges(gs_graph *g, int32_t count, const int32_t *pose_ids,
                              const int32_t *lm_ids, const double *z_xy,
                              const double *information /* count*4, or NULL -> cfg.cone_information*I */);

/* gs_set_fixed_* <- dynamic_cast<...>(vertex(id))->setFixed(true)                 (src/slam.cpp:464-474) */
int  gs_set_fixed_pose(gs_graph *g, int32_t pose_id, int32_t fixed);
int  gs_set_fixed_landmark(gs_graph *g, int32_t lm_id, int32_t fixed);
int  gs_set_pose_estimate(gs_graph *g, int32_t pose_id, const double est_xytheta[3]);
int  gs_set_landmark_estimate(gs_graph *g, int32_t lm_id, const double est_xy[2]);

/* ---- read-back (A11) -----------------------------------------------------
 * gs_get_pose     <- static_cast<VertexSE2*>(vertex(id))->estimate().toVector()   (src/slam.cpp:418-420,451-452)
 * gs_get_landmark <- static_cast<VertexPointXY*>(vertex(j))->estimate()           (src/slam.cpp:719-720) */
int  gs_get_pose(gs_graph *g, int32_t pose_id, double out_xytheta[3]);
int  gs_get_landmark(gs_graph *g, int32_t lm_id, double out_xy[2]);
int  gs_num_poses(gs_graph *g);
int  gs_num_landmarks(gs_graph *g);
int  gs_num_odometry_edges(gs_graph *g);
int  gs_num_observation_edges(gs_graph *g);
/* all vertices in insertion order; ids may be NULL */
int  gs_get_poses(gs_graph *g, int32_t capacity, int32_t *out_ids, double *out_xytheta);
int  gs_get_landmarks(gs_graph *g, int32_t capacity, int32_t *out_ids, double *out_xy);

/* ---- optimisation (A3-A10) ------------------------------------------------
 * gs_initialize_optimization <- m_optimizer.initializeOptimization()              (src/slam.cpp:480)
 *      + g2o BlockSolver::buildStructure: index maps, ordering, symbolic plan, upload to HBM.
 * gs_optimize                <- m_optimizer.optimize(10)                          (src/slam.cpp:481)
 *      runs `iterations` x (computeActiveErrors, buildSystem, solve, update) on the device,
 *      no damping, no line search, no convergence test; returns iterations done.
 *      Failure semantics are g2o's: when the factorisation of an iteration fails, that iteration's
 *      update and all later ones are NOT applied — the estimates stay at the last good iterate —
 *      and the call returns 0 (stats->iterations = updates applied, stats->numeric_failure = code).
 *      The default solver is an LDL^T like the Eigen 3.3.4 SimplicialLDLT behind g2o's
 *      LinearSolverEigen and fails like it on a pivot d == 0 only
 *      (thirdparty/Eigen/src/SparseCholesky/SimplicialCholesky_impl.h:172-176), plus on NaN;
 *      the Cholesky fallback kernel (factor_variant 4) fails on d <= 0 like SimplicialLLT.
 * gs_iterate                 one Gauss-Newton iteration, asynchronous on the handle's stream
 *      (the bench's "step"); estimates stay in HBM until gs_sync_estimates / gs_optimize.
 *      A failed iteration applies no update either; gs_stream_synchronize / gs_sync_estimates
 *      report it (GS_ERR_NUMERIC, or GS_ERR_TIMEOUT when a whole-tree launch gave up waiting for
 *      a front: the handle then switches to one launch per level) once and clear the condition.
 * gs_optimize_until          config 2 of BASELINE.json ("optimise to convergence"): the reference has no
 *      stop rule, this is the build-defined one.  Iterates until the chi2 at two
 *      consecutive linearisation points differs by <= rel_chi2_tol * chi2 (checked on the device
 *      value every iteration), at most max_iterations; returns the iterations whose update
 *      was applied, 0 on failure as gs_optimize. */
int  gs_initialize_optimization(gs_graph *g);
int  gs_optimize(gs_graph *g, int32_t iterations, gs_stats *stats /* may be NULL */);
int  gs_optimize_until(gs_graph *g, int32_t max_iterations, double rel_chi2_tol, gs_stats *stats /* may be NULL */);
int  gs_iterate(gs_graph *g);
int  gs_sync_estimates(gs_graph *g);      /* device -> host estimates, waits for the stream   */
int  gs_stream_synchronize(gs_graph *g);
/* computeActiveErrors + activeChi2 at the current estimates (device) */
int  gs_chi2(gs_graph *g, double *out_chi2);
int  gs_get_stats(gs_graph *g, gs_stats *stats);      /* plan statistics after initialize */

/* ---- measurement / parity hooks (tuning, fault injection and timestamps: include/graphslam_debug.h) ----------
 * gs_linearize: one A5+A6+A7 pass (the roofline kernel) on the stream, nothing else.
 * gs_time_linearize: `reps` back-to-back passes bracketed by HIP events on the handle's
 *      stream; returns the mean milliseconds per pass in *out_ms_per_pass.
 * gs_linearize_bytes: algorithmic bytes of one pass, SURVEY §8(d):
 *      E_pp*152 + E_pl*96 + N*120 + M*64.
 * gs_export_system: copy the block-sparse H and b of the last linearisation to the host
 *      (vertex arrays in insertion order, edge arrays in the order reported by *_edge_order):
 *      Hpp_diag [N*9], Hll_diag [M*4], Hpp_off [Epp*9] (= A^T Omega B), Hpl [Epl*6] (= A^T Omega B,
 *      3x2 row-major), b_pose [N*3], b_lm [M*2].  Any pointer may be NULL.
 * gs_export_delta: the last solve's increment, per vertex in insertion order
 *      (zeros for fixed vertices): dpose [N*3], dlm [M*2]. */
int  gs_linearize(gs_graph *g);
int  gs_time_linearize(gs_graph *g, int32_t reps, double *out_ms_per_pass);

int64_t gs_linearize_bytes(gs_graph *g);
int  gs_export_system(gs_graph *g, double *Hpp_diag, double *Hll_diag, double *Hpp_off,
                      double *Hpl, double *b_pose, double *b_lm,
                      int32_t *odometry_edge_order, int32_t *observation_edge_order);
int  gs_export_delta(gs_graph *g, double *dpose, double *dlm);
/* per-phase timing of `reps` full iterations (events on the stream); fills stats->ms_* with
 * per-iteration means.  Estimates are restored afterwards. */
int  gs_time_iterations(gs_graph *g, int32_t reps, gs_stats *stats);

/* ---- plan export (host logic, no device work; for tests of the symbolic phase) ---- */
typedef struct gs_plan_info {
    int32_t n_scalar;        /* free scalar unknowns                                          */
    int32_t n_fronts;
    int32_t n_levels;
    int32_t max_front;
    int64_t l_doubles;       /* factor storage                                                */
    int64_t u_doubles;       /* update-matrix storage                                         */
    int64_t n_asm_blocks;    /* original-entry assembly records                               */
    int64_t n_child_map;     /* extend-add map entries                                        */
} gs_plan_info;
/* builds the plan on the host only (no HIP), usable without a device */
int  gs_plan_build_host(gs_graph *g, gs_plan_info *info);
/* Append-only growth.  Per keyframe the reference adds one pose vertex with its odometry edge (src/slam.cpp:433-459), observation edges
 * to cones of the map (:537-550) and the cones it is the first to see with their first observation (:525-535); g2o's
 * initializeOptimization rebuilds everything at the next optimize() (:480).  Here, when the only changes since the last structure phase
 * are of that kind — new poses, new landmarks, edges whose pose end is a new pose; at most 16 poses / 16 landmarks / 512 + 64 edges since
 * the last full phase; every front stays within its form (63 scalars, or 159 in a plan with workgroup fronts) — gs_initialize_optimization /
 * gs_optimize keep the plan: the new vertices become
 * pivots of the root front, the fronts between a neighbour's front and the root gain them as boundary rows, and only those fronts' tables
 * are rebuilt (csrc/gs_plan.cpp grow_plan, csrc/gs_api.cpp upload_growth).  Anything else (a fixed flag, an edge between old vertices,
 * a graph below 128 poses, where there is nothing to gain; switches: gs_debug_options.grow / grow_min_poses, graphslam_debug.h) is a full structure phase.
 * gs_plan_growths: steps absorbed by the current plan; gs_growth_refusal: why the last change was NOT absorbed ("" if it was). */
int  gs_plan_growths(gs_graph *g);
const char *gs_growth_refusal(gs_graph *g);
/* flat int32 dump of the plan, see csrc/gs_plan.hpp for the layout; call with out==NULL
 * to get the required length in *out_len. */
int  gs_plan_export(gs_graph *g, int32_t *out, int64_t *out_len);

/* ---- front end (A0, A1) ----------------------------------------------------
 * gs_polar_to_xy_batch  <- Slam::Spherical2Cartesian + transformConeToCoG    (src/slam.cpp:637-654, 513-523)
 *      in : az_deg[n], zen_deg[n], dist[n]   out: xy[n*2] (CoG-frame x,y)
 * gs_cone_to_global_batch <- Slam::coneToGlobal                               (src/slam.cpp:499-510)
 *      in : pose_of_obs[n] index into poses[npose*3]; out: xy_global[n*2]
 * gs_associate_batch    <- association loop of Slam::addConesToMap against a FIXED map
 *      (src/slam.cpp:570-607; localizer variant :350-382): for each observation the LOWEST map
 *      index j with |type_j - type_i| < type_tol and Euclidean distance < threshold, else -1.
 *      obs is the reference's m_coneCollector layout, column-major 4 x n: (az, zen, dist, type).
 * All three run on the device of `g`. */
int  gs_polar_to_xy_batch(gs_graph *g, int32_t n, const double *az_deg, const double *zen_deg,
                          const double *dist, double *out_xy);
int  gs_cone_to_global_batch(gs_graph *g, int32_t n, const double *poses_xytheta, int32_t n_poses,
                             const int32_t *pose_of_obs, const double *obs_4xn, double *out_xy_global);
int  gs_associate_batch(gs_graph *g, int32_t n, const double *poses_xytheta, int32_t n_poses,
                        const int32_t *pose_of_obs, const double *obs_4xn,
                        int32_t n_map, const double *map_xy, const int32_t *map_type,
                        double threshold, double type_tol, int32_t *out_index);

/* gs_associate_resident: the same association with EVERYTHING resident — the map of gs_map_append (below; its hashed uniform grid is built on
 *      the device, once per map change), poses, observations and the result in DEVICE memory — asynchronous on the handle's stream:
 *      nothing crosses PCIe, nothing waits (gs_stream_synchronize when the caller needs the indices).  The batched form of the loop
 *      that SURVEY 8(a) calls the dominant front-end cost at scale (src/slam.cpp:570-607). */
int  gs_associate_resident(gs_graph *g, int32_t n, const double *dev_poses_xytheta, int32_t n_poses, const int32_t *dev_pose_of_obs,
                           const double *dev_obs_4xn, double threshold, double type_tol, int32_t *dev_out_index);

/* ---- the per-keyframe front end: A0 + A1 fused, against a map that stays resident in HBM -------------------------
 * gs_map_clear / gs_map_append / gs_map_set_xy / gs_map_size: the device mirror of Slam::m_map (src/slam.hpp: std::vector<Cone>):
 *      cones are appended in map order (index = Cone id, src/slam.cpp:556,610) and their positions rewritten after
 *      updateMap (src/slam.cpp:713-732).  Appends are asynchronous on the handle's stream.
 * gs_frame_frontend <- the per-frame arithmetic of addConesToMap / localizer (src/slam.cpp:570-607, 350-382): for the k
 *      observations of ONE frame (collector layout, column-major 4 x k) the CoG-frame XY (edge measurement), the global XY
 *      seen from `pose`, and the LOWEST map index with a matching type within `threshold` (-1: none) — one launch, one wait,
 *      no allocation.  signed_type != 0 selects the localizer's test `(type_j - (int)type_i) < type_tol` (no fabs, :360). */
int  gs_map_clear(gs_graph *g);
int  gs_map_append(gs_graph *g, int32_t n, const double *xy, const int32_t *type);
int  gs_map_set_xy(gs_graph *g, int32_t first, int32_t n, const double *xy);
int  gs_map_size(gs_graph *g);
int  gs_frame_frontend(gs_graph *g, const double pose_xytheta[3], const double *obs_4xk, int32_t k, double threshold,
                       double type_tol, int32_t signed_type, double *out_zxy, double *out_gxy, int32_t *out_index);

/* ---- multi-GPU: pose-window shards -----------------------------
 * Each rank owns a contiguous pose window and builds the SAME global plan; it linearises and
 * factorises only its own subtrees, exports the update contributions to the shared top of the
 * assembly tree into a dense exchange buffer (device memory), the caller all-reduces that
 * buffer (RCCL sum, fp64), and every rank finishes the top of the tree redundantly. */
int  gs_dist_configure(gs_graph *g, int32_t rank, int32_t world_size);   /* before gs_initialize_optimization */
/* Rank-local ingestion (optional).  By default every rank is given the WHOLE graph and finds out in one pass over all observation edges which
 * windows see which landmark.  A rank that is told so need not hold the other windows' observation edges at all: its plan is built from the edges
 * of its own window's poses, of every window's FIRST pose (the separators of the shared top) and of the fixed poses; everything else of the other
 * windows enters through two 64-bit masks per landmark — bit w set: an interior pose / the first pose of window w observes it.  Vertices and
 * odometry edges are still added on every rank (they carry the global indices the exchange slots are agreed on), observation edges grouped by pose.
 *   gs_dist_window_starts           insertion index of the first free pose of each window, world + 1 entries (the last: n_poses): which poses are whose
 *   gs_dist_local_landmark_windows  this rank's OWN bits, from the edges it holds (the ranks' bits are disjoint: an all-reduce SUM over the ranks —
 *                                   or an OR in a single process — gives the masks of the whole graph)
 *   gs_dist_share_landmark_windows  the two steps around the exchange in one call, over the handle's own RCCL communicator (gs_dist_comm_init first): own bits ->
 *                                   ncclAllReduce(uint64, sum) -> gs_dist_set_landmark_windows.  Collective: every rank calls it.
 *   gs_dist_set_landmark_windows    hands the whole-graph masks to the handle (n_landmarks = 0: back to the default); the next structure phase
 *                                   uses them and fails if an edge it holds contradicts them, or if the graph cannot be planned by windows
 *                                   (more than 64 ranks, fewer than 4 free poses per window, odometry edges between two windows' interiors).
 * The reference has no counterpart (one process, one optimiser: src/slam.cpp:53-65); SURVEY 8e's pose windows, without "every rank ingests everything". */
int  gs_dist_window_starts(gs_graph *g, int32_t *out_first_pose, int32_t capacity);
int  gs_dist_local_landmark_windows(gs_graph *g, uint64_t *seen_interior, uint64_t *seen_first, int32_t n_landmarks);
int  gs_dist_set_landmark_windows(gs_graph *g, const uint64_t *seen_interior, const uint64_t *seen_first, int32_t n_landmarks);
int  gs_dist_share_landmark_windows(gs_graph *g);
int64_t gs_dist_exchange_doubles(gs_graph *g);         /* length of the exchange buffer (after initialize)   */
int  gs_dist_set_exchange_buffer(gs_graph *g, void *device_ptr);   /* NULL: the library allocates its own   */
int  gs_dist_iterate_local(gs_graph *g);               /* linearise own edges + own subtrees + contribution  */
int  gs_dist_iterate_finish(gs_graph *g);              /* after the all-reduce: shared top, solve, update    */
/* The all-reduce INSIDE the library (the host side stays C++: the microservice needs no Python and no torch): RCCL is resolved at run
 * time (the copy the process has loaded already, else librccl.so), never linked.
 * gs_dist_unique_id        ncclGetUniqueId: 128 bytes, made by one rank and handed to the others by whatever channel the ranks share
 *                          (the microservices share an OD4 session: reference src/opendlv-logic-cfsd18-sensation-slam.cpp:62).
 * gs_dist_comm_init        ncclCommInitRank on the handle's device (collective: every rank calls it); the handle owns the communicator.
 * gs_dist_set_communicator adopt a caller-owned ncclComm_t instead.
 * gs_dist_iterate          one sharded Gauss-Newton iteration, all enqueued from C++ on the handle's stream: local half ->
 *                          ncclAllReduce(sum, fp64) of the exchange buffer (the shared rows of Omega / xi) -> shared top, solve, update.
 * gs_dist_optimize         Slam's optimize(10) (src/slam.cpp:481) on a sharded graph: `iterations` x gs_dist_iterate, g2o's failure rule
 *                          across ranks; returns the updates applied (0: a factorisation failed on some rank).  A whole-tree launch that gives up on a
 *                          front's flag on some rank (no property of H) is repaired inside the call: that rank switches to one launch per level, every rank
 *                          runs the iterations that were not applied again; gs_stats.first_failure = 2 / 4 says it happened (here / on another rank).  If the
 *                          launch gave up BEHIND the exchange the ranks have applied different numbers of updates: all return GS_ERR_TIMEOUT (estimates to be set again). */
int  gs_dist_unique_id(void *out_128_bytes);
int  gs_dist_comm_init(gs_graph *g, const void *unique_id_128_bytes, int32_t rank, int32_t world_size);
int  gs_dist_set_communicator(gs_graph *g, void *nccl_comm);
int  gs_dist_iterate(gs_graph *g);
int  gs_dist_optimize(gs_graph *g, int32_t iterations, gs_stats *stats /* may be NULL */);
/* host copies of the exchange buffer (tests; all-reduce over a CPU backend when ranks share one GPU) */
int  gs_dist_read_exchange(gs_graph *g, double *host_out);
int  gs_dist_write_exchange(gs_graph *g, const double *host_in);
/* per vertex (insertion order): known = this rank tracks its estimate; primary = exactly one rank per vertex
 * (shared and fixed vertices: rank 0), so summing primary-masked estimates over the ranks merges them */
int  gs_dist_known(gs_graph *g, uint8_t *pose_known, uint8_t *lm_known, uint8_t *pose_primary, uint8_t *lm_primary);

/* ---- Slam-level host mirror (rows f-1/f-2 of SURVEY §8f) ----------------------
 * gs_slam_perform <- Slam::performSLAM (src/slam.cpp:298-338): the 200 m odometry guard, the heading compensated by
 *      yawRate * dt (0 < dt < 1 s, dt from gs_slam_set_sample_times), addPoseToGraph, then — two independent ifs as in
 *      the reference (:329-334) — addConesToMap (loop-closure trigger, optimizeGraph, updateMap) while the loop is open
 *      and the localizer once it is closed, i.e. BOTH in the frame that closes it. */
int  gs_slam_create(const gs_config *cfg, gs_slam **out);
int  gs_slam_destroy(gs_slam *s);
int  gs_slam_perform(gs_slam *s, const double odometry_xytheta[3], const double *cones_4xk, int32_t k);
int  gs_slam_map_size(gs_slam *s);
int  gs_slam_get_map(gs_slam *s, int32_t capacity, double *out_xy, int32_t *out_type);
int  gs_slam_loop_closed(gs_slam *s);
int  gs_slam_current_cone_index(gs_slam *s);
int  gs_slam_get_send_pose(gs_slam *s, double out_xytheta[3]);
gs_graph *gs_slam_graph(gs_slam *s);
/* ---- frame collector and output encoders (row f-2) ---------------------------
 * gs_slam_collect_direction / _distance / _type <- Slam::nextCone (src/slam.cpp:67-152): one field of column objectId of
 *      the 4 x 1000 collector; returns 1 when the message opened a new frame, 0 otherwise, < 0 on error
 *      (objectId >= 1000 is refused; the reference indexes unchecked).
 * gs_slam_collect_flush <- Slam::initializeCollection (src/slam.cpp:221-257) after its wait, without the keyframe gate:
 *      extracts the leftmost lastObjectId + 1 columns (also copied to cones_out_4xk when not NULL, count in *k_out),
 *      resets the collector and runs gs_slam_perform on them with the given odometry pose.
 * gs_slam_encode_cones <- Slam::sendCones + Cone::getDirection / getDistance (src/slam.cpp:656-677, src/cone.cpp:34-53):
 *      conesPerPacket cones from currentConeIndex on, wrapping around the map, seen from the send pose; float32 fields
 *      as in the messages (azimuth in degrees, zenith is always 0).  cfg.reference_quirks keeps the reference's heading
 *      unit slip. */
int  gs_slam_collect_direction(gs_slam *s, uint32_t object_id, double azimuth_deg, double zenith_deg);
int  gs_slam_collect_distance(gs_slam *s, uint32_t object_id, double distance);
int  gs_slam_collect_type(gs_slam *s, uint32_t object_id, uint32_t type);
int  gs_slam_collect_flush(gs_slam *s, const double pose_xytheta[3], int32_t *k_out, double *cones_out_4xk);
/* the first half of gs_slam_collect_flush alone: extract the leftmost lastObjectId + 1 columns and reset the collector */
int  gs_slam_collect_extract(gs_slam *s, int32_t *k_out, double *cones_out_4xk);
int  gs_slam_encode_cones(gs_slam *s, int32_t cones_per_packet, float *azimuth_deg, float *distance, int32_t *type);
/* gs_cone_encode <- Cone::getDirection / Cone::getDistance (src/cone.cpp:34-53) for ONE cone seen from `pose`: the float32
 * azimuthAngle (degrees) and distance fields (zenithAngle is always 0).  reference_quirks != 0 keeps the reference's unit slip
 * (heading * 1 / RAD2DEG, src/cone.cpp:37-39).  Stateless; gs_slam_encode_cones calls it per cone.  Pinned against the reference's
 * own cone.cpp (oracle/_ref/libref_cone.so, tests/test_cone.py). */
int  gs_cone_encode(const double cone_xy[2], const double pose_xytheta[3], int32_t reference_quirks, float *azimuth_deg, float *distance);
/* ---- odometry intake and pose output (row f-4), host side -------------------------
 * gs_wgs84_to_cartesian / gs_wgs84_from_cartesian <- wgs84::toCartesian / fromCartesian (src/WGS84toCartesian.hpp:39-113,
 *      119-146): ellipsoidal polyconic projection about the reference point and its step-search inverse; arrays are
 *      {latitude, longitude} in degrees and {x, y} in metres.
 * gs_slam_set_gps_reference       <- m_gpsReference (command line of the microservice)
 * gs_slam_next_wgs84 / _heading   <- Slam::nextSplitPose (src/slam.cpp:154-185; heading wrapped with the float PI)
 * gs_slam_next_geolocation        <- Slam::nextPose (src/slam.cpp:187-209)
 * gs_slam_next_yaw_rate           <- Slam::nextYawRate (src/slam.cpp:211-219)
 * gs_slam_get_odometry            -> m_odometryData (x, y, heading) and m_yawRate: the pose gs_slam_perform is fed
 * gs_slam_encode_pose             <- Slam::sendPose (src/slam.cpp:679-695): {longitude, latitude, heading} as float32;
 *      cfg.reference_quirks keeps the reference's swapped latitude / longitude fields. */
int  gs_wgs84_to_cartesian(const double ref_latlon_deg[2], const double pos_latlon_deg[2], double out_xy[2]);
int  gs_wgs84_from_cartesian(const double ref_latlon_deg[2], const double xy[2], double out_latlon_deg[2]);
int  gs_slam_set_gps_reference(gs_slam *s, double latitude_deg, double longitude_deg);
int  gs_slam_next_wgs84(gs_slam *s, double latitude_deg, double longitude_deg);
int  gs_slam_next_heading(gs_slam *s, double north_heading);
int  gs_slam_next_geolocation(gs_slam *s, double latitude_deg, double longitude_deg, double heading);
int  gs_slam_next_yaw_rate(gs_slam *s, double angular_velocity_z);
/* m_yawReceivedTime / m_lastTimeStamp (src/slam.cpp:216; :73,102,129): sample times (microseconds) of the last yaw-rate
 * message and the last cone message; performSLAM's dt = |difference| / 1e6 (src/slam.cpp:309) */
int  gs_slam_set_sample_times(gs_slam *s, int64_t yaw_received_us, int64_t last_cone_us);
int  gs_slam_get_odometry(gs_slam *s, double out_xy_heading_yawrate[4]);
int  gs_slam_encode_pose(gs_slam *s, float out_lon_lat_heading[3]);

/* ---- the microservice shell (row f-3), transport-independent ---------------------------------------------------------
 * Mirrors main() of the reference (src/opendlv-logic-cfsd18-sensation-slam.cpp:49-119) around the Slam mirror: the same
 * command-line keys and the "at least 10 arguments" rule (:52), the seven data triggers behind their senderStamp filters
 * (:71-108), the gathering window and the keyframe gate (src/slam.cpp:221-257, 286-295), and the messages the localizer
 * publishes (sendPose + sendCones, :404-410, 656-695).  Messages cross this boundary decoded; csrc/gs_shell_cluon.cpp binds
 * it to cluon's OD4Session (Envelope decode / od4.send).  Type ids are the message set's: 19 GeodeticWgs84Reading
 * {v0 latitude, v1 longitude}, 1051 GeodeticHeadingReading {v0 northHeading}, 1116 Geolocation {v0 latitude, v1 longitude,
 * v2 heading}, 1031 AngularVelocityReading {v0 angularVelocityZ}, 1133 ObjectDirection {object_id, v0 azimuth, v1 zenith},
 * 1134 ObjectDistance {object_id, v0 distance}, 1131 ObjectType {object_id, v0 type}. */
typedef struct gs_shell gs_shell;
typedef struct gs_shell_msg {
    int32_t  data_type;
    uint32_t sender_stamp;
    int64_t  sample_time_us;
    uint32_t object_id;
    uint32_t reserved;
    double   v[3];
} gs_shell_msg;
/* argv as main() receives it; fewer than 10 arguments or a missing key -> GS_ERR_INVALID with the usage text in
 * gs_last_error (the reference prints it and returns 1).  device as in gs_config (-2: host-only, for tests of the dispatch). */
int  gs_shell_create(int32_t argc, const char *const *argv, int32_t device, gs_shell **out);
int  gs_shell_destroy(gs_shell *sh);
/* one incoming message; now_us = the caller's clock (starts the gathering window of a frame).  1 taken, 0 ignored. */
int  gs_shell_on_message(gs_shell *sh, const gs_shell_msg *m, int64_t now_us);
/* runs the frame whose gathering window (--gatheringTimeMs) has passed: extract, keyframe gate (--timeBetweenKeyframes, ms),
 * performSLAM, outputs.  1 performSLAM ran, 0 nothing due / not a keyframe. */
int  gs_shell_poll(gs_shell *sh, int64_t now_us);
int  gs_shell_pending_output(gs_shell *sh);
int  gs_shell_take_output(gs_shell *sh, int32_t capacity, gs_shell_msg *out);
int  gs_shell_counters(gs_shell *sh, int64_t out_run_gated[2]);
int  gs_shell_cid(gs_shell *sh);
gs_slam *gs_shell_slam(gs_shell *sh);

#ifdef __cplusplus
}
#endif
#endif /* GRAPHSLAM_H */
