/*
 * graphslam_debug.h — tuning, fault-injection and measurement hooks of the MI355X GraphSLAM back-end.
 *
 * NOT part of the drop-in boundary (include/graphslam.h): nothing here has a counterpart in the reference (its optimiser is a
 * g2o::SparseOptimizer with no such knobs, reference src/slam.cpp:53-65; it has no fault injection).  The tests, bench.py
 * and the scripts under scripts/ use these entry points for A/B timing, for the parity checks of every launch mode and for the
 * failure-semantics tests.  A consumer of the library (the microservice) never needs this header.
 */
#ifndef GRAPHSLAM_DEBUG_H
#define GRAPHSLAM_DEBUG_H

#include "graphslam.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Every tuning switch of a handle in ONE place.  gs_create fills it once from the environment variable named with each field (so a
 * script can still steer a whole process without code) and never looks at the environment again; gs_debug_set_options replaces it.
 * -1 (or 0 where stated) = the library's default.  "plan" fields act at the next structure phase (gs_initialize_optimization /
 * the first gs_optimize after a graph change), "call" fields at the next call that consults them.
 *
 *   launch shape of the solver — numerically neutral: the GPU suite asserts bitwise equal results across them
 *     tree            GS_TREE          plan  1 whole-tree launches (fronts wait for their children's flags); 0 one launch per level
 *     block_fronts    GS_BLOCK_FRONTS  plan  levels of at most this many fronts run a WORKGROUP per front (default 512)
 *     leaf_kernel     GS_LEAF_KERNEL   plan  0 never a separate leaf launch, 2 always, -1 by leaf_min
 *     leaf_min        GS_LEAF_MIN      plan  separate leaf launches only above this many leaves (default 2048)
 *     bs_wide         GS_BS_WIDE       plan  backward solve: levels wider than this get a light launch of their own (default 2048)
 *     leaf_nt3        GS_LEAF_NT3      plan  1 three-tile-row leaf instance when every leaf has <= 47 scalars (default), 0 off
 *     f3_lds_kb       GS_F3_LDS_KB     plan  occupancy experiments: LDS per workgroup of the per-level factor launches, KB (0 = need)
 *     subtree         GS_SUBTREE       plan  1: a level-1 front and the leaves below it run in ONE workgroup (k_factor3_sub: the leaves'
 *                                      update matrices stay in LDS; measured slower); 0 (default): leaf launch + flagged
 *                                      launch from level 1 up
 *     tickets         GS_TICKETS       plan  1: a workgroup of a whole-tree launch takes its number from a counter in HBM — the launch's own order,
 *                                      whatever order the hardware dispatches in (measured: -2.4 % iterations/s at 100k poses, -5 % at 10k, -9 % on
 *                                      24-cones-in-view tracks: one same-address atomic per workgroup); 0 (default): blockIdx — grid-order dispatch,
 *                                      with the bounded polls, the per-level fallback and the retry behind it
 *   plan shape — changes the elimination order, hence the last bits of the result (all are exact factorisations)
 *     leaf_poses      GS_LEAF_POSES    plan  nested-dissection leaf size in poses (0 = gs_config.leaf_poses / default 8)
 *     cluster_ways    GS_CLUSTER_WAYS  plan  fan-out of the multi-way split above the leaves (0 = default 8; 2 = binary)
 *     ell_lanes       GS_ELL_LANES     plan  lanes per pose of the observation-edge layout (0 = by size)
 *     big_cluster     GS_BIG_CLUSTER   plan  second-pass bound of a cluster front (-1 = by the view; 0 = wave fronts only)
 *     grow_headroom   GS_GROW_HEADROOM plan  scalars a cluster front stays below a wave's 63 (-1 = default 6)
 *     factor_variant  GS_FACTOR_VARIANT plan 0 = gs_config.factor_variant; 3 matrix-core LDL^T fronts, 4 block-per-front VALU
 *   append-only growth
 *     grow            GS_GROW          call  0: every graph change is a full structure phase (default 1)
 *     grow_min_poses  GS_GROW_MIN_POSES call graphs below this many poses always rebuild (default 128)
 *   front end
 *     assoc_grid      GS_ASSOC_GRID    call  batched association: 1 uniform grid, 0 brute force, -1 by map size (default)
 *   multi-GPU
 *     force_shared_top GS_FORCE_SHARED_TOP plan  world 1 only: the top k levels of the tree are treated as the SHARED top of a sharded
 *                                      graph (contribution -> exchange buffer -> all-reduce -> redundant top), so that the collective
 *                                      path runs with a non-empty exchange buffer on one GPU (default 0 = off)
 *     shard_by_window GS_SHARD_BY_WINDOW plan 1 (default): a rank builds the top of the tree from per-landmark window masks and touches only its own
 *                                      window's edges beyond one pass (gs_plan.cpp, nd_top); 0: the general recursion over every edge (the same plan
 *                                      for equal windows, a power of two of them)
 *   experiments / diagnostics
 *     host_trig       GS_HOST_TRIG     plan  1: cos / sin of the INITIAL pose angles from the host's libm (scripts/parity_spread.py)
 *     pool_poison     GS_POOL_POISON   plan  1: device chunks are filled with 0xFF when taken and when released
 *     plan_timing     GS_PLAN_TIMING   plan  1: per-phase wall times of the structure phase on stderr
 *     dbg             GS_DBG           plan  in-kernel phase timestamps: 8 | level_count << 8, 16 | level_position << 8
 * Two process-wide variables remain outside the struct: GS_THREADS (host threads of the plan build) and, Python binding only, GS_LIB. */
typedef struct gs_debug_options {
    int32_t struct_size;
    int32_t tree, block_fronts, leaf_kernel, leaf_min, bs_wide, leaf_nt3, f3_lds_kb, reserved0;
    int32_t leaf_poses, cluster_ways, ell_lanes, big_cluster, grow_headroom, factor_variant;
    int32_t grow, grow_min_poses;
    int32_t assoc_grid;
    int32_t force_shared_top;
    int32_t host_trig, pool_poison, plan_timing, dbg;
    int32_t subtree;
    int32_t tickets;
    int32_t shard_by_window;
    int32_t reserved[5];
} gs_debug_options;

int gs_debug_options_default(gs_debug_options *o);                     /* the library's defaults (the environment is NOT consulted) */
int gs_debug_get_options(gs_graph *g, gs_debug_options *o);            /* what the handle runs with */
int gs_debug_set_options(gs_graph *g, const gs_debug_options *o);      /* "plan" fields: the next structure phase is a full one */

/* Tuning aid: with dbg = 8 | (count << 8), the factor / backsolve kernels of the level that holds `count` fronts record 100 MHz
 * timestamps at their phase boundaries for that level's first front (factor: slots 0.., backsolve: slots 32..).  Copies the 64 slots out. */
int gs_debug_timestamps(gs_graph *g, int64_t *out64);
/* Tuning aid (F3_DONE_TS builds of the library only, zeros otherwise): 100 MHz completion time of every front in the
 * last factor launch ([0, n)) and the last backward-solve launch ([n, 2n)); returns n. */
int gs_debug_front_times(gs_graph *g, int64_t *out, int64_t capacity);
/* Fault injection (tests of the failure semantics; the reference has none): the k-th iteration enqueued after
 * this call reports `code` (1 = zero pivot, 2 = front-flag timeout) from its first front; k = 0 disarms. */
int gs_debug_fail_at_iteration(gs_graph *g, int32_t k, int32_t code);
/* Measurement hook of the batched association (bench.py's `association` entry): `reps` launches of gs_associate_resident, each with a HIP
 * start / stop event pair attached to the query kernel's dispatch (its own begin -> end, on the handle's stream); mean ms per launch. */
int gs_debug_time_associate_resident(gs_graph *g, int32_t n, const double *dev_poses_xytheta, int32_t n_poses, const int32_t *dev_pose_of_obs,
                                     const double *dev_obs_4xn, double threshold, double type_tol, int32_t *dev_out_index, int32_t reps, double *out_ms);
/* Measurement hook of the sharded iteration (bench.py's `ms_exchange`): `reps` RCCL all-reduces of the exchange buffer back to back on the
 * handle's stream, HIP events around them; mean ms per all-reduce.  Collective: every rank of the communicator calls it. */
int gs_debug_time_exchange(gs_graph *g, int32_t reps, double *out_ms);
/* The factor-kernel variant a plan with the given largest front and H-arena size (doubles) is given for a requested variant
 * (gs_config.factor_variant; 0 = default): 3 = matrix-core LDL^T fronts with 32-bit byte offsets into the arena (a wave per front of
 * <= 63 scalars, a workgroup per front of 64 .. 159; arena < 2^29 doubles), 4 = block-per-front kernel with 64-bit addressing
 * (anything else).  Pure function; what gs_initialize_optimization applies. */
int gs_debug_select_factor_variant(int32_t requested, int32_t max_front, int64_t arena_doubles);

#ifdef __cplusplus
}
#endif
#endif /* GRAPHSLAM_DEBUG_H */
