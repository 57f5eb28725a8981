/*
 * graphslam_oracle.h — CPU ORACLE (test infrastructure, NOT product code).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 * The product library (libgraphslam_hip.so) never links or calls it.
 *
 * PARITY STATUS: "parity unpinned" for the g2o arithmetic (A5-A7, A9, A10): the reference
 * holds no golden vectors, known-answer tests or fixtures for this path (its only test is
 * REQUIRE(5+6==11), reference test/tests-logic-cfsd18-sensation-slam.cpp:26-30) and g2o
 * (github.com/RainerKuemmerle/g2o, unpinned HEAD of mid-2018, reference Dockerfile.amd64:32)
 * is absent from the container.  The restatement follows g2o's published EdgeSE2 /
 * EdgeSE2PointXY / VertexSE2 / BlockSolver algorithms and is checked by
 * finite-difference Jacobians, closed-form micro-graphs and solver-independence properties.
 * The linear solve (A8) IS pinned: oracle/_ref/libref_eigen.so compiles the reference's own
 * vendored Eigen 3.3.4 SimplicialLDLT + AMD (reference thirdparty/Eigen/src/SparseCholesky/
 * SimplicialCholesky.h:421-494) where it lies, and tests compare this file's LDLT against it.
 */
#ifndef GRAPHSLAM_ORACLE_H
#define GRAPHSLAM_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_graph orc_graph;

/* external solver hook: factor+solve an upper-CCS SPD system; analyze!=0 on the first call
 * after a structure change.  Returns 0 on success. */
typedef int (*orc_solver_fn)(void *ctx, int analyze, int n, const int *colptr, const int *rowind,
                             const double *values, const double *b, double *x);

orc_graph *orc_create(void);
void orc_destroy(orc_graph *g);

/* A2: vertices/edges are addressed by insertion index (0-based) per kind */
int  orc_add_pose(orc_graph *g, const double est[3]);
int  orc_add_landmark(orc_graph *g, const double est[2]);
int  orc_add_odometry_edge(orc_graph *g, int i, int j, const double z[3], const double info[9]);
int  orc_add_observation_edge(orc_graph *g, int p, int l, const double z[2], const double info[4]);
int  orc_add_poses(orc_graph *g, int n, const double *est);
int  orc_add_landmarks(orc_graph *g, int n, const double *est);
int  orc_add_odometry_edges(orc_graph *g, int n, const int *i, const int *j, const double *z, const double *info);
int  orc_add_observation_edges(orc_graph *g, int n, const int *p, const int *l, const double *z, const double *info);
void orc_set_fixed_pose(orc_graph *g, int i, int fixed);
void orc_set_fixed_landmark(orc_graph *g, int l, int fixed);
int  orc_num_poses(const orc_graph *g);
int  orc_num_landmarks(const orc_graph *g);
int  orc_num_odometry_edges(const orc_graph *g);
int  orc_num_observation_edges(const orc_graph *g);
void orc_get_poses(const orc_graph *g, double *out);       /* [N*3] */
void orc_get_landmarks(const orc_graph *g, double *out);   /* [M*2] */
void orc_set_poses(orc_graph *g, const double *in);
void orc_set_landmarks(orc_graph *g, const double *in);

/* A0: reference src/slam.cpp:513-523, 637-654, 499-510 */
void orc_transform_cone_to_cog(double angle_deg, double distance, double lidar_to_cog, double out[2]);
void orc_spherical_to_cartesian(double az_deg, double zen_deg, double dist, double lidar_to_cog, double out[3]);
void orc_cone_to_global(const double pose[3], const double obs[4], double lidar_to_cog, double out[3]);
void orc_polar_to_xy_batch(int n, const double *az, const double *zen, const double *dist,
                           double lidar_to_cog, double *out_xy);
void orc_cone_to_global_batch(int n, const double *poses, const int *pose_of_obs, const double *obs_4xn,
                              double lidar_to_cog, double *out_xy);
/* A1 against a fixed map: first map index (insertion order) with same type and dist < thr */
void orc_associate_fixed_map(int n, const double *poses, const int *pose_of_obs, const double *obs_4xn,
                             int n_map, const double *map_xy, const int *map_type,
                             double thr, double type_tol, double lidar_to_cog, int *out_index);

/* SE2 helpers, SURVEY §8-A.1 */
double orc_normalize_theta(double th);
void orc_se2_compose(const double a[3], const double b[3], double out[3]);
void orc_se2_inverse(const double a[3], double out[3]);

/* edge arithmetic §8-A.2/3: errors and Jacobians (row-major) */
void orc_edge_se2(const double xi[3], const double xj[3], const double z[3],
                  double e[3], double A[9], double B[9]);
void orc_edge_se2_pointxy(const double xp[3], const double l[2], const double z[2],
                          double e[2], double A[6], double B[4]);

/* A5 */
double orc_chi2(const orc_graph *g);

/* A6+A7 per block, same layout as gs_export_system (vertex arrays in insertion order):
 * Hpp_diag [N*9], Hll_diag [M*4], Hpp_off [Epp*9], Hpl [Epl*6], b_pose [N*3], b_lm [M*2];
 * contributions of edges into fixed vertices are skipped exactly as §8-A.4 says. */
void orc_linearize_blocks(const orc_graph *g, double *Hpp_diag, double *Hll_diag, double *Hpp_off,
                          double *Hpl, double *b_pose, double *b_lm);

/* A4+A6+A7 into the scalar upper CCS g2o hands to Eigen: free landmarks first, then free poses */
int  orc_build_system(orc_graph *g);          /* returns n (free scalars) */
int  orc_system_n(const orc_graph *g);
int  orc_system_nnz(const orc_graph *g);
const int *orc_system_colptr(const orc_graph *g);
const int *orc_system_rowind(const orc_graph *g);
const double *orc_system_values(const orc_graph *g);
const double *orc_system_b(const orc_graph *g);

/* A8 own restatement of Eigen's up-looking LDL^T.  ordering: 0 natural, 1 track interleave, 2 caller-supplied (orc_set_elimination_order) */
int  orc_solve_ldlt(orc_graph *g, int ordering, double *x /* [n] */);
/* ordering 2 = a caller-supplied elimination order (perm[new] = old scalar, numbering of orc_vertex_offsets): the order of
 * another exact solver, so that what differs between the two increments is arithmetic, not ordering */
int  orc_set_elimination_order(orc_graph *g, const int *perm, int n);
int  orc_vertex_offsets(const orc_graph *g, int *pose_off /* [np] */, int *lm_off /* [nl] */);   /* returns n; -1 before orc_build_system */
/* A9 */
void orc_apply_update(orc_graph *g, const double *x);
/* last increment per vertex in insertion order (zeros for fixed) */
void orc_get_delta(const orc_graph *g, double *dpose, double *dlm);

/* A10: iterations x (errors, build, solve, update); chi2_out[it] = chi2 at the linearisation
 * point of iteration it (may be NULL); solver==NULL -> own LDLT with `ordering`.
 * timings_ms[5] (may be NULL) accumulates linearise+assemble / analyse / factor+solve / update / total.
 * returns the iterations whose update was applied.  g2o's SparseOptimizer::optimize leaves its loop at the first
 * failed solve, BEFORE that iteration's update, and returns 0 then (reference call site src/slam.cpp:481); the
 * vertices keep the previous iterate.  *failed (may be NULL) tells the two outcomes apart. */
int  orc_optimize(orc_graph *g, int iterations, int ordering, orc_solver_fn solver, void *solver_ctx,
                  double *chi2_out, double *timings_ms);
/* the same with the build-defined stop rule of BASELINE config 2 ("optimise to convergence"; the reference has none,
 * SURVEY §0.5): after the update of iteration it >= 1, stop if |chi2[it-1] - chi2[it]| <= rel_chi2_tol * chi2[it]
 * (chi2 at the linearisation points).  rel_chi2_tol < 0: no stop rule. */
int  orc_optimize_until(orc_graph *g, int max_iterations, double rel_chi2_tol, int ordering, orc_solver_fn solver,
                        void *solver_ctx, double *chi2_out, double *timings_ms, int *failed);

#ifdef __cplusplus
}
#endif
#endif
