/*
 * graphslam_oracle.c — CPU ORACLE (test infrastructure, NOT product code; see the header for
 * who may use it and for the parity status: g2o arithmetic "parity unpinned", A8 pinned
 * against the reference's vendored Eigen via oracle/_ref).
 *
 * Plain C restatement, fp64, single thread, of the hot path the reference drives through
 * g2o + Eigen (reference src/slam.cpp:461-484 -> g2o -> thirdparty/Eigen), following
 * SURVEY.md §8-A.  Every function cites what it restates.
 */
#include "graphslam_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* constants exactly as the reference declares them (src/slam.hpp:134-136): PI is a FLOAT literal */
static const double ORC_DEG2RAD = 0.017453292522222;
static const double ORC_RAD2DEG = 57.295779513082325;
static const double ORC_PI = 3.14159265f;

struct orc_graph {
    int np, cap_p; double *pose; unsigned char *pfix;          /* [np*3] */
    int nl, cap_l; double *lm;   unsigned char *lfix;          /* [nl*2] */
    int npp, cap_pp; int *pp_i, *pp_j; double *pp_z, *pp_info; /* z [3], info [9] */
    int npl, cap_pl; int *pl_p, *pl_l; double *pl_z, *pl_info; /* z [2], info [4] */
    /* system (scalar upper CCS) */
    int n, nnz; int *colptr, *rowind; double *values, *b;
    int *pose_off, *lm_off;                                     /* scalar offset or -1 */
    int *pp_pos, *pl_pos;                                       /* positions of edge off-diagonal entries */
    int *pdiag_pos, *ldiag_pos;                                 /* positions of diagonal block entries */
    int structure_valid;
    /* LDLT workspace */
    int *perm, *iperm, *Cp, *Ci; double *Cx; int *cmap;         /* permuted matrix + map from A entries */
    int *parent, *Lp, *Li, *Lnz, *flag, *pattern; double *Lx, *D, *y;
    int ldlt_ordering, ldlt_valid;
    int *user_perm; int user_perm_n;                              /* ordering 2: caller-supplied elimination order (perm[new] = old) */
    double *delta;                                              /* last x [n] */
};

static double now_ms(void) {
    struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

orc_graph *orc_create(void) { return (orc_graph *)calloc(1, sizeof(orc_graph)); }

static void free_system(orc_graph *g) {
    free(g->colptr); free(g->rowind); free(g->values); free(g->b);
    free(g->pose_off); free(g->lm_off); free(g->pp_pos); free(g->pl_pos);
    free(g->pdiag_pos); free(g->ldiag_pos);
    free(g->perm); free(g->iperm); free(g->Cp); free(g->Ci); free(g->Cx); free(g->cmap);
    free(g->parent); free(g->Lp); free(g->Li); free(g->Lnz); free(g->flag); free(g->pattern);
    free(g->Lx); free(g->D); free(g->y); free(g->delta);
    g->colptr = g->rowind = NULL; g->values = g->b = NULL;
    g->pose_off = g->lm_off = g->pp_pos = g->pl_pos = g->pdiag_pos = g->ldiag_pos = NULL;
    g->perm = g->iperm = g->Cp = g->Ci = g->cmap = NULL; g->Cx = NULL;
    g->parent = g->Lp = g->Li = g->Lnz = g->flag = g->pattern = NULL;
    g->Lx = g->D = g->y = g->delta = NULL;
    g->structure_valid = 0; g->ldlt_valid = 0;
}

void orc_destroy(orc_graph *g) {
    if (!g) return;
    free_system(g);
    free(g->pose); free(g->pfix); free(g->lm); free(g->lfix);
    free(g->pp_i); free(g->pp_j); free(g->pp_z); free(g->pp_info);
    free(g->pl_p); free(g->pl_l); free(g->pl_z); free(g->pl_info);
    free(g->user_perm);
    free(g);
}

#define GROW(ptr, cap, need, elems, type) do { if ((need) > (cap)) { int nc = (cap) ? (cap) * 2 : 1024; \
    while (nc < (need)) nc *= 2; (ptr) = (type *)realloc((ptr), (size_t)nc * (elems) * sizeof(type)); } } while (0)

/* A2: reference src/slam.cpp:434-438 (VertexSE2) */
int orc_add_pose(orc_graph *g, const double est[3]) {
    int need = g->np + 1;
    if (need > g->cap_p) { int nc = g->cap_p ? g->cap_p * 2 : 1024;
        g->pose = (double *)realloc(g->pose, (size_t)nc * 3 * sizeof(double));
        g->pfix = (unsigned char *)realloc(g->pfix, (size_t)nc); g->cap_p = nc; }
    memcpy(g->pose + 3 * g->np, est, 3 * sizeof(double)); g->pfix[g->np] = 0;
    g->structure_valid = 0; return g->np++;
}
/* A2: reference src/slam.cpp:527-531 (VertexPointXY) */
int orc_add_landmark(orc_graph *g, const double est[2]) {
    int need = g->nl + 1;
    if (need > g->cap_l) { int nc = g->cap_l ? g->cap_l * 2 : 1024;
        g->lm = (double *)realloc(g->lm, (size_t)nc * 2 * sizeof(double));
        g->lfix = (unsigned char *)realloc(g->lfix, (size_t)nc); g->cap_l = nc; }
    memcpy(g->lm + 2 * g->nl, est, 2 * sizeof(double)); g->lfix[g->nl] = 0;
    g->structure_valid = 0; return g->nl++;
}
/* A2: reference src/slam.cpp:447-457 (EdgeSE2) */
int orc_add_odometry_edge(orc_graph *g, int i, int j, const double z[3], const double info[9]) {
    if (i < 0 || j < 0 || i >= g->np || j >= g->np) return -1;
    int need = g->npp + 1;
    if (need > g->cap_pp) { int nc = g->cap_pp ? g->cap_pp * 2 : 1024;
        g->pp_i = (int *)realloc(g->pp_i, (size_t)nc * sizeof(int));
        g->pp_j = (int *)realloc(g->pp_j, (size_t)nc * sizeof(int));
        g->pp_z = (double *)realloc(g->pp_z, (size_t)nc * 3 * sizeof(double));
        g->pp_info = (double *)realloc(g->pp_info, (size_t)nc * 9 * sizeof(double)); g->cap_pp = nc; }
    g->pp_i[g->npp] = i; g->pp_j[g->npp] = j;
    memcpy(g->pp_z + 3 * g->npp, z, 3 * sizeof(double));
    memcpy(g->pp_info + 9 * g->npp, info, 9 * sizeof(double));
    g->structure_valid = 0; return g->npp++;
}
/* A2: reference src/slam.cpp:538-547 (EdgeSE2PointXY) */
int orc_add_observation_edge(orc_graph *g, int p, int l, const double z[2], const double info[4]) {
    if (p < 0 || l < 0 || p >= g->np || l >= g->nl) return -1;
    int need = g->npl + 1;
    if (need > g->cap_pl) { int nc = g->cap_pl ? g->cap_pl * 2 : 1024;
        g->pl_p = (int *)realloc(g->pl_p, (size_t)nc * sizeof(int));
        g->pl_l = (int *)realloc(g->pl_l, (size_t)nc * sizeof(int));
        g->pl_z = (double *)realloc(g->pl_z, (size_t)nc * 2 * sizeof(double));
        g->pl_info = (double *)realloc(g->pl_info, (size_t)nc * 4 * sizeof(double)); g->cap_pl = nc; }
    g->pl_p[g->npl] = p; g->pl_l[g->npl] = l;
    memcpy(g->pl_z + 2 * g->npl, z, 2 * sizeof(double));
    memcpy(g->pl_info + 4 * g->npl, info, 4 * sizeof(double));
    g->structure_valid = 0; return g->npl++;
}
/* bulk variants (test convenience) */
int orc_add_poses(orc_graph *g, int n, const double *est) { for (int i = 0; i < n; ++i) orc_add_pose(g, est + 3 * i); return g->np; }
int orc_add_landmarks(orc_graph *g, int n, const double *est) { for (int i = 0; i < n; ++i) orc_add_landmark(g, est + 2 * i); return g->nl; }
int orc_add_odometry_edges(orc_graph *g, int n, const int *i, const int *j, const double *z, const double *info) {
    for (int k = 0; k < n; ++k) if (orc_add_odometry_edge(g, i[k], j[k], z + 3 * k, info + 9 * k) < 0) return -1;
    return g->npp; }
int orc_add_observation_edges(orc_graph *g, int n, const int *p, const int *l, const double *z, const double *info) {
    for (int k = 0; k < n; ++k) if (orc_add_observation_edge(g, p[k], l[k], z + 2 * k, info + 4 * k) < 0) return -1;
    return g->npl; }
/* A3: reference src/slam.cpp:464-474 */
void orc_set_fixed_pose(orc_graph *g, int i, int fixed) { if (i >= 0 && i < g->np) { g->pfix[i] = (unsigned char)(fixed != 0); g->structure_valid = 0; } }
void orc_set_fixed_landmark(orc_graph *g, int l, int fixed) { if (l >= 0 && l < g->nl) { g->lfix[l] = (unsigned char)(fixed != 0); g->structure_valid = 0; } }
int orc_num_poses(const orc_graph *g) { return g->np; }
int orc_num_landmarks(const orc_graph *g) { return g->nl; }
int orc_num_odometry_edges(const orc_graph *g) { return g->npp; }
int orc_num_observation_edges(const orc_graph *g) { return g->npl; }
void orc_get_poses(const orc_graph *g, double *out) { memcpy(out, g->pose, (size_t)g->np * 3 * sizeof(double)); }
void orc_get_landmarks(const orc_graph *g, double *out) { memcpy(out, g->lm, (size_t)g->nl * 2 * sizeof(double)); }
void orc_set_poses(orc_graph *g, const double *in) { memcpy(g->pose, in, (size_t)g->np * 3 * sizeof(double)); }
void orc_set_landmarks(orc_graph *g, const double *in) { memcpy(g->lm, in, (size_t)g->nl * 2 * sizeof(double)); }

/* ---------------- A0: polar -> XY ---------------- */

/* Slam::transformConeToCoG, reference src/slam.cpp:513-523 (sign is NaN at angle == 0: kept) */
void orc_transform_cone_to_cog(double angle, double distance, double lidar_to_cog, double out[2]) {
    double sign = angle / fabs(angle);
    angle = ORC_PI - fabs(angle * ORC_DEG2RAD);
    double distance_new = sqrt(lidar_to_cog * lidar_to_cog + distance * distance
                               - 2 * lidar_to_cog * distance * cos(angle));
    double angle_new = asin((sin(angle) * distance) / distance_new) * ORC_RAD2DEG;
    out[0] = angle_new * sign; out[1] = distance_new;
}
/* Slam::Spherical2Cartesian, reference src/slam.cpp:637-654 */
void orc_spherical_to_cartesian(double az, double zen, double dist, double lidar_to_cog, double out[3]) {
    double t[2]; orc_transform_cone_to_cog(az, dist, lidar_to_cog, t);
    az = t[0]; dist = t[1];
    out[0] = dist * cos(zen * ORC_DEG2RAD) * cos(az * ORC_DEG2RAD);
    out[1] = dist * cos(zen * ORC_DEG2RAD) * sin(az * ORC_DEG2RAD);
    out[2] = dist * sin(zen * ORC_DEG2RAD);
}
/* Slam::coneToGlobal, reference src/slam.cpp:499-510; obs = (az, zen, dist, type) */
void orc_cone_to_global(const double pose[3], const double obs[4], double lidar_to_cog, double out[3]) {
    double c[3]; orc_spherical_to_cartesian(obs[0], obs[1], obs[2], lidar_to_cog, c);
    double nx = c[0] * cos(pose[2]) - c[1] * sin(pose[2]);
    double ny = c[0] * sin(pose[2]) + c[1] * cos(pose[2]);
    out[0] = nx + pose[0]; out[1] = ny + pose[1]; out[2] = obs[3];
}
void orc_polar_to_xy_batch(int n, const double *az, const double *zen, const double *dist,
                           double lidar_to_cog, double *out_xy) {
    for (int i = 0; i < n; ++i) { double c[3];
        orc_spherical_to_cartesian(az[i], zen[i], dist[i], lidar_to_cog, c);
        out_xy[2 * i] = c[0]; out_xy[2 * i + 1] = c[1]; }
}

void orc_cone_to_global_batch(int n, const double *poses, const int *pose_of_obs, const double *obs,
                              double lidar_to_cog, double *out_xy) {
    for (int i = 0; i < n; ++i) { double c[3];
        orc_cone_to_global(poses + 3 * pose_of_obs[i], obs + 4 * i, lidar_to_cog, c);
        out_xy[2 * i] = c[0]; out_xy[2 * i + 1] = c[1]; }
}

/* ---------------- A1: association against a fixed map ----------------
 * inner loop of Slam::addConesToMap, reference src/slam.cpp:570-607 with distanceBetweenCones
 * :708-711: scan the map in insertion order, take the FIRST j with fabs(type_j - type_i) < tol
 * and Euclidean distance < threshold. */
void orc_associate_fixed_map(int n, const double *poses, const int *pose_of_obs, const double *obs,
                             int n_map, const double *map_xy, const int *map_type,
                             double thr, double type_tol, double lidar_to_cog, int *out_index) {
    for (int i = 0; i < n; ++i) {
        double gc[3]; orc_cone_to_global(poses + 3 * pose_of_obs[i], obs + 4 * i, lidar_to_cog, gc);
        int found = -1;
        for (int j = 0; j < n_map && found < 0; ++j) {
            if (fabs((double)map_type[j] - obs[4 * i + 3]) < type_tol) {
                double dx = map_xy[2 * j] - gc[0], dy = map_xy[2 * j + 1] - gc[1];
                double d = sqrt(dx * dx + dy * dy);
                if (d < thr) found = j;
            }
        }
        out_index[i] = found;
    }
}

/* ---------------- SE2, SURVEY §8-A.1 (g2o se2.h) ---------------- */
double orc_normalize_theta(double th) {
    if (th >= -M_PI && th < M_PI) return th;
    double m = floor(th / (2 * M_PI));
    th = th - m * 2 * M_PI;
    if (th >= M_PI) th -= 2 * M_PI;
    if (th < -M_PI) th += 2 * M_PI;
    return th;
}
/* a*b : t = t_a + R(th_a) t_b, th = normalize(th_a + th_b) */
void orc_se2_compose(const double a[3], const double b[3], double out[3]) {
    double c = cos(a[2]), s = sin(a[2]);
    double x = a[0] + (c * b[0] - s * b[1]);
    double y = a[1] + (s * b[0] + c * b[1]);
    out[0] = x; out[1] = y; out[2] = orc_normalize_theta(a[2] + b[2]);
}
/* a^-1 : th' = normalize(-th), t' = R(th') * (-t) */
void orc_se2_inverse(const double a[3], double out[3]) {
    double th = orc_normalize_theta(-a[2]);
    double c = cos(th), s = sin(th);
    double tx = -a[0], ty = -a[1];
    out[0] = c * tx - s * ty; out[1] = s * tx + c * ty; out[2] = th;
}

/* EdgeSE2 (g2o edge_se2.h/.cpp; §8-A.2): e = vec(z^-1 * (xi^-1 * xj)); analytic Jacobians */
void orc_edge_se2(const double xi[3], const double xj[3], const double z[3],
                  double e[3], double A[9], double B[9]) {
    double zi[3], xii[3], rel[3], d[3];
    orc_se2_inverse(z, zi);
    orc_se2_inverse(xi, xii);
    orc_se2_compose(xii, xj, rel);
    orc_se2_compose(zi, rel, d);
    e[0] = d[0]; e[1] = d[1]; e[2] = d[2];
    if (!A) return;
    double si = sin(xi[2]), ci = cos(xi[2]);
    double dx = xj[0] - xi[0], dy = xj[1] - xi[1];
    double Ji[9] = { -ci, -si, -si * dx + ci * dy,
                      si, -ci, -ci * dx - si * dy,
                      0, 0, -1 };
    double Jj[9] = { ci, si, 0, -si, ci, 0, 0, 0, 1 };
    double cz = cos(zi[2]), sz = sin(zi[2]);
    double Z[9] = { cz, -sz, 0, sz, cz, 0, 0, 0, 1 };
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) {
        double a = 0, b = 0;
        for (int k = 0; k < 3; ++k) { a += Z[3 * r + k] * Ji[3 * k + c]; b += Z[3 * r + k] * Jj[3 * k + c]; }
        A[3 * r + c] = a; B[3 * r + c] = b;
    }
}
/* EdgeSE2PointXY (g2o edge_se2_pointxy.h/.cpp; §8-A.3): e = (xp^-1 * l) - z */
void orc_edge_se2_pointxy(const double xp[3], const double l[2], const double z[2],
                          double e[2], double A[6], double B[4]) {
    double inv[3]; orc_se2_inverse(xp, inv);
    double c = cos(inv[2]), s = sin(inv[2]);
    e[0] = (c * l[0] - s * l[1]) + inv[0] - z[0];
    e[1] = (s * l[0] + c * l[1]) + inv[1] - z[1];
    if (!A) return;
    double x1 = xp[0], y1 = xp[1], th1 = xp[2], x2 = l[0], y2 = l[1];
    double a1 = cos(th1), a2 = -a1, a3 = sin(th1);
    A[0] = a2; A[1] = -a3; A[2] = a1 * y2 - a1 * y1 - a3 * x2 + a3 * x1;
    A[3] = a3; A[4] = a2;  A[5] = -a3 * y2 + a3 * y1 - a1 * x2 + a1 * x1;
    B[0] = a1; B[1] = a3; B[2] = -a3; B[3] = a1;
}

/* A5: computeActiveErrors + activeChi2 */
double orc_chi2(const orc_graph *g) {
    double chi = 0;
    for (int k = 0; k < g->npp; ++k) {
        if (g->pfix[g->pp_i[k]] && g->pfix[g->pp_j[k]]) continue;   /* edge between fixed vertices is inactive */
        double e[3]; orc_edge_se2(g->pose + 3 * g->pp_i[k], g->pose + 3 * g->pp_j[k], g->pp_z + 3 * k, e, NULL, NULL);
        const double *W = g->pp_info + 9 * k;
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) chi += e[r] * W[3 * r + c] * e[c];
    }
    for (int k = 0; k < g->npl; ++k) {
        if (g->pfix[g->pl_p[k]] && g->lfix[g->pl_l[k]]) continue;
        double e[2]; orc_edge_se2_pointxy(g->pose + 3 * g->pl_p[k], g->lm + 2 * g->pl_l[k], g->pl_z + 2 * k, e, NULL, NULL);
        const double *W = g->pl_info + 4 * k;
        chi += e[0] * (W[0] * e[0] + W[1] * e[1]) + e[1] * (W[2] * e[0] + W[3] * e[1]);
    }
    return chi;
}

/* ---- per-edge quadratic form, g2o BaseBinaryEdge::constructQuadraticForm, §8-A.4 ---- */
static void quad_pp(const orc_graph *g, int k, double Hii[9], double Hij[9], double Hjj[9], double bi[3], double bj[3]) {
    double e[3], A[9], B[9];
    orc_edge_se2(g->pose + 3 * g->pp_i[k], g->pose + 3 * g->pp_j[k], g->pp_z + 3 * k, e, A, B);
    const double *W = g->pp_info + 9 * k;
    double WA[9], WB[9], We[3];
    for (int r = 0; r < 3; ++r) { We[r] = 0;
        for (int c = 0; c < 3; ++c) { double a = 0, b = 0;
            for (int t = 0; t < 3; ++t) { a += W[3 * r + t] * A[3 * t + c]; b += W[3 * r + t] * B[3 * t + c]; }
            WA[3 * r + c] = a; WB[3 * r + c] = b; We[r] += W[3 * r + c] * e[c]; } }
    for (int r = 0; r < 3; ++r) { double s1 = 0, s2 = 0;
        for (int c = 0; c < 3; ++c) { double a = 0, b = 0, d = 0;
            for (int t = 0; t < 3; ++t) { a += A[3 * t + r] * WA[3 * t + c]; b += A[3 * t + r] * WB[3 * t + c]; d += B[3 * t + r] * WB[3 * t + c]; }
            Hii[3 * r + c] = a; Hij[3 * r + c] = b; Hjj[3 * r + c] = d; }
        for (int t = 0; t < 3; ++t) { s1 += A[3 * t + r] * We[t]; s2 += B[3 * t + r] * We[t]; }
        bi[r] = -s1; bj[r] = -s2; }
}
static void quad_pl(const orc_graph *g, int k, double Hpp[9], double Hpl[6], double Hll[4], double bp[3], double bl[2]) {
    double e[2], A[6], B[4];
    orc_edge_se2_pointxy(g->pose + 3 * g->pl_p[k], g->lm + 2 * g->pl_l[k], g->pl_z + 2 * k, e, A, B);
    const double *W = g->pl_info + 4 * k;
    double WA[6], WB[4], We[2];
    for (int r = 0; r < 2; ++r) {
        for (int c = 0; c < 3; ++c) WA[3 * r + c] = W[2 * r] * A[c] + W[2 * r + 1] * A[3 + c];
        for (int c = 0; c < 2; ++c) WB[2 * r + c] = W[2 * r] * B[c] + W[2 * r + 1] * B[2 + c];
        We[r] = W[2 * r] * e[0] + W[2 * r + 1] * e[1]; }
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) Hpp[3 * r + c] = A[r] * WA[c] + A[3 + r] * WA[3 + c];
        for (int c = 0; c < 2; ++c) Hpl[2 * r + c] = A[r] * WB[c] + A[3 + r] * WB[2 + c];
        bp[r] = -(A[r] * We[0] + A[3 + r] * We[1]); }
    for (int r = 0; r < 2; ++r) {
        for (int c = 0; c < 2; ++c) Hll[2 * r + c] = B[r] * WB[c] + B[2 + r] * WB[2 + c];
        bl[r] = -(B[r] * We[0] + B[2 + r] * We[1]); }
}

/* A6+A7 per block (layout of gs_export_system) */
void orc_linearize_blocks(const orc_graph *g, double *Hpp_diag, double *Hll_diag, double *Hpp_off,
                          double *Hpl, double *b_pose, double *b_lm) {
    memset(Hpp_diag, 0, (size_t)g->np * 9 * sizeof(double));
    memset(Hll_diag, 0, (size_t)g->nl * 4 * sizeof(double));
    memset(b_pose, 0, (size_t)g->np * 3 * sizeof(double));
    memset(b_lm, 0, (size_t)g->nl * 2 * sizeof(double));
    for (int k = 0; k < g->npp; ++k) {
        int i = g->pp_i[k], j = g->pp_j[k];
        double Hii[9], Hij[9], Hjj[9], bi[3], bj[3]; quad_pp(g, k, Hii, Hij, Hjj, bi, bj);
        if (!g->pfix[i]) { for (int t = 0; t < 9; ++t) Hpp_diag[9 * i + t] += Hii[t]; for (int t = 0; t < 3; ++t) b_pose[3 * i + t] += bi[t]; }
        if (!g->pfix[j]) { for (int t = 0; t < 9; ++t) Hpp_diag[9 * j + t] += Hjj[t]; for (int t = 0; t < 3; ++t) b_pose[3 * j + t] += bj[t]; }
        for (int t = 0; t < 9; ++t) Hpp_off[9 * k + t] = (!g->pfix[i] && !g->pfix[j]) ? Hij[t] : 0.0;
    }
    for (int k = 0; k < g->npl; ++k) {
        int p = g->pl_p[k], l = g->pl_l[k];
        double Hpp[9], W[6], Hll[4], bp[3], bl[2]; quad_pl(g, k, Hpp, W, Hll, bp, bl);
        if (!g->pfix[p]) { for (int t = 0; t < 9; ++t) Hpp_diag[9 * p + t] += Hpp[t]; for (int t = 0; t < 3; ++t) b_pose[3 * p + t] += bp[t]; }
        if (!g->lfix[l]) { for (int t = 0; t < 4; ++t) Hll_diag[4 * l + t] += Hll[t]; for (int t = 0; t < 2; ++t) b_lm[2 * l + t] += bl[t]; }
        for (int t = 0; t < 6; ++t) Hpl[6 * k + t] = (!g->pfix[p] && !g->lfix[l]) ? W[t] : 0.0;
    }
}

/* ---------------- A4: structure (g2o BlockSolver::buildStructure + LinearSolverEigen CCS fill) ----
 * index map: active free vertices sorted by id => landmarks (ids 0..) first, then poses (1000..)
 * (§8-A.6); scalar upper-triangular CCS. */
typedef struct { int row, col; } rc_t;
static int rc_cmp(const void *a, const void *b) {
    const rc_t *x = (const rc_t *)a, *y = (const rc_t *)b;
    if (x->col != y->col) return x->col < y->col ? -1 : 1;
    return x->row < y->row ? -1 : (x->row > y->row);
}
static int find_pos(const orc_graph *g, int row, int col) {
    int lo = g->colptr[col], hi = g->colptr[col + 1] - 1;
    while (lo <= hi) { int mid = (lo + hi) >> 1; int r = g->rowind[mid];
        if (r == row) return mid; if (r < row) lo = mid + 1; else hi = mid - 1; }
    return -1;
}
static int build_structure(orc_graph *g) {
    free_system(g);
    g->lm_off = (int *)malloc((size_t)(g->nl + 1) * sizeof(int));
    g->pose_off = (int *)malloc((size_t)(g->np + 1) * sizeof(int));
    int n = 0;
    for (int l = 0; l < g->nl; ++l) { if (g->lfix[l]) g->lm_off[l] = -1; else { g->lm_off[l] = n; n += 2; } }
    for (int p = 0; p < g->np; ++p) { if (g->pfix[p]) g->pose_off[p] = -1; else { g->pose_off[p] = n; n += 3; } }
    g->n = n;
    /* collect pattern entries (upper) */
    size_t cap = (size_t)g->nl * 3 + (size_t)g->np * 6 + (size_t)g->npp * 9 + (size_t)g->npl * 6 + 16;
    rc_t *ent = (rc_t *)malloc(cap * sizeof(rc_t)); size_t ne = 0;
    for (int l = 0; l < g->nl; ++l) if (g->lm_off[l] >= 0) { int o = g->lm_off[l];
        for (int c = 0; c < 2; ++c) for (int r = 0; r <= c; ++r) { ent[ne].row = o + r; ent[ne].col = o + c; ++ne; } }
    for (int p = 0; p < g->np; ++p) if (g->pose_off[p] >= 0) { int o = g->pose_off[p];
        for (int c = 0; c < 3; ++c) for (int r = 0; r <= c; ++r) { ent[ne].row = o + r; ent[ne].col = o + c; ++ne; } }
    for (int k = 0; k < g->npp; ++k) { int oi = g->pose_off[g->pp_i[k]], oj = g->pose_off[g->pp_j[k]];
        if (oi < 0 || oj < 0 || oi == oj) continue;
        int lo = oi < oj ? oi : oj, hi = oi < oj ? oj : oi;
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { ent[ne].row = lo + r; ent[ne].col = hi + c; ++ne; } }
    for (int k = 0; k < g->npl; ++k) { int op = g->pose_off[g->pl_p[k]], ol = g->lm_off[g->pl_l[k]];
        if (op < 0 || ol < 0) continue;                      /* landmarks precede poses: ol < op */
        for (int r = 0; r < 2; ++r) for (int c = 0; c < 3; ++c) { ent[ne].row = ol + r; ent[ne].col = op + c; ++ne; } }
    qsort(ent, ne, sizeof(rc_t), rc_cmp);
    size_t nu = 0;
    for (size_t t = 0; t < ne; ++t) if (t == 0 || ent[t].row != ent[nu - 1].row || ent[t].col != ent[nu - 1].col) ent[nu++] = ent[t];
    g->nnz = (int)nu;
    g->colptr = (int *)calloc((size_t)n + 1, sizeof(int));
    g->rowind = (int *)malloc((nu + 1) * sizeof(int));
    g->values = (double *)calloc(nu + 1, sizeof(double));
    g->b = (double *)calloc((size_t)n + 1, sizeof(double));
    g->delta = (double *)calloc((size_t)n + 1, sizeof(double));
    for (size_t t = 0; t < nu; ++t) { g->colptr[ent[t].col + 1]++; g->rowind[t] = ent[t].row; }
    for (int c = 0; c < n; ++c) g->colptr[c + 1] += g->colptr[c];
    free(ent);
    /* destination positions */
    g->ldiag_pos = (int *)malloc((size_t)(g->nl + 1) * 4 * sizeof(int));
    g->pdiag_pos = (int *)malloc((size_t)(g->np + 1) * 9 * sizeof(int));
    g->pp_pos = (int *)malloc((size_t)(g->npp + 1) * 9 * sizeof(int));
    g->pl_pos = (int *)malloc((size_t)(g->npl + 1) * 6 * sizeof(int));
    for (int l = 0; l < g->nl; ++l) for (int r = 0; r < 2; ++r) for (int c = 0; c < 2; ++c)
        g->ldiag_pos[4 * l + 2 * r + c] = (g->lm_off[l] >= 0 && r <= c) ? find_pos(g, g->lm_off[l] + r, g->lm_off[l] + c) : -1;
    for (int p = 0; p < g->np; ++p) for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c)
        g->pdiag_pos[9 * p + 3 * r + c] = (g->pose_off[p] >= 0 && r <= c) ? find_pos(g, g->pose_off[p] + r, g->pose_off[p] + c) : -1;
    for (int k = 0; k < g->npp; ++k) { int oi = g->pose_off[g->pp_i[k]], oj = g->pose_off[g->pp_j[k]];
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) {
            int pos = -1;                                    /* Hij[r][c] sits at (oi+r, oj+c); mirrored if oi > oj */
            if (oi >= 0 && oj >= 0 && oi != oj) pos = (oi < oj) ? find_pos(g, oi + r, oj + c) : find_pos(g, oj + c, oi + r);
            g->pp_pos[9 * k + 3 * r + c] = pos; } }
    for (int k = 0; k < g->npl; ++k) { int op = g->pose_off[g->pl_p[k]], ol = g->lm_off[g->pl_l[k]];
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 2; ++c)    /* Hpl[r][c] at (op+r, ol+c) -> upper: (ol+c, op+r) */
            g->pl_pos[6 * k + 2 * r + c] = (op >= 0 && ol >= 0) ? find_pos(g, ol + c, op + r) : -1; }
    g->structure_valid = 1; g->ldlt_valid = 0;
    return n;
}

/* A6+A7 into the CCS (g2o BlockSolver::buildSystem) */
int orc_build_system(orc_graph *g) {
    if (!g->structure_valid) build_structure(g);
    memset(g->values, 0, (size_t)g->nnz * sizeof(double));
    memset(g->b, 0, (size_t)g->n * sizeof(double));
    for (int k = 0; k < g->npp; ++k) {
        int i = g->pp_i[k], j = g->pp_j[k]; int oi = g->pose_off[i], oj = g->pose_off[j];
        if (oi < 0 && oj < 0) continue;
        double Hii[9], Hij[9], Hjj[9], bi[3], bj[3]; quad_pp(g, k, Hii, Hij, Hjj, bi, bj);
        if (oi >= 0) { for (int t = 0; t < 9; ++t) { int pos = g->pdiag_pos[9 * i + t]; if (pos >= 0) g->values[pos] += Hii[t]; }
                       for (int t = 0; t < 3; ++t) g->b[oi + t] += bi[t]; }
        if (oj >= 0) { for (int t = 0; t < 9; ++t) { int pos = g->pdiag_pos[9 * j + t]; if (pos >= 0) g->values[pos] += Hjj[t]; }
                       for (int t = 0; t < 3; ++t) g->b[oj + t] += bj[t]; }
        for (int t = 0; t < 9; ++t) { int pos = g->pp_pos[9 * k + t]; if (pos >= 0) g->values[pos] += Hij[t]; }
    }
    for (int k = 0; k < g->npl; ++k) {
        int p = g->pl_p[k], l = g->pl_l[k]; int op = g->pose_off[p], ol = g->lm_off[l];
        if (op < 0 && ol < 0) continue;
        double Hpp[9], W[6], Hll[4], bp[3], bl[2]; quad_pl(g, k, Hpp, W, Hll, bp, bl);
        if (op >= 0) { for (int t = 0; t < 9; ++t) { int pos = g->pdiag_pos[9 * p + t]; if (pos >= 0) g->values[pos] += Hpp[t]; }
                       for (int t = 0; t < 3; ++t) g->b[op + t] += bp[t]; }
        if (ol >= 0) { for (int t = 0; t < 4; ++t) { int pos = g->ldiag_pos[4 * l + t]; if (pos >= 0) g->values[pos] += Hll[t]; }
                       for (int t = 0; t < 2; ++t) g->b[ol + t] += bl[t]; }
        for (int t = 0; t < 6; ++t) { int pos = g->pl_pos[6 * k + t]; if (pos >= 0) g->values[pos] += W[t]; }
    }
    return g->n;
}
int orc_system_n(const orc_graph *g) { return g->n; }
int orc_system_nnz(const orc_graph *g) { return g->nnz; }
const int *orc_system_colptr(const orc_graph *g) { return g->colptr; }
const int *orc_system_rowind(const orc_graph *g) { return g->rowind; }
const double *orc_system_values(const orc_graph *g) { return g->values; }
const double *orc_system_b(const orc_graph *g) { return g->b; }

/* ---------------- A8: up-looking simplicial LDL^T ----------------
 * Restates what Eigen::SimplicialLDLT<SparseMatrix<double>,Upper> executes:
 * symbolic = elimination tree + column counts (reference thirdparty/Eigen/src/SparseCholesky/
 * SimplicialCholesky_impl.h:51-98), numeric = up-looking factorisation (:101-190), solve =
 * P, L, D, L^T, P^T (SimplicialCholesky.h:530-560).  Ordering here is either natural or a
 * track-interleave order (pose, then the landmarks whose last observer it is) instead of AMD;
 * any exact ordering gives the same x to rounding, which tests check against the Eigen build. */
static void ldlt_symbolic(orc_graph *g, int ordering) {
    int n = g->n;
    g->perm = (int *)malloc((size_t)(n + 1) * sizeof(int));   /* perm[new] = old */
    g->iperm = (int *)malloc((size_t)(n + 1) * sizeof(int));
    if (ordering == 0) { for (int i = 0; i < n; ++i) g->perm[i] = i; }
    else if (ordering == 2 && g->user_perm && g->user_perm_n == n) { memcpy(g->perm, g->user_perm, (size_t)n * sizeof(int)); }
    else {
        int *last = (int *)malloc((size_t)(g->nl + 1) * sizeof(int));
        for (int l = 0; l < g->nl; ++l) last[l] = -1;
        for (int k = 0; k < g->npl; ++k) { int p = g->pl_p[k], l = g->pl_l[k];
            if (g->pose_off[p] >= 0 && p > last[l]) last[l] = p; }
        /* bucket landmarks by last free observer */
        int *head = (int *)malloc((size_t)(g->np + 1) * sizeof(int)), *next = (int *)malloc((size_t)(g->nl + 1) * sizeof(int));
        for (int p = 0; p < g->np; ++p) head[p] = -1;
        int pos = 0;
        for (int l = g->nl - 1; l >= 0; --l) { if (g->lm_off[l] < 0) continue;
            if (last[l] < 0) { g->perm[pos++] = g->lm_off[l]; g->perm[pos++] = g->lm_off[l] + 1; }
            else { next[l] = head[last[l]]; head[last[l]] = l; } }
        for (int p = 0; p < g->np; ++p) {
            if (g->pose_off[p] >= 0) { for (int t = 0; t < 3; ++t) g->perm[pos++] = g->pose_off[p] + t; }
            for (int l = head[p]; l >= 0; l = next[l]) { g->perm[pos++] = g->lm_off[l]; g->perm[pos++] = g->lm_off[l] + 1; }
        }
        free(last); free(head); free(next);
    }
    for (int i = 0; i < n; ++i) g->iperm[g->perm[i]] = i;
    /* C = P A P^T, upper */
    int nnz = g->nnz;
    g->Cp = (int *)calloc((size_t)n + 1, sizeof(int)); g->Ci = (int *)malloc((size_t)(nnz + 1) * sizeof(int));
    g->Cx = (double *)malloc((size_t)(nnz + 1) * sizeof(double)); g->cmap = (int *)malloc((size_t)(nnz + 1) * sizeof(int));
    for (int c = 0; c < n; ++c) for (int p = g->colptr[c]; p < g->colptr[c + 1]; ++p) {
        int r2 = g->iperm[g->rowind[p]], c2 = g->iperm[c]; int cc = r2 > c2 ? r2 : c2; g->Cp[cc + 1]++; }
    for (int c = 0; c < n; ++c) g->Cp[c + 1] += g->Cp[c];
    int *fill = (int *)malloc((size_t)(n + 1) * sizeof(int)); memcpy(fill, g->Cp, (size_t)n * sizeof(int));
    for (int c = 0; c < n; ++c) for (int p = g->colptr[c]; p < g->colptr[c + 1]; ++p) {
        int r2 = g->iperm[g->rowind[p]], c2 = g->iperm[c]; int rr = r2 < c2 ? r2 : c2, cc = r2 > c2 ? r2 : c2;
        int q = fill[cc]++; g->Ci[q] = rr; g->cmap[p] = q; }
    free(fill);
    /* etree + counts */
    g->parent = (int *)malloc((size_t)(n + 1) * sizeof(int)); g->Lnz = (int *)calloc((size_t)n + 1, sizeof(int));
    g->flag = (int *)malloc((size_t)(n + 1) * sizeof(int)); g->Lp = (int *)malloc((size_t)(n + 2) * sizeof(int));
    for (int k = 0; k < n; ++k) { g->parent[k] = -1; g->flag[k] = k; g->Lnz[k] = 0;
        for (int p = g->Cp[k]; p < g->Cp[k + 1]; ++p) { int i = g->Ci[p];
            if (i < k) for (; g->flag[i] != k; i = g->parent[i]) { if (g->parent[i] == -1) g->parent[i] = k; g->Lnz[i]++; g->flag[i] = k; } } }
    g->Lp[0] = 0; for (int k = 0; k < n; ++k) g->Lp[k + 1] = g->Lp[k] + g->Lnz[k];
    size_t lnz = (size_t)g->Lp[n];
    g->Li = (int *)malloc((lnz + 1) * sizeof(int)); g->Lx = (double *)malloc((lnz + 1) * sizeof(double));
    g->D = (double *)malloc((size_t)(n + 1) * sizeof(double)); g->y = (double *)calloc((size_t)n + 1, sizeof(double));
    g->pattern = (int *)malloc((size_t)(n + 1) * sizeof(int));
    g->ldlt_ordering = ordering; g->ldlt_valid = 1;
}
static int ldlt_numeric(orc_graph *g) {
    int n = g->n;
    for (int p = 0; p < g->nnz; ++p) g->Cx[g->cmap[p]] = g->values[p];
    for (int k = 0; k < n; ++k) {
        g->y[k] = 0.0; int top = n; g->flag[k] = k; g->Lnz[k] = 0;
        for (int p = g->Cp[k]; p < g->Cp[k + 1]; ++p) { int i = g->Ci[p];
            if (i <= k) { g->y[i] += g->Cx[p]; int len = 0;
                for (; g->flag[i] != k; i = g->parent[i]) { g->pattern[len++] = i; g->flag[i] = k; }
                while (len > 0) g->pattern[--top] = g->pattern[--len]; } }
        double d = g->y[k]; g->y[k] = 0.0;
        for (; top < n; ++top) { int i = g->pattern[top]; double yi = g->y[i]; g->y[i] = 0.0;
            double lki = yi / g->D[i]; int p2 = g->Lp[i] + g->Lnz[i];
            for (int p = g->Lp[i]; p < p2; ++p) g->y[g->Li[p]] -= g->Lx[p] * yi;
            d -= lki * yi; g->Li[p2] = k; g->Lx[p2] = lki; g->Lnz[i]++; }
        g->D[k] = d;
        if (d == 0.0) return -1;
    }
    return 0;
}
/* ordering 2: the elimination order of ANOTHER solver (e.g. the nested-dissection order of the HIP back-end's plan), so that
 * its increment can be compared with this LDL^T's under the SAME order — what is left then is arithmetic, not ordering.
 * perm[new] = old in this oracle's scalar numbering (orc_vertex_offsets); checked to be a permutation. */
int orc_set_elimination_order(orc_graph *g, const int *perm, int n) {
    if (!g || !perm || n <= 0) return -1;
    char *seen = (char *)calloc((size_t)n, 1);
    for (int i = 0; i < n; ++i) { if (perm[i] < 0 || perm[i] >= n || seen[perm[i]]) { free(seen); return -2; } seen[perm[i]] = 1; }
    free(seen);
    free(g->user_perm); g->user_perm = (int *)malloc((size_t)n * sizeof(int)); memcpy(g->user_perm, perm, (size_t)n * sizeof(int));
    g->user_perm_n = n; g->ldlt_valid = 0;
    return 0;
}
/* first scalar of every vertex in the system's numbering (free landmarks first, then free poses: g2o's hessian index order,
 * SURVEY 8-A.6), -1 for fixed vertices; valid after orc_build_system */
int orc_vertex_offsets(const orc_graph *g, int *pose_off, int *lm_off) {
    if (!g || !g->structure_valid) return -1;
    memcpy(pose_off, g->pose_off, (size_t)g->np * sizeof(int)); memcpy(lm_off, g->lm_off, (size_t)g->nl * sizeof(int));
    return g->n;
}
int orc_solve_ldlt(orc_graph *g, int ordering, double *x) {
    if (!g->structure_valid) return -1;
    if (!g->ldlt_valid || g->ldlt_ordering != ordering) {
        free(g->perm); free(g->iperm); free(g->Cp); free(g->Ci); free(g->Cx); free(g->cmap);
        free(g->parent); free(g->Lp); free(g->Li); free(g->Lnz); free(g->flag); free(g->pattern);
        free(g->Lx); free(g->D); free(g->y);
        ldlt_symbolic(g, ordering);
    }
    if (ldlt_numeric(g) != 0) return -2;
    int n = g->n; double *w = (double *)malloc((size_t)(n + 1) * sizeof(double));
    for (int i = 0; i < n; ++i) w[i] = g->b[g->perm[i]];
    for (int j = 0; j < n; ++j) { double wj = w[j]; for (int p = g->Lp[j]; p < g->Lp[j] + g->Lnz[j]; ++p) w[g->Li[p]] -= g->Lx[p] * wj; }
    for (int j = 0; j < n; ++j) w[j] /= g->D[j];
    for (int j = n - 1; j >= 0; --j) { double s = w[j]; for (int p = g->Lp[j]; p < g->Lp[j] + g->Lnz[j]; ++p) s -= g->Lx[p] * w[g->Li[p]]; w[j] = s; }
    for (int i = 0; i < n; ++i) x[g->perm[i]] = w[i];
    free(w);
    return 0;
}

/* A9: g2o VertexSE2::oplusImpl / VertexPointXY::oplusImpl, §8-A.5 */
void orc_apply_update(orc_graph *g, const double *x) {
    if (x != g->delta) memcpy(g->delta, x, (size_t)g->n * sizeof(double));
    for (int l = 0; l < g->nl; ++l) { int o = g->lm_off[l]; if (o < 0) continue;
        g->lm[2 * l] += x[o]; g->lm[2 * l + 1] += x[o + 1]; }
    for (int p = 0; p < g->np; ++p) { int o = g->pose_off[p]; if (o < 0) continue;
        g->pose[3 * p] += x[o]; g->pose[3 * p + 1] += x[o + 1];
        g->pose[3 * p + 2] = orc_normalize_theta(g->pose[3 * p + 2] + x[o + 2]); }
}
void orc_get_delta(const orc_graph *g, double *dpose, double *dlm) {
    for (int p = 0; p < g->np; ++p) { int o = g->pose_off ? g->pose_off[p] : -1;
        for (int t = 0; t < 3; ++t) dpose[3 * p + t] = (o >= 0) ? g->delta[o + t] : 0.0; }
    for (int l = 0; l < g->nl; ++l) { int o = g->lm_off ? g->lm_off[l] : -1;
        for (int t = 0; t < 2; ++t) dlm[2 * l + t] = (o >= 0) ? g->delta[o + t] : 0.0; }
}

/* A10: g2o SparseOptimizer::optimize + OptimizationAlgorithmGaussNewton::solve, §8-A.7 */
int orc_optimize_until(orc_graph *g, int iterations, double rel_tol, int ordering, orc_solver_fn solver, void *ctx,
                       double *chi2_out, double *tm, int *failed) {
    double t_all = now_ms();
    int done = 0, analyze;
    if (failed) *failed = 0;
    if (!g->structure_valid) build_structure(g);
    if (g->n == 0) return 0;
    analyze = 1;                                   /* initializeOptimization precedes every optimize() */
    double *x = (double *)calloc((size_t)g->n + 1, sizeof(double));
    double prev = -1.0;
    for (int it = 0; it < iterations; ++it) {
        double t0 = now_ms();
        const double chi = orc_chi2(g);
        if (chi2_out) chi2_out[it] = chi;
        orc_build_system(g);
        double t1 = now_ms();
        int rc;
        if (solver) rc = solver(ctx, analyze, g->n, g->colptr, g->rowind, g->values, g->b, x);
        else { if (analyze) g->ldlt_valid = 0; rc = orc_solve_ldlt(g, ordering, x); }
        analyze = 0;
        double t2 = now_ms();
        if (rc != 0) { if (failed) *failed = 1; break; }   /* g2o: Fail before update(); the previous iterate stays */
        orc_apply_update(g, x);
        double t3 = now_ms();
        if (tm) { tm[0] += t1 - t0; tm[2] += t2 - t1; tm[3] += t3 - t2; }
        ++done;
        if (rel_tol >= 0.0 && prev >= 0.0 && fabs(prev - chi) <= rel_tol * chi) break;
        prev = chi;
    }
    free(x);
    if (tm) tm[4] += now_ms() - t_all;
    return done;
}
int orc_optimize(orc_graph *g, int iterations, int ordering, orc_solver_fn solver, void *ctx,
                 double *chi2_out, double *tm) {
    return orc_optimize_until(g, iterations, -1.0, ordering, solver, ctx, chi2_out, tm, NULL);
}
