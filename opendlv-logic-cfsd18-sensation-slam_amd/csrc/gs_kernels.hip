// gs_kernels.hip — hand-written HIP kernels (gfx950 / CDNA4, wave64) of the GraphSLAM hot path.
//
//   A0  k_polar_to_xy, k_cone_to_global    Slam::transformConeToCoG / Spherical2Cartesian / coneToGlobal
//                                          (reference src/slam.cpp:513-523, 637-654, 499-510)
//   A1  k_associate                        association loop of Slam::addConesToMap vs a fixed map
//                                          (reference src/slam.cpp:570-607, 708-711)
//   A5-A7 k_linearize_ell (+ _finalize)    g2o computeError + linearizeOplus + constructQuadraticForm of
//                                          EdgeSE2 / EdgeSE2PointXY, summed per vertex (SURVEY.md §8-A.2-4;
//                                          driven from reference src/slam.cpp:481); k_linearize_*_gather =
//                                          general-graph fallback
//   A8  k_factor_level / k_backsolve_level  multifrontal Cholesky of the joint system (replaces Eigen
//                                          SimplicialLDLT, reference thirdparty/Eigen/src/SparseCholesky/
//                                          SimplicialCholesky_impl.h:101-190), forward solve fused as an extra row
//   A9  k_update                           VertexSE2::oplusImpl / VertexPointXY::oplusImpl (§8-A.5)
//
// All fp64; every sum has a fixed order (no floating-point atomics) so results are bitwise reproducible.
#include "gs_device.hpp"
#include <hip/hip_ext.h>
#include <algorithm>
#include <mutex>
#include <set>
#include <type_traits>
#include <utility>

namespace gs {

static constexpr int WAVE = 64;
static constexpr int LM_REC = 8;       // doubles per (wave tile, landmark) partial-sum record: {H00, H01, H11, b0, b1, -, -, -} = one 64-byte line (gs_device.hpp: lm_part)
// GS_G2O_ORDER (default 1): the edge residuals in g2o's operation order (inverse, then compose) without fused
// multiply-adds; 0 = differences first (fewer rounding errors, but not the reference's numbers) — A/B builds only
#ifndef GS_G2O_ORDER
#define GS_G2O_ORDER 1
#endif

// constants exactly as the reference declares them (src/slam.hpp:134-136; PI is a FLOAT literal)
__device__ static constexpr double kDeg2Rad = 0.017453292522222;
__device__ static constexpr double kRad2Deg = 57.295779513082325;
__device__ static constexpr double kPiF = (double)3.14159265f;
__device__ static constexpr double kPi = 3.14159265358979323846;

// g2o normalize_theta
__device__ __forceinline__ double normalize_theta(double th) {
#pragma clang fp contract(off)                                      // m * 2 pi is rounded before the subtraction, as on the CPU path
    if (th >= -kPi && th < kPi) return th;
    double m = floor(th / (2.0 * kPi));
    th = th - m * 2.0 * kPi;
    if (th >= kPi) th -= 2.0 * kPi;
    if (th < -kPi) th += 2.0 * kPi;
    return th;
}

// ------------------------------------------------------------------ A0
__device__ __forceinline__ void polar_to_xy(double az, double zen, double dist, double lidar, double &x, double &y) {
    // transformConeToCoG: sign is NaN at az == 0 (kept)
    double sign = az / fabs(az);
    double ang = kPiF - fabs(az * kDeg2Rad);
    double sa, ca; sincos(ang, &sa, &ca);                           // (one range reduction for the pair: the batched association is bound by these calls)
    double dnew = sqrt(lidar * lidar + dist * dist - 2.0 * lidar * dist * ca);
    double anew = asin((sa * dist) / dnew) * kRad2Deg;
    double az2 = anew * sign;
    // Spherical2Cartesian
    double cz = cos(zen * kDeg2Rad);
    double s2, c2; sincos(az2 * kDeg2Rad, &s2, &c2);
    x = dnew * cz * c2;
    y = dnew * cz * s2;
}

__global__ void k_polar_to_xy(int n, const double *__restrict__ az, const double *__restrict__ zen,
                              const double *__restrict__ dist, double lidar, double *__restrict__ out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double x, y; polar_to_xy(az[i], zen[i], dist[i], lidar, x, y);
    out[2 * i] = x; out[2 * i + 1] = y;
}

__device__ __forceinline__ void cone_to_global_cs(double px, double py, double c, double s, const double *obs, double lidar, double &gx, double &gy) {
    double x, y; polar_to_xy(obs[0], obs[1], obs[2], lidar, x, y);
    gx = (x * c - y * s) + px;
    gy = (x * s + y * c) + py;
}
__device__ __forceinline__ void cone_to_global(const double *pose, const double *obs, double lidar, double &gx, double &gy) {
    double s, c; sincos(pose[2], &s, &c);
    cone_to_global_cs(pose[0], pose[1], c, s, obs, lidar, gx, gy);
}

__global__ void k_cone_to_global(int n, const double *__restrict__ poses, const int32_t *__restrict__ pose_of_obs,
                                 const double *__restrict__ obs, double lidar, double *__restrict__ out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double gx, gy; cone_to_global(poses + 3 * pose_of_obs[i], obs + 4 * i, lidar, gx, gy);
    out[2 * i] = gx; out[2 * i + 1] = gy;
}

// ------------------------------------------------------------------ A1
// One thread per observation; the map is streamed through LDS in tiles and every thread keeps the LOWEST
// matching index, which equals the reference's first-match-in-insertion-order scan.
static constexpr int ASSOC_TILE = 1024;
__global__ void __launch_bounds__(256) k_associate(int n, const double *__restrict__ poses,
        const int32_t *__restrict__ pose_of_obs, const double *__restrict__ obs, double lidar, int n_map,
        const double *__restrict__ map_xy, const int32_t *__restrict__ map_type, double thr, double type_tol,
        int32_t *__restrict__ out) {
    __shared__ double sx[ASSOC_TILE], sy[ASSOC_TILE];
    __shared__ int32_t st[ASSOC_TILE];
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    double gx = 0, gy = 0, ty = 0; bool live = i < n;
    if (live) { cone_to_global(poses + 3 * pose_of_obs[i], obs + 4 * i, lidar, gx, gy); ty = obs[4 * i + 3]; }
    int found = -1;
    for (int base = 0; base < n_map; base += ASSOC_TILE) {
        int cnt = min(ASSOC_TILE, n_map - base);
        __syncthreads();
        for (int t = threadIdx.x; t < cnt; t += blockDim.x) { sx[t] = map_xy[2 * (base + t)]; sy[t] = map_xy[2 * (base + t) + 1]; st[t] = map_type[base + t]; }
        __syncthreads();
        if (live && found < 0) {
            for (int t = 0; t < cnt; ++t) {
                if (fabs((double)st[t] - ty) < type_tol) {
                    double dx = sx[t] - gx, dy = sy[t] - gy;
                    if (sqrt(dx * dx + dy * dy) < thr) { found = base + t; break; }
                }
            }
        }
    }
    if (live) out[i] = found;
}

__global__ void k_pose_trig(int n, const double *__restrict__ pose_est, double *__restrict__ pose_cs);
// ---- batched A1 with a HASHED uniform grid built ON THE DEVICE (round 4).  Cell edge = a hair above the threshold, so every cone within
// the threshold of a query sits in the 3 x 3 cells around it; a cell (cx, cy) — unbounded integers, no bounding box, no coarsening for
// sparse maps (a 25 km lap in a dense grid had 20 m cells of ~8 cones each: ~24 candidates per query) — hashes to one of B = 2^k >= 4 n_map
// buckets.  Build: cones per bucket (k_grid_count, integer atomics: the counts do not depend on the order), exclusive scan (k_grid_scan),
// fill (k_grid_fill: the order inside a bucket is whatever the atomics give — the query takes the LOWEST matching index over its nine
// buckets, which does not depend on it; a colliding cone of some far cell is one more candidate that fails the distance test).  Same
// pair test, same arithmetic as k_associate => identical indices.  No host pass over the map, no host round trip.
struct GridParams { double inv_cell; uint32_t mask; int pad; };
__device__ __forceinline__ long long grid_coord(double v, double inv_cell) { return (long long)fmin(fmax(floor(v * inv_cell), -4.0e15), 4.0e15); }   // (finite input; clamped: no overflow of the cast)
__device__ __forceinline__ uint32_t grid_bucket(long long cx, long long cy, uint32_t mask) {
    return (((uint32_t)cx * 73856093u) ^ ((uint32_t)cy * 19349663u)) & mask;
}
__global__ void __launch_bounds__(256) k_grid_count(int n_map, const double *__restrict__ map_xy, GridParams g, int32_t *__restrict__ count) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n_map) return;
    const double x = map_xy[2 * j], y = map_xy[2 * j + 1];
    if (!(isfinite(x) && isfinite(y))) return;                     // never within any threshold of anything: not in the grid
    atomicAdd(count + grid_bucket(grid_coord(x, g.inv_cell), grid_coord(y, g.inv_cell), g.mask), 1);
}
// exclusive scan of count[0 .. n) into start[0 .. n], one workgroup
__global__ void __launch_bounds__(1024) k_grid_scan(int n, const int32_t *__restrict__ count, int32_t *__restrict__ start) {
    __shared__ int wsum[16]; __shared__ int carry_s;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 4096) {                     // four consecutive buckets per thread
        const int i0 = base + 4 * tid;
        int v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = i0 + k < n ? count[i0 + k] : 0;
        const int tsum = v[0] + v[1] + v[2] + v[3];
        int x = tsum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int y = __shfl_up(x, off, WAVE); if (lane >= off) x += y; }
        if (lane == 63) wsum[w] = x;
        __syncthreads();
        int pre = carry_s;
        for (int k = 0; k < w; ++k) pre += wsum[k];
        int run = pre + x - tsum;
#pragma unroll
        for (int k = 0; k < 4; ++k) { if (i0 + k < n) start[i0 + k] = run; run += v[k]; }
        __syncthreads();
        if (tid == 1023) carry_s = pre + x;
        __syncthreads();
    }
    if (tid == 0) start[n] = carry_s;
}
__global__ void __launch_bounds__(256) k_grid_fill(int n_map, const double *__restrict__ map_xy, GridParams g, const int32_t *__restrict__ start,
                                                   int32_t *__restrict__ cursor, int32_t *__restrict__ items) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n_map) return;
    const double x = map_xy[2 * j], y = map_xy[2 * j + 1];
    if (!(isfinite(x) && isfinite(y))) return;
    const uint32_t c = grid_bucket(grid_coord(x, g.inv_cell), grid_coord(y, g.inv_cell), g.mask);
    items[start[c] + atomicAdd(cursor + c, 1)] = j;
}
__global__ void __launch_bounds__(256) k_associate_grid_dev(int n, const double *__restrict__ poses, const int32_t *__restrict__ pose_of_obs,
        const double *__restrict__ obs, double lidar, const double *__restrict__ map_xy, const int32_t *__restrict__ map_type, double thr, double type_tol,
        GridParams g, const int32_t *__restrict__ cell_start, const int32_t *__restrict__ cell_items, int32_t *__restrict__ out,
        const double *__restrict__ pose_cs /* cos, sin of every pose's heading (k_pose_trig): the observations of a pose share them */) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double4 o4 = reinterpret_cast<const double4 *>(obs)[i];      // the observation's 32 bytes in one load (az, zen, dist, type)
    const double ob[4] = {o4.x, o4.y, o4.z, o4.w};
    const int pi = pose_of_obs[i]; const double2 cs = reinterpret_cast<const double2 *>(pose_cs)[pi];
    double gx, gy; cone_to_global_cs(poses[3 * pi], poses[3 * pi + 1], cs.x, cs.y, ob, lidar, gx, gy);
    const double ty = ob[3];
    int found = 0x7fffffff;
    if (isfinite(gx) && isfinite(gy)) {                              // (NaN: azimuth 0, SURVEY 8-B.3 — no match)
        const long long cx = grid_coord(gx, g.inv_cell), cy = grid_coord(gy, g.inv_cell);
        int s0[9], s1[9];                                            // the nine buckets' ranges first: independent loads
#pragma unroll
        for (int k = 0; k < 9; ++k) { const uint32_t bk = grid_bucket(cx + (k % 3) - 1, cy + (k / 3) - 1, g.mask); s0[k] = cell_start[bk]; s1[k] = cell_start[bk + 1]; }
#pragma unroll
        for (int k = 0; k < 9; ++k)
            for (int q = s0[k]; q < s1[k]; ++q) { const int j = cell_items[q];
                if (j < found && fabs((double)map_type[j] - ty) < type_tol) {
                    const double ddx = map_xy[2 * j] - gx, ddy = map_xy[2 * j + 1] - gy;
                    if (sqrt(ddx * ddx + ddy * ddy) < thr) found = j; } }
    }
    out[i] = found == 0x7fffffff ? -1 : found;
}
static GridParams grid_params_of(double thr, long long buckets) { GridParams g; g.inv_cell = 1.0 / (thr * (1.0 + 1e-9)); g.mask = (uint32_t)(buckets - 1); g.pad = 0; return g; }
// buckets: a power of two; count / start / cursor: buckets + 1 ints each, items: n_map ints
void launch_grid_build(int n_map, const double *map_xy, double thr, long long buckets, int32_t *count, int32_t *start, int32_t *cursor, int32_t *items, hipStream_t st) {
    const GridParams g = grid_params_of(thr, buckets);
    hipMemsetAsync(count, 0, (size_t)(buckets + 1) * sizeof(int32_t), st); hipMemsetAsync(cursor, 0, (size_t)(buckets + 1) * sizeof(int32_t), st);
    if (n_map > 0) hipLaunchKernelGGL(k_grid_count, dim3((n_map + 255) / 256), dim3(256), 0, st, n_map, map_xy, g, count);
    hipLaunchKernelGGL(k_grid_scan, dim3(1), dim3(1024), 0, st, (int)buckets, count, start);
    if (n_map > 0) hipLaunchKernelGGL(k_grid_fill, dim3((n_map + 255) / 256), dim3(256), 0, st, n_map, map_xy, g, start, cursor, items);
}
void launch_associate_grid_dev(int n, const double *poses, const int32_t *pose_of_obs, const double *obs, double lidar, const double *map_xy,
                               const int32_t *map_type, double thr, double type_tol, long long buckets, const int32_t *cell_start, const int32_t *cell_items,
                               int32_t *out, int n_poses, double *pose_cs_scratch, hipStream_t st, hipEvent_t start, hipEvent_t stop) {
    if (n <= 0) return;
    // cos / sin of every pose once (k_pose_trig), then the queries: start is attached to the first dispatch, stop to the second
    hipExtLaunchKernelGGL(k_pose_trig, dim3((std::max(n_poses, 1) + 255) / 256), dim3(256), 0, st, start, nullptr, 0, n_poses, poses, pose_cs_scratch);
    hipExtLaunchKernelGGL(k_associate_grid_dev, dim3((n + 255) / 256), dim3(256), 0, st, nullptr, stop, 0, n, poses, pose_of_obs, obs, lidar, map_xy, map_type,
                          thr, type_tol, grid_params_of(thr, buckets), cell_start, cell_items, out, (const double *)pose_cs_scratch);
}


// One keyframe's front end in ONE launch (A0 + A1 fused): workgroup i takes observation i of the frame — polar -> CoG-frame
// XY (the edge measurement), -> global XY (the association query) — and its 256 threads scan the resident map for the
// LOWEST index j with a matching type and distance < thr (= the reference's first match in insertion order, reference
// src/slam.cpp:570-607; signed_type = the localizer's test without fabs, :360).  in = {pose x, y, theta, obs[4 * k]}.
__global__ void __launch_bounds__(256) k_frame_frontend(int k, const double *__restrict__ in, double lidar, int n_map,
        const double *__restrict__ map_xy, const int32_t *__restrict__ map_type, double thr, double type_tol, int signed_type,
        double *__restrict__ out_z, double *__restrict__ out_g, int32_t *__restrict__ out_idx) {
    __shared__ int red[4];
    const int i = blockIdx.x, tid = threadIdx.x;
    const double *obs = in + 3 + 4 * i;
    double zx, zy; polar_to_xy(obs[0], obs[1], obs[2], lidar, zx, zy);
    const double c = cos(in[2]), s = sin(in[2]);
    const double gx = (zx * c - zy * s) + in[0], gy = (zx * s + zy * c) + in[1];
    const double ty = obs[3]; const int tyi = (int)ty;
    int found = 0x7fffffff;
    for (int j = tid; j < n_map; j += 256) {
        const bool type_ok = signed_type ? ((double)(map_type[j] - tyi) < type_tol) : (fabs((double)map_type[j] - ty) < type_tol);
        if (type_ok) { const double dx = map_xy[2 * j] - gx, dy = map_xy[2 * j + 1] - gy;
            if (sqrt(dx * dx + dy * dy) < thr) { found = j; break; } }     // ascending j per thread: its first hit is its lowest
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) found = min(found, __shfl_down(found, off, WAVE));
    if ((tid & 63) == 0) red[tid >> 6] = found;
    __syncthreads();
    if (tid == 0) { found = min(min(red[0], red[1]), min(red[2], red[3]));
        out_idx[i] = found == 0x7fffffff ? -1 : found;
        out_z[2 * i] = zx; out_z[2 * i + 1] = zy; out_g[2 * i] = gx; out_g[2 * i + 1] = gy; }
}
void launch_frame_frontend(int k, const double *in, double lidar, int n_map, const double *map_xy, const int32_t *map_type,
                           double thr, double type_tol, int signed_type, double *out_z, double *out_g, int32_t *out_idx, hipStream_t st) {
    if (k > 0) hipLaunchKernelGGL(k_frame_frontend, dim3(k), dim3(256), 0, st, k, in, lidar, n_map, map_xy, map_type, thr, type_tol, signed_type, out_z, out_g, out_idx);
}

// ------------------------------------------------------------------ A5-A7
// EdgeSE2PointXY with the pose's cos/sin already known: error, Jacobian rows A0/A1 (2x3); B = R(theta)^T
__device__ __forceinline__ void edge_pl(double px, double py, double c, double s, double lx, double ly, double zx, double zy,
                                        double &ex, double &ey, double A0[3], double A1[3]) {
#if !GS_G2O_ORDER
    double dx = lx - px, dy = ly - py;
#endif
#if GS_G2O_ORDER
    {   // g2o EdgeSE2PointXY::computeError: (x_p^-1 * l) - z with SE2::inverse() = (R^T (-t), -theta) and SE2 * point =
        // R p + t, in g2o's operation order and without fused multiply-adds: at kilometre coordinates the two products
        // cancel to the metre-sized residual and HOW they are rounded is 1e-12 m of it — the CPU path's rounding is the bar
#pragma clang fp contract(off)
        const double ix = -(c * px + s * py), iy = s * px - c * py;
        ex = ((c * lx + s * ly) + ix) - zx;
        ey = ((c * ly - s * lx) + iy) - zy;
    }
#else
    ex = (c * dx + s * dy) - zx;
    ey = (-s * dx + c * dy) - zy;
#endif
    A0[0] = -c; A0[1] = -s; A1[0] = s;  A1[1] = -c;
#if GS_G2O_ORDER
    {   // g2o EdgeSE2PointXY::linearizeOplus writes the lever-arm entries as  a1 y2 - a1 y1 - a3 x2 + a3 x1  and
        // -a3 y2 + a3 y1 - a1 x2 + a1 x1  (a1 = cos, a3 = sin; left to right, every product rounded): at kilometre coordinates that
        // is 1e-13 of relative rounding in H_pp and H_pl which "differences first" does not have — and cond(H) ~ 1e13 on a 25 km lap
        // turns exactly that 1e-13 into the per-cent of the first increment by which the GPU path stood apart from every CPU path
        // (measured: profiles/r03_parity_spread_cfg4*.json).  The reference's numbers are the bar: same order, no contraction.
#pragma clang fp contract(off)
        A0[2] = ((c * ly - c * py) - s * lx) + s * px;
        A1[2] = (((-s) * ly + s * py) - c * lx) + c * px;
    }
#else
    A0[2] = c * dy - s * dx;
    A1[2] = -s * dy - c * dx;
#endif
}

// One observation edge: everything constructQuadraticForm produces, packed symmetric.
//   Hp[6] = A^T W A (xx xy xt yy yt tt), bp[3] = -A^T W e, W6 = A^T W B (3x2 row-major),
//   Hl[3] = B^T W B (00 01 11), bl[2] = -B^T W e, chi = e^T W e
struct PlQuad { double Hp[6], bp[3], W6[6], Hl[3], bl[2], chi; };
__device__ __forceinline__ void quad_pl(double px, double py, double c, double s, double lx, double ly, double zx, double zy,
                                        double w00, double w01, double w11, PlQuad &q) {
    double ex, ey, A0[3], A1[3];
    edge_pl(px, py, c, s, lx, ly, zx, zy, ex, ey, A0, A1);
    double We0 = w00 * ex + w01 * ey, We1 = w01 * ex + w11 * ey;
    q.chi = ex * We0 + ey * We1;
    double WA0[3], WA1[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { WA0[k] = w00 * A0[k] + w01 * A1[k]; WA1[k] = w01 * A0[k] + w11 * A1[k]; }
    q.Hp[0] = A0[0] * WA0[0] + A1[0] * WA1[0]; q.Hp[1] = A0[0] * WA0[1] + A1[0] * WA1[1]; q.Hp[2] = A0[0] * WA0[2] + A1[0] * WA1[2];
    q.Hp[3] = A0[1] * WA0[1] + A1[1] * WA1[1]; q.Hp[4] = A0[1] * WA0[2] + A1[1] * WA1[2]; q.Hp[5] = A0[2] * WA0[2] + A1[2] * WA1[2];
#pragma unroll
    for (int r = 0; r < 3; ++r) q.bp[r] = -(A0[r] * We0 + A1[r] * We1);
    double WB0[2] = {w00 * c - w01 * s, w00 * s + w01 * c};            // W * B, B rows (c, s), (-s, c)
    double WB1[2] = {w01 * c - w11 * s, w01 * s + w11 * c};
#pragma unroll
    for (int r = 0; r < 3; ++r) { q.W6[2 * r] = A0[r] * WB0[0] + A1[r] * WB1[0]; q.W6[2 * r + 1] = A0[r] * WB0[1] + A1[r] * WB1[1]; }
    q.Hl[0] = c * WB0[0] - s * WB1[0]; q.Hl[1] = c * WB0[1] - s * WB1[1]; q.Hl[2] = s * WB0[1] + c * WB1[1];
    q.bl[0] = -(c * We0 - s * We1); q.bl[1] = -(s * We0 + c * We1);
}

// One (pose, odometry edge) incidence with its operands already in registers: adds the edge's share for that
// endpoint to H (6 packed) and b, writes the off-diagonal block if this endpoint owns the edge (role 0 = i
// endpoint); returns the chi2 of an owned edge.
// device-scope store (sc1): see st_off_wt below
template <class Tp> __device__ __forceinline__ void st_wt(Tp *ptr, Tp v) { __hip_atomic_store(ptr, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#ifndef LIN_NT
#define LIN_NT 0
#endif
#if LIN_NT & 4
#define ST_O(ptr, v) st_wt(ptr, v)
#else
#define ST_O(ptr, v) (*(ptr) = (v))
#endif
template <bool WRITE_H>
__device__ __forceinline__ double pp_incidence(const DevGraph &d, int k, int role, const double xi[3], const double xj[3],
                                               double ci, double si, const double zinv5[5], const double w[6],
                                               bool fi, bool fj, double H[6], double b[3], double *hoff = nullptr, int64_t hoff_stride = 0) {
    // (hoff: where the off-diagonal block of an owned edge goes, plane stride hoff_stride; default = the edge's slot of Hpp_off)
    // Structure of the EdgeSE2 Jacobians (g2o EdgeSE2::linearizeOplus): with M = rot(z^-1) * R_i^T = [[m0, m1], [-m1, m0]],
    //   B = [[m0, m1, 0], [-m1, m0, 0], [0, 0, 1]],   A = [-B(:,0), -B(:,1), a2],  a2 = (a02, a12, -1),
    // so with G = B^T W B and wa2 = W a2 everything needed is G (6), wa2 (3) and three dot products:
    //   A^T W A = [[g00, g01, -t0], [., g11, -t1], [., ., t2]],  A^T W B = [[-g00, -g01, -g02], [-g01, -g11, -g12], [t0, t1, wa2_2]],
    //   t0 = b0.wa2, t1 = b1.wa2, t2 = a2.wa2.   (Same quantities as the dense products, ~1/3 of the live values.)
    double chi = 0.0;
    const double dx = xj[0] - xi[0], dy = xj[1] - xi[1];
    const double rx = ci * dx + si * dy, ry = -si * dx + ci * dy;     // rel = xi^-1 * xj
    const double rth = normalize_theta(normalize_theta(-xi[2]) + xj[2]);
    const double cz = zinv5[3], sz = zinv5[4];
#if GS_G2O_ORDER
    double e0, e1;
    {   // g2o EdgeSE2::computeError: z^-1 * (x_i^-1 * x_j) as inverse-then-compose (SE2::inverse = (R^T (-t), -theta), SE2 * SE2 =
        // (R_a t_b + t_a, theta_a + theta_b)), every product rounded on its own (no fused multiply-add): at the reference's
        // initial point these residuals are zero in exact arithmetic and what is computed IS the rounding — follow the CPU path's
#pragma clang fp contract(off)
        const double ix = -(ci * xi[0] + si * xi[1]), iy = si * xi[0] - ci * xi[1];
        const double qx = ix + (ci * xj[0] + si * xj[1]), qy = iy + (ci * xj[1] - si * xj[0]);
        e0 = zinv5[0] + (cz * qx - sz * qy); e1 = zinv5[1] + (sz * qx + cz * qy);
    }
    const double e2 = normalize_theta(zinv5[2] + rth);
#else
    const double e0 = zinv5[0] + (cz * rx - sz * ry), e1 = zinv5[1] + (sz * rx + cz * ry), e2 = normalize_theta(zinv5[2] + rth);
#endif
    const double w00 = w[0], w01 = w[1], w02 = w[2], w11 = w[3], w12 = w[4], w22 = w[5];
    const double We0 = w00 * e0 + w01 * e1 + w02 * e2, We1 = w01 * e0 + w11 * e1 + w12 * e2, We2 = w02 * e0 + w12 * e1 + w22 * e2;
    if (role == 0 && !(fi && fj)) chi = e0 * We0 + e1 * We1 + e2 * We2;
    if (WRITE_H) {
        const double m0 = cz * ci + sz * si, m1 = cz * si - sz * ci;
        const double bw0 = m0 * We0 - m1 * We1, bw1 = m1 * We0 + m0 * We1;      // b0.We, b1.We
        // G = B^T W B
        const double wb00 = w00 * m0 - w01 * m1, wb01 = w00 * m1 + w01 * m0;      // (W B)[0][0..1]
        const double wb10 = w01 * m0 - w11 * m1, wb11 = w01 * m1 + w11 * m0;      // (W B)[1][0..1]
        const double g00 = m0 * wb00 - m1 * wb10, g01 = m0 * wb01 - m1 * wb11, g11 = m1 * wb01 + m0 * wb11;
        const double g02 = m0 * w02 - m1 * w12, g12 = m1 * w02 + m0 * w12;
        if (role == 0) {
            const double a02 = cz * ry + sz * rx, a12 = sz * ry - cz * rx;
            const double wa0 = w00 * a02 + w01 * a12 - w02, wa1 = w01 * a02 + w11 * a12 - w12, wa2 = w02 * a02 + w12 * a12 - w22;
            const double t0 = m0 * wa0 - m1 * wa1, t1 = m1 * wa0 + m0 * wa1, t2 = a02 * wa0 + a12 * wa1 - wa2;
            H[0] += g00; H[1] += g01; H[2] -= t0; H[3] += g11; H[4] -= t1; H[5] += t2;
            b[0] += bw0; b[1] += bw1; b[2] -= a02 * We0 + a12 * We1 - We2;
            const bool both = !fi && !fj;
            const int64_t E = hoff ? hoff_stride : (int64_t)d.Epp;
            double *o = hoff ? hoff : d.Hpp_off + k;
            ST_O(o, both ? -g00 : 0.0);         ST_O(o + E, both ? -g01 : 0.0);     ST_O(o + 2 * E, both ? -g02 : 0.0);
            ST_O(o + 3 * E, both ? -g01 : 0.0); ST_O(o + 4 * E, both ? -g11 : 0.0); ST_O(o + 5 * E, both ? -g12 : 0.0);
            ST_O(o + 6 * E, both ? t0 : 0.0);   ST_O(o + 7 * E, both ? t1 : 0.0);   ST_O(o + 8 * E, both ? wa2 : 0.0);
        } else {
            H[0] += g00; H[1] += g01; H[2] += g02; H[3] += g11; H[4] += g12; H[5] += w22;
            b[0] -= bw0; b[1] -= bw1; b[2] -= We2;
        }
    }
    return chi;
}
// the same with the operands fetched here.  An incidence record is 8 bytes: {edge (-1: another shard evaluates it), other endpoint | role << 31}
// (role 0: the pose that holds the record is the edge's i endpoint, 1: its j endpoint) — the pose itself is known to whoever walks its
// records (16-byte {edge, role, i, j} records until round 3: 32 of the pass's ~150 excess bytes per pose)
template <bool WRITE_H>
__device__ __forceinline__ double pp_incidence_rec(const DevGraph &d, const int2 inc, int p, double H[6], double b[3], bool have_cs = false, double own_c = 1.0, double own_s = 0.0) {
    const int k = inc.x, role = (int)((uint32_t)inc.y >> 31), other = inc.y & 0x7fffffff;
    if (k < 0) return 0.0;                                                       // evaluated by another shard
    const int i = role ? other : p, j = role ? p : other;
    double xi[3], xj[3], z5[5], w[6];
#pragma unroll
    for (int t = 0; t < 3; ++t) { xi[t] = d.pose_est[3 * i + t]; xj[t] = d.pose_est[3 * j + t]; }
#pragma unroll
    for (int t = 0; t < 5; ++t) z5[t] = d.pp_zinv[5 * (int64_t)k + t];
#pragma unroll
    for (int t = 0; t < 6; ++t) w[t] = d.pp_info[6 * (int64_t)k + t];
    double si, ci;
    if (have_cs && i == p) { si = own_s; ci = own_c; }
    else if (have_cs) { const double2 t2 = reinterpret_cast<const double2 *>(d.pose_cs)[i]; ci = t2.x; si = t2.y; }   // fused kernel: cached
    else sincos(xi[2], &si, &ci);
    return pp_incidence<WRITE_H>(d, k, role, xi, xj, ci, si, z5, w, d.pose_fixed[i], d.pose_fixed[j], H, b);
}
template <bool WRITE_H>
__device__ __forceinline__ double pp_incidence_q(const DevGraph &d, int q, int p, double H[6], double b[3], bool have_cs = false, double own_c = 1.0, double own_s = 0.0) {
    return pp_incidence_rec<WRITE_H>(d, reinterpret_cast<const int2 *>(d.ppinc)[q], p, H, b, have_cs, own_c, own_s);
}

// one partial-sum record (64-byte aligned line): two 16-byte loads and one 8-byte load instead of five scattered 8-byte ones
__device__ __forceinline__ void lm_rec_load(const double *lm_part, int64_t slot, double (&v)[5]) {
    const double2 *r = reinterpret_cast<const double2 *>(lm_part + slot * LM_REC);
    const double2 a = r[0], b = r[1]; v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y; v[4] = lm_part[slot * LM_REC + 4];
}
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, WAVE);
    return v;                                                      // valid in lane 0
}
__device__ __forceinline__ double block_sum(double v, double *red) {
    // fixed-order tree: wave shuffle, then the wave partials in wave order (red has >= blockDim/64 slots)
    v = wave_sum(v);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    double r = 0;
    if (threadIdx.x == 0) { for (int k = 0; k < (int)((blockDim.x + 63) >> 6); ++k) r += red[k]; }
    return r;
}

// ---- general fallback: gather kernels on the same ELL arrays (any slot count R; also the chi2-only pass).
//      thread per pose / thread per landmark, every sum in fixed order.
template <bool WRITE_H>
__global__ void __launch_bounds__(256) k_linearize_pose_gather(DevGraph d) {
    __shared__ double red[8];
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    double chi = 0.0;
    if (p < d.N) {
        const double px = d.pose_est[3 * p], py = d.pose_est[3 * p + 1];
        double s, c; sincos(d.pose_est[3 * p + 2], &s, &c);
        const bool fp = d.pose_fixed[p];
        double H[6] = {0, 0, 0, 0, 0, 0}, b[3] = {0, 0, 0};
        const int64_t L = d.ell_len, S = (int64_t)d.ell_T * d.ell_np;
        const bool laid_out = p >= d.ell_p0 && p < d.ell_p0 + d.ell_np;      // pose-window shards: the layout covers the poses this rank sweeps
        for (int sl = 0; laid_out && sl < d.ell_R * d.ell_T; ++sl) {
            const int64_t e = (int64_t)(sl / d.ell_T) * S + (int64_t)d.ell_T * (p - d.ell_p0) + (sl % d.ell_T);
            const int l = d.ell_l[e];
            if (l < 0) continue;
            PlQuad q;
            quad_pl(px, py, c, s, d.lm_est[2 * l], d.lm_est[2 * l + 1], d.ell_z[e], d.ell_z[L + e],
                    d.ell_w[e], d.ell_w[L + e], d.ell_w[2 * L + e], q);
            const bool fl = d.lm_fixed[l];
            if (!(fp && fl)) chi += q.chi;
            if (WRITE_H) {
#pragma unroll
                for (int k = 0; k < 6; ++k) H[k] += q.Hp[k];
#pragma unroll
                for (int k = 0; k < 3; ++k) b[k] += q.bp[k];
                const bool both = !fp && !fl;
#pragma unroll
                for (int k = 0; k < 6; ++k) d.Hpl[k * L + e] = both ? q.W6[k] : 0.0;
            }
        }
        for (int q = d.ppadj_start[p]; q < d.ppadj_start[p + 1]; ++q) chi += pp_incidence_q<WRITE_H>(d, q, p, H, b);
        if (WRITE_H) {
#pragma unroll
            for (int k = 0; k < 6; ++k) d.Hpp_diag[(int64_t)k * d.N + p] = fp ? 0.0 : H[k];
#pragma unroll
            for (int k = 0; k < 3; ++k) d.b_pose[(int64_t)k * d.N + p] = fp ? 0.0 : b[k];
        }
    }
    double tot = block_sum(chi, red);
    if (threadIdx.x == 0) d.chi2_partial[blockIdx.x] = tot;
}

__global__ void __launch_bounds__(256) k_linearize_lm_gather(DevGraph d) {
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= d.M) return;
    double h00 = 0, h01 = 0, h11 = 0, b0 = 0, b1 = 0;
    if (!d.lm_fixed[l]) {
        const double lx = d.lm_est[2 * l], ly = d.lm_est[2 * l + 1];
        const int64_t L = d.ell_len, S = (int64_t)d.ell_T * d.ell_np;
        for (int q = d.lm_start[l]; q < d.lm_start[l + 1]; ++q) {
            const int64_t e = d.lm_edges[q];
            const int p = d.ell_p0 + (int)((e % S) / d.ell_T);
            double s, c; sincos(d.pose_est[3 * p + 2], &s, &c);
            PlQuad r;
            quad_pl(d.pose_est[3 * p], d.pose_est[3 * p + 1], c, s, lx, ly, d.ell_z[e], d.ell_z[L + e],
                    d.ell_w[e], d.ell_w[L + e], d.ell_w[2 * L + e], r);
            h00 += r.Hl[0]; h01 += r.Hl[1]; h11 += r.Hl[2]; b0 += r.bl[0]; b1 += r.bl[1];
        }
    }
    d.Hll_diag[l] = h00; d.Hll_diag[(int64_t)d.M + l] = h01; d.Hll_diag[2 * (int64_t)d.M + l] = h11;
    d.b_lm[l] = b0; d.b_lm[(int64_t)d.M + l] = b1;
}

// ---- fused kernel: T lanes per pose, every wave independent (no block barrier).
// The pass is bound by load latency, not arithmetic, so it is shaped for memory-level parallelism:
//   1. each thread issues, back to back, the loads of ALL its observation edges (<= LIN_R slots of the ELL
//      streams, coalesced: consecutive lanes read consecutive addresses in every slot), its pose state, its
//      odometry incidence record and the descriptors of its landmark-sum items;
//   2. second level, again all at once: landmark gathers and the incidence's endpoint states / z^-1 / information;
//   3. quadratic forms in registers; H_pl blocks are streamed out coalesced (8 B per lane per component); the
//      pose-side sums never leave registers (xor-shuffle over the T lanes of a pose);
//   4. landmark-side shares go through a wave-private LDS region; each lane sums (landmark group, component)
//      items in a fixed order into the per-(wave tile, landmark) partial slot.
static constexpr int LIN_R = 4;                   // == gs::LIN_R: observation slots per lane
#ifndef LIN_WAVES_PER_SIMD
#define LIN_WAVES_PER_SIMD 4
#endif
// base[byte_off / sizeof(T)] with a wave-uniform base and a 32-bit per-lane byte offset: compiles to the
// "SGPR base + VGPR offset" global addressing form, so no 64-bit per-lane address has to live in VGPRs.
template <class Tp> __device__ __forceinline__ Tp ld_off(const Tp *base, uint32_t byte_off) {
    return *reinterpret_cast<const Tp *>(reinterpret_cast<const char *>(base) + byte_off);
}
template <class Tp> __device__ __forceinline__ void st_off(Tp *base, uint32_t byte_off, Tp v) {
    *reinterpret_cast<Tp *>(reinterpret_cast<char *>(base) + byte_off) = v;
}
// streaming variants (read-once / written-once planes of the linearisation pass): non-temporal hint
template <class Tp> __device__ __forceinline__ Tp ld_off_nt(const Tp *base, uint32_t byte_off) {
    return __builtin_nontemporal_load(reinterpret_cast<const Tp *>(reinterpret_cast<const char *>(base) + byte_off));
}
template <class Tp> __device__ __forceinline__ void st_off_nt(Tp *base, uint32_t byte_off, Tp v) {
    __builtin_nontemporal_store(v, reinterpret_cast<Tp *>(reinterpret_cast<char *>(base) + byte_off));
}
// device-scope store (sc1): written through the XCD's L2 instead of staying dirty in it until the end-of-kernel
// write-back that makes a kernel's output visible to the other XCDs
template <class Tp> __device__ __forceinline__ void st_off_wt(Tp *base, uint32_t byte_off, Tp v) {
    __hip_atomic_store(reinterpret_cast<Tp *>(reinterpret_cast<char *>(base) + byte_off), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// LIN_NT bit 0: non-temporal loads of the read-once ELL streams, bit 1: non-temporal stores of the Hpl planes.
// Measured on MI355X, cfg4 (105 MB per pass, fits the 256 MB Infinity Cache), microseconds per launch:
//   back-to-back launches  nt=0 26.1  nt=1 30.4  nt=2 26.8  nt=3 29.8   (streams still cached from the last launch)
//   inside a GN iteration  nt=0 37.5  nt=1 33.0  nt=2 38.7  nt=3 33.8   (streams cold: the solver moved ~1 GB since)
// and cfg5 (1 GB per pass) back to back: nt=0 230, nt=3 210.  The iteration is what ships: loads non-temporal.
// Re-measured with the final solver (L panels non-temporal): nt=0 30.7 us in-iteration but the factor phase after it
// 252 us and 2540 it/s; nt=1 31.7 us, factor 244 us, 2600 it/s — streaming the read-once inputs past the caches is
// worth more to the kernels that follow than to this one.
// Round 3, re-measured with this round's solver (scripts/r3_f.sh, same box, whole iterations; microseconds lin / factor, it/s):
//   cfg4  nt=0 28.5 / 170, 3638   nt=1 32.0 / 170, 3590   nt=3 32.5 / 173, 3545   nt=5 (write-through stores) 32.0 / 170, 3598
//   cfg5  nt=0 209 / 960, 626     nt=1 209 / 965, 622     nt=5 207 / 972, 622
// the factor phase no longer gains from streamed inputs (its own traffic is streamed now), the pass itself loses 3.5 us at
// cfg4 with them: plain loads.
#ifndef LIN_NT
#define LIN_NT 0
#endif
#if LIN_NT & 1
#define LD_S ld_off_nt
#else
#define LD_S ld_off
#endif
#if LIN_NT & 4
#define ST_S st_off_wt
#elif LIN_NT & 2
#define ST_S st_off_nt
#else
#define ST_S st_off
#endif
// LIN_TS (tuning builds only): 100 MHz timestamps of three wave tiles (first, middle, last) at the phase boundaries,
// without extra waits — they show where a wave stalls.  Slots 40 + 8 * {0, 1, 2} + phase of dbg_ts.
#ifndef LIN_TS
#define LIN_TS 0
#endif
// LIN_ABL (tuning builds only, results are WRONG): pieces of the pass switched off to attribute its HBM traffic and time — bit 0 no odometry
// incidences, bit 1 no landmark gathers, bit 2 no H_pl stores, bit 3 no landmark-group phase (scripts/r4_b.sh)
#ifndef LIN_ABL
#define LIN_ABL 0
#endif
#if LIN_TS
#define LTS(i) do { if (ts_k >= 0 && lane == 0) d.dbg_ts[40 + 8 * ts_k + (i)] = wall_clock64(); } while (0)
#else
#define LTS(i) do { } while (0)
#endif
template <int T>
__global__ void __launch_bounds__(256, LIN_WAVES_PER_SIMD) k_linearize_ell(DevGraph d) {
    constexpr int PW = 64 / T;
    __shared__ double s_lc[4][5][LIN_R * 64];       // 40 KB per block = 4 blocks per CU in the 160 KB LDS: 4 waves per SIMD
    // the wave id is wave-uniform: say so (readfirstlane), otherwise everything derived from it is treated as divergent
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int wt = d.wt_lo + blockIdx.x * 4 + wave;                // a shard only sweeps the wave tiles it has edges in
    if (wt >= d.wt_hi) return;                                     // whole wave leaves; no block-level barrier below
#if LIN_TS
    const int ts_k = wt == 0 ? 0 : (wt == d.n_wtiles / 2 ? 1 : (wt == d.n_wtiles - 1 ? 2 : -1));
#endif
    LTS(0);
    const int p = wt * PW + lane / T, h = lane % T;
    const bool live = p < d.N;
    const int64_t L = d.ell_len, S = (int64_t)T * d.ell_np;         // the layout starts at pose ell_p0 = wt_lo * PW (pose-window shards: the swept tiles only)
    const int R = d.ell_R;
    // ---- first round trip: pose state, incidence range, tile descriptor AND the streams of slots 0-1 — none of these
    // addresses depends on a loaded value, so everything is issued before the first wait
    const uint32_t off8 = (uint32_t)(T * (p - d.ell_p0) + h) * 8u;   // byte offset of this lane's edge inside one ELL slot plane (< 4 GiB: host-checked)
    const uint32_t plane8 = (uint32_t)S * 8u;
    struct Slots { int l[2]; uint32_t dst[2]; double zx[2], zy[2], w00[2], w01[2], w11[2]; };
    auto load_slots = [&](int c, Slots &e) {
#pragma unroll
        for (int j = 0; j < 2; ++j) { const int i = c + j;
            e.l[j] = -1; e.dst[j] = 0xFFFFu; e.zx[j] = e.zy[j] = e.w00[j] = e.w01[j] = e.w11[j] = 0.0;
            if (live && i < R) { const uint32_t o = off8 + (uint32_t)i * plane8;      // SGPR base + 32-bit lane offset addressing
                e.l[j] = LD_S(d.ell_l, o >> 1); e.dst[j] = LD_S(d.ell_dst, o >> 2); e.zx[j] = LD_S(d.ell_z, o); e.zy[j] = LD_S(d.ell_z + L, o);
                e.w00[j] = LD_S(d.ell_w, o); e.w01[j] = LD_S(d.ell_w + L, o); e.w11[j] = LD_S(d.ell_w + 2 * L, o); } }
    };
    double px = 0, py = 0; bool fp = true; int q0 = 0, q1 = 0;       // theta itself is not needed: its cos/sin are cached per pose
    if (live) { px = d.pose_est[3 * p]; py = d.pose_est[3 * p + 1]; fp = d.pose_fixed[p];
                q0 = d.ppadj_start[p]; q1 = d.ppadj_start[p + 1]; }
    const int4 wd = reinterpret_cast<const int4 *>(d.wt_desc)[wt];        // {first group, #groups, first position, #positions}
    double sn = 0.0, cs = 1.0;                                     // cos/sin of theta are kept per pose (k_update): no sincos here
    if (live) { const double2 t2 = reinterpret_cast<const double2 *>(d.pose_cs)[p]; cs = t2.x; sn = t2.y; }
    Slots e0; load_slots(0, e0);
    // ---- second round trip: what needs a loaded index — first odometry incidence, landmark-sum item descriptors (and,
    // inside the chunk, the landmark estimates of slots 0-1)
    int2 inc0 = make_int2(-1, 0);                                 // this lane's first odometry incidence, fetched now, used after the edges
    if (q0 + h < q1) inc0 = reinterpret_cast<const int2 *>(d.ppinc)[q0 + h];
    const int g0 = wd.x, ng = wd.y, nitems = ng * 5;
    int it_s[2] = {0, 0}, it_e[2] = {0, 0}, it_slot[2] = {0, 0};
#pragma unroll
    for (int u = 0; u < 2; ++u) { const int item = lane + 64 * u;
        if (item < nitems) { const int2 gt = reinterpret_cast<const int2 *>(d.grp_tab)[g0 + item % ng];      // {first | end << 16 (tile-local positions), partial-sum slot}: one 8-byte load
            it_s[u] = gt.x & 0xffff; it_e[u] = gt.x >> 16; it_slot[u] = gt.y; } }
    // ---- observation edges, two slots at a time: loads of both slots, both landmark gathers, then the arithmetic.
    // (All four slots at once need ~170 VGPRs = 3 waves per SIMD, and 100k poses are 3125 waves for 3072 slots: a
    // second round for 53 waves.  Two at a time fit 128 VGPRs = 4 waves per SIMD: one round, and the other three
    // waves of the SIMD cover the shorter per-thread load queue.)
    double H[6] = {0, 0, 0, 0, 0, 0}, b[3] = {0, 0, 0}, chi = 0.0;
    auto do_slots = [&](int c, const Slots &e) {
        double lx[2], ly[2]; bool fl[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) { lx[j] = ly[j] = 0.0; fl[j] = true;
            if (e.l[j] >= 0) {
#if LIN_ABL & 2
                lx[j] = 1.0; ly[j] = 2.0; fl[j] = false;
#else
                lx[j] = d.lm_est[2 * e.l[j]]; ly[j] = d.lm_est[2 * e.l[j] + 1]; fl[j] = d.lm_fixed[e.l[j]];
#endif
            } }
#pragma unroll
        for (int j = 0; j < 2; ++j) { const int i = c + j;
            double hl0 = 0, hl1 = 0, hl2 = 0, bl0 = 0, bl1 = 0;
            if (e.l[j] >= 0) {
                const uint32_t o = off8 + (uint32_t)i * plane8;
                PlQuad q;
                quad_pl(px, py, cs, sn, lx[j], ly[j], e.zx[j], e.zy[j], e.w00[j], e.w01[j], e.w11[j], q);
                if (!(fp && fl[j])) chi += q.chi;
                const bool both = !fp && !fl[j];
#if !(LIN_ABL & 4)
#pragma unroll
                for (int k = 0; k < 6; ++k) ST_S(d.Hpl + k * L, o, both ? q.W6[k] : 0.0);
#endif
#pragma unroll
                for (int k = 0; k < 6; ++k) H[k] += q.Hp[k];
#pragma unroll
                for (int k = 0; k < 3; ++k) b[k] += q.bp[k];
                hl0 = q.Hl[0]; hl1 = q.Hl[1]; hl2 = q.Hl[2]; bl0 = q.bl[0]; bl1 = q.bl[1];
            }
            if (e.dst[j] != 0xFFFFu) { const int pos = e.dst[j];      // an edge another shard evaluates (l < 0) still owns its position: zeros
                s_lc[wave][0][pos] = hl0; s_lc[wave][1][pos] = hl1; s_lc[wave][2][pos] = hl2; s_lc[wave][3][pos] = bl0; s_lc[wave][4][pos] = bl1; }
        }
    };
    LTS(1);
    do_slots(0, e0);
    LTS(2);
    if (R > 2) { Slots e1; load_slots(2, e1); do_slots(2, e1); }     // uniform (prefetching these under slots 0-1: measured, no change)
    LTS(3);
    // ---- odometry incidences: lane h takes incidences q0+h, q0+h+T, ... of its pose
#if !(LIN_ABL & 1)
    if (q0 + h < q1) chi += pp_incidence_rec<true>(d, inc0, p, H, b, true, cs, sn);
    for (int q = q0 + h + T; q < q1; q += T) chi += pp_incidence_q<true>(d, q, p, H, b, true, cs, sn);
#endif
    LTS(4);
    // ---- pose sums: xor-shuffle over the T lanes of the pose, then each lane stores its share of the 9 components
#pragma unroll
    for (int off = 1; off < T; off <<= 1) {
#pragma unroll
        for (int k = 0; k < 6; ++k) H[k] += __shfl_xor(H[k], off, WAVE);
#pragma unroll
        for (int k = 0; k < 3; ++k) b[k] += __shfl_xor(b[k], off, WAVE);
    }
    if (live) {                                                    // lane h stores planes h, h + T, ...: the 9 planes (Hpp_diag, then
        double *plane0 = d.Hpp_diag + p;                           // b_pose) are contiguous in the arena, stride N
#pragma unroll
        for (int j = 0; j * T < 9; ++j) {
            double v = (j * T < 6) ? H[(j * T) % 6] : b[(j * T - 6 + 3) % 3];
#pragma unroll
            for (int t = 1; t < T; ++t) if (j * T + t < 9) { const int k = j * T + t; v = (h == t) ? (k < 6 ? H[k % 6] : b[(k + 3 - 6) % 3]) : v; }
            const int k = j * T + h;
            if (j * T + T - 1 < 9 || k < 9) ST_O(plane0 + (int64_t)k * d.N, fp ? 0.0 : v);
        }
    }
    LTS(5);
    // ---- landmark groups of this wave tile (wave-private LDS region; same-wave LDS accesses are ordered)
    wave_lds_sync();
#if LIN_ABL & 8
    if (lane == 0) d.chi2_partial[wt] = chi + s_lc[wave][0][0]; return;
#endif
#pragma unroll
    for (int u = 0; u < 2; ++u) { const int item = lane + 64 * u;
        if (item < nitems) { const int comp = item / ng;
            double sum = 0.0;
            const double *col = s_lc[wave][comp]; const int e = it_e[u];
            for (int q = it_s[u]; q < e; q += 4) {                 // four LDS reads in flight, added in position order
                const double a0 = col[q], a1 = col[min(q + 1, e - 1)], a2 = col[min(q + 2, e - 1)], a3 = col[min(q + 3, e - 1)];
                sum += a0; sum += (q + 1 < e) ? a1 : 0.0; sum += (q + 2 < e) ? a2 : 0.0; sum += (q + 3 < e) ? a3 : 0.0; }
            ST_O(d.lm_part + (int64_t)it_slot[u] * LM_REC + comp, sum); } }
    for (int item = lane + 128; item < nitems; item += 64) { const int comp = item / ng, gl = item % ng;
        double sum = 0.0;
        const int2 gt = reinterpret_cast<const int2 *>(d.grp_tab)[g0 + gl];
        for (int q = gt.x & 0xffff; q < (gt.x >> 16); ++q) sum += s_lc[wave][comp][q];
        d.lm_part[(int64_t)gt.y * LM_REC + comp] = sum; }
    chi = wave_sum(chi);
    if (lane == 0) d.chi2_partial[wt] = chi;
    LTS(6);
}

// landmark diagonal blocks from the per-(wave tile, landmark) partials (slots ordered by landmark, then tile)
// + the chi2 total (fixed order)
__global__ void __launch_bounds__(256) k_linearize_finalize(DevGraph d, int n_partial) {
    __shared__ double red[8];
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l < d.M) {
        double a[5] = {0, 0, 0, 0, 0};
        if (!d.lm_fixed[l]) {
            for (int q = d.lm_grp_start[l]; q < d.lm_grp_start[l + 1]; ++q) {
#pragma unroll
                for (int k = 0; k < 5; ++k) a[k] += d.lm_part[(int64_t)q * LM_REC + k]; }
        }
        d.Hll_diag[l] = a[0]; d.Hll_diag[(int64_t)d.M + l] = a[1]; d.Hll_diag[2 * (int64_t)d.M + l] = a[2];
        d.b_lm[l] = a[3]; d.b_lm[(int64_t)d.M + l] = a[4];
    }
    if (blockIdx.x == gridDim.x - 1) {
        double s = 0.0;
        for (int k = threadIdx.x; k < n_partial; k += 256) s += d.chi2_partial[k];
        const double tot = block_sum(s, red);
        if (threadIdx.x == 0) d.chi2[0] = tot;
    }
}

__global__ void __launch_bounds__(256) k_reduce_chi2(DevGraph d, int n_partial) {
    __shared__ double red[8];
    double s = 0.0;
    for (int k = threadIdx.x; k < n_partial; k += 256) s += d.chi2_partial[k];
    const double tot = block_sum(s, red);
    if (threadIdx.x == 0) d.chi2[0] = tot;
}

void launch_linearize_gather(const DevGraph &d, hipStream_t st) {
    const int gp = (d.N + 255) / 256, gl = (d.M + 255) / 256;
    if (gp > 0) hipLaunchKernelGGL(k_linearize_pose_gather<true>, dim3(gp), dim3(256), 0, st, d);
    if (gl > 0) hipLaunchKernelGGL(k_linearize_lm_gather, dim3(gl), dim3(256), 0, st, d);
    hipLaunchKernelGGL(k_reduce_chi2, dim3(1), dim3(256), 0, st, d, gp);
}
// materialise H_ll, b_l and the chi2 total from the fused kernel's partials (export / chi2 queries only: inside an
// iteration the front assembly sums the landmark slots itself and k_update totals chi2)
void launch_linearize_finalize(const DevGraph &d, hipStream_t st) {
    if (d.n_wtiles > 0)
        hipLaunchKernelGGL(k_linearize_finalize, dim3(max(1, (d.M + 255) / 256)), dim3(256), 0, st, d, d.n_wtiles + (d.tN > 0 ? 1 : 0));     // (+ the tail's partial)
}
// start / stop (optional): HIP events attached to THIS dispatch (hipExtLaunchKernelGGL) — the kernel's own begin and end as the
// command processor stamps them, what a kernel trace reports; an event recorded before / after the launch also holds the
// hand-over from the previous kernel of the stream
void launch_linearize(const DevGraph &d, hipStream_t st, hipEvent_t start, hipEvent_t stop) {
    if (d.n_wtiles <= 0) { launch_linearize_gather(d, st); return; }
    if (d.wt_hi <= d.wt_lo) return;
    const dim3 grid((d.wt_hi - d.wt_lo + 3) / 4), block(256);
    // (without events: the plain launch — the kernel trace of a lap-sized optimize(10) shows 3 us between k_update and a linearisation
    // dispatched through hipExtLaunchKernelGGL and none in front of the plainly launched kernels, scripts/r4_g.sh)
    if (!start && !stop) { switch (d.ell_T) {
        case 1: hipLaunchKernelGGL(k_linearize_ell<1>, grid, block, 0, st, d); break;
        case 2: hipLaunchKernelGGL(k_linearize_ell<2>, grid, block, 0, st, d); break;
        case 4: hipLaunchKernelGGL(k_linearize_ell<4>, grid, block, 0, st, d); break;
        default: hipLaunchKernelGGL(k_linearize_ell<8>, grid, block, 0, st, d); break; }
        return; }
    switch (d.ell_T) {
        case 1: hipExtLaunchKernelGGL(k_linearize_ell<1>, grid, block, 0, st, start, stop, 0, d); break;
        case 2: hipExtLaunchKernelGGL(k_linearize_ell<2>, grid, block, 0, st, start, stop, 0, d); break;
        case 4: hipExtLaunchKernelGGL(k_linearize_ell<4>, grid, block, 0, st, start, stop, 0, d); break;
        default: hipExtLaunchKernelGGL(k_linearize_ell<8>, grid, block, 0, st, start, stop, 0, d); break;
    }
}
// ---- the tail of a grown plan (grow_plan: poses, cones and edges appended since the plan was built; reference src/slam.cpp:433-459,
// 525-550 add exactly these).  ONE workgroup, behind the main pass; no atomics, every sum in a fixed order:
//   1. a thread per tail observation edge: the same quad_pl as the main kernel; H_pl block to the tail arena, the pose-side and the
//      landmark-side shares to LDS;
//   2. a thread per tail pose sums its edges' pose-side shares in edge order (t_pose_start / t_pose_edges); a thread per landmark
//      the tail touches sums its edges' landmark-side shares in edge order (t_lt_*) and ADDS the sum to the landmark's first
//      partial-sum slot (an old cone: the fronts sum a cone's slots in order) or stores it in the tail arena (a tail cone: all of
//      its edges are tail edges) — one writer per address;
//   3. thread 0 walks the tail odometry edges in order: the new pose's share into the LDS accumulators, the older end's share into
//      its accumulator (a tail pose) or, read-modify-write, into the old pose's H_pp / b entries; off-diagonal blocks to the tail arena;
//   4. the tail poses' diagonal blocks and rhs out; chi2 of the tail (edges in order) as one more partial for k_update.
__global__ void __launch_bounds__(256) k_linearize_tail(DevGraph d) {
    constexpr int EC = 512, PC = 16;                                 // == gs::TAIL_PL, gs::TAIL_POSES (launch_linearize_tail checks)
    __shared__ double s_p[9][EC], s_l[5][EC], s_chi[EC], s_acc[PC][9];
    const int tid = threadIdx.x;
    for (int e = tid; e < d.tEpl; e += 256) {
        const int p = d.t_pl[2 * e], l = d.t_pl[2 * e + 1];
        const double2 cs = reinterpret_cast<const double2 *>(d.pose_cs)[p];
        PlQuad q;
        quad_pl(d.pose_est[3 * p], d.pose_est[3 * p + 1], cs.x, cs.y, d.lm_est[2 * l], d.lm_est[2 * l + 1], d.t_pl_z[2 * e], d.t_pl_z[2 * e + 1],
                d.t_pl_w[3 * e], d.t_pl_w[3 * e + 1], d.t_pl_w[3 * e + 2], q);
        const bool fl = d.lm_fixed[l];
#pragma unroll
        for (int k = 0; k < 6; ++k) { s_p[k][e] = q.Hp[k]; d.t_Hpl[(int64_t)k * d.tcapEpl + e] = fl ? 0.0 : q.W6[k]; }
#pragma unroll
        for (int k = 0; k < 3; ++k) { s_p[6 + k][e] = q.bp[k]; s_l[k][e] = q.Hl[k]; }
        s_l[3][e] = q.bl[0]; s_l[4][e] = q.bl[1]; s_chi[e] = q.chi;   // (a tail pose is free: the edge counts)
    }
    __syncthreads();
    if (tid < d.tN) {                                                // pose side
        double a[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int q = d.t_pose_start[tid]; q < d.t_pose_start[tid + 1]; ++q) { const int e = d.t_pose_edges[q];
#pragma unroll
            for (int k = 0; k < 9; ++k) a[k] += s_p[k][e]; }
#pragma unroll
        for (int k = 0; k < 9; ++k) s_acc[tid][k] = a[k];
    }
    for (int j = tid - 64; j >= 0 && j < d.tLt; j += 192) {          // landmark side (threads 64 ..)
        const int l = d.t_lt_id[j];
        double a[5] = {0, 0, 0, 0, 0};
        for (int q = d.t_lt_start[j]; q < d.t_lt_start[j + 1]; ++q) { const int e = d.t_lt_edges[q];
#pragma unroll
            for (int k = 0; k < 5; ++k) a[k] += s_l[k][e]; }
        if (l >= d.M) { const int o = l - d.M; const int64_t S = d.tcapM;
            d.t_Hll_diag[o] = a[0]; d.t_Hll_diag[S + o] = a[1]; d.t_Hll_diag[2 * S + o] = a[2]; d.t_b_lm[o] = a[3]; d.t_b_lm[S + o] = a[4]; }
        else { const int slot = d.lm_grp_start[l];
#pragma unroll
            for (int k = 0; k < 5; ++k) d.lm_part[(int64_t)slot * LM_REC + k] += a[k]; }
    }
    __syncthreads();
    double chi = 0.0;
    if (tid == 0) {
        for (int kk = 0; kk < d.tEpp; ++kk) { const int i = d.t_pp_ij[2 * kk], j = d.t_pp_ij[2 * kk + 1];
            const int64_t k = (int64_t)d.Epp + kk;
            double xi[3], xj[3], z5[5], w[6];
#pragma unroll
            for (int c = 0; c < 3; ++c) { xi[c] = d.pose_est[3 * i + c]; xj[c] = d.pose_est[3 * j + c]; }
#pragma unroll
            for (int c = 0; c < 5; ++c) z5[c] = d.pp_zinv[5 * k + c];
#pragma unroll
            for (int c = 0; c < 6; ++c) w[c] = d.pp_info[6 * k + c];
            const double2 ci = reinterpret_cast<const double2 *>(d.pose_cs)[i];
            const bool fi = d.pose_fixed[i], fj = d.pose_fixed[j];
            double Hi[6] = {0, 0, 0, 0, 0, 0}, bi[3] = {0, 0, 0}, Hj[6] = {0, 0, 0, 0, 0, 0}, bj[3] = {0, 0, 0};
            chi += pp_incidence<true>(d, (int)k, 0, xi, xj, ci.x, ci.y, z5, w, fi, fj, Hi, bi, d.t_Hpp_off + kk, d.tcapEpp);
            pp_incidence<true>(d, (int)k, 1, xi, xj, ci.x, ci.y, z5, w, fi, fj, Hj, bj);
            auto share = [&](int v, const double *H, const double *b) {   // endpoint v's share of the edge
                if (d.pose_fixed[v]) return;
                if (v >= d.N) {
#pragma unroll
                    for (int c = 0; c < 6; ++c) s_acc[v - d.N][c] += H[c];
#pragma unroll
                    for (int c = 0; c < 3; ++c) s_acc[v - d.N][6 + c] += b[c]; }
                else { const int64_t S = d.N;
#pragma unroll
                    for (int c = 0; c < 6; ++c) d.Hpp_diag[c * S + v] += H[c];
#pragma unroll
                    for (int c = 0; c < 3; ++c) d.b_pose[c * S + v] += b[c]; } };
            share(i, Hi, bi); share(j, Hj, bj);
        }
        for (int e = 0; e < d.tEpl; ++e) chi += s_chi[e];
        d.chi2_partial[d.n_wtiles] = chi;                            // one more partial for k_update's total
    }
    __syncthreads();
    if (tid < d.tN) {
#pragma unroll
        for (int c = 0; c < 6; ++c) d.t_Hpp_diag[(int64_t)c * d.tcapN + tid] = s_acc[tid][c];
#pragma unroll
        for (int c = 0; c < 3; ++c) d.t_b_pose[(int64_t)c * d.tcapN + tid] = s_acc[tid][6 + c];
    }
}
void launch_linearize_tail(const DevGraph &d, hipStream_t st) {
    if (d.tN > 0 && d.tcapEpl <= 512 && d.tcapN <= 16) hipLaunchKernelGGL(k_linearize_tail, dim3(1), dim3(256), 0, st, d);
}
void launch_chi2_only(const DevGraph &d, hipStream_t st) {
    if (d.tN > 0 && d.n_wtiles > 0) { launch_linearize(d, st); launch_linearize_tail(d, st); launch_linearize_finalize(d, st); return; }   // a grown plan: the full pass (the gather kernels do not know the tail)
    const int gp = (d.N + 255) / 256;
    if (gp > 0) hipLaunchKernelGGL(k_linearize_pose_gather<false>, dim3(gp), dim3(256), 0, st, d);
    hipLaunchKernelGGL(k_reduce_chi2, dim3(1), dim3(256), 0, st, d, gp);
}

// ------------------------------------------------------------------ A8 factorisation
// One workgroup per front.  The (f+1) x f frontal matrix (last row = rhs) lives in LDS, column-major with
// an odd leading dimension; lower triangle only.  assemble originals -> extend-add children -> partial
// Cholesky of the npiv pivot columns -> L panel and update matrix to HBM.
__device__ __forceinline__ void apply_asm(const DevGraph &d, const int32_t *rec, double *F, int ld, int f) {
    const int kind = rec[0], src = rec[1], r0 = rec[2], c0 = rec[3];
    switch (kind) {
        case 0: {   // pose diagonal (packed xx xy xt yy yt tt) + rhs
            const double *H = d.Hpp_diag + src; const int64_t S = d.N;
            F[(c0 + 0) * ld + r0 + 0] += H[0];     F[(c0 + 0) * ld + r0 + 1] += H[S];     F[(c0 + 0) * ld + r0 + 2] += H[2 * S];
            F[(c0 + 1) * ld + r0 + 1] += H[3 * S]; F[(c0 + 1) * ld + r0 + 2] += H[4 * S]; F[(c0 + 2) * ld + r0 + 2] += H[5 * S];
#pragma unroll
            for (int c = 0; c < 3; ++c) F[(c0 + c) * ld + f] += d.b_pose[c * S + src];
        } break;
        case 1: {   // landmark diagonal (packed 00 01 11) + rhs
            if (d.n_wtiles > 0) {
                // fused linearisation: sum the landmark's per-(wave tile) partial slots here, in slot order
                double a[5] = {0, 0, 0, 0, 0};
                for (int q = d.lm_grp_start[src]; q < d.lm_grp_start[src + 1]; ++q) {
#pragma unroll
                    for (int k = 0; k < 5; ++k) a[k] += d.lm_part[(int64_t)q * LM_REC + k]; }
                F[(c0 + 0) * ld + r0 + 0] += a[0]; F[(c0 + 0) * ld + r0 + 1] += a[1]; F[(c0 + 1) * ld + r0 + 1] += a[2];
                F[(c0 + 0) * ld + f] += a[3]; F[(c0 + 1) * ld + f] += a[4];
            } else {
                const double *H = d.Hll_diag + src; const int64_t S = d.M;
                F[(c0 + 0) * ld + r0 + 0] += H[0]; F[(c0 + 0) * ld + r0 + 1] += H[S]; F[(c0 + 1) * ld + r0 + 1] += H[2 * S];
                F[(c0 + 0) * ld + f] += d.b_lm[src]; F[(c0 + 1) * ld + f] += d.b_lm[S + src];
            }
        } break;
        case 2: case 3: {
            const double *H = d.Hpp_off + src; const int64_t S = d.Epp;
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) F[(c0 + b) * ld + r0 + a] += (kind == 2) ? H[(3 * a + b) * S] : H[(3 * b + a) * S];
        } break;
        case 4: {   // 3x2 as is
            const double *H = d.Hpl + src; const int64_t S = d.ell_len;
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) F[(c0 + b) * ld + r0 + a] += H[(2 * a + b) * S];
        } break;
        default: {  // 2x3 transposed
            const double *H = d.Hpl + src; const int64_t S = d.ell_len;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) F[(c0 + b) * ld + r0 + a] += H[(2 * b + a) * S];
        } break;
    }
}

// mode 0: a front of this rank's own subtree (or the single-GPU case): originals + all children, factorise.
// mode 1: this rank's CONTRIBUTION to a shared front (pose-window shards): originals it evaluated + the update
//         matrices of the children it owns, written to the front's slot of the exchange buffer; no factorisation.
// mode 2: a shared front after the all-reduce: start from the summed slot, add the shared children, factorise.
enum { FRONT_OWN = 0, FRONT_CONTRIB = 1, FRONT_TOP = 2 };
// pose-window shards: a rank's failure flag travels with its contribution (one extra double behind the shared fronts'
// slots, summed by the same all-reduce), so that EVERY rank skips the update of an iteration in which ANY rank met a
// zero pivot — g2o stops the whole optimisation at that iteration and keeps the previous iterate.  The local phase
// (linearise, own subtrees) has completed on this stream when the contribution launch starts.
__device__ __forceinline__ void contrib_publish_fail(const DevGraph &d) {
    // two slots at the tail of the exchange buffer, summed over the ranks by the same all-reduce: [0] a zero pivot (or a failure heard of earlier) somewhere,
    // [1] a whole-tree launch that gave up on a flag — not a property of H: the ranks can run the iteration again (gs_dist_optimize does)
    if (d.xfail_off >= 0) { const int c = d.fail[0]; d.exchange[d.xfail_off] = (c != 0 && c != 2 && c != 4) ? 1.0 : 0.0; d.exchange[d.xfail_off + 1] = (c == 2 || c == 4) ? 1.0 : 0.0; }
}
template <bool USE_LDS>
__global__ void __launch_bounds__(256) k_factor_level(DevGraph d, int level_off, int mode) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int s = d.level_fronts[level_off + blockIdx.x];
    const DevFront fr = d.fronts[s];
    const int npiv = fr.npiv, nbnd = fr.nbnd, f = npiv + nbnd, ld = (f + 1) | 1;
    double *F = USE_LDS ? smem : d.front_ws + (int64_t)blockIdx.x * d.front_ws_stride;
    const int tid = threadIdx.x;
    if (mode == FRONT_TOP) {
        const double *X = d.exchange + d.x_off[s];
        for (int idx = tid; idx < (f + 1) * f; idx += 256) { int c = idx / (f + 1), r = idx - c * (f + 1); F[c * ld + r] = X[idx]; }
        __syncthreads();
    } else {
        for (int idx = tid; idx < ld * f; idx += 256) F[idx] = 0.0;
        __syncthreads();
        const int nuniq = fr.asm_cnt - fr.asm_dup;
        for (int t = tid; t < nuniq; t += 256) apply_asm(d, d.asm_recs + 4 * (int64_t)(fr.asm_off + t), F, ld, f);
        __syncthreads();
        if (fr.asm_dup > 0) {
            if (tid == 0) for (int t = nuniq; t < fr.asm_cnt; ++t) apply_asm(d, d.asm_recs + 4 * (int64_t)(fr.asm_off + t), F, ld, f);
            __syncthreads();
        }
    }
    for (int ci = 0; ci < fr.child_cnt; ++ci) {
        const int c = d.children[fr.child_off + ci];
        const DevFront ch = d.fronts[c];
        if (mode == FRONT_CONTRIB && ch.owner != d.rank) continue;      // block-uniform
        if (mode == FRONT_TOP && ch.owner >= 0) continue;               // owned children came in through the all-reduce
        const int nb = ch.nbnd, ldu = nb + 1;
        const double *U = d.Ubuf + ch.U_off;
        const int32_t *map = d.child_map + ch.map_off;
        for (int idx = tid; idx < ldu * nb; idx += 256) {
            int col = idx / ldu, row = idx - col * ldu;
            if (row >= col) {
                int pr = (row == nb) ? f : map[row], pc = map[col];
                F[pc * ld + pr] += U[idx];
            }
        }
        __syncthreads();
    }
    if (mode == FRONT_CONTRIB) {
        double *X = d.exchange + d.x_off[s];
        for (int idx = tid; idx < (f + 1) * f; idx += 256) { int c = idx / (f + 1), r = idx - c * (f + 1); X[idx] = (r >= c) ? F[c * ld + r] : 0.0; }
        if (blockIdx.x == 0 && tid == 0) contrib_publish_fail(d);
        return;
    }
    // right-looking partial Cholesky
    const int tx = tid & 15, ty = tid >> 4;
    for (int k = 0; k < npiv; ++k) {
        double piv = F[k * ld + k];
        if (!(piv > 0.0)) { if (tid == 0) atomicMax(d.fail, 1); piv = 1.0; }        // Cholesky (sqrt): fails on d <= 0 like Eigen's SimplicialLLT
        const double dd = sqrt(piv), inv = 1.0 / dd;
        __syncthreads();
        for (int r = k + 1 + tid; r <= f; r += 256) F[k * ld + r] *= inv;
        if (tid == 0) F[k * ld + k] = dd;
        __syncthreads();
        for (int c = k + 1 + ty; c < f; c += 16) {
            const double lc = F[k * ld + c];
            for (int r = c + tx; r <= f; r += 16) F[c * ld + r] -= F[k * ld + r] * lc;
        }
        __syncthreads();
    }
    // L panel: (f+1) x npiv, ld = f+1
    double *L = d.Lbuf + fr.L_off;
    const int ldl = f + 1;
    for (int idx = tid; idx < ldl * npiv; idx += 256) { int c = idx / ldl, r = idx - c * ldl; L[idx] = F[c * ld + r]; }
    double *U = d.Ubuf + fr.U_off;
    const int ldu = nbnd + 1;
    for (int idx = tid; idx < ldu * nbnd; idx += 256) { int c = idx / ldu, r = idx - c * ldu;
        U[idx] = (r >= c) ? F[(npiv + c) * ld + npiv + r] : 0.0; }
}

static constexpr int LDS_LIMIT_BYTES = 160 * 1024;
// hipFuncAttributeMaxDynamicSharedMemorySize is a property of (function, DEVICE): a process that opens handles on two
// devices must set it on each (a process-wide "done" flag would launch the 80 KB LDS kernels on the second device without it)
static void allow_max_lds(const void *fn) {
    static std::mutex mu; static std::set<std::pair<int, const void *>> done;
    int dev = 0; (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(mu);
    if (done.insert({dev, fn}).second) (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_LIMIT_BYTES);
}
int factor_lds_limit_f() {
    int f = 1;
    while ((int64_t)(((f + 2) | 1)) * (f + 1) * 8 <= LDS_LIMIT_BYTES) ++f;
    return f;
}

// tile-image layout of a 64 x 64 front: the 10 lower 16 x 16 tiles, tile (I, J) at index I(I+1)/2 + J, row-major
// inside a tile.  It is the LDS staging layout of the matrix-core front kernels: lane l of register q of tile t holds
// (row 16I + (l >> 4) + 4q, col 16J + (l & 15)), the C/D layout of v_mfma_f64_16x16x4_f64.
static constexpr int MF_IMG = 2560;
__device__ __forceinline__ int mf_tile(int I, int J) { return ((I * (I + 1)) >> 1) + J; }

__device__ __forceinline__ double lane_bcast(double v, int src_lane) {      // src_lane is wave-uniform
    int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
    return __hiloint2double(hi, lo);
}
typedef double v4d __attribute__((ext_vector_type(4)));

// F3_NT: bit 0 = the L panels (written once by the factor kernel, read once by the backward solve) move with
// non-temporal stores / loads, so that 2 x 141 MB per iteration do not sweep the caches the update matrices and the
// original values live in.  Measured at cfg4 (factor phase, us / GN iterations per s): 0: 281 / 2310, 1: 255 / 2480,
// 3 (+ records): 256 / 2485, 5 (+ original values): 267 / 2410, 7: 268 / 2415.
#ifndef F3_NT
#define F3_NT 1
#endif
#if F3_NT & 8      // bit 3 (experiment): the L panels written THROUGH the XCD's L2 (sc1) — nothing of them stays dirty until the end-of-kernel write-back
#define F3_ST_L(ptr, v) __hip_atomic_store((ptr), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define F3_LD_L(ptr) __builtin_nontemporal_load(ptr)
#elif F3_NT & 1
#define F3_ST_L(ptr, v) __builtin_nontemporal_store((v), (ptr))
#define F3_LD_L(ptr) __builtin_nontemporal_load(ptr)
#else
#define F3_ST_L(ptr, v) (*(ptr) = (v))
#define F3_LD_L(ptr) (*(ptr))
#endif
#if F3_NT & 2      // bit 1: the scalar assembly records (read once per iteration)
#define F3_LD_REC(ptr) __builtin_nontemporal_load(ptr)
#else
#define F3_LD_REC(ptr) (*(ptr))
#endif
#if F3_NT & 4      // bit 2: the original H values (written by the linearisation pass, read once here)
#define F3_LD_VAL(base, off) ld_off_nt(base, off)
#else
#define F3_LD_VAL(base, off) ld_off(base, off)
#endif
// ---- variant 3: latency-shaped wave-per-front kernels (f <= 63).  The time of a level is the latency of ONE front
// (the top of the tree has fewer fronts than the chip has wave slots, and the leaf level is a few rounds of that
// latency), so these kernels are built around the number of dependent memory round trips per front:
//   1. one 96-byte descriptor per level position (front, its first two children, arena offsets): one coalesced load;
//   2. assembly records + the children's inverse row maps;                     (one round trip)
//   3. the original H values, then the children's update matrices: read in storage order (contiguous loads) and added
//      into the front's LDS image by source, children in list order (sum order originals + child 0 + child 1 + ...); see
//      "by source" below.  (Round 1 gathered by destination, 40 scattered loads per lane and child, to avoid the LDS
//      read-modify-write; round 2 measured what a scattered wave access costs — ~60 cycles of address processing per
//      instruction — and that a packed matrix is only 7-14 contiguous loads per lane.  LDS f64 atomics stay out:
//      ~270 cycles per wave instruction.)
//   4. LDL^T panels of 4 pivots on the fp64 matrix cores: T(I,J) -= (c/d) c^T, one reciprocal per pivot instead of a
//      square root and a division; the rhs is row f of the front, so row f of the L panel is D^-1 L^-1 b and the
//      backward solve has a unit diagonal (no division there either).
// Record formats (built in gs_api.cpp upload_graph):
//   f3_desc [level position][24] int32: 0 front, 1 npiv, 2 nbnd, 3 asm_off, 4 #unique records, 5 #duplicate records,
//      6 #children, 7 child_off, 8-9 L_off, 10 piv0, 11 bnd_off, 12-13 child front (-1 none), 14 offset of the front's children table in f3_x (F3X ints per child: its 64-entry row
//      table, then {front, update-matrix offset, size, owner}), 15 unused,
//      16-17 child owner, 18-19 exchange slot offset, 20 sc_off, 21 #scalar records (multiple of 64), 22 lm_off, 23 #landmark records
//   sc3 [scalar][2]: {offset of the value in H_arena, offset in the staging image}: the original blocks flattened to
//      scalars, so the assembly is branch-free (load record, load value, one LDS store)
//   lm3 [record][4]: {#slots, first slot, r0, c0}: landmark diagonal blocks of the fused linearisation (sum of the
//      per-wave-tile partial slots, in slot order)
//   asm3 [record][4]: kind | count << 8, src, r0, c0 — as asm_recs, except that a landmark diagonal record carries its
//      partial-sum slot range (src = first slot, count = #slots) so that no index load precedes the value loads
//   update matrices are PACKED lower triangles: boundary row r' (0 .. nbnd, the last one = rhs) holds its columns
//      0 .. min(r', nbnd - 1) at r'(r'+1)/2 — half the bytes of 16 x 16 tile images, and a parent row's columns are
//      contiguous (fewer cache lines per gather).  Behind each: one double that stays zero (clamped gathers read it)
//      and one that swallows clamped stores.  offset(r', c') = rowpart(r') + colpart(c'), so both the child's store and
//      the leaf instance's by-destination store addresses an element with one add:
//   child table [64] int32 (behind the descriptor for the first two children, in f3_x for all): for boundary row r' of
//      the CHILD (r' = its nbnd: the rhs row) the place of the parent row it lands on in the parent's tile image, as
//      {row part (low 16 bits), column part (high 16 bits)}: image index of (R, C) = rowpart(R) + colpart(C).
//      own store table [64] (leaf instance): row of the front -> {rowpart = r'(r'+1)/2 * 8, colpart = r' * 8} byte
//      offsets in its packed update matrix, -30000 = none (the sum goes negative, the store is clamped to the spare
//      double).  All three sit right behind the descriptor (f3_desc stride 224 ints): known after the FIRST round trip.
struct F3 {
    int s, npiv, nbnd, asm_off, asm_uniq, asm_dup, nchild, child_off, piv0, bnd_off, c_id[2], x_tab, c_owner[2];
    int sc_off, sc_cnt, lm_off, lm_cnt, u_off, u_size, c_uoff[2], c_usize[2], parent, level;
    int64_t L_off, x_off;
};
static constexpr int F3_INTS = 32, F3_STRIDE = 224;   // 32 descriptor ints, table of child 0, table of child 1, own store table (64 ints each)
__device__ __forceinline__ F3 f3_load(const int32_t *desc, int idx, int lane) {
    const int v = (lane < F3_INTS) ? desc[(int64_t)idx * F3_STRIDE + lane] : 0;
    auto g = [&](int i) { return __builtin_amdgcn_readlane(v, i); };
    F3 r;
    r.s = g(0); r.npiv = g(1); r.nbnd = g(2); r.asm_off = g(3); r.asm_uniq = g(4); r.asm_dup = g(5); r.nchild = g(6); r.child_off = g(7);
    r.L_off = (int64_t)(uint32_t)g(8) | ((int64_t)g(9) << 32); r.piv0 = g(10); r.bnd_off = g(11);
    r.c_id[0] = g(12); r.c_id[1] = g(13); r.x_tab = g(14); r.c_owner[0] = g(16); r.c_owner[1] = g(17);
    r.x_off = (int64_t)(uint32_t)g(18) | ((int64_t)g(19) << 32);
    r.sc_off = g(20); r.sc_cnt = g(21); r.lm_off = g(22); r.lm_cnt = g(23);
    r.u_off = g(24); r.u_size = g(25); r.c_uoff[0] = g(26); r.c_uoff[1] = g(27); r.c_usize[0] = g(28); r.c_usize[1] = g(29); r.parent = g(30); r.level = g(31);
    return r;
}

// values of one assembly record (phase A: loads only)
__device__ __forceinline__ void asm3_load(const DevGraph &d, int kind_cnt, int src, double (&v)[9]) {
    const int kind = kind_cnt & 0xff;
#pragma unroll
    for (int k = 0; k < 9; ++k) v[k] = 0.0;
    switch (kind) {
        case 0: { const bool tl = src >= d.N;                           // a tail pose (grow_plan): its blocks live in the tail arenas
            const int64_t S = tl ? d.tcapN : d.N; const int i = tl ? src - d.N : src;
            const double *Hd = tl ? d.t_Hpp_diag : d.Hpp_diag, *bp = tl ? d.t_b_pose : d.b_pose;
#pragma unroll
            for (int k = 0; k < 6; ++k) v[k] = Hd[k * S + i];
#pragma unroll
            for (int k = 0; k < 3; ++k) v[6 + k] = bp[k * S + i]; } break;
        case 1: {
            if (d.n_wtiles > 0) { const int cnt = kind_cnt >> 8;
                for (int q0 = 0; q0 < cnt; q0 += 4) {                // four slots' loads in flight, added in slot order
                    double t[4][5];
#pragma unroll
                    for (int j = 0; j < 4; ++j) { const int q = min(q0 + j, cnt - 1);
                        lm_rec_load(d.lm_part, src + q, t[j]); }
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int k = 0; k < 5; ++k) v[k] += (q0 + j < cnt) ? t[j][k] : 0.0; }
            } else { const int64_t S = d.M;
                v[0] = d.Hll_diag[src]; v[1] = d.Hll_diag[S + src]; v[2] = d.Hll_diag[2 * S + src]; v[3] = d.b_lm[src]; v[4] = d.b_lm[S + src]; }
        } break;
        case 6: { const int64_t S = d.tcapM; const int i = src - d.M;        // a landmark appended after the plan was built: summed by k_linearize_tail
            v[0] = d.t_Hll_diag[i]; v[1] = d.t_Hll_diag[S + i]; v[2] = d.t_Hll_diag[2 * S + i]; v[3] = d.t_b_lm[i]; v[4] = d.t_b_lm[S + i]; } break;
        case 2: case 3: { const bool tl = src >= d.Epp;
            const int64_t S = tl ? d.tcapEpp : d.Epp; const double *H = tl ? d.t_Hpp_off + (src - d.Epp) : d.Hpp_off + src;
#pragma unroll
            for (int k = 0; k < 9; ++k) v[k] = H[k * S]; } break;
        default: { const bool tl = src >= d.ell_len;
            const int64_t S = tl ? (int64_t)d.tcapEpl : d.ell_len; const double *H = tl ? d.t_Hpl + (src - d.ell_len) : d.Hpl + src;
#pragma unroll
            for (int k = 0; k < 6; ++k) v[k] = H[k * S]; } break;
    }
}
// staging of a front's original entries before they go to the accumulators.  StageT<false>: the 64 x 64 tile image
// (20 KB).  StageT<true>: only the pivot columns, column-major (f+1) x npiv with an odd leading dimension (~9 KB) —
// leaves have nothing else to stage, and the smaller footprint is what lets three workgroups share a CU.
template <bool PANEL> struct StageT;
template <> struct StageT<false> { double *F; int f, ld;
    __device__ __forceinline__ double &at(int r, int c) const { return F[mf_tile(r >> 4, c >> 4) * 256 + (r & 15) * 16 + (c & 15)]; }
    __device__ __forceinline__ int img(int off) const { return off; } };
template <> struct StageT<true> { double *F; int f, ld;
    __device__ __forceinline__ double &at(int r, int c) const { return F[c * ld + r]; }
    __device__ __forceinline__ int img(int off) const {            // tile-image offset (the records' format) -> panel offset
        const int t = off >> 8, I = t >= 6 ? 3 : (t >= 3 ? 2 : (t >= 1 ? 1 : 0)), J = t - ((I * (I + 1)) >> 1);
        return (16 * J + (off & 15)) * ld + 16 * I + ((off >> 4) & 15); } };
// phase B: the values into the staging image
template <bool ADD, class St>
__device__ __forceinline__ void asm3_put(const St &P, int kind_cnt, int r0, int c0, const double (&v)[9]) {
    const int kind = kind_cnt & 0xff, f = P.f;
    auto put = [&](int r, int c, double x) { if (ADD) P.at(r, c) += x; else P.at(r, c) = x; };
    switch (kind) {
        case 0:
            put(r0, c0, v[0]); put(r0 + 1, c0, v[1]); put(r0 + 2, c0, v[2]);
            put(r0 + 1, c0 + 1, v[3]); put(r0 + 2, c0 + 1, v[4]); put(r0 + 2, c0 + 2, v[5]);
            put(f, c0, v[6]); put(f, c0 + 1, v[7]); put(f, c0 + 2, v[8]);
            break;
        case 1: case 6:
            put(r0, c0, v[0]); put(r0 + 1, c0, v[1]); put(r0 + 1, c0 + 1, v[2]); put(f, c0, v[3]); put(f, c0 + 1, v[4]);
            break;
        case 2: case 3:
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) put(r0 + a, c0 + b, (kind == 2) ? v[3 * a + b] : v[3 * b + a]);
            break;
        case 4:
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) put(r0 + a, c0 + b, v[2 * a + b]);
            break;
        default:
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) put(r0 + a, c0 + b, v[2 * b + a]);
            break;
    }
}

// device-scope (sc1) load: fetched from the memory side, not from this XCD's L2 — data another XCD wrote during THIS kernel
__device__ __forceinline__ double ld_off_coh(const double *base, uint32_t byte_off) {
    return __hip_atomic_load(reinterpret_cast<const double *>(reinterpret_cast<const char *>(base) + byte_off), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// ---- by source (round 2, non-leaf fronts).  Measured on MI355X: a wave's SCATTERED 8-byte global access costs ~60
// cycles of address processing per instruction — the 80 by-destination loads of two children are 2 us of every level's
// chain, the 40 by-destination stores of the update matrix 1 us — while a CONTIGUOUS one (lane i at byte 8 i) costs a
// few.  A packed update matrix is 7-14 doubles per lane when read or written in storage order, so the non-leaf fronts
// move it that way and do the permutation in LDS: a child's element idx = 64 k + lane is boundary entry (r', c') of the
// child (a function of idx alone: each lane keeps its 14 (r', c') pairs for the whole kernel), the child's table maps r'
// and c' to their place in the parent's tile image {row part (low 16 bits), column part (high 16 bits)}, and the lane adds
// its element there (read, add, write; children in list order: the sum order originals + child 0 + child 1 + ... is the
// by-destination one's).  The places depend on the plan only: a front that waits for its children computes them before
// the wait.  Going out, the accumulators pass through the image and every lane reads its elements' places back.
static constexpr int F3_KREG = 14;                                   // elements per lane kept in registers: update matrices up to 896 doubles (boundary <= 40)
__device__ __forceinline__ int f3_rc_of(int idx) {                   // packed index -> r' | c' << 8 (row r' holds columns 0 .. r' at r'(r'+1)/2)
    int r = (int)((__fsqrt_rn(8.0f * (float)idx + 1.0f) - 1.0f) * 0.5f);
    if (((r * (r + 1)) >> 1) > idx) --r;
    if ((((r + 1) * (r + 2)) >> 1) <= idx) ++r;
    return r | ((idx - ((r * (r + 1)) >> 1)) << 8);
}
__device__ __forceinline__ int f3_img_rowpart(int R) { const int I = R >> 4; return (((I * (I + 1)) >> 1) << 8) + ((R & 15) << 4); }
__device__ __forceinline__ int f3_img_colpart(int C) { return ((C >> 4) << 8) + (C & 15); }
// places in the parent's image of a child's elements: tab = the child's table (lane r' holds {row part, column part} of row r').
// Everything below is branch-free per lane (a lane beyond the matrix loads the zero double behind it and adds it to image
// element 1 — row 0, column 1: upper triangle of a diagonal tile, don't-care everywhere) so that the loads, the LDS reads
// and the LDS writes of a group each go out back to back: one latency per group, not one per element.  Elements
// [K0, K1) of every lane; the callers take 0-8 always and 8-14 for matrices of more than 512 doubles.
template <int K0, int K1>
__device__ __forceinline__ void f3_child_places(int tab, const int (&rc)[F3_KREG], int usize, int lane, int (&dst)[F3_KREG]) {
#pragma unroll
    for (int k = K0; k < K1; ++k) {
        const int p = (__shfl(tab, rc[k] & 0xff, WAVE) & 0xffff) + (int)((uint32_t)__shfl(tab, rc[k] >> 8, WAVE) >> 16);
        dst[k] = (64 * k + lane < usize) ? p : 1;
        asm volatile("" : "+v"(dst[k])); }                         // computed HERE (before the caller's wait), not sunk behind it
}
template <int K0, int K1>
__device__ __forceinline__ void f3_child_loads(const double *Uc, int usize, int lane, double (&st)[F3_KREG]) {
#pragma unroll
    for (int k = K0; k < K1; ++k) st[k] = ld_off_coh(Uc, (uint32_t)min(64 * k + lane, usize) * 8u);
}
template <int K0, int K1>
__device__ __forceinline__ void f3_child_scatter(double *img, const int (&dst)[F3_KREG], const double (&st)[F3_KREG]) {
    double o[F3_KREG];                                               // a child's places are distinct: all reads, then all writes
#pragma unroll
    for (int k = K0; k < K1; ++k) o[k] = img[dst[k]];
#pragma unroll
    for (int k = K0; k < K1; ++k) img[dst[k]] = o[k] + st[k];
}
// elements 896 .. of a large update matrix (boundary > 40 rows), one row of 64 at a time
__device__ __forceinline__ void f3_child_tail(double *img, const double *Uc, int usize, int tab, int lane) {
    for (int base = 64 * F3_KREG; base < usize; base += 64) {
        const int idx = base + lane, rc = f3_rc_of(min(idx, usize - 1));
        const int p = (__shfl(tab, rc & 0xff, WAVE) & 0xffff) + (int)((uint32_t)__shfl(tab, rc >> 8, WAVE) >> 16);
        const int dst = idx < usize ? p : 1;
        const double v = ld_off_coh(Uc, (uint32_t)min(idx, usize) * 8u);
        img[dst] += v;
    }
}

#ifndef F3_RCP_NEWTON
#define F3_RCP_NEWTON 2
#endif
#ifndef F3_EXACT_DIV
#define F3_EXACT_DIV 0
#endif
__device__ __forceinline__ double rcp_f64(double x) {             // reciprocal to ~1 ulp: hardware seed + Newton steps
    double r = __builtin_amdgcn_rcp(x);
    double e = fma(-x, r, 1.0); r = fma(r, e, r);
#if F3_RCP_NEWTON >= 2
    e = fma(-x, r, 1.0); r = fma(r, e, r);
#endif
    return r;
}

// the pivots of the panel k0 .. k0+3 on its four columns, one row per lane: p <- l (0 above the diagonal, 1 on it), dd <- the pivots (0
// for a column that is not a pivot).  (Measured alternative: every lane factorises the 4 x 4 diagonal block itself from LDS
// broadcast reads and eliminates its own row against it — bit-identical, no lane broadcasts in the chain of four
// reciprocals, 35 more VALU instructions per panel: the chain of the upper levels unchanged, the leaf level slower,
// factor 0.179 against 0.170 ms.)
__device__ __forceinline__ void f3_panel_pivots(int k0, bool &bad, double (&p)[4], double (&dd)[4], int npiv, int lane) {    // k0 (first pivot of the panel): uniform
    // one pivot of the panel.  If every column of this panel is a pivot (all panels of a front but possibly the last)
    // there are no uniform branches between the pivots and no merges of the p[] registers after them.
    auto pivot = [&](int j, bool is_pivot) {
        const int col = k0 + j;
        dd[j] = 0.0;
        if (is_pivot) {
            const double piv = lane_bcast(p[j], col);
            bad = bad || !(fabs(piv) > 0.0);                        // LDL^T: a ZERO pivot fails (Eigen 3.3.4 SimplicialCholesky_impl.h:172-176: d == 0), a negative one does not; NaN is reported too
#if F3_EXACT_DIV                                                     // (experiment: a division per entry instead of one reciprocal per pivot — scripts/r3_j.sh)
            const double lj = (lane >= col) ? p[j] / piv : 0.0;
#else
            const double inv = rcp_f64(piv);
            const double lj = (lane >= col) ? p[j] * inv : 0.0;    // row col itself gets d / d = 1: a dead row in every later use
#endif
#pragma unroll
            for (int j2 = j + 1; j2 < 4; ++j2) { const double c2 = lane_bcast(p[j], k0 + j2); p[j2] -= lj * c2; }
            p[j] = lj;
            dd[j] = piv;
        } else p[j] = 0.0;                                          // not a pivot: contributes nothing to the update
    };
    if (k0 + 4 <= npiv) {                                           // uniform
#pragma unroll
        for (int j = 0; j < 4; ++j) pivot(j, true);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) pivot(j, k0 + j < npiv);
    }
}

template <int B, int NT = 4>      // panel B: pivots 4B .. 4B+3 (tile column J0 = B / 4), LDL^T; NT = tile rows of the front (3: fronts of <= 47 scalars)
__device__ __forceinline__ bool f3_panel_step(bool &bad, v4d (&acc)[NT * (NT + 1) / 2], double *Pn, double *L, int npiv, int f, int lane) {
    constexpr int k0 = 4 * B, J0 = B / 4, jc = (B % 4) * 4;
    static_assert(J0 < NT, "panel beyond the front's tile rows");
    if (k0 >= npiv) return false;                                   // uniform
    const int lc = lane & 15, lr = lane >> 4;
    // The wide levels of the tree are bound by instruction issue, so this function is written for few instructions:
    // one exec-mask region for the spill and one for the L stores, no uniform branches inside them, the positivity
    // test as one compare per pivot into a flag that is looked at once per front (no protective select: a front with a
    // bad pivot produces garbage, the solve is reported as failed).  (One instance per panel, sixteen in a row: a loop
    // over the panels of a tile column was measured slower here — 0.186 against 0.176 ms — and faster in the four-wave
    // form below, which runs every front on a CU whose instruction cache has not seen the code.)
    // 1. the panel's four columns (rows of tiles J0..3) to LDS, row-major 64 x 4 (tile rows beyond f hold zeros)
    if (lc >= jc && lc < jc + 4) {
#pragma unroll
        for (int I = J0; I < NT; ++I)
#pragma unroll
            for (int q = 0; q < 4; ++q) Pn[(16 * I + lr + 4 * q) * 4 + (lc - jc)] = acc[mf_tile(I, J0)][q];
    }
    wave_lds_sync();
    // 2. factorise the panel, lane r = row r:  l = c / d from the diagonal down (the diagonal itself becomes 1), 0 above
    double p[4], dd[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) p[j] = Pn[lane * 4 + j];
    f3_panel_pivots(k0, bad, p, dd, npiv, lane);
    // the panel's columns of L (the diagonal is 1), one masked region.  Rows above the panel are zero and nobody reads them (the
    // backward solve uses a column from its diagonal down): not stored — whole 64-byte groups of them, a fifth of a leaf's L bytes
    if (lane >= (k0 & ~7) && lane <= f) {
        double *Lc = L + (int64_t)k0 * (f + 1) + lane;
#pragma unroll
        for (int j = 0; j < 4; ++j) if (k0 + j < npiv) F3_ST_L(Lc + (int64_t)j * (f + 1), p[j]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) Pn[lane * 4 + j] = p[j];
    wave_lds_sync();
    // 3. trailing update on the matrix cores: T(I,J) -= A_I B_J^T, A = l (scaled), B = l * d (the unscaled column)
    const double dk = lr == 0 ? dd[0] : (lr == 1 ? dd[1] : (lr == 2 ? dd[2] : dd[3]));      // k = lane >> 4
    double a[NT];
#pragma unroll
    for (int I = J0; I < NT; ++I) a[I] = Pn[(16 * I + lc) * 4 + lr];  // A[i = lane & 15][k = lane >> 4] = l[16 I + i][k]
#pragma unroll
    for (int I = J0; I < NT; ++I) if (16 * I <= f) {
#pragma unroll
        for (int J = J0; J <= I; ++J)
            acc[mf_tile(I, J)] = __builtin_amdgcn_mfma_f64_16x16x4f64(-a[I], a[J] * dk, acc[mf_tile(I, J)], 0, 0, 0); }
    wave_lds_sync();
    return true;
}

// ---- whole-tree launches (single GPU).  One launch factorises every level: wave w takes level position w, so a
// front's children always sit in EARLIER workgroups.  Dispatch is in workgroup order per XCD, hence whatever a resident
// wave waits for is resident too or already done — no level barrier, no kernel boundary between levels, and a front's
// descriptor / record / value round trips run while its children still work.  A front publishes its update matrix
// with device-scope (sc1, write-through) stores, drains them (s_waitcnt), then sets done[front] = epoch; the parent
// polls that flag and gathers with device-scope loads (the XCDs' L2s are not coherent with each other inside a kernel).
// The poll is bounded: on expiry the front carries on, reports through d.fail, and the grid still drains.
#ifndef F3_POLL_SLEEP
#define F3_POLL_SLEEP 4
#endif
// `fail` = d.fail: every 64 polls the waiter also looks at the launch's failure code and LEAVES AT ONCE when any front has
// reported one (returns true: nothing more to report; the iteration applies no update anyway) — so after the first expiry
// of a poll budget (~30 ms) the rest of the grid drains immediately instead of every front spending its own budget.
__device__ __forceinline__ bool f3_wait_flag(const int32_t *flag, int epoch, const int32_t *fail) {
    for (int it = 0; it < (1 << 18); ++it) {
        if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); return true; }
        if ((it & 63) == 63 && __hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return true;
        __builtin_amdgcn_s_sleep(F3_POLL_SLEEP);
    }
    return false;
}
// "Not written yet": a signalling-NaN bit pattern (both 32-bit halves equal, so hipMemsetD32 fills it).  Arithmetic
// never produces a signalling NaN, so no computed value can be mistaken for it.  The whole-tree backward solve hands
// a parent's solution to its children through the data itself: every factor front resets its own rows of xe to this
// pattern, the parent stores its values (write-through, no wait for the acknowledgement, no flag) and a child polls the
// rows it needs until none of them holds the pattern — one memory round trip per level less than store / wait / flag.
#define F3_UNSET_BITS 0x7FF4A5A57FF4A5A5ull
#define F3_UNSET_WORD 0x7FF4A5A5u
__device__ __forceinline__ bool f3_is_unset(double v) { return (unsigned long long)__double_as_longlong(v) == F3_UNSET_BITS; }
__device__ __forceinline__ double f3_unset() { return __longlong_as_double((long long)F3_UNSET_BITS); }
__device__ __forceinline__ void f3_publish(int32_t *flag, int epoch, int lane) {
    __builtin_amdgcn_s_waitcnt(0);                                  // this wave's write-through stores have been acknowledged
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    if (lane == 0) __hip_atomic_store(flag, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- tickets (round 4).  The whole-tree launches rest on "a front's children sit in earlier workgroups": with workgroups handed out in
// grid order, whatever a resident wave waits for is resident or done.  Grid-order dispatch is how the hardware behaves, not something the
// programming model promises.  With tickets the order is the launch's own: a workgroup's first act is to take the next number of a counter in
// HBM, and that number — not blockIdx — names its work.  The numbers taken at any moment are a prefix of the launch, so a waiting workgroup's
// children belong to workgroups that have started: progress no longer depends on the dispatcher.  One device-scope atomic per workgroup; the
// counter is never reset inside an iteration — the host knows how many workgroups every ticketed launch of the stream has had (ticket_base).
__device__ __forceinline__ int wg_ticket(const DevGraph &d, double *slot) {      // slot: 8 bytes of LDS nobody else touches before the second barrier
    if (d.tickets == nullptr) return (int)blockIdx.x;
    volatile unsigned *s = reinterpret_cast<volatile unsigned *>(slot);
    if (threadIdx.x == 0) *s = atomicAdd(d.tickets, 1u) - d.ticket_base;
    __syncthreads();
    const unsigned t = *s;
    __syncthreads();
    return __builtin_amdgcn_readfirstlane((int)t);
}
// two children of a front into its LDS image, by source.  flag != nullptr: a child of the SAME launch — its flag is awaited
// (the places are computed first: nothing but the loads and the read-add-writes is behind the wait).  The loads are
// device-scope either way: in storage order every line is fetched once, so there is nothing a cached load would save.
__device__ __forceinline__ bool f3_gather_pair(double *img, const int (&rc)[F3_KREG], int lane, int epoch, const int32_t *fail,
        bool onA, const double *UA, int uszA, int tabA, const int32_t *flagA, bool onB, const double *UB, int uszB, int tabB, const int32_t *flagB) {
    int dA[F3_KREG], dB[F3_KREG]; double sA[F3_KREG], sB[F3_KREG];
    const bool bigA = uszA > 512, bigB = uszB > 512;                 // uniform
    if (onA) { f3_child_places<0, 8>(tabA, rc, uszA, lane, dA); if (bigA) f3_child_places<8, F3_KREG>(tabA, rc, uszA, lane, dA); }
    if (onB) { f3_child_places<0, 8>(tabB, rc, uszB, lane, dB); if (bigB) f3_child_places<8, F3_KREG>(tabB, rc, uszB, lane, dB); }
    bool ok = true;                                                  // child A's loads are issued while child B may still be working
    if (onA) { if (flagA) ok = f3_wait_flag(flagA, epoch, fail) && ok;
        f3_child_loads<0, 8>(UA, uszA, lane, sA); if (bigA) f3_child_loads<8, F3_KREG>(UA, uszA, lane, sA); }
    if (onB) { if (flagB) ok = f3_wait_flag(flagB, epoch, fail) && ok;
        f3_child_loads<0, 8>(UB, uszB, lane, sB); if (bigB) f3_child_loads<8, F3_KREG>(UB, uszB, lane, sB); }
    if (onA) { f3_child_scatter<0, 8>(img, dA, sA); if (bigA) f3_child_scatter<8, F3_KREG>(img, dA, sA);
        if (uszA > 64 * F3_KREG) f3_child_tail(img, UA, uszA, tabA, lane); }
    if (onB) { f3_child_scatter<0, 8>(img, dB, sB); if (bigB) f3_child_scatter<8, F3_KREG>(img, dB, sB);
        if (uszB > 64 * F3_KREG) f3_child_tail(img, UB, uszB, tabB, lane); }
    return ok;
}

#ifndef F3_DONE_TS
#define F3_DONE_TS 0
#endif
// ---- four waves per front: the upper levels of a whole-tree launch (round 2).  Up there the chain of dependent fronts is
// the whole cost and the chip is nearly empty, so a front gets a workgroup: the ten accumulator tiles are spread over the
// four waves (tile t belongs to wave t mod 4: three tiles at most, MI355X runs fp64 MFMA at the vector rate — one wave's ten
// tiles are ~1000 of a panel's ~2200 cycles), the records, the children's elements and the update matrix over 256
// threads.  Every wave factorises the panel itself (the same instructions on the same values, in lockstep on four SIMDs),
// so a panel costs one workgroup barrier.  Element by element the arithmetic and its order are the wave-per-front
// kernel's: the results are bit-identical.
__device__ __forceinline__ int f3_tile_row(int t) { return t >= 6 ? 3 : (t >= 3 ? 2 : (t >= 1 ? 1 : 0)); }
__device__ __forceinline__ bool f3_block_panel(int B, bool &bad, v4d (&acc)[3], double *Pn, double *Pw, double *L, int npiv, int f, int wave, int lane) {
    const int k0 = 4 * B, J0 = B >> 2, jc = (B & 3) * 4;             // uniform
    if (k0 >= npiv) return false;                                   // uniform over the workgroup
    const int lc = lane & 15, lr = lane >> 4;
    double *Pb = Pn + (B & 1) * 256;                                 // two buffers: the next panel's spill must not meet this panel's readers
    // 1. the panel's four columns out of whichever waves own the tiles of tile column J0
#pragma unroll
    for (int s = 0; s < 3; ++s) { const int t = 4 * s + wave, I = f3_tile_row(t), J = t - ((I * (I + 1)) >> 1);     // uniform per wave
        if (t < 10 && J == J0 && lc >= jc && lc < jc + 4) {
#pragma unroll
            for (int q = 0; q < 4; ++q) Pb[(16 * I + lr + 4 * q) * 4 + (lc - jc)] = acc[s][q]; } }
    __syncthreads();
    // 2. every wave factorises the panel (lane r = row r)
    double p[4], dd[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) p[j] = Pb[lane * 4 + j];
    f3_panel_pivots(k0, bad, p, dd, npiv, lane);
    if (wave == 0 && lane >= (k0 & ~7) && lane <= f) {             // (rows above the panel: zero, never read, not stored)
        double *Lc = L + (int64_t)k0 * (f + 1) + lane;
#pragma unroll
        for (int j = 0; j < 4; ++j) if (k0 + j < npiv) F3_ST_L(Lc + (int64_t)j * (f + 1), p[j]);
    }
    // 3. trailing update of the wave's own tiles; the operand layout comes through the wave's private buffer (no barrier)
#pragma unroll
    for (int j = 0; j < 4; ++j) Pw[lane * 4 + j] = p[j];
    const double dk = lr == 0 ? dd[0] : (lr == 1 ? dd[1] : (lr == 2 ? dd[2] : dd[3]));
#pragma unroll
    for (int s = 0; s < 3; ++s) { const int t = 4 * s + wave, I = f3_tile_row(t), J = t - ((I * (I + 1)) >> 1);
        if (t < 10 && J >= J0) {
            const double aI = Pw[(16 * I + lc) * 4 + lr], aJ = Pw[(16 * J + lc) * 4 + lr];
            acc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(-aI, aJ * dk, acc[s], 0, 0, 0); } }
    wave_lds_sync();
    return true;
}
// children into the image by source, 256 threads: element 256 k + tid, k < 4 (matrices up to 1024 doubles), the rest in a loop
__device__ __forceinline__ void f3_block_places(int tab, const int (&rc)[4], int usz, int tid, int (&dst)[4]) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int pl = (__shfl(tab, rc[k] & 0xff, WAVE) & 0xffff) + (int)((uint32_t)__shfl(tab, rc[k] >> 8, WAVE) >> 16);
        dst[k] = (256 * k + tid < usz) ? pl : 1;
        asm volatile("" : "+v"(dst[k])); }                           // computed HERE (before the caller's wait)
}
__device__ __forceinline__ void f3_block_loads(const double *Uc, int usz, int tid, double (&st)[4]) {
#pragma unroll
    for (int k = 0; k < 4; ++k) st[k] = ld_off_coh(Uc, (uint32_t)min(256 * k + tid, usz) * 8u);
}
__device__ __forceinline__ void f3_block_scatter(double *img, const double *Uc, int usz, int tab, int tid, const int (&dst)[4], const double (&st)[4]) {
    double o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = img[dst[k]];
#pragma unroll
    for (int k = 0; k < 4; ++k) img[dst[k]] = o[k] + st[k];
    for (int base = 1024; base < usz; base += 256) {                // boundary > 43 rows
        const int idx = base + tid, rcv = f3_rc_of(min(idx, usz - 1));
        const int pl = (__shfl(tab, rcv & 0xff, WAVE) & 0xffff) + (int)((uint32_t)__shfl(tab, rcv >> 8, WAVE) >> 16);
        const double v = ld_off_coh(Uc, (uint32_t)min(idx, usz) * 8u);
        img[idx < usz ? pl : 1] += v; }
}
// a leaf's Schur complement (its accumulators, NT tile rows) added into its parent's LDS image: element (R, C) of the leaf front, R >= C,
// both beyond its pivots (R = f: the rhs row), lands at rowpart(tab[R - npiv]) + colpart(tab[C - npiv]) — tab = the leaf's row table in the
// parent (lane r' holds the place of boundary row r').  A child's places are distinct; lanes without an
// element add 0 to image element 1 (don't care, as in the by-source gather).  The SAME values in the SAME order as a leaf that stores
// its update matrix and a parent that gathers it: bit-identical.
template <int NT>
__device__ __forceinline__ void f3_sub_add(double *img, const v4d (&acc)[NT * (NT + 1) / 2], int tab, int npiv, int f, int lane) {
    const int lc = lane & 15, lr = lane >> 4;
    int cpart[NT], rpart[NT][4]; bool cok[NT], rok[NT][4];
#pragma unroll
    for (int J = 0; J < NT; ++J) { const int C = 16 * J + lc; cok[J] = C >= npiv && C < f;
        cpart[J] = (int)((uint32_t)__shfl(tab, min(max(C - npiv, 0), 63), WAVE) >> 16); }
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int R = 16 * I + lr + 4 * q; rok[I][q] = R >= npiv && R <= f;
            rpart[I][q] = __shfl(tab, min(max(R - npiv, 0), 63), WAVE) & 0xffff; }
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
        for (int J = 0; J <= I; ++J) { const int t = mf_tile(I, J);      // a tile at a time (few registers): four reads, four writes
            double o[4]; int pl[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) { const bool ok = rok[I][q] && cok[J] && (I != J || lr + 4 * q >= lc);
                pl[q] = ok ? rpart[I][q] + cpart[J] : 1; o[q] = img[pl[q]] + (ok ? (double)acc[t][q] : 0.0); }
#pragma unroll
            for (int q = 0; q < 4; ++q) img[pl[q]] = o[q]; }
}
template <bool TREE, bool LEAF, int NT, bool MERGED = false>
__device__ __forceinline__ void f3_wave_front(const DevGraph &d, int pos, int mode, int leaf_slot, double *smem, int wave, int lane, bool ts_on, bool first,
                                              v4d (*acc_out)[NT * (NT + 1) / 2] = nullptr, int *npiv_out = nullptr, int *f_out = nullptr);
// SUB: the front's children are LEAVES (level 0, <= 47 scalars) factorised by this workgroup's own waves, four at a time, their Schur
// complements added into the image in list order — nothing of them goes to HBM but their L panels (k_factor3_sub)
template <bool SUB = false, int NTL = 3>      // NTL: tile rows of the leaves (SUB)
__device__ __forceinline__ void f3_block_front(const DevGraph &d, int pos, int mode, int leaf_slot, double *smem, bool ts_on, int sub_leaf_slot = 0) {
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
#define F3B_TS(i) do { if (ts_on) { __builtin_amdgcn_s_waitcnt(0); if (tid == 0) d.dbg_ts[i] = wall_clock64(); } } while (0)
    F3B_TS(0);
    const int tab0 = d.f3_desc[(int64_t)pos * F3_STRIDE + F3_INTS + lane], tab1 = d.f3_desc[(int64_t)pos * F3_STRIDE + F3_INTS + 64 + lane];
    const F3 fr = f3_load(d.f3_desc, pos, lane);
    const int npiv = fr.npiv, f = npiv + fr.nbnd;
    if (wave == 0 && lane < npiv) d.xe[fr.piv0 + lane] = f3_unset();
    double *img = smem, *Pn = smem + MF_IMG, *Pw = smem + MF_IMG + 512 + wave * 256;
    StageT<false> P{img, f, (f + 1) | 1};
    int rc[4], own[4];                                               // (r', c') of this thread's elements of any packed matrix; their places in THIS front's image
    auto own_places = [&]() {
#pragma unroll
        for (int k = 0; k < 4; ++k) { rc[k] = f3_rc_of(256 * k + tid);
            own[k] = min(f3_img_rowpart(npiv + (rc[k] & 0xff)) + f3_img_colpart(npiv + (rc[k] >> 8)), MF_IMG - 1);
            asm volatile("" : "+v"(own[k])); } };
    if constexpr (!SUB) own_places();                                // before the wait for the children (SUB: nothing to wait for — computed behind the leaves, not kept alive across them)
    // ---- records (scalar records come in multiples of 64: a wave's 64 are all there or none)
    // mode TOP (the shared top of a sharded graph): the originals AND the contributions of the ranks' own subtrees arrive summed in
    // the front's exchange slot; only the shared children are gathered
    const bool top = mode == FRONT_TOP;
    const int nsc = top ? 0 : fr.sc_cnt, nlm = top ? 0 : fr.lm_cnt;
    const int2 *sc3 = reinterpret_cast<const int2 *>(d.sc3) + fr.sc_off;
    int2 sc[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) { sc[u] = make_int2(0, 1);
        if (256 * u + 64 * wave < nsc) { const long long v = F3_LD_REC(reinterpret_cast<const long long *>(sc3 + 256 * u + tid)); sc[u] = make_int2((int)(v & 0xffffffffLL), (int)(v >> 32)); } }
    int4 lmr = make_int4(0, 0, 0, 0);
    if (tid < nlm) lmr = reinterpret_cast<const int4 *>(d.lm3)[fr.lm_off + tid];
#pragma unroll
    for (int k = 0; k < 10; ++k) img[256 * k + tid] = 0.0;
    __syncthreads();
    // ---- the original values
    if (top) {
        const double *X = d.exchange + fr.x_off;                     // slot layout: (f+1) x f column-major, ld = f+1
        for (int c = wave; c < f; c += 4) if (lane >= c && lane <= f) P.at(lane, c) = X[c * (f + 1) + lane];
        __syncthreads();
    } else {
        double val[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) if (256 * u + 64 * wave < nsc) val[u] = F3_LD_VAL(d.H_arena, (uint32_t)sc[u].x * 8u);
        double lv[5] = {0, 0, 0, 0, 0};
        if (tid < nlm) {
            for (int q0 = 0; q0 < lmr.x; q0 += 4) {                  // four slots' loads in flight, added in slot order
                double t4[4][5];
#pragma unroll
                for (int j = 0; j < 4; ++j) { const int q = min(q0 + j, lmr.x - 1);
                    lm_rec_load(d.lm_part, lmr.y + q, t4[j]); }
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int k = 0; k < 5; ++k) lv[k] += (q0 + j < lmr.x) ? t4[j][k] : 0.0; } }
#pragma unroll
        for (int u = 0; u < 2; ++u) if (256 * u + 64 * wave < nsc) img[sc[u].y] = val[u];
        for (int base = 512; base + 64 * wave < nsc; base += 256) { const int2 r = sc3[base + tid]; img[r.y] = ld_off(d.H_arena, (uint32_t)r.x * 8u); }
        if (tid < nlm) { const int r0 = lmr.z, c0 = lmr.w;
            P.at(r0, c0) = lv[0]; P.at(r0 + 1, c0) = lv[1]; P.at(r0 + 1, c0 + 1) = lv[2]; P.at(f, c0) = lv[3]; P.at(f, c0 + 1) = lv[4]; }
        for (int t = 256 + tid; t < nlm; t += 256) {
            const int4 r = reinterpret_cast<const int4 *>(d.lm3)[fr.lm_off + t]; double a[5] = {0, 0, 0, 0, 0};
            for (int q = 0; q < r.x; ++q)
#pragma unroll
                for (int k = 0; k < 5; ++k) a[k] += d.lm_part[(int64_t)(r.y + q) * LM_REC + k];
            P.at(r.z, r.w) = a[0]; P.at(r.z + 1, r.w) = a[1]; P.at(r.z + 1, r.w + 1) = a[2]; P.at(f, r.w) = a[3]; P.at(f, r.w + 1) = a[4]; }
        __syncthreads();
        if (fr.asm_dup > 0) {                                        // parallel edges: added one by one, in record order
            if (tid == 0) for (int t = fr.asm_uniq; t < fr.asm_uniq + fr.asm_dup; ++t) {
                const int4 r = reinterpret_cast<const int4 *>(d.asm3)[fr.asm_off + t];
                double w[9]; asm3_load(d, r.x, r.y, w); asm3_put<true>(P, r.x, r.z, r.w, w); }
            __syncthreads();
        }
    }
    F3B_TS(5);
    if constexpr (SUB) {
        // ---- the children are leaves of this workgroup: wave w factorises children w, w + 4, ... (the leaf instance's code, its staging slot
        // behind the image), then the four Schur complements go into the image one after the other, in list order
        const int32_t *xt = d.f3_x + fr.x_tab; const int XS = d.f3x_stride, XH = XS - 8;
        for (int e0 = 0; e0 < fr.nchild; e0 += 4) {                  // (uniform over the workgroup)
            const int e = e0 + wave; const bool has = e < fr.nchild;  // uniform per wave
            v4d cacc[NTL * (NTL + 1) / 2]; int ctab = 0, cnp = 0, cf = 0;
            if (has) { ctab = xt[e * XS + lane]; const int hd = xt[e * XS + XH + (lane & 7)];
                f3_wave_front<true, true, NTL, true>(d, __builtin_amdgcn_readlane(hd, 4), FRONT_OWN, sub_leaf_slot, smem + MF_IMG, wave, lane, false, false, &cacc, &cnp, &cf); }
            __syncthreads();
#pragma unroll 1
            for (int k = 0; k < 4; ++k) { if (wave == k && has) f3_sub_add<NTL>(img, cacc, ctab, cnp, cf, lane); __syncthreads(); }
        }
        own_places();
    } else
    // ---- the children, by source, in list order, two at a time: places before the waits, both children's loads in flight
    // together, a barrier between two children's read-add-writes (their places overlap)
    if (fr.nchild > 0) {
        const bool plain = leaf_slot != 0 && fr.level <= leaf_slot;   // (leaf_slot here: the highest level whose fronts have all their children in earlier launches)
        const int32_t *xt = d.f3_x + fr.x_tab;
        bool okw = true;
        for (int e = 0; e < fr.nchild; e += 2) {
            const bool hasB = e + 1 < fr.nchild;
            int tA, a_id, a_uoff, a_usz, a_own, tB, b_id, b_uoff, b_usz, b_own;
            if (e == 0) { tA = tab0; a_id = fr.c_id[0]; a_uoff = fr.c_uoff[0]; a_usz = fr.c_usize[0]; a_own = fr.c_owner[0];
                          tB = tab1; b_id = fr.c_id[1]; b_uoff = fr.c_uoff[1]; b_usz = fr.c_usize[1]; b_own = fr.c_owner[1]; }
            else { const int eb = hasB ? e + 1 : e;
                const int XS = d.f3x_stride, XH = XS - 8;            // per-child stride of f3_x (72, or 168 when the plan holds a big front), header behind the table
                tA = xt[e * XS + lane]; const int hA = xt[e * XS + XH + (lane & 7)]; tB = xt[eb * XS + lane]; const int hB = xt[eb * XS + XH + (lane & 7)];
                a_id = __builtin_amdgcn_readlane(hA, 0); a_uoff = __builtin_amdgcn_readlane(hA, 1); a_usz = __builtin_amdgcn_readlane(hA, 2); a_own = __builtin_amdgcn_readlane(hA, 3);
                b_id = __builtin_amdgcn_readlane(hB, 0); b_uoff = __builtin_amdgcn_readlane(hB, 1); b_usz = __builtin_amdgcn_readlane(hB, 2); b_own = __builtin_amdgcn_readlane(hB, 3); }
            const bool onA = !(top && a_own >= 0), onB = hasB && !(top && b_own >= 0);      // (uniform) a rank's own subtree: its update matrix came with the exchange slot
            int dA[4], dB[4]; double sA[4], sB[4];
            if (onA) f3_block_places(tA, rc, a_usz, tid, dA);
            if (onB) f3_block_places(tB, rc, b_usz, tid, dB);
            if (onA) { if (!plain) okw = f3_wait_flag(d.done_f + a_id, d.epoch, d.fail) && okw;
                f3_block_loads(d.Uimg + a_uoff, a_usz, tid, sA); }
            if (onB) { if (!plain) okw = f3_wait_flag(d.done_f + b_id, d.epoch, d.fail) && okw;
                f3_block_loads(d.Uimg + b_uoff, b_usz, tid, sB); }
            if (onA) { f3_block_scatter(img, d.Uimg + a_uoff, a_usz, tA, tid, dA, sA); __syncthreads(); }
            if (onB) { f3_block_scatter(img, d.Uimg + b_uoff, b_usz, tB, tid, dB, sB); __syncthreads(); }
        }
        if (!okw && tid == 0) atomicMax(d.fail, 2);
    }
    F3B_TS(10);
    // ---- accumulators: wave w holds tiles w, w + 4, w + 8
    v4d acc[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) { const int t = 4 * s + wave;
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[s][q] = t < 10 ? img[t * 256 + q * 64 + lane] : 0.0; }
    F3B_TS(6);
    double *L = d.Lbuf + fr.L_off;
    bool go = true, bad = false;
#pragma clang loop unroll(disable)
    for (int B = 0; B < 16 && go; ++B) go = f3_block_panel(B, bad, acc, Pn, Pw, L, npiv, f, wave, lane);
    if (d.inject_iter != 0 && d.iter == d.inject_iter && pos == 0 && tid == 0) atomicMax(d.fail, d.inject_code);   // gs_debug_fail_at_iteration (fault injection for tests)
    if (bad && tid == 0) atomicMax(d.fail, 1);
    F3B_TS(7);
    // ---- Schur complement out through the image, contiguous stores
    __syncthreads();                                                 // nobody reads the image any more (the accumulators were loaded long ago): kept for the panel buffers' sake
#pragma unroll
    for (int s = 0; s < 3; ++s) { const int t = 4 * s + wave;
        if (t < 10) {
#pragma unroll
            for (int q = 0; q < 4; ++q) img[t * 256 + q * 64 + lane] = acc[s][q]; } }
    __syncthreads();
    {
        double *U = d.Uimg + fr.u_off; const int usz = fr.u_size;
        double v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = img[own[k]];
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int idx = 256 * k + tid; if (256 * k < usz) st_off_wt(U, (uint32_t)(idx < usz ? idx : usz + 1) * 8u, v[k]); }
        for (int base = 1024; base < usz; base += 256) { const int idx = base + tid; const int rcv = f3_rc_of(min(idx, usz - 1));
            const double w = img[min(f3_img_rowpart(npiv + (rcv & 0xff)) + f3_img_colpart(npiv + (rcv >> 8)), MF_IMG - 1)];
            st_off_wt(U, (uint32_t)(idx < usz ? idx : usz + 1) * 8u, w); }
    }
#if F3_DONE_TS
    if (tid == 0) d.done_ts[fr.s] = wall_clock64();
#endif
    __builtin_amdgcn_s_waitcnt(0);                                  // every wave's write-through stores have been acknowledged
    __syncthreads();
    if (tid == 0) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __hip_atomic_store(d.done_f + fr.s, d.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    F3B_TS(8);
#undef F3B_TS
}

// LEAF: fronts without children (level 0 of a rank's own subtrees, mode OWN only): no gather registers, pivot-column
// staging => 3 waves per SIMD and 3 workgroups per CU instead of 2 (halving the resident waves was measured to cost
// the leaf level x1.67: it is bound by resident waves x front latency, not yet by bandwidth)
// NT = tile rows held in the accumulators: 4 (fronts up to 63 scalars) or, leaf instance only, 3 (every leaf <= 47 scalars:
// 6 tiles instead of 10 — a third fewer accumulator registers and spill traffic, five waves per SIMD instead of four)
// one front on one wave (f <= 63): the body of the wave-per-front kernels.  pos = the front's level position (index of its
// descriptor), smem = the workgroup's dynamic LDS (the wave takes slot `wave`), first = the launch's first position
// MERGED (leaf instance only): the leaf runs INSIDE its parent's workgroup (k_factor3_sub): its Schur complement is not stored — the
// accumulators are handed back (acc_out, with the front's npiv and f) and the caller adds them into the parent's LDS image.
template <bool TREE, bool LEAF, int NT, bool MERGED>
__device__ __forceinline__ void f3_wave_front(const DevGraph &d, int pos, int mode, int leaf_slot, double *smem, int wave, int lane, bool ts_on, bool first,
                                              v4d (*acc_out)[NT * (NT + 1) / 2], int *npiv_out, int *f_out) {
    static_assert(LEAF || NT == 4, "only the leaf instance has a three-tile-row form");
    static_assert(!MERGED || LEAF, "only a leaf runs inside its parent's workgroup");
    constexpr int NTILE = NT * (NT + 1) / 2;
#define F3_TS(i) do { if (ts_on) { __builtin_amdgcn_s_waitcnt(0); if (lane == 0) d.dbg_ts[i] = wall_clock64(); } } while (0)
    F3_TS(0);
    int pv[2];                                                       // the children's row tables ride behind the descriptor
    pv[0] = d.f3_desc[(int64_t)pos * F3_STRIDE + F3_INTS + lane];
    pv[1] = d.f3_desc[(int64_t)pos * F3_STRIDE + F3_INTS + 64 + lane];
    const int sv = d.f3_desc[(int64_t)pos * F3_STRIDE + F3_INTS + 128 + lane];     // own store table
    const F3 fr = f3_load(d.f3_desc, pos, lane);
    const int npiv = fr.npiv, f = npiv + fr.nbnd;
    if (lane < npiv) d.xe[fr.piv0 + lane] = f3_unset();             // this front's solution rows: not written yet (the whole-tree backward solve polls them)
    StageT<LEAF> P{smem + (int64_t)wave * (LEAF ? leaf_slot : MF_IMG), f, (f + 1) | 1};
    const int lc = lane & 15, lr = lane >> 4;
    F3_TS(1);
    const bool top = mode == FRONT_TOP;
    bool use[2];
#pragma unroll
    for (int k = 0; k < 2; ++k)
        use[k] = !LEAF && fr.c_id[k] >= 0 && !(mode == FRONT_CONTRIB && fr.c_owner[k] != d.rank) && !(top && fr.c_owner[k] >= 0);   // uniform
    int rc[F3_KREG], own[F3_KREG];                                   // (r', c') of this lane's elements of any packed update matrix; their places in THIS front's image
    if constexpr (!LEAF) {
#pragma unroll
        for (int k = 0; k < F3_KREG; ++k) { rc[k] = f3_rc_of(64 * k + lane);
            own[k] = min(f3_img_rowpart(npiv + (rc[k] & 0xff)) + f3_img_colpart(npiv + (rc[k] >> 8)), MF_IMG - 1);
            asm volatile("" : "+v"(own[k])); } }                    // computed HERE (before the wait), not rematerialised behind the panels
    // ---- round trip 2b: scalar assembly records (eight per lane up front), landmark records
#ifndef F3_LEAF_BLOCKS
#define F3_LEAF_BLOCKS 1
#endif
    // leaves (F3_LEAF_BLOCKS): their level is bound by bytes, not by a lone wave's instruction count, and a block record
    // (16 bytes for the 5-9 scalars of one H block) is a quarter of the scalar records' bytes
    constexpr bool BLK = LEAF && F3_LEAF_BLOCKS;
    const int nsc = (top || BLK) ? 0 : fr.sc_cnt, nlm = (top || BLK) ? 0 : fr.lm_cnt;
    int4 brec[2];
    if (BLK) {
#pragma unroll
        for (int u = 0; u < 2; ++u) { const int t = lane + 64 * u;
            brec[u] = (t < fr.asm_uniq) ? reinterpret_cast<const int4 *>(d.asm3)[fr.asm_off + t] : make_int4(-1, 0, 0, 0); } }
    const int2 *sc3 = reinterpret_cast<const int2 *>(d.sc3) + fr.sc_off;
    int2 sc[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { sc[u] = make_int2(0, 1);
        if (64 * u < nsc) { const long long v = F3_LD_REC(reinterpret_cast<const long long *>(sc3 + 64 * u + lane)); sc[u] = make_int2((int)(v & 0xffffffffLL), (int)(v >> 32)); } }   // uniform predicate
    int4 lmr = make_int4(0, 0, 0, 0);
    if (lane < nlm) lmr = reinterpret_cast<const int4 *>(d.lm3)[fr.lm_off + lane];
    F3_TS(2);
    // the staging image: originals only touch pivot columns, i.e. tile columns 0 .. (npiv - 1) / 16
    const int Jmax = top ? 3 : ((npiv - 1) >> 4);
    if (LEAF) { for (int idx = lane; idx < P.ld * npiv; idx += 64) P.F[idx] = 0.0; }
    else {
#pragma unroll
        for (int I = 0; I < 4; ++I)
#pragma unroll
            for (int J = 0; J <= I; ++J) {                             // every tile: the children add into the boundary block as well
#pragma unroll
                for (int q = 0; q < 4; ++q) P.F[mf_tile(I, J) * 256 + q * 64 + lane] = 0.0; }
    }
    wave_lds_sync();
    F3_TS(3);
    // ---- round trip 3: the original values
    F3_TS(4);
    if (top) {
        const double *X = d.exchange + fr.x_off;                     // slot layout: (f+1) x f column-major, ld = f+1
        for (int c = 0; c < f; ++c) if (lane >= c && lane <= f) P.at(lane, c) = X[c * (f + 1) + lane];
    } else {
        if (BLK) {
            double bv[2][9];
#pragma unroll
            for (int u = 0; u < 2; ++u) if (brec[u].x >= 0) asm3_load(d, brec[u].x, brec[u].y, bv[u]);
#pragma unroll
            for (int u = 0; u < 2; ++u) if (brec[u].x >= 0) asm3_put<false>(P, brec[u].x, brec[u].z, brec[u].w, bv[u]);
            for (int t = lane + 128; t < fr.asm_uniq; t += 64) {
                const int4 r = reinterpret_cast<const int4 *>(d.asm3)[fr.asm_off + t];
                double w[9]; asm3_load(d, r.x, r.y, w); asm3_put<false>(P, r.x, r.z, r.w, w); }
        }
        double val[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) if (64 * u < nsc) val[u] = F3_LD_VAL(d.H_arena, (uint32_t)sc[u].x * 8u);
        double lv[5] = {0, 0, 0, 0, 0};
        if (lane < nlm) {
            for (int q0 = 0; q0 < lmr.x; q0 += 4) {                  // four slots' loads in flight, added in slot order
                double t[4][5];
#pragma unroll
                for (int j = 0; j < 4; ++j) { const int q = min(q0 + j, lmr.x - 1);
                    lm_rec_load(d.lm_part, lmr.y + q, t[j]); }
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int k = 0; k < 5; ++k) lv[k] += (q0 + j < lmr.x) ? t[j][k] : 0.0; } }
#pragma unroll
        for (int u = 0; u < 8; ++u) if (64 * u < nsc) P.F[P.img(sc[u].y)] = val[u];
        for (int base = 512; base < nsc; base += 64) { const int2 r = sc3[base + lane]; P.F[P.img(r.y)] = ld_off(d.H_arena, (uint32_t)r.x * 8u); }   // fronts with > 512 scalars
        if (lane < nlm) { const int r0 = lmr.z, c0 = lmr.w;
            P.at(r0, c0) = lv[0]; P.at(r0 + 1, c0) = lv[1]; P.at(r0 + 1, c0 + 1) = lv[2]; P.at(f, c0) = lv[3]; P.at(f, c0 + 1) = lv[4]; }
        for (int t = 64 + lane; t < nlm; t += 64) {                  // fronts with > 64 landmark pivots (not on f <= 63 fronts; kept for safety)
            const int4 r = reinterpret_cast<const int4 *>(d.lm3)[fr.lm_off + t]; double a[5] = {0, 0, 0, 0, 0};
            for (int q = 0; q < r.x; ++q)
#pragma unroll
                for (int k = 0; k < 5; ++k) a[k] += d.lm_part[(int64_t)(r.y + q) * LM_REC + k];
            P.at(r.z, r.w) = a[0]; P.at(r.z + 1, r.w) = a[1]; P.at(r.z + 1, r.w + 1) = a[2]; P.at(f, r.w) = a[3]; P.at(f, r.w + 1) = a[4]; }
        wave_lds_sync();
        if (fr.asm_dup > 0) {                                        // parallel edges: added one by one, in record order
            if (lane == 0) for (int t = fr.asm_uniq; t < fr.asm_uniq + fr.asm_dup; ++t) {
                const int4 r = reinterpret_cast<const int4 *>(d.asm3)[fr.asm_off + t];
                double w[9]; asm3_load(d, r.x, r.y, w); asm3_put<true>(P, r.x, r.z, r.w, w); }
        }
    }
    wave_lds_sync();
    F3_TS(5);
    // ---- the children's update matrices, by source, into the image: originals + child 0 + child 1 + ... (list order), two
    // at a time.  The first two children's tables came with the descriptor; further ones (the multi-way splits above the
    // leaves have up to 8 children) come from the front's children table (f3_x: row table + {front, offset, size, owner} per
    // child), the next pair's while this pair is in flight.
    if constexpr (!LEAF) if (fr.nchild > 0) {
        const bool plain = !TREE || (leaf_slot != 0 && fr.level <= leaf_slot);     // (non-leaf instances: leaf_slot = the highest level whose fronts have ALL their children in EARLIER launches — complete and visible, no flags to wait for; 0 = none)
        const int32_t *xt = d.f3_x + fr.x_tab;
        int tA = pv[0], tB = pv[1];
        int a_id = fr.c_id[0], a_uoff = fr.c_uoff[0], a_usz = fr.c_usize[0], a_own = fr.c_owner[0];
        int b_id = fr.c_id[1], b_uoff = fr.c_uoff[1], b_usz = fr.c_usize[1], b_own = fr.c_owner[1];
        bool okw = true;
        for (int e = 0; e < fr.nchild; e += 2) {
            const bool onA = !(mode == FRONT_CONTRIB && a_own != d.rank) && !(top && a_own >= 0);
            const bool onB = e + 1 < fr.nchild && !(mode == FRONT_CONTRIB && b_own != d.rank) && !(top && b_own >= 0);
            int nA = 0, nhA = 0, nB = 0, nhB = 0;
            const bool more = e + 2 < fr.nchild;
            if (more) { const int na = e + 2, nb = min(e + 3, fr.nchild - 1);
                const int XS = d.f3x_stride, XH = XS - 8;
                nA = xt[na * XS + lane]; nhA = xt[na * XS + XH + (lane & 7)]; nB = xt[nb * XS + lane]; nhB = xt[nb * XS + XH + (lane & 7)]; }
            okw = f3_gather_pair(P.F, rc, lane, d.epoch, d.fail, onA, d.Uimg + a_uoff, a_usz, tA, plain ? nullptr : d.done_f + a_id,
                                 onB, d.Uimg + b_uoff, b_usz, tB, plain ? nullptr : d.done_f + b_id) && okw;
            if (e == 0) F3_TS(9);
            if (more) { tA = nA; tB = nB;
                a_id = __builtin_amdgcn_readlane(nhA, 0); a_uoff = __builtin_amdgcn_readlane(nhA, 1); a_usz = __builtin_amdgcn_readlane(nhA, 2); a_own = __builtin_amdgcn_readlane(nhA, 3);
                b_id = __builtin_amdgcn_readlane(nhB, 0); b_uoff = __builtin_amdgcn_readlane(nhB, 1); b_usz = __builtin_amdgcn_readlane(nhB, 2); b_own = __builtin_amdgcn_readlane(nhB, 3); }
        }
        if (!okw && lane == 0) atomicMax(d.fail, 2);
        wave_lds_sync();
    }
    F3_TS(10);
    // ---- accumulators <- the image
    v4d acc[NTILE];
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
        for (int J = 0; J <= I; ++J) { const int t = mf_tile(I, J);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (LEAF) { const int row = 16 * I + lr + 4 * q, col = 16 * J + lc;       // pivot-column panel: (row, col) at col * ld + row
                    acc[t][q] = (J <= Jmax && col < npiv && row <= f) ? P.F[min(col, npiv - 1) * P.ld + min(row, f)] : 0.0; }
                else acc[t][q] = P.F[t * 256 + q * 64 + lane]; } }
    wave_lds_sync();
    if constexpr (!LEAF) if (mode == FRONT_CONTRIB) {                            // this rank's share of a shared front -> exchange slot
#pragma unroll
        for (int t = 0; t < 10; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) P.F[t * 256 + q * 64 + lane] = acc[t][q];
        wave_lds_sync();
        double *X = d.exchange + fr.x_off;
        for (int c = 0; c < f; ++c) if (lane <= f) X[c * (f + 1) + lane] = (lane >= c) ? P.at(lane, c) : 0.0;
        if (first && lane == 0) contrib_publish_fail(d);
        return;
    }
    F3_TS(6);
    // ---- panels of 4 pivots; the LDS image is free now, its first 256 doubles serve as the panel buffer
    double *Pn = P.F;
    double *L = d.Lbuf + fr.L_off;
    bool go = true, bad = false;
    go = go && f3_panel_step<0, NT>(bad, acc, Pn, L, npiv, f, lane);   go = go && f3_panel_step<1, NT>(bad, acc, Pn, L, npiv, f, lane);
    go = go && f3_panel_step<2, NT>(bad, acc, Pn, L, npiv, f, lane);   go = go && f3_panel_step<3, NT>(bad, acc, Pn, L, npiv, f, lane);
    go = go && f3_panel_step<4, NT>(bad, acc, Pn, L, npiv, f, lane);   go = go && f3_panel_step<5, NT>(bad, acc, Pn, L, npiv, f, lane);
    go = go && f3_panel_step<6, NT>(bad, acc, Pn, L, npiv, f, lane);   go = go && f3_panel_step<7, NT>(bad, acc, Pn, L, npiv, f, lane);
    go = go && f3_panel_step<8, NT>(bad, acc, Pn, L, npiv, f, lane);   go = go && f3_panel_step<9, NT>(bad, acc, Pn, L, npiv, f, lane);
    go = go && f3_panel_step<10, NT>(bad, acc, Pn, L, npiv, f, lane);  go = go && f3_panel_step<11, NT>(bad, acc, Pn, L, npiv, f, lane);
    if constexpr (NT == 4) {
    go = go && f3_panel_step<12, NT>(bad, acc, Pn, L, npiv, f, lane);  go = go && f3_panel_step<13, NT>(bad, acc, Pn, L, npiv, f, lane);
    go = go && f3_panel_step<14, NT>(bad, acc, Pn, L, npiv, f, lane);  go = go && f3_panel_step<15, NT>(bad, acc, Pn, L, npiv, f, lane); }
    if (d.inject_iter != 0 && d.iter == d.inject_iter && pos == 0 && lane == 0) atomicMax(d.fail, d.inject_code);   // gs_debug_fail_at_iteration (fault injection for tests)
    if (bad && lane == 0) atomicMax(d.fail, 1);
    F3_TS(7);
    if constexpr (MERGED) {                                          // the parent's workgroup takes the Schur complement from here
#pragma unroll
        for (int t = 0; t < NTILE; ++t) (*acc_out)[t] = acc[t];
        *npiv_out = npiv; *f_out = f;
#if F3_DONE_TS
        if (lane == 0) d.done_ts[fr.s] = wall_clock64();
#endif
        if (lane == 0) d.done_f[fr.s] = d.epoch;
        return;
    }
    // ---- Schur complement out, packed: element (row, col) -> rowpart(row) + colpart(col) from the front's own table; pivot
    // rows / columns make the sum negative and the upper-triangle lanes of the diagonal tiles are forced there: the
    // unsigned min sends all of those to the spare double behind the matrix
    if constexpr (!LEAF) {
        // non-leaf fronts: by source — the accumulators pass through the image, every lane reads its elements (boundary entry
        // (r', c') = image (npiv + r', npiv + c')) and the stores are contiguous: 13 instead of 40 scattered ones
        double *U = d.Uimg + fr.u_off;
        const int usz = fr.u_size;
#pragma unroll
        for (int t = 0; t < 10; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) P.F[t * 256 + q * 64 + lane] = acc[t][q];
        wave_lds_sync();
        auto place = [&](int rcv) { return min(f3_img_rowpart(npiv + (rcv & 0xff)) + f3_img_colpart(npiv + (rcv >> 8)), MF_IMG - 1); };
        auto out = [&](auto K0c, auto K1c) {                         // branch-free per lane: lanes beyond the matrix store to the spare double behind it
            constexpr int K0 = decltype(K0c)::value, K1 = decltype(K1c)::value;
            double v[F3_KREG];
#pragma unroll
            for (int k = K0; k < K1; ++k) v[k] = P.F[own[k]];
#pragma unroll
            for (int k = K0; k < K1; ++k) { const int idx = 64 * k + lane; const uint32_t o = (uint32_t)(idx < usz ? idx : usz + 1) * 8u;
                if (TREE) st_off_wt(U, o, v[k]); else st_off(U, o, v[k]); } };
        out(std::integral_constant<int, 0>{}, std::integral_constant<int, 8>{});
        if (usz > 512) out(std::integral_constant<int, 8>{}, std::integral_constant<int, F3_KREG>{});
        for (int base = 64 * F3_KREG; base < usz; base += 64) {      // boundary > 40 rows
            const int idx = base + lane;
            const double v = P.F[place(f3_rc_of(min(idx, usz - 1)))];
            const uint32_t o = (uint32_t)(idx < usz ? idx : usz + 1) * 8u;
            if (TREE) st_off_wt(U, o, v); else st_off(U, o, v); }
    } else {
        double *U = d.Uimg + fr.u_off;
        const uint32_t dump = (uint32_t)(fr.u_size + 1) * 8u;
        int co[NT], rov[NT][4];                                      // all lane permutes of the store table first: one LDS latency, not one per row
#pragma unroll
        for (int J = 0; J < NT; ++J) co[J] = __shfl(sv, 16 * J + lc, WAVE) >> 16;
#pragma unroll
        for (int I = 0; I < NT; ++I)
#pragma unroll
            for (int q = 0; q < 4; ++q) rov[I][q] = (int)(short)(__shfl(sv, 16 * I + lr + 4 * q, WAVE) & 0xffff);
#pragma unroll
        for (int I = 0; I < NT; ++I) {
            if (16 * I + 15 < npiv || 16 * I > f) continue;         // uniform
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int ro = rov[I][q];
#pragma unroll
                for (int J = 0; J <= I; ++J) {
                    if (16 * J + 15 < npiv) continue;                // uniform
                    int off = ro + co[J];
                    if (I == J) off = (lr + 4 * q < lc) ? -1 : off;
#ifndef F3_LEAF_UWT
#define F3_LEAF_UWT 0
#endif
                    if ((TREE && !LEAF) || F3_LEAF_UWT) st_off_wt(U, min((uint32_t)off, dump), (double)acc[mf_tile(I, J)][q]);      // consumed inside this launch: write through
                    else st_off(U, min((uint32_t)off, dump), (double)acc[mf_tile(I, J)][q]);                        // consumed by a later launch
                }
            }
        }
    }
#if F3_DONE_TS
    if (lane == 0) d.done_ts[fr.s] = wall_clock64();
#endif
    if (TREE && !LEAF) f3_publish(d.done_f + fr.s, d.epoch, lane);
    if (TREE && LEAF && lane == 0) d.done_f[fr.s] = d.epoch;        // read by the next launch only (third and later children of a front)
    F3_TS(8);
}

// ---- the bottom of the tree in one workgroup (round 4): a level-1 front and the leaves below it.  Before: the leaf launch wrote every
// leaf's update matrix to HBM (45 MB at 100k poses, 0.4 GB at 1M), a kernel boundary, and 2 048 (16 384) level-1 fronts read them back
// six at a time — 20 us of the 168 us factor phase at cfg4, 205 of 990 at cfg5.  Now workgroup b takes level-1 front b: its waves
// factorise the leaves (the leaf instance's code), add their Schur complements into the front's LDS image in list order, and the four
// waves factorise the front itself (the block form).  Same arithmetic, same order: bit-identical to the two-launch path.
template <int NTL>      // tile rows of the leaves: 3 when every leaf has <= 47 scalars, else 4
__global__ void __launch_bounds__(256, NTL == 3 ? 3 : 2) k_factor3_sub(DevGraph d, int first_pos, int count, int leaf_slot) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    if ((int)blockIdx.x >= count) return;
    f3_block_front<true, NTL>(d, first_pos + (int)blockIdx.x, FRONT_OWN, 0, smem, false, leaf_slot);
}

#ifndef F3_LEAF4_WPS
#define F3_LEAF4_WPS 3      // waves per SIMD the four-tile-row leaf instance is compiled for
#endif
template <bool TREE, bool LEAF, int NT = 4>
__global__ void __launch_bounds__(256, LEAF ? (NT == 3 ? 5 : F3_LEAF4_WPS) : 2) k_factor3(DevGraph d, int level_off, int count, int mode, int leaf_slot, int n_wave_fronts) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    // whole-tree launches: the first n_wave_fronts level positions one wave each (four per workgroup), the rest — the upper
    // levels — one workgroup each
    int bid = (int)blockIdx.x;
    if constexpr (TREE && !LEAF) {
        bid = wg_ticket(d, smem);                                    // (the flagged launch: its fronts wait for each other)
        const int wave_blocks = (n_wave_fronts + 3) >> 2;
        if (bid >= wave_blocks) { const int pos = level_off + n_wave_fronts + (bid - wave_blocks);
            f3_block_front(d, pos, mode, leaf_slot, smem, (d.dbg & 16) && pos == (d.dbg >> 8)); return; }
    }
    const int fi = bid * 4 + wave;
    if (fi >= (TREE && !LEAF ? n_wave_fronts : count)) return;      // whole wave leaves; no block barrier below
    const bool ts_on = ((d.dbg & 8) && count == (d.dbg >> 8) && fi == 0) || ((d.dbg & 16) && level_off + fi == (d.dbg >> 8));   // 16: probe the front at a level POSITION
    f3_wave_front<TREE, LEAF, NT>(d, level_off + fi, mode, leaf_slot, smem, wave, lane, ts_on, fi == 0);
}

// backward solve of variant 3's LDL^T panels (unit diagonal): x_piv = L11^-T (y - L21^T x_bnd), one wave per front
template <bool TREE>      // TREE: the front polls its boundary rows of the solution until its ancestors have written them
__device__ __forceinline__ void bs3_wave_front(const DevGraph &d, int pos, double *smem, int slot_doubles, int wave, int lane, bool ts_on) {
    F3_TS(32);
    const F3 fr = f3_load(d.f3_desc, pos, lane);
    const int npiv = fr.npiv, nbnd = fr.nbnd, f = npiv + nbnd, ldl = f + 1, lds = (f + 1) | 1;
    F3_TS(33);
    double *S = smem + (int64_t)wave * slot_doubles;
    const double *L = d.Lbuf + fr.L_off;
    const int row = (lane < nbnd) ? d.bnd_rows[fr.bnd_off + lane] : -1;
    // the L panel, column by column (coalesced), four loads in flight per lane
    const int lrow = min(lane, f);                                   // lanes beyond the panel re-read row f and drop it
    for (int c0 = 0; c0 < npiv; c0 += 32) {                          // 32 column loads in flight per lane: one round trip for npiv <= 32
        double t[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) { const int c = min(c0 + j, npiv - 1);      // rows above the diagonal are not stored: those lanes re-read the diagonal's 64-byte group
            t[j] = F3_LD_L(&L[(int64_t)c * ldl + max(lrow, min(c & ~7, f))]); }
#pragma unroll
        for (int j = 0; j < 32; ++j) if (c0 + j < npiv && lane <= f) S[(c0 + j) * lds + lane] = t[j];
    }
    F3_TS(34);
    const int me = min(lane, npiv - 1);                              // lanes >= npiv compute on a valid column and drop the result
    double w;
    if (TREE) {
        // Everything that does not depend on the parent happens BEFORE the wait: the panel is in LDS, and this lane's
        // share of it — its column of L21 (the boundary mat-vec) and of L11 (the substitution) — goes on into registers,
        // so that after the flag only the gather of x_bnd and two chains of (readlane, fma) remain.
        wave_lds_sync();
        double l21[64], lcol[32];
        const int nb1 = max(nbnd - 1, 0);
#pragma unroll
        for (int r = 0; r < 64; ++r) l21[r] = S[me * lds + npiv + min(r, nb1)];
#pragma unroll
        for (int cc = 1; cc < 32; ++cc) { lcol[cc] = S[min(min(lane, cc - 1), npiv - 1) * lds + min(cc, f)];
            lcol[cc] = lane < cc ? lcol[cc] : 0.0; }                  // the mask of the substitution step goes into the coefficient HERE, before the wait
        w = S[me * lds + f];
        double xb = 0.0;                                             // 0 beyond the boundary: those terms vanish
        if (nbnd > 0) {                                              // the boundary rows belong to the ancestors: poll the values themselves
            bool got = false, failed_already = false;
            for (int it = 0; it < (1 << 18); ++it) {
                if (row >= 0) xb = ld_off_coh(d.xe, (uint32_t)row * 8u);
                if (!__any(row >= 0 && f3_is_unset(xb))) { got = true; break; }
                if ((it & 63) == 63 && __hip_atomic_load(d.fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { failed_already = true; break; }   // some front reported: leave at once
                __builtin_amdgcn_s_sleep(F3_POLL_SLEEP);
            }
            if (!got) { if (lane == 0 && !failed_already) atomicMax(d.fail, 2); if (f3_is_unset(xb)) xb = 0.0; }     // bounded: report, carry on, drain
        }
        F3_TS(35);
#pragma unroll
        for (int r0 = 0; r0 < 64; r0 += 8) if (r0 < nbnd) {          // uniform; groups of 8: at most 7 terms beyond the boundary (their x is 0)
#pragma unroll
            for (int j = 0; j < 8; ++j) w -= l21[r0 + j] * lane_bcast(xb, r0 + j); }
        F3_TS(36);
        for (int cc = npiv - 1; cc >= 32; --cc) {                    // fronts with more than 32 pivots: the upper steps from LDS
            const double lv = S[min(lane, cc - 1) * lds + cc]; const double xcc = lane_bcast(w, cc);
            if (lane < cc) w -= lv * xcc; }
#pragma unroll
        for (int cc = 31; cc >= 1; --cc) if (cc < npiv) {            // uniform
            const double xcc = lane_bcast(w, cc); w -= lcol[cc] * xcc; }   // lanes >= cc: coefficient 0 (a select per step was half of the step's dependent chain)
    } else {
        const double xb = (row >= 0) ? d.xe[row] : 0.0;
        wave_lds_sync();
        F3_TS(35);
        w = S[me * lds + f];
        for (int r0 = 0; r0 < nbnd; r0 += 16) {                      // 16 LDS reads in flight, then the FMAs in row order
            double l16[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) l16[j] = S[me * lds + npiv + min(r0 + j, nbnd - 1)];
#pragma unroll
            for (int j = 0; j < 16; ++j) w -= ((r0 + j < nbnd) ? l16[j] : 0.0) * lane_bcast(xb, min(r0 + j, 63));
        }
        F3_TS(36);
        for (int c1 = npiv - 1; c1 > 0; c1 -= 4) {                   // the L11 entries of four steps are read ahead of the chain
            double l4[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { const int cc = max(c1 - j, 1); l4[j] = S[min(lane, cc - 1) * lds + cc]; }
#pragma unroll
            for (int j = 0; j < 4; ++j) { const int cc = c1 - j;
                if (cc > 0) { const double xcc = lane_bcast(w, cc); if (lane < cc) w -= l4[j] * xcc; } }
        }
    }
    F3_TS(37);
    if (lane < npiv) { if (TREE) st_off_wt(d.xe, (uint32_t)(fr.piv0 + lane) * 8u, w); else d.xe[fr.piv0 + lane] = w; }
#if F3_DONE_TS
    if (lane == 0) d.done_ts[d.n_fronts + fr.s] = wall_clock64();
#endif
    F3_TS(38);
}
template <bool TREE>      // TREE: one launch, root first (wave w takes level position count - 1 - w), a front waits for its parent's flag
__global__ void __launch_bounds__(256) k_backsolve3(DevGraph d, int level_off, int count, int slot_doubles) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int bid = TREE ? wg_ticket(d, smem) : (int)blockIdx.x;     // (TREE: a front polls its ancestors' values — they must belong to workgroups that have started)
    const int fi = bid * 4 + wave;
    if (fi >= count) return;
    const int pos = TREE ? level_off + (count - 1 - fi) : level_off + fi;
    const bool ts_on = ((d.dbg & 8) && count == (d.dbg >> 8) && fi == 0) || ((d.dbg & 16) && pos == (d.dbg >> 8));   // 16: probe the front at a level POSITION
    bs3_wave_front<TREE>(d, pos, smem, slot_doubles, wave, lane, ts_on);
}

// ------------------------------------------------------------------ fronts of 64 .. 159 scalars (round 3)
// A frame of the reference holds every cone within coneMappingThreshold (50 m in usecase/docker-compose.yml:16, reference
// src/slam.cpp:608): 16-24 cones in view instead of the synthetic track's 8, separators of one pose + those cones and fronts of
// 100-150 scalars.  Such a front gets a WORKGROUP and the same LDL^T on the fp64 matrix cores as the small ones: NT tile rows
// (7: f <= 111, 10: f <= 159), the NT (NT + 1) / 2 lower 16 x 16 tiles spread over the four waves' accumulators (tile t on wave
// t mod 4), panels of 4 pivots factorised redundantly by every wave (a lane owns rows lane, lane + 64, lane + 128), children
// added by source through the LDS image, the update matrix out in storage order, the same flags.  The choice is PER FRONT
// (a table maps workgroups to level positions: k_factor3_tab / k_backsolve3_tab), a small front of the same tree still runs on a
// wave.  Record formats are the small fronts' (packed update matrices, scalar records, tile-image offsets: all generic in the
// number of tile rows); a child's row table has BIG_TAB entries in f3_x when the plan holds a big front (d.f3x_stride).
static constexpr int BIG_TAB = 160;
enum { WG_WAVES = 0, WG_BLOCK4 = 1, WG_BIG7 = 2, WG_BIG10 = 3, WG_BIG5 = 4 };
__device__ __forceinline__ int f3_tile_row_any(int t) {            // row I of tile t = I (I + 1) / 2 + J, t < 55; integer compares only (t is wave-uniform: scalar code)
    return (t >= 1) + (t >= 3) + (t >= 6) + (t >= 10) + (t >= 15) + (t >= 21) + (t >= 28) + (t >= 36) + (t >= 45);
}
template <int NT> struct BigDims {
    static constexpr int NTILE = NT * (NT + 1) / 2, TPW = (NTILE + 3) / 4, ROWS = 16 * NT, IMG = NTILE * 256, NR = (ROWS + 63) / 64;
    static constexpr int PANEL = ROWS * 4;                            // one panel buffer (ROWS x 4)
    static constexpr int LDS_DOUBLES = IMG + 2 * PANEL + 4 * PANEL + BIG_TAB / 2;      // image, two spill buffers, a private buffer per wave, one child table
};
template <int NT>
__device__ __forceinline__ bool f3_big_panel(int B, bool &bad, v4d (&acc)[BigDims<NT>::TPW], double *Pn, double *Pw, double *L, int npiv, int f, int wave, int lane) {
    using D = BigDims<NT>;
    const int k0 = 4 * B, J0 = B >> 2, jc = (B & 3) * 4;             // uniform
    if (k0 >= npiv) return false;
    const int lc = lane & 15, lr = lane >> 4;
    double *Pb = Pn + (B & 1) * D::PANEL;
    // 1. the panel's four columns out of whichever waves own the tiles of tile column J0
#pragma unroll
    for (int s = 0; s < D::TPW; ++s) { const int t = 4 * s + wave, I = f3_tile_row_any(t), J = t - ((I * (I + 1)) >> 1);   // uniform per wave
        if (t < D::NTILE && J == J0 && lc >= jc && lc < jc + 4) {
#pragma unroll
            for (int q = 0; q < 4; ++q) Pb[(16 * I + lr + 4 * q) * 4 + (lc - jc)] = acc[s][q]; } }
    __syncthreads();
    // 2. every wave factorises the panel; a lane owns rows lane + 64 m.  What the rows need from each other are the entries of the
    // panel's 4 x 4 diagonal block: every lane reads it (LDS broadcast) and eliminates it alongside its own rows — the same
    // operations on the same values as a lane-to-lane broadcast of the pivot rows, without a register picked by a runtime index
    double p[D::NR][4], dd[4], Bd[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c <= r; ++c) Bd[r][c] = Pb[(k0 + r) * 4 + c];
#pragma unroll
    for (int m = 0; m < D::NR; ++m)
#pragma unroll
        for (int j = 0; j < 4; ++j) p[m][j] = (lane + 64 * m < D::ROWS) ? Pb[(lane + 64 * m) * 4 + j] : 0.0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        dd[j] = 0.0;
        if (k0 + j < npiv) {                                        // uniform
            const int col = k0 + j;
            const double piv = Bd[j][j];
            bad = bad || !(fabs(piv) > 0.0);                        // LDL^T: a ZERO pivot fails (Eigen SimplicialCholesky_impl.h:172-176), NaN is reported too
            const double inv = rcp_f64(piv);
#pragma unroll
            for (int m = 0; m < D::NR; ++m) {
                const double lj = (lane + 64 * m >= col) ? p[m][j] * inv : 0.0;      // row col itself becomes d / d = 1: a dead row in every later use
#pragma unroll
                for (int j2 = j + 1; j2 < 4; ++j2) p[m][j2] -= lj * Bd[j2][j];       // the unscaled column at the later pivot rows
                p[m][j] = lj; }
#pragma unroll
            for (int j2 = j + 1; j2 < 4; ++j2) { const double lb = Bd[j2][j] * inv;    // the block's own rows, the same way
#pragma unroll
                for (int j3 = j + 1; j3 <= j2; ++j3) Bd[j2][j3] -= lb * Bd[j3][j]; }
            dd[j] = piv;
        } else {
#pragma unroll
            for (int m = 0; m < D::NR; ++m) p[m][j] = 0.0; }
    }
    if (wave == 0) {
#pragma unroll
        for (int m = 0; m < D::NR; ++m) if (lane + 64 * m >= (k0 & ~7) && lane + 64 * m <= f) {      // (rows above the panel: zero, never read, not stored)
            double *Lc = L + (int64_t)k0 * (f + 1) + lane + 64 * m;
#pragma unroll
            for (int j = 0; j < 4; ++j) if (k0 + j < npiv) F3_ST_L(Lc + (int64_t)j * (f + 1), p[m][j]); }
    }
    // 3. trailing update of the wave's own tiles; operands through the wave's private buffer
#pragma unroll
    for (int m = 0; m < D::NR; ++m) if (lane + 64 * m < D::ROWS) {
#pragma unroll
        for (int j = 0; j < 4; ++j) Pw[(lane + 64 * m) * 4 + j] = p[m][j]; }
    const double dk = lr == 0 ? dd[0] : (lr == 1 ? dd[1] : (lr == 2 ? dd[2] : dd[3]));
    // (static tile coordinates — every wave walking all NT (NT + 1) / 2 tiles with a uniform test each, operands preloaded per tile
    // row — were measured SLOWER: 548 against 596 it/s on the 24-cones-in-view track; the code is three times the size.  So were
    // tile ROWS per wave (accumulator [r][J] = tile (w + 4 r, J), all register indices compile-time, operands read once per tile
    // row): 3.2 us per panel of a 153-scalar front against 2.3 — thirty uniform tests per loop at ~25 cycles each —, and the same
    // with one jump on the tile column instead of the tests: 4.2 us, ten copies of the loop thrash the instruction cache.  The
    // panel of a ten-tile-row front is 1.0 us of matrix cores (14 tiles per wave at 67 ns), 0.5 of pivots, 0.6 of LDS round trips.)
#pragma unroll
    for (int s = 0; s < D::TPW; ++s) { const int t = 4 * s + wave, I = f3_tile_row_any(t), J = t - ((I * (I + 1)) >> 1);
        if (t < D::NTILE && J >= J0 && 16 * I <= f) {
            const double aI = Pw[(16 * I + lc) * 4 + lr], aJ = Pw[(16 * J + lc) * 4 + lr];
            acc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(-aI, aJ * dk, acc[s], 0, 0, 0); } }
    wave_lds_sync();
    return true;
}
// mode (pose-window shards, as for the small fronts): FRONT_OWN a front of this rank's own subtree; FRONT_CONTRIB this rank's share of a
// shared front — the originals it evaluated + the update matrices of the children it owns — written to the front's exchange slot, no
// factorisation; FRONT_TOP a shared front after the all-reduce: the image starts from the summed slot, only the shared children are added
template <int NT>
__device__ __forceinline__ void f3_big_front(const DevGraph &d, int pos, double *smem, int mode = FRONT_OWN) {
    using D = BigDims<NT>;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const F3 fr = f3_load(d.f3_desc, pos, lane);
    const int npiv = fr.npiv, f = npiv + fr.nbnd;
    const bool ts_on = (d.dbg & 16) && pos == (d.dbg >> 8);          // GS_DBG = 16 | position << 8: phase timestamps of this front (scripts/big_probe.py)
#define BIG_TS(i) do { if (ts_on) { __builtin_amdgcn_s_waitcnt(0); if (tid == 0) d.dbg_ts[i] = wall_clock64(); } } while (0)
    BIG_TS(0);
    for (int r = tid; r < npiv; r += 256) d.xe[fr.piv0 + r] = f3_unset();        // this front's solution rows: not written yet
    double *img = smem, *Pn = smem + D::IMG, *Pw = Pn + 2 * D::PANEL + wave * D::PANEL;
    int32_t *tab = reinterpret_cast<int32_t *>(Pn + 6 * D::PANEL);
    StageT<false> P{img, f, (f + 1) | 1};
    for (int k = tid; k < D::IMG; k += 256) img[k] = 0.0;
    __syncthreads();
    // ---- the original values: scalar records {offset in H_arena, offset in the image} (padded to multiples of 64), landmark
    // diagonal blocks of the fused linearisation as sums of their partial slots, parallel edges one by one
    if (mode == FRONT_TOP) {                                         // the all-reduced slot: (f + 1) x f column-major, ld = f + 1
        const double *X = d.exchange + fr.x_off;
        for (int idx = tid; idx < (f + 1) * f; idx += 256) { const int c = idx / (f + 1), r = idx - c * (f + 1); if (r >= c) P.at(r, c) = X[idx]; }
        __syncthreads();
    } else
    { const int nsc = fr.sc_cnt, nlm = fr.lm_cnt;
      const int2 *sc3 = reinterpret_cast<const int2 *>(d.sc3) + fr.sc_off;
      for (int base = 0; base < nsc; base += 1024) {                 // four records per thread in flight
          int2 r4[4]; double v4[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) { const int i = base + 256 * u + tid; r4[u] = i < nsc ? sc3[i] : make_int2(0, 1); }
#pragma unroll
          for (int u = 0; u < 4; ++u) v4[u] = ld_off(d.H_arena, (uint32_t)r4[u].x * 8u);
#pragma unroll
          for (int u = 0; u < 4; ++u) if (base + 256 * u + tid < nsc) img[r4[u].y] = v4[u]; }
      // (with 24 cones in view a landmark is seen from ~30 wave tiles: eight slots' loads in flight, added in slot order)
      for (int t = tid; t < nlm; t += 256) {
          const int4 r = reinterpret_cast<const int4 *>(d.lm3)[fr.lm_off + t]; double a[5] = {0, 0, 0, 0, 0};
          for (int q0 = 0; q0 < r.x; q0 += 8) {
              double t8[8][5];
#pragma unroll
              for (int j = 0; j < 8; ++j) { const int q = min(q0 + j, r.x - 1);
                  lm_rec_load(d.lm_part, r.y + q, t8[j]); }
#pragma unroll
              for (int j = 0; j < 8; ++j)
#pragma unroll
                  for (int k = 0; k < 5; ++k) a[k] += (q0 + j < r.x) ? t8[j][k] : 0.0; }
          P.at(r.z, r.w) = a[0]; P.at(r.z + 1, r.w) = a[1]; P.at(r.z + 1, r.w + 1) = a[2]; P.at(f, r.w) = a[3]; P.at(f, r.w + 1) = a[4]; }
      __syncthreads();
      if (fr.asm_dup > 0) {
          if (tid == 0) for (int t = fr.asm_uniq; t < fr.asm_uniq + fr.asm_dup; ++t) {
              const int4 r = reinterpret_cast<const int4 *>(d.asm3)[fr.asm_off + t];
              double w[9]; asm3_load(d, r.x, r.y, w); asm3_put<true>(P, r.x, r.z, r.w, w); }
          __syncthreads(); } }
    BIG_TS(1);
    // ---- the children, by source, in list order: table to LDS, the child's flag, its packed update matrix in storage order
    bool okw = true;
    for (int e = 0; e < fr.nchild; ++e) {
        const int32_t *xt = d.f3_x + fr.x_tab + (int64_t)e * d.f3x_stride;
        const int hv = xt[BIG_TAB + (lane & 7)];
        const int c_id = __builtin_amdgcn_readlane(hv, 0), c_uoff = __builtin_amdgcn_readlane(hv, 1), c_usz = __builtin_amdgcn_readlane(hv, 2), c_own = __builtin_amdgcn_readlane(hv, 3);
        if ((mode == FRONT_CONTRIB && c_own != d.rank) || (mode == FRONT_TOP && c_own >= 0)) continue;      // (uniform) another rank's subtree / came in with the slot
        if (tid < BIG_TAB) tab[tid] = xt[tid];
        okw = f3_wait_flag(d.done_f + c_id, d.epoch, d.fail) && okw;       // a child of an earlier launch has its flag set already
        __syncthreads();
        const double *Uc = d.Uimg + c_uoff;
        for (int base = 0; base < c_usz; base += 4096) {             // sixteen loads per thread in flight
            double v16[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) { const int idx = base + 256 * u + tid; v16[u] = ld_off_coh(Uc, (uint32_t)min(idx, c_usz) * 8u); }      // beyond the matrix: the zero double behind it
#pragma unroll
            for (int u = 0; u < 16; ++u) { const int idx = base + 256 * u + tid;
                if (idx < c_usz) { const int rc = f3_rc_of(idx); img[(tab[rc & 0xff] & 0xffff) + (int)((uint32_t)tab[rc >> 8] >> 16)] += v16[u]; } }      // a child's places are distinct
        }
        __syncthreads();
    }
    if (!okw && tid == 0) atomicMax(d.fail, 2);
    BIG_TS(2);
    if (mode == FRONT_CONTRIB) {                                     // this rank's share of a shared front -> its exchange slot
        double *X = d.exchange + fr.x_off;
        for (int idx = tid; idx < (f + 1) * f; idx += 256) { const int c = idx / (f + 1), r = idx - c * (f + 1); X[idx] = r >= c ? P.at(r, c) : 0.0; }
        if (blockIdx.x == 0 && tid == 0) contrib_publish_fail(d);
        return;
    }
    // ---- accumulators: wave w holds tiles w, w + 4, ...
    v4d acc[D::TPW];
#pragma unroll
    for (int s = 0; s < D::TPW; ++s)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[s][q] = 4 * s + wave < D::NTILE ? img[(4 * s + wave) * 256 + q * 64 + lane] : 0.0;
    double *L = d.Lbuf + fr.L_off;
    bool go = true, bad = false;
#pragma clang loop unroll(disable)
    BIG_TS(3);
    for (int B = 0; B < 4 * NT && go; ++B) { go = f3_big_panel<NT>(B, bad, acc, Pn, Pw, L, npiv, f, wave, lane); if (B < 16) BIG_TS(4 + B); }
    BIG_TS(20);
    if (d.inject_iter != 0 && d.iter == d.inject_iter && pos == 0 && tid == 0) atomicMax(d.fail, d.inject_code);   // gs_debug_fail_at_iteration
    if (bad && tid == 0) atomicMax(d.fail, 1);
    // ---- Schur complement out through the image, contiguous write-through stores
    __syncthreads();
#pragma unroll
    for (int s = 0; s < D::TPW; ++s) { const int t = 4 * s + wave;
        if (t < D::NTILE) {
#pragma unroll
            for (int q = 0; q < 4; ++q) img[t * 256 + q * 64 + lane] = acc[s][q]; } }
    __syncthreads();
    { double *U = d.Uimg + fr.u_off; const int usz = fr.u_size;
      for (int idx = tid; idx < usz; idx += 256) { const int rc = f3_rc_of(idx);
          st_off_wt(U, (uint32_t)idx * 8u, img[f3_img_rowpart(npiv + (rc & 0xff)) + f3_img_colpart(npiv + (rc >> 8))]); } }
#if F3_DONE_TS
    if (tid == 0) d.done_ts[fr.s] = wall_clock64();
#endif
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (tid == 0) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __hip_atomic_store(d.done_f + fr.s, d.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    BIG_TS(21);
#undef BIG_TS
}
// LDS of the backward solve of a big front: boundary values, right-hand side, L21 (column-major as stored) and L11 strictly
// lower, packed by ROWS (row r holds its r columns at r (r - 1) / 2): f^2 / 2 doubles at most
__host__ __device__ constexpr int bs3_big_lds_doubles(int f) { return 2 * BIG_TAB + (f * f) / 2 + f + 64; }
template <bool TREE>
__device__ __forceinline__ void bs3_big_front(const DevGraph &d, int pos, double *smem) {
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const F3 fr = f3_load(d.f3_desc, pos, lane);
    const int npiv = fr.npiv, nbnd = fr.nbnd, f = npiv + nbnd, ldl = f + 1;
    double *xb = smem, *w = smem + BIG_TAB, *S21 = smem + 2 * BIG_TAB, *S11 = S21 + npiv * nbnd;
    const double *L = d.Lbuf + fr.L_off;
    // ---- everything that does not depend on the ancestors first: the panel into LDS.  Column c's rows c + 1 .. f are contiguous:
    // a lane takes rows c + 1 + lane + 64 m; four columns per wave in flight
    for (int c0 = 4 * wave; c0 < npiv; c0 += 16) {
        double v[4][3];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int m = 0; m < 3; ++m) { const int c = min(c0 + j, npiv - 1), r = c + 1 + lane + 64 * m; v[j][m] = F3_LD_L(&L[(int64_t)c * ldl + min(r, f)]); }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int m = 0; m < 3; ++m) { const int c = c0 + j, r = c + 1 + lane + 64 * m;
                if (c < npiv && r <= f) { if (r < npiv) S11[((r * (r - 1)) >> 1) + c] = v[j][m]; else if (r < f) S21[c * nbnd + (r - npiv)] = v[j][m]; else w[c] = v[j][m]; } }
    }
    // ---- the boundary rows belong to the ancestors: poll the values themselves (whole-tree launch), or read them
    bool got = true, failed_already = false;
    if (TREE) {
        for (int r0 = 0; r0 < nbnd; r0 += 256) { const int r = r0 + tid; const int row = r < nbnd ? d.bnd_rows[fr.bnd_off + r] : -1;
            double x = 0.0; bool mine = false;
            for (int it = 0; it < (1 << 18); ++it) {
                if (row >= 0) x = ld_off_coh(d.xe, (uint32_t)row * 8u);
                mine = row >= 0 && f3_is_unset(x);
                if (!__any(mine)) break;
                if ((it & 63) == 63 && __hip_atomic_load(d.fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { failed_already = true; break; }
                __builtin_amdgcn_s_sleep(F3_POLL_SLEEP); }
            if (mine) { got = false; x = 0.0; }
            if (r < nbnd) xb[r] = x; }
        if (__any(!got) && !failed_already && lane == 0) atomicMax(d.fail, 2);       // bounded: report, carry on, drain
    } else {
        for (int r = tid; r < nbnd; r += 256) xb[r] = d.xe[d.bnd_rows[fr.bnd_off + r]];
    }
    __syncthreads();
    // ---- w_c = y_c - sum_r L21[r][c] x_bnd[r]: a wave per column, lanes over the boundary rows (fixed order: lane strides, then the shuffle tree)
    for (int c = wave; c < npiv; c += 4) {
        double a = 0.0;
        for (int r = lane; r < nbnd; r += 64) a += S21[c * nbnd + r] * xb[r];
        a = wave_sum(a);
        if (lane == 0) w[c] -= a; }
    __syncthreads();
    // ---- L11^T x = w, unit diagonal, column oriented on wave 0: lane j keeps w_j, w_{j + 64}, w_{j + 128}
    if (wave == 0) {
        double wr[3];
#pragma unroll
        for (int m = 0; m < 3; ++m) wr[m] = (lane + 64 * m < npiv) ? w[lane + 64 * m] : 0.0;
        for (int cc = npiv - 1; cc >= 1; --cc) {
            const int mc = cc >> 6;                                  // uniform
            const double src = mc == 0 ? wr[0] : (mc == 1 ? wr[1] : wr[2]);
            const double xcc = lane_bcast(src, cc & 63);
            const double *row = S11 + ((cc * (cc - 1)) >> 1);
#pragma unroll
            for (int m = 0; m < 3; ++m) { const int j = lane + 64 * m; if (64 * m < cc) { const double lv = j < cc ? row[j] : 0.0; wr[m] -= lv * xcc; } }
        }
#pragma unroll
        for (int m = 0; m < 3; ++m) { const int j = lane + 64 * m;
            if (j < npiv) { if (TREE) st_off_wt(d.xe, (uint32_t)(fr.piv0 + j) * 8u, wr[m]); else d.xe[fr.piv0 + j] = wr[m]; } }
#if F3_DONE_TS
        if (lane == 0) d.done_ts[d.n_fronts + fr.s] = wall_clock64();
#endif
    }
}
// Table-driven launches (plans that hold a big front): workgroup b takes wgt[b] = {first level position, kind | count << 8} — up to
// four small fronts a wave each, a small front on four waves, or a big front.  Factor: the table follows the level positions
// (children in earlier workgroups); backward solve: the reverse (ancestors in earlier workgroups).
// CLASS: the launches are cut by LDS need, and each class is its own kernel so that the many fronts just beyond a wave do not
// inherit the registers (and with them the occupancy) of the others: 0 = fronts of 64-79 scalars (three workgroups per CU),
// 1 = small fronts (a wave or four each) and fronts of 80-111 (two per CU), 2 = fronts of 112-159 (one per CU)
template <int CLASS>
__global__ void __launch_bounds__(256, CLASS == 0 ? 3 : (CLASS == 1 ? 2 : 1)) k_factor3_tab(DevGraph d, const int2 *__restrict__ wgt, int leaf_launch_preceded, int mode) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int bid = wg_ticket(d, smem);
    const int2 e = wgt[bid];
    const int pos = __builtin_amdgcn_readfirstlane(e.x), kind = __builtin_amdgcn_readfirstlane(e.y) & 0xff, cnt = __builtin_amdgcn_readfirstlane(e.y) >> 8;
    if constexpr (CLASS == 0) { f3_big_front<5>(d, pos, smem, mode); }
    else if constexpr (CLASS == 2) { f3_big_front<10>(d, pos, smem, mode); }
    else {
        // (the host never puts a four-wave entry into a CONTRIB table: that form has the modes OWN and TOP)
        if (kind == WG_WAVES) { if (wave < cnt) f3_wave_front<true, false, 4>(d, pos + wave, mode, leaf_launch_preceded, smem, wave, lane, false, bid == 0 && wave == 0); }
        else if (kind == WG_BLOCK4) f3_block_front(d, pos, mode, leaf_launch_preceded, smem, false);
        else f3_big_front<7>(d, pos, smem, mode);
    }
}
template <bool BIG>      // BIG: a launch of big fronts only (few registers: as many workgroups per CU as the LDS allows); else small fronts, a wave each
__global__ void __launch_bounds__(256, BIG ? 4 : 2) k_backsolve3_tab(DevGraph d, const int2 *__restrict__ wgt, int slot_doubles) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int2 e = wgt[wg_ticket(d, smem)];
    const int pos = __builtin_amdgcn_readfirstlane(e.x), cnt = __builtin_amdgcn_readfirstlane(e.y) >> 8;
    if constexpr (BIG) bs3_big_front<true>(d, pos, smem);
    else { if (wave < cnt) bs3_wave_front<true>(d, pos - wave, smem, slot_doubles, wave, lane, false); }     // root first: positions downwards
}
// the kernel's copy of the handle's device view for a ticketed launch of `grid` workgroups: the counter's value when its first workgroup starts
static DevGraph ticketed(const DevGraph &d, unsigned grid) { DevGraph dd = d; if (d.tickets) d.ticket_base += grid; return dd; }
// LDS of one workgroup of a table-driven launch, by kind (a launch takes the maximum over its workgroups: the host cuts the table
// into launches of equal need, so that the many fronts just beyond a wave do not run at the occupancy of the ten-tile-row ones)
size_t factor_tab_lds_bytes(int kind) {
    switch (kind) {
        case WG_BIG5: return (size_t)BigDims<5>::LDS_DOUBLES * sizeof(double);
        case WG_BIG7: return (size_t)BigDims<7>::LDS_DOUBLES * sizeof(double);
        case WG_BIG10: return (size_t)BigDims<10>::LDS_DOUBLES * sizeof(double);
        default: return (size_t)MF_IMG * 4 * sizeof(double);
    }
}
size_t backsolve_tab_lds_bytes(int kind, int f_or_slot_f, int npiv_small) {      // waves: the slot of the largest small front; big: the front's own need
    if (kind == WG_WAVES || kind == WG_BLOCK4) { const int slot = ((((f_or_slot_f + 1) | 1) * std::max(npiv_small, 1)) + 1) & ~1; return (size_t)slot * 4 * sizeof(double); }
    return (size_t)bs3_big_lds_doubles(f_or_slot_f) * sizeof(double);
}
void launch_factor_tab(const DevGraph &d, const int2 *wgt, int n_wg, int leaf_launch_preceded, size_t lds_bytes, int cls, hipStream_t st, int mode) {
    if (n_wg <= 0) return;                                           // (a launch holds workgroups of ONE class: the host cut the table that way)
    if (cls == 0) { allow_max_lds((const void *)k_factor3_tab<0>);
        hipLaunchKernelGGL(k_factor3_tab<0>, dim3(n_wg), dim3(256), lds_bytes, st, ticketed(d, (unsigned)n_wg), wgt, leaf_launch_preceded, mode); }
    else if (cls == 2) { allow_max_lds((const void *)k_factor3_tab<2>);
        hipLaunchKernelGGL(k_factor3_tab<2>, dim3(n_wg), dim3(256), lds_bytes, st, ticketed(d, (unsigned)n_wg), wgt, leaf_launch_preceded, mode); }
    else { allow_max_lds((const void *)k_factor3_tab<1>);
        hipLaunchKernelGGL(k_factor3_tab<1>, dim3(n_wg), dim3(256), lds_bytes, st, ticketed(d, (unsigned)n_wg), wgt, leaf_launch_preceded, mode); }
}
void launch_backsolve_tab(const DevGraph &d, const int2 *wgt, int n_wg, int max_npiv_small, int max_f_small, size_t lds_bytes, int cls, hipStream_t st) {
    if (n_wg <= 0) return;
    const int slot = ((((max_f_small + 1) | 1) * std::max(max_npiv_small, 1)) + 1) & ~1;
    if (cls == 1) { allow_max_lds((const void *)k_backsolve3_tab<true>);
        hipLaunchKernelGGL(k_backsolve3_tab<true>, dim3(n_wg), dim3(256), lds_bytes, st, ticketed(d, (unsigned)n_wg), wgt, slot); }
    else { allow_max_lds((const void *)k_backsolve3_tab<false>);
        hipLaunchKernelGGL(k_backsolve3_tab<false>, dim3(n_wg), dim3(256), lds_bytes, st, ticketed(d, (unsigned)n_wg), wgt, slot); }
}

// ---- structure phase on the device: the ELL streams of the observation edges, permuted out of the insertion-order
// arrays (which travel to HBM, unprocessed, while the host still builds the plan).  ell_ins[e] = insertion index of the
// edge at ELL position e (-1: empty slot); pl_rank (pose-window shards): edges another rank evaluates stay empty here.
__global__ void __launch_bounds__(256) k_build_ell(int64_t L, const int32_t *__restrict__ ell_ins, const int32_t *__restrict__ raw_l,
        const double *__restrict__ raw_z, const double *__restrict__ raw_info, const int32_t *__restrict__ pl_rank, int rank,
        int32_t *__restrict__ ell_l, double *__restrict__ ell_z, double *__restrict__ ell_w) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= L) return;
    int k = ell_ins[e];
    if (k >= 0 && pl_rank && pl_rank[k] != rank) k = -1;
    int l = -1; double zx = 0, zy = 0, w0 = 0, w1 = 0, w2 = 0;
    if (k >= 0) { l = raw_l[k]; zx = raw_z[2 * (int64_t)k]; zy = raw_z[2 * (int64_t)k + 1];
        w0 = raw_info[3 * (int64_t)k]; w1 = raw_info[3 * (int64_t)k + 1]; w2 = raw_info[3 * (int64_t)k + 2]; }
    ell_l[e] = l; ell_z[e] = zx; ell_z[L + e] = zy; ell_w[e] = w0; ell_w[L + e] = w1; ell_w[2 * L + e] = w2;
}
void launch_build_ell(int64_t L, const int32_t *ell_ins, const int32_t *raw_l, const double *raw_z, const double *raw_info,
                      const int32_t *pl_rank, int rank, int32_t *ell_l, double *ell_z, double *ell_w, hipStream_t st) {
    if (L > 0) hipLaunchKernelGGL(k_build_ell, dim3((unsigned)((L + 255) / 256)), dim3(256), 0, st, L, ell_ins, raw_l, raw_z, raw_info, pl_rank, rank, ell_l, ell_z, ell_w);
}

// ---- structure phase on the device: the per-level-position descriptors (f3_desc: 32 ints + the first two children's
// row tables + the front's own store table) and the children tables (f3_x) of the variant-3 kernels, built from the
// compact plan arrays that are in HBM anyway (fronts, children, child maps).  One wave per level position.  On the host
// these tables were 35 MB built on threads and copied over PCIe: 9 ms of the structure phase at 100k poses.
// pack(i, col_ok): boundary row i of a front inside its packed update matrix, as byte offsets {row part i(i+1)/2 * 8 (low 16
// bits), column part i * 8 (high 16 bits)}; -30000 = none (the sum goes negative and the gather clamps to the zero slot).
__device__ __forceinline__ int32_t f3_pack(int i, bool col_ok) {
    const int ro8 = ((i * (i + 1)) >> 1) * 8, co8 = col_ok ? i * 8 : -30000;
    return (int32_t)((uint32_t)(ro8 & 0xffff) | ((uint32_t)(co8 & 0xffff) << 16));
}
__global__ void __launch_bounds__(256) k_build_f3(int nq, const int32_t *__restrict__ lf, const DevFront *__restrict__ fronts,
        const int32_t *__restrict__ children, const int32_t *__restrict__ child_map, const int32_t *__restrict__ u3_off,
        const int32_t *__restrict__ u3_size, const int32_t *__restrict__ bf /*[front][8]*/, const int32_t *__restrict__ xrow_off,
        const int64_t *__restrict__ x_off /* nullable */, int32_t *__restrict__ f3_desc, int32_t *__restrict__ f3_x, int x_stride,
        const int32_t *__restrict__ list /* nullable: the level positions to (re)build */, const int32_t *__restrict__ pos_of /* nullable: front -> level position */) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int qi = blockIdx.x * 4 + wave;
    if (qi >= nq) return;
    const int q = list ? list[qi] : qi;
    const int32_t none = (int32_t)((uint32_t)((-30000) & 0xffff) | ((uint32_t)(-30000) << 16));
    const int s = lf[q];
    const DevFront F = fronts[s];
    const int f = F.npiv + F.nbnd;
    int32_t *r = f3_desc + (int64_t)q * F3_STRIDE;
    // ---- the 32 descriptor ints: lane i computes entry i
    { int c0 = -1, c1 = -1;
      if (F.child_cnt > 0) c0 = children[F.child_off];
      if (F.child_cnt > 1) c1 = children[F.child_off + 1];
      const int64_t xo = x_off ? x_off[s] : 0;
      int v = 0;
      switch (lane) {
          case 0: v = s; break;            case 1: v = F.npiv; break;       case 2: v = F.nbnd; break;      case 3: v = F.asm_off; break;
          case 4: v = F.asm_cnt - F.asm_dup; break;                         case 5: v = F.asm_dup; break;   case 6: v = F.child_cnt; break;
          case 7: v = F.child_off; break;  case 8: v = (int32_t)(F.L_off & 0xffffffffLL); break;           case 9: v = (int32_t)(F.L_off >> 32); break;
          case 10: v = F.piv0; break;      case 11: v = (int32_t)F.bnd_off; break;
          case 12: v = c0; break;          case 13: v = c1; break;          case 14: v = xrow_off[q]; break;
          case 16: v = c0 >= 0 ? fronts[c0].owner : 0; break;               case 17: v = c1 >= 0 ? fronts[c1].owner : 0; break;
          case 18: v = (int32_t)(xo & 0xffffffffLL); break;                 case 19: v = (int32_t)(xo >> 32); break;
          case 20: v = bf[8 * s + 3]; break; case 21: v = bf[8 * s + 4]; break; case 22: v = bf[8 * s + 5]; break; case 23: v = bf[8 * s + 6]; break;
          case 24: v = u3_off[s]; break;   case 25: v = u3_size[s]; break;
          case 26: v = c0 >= 0 ? u3_off[c0] : 0; break;   case 27: v = c1 >= 0 ? u3_off[c1] : 0; break;
          case 28: v = c0 >= 0 ? u3_size[c0] : 0; break;  case 29: v = c1 >= 0 ? u3_size[c1] : 0; break;
          // whole-tree backward solve: wait for a parent of the SAME launch only (own in own, shared in shared)
          case 30: v = (F.parent >= 0 && fronts[F.parent].owner == F.owner) ? F.parent : -1; break;
          case 31: v = F.level; break;
          default: v = 0; break;
      }
      if (lane < F3_INTS) r[lane] = v; }
    // ---- the front's own store table (leaf instance): row of the front -> its place in the front's packed update matrix
    r[160 + lane] = (lane >= F.npiv && lane <= f) ? f3_pack(lane - F.npiv, lane < f) : none;
    if (F.child_cnt < 1) r[32 + lane] = 0;
    if (F.child_cnt < 2) r[96 + lane] = 0;
    // ---- children: row table + header {front, update-matrix offset, size, owner}; x_stride ints per child (72: a 64-entry table;
    // 168: a 160-entry one, plans that hold a front of more than 63 scalars), the header in its last 8 ints
    int32_t *xt = f3_x + xrow_off[q];
    for (int k = 0; k < F.child_cnt; ++k) {
        const int c = children[F.child_off + k];
        const int nbc = fronts[c].nbnd; const int64_t mo = fronts[c].map_off;
        // by source: boundary row r' of the child (r' = nbc: its rhs row) -> the place of its parent row in the parent's tile image
        for (int r0 = 0; r0 < x_stride - 8; r0 += 64) { const int rp = r0 + lane;
            int32_t v = 0;
            if (rp <= nbc) { const int R = rp < nbc ? child_map[mo + rp] : f;
                v = (int32_t)((uint32_t)f3_img_rowpart(R) | ((uint32_t)f3_img_colpart(R) << 16)); }
            if (rp < x_stride - 8) xt[k * x_stride + rp] = v;
            if (k < 2 && r0 == 0) r[32 + 64 * k + lane] = v; }
        if (lane < 8) xt[k * x_stride + (x_stride - 8) + lane] = lane == 0 ? c : (lane == 1 ? u3_off[c] : (lane == 2 ? u3_size[c] : (lane == 3 ? fronts[c].owner : (lane == 4 && pos_of ? pos_of[c] : 0))));   // {front, update-matrix offset, size, owner, level position}
    }
}
void launch_build_f3(int nq, const int32_t *lf, const DevFront *fronts, const int32_t *children, const int32_t *child_map,
                     const int32_t *u3_off, const int32_t *u3_size, const int32_t *bf, const int32_t *xrow_off, const int64_t *x_off,
                     int32_t *f3_desc, int32_t *f3_x, int x_stride, hipStream_t st, const int32_t *list, const int32_t *pos_of) {
    if (nq > 0) hipLaunchKernelGGL(k_build_f3, dim3((nq + 3) / 4), dim3(256), 0, st, nq, lf, fronts, children, child_map, u3_off, u3_size, bf, xrow_off, x_off, f3_desc, f3_x, x_stride, list, pos_of);
}
// growth: scattered rows of fronts / u3_off / u3_size / bf from one packed patch buffer (32 ints per front: front, DevFront as 20 ints,
// u3_off, u3_size, bf[8], one spare)
__global__ void __launch_bounds__(256) k_apply_front_patch(int n, const int32_t *__restrict__ patch, DevFront *fronts, int32_t *u3_off, int32_t *u3_size, int32_t *bf) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x, i = t >> 5, c = t & 31;
    if (i >= n) return;
    static_assert(sizeof(DevFront) == 80, "DevFront travels as 20 ints");
    const int32_t *r = patch + 32 * (int64_t)i; const int s = r[0]; const int32_t v = r[c];
    if (c >= 1 && c <= 20) reinterpret_cast<int32_t *>(fronts + s)[c - 1] = v;
    else if (c == 21) u3_off[s] = v;
    else if (c == 22) u3_size[s] = v;
    else if (c >= 23 && c <= 30) bf[8 * (int64_t)s + (c - 23)] = v;
}
void launch_apply_front_patch(int n, const int32_t *patch, DevFront *fronts, int32_t *u3_off, int32_t *u3_size, int32_t *bf, hipStream_t st) {
    if (n > 0) hipLaunchKernelGGL(k_apply_front_patch, dim3((n * 32 + 255) / 256), dim3(256), 0, st, n, patch, fronts, u3_off, u3_size, bf);
}
// landmark-diagonal block records of the fused linearisation: {kind 1, landmark} -> {1 | #partial slots << 8, first slot}
__global__ void __launch_bounds__(256) k_patch_asm3(int64_t n, int32_t *__restrict__ asm3, const int32_t *__restrict__ lm_grp_start) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    if (asm3[4 * t] == 1) { const int l = asm3[4 * t + 1]; const int q0 = lm_grp_start[l], q1 = lm_grp_start[l + 1];
        asm3[4 * t] = 1 | ((q1 - q0) << 8); asm3[4 * t + 1] = q0; }
}
void launch_patch_asm3(int64_t n, int32_t *asm3, const int32_t *lm_grp_start, hipStream_t st) {
    if (n > 0) hipLaunchKernelGGL(k_patch_asm3, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, asm3, lm_grp_start);
}

// ---- structure phase on the device: the scalar assembly records (sc3) and the fused landmark records (lm3) of every
// front, expanded from the block records (asm3) that are in HBM already.  One wave per front; a lane takes a block
// record, a wave prefix sum places its 5 / 6 / 9 scalars.  (On the host this expansion was 33 of the 110 ms of the
// structure phase at 100k poses, plus 67 MB over PCIe.)
__global__ void __launch_bounds__(256) k_build_sc3(const int32_t *__restrict__ bf, const int32_t *__restrict__ asm3, int32_t *__restrict__ sc3,
                                                   int32_t *__restrict__ lm3, int n_fronts, Sc3Args A, const int32_t *__restrict__ list /* nullable */) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int si = blockIdx.x * 4 + wave;
    if (si >= n_fronts) return;
    const int s = list ? list[si] : si;
    const int asm_off = bf[8 * s], n_uniq = bf[8 * s + 1], f = bf[8 * s + 2], sc_off = bf[8 * s + 3], sc_cnt = bf[8 * s + 4], lm_off = bf[8 * s + 5];
    int sbase = sc_off, lbase = lm_off;
    auto img = [](int r, int c) { const int I = r >> 4, J = c >> 4; return (((I * (I + 1)) >> 1) + J) * 256 + (r & 15) * 16 + (c & 15); };
    for (int t0 = 0; t0 < n_uniq; t0 += 64) {
        const int t = t0 + lane; const bool on = t < n_uniq;
        const int4 r = on ? reinterpret_cast<const int4 *>(asm3)[asm_off + t] : make_int4(-1, 0, 0, 0);
        const int kind = r.x & 0xff;
        const bool islm = on && kind == 1 && A.fused;
        int nsc = !on ? 0 : (kind == 0 ? 9 : (kind == 1 ? (A.fused ? 0 : 5) : (kind <= 3 ? 9 : (kind == 6 ? 5 : 6))));
        int ps = nsc, pl = islm ? 1 : 0;                                  // inclusive prefix sums over the lanes
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int a = __shfl_up(ps, o, WAVE), b = __shfl_up(pl, o, WAVE); if (lane >= o) { ps += a; pl += b; } }
        const int tot_s = __shfl(ps, 63, WAVE), tot_l = __shfl(pl, 63, WAVE);
        int q = sbase + ps - nsc;
        auto add = [&](int64_t src, int rr, int cc) { sc3[2 * (int64_t)q] = (int32_t)src; sc3[2 * (int64_t)q + 1] = img(rr, cc); ++q; };
        const int64_t src = r.y; const int r0 = r.z, c0 = r.w;
        if (on) switch (kind) {
            case 0: { const bool tl = src >= A.N;                          // sources beyond the base counts: the tail arenas (grow_plan)
                const int64_t N = tl ? A.tcapN : A.N, H = tl ? A.toff[0] - A.N : A.off[0], B = tl ? A.toff[1] - A.N : A.off[1];
                add(H + src, r0, c0); add(H + N + src, r0 + 1, c0); add(H + 2 * N + src, r0 + 2, c0);
                add(H + 3 * N + src, r0 + 1, c0 + 1); add(H + 4 * N + src, r0 + 2, c0 + 1); add(H + 5 * N + src, r0 + 2, c0 + 2);
                for (int k = 0; k < 3; ++k) add(B + k * N + src, f, c0 + k); } break;
            case 1:
                if (A.fused) { int32_t *o = lm3 + 4 * (int64_t)(lbase + pl - 1); o[0] = r.x >> 8; o[1] = r.y; o[2] = r0; o[3] = c0; }
                else { const int64_t H = A.off[5], B = A.off[6], M = A.M;
                    add(H + src, r0, c0); add(H + M + src, r0 + 1, c0); add(H + 2 * M + src, r0 + 1, c0 + 1); add(B + src, f, c0); add(B + M + src, f, c0 + 1); }
                break;
            case 6: { const int64_t H = A.toff[4] - A.M, B = A.toff[5] - A.M, M = A.tcapM;   // tail landmark: diagonal block + rhs out of the tail arena
                add(H + src, r0, c0); add(H + M + src, r0 + 1, c0); add(H + 2 * M + src, r0 + 1, c0 + 1); add(B + src, f, c0); add(B + M + src, f, c0 + 1); } break;
            case 2: case 3: { const bool tl = src >= A.Epp;
                const int64_t E = tl ? A.tcapEpp : A.Epp, H = tl ? A.toff[2] - A.Epp : A.off[2];
                for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) add(H + (kind == 2 ? 3 * a + b : 3 * b + a) * E + src, r0 + a, c0 + b); } break;
            default: { const bool tl = src >= A.L;                       // kinds 4, 5
                const int64_t L = tl ? (int64_t)A.tcapEpl : A.L, H = tl ? A.toff[3] - A.L : A.off[3];
                if (kind == 4) { for (int a = 0; a < 3; ++a) for (int b = 0; b < 2; ++b) add(H + (2 * a + b) * L + src, r0 + a, c0 + b); }
                else { for (int a = 0; a < 2; ++a) for (int b = 0; b < 3; ++b) add(H + (2 * b + a) * L + src, r0 + a, c0 + b); } } break;
        }
        sbase += tot_s; lbase += tot_l;
    }
    for (int q = sbase + lane; q < sc_off + sc_cnt; q += 64) { sc3[2 * (int64_t)q] = 0; sc3[2 * (int64_t)q + 1] = 1; }      // padding: (value 0 -> a don't-care slot)
}
void launch_build_sc3(const int32_t *bf, const int32_t *asm3, int32_t *sc3, int32_t *lm3, int n_fronts, const Sc3Args &A, hipStream_t st, const int32_t *list) {
    if (n_fronts > 0) hipLaunchKernelGGL(k_build_sc3, dim3((n_fronts + 3) / 4), dim3(256), 0, st, bf, asm3, sc3, lm3, n_fronts, A, list);
}

// whole-tree launches of variant 3 (own fronts of a single-GPU graph): the leaf instance (positions [0, n_leaf)), then — n_sub > 0 —
// the bottom subtrees (level-1 fronts at positions [sub_first, sub_first + n_sub), each with the leaves below it, which are the
// positions [n_leaf, sub_first) and get no launch of their own), then every level above in one flagged launch
void launch_factor_tree(const DevGraph &d, int n_leaf, int leaf_slot, int leaf_max_f, int count, int n_block, int sub_first, int n_sub, hipStream_t st) {
    if (count <= 0) return;
    allow_max_lds((const void *)k_factor3<true, false>); allow_max_lds((const void *)k_factor3<true, true>); allow_max_lds((const void *)k_factor3<true, true, 3>);
    // level 0 (no children) through the high-occupancy leaf instance (three tile rows when every leaf has <= 47 scalars),
    // everything above in one launch whose fronts wait on flags
    const int nt3 = d.leaf_nt3;                                      // gs_debug_options.leaf_nt3
    if (n_leaf > 0 && leaf_max_f <= 47 && nt3) hipLaunchKernelGGL((k_factor3<true, true, 3>), dim3((n_leaf + 3) / 4), dim3(256), (size_t)leaf_slot * 4 * sizeof(double), st, d, 0, n_leaf, FRONT_OWN, leaf_slot, 0);
    else if (n_leaf > 0) hipLaunchKernelGGL((k_factor3<true, true>), dim3((n_leaf + 3) / 4), dim3(256), (size_t)leaf_slot * 4 * sizeof(double), st, d, 0, n_leaf, FRONT_OWN, leaf_slot, 0);
    int first = n_leaf, plain_level = n_leaf > 0 ? 1 : 0;
    if (n_sub > 0) { allow_max_lds((const void *)k_factor3_sub<3>); allow_max_lds((const void *)k_factor3_sub<4>);
        if (leaf_max_f <= 47 && nt3) hipLaunchKernelGGL(k_factor3_sub<3>, dim3(n_sub), dim3(256), factor_sub_lds_bytes(leaf_slot), st, d, sub_first, n_sub, leaf_slot);
        else hipLaunchKernelGGL(k_factor3_sub<4>, dim3(n_sub), dim3(256), factor_sub_lds_bytes(leaf_slot), st, d, sub_first, n_sub, leaf_slot);
        first = sub_first + n_sub; plain_level = 2; }
    // the last n_block level positions (whole upper levels) get a workgroup each, the others a wave each
    if (count > first) { const int nw = count - first - n_block; const unsigned grid = (unsigned)((nw + 3) / 4 + n_block);
        hipLaunchKernelGGL((k_factor3<true, false>), dim3(grid), dim3(256), (size_t)MF_IMG * 4 * sizeof(double), st, ticketed(d, grid), first, count - first, FRONT_OWN, plain_level, nw); }   // leaf_slot argument: the highest level whose fronts have all their children in earlier launches
}
size_t factor_sub_lds_bytes(int leaf_slot) { return (size_t)(MF_IMG + std::max(1536, 4 * leaf_slot)) * sizeof(double); }
// the shared top of a sharded graph (mode TOP: fronts start from the all-reduced exchange slots and gather their shared
// children only): one flagged launch as well
void launch_factor_tree_top(const DevGraph &d, int first, int count, hipStream_t st) {
    if (count <= 0) return;
    allow_max_lds((const void *)k_factor3<true, false>);
    // every shared front four waves (the chain of the top levels is all there is to this launch): n_wave_fronts = 0
    hipLaunchKernelGGL((k_factor3<true, false>), dim3(count), dim3(256), (size_t)MF_IMG * 4 * sizeof(double), st, ticketed(d, (unsigned)count), first, count, FRONT_TOP, 0, 0);
}
void launch_backsolve_tree(const DevGraph &d, int first, int count, int max_npiv, int max_f, hipStream_t st) {
    if (count <= 0) return;                                          // positions [first, first + count), root first
    const int slot = ((((max_f + 1) | 1) * max_npiv) + 1) & ~1;
    allow_max_lds((const void *)k_backsolve3<true>);
    hipLaunchKernelGGL(k_backsolve3<true>, dim3((count + 3) / 4), dim3(256), (size_t)slot * 4 * sizeof(double), st, ticketed(d, (unsigned)((count + 3) / 4)), first, count, slot);
}

void launch_factor_level(const DevGraph &d, int level_off, int count, int max_f, int mode, hipStream_t st) {
    if (count <= 0) return;
    if (max_f <= 63 && d.factor_variant == 3) {
        allow_max_lds((const void *)k_factor3<false, false>);
        const size_t lds3 = std::max((size_t)MF_IMG * 4 * sizeof(double), (size_t)d.f3_lds_kb * 1024);   // gs_debug_options.f3_lds_kb: occupancy experiments (more LDS per block = fewer resident blocks)
        hipLaunchKernelGGL((k_factor3<false, false>), dim3((count + 3) / 4), dim3(256), lds3, st, d, level_off, count, mode, 0, count);
        return;
    }
    int64_t bytes = (int64_t)((max_f + 1) | 1) * max_f * 8;
    if (bytes <= LDS_LIMIT_BYTES) {
        allow_max_lds((const void *)k_factor_level<true>);
        hipLaunchKernelGGL(k_factor_level<true>, dim3(count), dim3(256), (size_t)bytes, st, d, level_off, mode);
    } else {
        hipLaunchKernelGGL(k_factor_level<false>, dim3(count), dim3(256), 0, st, d, level_off, mode);
    }
}

// ------------------------------------------------------------------ A8 backward solve
// x_piv = L11^-T (y1 - L21^T x_bnd), fronts of one level in parallel, root level first.
__global__ void __launch_bounds__(256) k_backsolve_level(DevGraph d, int level_off) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int s = d.level_fronts[level_off + blockIdx.x];
    const DevFront fr = d.fronts[s];
    const int npiv = fr.npiv, nbnd = fr.nbnd, f = npiv + nbnd, ldl = f + 1;
    double *xb = smem, *w = smem + nbnd;
    const double *L = d.Lbuf + fr.L_off;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int r = tid; r < nbnd; r += 256) xb[r] = d.xe[d.bnd_rows[fr.bnd_off + r]];
    __syncthreads();
    for (int c = wave; c < npiv; c += 4) {
        const double *col = L + (int64_t)c * ldl;
        double acc = 0.0;
        for (int r = lane; r < nbnd; r += 64) acc += col[npiv + r] * xb[r];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, WAVE);
        if (lane == 0) w[c] = col[f] - acc;
    }
    __syncthreads();
    if (wave == 0) {
        for (int c = npiv - 1; c >= 0; --c) {
            const double *col = L + (int64_t)c * ldl;
            double acc = 0.0;
            for (int r = c + 1 + lane; r < npiv; r += 64) acc += col[r] * w[r];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, WAVE);
            if (lane == 0) w[c] = (w[c] - acc) / col[c];
            __builtin_amdgcn_wave_barrier();
            __threadfence_block();
        }
        for (int c = lane; c < npiv; c += 64) d.xe[fr.piv0 + c] = w[c];
    }
}

// one WAVE per front, no reductions: the L panel is staged coalesced into LDS (odd leading dimension), lane c owns
// pivot column c:  w_c = y1_c - sum_r L21[r][c] x_bnd[r]  (x_bnd broadcast by readlane), then the column-oriented
// backward substitution  x_cc = w_cc / L11[cc][cc];  w_j -= L11[cc][j] x_cc  (j < cc).
__global__ void __launch_bounds__(256) k_backsolve_wave(DevGraph d, int level_off, int count, int slot_doubles) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    // the wave id is wave-uniform: say so (readfirstlane), otherwise everything derived from it (front id, f, loop
    // bounds, readlane sources) is treated as divergent and every readlane becomes a waterfall loop
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int fi = blockIdx.x * 4 + wave;
    if (fi >= count) return;
    const int s = d.level_fronts[level_off + fi];
    const DevFront fr = d.fronts[s];
    const int npiv = fr.npiv, nbnd = fr.nbnd, f = npiv + nbnd, ldl = f + 1, lds = (f + 1) | 1;
    double *S = smem + (int64_t)wave * slot_doubles;
    const double *L = d.Lbuf + fr.L_off;
    double xb = 0.0;
    if (lane < nbnd) xb = d.xe[d.bnd_rows[fr.bnd_off + lane]];
    for (int c = 0; c < npiv; ++c) if (lane <= f) S[c * lds + lane] = L[(int64_t)c * ldl + lane];
    wave_lds_sync();
    double w = (lane < npiv) ? S[lane * lds + f] : 0.0;
    for (int r = 0; r < nbnd; ++r) {
        const double xr = lane_bcast(xb, r);
        if (lane < npiv) w -= S[lane * lds + npiv + r] * xr;
    }
    for (int cc = npiv - 1; cc >= 0; --cc) {
        const double xcc = lane_bcast(w, cc) / S[cc * lds + cc];
        if (lane == cc) w = xcc;
        if (lane < cc) w -= S[lane * lds + cc] * xcc;
    }
    if (lane < npiv) d.xe[fr.piv0 + lane] = w;
}

void launch_backsolve_level(const DevGraph &d, int level_off, int count, int max_npiv, int max_nbnd, hipStream_t st) {
    if (count <= 0) return;
    if (d.factor_variant == 3) {                                     // LDL^T panels: unit-diagonal backward solve
        const int f = max_npiv + max_nbnd, slot = ((((f + 1) | 1) * max_npiv) + 1) & ~1;
        allow_max_lds((const void *)k_backsolve3<false>);
        hipLaunchKernelGGL(k_backsolve3<false>, dim3((count + 3) / 4), dim3(256), (size_t)slot * 4 * sizeof(double), st, d, level_off, count, slot);
        return;
    }
    if (max_npiv + max_nbnd <= 63) {
        const int f = max_npiv + max_nbnd, slot = ((((f + 1) | 1) * max_npiv) + 1) & ~1;
        allow_max_lds((const void *)k_backsolve_wave);
        hipLaunchKernelGGL(k_backsolve_wave, dim3((count + 3) / 4), dim3(256), (size_t)slot * 4 * sizeof(double), st, d, level_off, count, slot);
        return;
    }
    size_t bytes = (size_t)(max_npiv + max_nbnd + 2) * 8;
    hipLaunchKernelGGL(k_backsolve_level, dim3(count), dim3(256), bytes, st, d, level_off);
}

// cos/sin of every pose angle, refreshed whenever the host hands over new estimates; inside the iteration k_update
// keeps it current, so the linearisation pass reads 16 bytes per pose instead of evaluating sincos per lane
__global__ void __launch_bounds__(256) k_pose_trig(int n, const double *__restrict__ pose_est, double *__restrict__ pose_cs) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    double sn, cs; sincos(pose_est[3 * t + 2], &sn, &cs);
    pose_cs[2 * t] = cs; pose_cs[2 * t + 1] = sn;
}
void launch_pose_trig(const DevGraph &d, hipStream_t st) {
    const int n = d.N + d.tN;
    if (n > 0) hipLaunchKernelGGL(k_pose_trig, dim3((n + 255) / 256), dim3(256), 0, st, n, d.pose_est, d.pose_cs);
}
void launch_pose_trig_range(const DevGraph &d, int first, int count, hipStream_t st) {
    if (count > 0) hipLaunchKernelGGL(k_pose_trig, dim3((count + 255) / 256), dim3(256), 0, st, count, d.pose_est + 3 * (int64_t)first, d.pose_cs + 2 * (int64_t)first);
}

// ------------------------------------------------------------------ A9
__global__ void __launch_bounds__(256) k_update(DevGraph d) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    // g2o: when the linear solve fails, OptimizationAlgorithmGaussNewton::solve returns Fail BEFORE update() and
    // SparseOptimizer::optimize leaves its loop: the vertices keep the previous iterate (call site reference
    // src/slam.cpp:481).  The iterations of a gs_optimize call are all enqueued up front, so the rule lives here: once
    // the failure flag is up (zero pivot, a whole-tree launch that gave up on a flag, or — pose-window shards — a
    // failure any rank reported with its contribution) no update is applied any more.  fail[1] counts applied updates.
    const bool peer_numeric = d.xfail_off >= 0 && d.exchange[d.xfail_off] != 0.0, peer_timeout = d.xfail_off >= 0 && d.exchange[d.xfail_off + 1] != 0.0;
    const bool peer_failed = peer_numeric || peer_timeout;
    // fail[2]: the iteration in which gs_optimize_until's stop rule fired (0: not yet).  Only an EARLIER iteration's verdict
    // gates this launch: the block that evaluates the rule below writes it while other blocks of the same launch may still
    // be reading — with the iteration number in the flag they all see "not before this iteration" whatever the timing.
    const int conv_it = d.fail[2];
    const bool stop = d.fail[0] != 0 || peer_failed || (conv_it != 0 && conv_it < d.iter);
    if (t == 0) { if (stop) { if (peer_failed) atomicCAS(d.fail, 0, peer_numeric ? 3 : 4); }      // 3: a zero pivot on another rank, 4: a flag timeout on another rank      // (3 only where nothing failed locally: the rank a flag timeout happened on must keep its 2 — it is the one that has to fall back to one launch per level;
                                                                                 // with atomicMax every rank, the origin included, read "another rank failed" and the timeout came back with every later call)
                  else atomicAdd(d.fail + 1, 1); }
    // pose-window shards: a rank only tracks the vertices of its own subtrees and of the shared top.
    // One thread per SCALAR of the estimates (3 N pose scalars, then 2 M landmark scalars): the estimate, the increment record and
    // the solution vector are read and written as fully coalesced 8-byte streams (a thread per vertex touched them with a 24-byte
    // stride: 1.0 TB/s at 1M poses); the lane that holds a pose's angle normalises it and refreshes the pose's cos / sin.
    // (the scalars of a grown plan's tail poses follow the landmarks': pose scalar index ps = t - 2 M)
    const int ps = t < 3 * d.N ? t : ((t >= 3 * d.N + 2 * d.M && t < 3 * (d.N + d.tN) + 2 * d.M) ? t - 2 * d.M : -1);
    if (stop) { }
    else if (ps >= 0) {
        const int p = ps / 3, c = ps - 3 * p;
        const int g = (p >= d.N || d.pose_known[p]) ? d.pose_gidx[p] : -1;
        double dx = 0.0;
        if (g >= 0) { dx = d.xe[g + c];
            double v = d.pose_est[ps] + dx;
            if (c == 2) { v = normalize_theta(v); double sn, cs; sincos(v, &sn, &cs); reinterpret_cast<double2 *>(d.pose_cs)[p] = make_double2(cs, sn); }
            d.pose_est[ps] = v; }
        d.dpose[ps] = dx;
    } else if (t < 3 * d.N + 2 * d.M || (t >= 3 * (d.N + d.tN) + 2 * d.M && t < 3 * (d.N + d.tN) + 2 * (d.M + d.tM))) {
        const int u = t < 3 * d.N + 2 * d.M ? t - 3 * d.N : t - 3 * (d.N + d.tN), l = u >> 1, c = u & 1;     // (tail landmarks' scalars come last: u = 2 M + ...)
        const int g = (l >= d.M || d.lm_known[l]) ? d.lm_gidx[l] : -1;
        double dx = 0.0;
        if (g >= 0) { dx = d.xe[g + c]; d.lm_est[u] += dx; }
        d.dlm[u] = dx;
    }
    // fused linearisation leaves one chi2 partial per wave tile: total them here (fixed order), no extra launch.  By the FIRST workgroup
    // (dispatched first: its chain of loads -> reduction runs beside the others; the last one, rounds 1-3, put it behind everything else)
    if (blockIdx.x == 0) {
        __shared__ double red[8];
        double tot = 0.0;
        if (d.n_wtiles > 0) {
            // (sixteen loads in flight per thread, added in the same order as one at a time: a thread's chain of dependent
            // load -> add was the whole duration of this kernel — 80 us at 1M poses, 7 of its 10 us at 100k)
            double s = 0.0;
            const int nsum = d.n_wtiles + (d.tN > 0 ? 1 : 0);      // a grown plan's tail leaves one more partial (k_linearize_tail)
            for (int k0 = threadIdx.x; k0 < nsum; k0 += 256 * 16) {
                double v[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) { const int k = k0 + 256 * j; v[j] = k < nsum ? d.chi2_partial[k] : 0.0; }
#pragma unroll
                for (int j = 0; j < 16; ++j) if (k0 + 256 * j < nsum) s += v[j]; }
            tot = block_sum(s, red);
            if (threadIdx.x == 0) d.chi2[0] = tot;
        } else if (threadIdx.x == 0) tot = d.chi2[0];              // gather kernels: k_reduce_chi2 totalled it already
        if (threadIdx.x == 0 && d.hist_slot >= 0) d.chi2[1 + d.hist_slot] = tot;   // gs_optimize's chi2 history (a copy kernel per iteration before)
        // stop rule of gs_optimize_until: this iteration's update has been applied; if the chi2 of its linearisation point
        // is within conv_tol (relative) of the previous one, no later update is applied
        if (threadIdx.x == 0 && d.conv_tol >= 0.0 && !stop) {
            const double prev = d.chi2[70];
            if (prev >= 0.0 && fabs(prev - tot) <= d.conv_tol * tot) d.fail[2] = d.iter;
            d.chi2[70] = tot;
        }
    }
}
void launch_update(const DevGraph &d, hipStream_t st) {
    int n = 3 * (d.N + d.tN) + 2 * (d.M + d.tM);
    if (n > 0) hipLaunchKernelGGL(k_update, dim3((n + 255) / 256), dim3(256), 0, st, d);
}

void launch_polar_to_xy(int n, const double *az, const double *zen, const double *dist, double lidar, double *out, hipStream_t st) {
    if (n > 0) hipLaunchKernelGGL(k_polar_to_xy, dim3((n + 255) / 256), dim3(256), 0, st, n, az, zen, dist, lidar, out);
}
void launch_cone_to_global(int n, const double *poses, const int32_t *pose_of_obs, const double *obs, double lidar,
                           double *out, hipStream_t st) {
    if (n > 0) hipLaunchKernelGGL(k_cone_to_global, dim3((n + 255) / 256), dim3(256), 0, st, n, poses, pose_of_obs, obs, lidar, out);
}
void launch_associate(int n, const double *poses, const int32_t *pose_of_obs, const double *obs, double lidar,
                      int n_map, const double *map_xy, const int32_t *map_type, double thr, double type_tol,
                      int32_t *out, hipStream_t st) {
    if (n > 0) hipLaunchKernelGGL(k_associate, dim3((n + 255) / 256), dim3(256), 0, st, n, poses, pose_of_obs, obs, lidar,
                                  n_map, map_xy, map_type, thr, type_tol, out);
}

}  // namespace gs
