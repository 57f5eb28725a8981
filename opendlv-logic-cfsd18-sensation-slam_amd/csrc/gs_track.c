/*
 * gs_track.c — deterministic synthetic cone-track generator (build-defined; the reference has
 * no generator, recorded data or fixtures: SURVEY.md §4, §8d).
 *
 * Produces what the reference's microservice would have received on a closed Formula-Student
 * style track, in the layouts the reference uses:
 *   - per keyframe an odometry pose (x, y, heading) as latched in Slam::m_odometryData
 *     (reference src/slam.cpp:173-182, 207-209)
 *   - per keyframe K = 8 cone observations as columns (azimuth deg, zenith deg, distance m, type)
 *     of the 4 x K collector matrix (reference src/slam.cpp:83-84,108,136)
 * plus ground truth (poses, cone positions, true cone id per observation) for RMSE and for
 * building bench graphs without running the sequential association.
 *
 * Track: a stadium (two straights, two half circles) of centre-line length L = 2.5 m * M,
 * cone pairs every 5 m at +-1.5 m lateral (left = blue, type 2; right = yellow, type 1; the first
 * two pairs big orange, type 4; colour codes as in reference viewerbuild/src/drawer.cpp:22-37).
 * N poses equally spaced over one lap starting 2.5 m before pair 0, so the last poses see pairs
 * 0..3 again (loop closure, reference src/slam.cpp:697-706).  Each pose sees the 4 pairs within
 * (0, 20] m of arc length ahead.  The LiDAR sits 1.5 m ahead of the centre of gravity
 * (reference src/slam.cpp:514), observations are polar in the LiDAR frame.
 *
 * Noise: odometry = truth + independent N(0, 0.05 m) on x, y and N(0, 0.005 rad) on the heading per
 * keyframe.  The reference's "odometry" is the absolute UKF/GPS geolocation latched per frame
 * (reference src/slam.cpp:186-210), i.e. a bounded, essentially uncorrelated fix, not dead reckoning.
 * (A drifting or strongly time-correlated error excites the very soft bending modes of a pure
 * relative-measurement graph: the reference's undamped Gauss-Newton then takes kilometre-sized steps
 * on the 25 km lap and even two exact CPU factorisations of the same system drift 3e-7 apart, which
 * makes a 1e-6 parity bar meaningless.  gs_track_generate_ex exposes the noise model for such studies.)
 * Observations: azimuth += N(0, 0.5 deg), distance += N(0, 0.05 m), zenith = 0; an azimuth of
 * exactly 0 is re-drawn (reference quirk: sign = angle/fabs(angle) is NaN at 0, src/slam.cpp:515).
 * RNG: MT19937-64 + Box-Muller, seeds 18 (reserved for track shape), 19 (odometry), 20 (observations).
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define GS_TRACK_K 8

/* ---- MT19937-64 (Matsumoto & Nishimura reference algorithm, restated) ---- */
typedef struct { uint64_t mt[312]; int idx; int have_spare; double spare; } gs_rng;

static void rng_seed(gs_rng *r, uint64_t seed) {
    r->mt[0] = seed;
    for (int i = 1; i < 312; ++i)
        r->mt[i] = 6364136223846793005ULL * (r->mt[i - 1] ^ (r->mt[i - 1] >> 62)) + (uint64_t)i;
    r->idx = 312; r->have_spare = 0; r->spare = 0.0;
}
static uint64_t rng_u64(gs_rng *r) {
    if (r->idx >= 312) {
        for (int i = 0; i < 312; ++i) {
            uint64_t x = (r->mt[i] & 0xFFFFFFFF80000000ULL) | (r->mt[(i + 1) % 312] & 0x7FFFFFFFULL);
            uint64_t xa = x >> 1;
            if (x & 1ULL) xa ^= 0xB5026F5AA96619E9ULL;
            r->mt[i] = r->mt[(i + 156) % 312] ^ xa;
        }
        r->idx = 0;
    }
    uint64_t y = r->mt[r->idx++];
    y ^= (y >> 29) & 0x5555555555555555ULL;
    y ^= (y << 17) & 0x71D67FFFEDA60000ULL;
    y ^= (y << 37) & 0xFFF7EEE000000000ULL;
    y ^= (y >> 43);
    return y;
}
static double rng_uniform(gs_rng *r) { /* (0,1) with 53 bits */
    return ((double)(rng_u64(r) >> 11) + 0.5) * (1.0 / 9007199254740992.0);
}
static double rng_normal(gs_rng *r) {
    if (r->have_spare) { r->have_spare = 0; return r->spare; }
    double u1 = rng_uniform(r), u2 = rng_uniform(r);
    double m = sqrt(-2.0 * log(u1));
    r->spare = m * sin(6.283185307179586476925 * u2);
    r->have_spare = 1;
    return m * cos(6.283185307179586476925 * u2);
}

/* ---- stadium centre line, exact arc-length parameterisation ---- */
typedef struct { double L, R, S; } stadium;

static void stadium_init(stadium *t, double L) {
    double R = L / (4.0 * M_PI);
    if (R < 9.0) R = 9.0;
    t->L = L; t->R = R; t->S = 0.5 * (L - 2.0 * M_PI * R);
}
/* position and heading at arc length s (any real, wrapped) */
static void stadium_at(const stadium *t, double s, double *x, double *y, double *th) {
    double L = t->L, R = t->R, S = t->S;
    s = fmod(s, L); if (s < 0) s += L;
    if (s < S) { *x = s; *y = 0.0; *th = 0.0; return; }
    s -= S;
    if (s < M_PI * R) { double a = s / R; *x = S + R * sin(a); *y = R - R * cos(a); *th = a; return; }
    s -= M_PI * R;
    if (s < S) { *x = S - s; *y = 2.0 * R; *th = M_PI; return; }
    s -= S;
    { double a = s / R; *x = -R * sin(a); *y = R + R * cos(a); *th = M_PI + a; }
}
static double wrap_pi(double a) {
    while (a >= M_PI) a -= 2.0 * M_PI;
    while (a < -M_PI) a += 2.0 * M_PI;
    return a;
}

/*
 * Generate one lap.
 *   n_poses  N, n_cones M (even, >= 24)
 * outputs (caller-allocated):
 *   truth_poses [N*3], odom_poses [N*3]
 *   cone_xy [M*2] ground truth, cone_type [M]     (cone id = 2*pair + side, side 0 = left)
 *   obs [N*K*4]   per pose K columns of (az deg, zen deg, dist m, type)   -- collector layout
 *   obs_cone [N*K] ground-truth cone id of every observation
 * returns 0, or -1 on bad arguments.
 */
static int track_generate_k(int32_t n_poses, int32_t n_cones, int32_t K, const double *noise /* [5] sig_xy sig_th revert sig_az_deg sig_d */,
                            double *truth_poses, double *odom_poses,
                            double *cone_xy, int32_t *cone_type,
                            double *obs, int32_t *obs_cone)
{
    /* K observations per pose = the K / 2 pairs within (0, 2.5 K] m ahead: K = 8 is the 20 m view of SURVEY 8d; K = 16 / 24 are
     * what the reference's coneMappingThreshold of 50 m (usecase/docker-compose.yml:16, src/slam.cpp:608) lets a frame hold */
    if (n_poses < 3 || n_cones < 24 || (n_cones & 1) || K < 2 || (K & 1) || K > 64 || 3 * K > n_cones) return -1;
    const int N = n_poses, M = n_cones, P = M / 2, Q = K / 2;
    const double L = 2.5 * (double)M;
    const double lidar = 1.5, half_width = 1.5, view = 5.0 * (double)Q, spacing = 5.0;
    stadium t; stadium_init(&t, L);
    (void)spacing;

    for (int p = 0; p < P; ++p) {
        double x, y, th; stadium_at(&t, 5.0 * p, &x, &y, &th);
        double nx = -sin(th), ny = cos(th);           /* left normal */
        cone_xy[(2 * p) * 2 + 0] = x + half_width * nx; cone_xy[(2 * p) * 2 + 1] = y + half_width * ny;
        cone_xy[(2 * p + 1) * 2 + 0] = x - half_width * nx; cone_xy[(2 * p + 1) * 2 + 1] = y - half_width * ny;
        cone_type[2 * p] = (p < 2) ? 4 : 2;
        cone_type[2 * p + 1] = (p < 2) ? 4 : 1;
    }

    gs_rng r_odo, r_obs; rng_seed(&r_odo, 19); rng_seed(&r_obs, 20);
    const double sig_xy = noise[0], sig_th = noise[1], revert = noise[2];
    const double sig_az = noise[3], sig_d = noise[4];
    const double ds = L / (double)N;
    double dx = 0, dy = 0, dth = 0;                   /* accumulated odometry drift */

    for (int k = 0; k < N; ++k) {
        double s = -2.5 + ds * k;
        double x, y, th; stadium_at(&t, s, &x, &y, &th);
        th = wrap_pi(th);
        truth_poses[3 * k + 0] = x; truth_poses[3 * k + 1] = y; truth_poses[3 * k + 2] = th;
        if (k > 0) { dx = revert * dx + sig_xy * rng_normal(&r_odo); dy = revert * dy + sig_xy * rng_normal(&r_odo);
                     dth = revert * dth + sig_th * rng_normal(&r_odo); }
        odom_poses[3 * k + 0] = x + dx; odom_poses[3 * k + 1] = y + dy; odom_poses[3 * k + 2] = wrap_pi(th + dth);

        /* first pair strictly ahead: smallest p with 5p > s (mod L) */
        double sm = fmod(s, L); if (sm < 0) sm += L;
        int p0 = (int)floor(sm / 5.0) + 1;
        double c = cos(th), sn = sin(th);
        double lx = x + lidar * c, ly = y + lidar * sn;
        int col = 0;
        for (int q = 0; q < Q; ++q) {
            int p = (p0 + q) % P;
            double ahead = 5.0 * (p0 + q) - sm;
            if (!(ahead > 0.0 && ahead <= view + 1e-9)) continue;   /* cannot happen with 5 m spacing */
            for (int side = 0; side < 2; ++side) {
                int id = 2 * p + side;
                double gx = cone_xy[2 * id] - lx, gy = cone_xy[2 * id + 1] - ly;
                double vx = c * gx + sn * gy, vy = -sn * gx + c * gy;       /* LiDAR frame */
                double az = atan2(vy, vx) * (180.0 / M_PI);
                double d = sqrt(vx * vx + vy * vy);
                double azn;
                do { azn = az + sig_az * rng_normal(&r_obs); } while (azn == 0.0);
                double dn = d + sig_d * rng_normal(&r_obs);
                if (dn < 0.05) dn = 0.05;
                double *o = obs + ((size_t)k * K + col) * 4;
                o[0] = azn; o[1] = 0.0; o[2] = dn; o[3] = (double)cone_type[id];
                obs_cone[(size_t)k * K + col] = id;
                ++col;
            }
        }
        for (; col < K; ++col) {                       /* never reached for valid tracks */
            double *o = obs + ((size_t)k * K + col) * 4;
            o[0] = 1.0; o[1] = 0.0; o[2] = 1e6; o[3] = 0.0;
            obs_cone[(size_t)k * K + col] = -1;
        }
    }
    return 0;
}

int gs_track_generate_ex(int32_t n_poses, int32_t n_cones, const double *noise, double *truth_poses, double *odom_poses,
                         double *cone_xy, int32_t *cone_type, double *obs, int32_t *obs_cone)
{
    return track_generate_k(n_poses, n_cones, GS_TRACK_K, noise, truth_poses, odom_poses, cone_xy, cone_type, obs, obs_cone);
}

/* the same lap with K observations per pose (obs [N*K*4], obs_cone [N*K]) */
int gs_track_generate_k(int32_t n_poses, int32_t n_cones, int32_t K, double *truth_poses, double *odom_poses,
                        double *cone_xy, int32_t *cone_type, double *obs, int32_t *obs_cone)
{
    const double noise[5] = {0.05, 0.005, 0.0, 0.5, 0.05};
    return track_generate_k(n_poses, n_cones, K, noise, truth_poses, odom_poses, cone_xy, cone_type, obs, obs_cone);
}

int gs_track_generate(int32_t n_poses, int32_t n_cones,
                      double *truth_poses, double *odom_poses,
                      double *cone_xy, int32_t *cone_type,
                      double *obs, int32_t *obs_cone)
{
    const double noise[5] = {0.05, 0.005, 0.0, 0.5, 0.05};
    return gs_track_generate_ex(n_poses, n_cones, noise, truth_poses, odom_poses, cone_xy, cone_type, obs, obs_cone);
}

int gs_track_obs_per_pose(void) { return GS_TRACK_K; }
