// A C++ consumer of the sharded entry points (tests/test_gpu_parity.py::test_cpp_consumer_runs_the_sharded_optimize_through_rccl):
// what the microservice — a C++ process, reference src/opendlv-logic-cfsd18-sensation-slam.cpp:49-119 — does to run its optimiser over
// pose windows, with nothing but the C-ABI: no Python, no torch, no HIP header.  Plain g++ -std=c++14 against include/graphslam.h.
//   1. build the graph the way Slam does (src/slam.cpp:433-459, 525-550): pose vertices with their odometry edges, cone vertices,
//      observation edges (measurements from the library's own polar -> XY);
//   2. handle A: gs_optimize(10)                                  (src/slam.cpp:480-481 on one GPU)
//   3. handle B: gs_dist_configure(0, 1) + a shared top forced on the single rank (include/graphslam_debug.h: the only way to put a
//      NON-EMPTY exchange buffer through RCCL on one GPU), gs_dist_unique_id -> gs_dist_comm_init -> gs_dist_optimize(10): local
//      half -> ncclAllReduce(sum, fp64) -> shared top, all enqueued by the library;
//   4. both must agree (1e-9 relative to the track's size) and report 10 iterations;
//   5. two more handles as the replicas of a world of 2 with RANK-LOCAL INGESTION: every vertex and odometry edge on both, a keyframe's observation
//      edges only on the replica that owns its pose (and, on both, those of the second window's first pose and of the fixed poses), the landmark
//      windows from gs_dist_local_landmark_windows OR-ed over the two and handed back with gs_dist_set_landmark_windows; 10 iterations through
//      the two halves with the exchange buffers summed on the host; the merged poses against handle A's.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../include/graphslam.h"
#include "../include/graphslam_debug.h"

extern "C" int gs_track_generate(int32_t, int32_t, double *, double *, double *, int32_t *, double *, int32_t *);
extern "C" int gs_track_obs_per_pose(void);

#define CHECK(call) do { int rc_ = (call); if (rc_ < 0) { std::fprintf(stderr, "%s -> %d: %s\n", #call, rc_, gs_last_error()); return 1; } } while (0)

static void se2_between(const double *a, const double *b, double *z) {      // z = a^-1 o b  (src/slam.cpp:451-454)
    const double c = std::cos(a[2]), s = std::sin(a[2]), dx = b[0] - a[0], dy = b[1] - a[1];
    z[0] = c * dx + s * dy; z[1] = -s * dx + c * dy; z[2] = std::atan2(std::sin(b[2] - a[2]), std::cos(b[2] - a[2]));
}

// rank >= 0: rank-local ingestion — the observation edges of this replica's poses, of the windows' first poses and of the fixed poses only
static int build(gs_graph *g, int N, int M, int K, const std::vector<double> &odom, const std::vector<double> &cxy,
                 const std::vector<double> &obs, const std::vector<int32_t> &ocone, const std::vector<double> &zxy, int rank = -1, int world = 1) {
    std::vector<char> seen(M, 0);
    std::vector<int32_t> first((size_t)world + 1, 0);
    if (rank >= 0) {                                                // the vertices first: the windows are counted in FREE poses
        for (int k = 0; k < N; ++k) {
            CHECK(gs_add_pose(g, 1000 + k, &odom[3 * k]));
            if (k > 0) { double z[3]; se2_between(&odom[3 * (k - 1)], &odom[3 * k], z);
                const double info[9] = {5, 0, 0, 0, 5, 0, 0, 0, 5};
                CHECK(gs_add_odometry_edge(g, 1000 + k - 1, 1000 + k, z, info)); } }
        CHECK(gs_set_fixed_pose(g, 1000, 1)); CHECK(gs_set_fixed_pose(g, 1001, 1));
        CHECK(gs_dist_configure(g, rank, world));
        CHECK(gs_dist_window_starts(g, first.data(), world + 1));
        int nobs = 0;
        for (int k = 0; k < N; ++k) {
            bool mine = k < 2 || (k >= first[(size_t)rank] && k < first[(size_t)rank + 1]);
            for (int w = 1; w < world; ++w) mine = mine || k == first[(size_t)w];
            for (int i = 0; i < K; ++i) { const int l = ocone[(size_t)k * K + i];
                if (!seen[l]) { seen[l] = 1; const double e[2] = {cxy[2 * l] + 0.3, cxy[2 * l + 1] - 0.2}; CHECK(gs_add_landmark(g, l, e)); }      // every cone on every replica, in the same order
                if (!mine) continue;
                const double info[4] = {0.01, 0, 0, 0.01};
                CHECK(gs_add_observation_edge(g, 1000 + k, l, &zxy[2 * ((size_t)k * K + i)], info)); ++nobs; } }
        int fixed = 0; for (int l = 0; l < M && fixed < 2; ++l) if (seen[l]) { CHECK(gs_set_fixed_landmark(g, l, 1)); ++fixed; }
        return nobs;
    }
    for (int k = 0; k < N; ++k) {
        CHECK(gs_add_pose(g, 1000 + k, &odom[3 * k]));
        if (k > 0) { double z[3]; se2_between(&odom[3 * (k - 1)], &odom[3 * k], z);
            const double info[9] = {5, 0, 0, 0, 5, 0, 0, 0, 5};
            CHECK(gs_add_odometry_edge(g, 1000 + k - 1, 1000 + k, z, info)); }
        for (int i = 0; i < K; ++i) { const int l = ocone[(size_t)k * K + i];
            if (!seen[l]) { seen[l] = 1; const double e[2] = {cxy[2 * l] + 0.3, cxy[2 * l + 1] - 0.2}; CHECK(gs_add_landmark(g, l, e)); }
            const double info[4] = {0.01, 0, 0, 0.01};
            CHECK(gs_add_observation_edge(g, 1000 + k, l, &zxy[2 * ((size_t)k * K + i)], info)); }
    }
    CHECK(gs_set_fixed_pose(g, 1000, 1)); CHECK(gs_set_fixed_pose(g, 1001, 1));             // src/slam.cpp:464-474
    int fixed = 0; for (int l = 0; l < M && fixed < 2; ++l) if (seen[l]) { CHECK(gs_set_fixed_landmark(g, l, 1)); ++fixed; }
    return 0;
}

int main() {
    const int N = 2000, M = 400, K = gs_track_obs_per_pose();
    std::vector<double> truth(3 * N), odom(3 * N), cxy(2 * M), obs((size_t)4 * K * N), zxy((size_t)2 * K * N); std::vector<int32_t> ctype(M), ocone((size_t)K * N);
    if (gs_track_generate(N, M, truth.data(), odom.data(), cxy.data(), ctype.data(), obs.data(), ocone.data()) != 0) return 3;
    gs_graph *A = nullptr, *B = nullptr;
    CHECK(gs_create(nullptr, &A)); CHECK(gs_create(nullptr, &B));
    { std::vector<double> az((size_t)K * N), zen((size_t)K * N), di((size_t)K * N);
      for (size_t i = 0; i < (size_t)K * N; ++i) { az[i] = obs[4 * i]; zen[i] = obs[4 * i + 1]; di[i] = obs[4 * i + 2]; }
      CHECK(gs_polar_to_xy_batch(A, K * N, az.data(), zen.data(), di.data(), zxy.data())); }
    if (build(A, N, M, K, odom, cxy, obs, ocone, zxy) || build(B, N, M, K, odom, cxy, obs, ocone, zxy)) return 1;
    gs_stats sa, sb;
    const int da = gs_optimize(A, 10, &sa); if (da < 0) { std::fprintf(stderr, "gs_optimize: %s\n", gs_last_error()); return 1; }
    // ---- the sharded path on one rank: a forced shared top, RCCL inside the library
    gs_debug_options o; CHECK(gs_debug_get_options(B, &o)); o.force_shared_top = 3; CHECK(gs_debug_set_options(B, &o));
    CHECK(gs_dist_configure(B, 0, 1));
    char id[128]; CHECK(gs_dist_unique_id(id));                     // (rank 0 makes it; the other ranks would receive these 128 bytes)
    CHECK(gs_dist_comm_init(B, id, 0, 1));
    CHECK(gs_dist_share_landmark_windows(B));                       // (a group of one: own bits -> ncclAllReduce(uint64, sum) -> back; what replicas without another channel call)
    const int db = gs_dist_optimize(B, 10, &sb); if (db < 0) { std::fprintf(stderr, "gs_dist_optimize: %s\n", gs_last_error()); return 1; }
    std::vector<double> pa(3 * N), pb(3 * N);
    CHECK(gs_get_poses(A, N, nullptr, pa.data())); CHECK(gs_get_poses(B, N, nullptr, pb.data()));
    double rms = 0, worst = 0;
    for (int k = 0; k < N; ++k) rms += pa[3 * k] * pa[3 * k] + pa[3 * k + 1] * pa[3 * k + 1];
    rms = std::sqrt(rms / N);
    for (int i = 0; i < 3 * N; ++i) worst = std::fmax(worst, std::fabs(pa[i] - pb[i]));
    std::printf("iterations %d %d  shared fronts %d  exchange doubles %lld  max |pose diff| / rms %.3g  chi2 %.6g -> %.6g\n", da, db, sb.n_shared_fronts,
                (long long)gs_dist_exchange_doubles(B), worst / rms, sa.chi2_initial, sa.chi2_final);
    const bool ok = da == 10 && db == 10 && sb.n_shared_fronts > 0 && gs_dist_exchange_doubles(B) > 2 && worst / rms < 1e-9 && sa.chi2_final < sa.chi2_initial;
    // ---- two replicas, rank-local ingestion
    bool ok2 = false;
    { const int world = 2; gs_graph *R[2] = {nullptr, nullptr}; int held[2] = {0, 0};
      for (int r = 0; r < world; ++r) { CHECK(gs_create(nullptr, &R[r])); held[r] = build(R[r], N, M, K, odom, cxy, obs, ocone, zxy, r, world); if (held[r] <= 0) return 1; }
      const int Mg = gs_num_landmarks(R[0]);
      std::vector<uint64_t> a((size_t)Mg, 0), b((size_t)Mg, 0), a1((size_t)Mg), b1((size_t)Mg);
      for (int r = 0; r < world; ++r) { CHECK(gs_dist_local_landmark_windows(R[r], a1.data(), b1.data(), Mg)); for (int l = 0; l < Mg; ++l) { a[l] |= a1[l]; b[l] |= b1[l]; } }
      for (int r = 0; r < world; ++r) { CHECK(gs_dist_set_landmark_windows(R[r], a.data(), b.data(), Mg)); CHECK(gs_initialize_optimization(R[r])); }
      const long long X = gs_dist_exchange_doubles(R[0]);
      if (X != gs_dist_exchange_doubles(R[1]) || X <= 2) { std::fprintf(stderr, "exchange layouts differ\n"); return 1; }
      std::vector<double> x0((size_t)X), x1((size_t)X);
      for (int it = 0; it < 10; ++it) {
          CHECK(gs_dist_iterate_local(R[0])); CHECK(gs_dist_iterate_local(R[1]));
          CHECK(gs_dist_read_exchange(R[0], x0.data())); CHECK(gs_dist_read_exchange(R[1], x1.data()));
          for (long long i = 0; i < X; ++i) x0[(size_t)i] += x1[(size_t)i];                       // (the all-reduce)
          CHECK(gs_dist_write_exchange(R[0], x0.data())); CHECK(gs_dist_write_exchange(R[1], x0.data()));
          CHECK(gs_dist_iterate_finish(R[0])); CHECK(gs_dist_iterate_finish(R[1])); }
      std::vector<double> pm(3 * (size_t)N, 0.0), pr(3 * (size_t)N); std::vector<uint8_t> pk((size_t)N), lk((size_t)Mg), pp((size_t)N), lp((size_t)Mg);
      for (int r = 0; r < world; ++r) { CHECK(gs_sync_estimates(R[r])); CHECK(gs_get_poses(R[r], N, nullptr, pr.data())); CHECK(gs_dist_known(R[r], pk.data(), lk.data(), pp.data(), lp.data()));
          for (int k = 0; k < N; ++k) if (pp[(size_t)k]) for (int t = 0; t < 3; ++t) pm[3 * (size_t)k + t] = pr[3 * (size_t)k + t]; }
      double worst2 = 0; for (int i = 0; i < 3 * N; ++i) worst2 = std::fmax(worst2, std::fabs(pa[i] - pm[(size_t)i]));
      std::printf("rank-local ingestion, world 2: %d + %d of %d observation edges held, %lld exchange doubles, max |pose diff| / rms %.3g\n", held[0], held[1], K * N, X, worst2 / rms);
      ok2 = worst2 / rms < 1e-7 && held[0] + held[1] < K * N + 4 * K * world;
      gs_destroy(R[0]); gs_destroy(R[1]); }
    gs_destroy(A); gs_destroy(B);
    std::puts(ok && ok2 ? "OK" : "MISMATCH");
    return ok && ok2 ? 0 : 2;
}
