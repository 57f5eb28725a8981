// ThreadSanitizer check of the structure phase's host threads (csrc/gs_plan.cpp on csrc/gs_parallel.hpp): a chain of poses, each seeing the K cones
// nearest along the track (edges grouped by pose, as the reference inserts them), planned as a single handle, as rank 3 of 8 pose windows from the
// window masks, and by the general recursion — twice each on one workspace (recycled arrays).  Large enough for every parallel region to split.
//   g++ -O1 -g -std=c++17 -fsanitize=thread -I opendlv-logic-cfsd18-sensation-slam_amd/csrc -I include tests/tools/plan_tsan.cpp \
//       opendlv-logic-cfsd18-sensation-slam_amd/csrc/gs_plan.cpp -o /tmp/plan_tsan -lpthread && GS_THREADS=8 /tmp/plan_tsan
#include "gs_host.hpp"
#include <cmath>
#include <cstdio>
int main(int argc, char **argv) {
    using namespace gs;
    const int N = argc > 1 ? std::atoi(argv[1]) : 300000, M = N / 10, K = 8;
    HostGraph g;
    for (int p = 0; p < N; ++p) { g.pose_id.push_back(p); g.pose_fixed.push_back(p == 0 || p == N / 3); const double a = 6.283185307179586 * p / N;
        g.pose_est.push_back(100 * std::cos(a)); g.pose_est.push_back(100 * std::sin(a)); g.pose_est.push_back(a); }
    for (int l = 0; l < M; ++l) { g.lm_id.push_back(l); g.lm_fixed.push_back(0); g.lm_est.push_back(0); g.lm_est.push_back(0); }
    for (int p = 0; p < N; ++p) {
        if (p > 0) { g.pp_i.push_back(p - 1); g.pp_j.push_back(p); for (int t = 0; t < 3; ++t) g.pp_z.push_back(0.1); for (int t = 0; t < 6; ++t) g.pp_info.push_back(t == 0 || t == 3 || t == 5); }
        for (int j = 0; j < K; ++j) { g.pl_p.push_back(p); g.pl_l.push_back((int)(((int64_t)p * M / N + j) % M)); g.pl_z.push_back(1); g.pl_z.push_back(0); g.pl_info.push_back(1); g.pl_info.push_back(0); g.pl_info.push_back(1); } }
    int bad = 0;
    auto run = [&](int world, int rank, bool by_window) {
        PlanOptions o; o.world = world; o.rank = rank; o.by_window = by_window;
        Plan plan; std::string err; std::shared_ptr<void> ws; std::vector<int32_t> first, again;
        if (!build_plan(g, o, plan, err, &ws)) { std::printf("build failed: %s\n", err.c_str()); ++bad; return; }
        export_plan(plan, first);
        if (!build_plan(g, o, plan, err, &ws)) { std::printf("re-plan failed: %s\n", err.c_str()); ++bad; return; }
        export_plan(plan, again);
        if (first != again) { std::printf("world %d rank %d: the re-plan on the recycled arrays differs\n", world, rank); ++bad; }
        std::printf("world %d rank %d by_window %d: %zu fronts, %d scalars\n", world, rank, (int)by_window, plan.fronts.size(), plan.n_scalar);
    };
    run(1, 0, true); run(8, 3, true); run(8, 3, false); run(8, 0, true);
    // the same edges in reverse order: not grouped by pose any more — the counting sorts, and the general recursion for a shard
    { const size_t E = g.pl_p.size(); HostGraph r = g;
      for (size_t k = 0; k < E; ++k) { r.pl_p[k] = g.pl_p[E - 1 - k]; r.pl_l[k] = g.pl_l[E - 1 - k]; }
      std::swap(g, r); run(1, 0, true); run(8, 5, true); std::swap(g, r); }
    std::printf("bad %d\n", bad);
    return bad != 0;
}
