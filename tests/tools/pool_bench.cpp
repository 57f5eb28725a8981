// cost of one (nearly empty) parallel region of csrc/gs_parallel.hpp: g++ -O2 -std=c++17 -I opendlv-logic-cfsd18-sensation-slam_amd/csrc tests/tools/pool_bench.cpp -o tests/_build/pc_bench -lpthread
// GPU box, 16 threads: 320 us with a thread created per part (rounds 1-3), 5.7 us with the pool.
#include "gs_parallel.hpp"
#include <chrono>
#include <cstdio>
#include <algorithm>
int main() {
    using namespace gs;
    std::vector<double> us;
    long sinks[64 * 16] = {0};
    for (int rep = 0; rep < 3; ++rep) {
        us.clear();
        for (int i = 0; i < 200; ++i) {
            auto t0 = std::chrono::steady_clock::now();
            parallel_chunks(1 << 20, 1, [&](int64_t b, int64_t e, int t) { long s = 0; for (int64_t k = b; k < e; k += 4096) s += k; sinks[16 * t] += s; });
            auto t1 = std::chrono::steady_clock::now();
            us.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());
        }
        std::sort(us.begin(), us.end());
        std::printf("threads %d: min %.1f  median %.1f  p90 %.1f  max %.1f us per call\n", host_threads(), us[0], us[100], us[180], us[199]);
    }
    return (int)(sinks[0] & 1);
}
