// ThreadSanitizer check of the host worker pool (csrc/gs_parallel.hpp): many regions in a row, three concurrent callers (two fall back to fresh threads),
// a region inside a region.  g++ -O1 -g -std=c++17 -fsanitize=thread -I opendlv-logic-cfsd18-sensation-slam_amd/csrc tests/tools/pool_tsan.cpp -o /tmp/pool_tsan -lpthread && GS_THREADS=6 /tmp/pool_tsan
#include "gs_parallel.hpp"
#include <cstdio>
#include <numeric>
int main() {
    using namespace gs;
    const int64_t n = 1 << 18;
    std::vector<long> a(n), out(64 * 8, 0);
    std::iota(a.begin(), a.end(), 0);
    long want = 0; for (auto v : a) want += v;
    int bad = 0;
    // many regions in a row from one caller, two concurrent callers (one falls back to fresh threads), a region inside a region
    auto caller = [&](int id) {
        for (int rep = 0; rep < 300; ++rep) {
            std::vector<long> part(64, 0);
            parallel_chunks(n, 1024, [&](int64_t b, int64_t e, int t) { long s = 0; for (int64_t k = b; k < e; ++k) s += a[k];
                if (rep % 50 == 0) parallel_chunks(1000, 10, [&](int64_t b2, int64_t e2, int) { for (int64_t k = b2; k < e2; ++k) s += 0 * k; });
                part[t] = s; });
            long tot = 0; for (auto v : part) tot += v;
            if (tot != want) ++bad;
        }
        out[8 * id] = bad;
    };
    std::thread t1(caller, 1), t2(caller, 2);
    caller(0);
    t1.join(); t2.join();
    std::printf("bad %d\n", bad);
    return bad != 0;
}
