// gs_parallel.hpp — host threads for the structure phase (plan build + table construction in upload_graph).
// The reference runs initializeOptimization / analyzePattern single-threaded (reference src/slam.cpp:480); here the
// structure phase is loops over poses / wave tiles / fronts that are independent once their output offsets are
// known, so it is split over the host cores next to the GPU (GS_THREADS, default min(16, hardware threads)).
#pragma once
#include <algorithm>
#include <cstdlib>
#include <thread>
#include <vector>

#include <cstdint>
#include <sys/mman.h>

namespace gs {

// A freshly allocated big array is first touched in 2 MB steps where the kernel hands out transparent huge pages on request (madvise mode):
// 512 times fewer page faults for the ~100 MB a structure phase at 100k poses fills for the first time.  A hint: harmless where it does not apply.
inline void hint_huge_pages(const void *p, size_t bytes) {
    const uintptr_t step = (uintptr_t)2 << 20, a = ((uintptr_t)p + step - 1) & ~(step - 1), e = ((uintptr_t)p + bytes) & ~(step - 1);
#ifndef GS_NO_HUGE      /* (tuning builds: -DGS_NO_HUGE=1 measures without the hint) */
    if (p && e > a) (void)madvise(reinterpret_cast<void *>(a), (size_t)(e - a), MADV_HUGEPAGE);
#else
    (void)a; (void)e;
#endif
}
template <class V> void hint_huge(const V &v) { hint_huge_pages(v.data(), v.capacity() * sizeof(*v.data())); }

inline int host_threads() {
    static int n = [] {
        int v = (int)std::thread::hardware_concurrency(); if (v <= 0) v = 1;
        v = std::min(v, 16);
        if (const char *e = std::getenv("GS_THREADS")) v = std::max(1, std::atoi(e));
        return v; }();
    return n;
}

// fn(begin, end, thread index): contiguous chunks of [0, n), one per thread; runs inline when the range is small
template <class F> void parallel_chunks(int64_t n, int64_t min_per_thread, F &&fn) {
    int T = (int)std::min<int64_t>(host_threads(), std::max<int64_t>(1, n / std::max<int64_t>(1, min_per_thread)));
    if (T <= 1) { fn((int64_t)0, n, 0); return; }
    std::vector<std::thread> th; th.reserve(T - 1);
    for (int t = 1; t < T; ++t) th.emplace_back([&, t] { fn(n * t / T, n * (t + 1) / T, t); });
    fn((int64_t)0, n / T, 0);
    for (auto &x : th) x.join();
}
inline int chunk_count(int64_t n, int64_t min_per_thread) {
    return (int)std::min<int64_t>(host_threads(), std::max<int64_t>(1, n / std::max<int64_t>(1, min_per_thread)));
}

// run independent tasks concurrently
template <class... Fs> void parallel_tasks(Fs &&...fs) {
    if (host_threads() <= 1) { (fs(), ...); return; }
    std::vector<std::thread> th;
    (th.emplace_back(std::forward<Fs>(fs)), ...);
    for (auto &x : th) x.join();
}

}  // namespace gs
