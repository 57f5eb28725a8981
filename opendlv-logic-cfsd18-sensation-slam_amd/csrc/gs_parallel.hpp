// gs_parallel.hpp — host threads for the structure phase (plan build + table construction in upload_graph).
// The reference runs initializeOptimization / analyzePattern single-threaded (reference src/slam.cpp:480); here the
// structure phase is loops over poses / wave tiles / fronts that are independent once their output offsets are
// known, so it is split over the host cores next to the GPU (GS_THREADS, default min(16, hardware threads)).
#pragma once
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <mutex>
#include <thread>
#include <type_traits>
#include <unistd.h>
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

// The workers of the structure phase.  A plan build is ~40 parallel regions of 0.1-1 ms each; creating and joining 15 threads per region cost
// 0.2-0.35 ms a time on the GPU box (tests/_build/pc_bench: ~20 us per thread) — 8-12 of a 20 ms structure phase at 100k poses.  The pool keeps
// host_threads() - 1 threads: between the regions of a build they spin on a generation counter (1.5 ms at most), then sleep on a condition
// variable.  One job at a time: a caller that finds the pool busy (another handle planning on another host thread, the upload helper) or that IS a
// worker runs its region the old way (fresh threads / in order on its own thread) — same parts, same results.  Parts, not threads, carry the index
// handed to fn: part t is run exactly once, by whoever takes it.  A forked child gets a pool of its own on first use.
class WorkerPool {
  public:
    static WorkerPool &get() {
        static WorkerPool *p = nullptr; static std::mutex mu; static pid_t owner = 0;
        std::lock_guard<std::mutex> lk(mu);
        if (!p || owner != getpid()) { p = new WorkerPool(host_threads() - 1); owner = getpid(); }     // (never destroyed: its threads end with the process)
        return *p;
    }
    static bool &in_worker() { static thread_local bool v = false; return v; }
    // runs parts 0 .. T-1 of job(ctx, part); false: the pool is busy (or has no workers) — the caller falls back.
    // A job lives on its caller's stack and is published by pointer.  A worker announces itself (active_) BEFORE it looks at the pointer; the caller takes
    // the pointer back when every part is done and leaves only when no worker is announced any more: nobody can hold a job that has gone, and a worker that
    // wakes late — the job it was woken for long finished, the next one half set up — finds either nothing or a complete job.
    bool run(int T, void (*fn)(void *, int), void *ctx) {
        if (workers_.empty() || in_worker() || !busy_.try_lock()) return false;
        Job job; job.fn = fn; job.ctx = ctx; job.parts = T; job.left.store(T, std::memory_order_relaxed);
        cur_.store(&job, std::memory_order_seq_cst);
        { std::lock_guard<std::mutex> lk(m_); gen_.fetch_add(1, std::memory_order_release); }
        if (sleepers_.load(std::memory_order_acquire) != 0) cv_.notify_all();
        work(&job, 0);
        while (job.left.load(std::memory_order_acquire) != 0) __builtin_ia32_pause();
        cur_.store(nullptr, std::memory_order_seq_cst);
        while (active_.load(std::memory_order_seq_cst) != 0) __builtin_ia32_pause();
        busy_.unlock();
        return true;
    }
  private:
    struct Job { void (*fn)(void *, int) = nullptr; void *ctx = nullptr; int parts = 0; std::atomic<int> next{0}, left{0}; std::atomic<uint64_t> taken{0}; };
    explicit WorkerPool(int n) { for (int i = 0; i < n; ++i) workers_.emplace_back([this, i] { loop(i + 1); }); for (auto &t : workers_) t.detach(); }
    // Part t goes to thread t when that thread is there to take it (the caller is thread 0, worker i thread i + 1): the same thread then walks the same
    // slice of the arrays in one region after the other — and in one plan build after the other —, which keeps a slice in ONE core's caches; whoever
    // is done with its own part takes what is left (a worker that sleeps does not hold its part back).  More than 64 parts: a plain counter.
    static void work(Job *j, int me) {
        if (j->parts > 64) {
            for (;;) { const int t = j->next.fetch_add(1, std::memory_order_acq_rel); if (t >= j->parts) return;
                j->fn(j->ctx, t); j->left.fetch_sub(1, std::memory_order_acq_rel); } }
        auto take = [&](int t) { const uint64_t bit = 1ull << t; return !(j->taken.fetch_or(bit, std::memory_order_acq_rel) & bit); };
        if (me < j->parts && take(me)) { j->fn(j->ctx, me); j->left.fetch_sub(1, std::memory_order_acq_rel); }
        for (int t = 0; t < j->parts; ++t)
            if (!(j->taken.load(std::memory_order_acquire) & (1ull << t)) && take(t)) { j->fn(j->ctx, t); j->left.fetch_sub(1, std::memory_order_acq_rel); }
    }
    void loop(int me) {
        in_worker() = true;
        uint64_t seen = gen_.load(std::memory_order_acquire);
        for (;;) {
            bool got = false;
            const auto until = std::chrono::steady_clock::now() + std::chrono::microseconds(1500);     // (waking a sleeping thread costs ~20 us — per thread, one after the other)
            while (!got) { for (int spin = 0; spin < 256 && !got; ++spin) { got = gen_.load(std::memory_order_acquire) != seen; if (!got) __builtin_ia32_pause(); }
                if (!got && std::chrono::steady_clock::now() > until) break; }
            if (!got) { std::unique_lock<std::mutex> lk(m_); sleepers_.fetch_add(1, std::memory_order_acq_rel);
                cv_.wait(lk, [&] { return gen_.load(std::memory_order_acquire) != seen; }); sleepers_.fetch_sub(1, std::memory_order_acq_rel); }
            seen = gen_.load(std::memory_order_acquire);
            active_.fetch_add(1, std::memory_order_seq_cst);
            if (Job *j = cur_.load(std::memory_order_seq_cst)) work(j, me);
            active_.fetch_sub(1, std::memory_order_seq_cst);
        }
    }
    std::vector<std::thread> workers_;
    std::mutex m_, busy_; std::condition_variable cv_;
    std::atomic<uint64_t> gen_{0}; std::atomic<int> sleepers_{0}, active_{0};
    std::atomic<Job *> cur_{nullptr};
};

// fn(begin, end, part index): contiguous chunks of [0, n), one per part (as many parts as host threads, fewer for small ranges); inline when the range is small
template <class F> void parallel_chunks(int64_t n, int64_t min_per_thread, F &&fn) {
    int T = (int)std::min<int64_t>(host_threads(), std::max<int64_t>(1, n / std::max<int64_t>(1, min_per_thread)));
    if (T <= 1) { fn((int64_t)0, n, 0); return; }
    typedef typename std::remove_reference<F>::type Fn;
    struct Ctx { Fn *fn; int64_t n; int T; } ctx{&fn, n, T};
    if (WorkerPool::get().run(T, [](void *c, int t) { auto *x = static_cast<Ctx *>(c); (*x->fn)(x->n * t / x->T, x->n * (t + 1) / x->T, t); }, &ctx)) return;
    if (WorkerPool::in_worker()) { for (int t = 0; t < T; ++t) fn(n * t / T, n * (t + 1) / T, t); return; }      // a region inside a region: in order, on this thread
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
