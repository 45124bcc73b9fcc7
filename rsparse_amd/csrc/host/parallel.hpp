// Small host-side helpers for the analysis stages that are worth threading (independent items, results independent of
// the thread count).  No device calls inside the workers.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <initializer_list>
#include <thread>
#include <vector>

namespace rsp {

// worker threads: RSP_HOST_THREADS, else OMP_NUM_THREADS (the GPU boxes set it to their CPU share), else the
// hardware's count capped at 16
// (thread-local override: the calling thread's next analyses use exactly this many; 0 = no override.  The self-test uses
// it instead of touching the process environment, which other threads may be reading.)
inline int64_t &host_threads_override()
{
    static thread_local int64_t value = 0;
    return value;
}

inline int64_t host_threads()
{
    if (host_threads_override() > 0) return std::min<int64_t>(host_threads_override(), 64);
    for (const char *name : {"RSP_HOST_THREADS", "OMP_NUM_THREADS"})
        if (const char *e = std::getenv(name))
            if (std::atoll(e) > 0) return std::min<int64_t>(std::atoll(e), 64);
    const unsigned hw = std::thread::hardware_concurrency();
    return std::max<int64_t>(1, std::min<int64_t>(hw ? hw : 1, 16));
}

// fn(begin, end) over [0, n) cut into contiguous chunks, one per thread; runs inline when n is small
template <class F>
inline void parallel_ranges(int64_t n, int64_t min_per_thread, F &&fn)
{
    const int64_t t = std::max<int64_t>(1, std::min<int64_t>(host_threads(), n / std::max<int64_t>(1, min_per_thread)));
    if (t <= 1) {
        fn((int64_t)0, n);
        return;
    }
    std::vector<std::thread> pool;
    for (int64_t k = 1; k < t; ++k) pool.emplace_back([&fn, n, t, k] { fn(n * k / t, n * (k + 1) / t); });
    fn((int64_t)0, n / t);
    for (auto &th : pool) th.join();
}

// std::sort semantics (same result as a sequential sort when equal elements are indistinguishable): chunks sorted on
// threads, then merged pairwise
template <class It>
inline void parallel_sort(It first, It last)
{
    const int64_t n = last - first;
    int64_t t = std::min<int64_t>(host_threads(), n >> 16);
    if (t <= 1) {
        std::sort(first, last);
        return;
    }
    std::vector<int64_t> cut(t + 1);
    for (int64_t k = 0; k <= t; ++k) cut[k] = n * k / t;
    {
        std::vector<std::thread> pool;
        for (int64_t k = 1; k < t; ++k) pool.emplace_back([=] { std::sort(first + cut[k], first + cut[k + 1]); });
        std::sort(first + cut[0], first + cut[1]);
        for (auto &th : pool) th.join();
    }
    for (int64_t step = 1; step < t; step *= 2) {
        std::vector<std::thread> pool;
        for (int64_t k = 0; k + step < t; k += 2 * step) {
            const int64_t a = cut[k], m = cut[k + step], b = cut[std::min<int64_t>(t, k + 2 * step)];
            pool.emplace_back([=] { std::inplace_merge(first + a, first + m, first + b); });
        }
        for (auto &th : pool) th.join();
    }
}

}  // namespace rsp
