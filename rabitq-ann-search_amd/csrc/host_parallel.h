// host_parallel.h — host worker threads of the library (repacker, file validation, builder stages).
//
// One process per GPU, eight per node: at most 64 threads each.  A worker's exception does not reach
// std::terminate: the first one is kept and rethrown on the calling thread after every worker has been
// joined, so the C ABI reports it as an error code (cphnsw_mi355x.hip: guarded).
#pragma once
#include <algorithm>
#include <atomic>
#include <cstddef>
#include <cstdlib>
#include <exception>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace cph {

inline size_t host_threads() {
    if (const char* e = getenv("CPH_BUILD_THREADS")) return (size_t)std::max(1, atoi(e));
    const unsigned hw = std::min(64u, std::thread::hardware_concurrency());
    return hw ? hw : 4;
}

// Runs body(t) on `nt` threads (t = 0..nt-1; the caller's thread is one of them) and rethrows the first exception.
inline void run_threads(size_t nt, const std::function<void(size_t)>& body) {
    if (nt <= 1) { body(0); return; }
    std::exception_ptr first;
    std::mutex mu;
    auto guarded = [&](size_t t) {
        try {
            body(t);
        } catch (...) {
            std::lock_guard<std::mutex> lk(mu);
            if (!first) first = std::current_exception();
        }
    };
    std::vector<std::thread> th;
    th.reserve(nt - 1);
    try {
        for (size_t t = 1; t < nt; ++t) th.emplace_back(guarded, t);
    } catch (...) {                       // thread creation failed: the started ones still have to be joined
        std::lock_guard<std::mutex> lk(mu);
        if (!first) first = std::current_exception();
    }
    guarded(0);
    for (auto& x : th) x.join();
    if (first) std::rethrow_exception(first);
}

// fn(lo, hi) over [0, n) in chunks handed out dynamically.
inline void parallel_for(size_t n, size_t min_chunk, const std::function<void(size_t, size_t)>& fn) {
    if (n == 0) return;
    const size_t nt = std::max<size_t>(1, std::min<size_t>(host_threads(), n / std::max<size_t>(min_chunk, 1)));
    if (nt <= 1) { fn(0, n); return; }
    std::atomic<size_t> next{0};
    std::atomic<bool> failed{false};
    const size_t chunk = std::max<size_t>(std::max<size_t>(min_chunk, 1), n / (nt * 16));
    run_threads(nt, [&](size_t) {
        try {
            while (!failed.load(std::memory_order_relaxed)) {
                const size_t lo = next.fetch_add(chunk);
                if (lo >= n) break;
                fn(lo, std::min(n, lo + chunk));
            }
        } catch (...) {
            failed.store(true, std::memory_order_relaxed);
            throw;
        }
    });
}

}  // namespace cph
