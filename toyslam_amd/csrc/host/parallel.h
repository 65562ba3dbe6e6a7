// parallel.h — the host side's only threading primitive: contiguous chunks of [0, n) on std::threads.
#pragma once
#include <sched.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <exception>
#include <thread>
#include <vector>

namespace tsgo {

// Upper bound of parallel_chunks' chunk count, whatever TSGO_HOST_THREADS says: callers keep per-chunk outputs in arrays of this size.
constexpr int kMaxHostThreads = 64;

inline int host_threads() {
    if (const char* e = getenv("TSGO_HOST_THREADS")) return std::max(1, std::min(kMaxHostThreads, atoi(e)));
    cpu_set_t set; CPU_ZERO(&set);
    int n = 0;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) n = CPU_COUNT(&set);
    if (n <= 0) n = (int)std::thread::hardware_concurrency();
    return std::max(1, std::min(16, n / 2));     // half the logical CPUs, at most 16 (one GPU's share of an 8-GPU host)
}

// f(chunk, begin, end) over [0, n) split into contiguous chunks, one std::thread each (at least `grain` items per thread).
// Every chunk runs exactly once whatever happens: if a thread cannot be created (EAGAIN under a thread limit — a server runs
// this from up to 64 connection threads) the chunks left over run on the calling thread; an exception thrown by f on a worker
// is carried back and rethrown here after every started thread has been joined (never std::terminate).
template <typename F> inline int parallel_chunks(int n, F f, int grain = 8) {
    const int nt = std::max(1, std::min(host_threads(), n / std::max(1, grain)));
    if (nt == 1) { f(0, 0, n); return 1; }
    std::vector<std::thread> th;
    std::vector<std::exception_ptr> err((size_t)nt);
    auto bounds = [n, nt](int c, int& b, int& e) { b = (int)((int64_t)n * c / nt); e = (int)((int64_t)n * (c + 1) / nt); };
    int started = 0;
    try {
        th.reserve((size_t)nt);
        for (; started < nt - 1; ++started) {
            int b, e; bounds(started, b, e);
            const int c = started;
            th.emplace_back([c, b, e, &f, &err] { try { f(c, b, e); } catch (...) { err[(size_t)c] = std::current_exception(); } });
        }
    } catch (...) {}                       // thread creation failed: `started` chunks are running, the rest run below
    for (int c = started; c < nt; ++c) {   // the last chunk always runs here (the caller would only wait otherwise)
        int b, e; bounds(c, b, e);
        try { f(c, b, e); } catch (...) { err[(size_t)c] = std::current_exception(); }
    }
    for (auto& t : th) t.join();
    for (auto& e : err) if (e) std::rethrow_exception(e);
    return nt;
}

}  // namespace tsgo
