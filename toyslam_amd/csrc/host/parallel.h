// parallel.h — the host side's only threading primitive: contiguous chunks of [0, n) on std::threads.
#pragma once
#include <sched.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <thread>
#include <vector>

namespace tsgo {

inline int host_threads() {
    if (const char* e = getenv("TSGO_HOST_THREADS")) return std::max(1, atoi(e));
    cpu_set_t set; CPU_ZERO(&set);
    int n = 0;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) n = CPU_COUNT(&set);
    if (n <= 0) n = (int)std::thread::hardware_concurrency();
    return std::max(1, std::min(16, n / 2));     // half the logical CPUs, at most 16 (one GPU's share of an 8-GPU host)
}

// f(chunk, begin, end) over [0, n) split into contiguous chunks, one std::thread each (at least `grain` items per thread).
template <typename F> inline int parallel_chunks(int n, F f, int grain = 8) {
    const int nt = std::max(1, std::min(host_threads(), n / std::max(1, grain)));
    if (nt == 1) { f(0, 0, n); return 1; }
    std::vector<std::thread> th;
    for (int c = 0; c < nt; ++c) {
        const int b = (int)((int64_t)n * c / nt), e = (int)((int64_t)n * (c + 1) / nt);
        th.emplace_back([=, &f] { f(c, b, e); });
    }
    for (auto& t : th) t.join();
    return nt;
}

}  // namespace tsgo
