// tsgo_hip.hip — device side of the C ABI in include/tsgo.h: buffers, launches, the Gauss-Newton
// loop with the reference's stop rules (remote/optimizer/OptimizerCpu.h:80-180), hipGraph replay of
// the PCG iteration, RCCL all-reduces for edge-sharded runs.  Kernels: tsgo_kernels.h.
//
// There is NO CPU fallback in this file: every entry point that computes needs a gfx950 device and
// returns an error otherwise.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/tsgo.h"
#ifdef TSGO_TESTING
#include "../../include/tsgo_testing.h"
#endif
#include "host/amg.h"
#include "host/errors.h"
#include "host/knobs.h"
#include "host/parallel.h"
#include "host/problem.h"
#include "tsgo_amg_kernels.h"
#include "tsgo_kernels.h"
#include "tsgo_sym_kernels.h"

namespace {

using namespace tsgo;

#define HIP_OK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess)                                                                          \
            return set_error(-10, std::string(#expr) + ": " + hipGetErrorString(e_));                  \
    } while (0)
#define NCCL_OK(expr)                                                                                  \
    do {                                                                                               \
        ncclResult_t e_ = (expr);                                                                      \
        if (e_ != ncclSuccess)                                                                         \
            return set_error(-11, std::string(#expr) + ": " + ncclGetErrorString(e_));                 \
    } while (0)

// launch a kernel template over the lanes-per-vertex parameter
#define LAUNCH_G(G, KERNEL, grid, stream, ...)                                                         \
    do {                                                                                               \
        switch (G) {                                                                                   \
            case 1: hipLaunchKernelGGL((KERNEL<T, 1>), dim3(grid), dim3(kBlock), 0, stream, __VA_ARGS__); break; \
            case 2: hipLaunchKernelGGL((KERNEL<T, 2>), dim3(grid), dim3(kBlock), 0, stream, __VA_ARGS__); break; \
            case 4: hipLaunchKernelGGL((KERNEL<T, 4>), dim3(grid), dim3(kBlock), 0, stream, __VA_ARGS__); break; \
            default: hipLaunchKernelGGL((KERNEL<T, 8>), dim3(grid), dim3(kBlock), 0, stream, __VA_ARGS__); break; \
        }                                                                                              \
    } while (0)
#define LAUNCH_GM(G, KERNEL, MODE, grid, stream, ...)                                                  \
    do {                                                                                               \
        switch (G) {                                                                                   \
            case 1: hipLaunchKernelGGL((KERNEL<T, 1, MODE>), dim3(grid), dim3(kBlock), 0, stream, __VA_ARGS__); break; \
            case 2: hipLaunchKernelGGL((KERNEL<T, 2, MODE>), dim3(grid), dim3(kBlock), 0, stream, __VA_ARGS__); break; \
            case 4: hipLaunchKernelGGL((KERNEL<T, 4, MODE>), dim3(grid), dim3(kBlock), 0, stream, __VA_ARGS__); break; \
            default: hipLaunchKernelGGL((KERNEL<T, 8, MODE>), dim3(grid), dim3(kBlock), 0, stream, __VA_ARGS__); break; \
        }                                                                                              \
    } while (0)

#define LAUNCH_GML(G, KERNEL, MODE, LOW, grid, stream, ...)                                             \
    do {                                                                                               \
        switch (G) {                                                                                   \
            case 1: hipLaunchKernelGGL((KERNEL<T, 1, MODE, LOW>), dim3(grid), dim3(kBlock), 0, stream, __VA_ARGS__); break; \
            case 2: hipLaunchKernelGGL((KERNEL<T, 2, MODE, LOW>), dim3(grid), dim3(kBlock), 0, stream, __VA_ARGS__); break; \
            case 4: hipLaunchKernelGGL((KERNEL<T, 4, MODE, LOW>), dim3(grid), dim3(kBlock), 0, stream, __VA_ARGS__); break; \
            default: hipLaunchKernelGGL((KERNEL<T, 8, MODE, LOW>), dim3(grid), dim3(kBlock), 0, stream, __VA_ARGS__); break; \
        }                                                                                              \
    } while (0)
#define LAUNCH_GML1(G, KERNEL, LOW, grid, stream, ...)                                                 \
    do {                                                                                               \
        switch (G) {                                                                                   \
            case 1: hipLaunchKernelGGL((KERNEL<T, 1, LOW>), dim3(grid), dim3(kBlock), 0, stream, __VA_ARGS__); break; \
            case 2: hipLaunchKernelGGL((KERNEL<T, 2, LOW>), dim3(grid), dim3(kBlock), 0, stream, __VA_ARGS__); break; \
            case 4: hipLaunchKernelGGL((KERNEL<T, 4, LOW>), dim3(grid), dim3(kBlock), 0, stream, __VA_ARGS__); break; \
            default: hipLaunchKernelGGL((KERNEL<T, 8, LOW>), dim3(grid), dim3(kBlock), 0, stream, __VA_ARGS__); break; \
        }                                                                                              \
    } while (0)

}  // namespace

#ifdef TSGO_TESTING
// An all-reduce among engine handles of ONE process (typically sharing one device): every rank stages its buffer in host
// memory, all ranks meet, every rank sums the staged buffers in rank order (same bits everywhere, as with RCCL) and copies
// the sum back.  Slow by design; it exists so that the edge-sharded device path (shard tables, ownership rules, per-rank
// level-0 lists, where the all-reduces sit) can be run with 2, 3, ... ranks on a box with a single GPU, where RCCL refuses
// two ranks on one device.  tests/test_gpu_sharded_inprocess.py.
struct tsgo_local_group {
    int world = 1;
    std::mutex m; std::condition_variable cv;
    int arrived = 0; long generation = 0;
    std::vector<std::vector<unsigned char>> stage;
    bool broken = false;                     // a rank gave up waiting: every later barrier fails at once
    // false: the other ranks did not arrive within kLocalBarrierSeconds (they took a different decision, or one of them
    // returned with an error) — the caller reports that instead of waiting forever
    bool barrier() {
        std::unique_lock<std::mutex> l(m);
        if (broken) return false;
        const long gen = generation;
        if (++arrived == world) { arrived = 0; ++generation; cv.notify_all(); return true; }
        if (!cv.wait_for(l, std::chrono::seconds(kLocalBarrierSeconds), [&] { return generation != gen || broken; }) || broken) {
            broken = true; cv.notify_all();
            return false;
        }
        return true;
    }
    static constexpr int kLocalBarrierSeconds = 120;
};
#else
struct tsgo_local_group;      // in-process all-reduce group: TSGO_TESTING builds only (include/tsgo_testing.h)
#endif

namespace {

// The multigrid pattern builder runs on a thread of its own and does its device work (symbolic products) on a stream of its own:
// the helpers below use the calling thread's stream, which is this one on the builder thread and the engine's elsewhere.
thread_local hipStream_t t_builder_stream = nullptr;

struct IEngine {
    virtual ~IEngine() {}
    virtual int set_graph(const tsgo_graph& g) = 0;
    virtual int optimize(int iterations, tsgo_stats* st) = 0;
    virtual int get_vertices(double* out) = 0;
    virtual int linearize(double* diag, double* grad, double* chi2) = 0;
    virtual int solve_step(double* delta, double* chi2, int* iters) = 0;
    virtual int time_kernel(int which, int reps, double* us, double* bytes) = 0;
    virtual int cycle_probe(int reps, tsgo_cycle_level* out, int cap) = 0;
    virtual int profile_iteration(int reps, tsgo_prof_entry* out, int cap) = 0;
    virtual int comm_selftest(int* ranks_out) = 0;
    virtual void reset_history() = 0;
    ncclComm_t comm = nullptr;
    tsgo_local_group* lgroup = nullptr;      // in-process stand-in for the communicator (tests on a one-GPU box; always null outside TSGO_TESTING builds)
};

constexpr int kRhoSteps = 16, kRhoBlocks = 64, kRhoEvery = 8;   // smoother-damping estimate: power steps, partial sums, refresh period
constexpr int kAmgStallIter = 400;      // a multigrid-preconditioned solve that has not even halved r^T M^-1 r by this iteration is treated as a
constexpr double kAmgStallRatio = 0.5;  // failed preconditioner (stagnation) and repeated with block-Jacobi; one that is converging, however slowly
                                        // (odometry-only graphs under the analytic Jacobians are beam-like: 600 iterations at 12k poses, where
                                        // block-Jacobi needs > 10^5), runs on to pcg_max_iters
constexpr double kMediumPairList = 6;   // Galerkin products whose average pair list is longer than this share an output block among 8 lanes,
constexpr double kLongPairList = 16;    // ... longer than this among 16 lanes ...
constexpr double kVeryLongPairList = 200; // ... or a whole wavefront
constexpr int kHierMaxAge = 4;          // a multigrid hierarchy serves at most this many consecutive linearisations (2: 4.32, 3: 4.25, 4: 4.16, 6: 4.16 ms per step at 100k poses, profiles/r03h_*) ...
constexpr int kYoungLins = 6, kYoungMaxAge = 2;   // ... a graph's first linearisations: two per hierarchy at most (do_linearize)
constexpr double kHostSlowFraction = 0.6;   // use_graphs = 2: eager launches need the host to be done enqueueing a burst well before the device is done running it.  At 100k
                                            // poses the host needs 0.4 of the solve's time (90 of 217 us per iteration) and eager wins by 2 %; at 10k poses 0.9 (90 of 99 us):
                                            // eager is then as fast on a good run (2.25 ms per step against 2.32) and 20-30 % slower on a bad one — replay
constexpr int kPaceLead = 1;             // do_solve_paced: iterations the host enqueues ahead of the one whose gate has run
constexpr int kHierSlack = 2;           // ... and is rebuilt as soon as a solve needs more than this many iterations over its first
constexpr int kPackedCycleMaxIters = 64; // a multigrid solve that needs more iterations than this is on an ill-conditioned graph: its cycle leaves the packed halves for f32
constexpr int kHierFreshAbove = 64;     // ... and at every linearisation while solves take more iterations than this (a build costs about four)
constexpr double kProfBlockUsPerLaunch = 10.0;   // tsgo_profile_iteration: host time allowed per enqueued launch + event behind the blocking kernel
constexpr int kChunk = 16;   // PCG iterations per captured hipGraph (even: the state ring has 2 slots)
constexpr int kChunkAmg = 2; // with the multigrid V-cycle an iteration is ~30 launches and a solve ~50 iterations

// Holds its stream for `ms` milliseconds (constant-rate 100 MHz counter), bounded: at most a few hundred ms whatever is asked.
__global__ void k_wait_ms(int ms) {
    const long long ticks = (long long)(ms < 0 ? 0 : (ms > 400 ? 400 : ms)) * 100000ll;
    const long long t0 = (long long)wall_clock64();
    while ((long long)wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}

template <typename T> struct DevLevel {      // device copy of one AmgLevel (host/amg.h) + its numeric arrays
    int n = 0, n_agg = 0, nnzA = 0, nnzP = 0, nnzT = 0, nnzNext = 0;
    double pairs_T = 0, pairs_A = 0;      // average pair-list lengths of the two Galerkin products
    int *A_ptr = nullptr, *A_col = nullptr, *A_row = nullptr, *diag = nullptr;
    int *P_ptr = nullptr, *P_col = nullptr, *P_row = nullptr, *p_self = nullptr, *ps_ptr = nullptr, *ps_x = nullptr, *ps_y = nullptr;
    int *R_ptr = nullptr, *R_col = nullptr, *r_to_p = nullptr, *p_to_r = nullptr;
    int *ts_ptr = nullptr, *ts_x = nullptr, *ts_y = nullptr, *as_ptr = nullptr, *as_x = nullptr, *as_y = nullptr, *as_mirror = nullptr, *as_upper = nullptr;
    int n_upper = 0;
    int *T_ptr = nullptr, *T_col = nullptr;      // pattern of T = A P (the factored dense level reads it: E = P - W T)
    bool patterns_up = false;      // A / P / R patterns are on the device already (the device built this level's pair-list products)
    bool products_dev = false;     // ts_* / as_* were built on the device (tsgo_sym_kernels.h): nothing of them exists on the host
    using H = HT<T>;                              // hierarchy storage type (tsgo_amg_kernels.h)
    T* rel = nullptr;
    H *A = nullptr, *Dinv = nullptr, *P = nullptr, *Tv = nullptr, *Rv = nullptr;
    uint32_t *Apm = nullptr, *Ppm = nullptr, *Rpm = nullptr;       // cycle format: the same blocks, plane-major within each row, f32 or packed half (tsgo_amg_kernels.h)
    T *r = nullptr, *z = nullptr, *res = nullptr, *z2 = nullptr;
};

template <typename T> struct Engine : IEngine {
    tsgo_config cfg;
    Problem pr;
    hipStream_t stream = nullptr, stream2 = nullptr;     // stream2: the pattern builder thread's
    hipStream_t cs() const { return t_builder_stream ? t_builder_stream : stream; }
    std::mutex slab_mu;                                    // the slab allocator is shared by the two threads of tsgo_set_graph
    // Device memory of a graph is bump-allocated from a few large slabs that the handle keeps: tsgo_set_graph with a new
    // structure frees nothing and allocates nothing as long as the new graph fits in what an earlier one needed (hipFree is
    // synchronous and a request makes ~230 allocations).  Slabs grow geometrically (64 MB ... 1 GB each) and are returned
    // to the driver when the handle is destroyed.
    struct Slab { char* base; size_t cap, used; };
    std::vector<Slab> slabs; size_t slab_total = 0;
    static constexpr size_t kSlabMin = size_t(64) << 20, kSlabMax = size_t(1) << 30;
    bool have_graph_data = false;
    double ms_setup = 0;
    // Structure of the graph the device tables were built for (a SLAM front-end resends the same
    // graph with new estimates).  A request with the same vertex ids / types, edge list and fixed list only refills
    // state and measurement planes: no layout build, no multigrid patterns, no table or hierarchy upload, no capture.
    struct Structure {
        std::vector<uint32_t> v_id, v_type, e_type, e_ids, fixed;
        bool same_as(const tsgo_graph& g) const {
            auto eq = [](const std::vector<uint32_t>& a, const uint32_t* b, size_t n) { return a.size() == n && (n == 0 || std::memcmp(a.data(), b, n * sizeof(uint32_t)) == 0); };
            return g.n_vertices >= 0 && g.n_edges >= 0 && g.n_fixed >= 0 && eq(v_id, g.v_id, (size_t)g.n_vertices) && eq(v_type, g.v_type, (size_t)g.n_vertices) &&
                   eq(e_type, g.e_type, (size_t)g.n_edges) && eq(e_ids, g.e_ids, 2 * (size_t)g.n_edges) && eq(fixed, g.fixed, (size_t)g.n_fixed);
        }
        void take(const tsgo_graph& g) {
            v_id.assign(g.v_id, g.v_id + g.n_vertices); v_type.assign(g.v_type, g.v_type + g.n_vertices);
            e_type.assign(g.e_type, g.e_type + g.n_edges); e_ids.assign(g.e_ids, g.e_ids + 2 * (size_t)g.n_edges);
            fixed.assign(g.fixed, g.fixed + g.n_fixed);
        }
    } structure;
    int structure_reuses = 0;
    bool last_set_reused = false;
    std::vector<double> lm_values;                 // per LM edge: (zx, zy, w0, w1), scratch of stage_values
    T* stage = nullptr; size_t stage_cap = 0;      // pinned staging for everything that goes to the device in type T
    T *st_p = nullptr, *st_l = nullptr, *st_o = nullptr;   // the static planes of the three tables (Table<T>::st, writable)

    // device
    T *ps = nullptr, *theta = nullptr, *lmrec = nullptr, *gauge_p = nullptr, *gauge_l = nullptr;
    Table<T> tp{}, tl{}, to{};
    T *part = nullptr, *dp = nullptr, *minv = nullptr, *r = nullptr, *p = nullptr, *q = nullptr, *x = nullptr, *zc = nullptr;
    float *zc32 = nullptr, *tvec32 = nullptr;      // f32 copies of the gathered records, read by the two products inside the multigrid cycle
    T *sbuf = nullptr, *tvec = nullptr, *ninv = nullptr, *dl = nullptr, *gpart[2] = {nullptr, nullptr}, *npart = nullptr;
    CgState<T>* st[2] = {nullptr, nullptr};
    CgState<T>* h_state = nullptr;     // pinned
    int* h_flag = nullptr;             // pinned, coherent: what the gate of the last launched iteration saw (k_iter_gate, do_solve_paced)
    T* h_scratch = nullptr;            // pinned, partial sums
    int nbP = 0, nbL = 0, nbC = 0;
    hipGraphExec_t cg_graph = nullptr;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    int predicted_cg = 0;
    double dev_us_per_iter = 0;        // wall time of the last multigrid solve over its iterations
    bool host_slow = false;            // use_graphs = 2: the host thread has been seen to enqueue too slowly for eager launches (do_solve_once)
    int n_host_slow = 0, n_decided = 0, n_slow_seen = 0;
    static constexpr int kDecideSolves = 3;
    bool replayed = false;             // the last tsgo_optimize replayed captured iterations (tsgo_stats.graph_replay)
    static constexpr int kAgeSlots = 16;
    int iters_by_age[kAgeSlots] = {};      // PCG iterations of the last solve that ran on a hierarchy of that age (do_solve_once's burst)
    // multigrid preconditioner.  Edge-sharded runs use it too: every rank holds the whole hierarchy (patterns from the whole graph,
    // level-0 blocks all-reduced, everything below computed redundantly); only the level-0 contribution lists are per shard
    bool amg_on = false;
    AmgSym amg;
    std::vector<DevLevel<T>> lv;
    int *sc_ptr = nullptr, *sc_optr = nullptr; uint32_t *sc_si = nullptr, *sc_sk = nullptr, *sc_os = nullptr;
    int *last_ptr = nullptr, *last_col = nullptr; int nb_last = 0, nnz_last = 0;
    using H = HT<T>;
    H* A_last = nullptr;
    T *inv_last = nullptr, *r_last = nullptr, *z_last = nullptr, *rzpart = nullptr;
    double ms_amg_symbolic = 0;
    T *omega_dev = nullptr, *one_dev = nullptr, *gscale_dev = nullptr, *pw_a = nullptr, *pw_b = nullptr, *rho_part = nullptr;
    T* h_rho = nullptr;                 // pinned
    std::vector<double> omega_host;    // smoother damping per level (diagnostics)
    int lin_count = 0;
    int n_lins = 0;                    // linearisations of this graph so far (reset with the solver state)
    int hier_age = -1, hier_max_age = kHierMaxAge, hier_slack = kHierSlack, iters_fresh = 0, iters_last = 0;   // -1: no valid hierarchy
    T* hist[kMaxWarm] = {};            // pose deltas of the last solves, newest first (warm start)
    bool have_prev = false;            // hist[0] holds the pose delta of the previous solve
    int n_prev = 0;                    // how many consecutive deltas are held
    int n_tested = 0;                  // orders 1..n_tested of the warm start's extrapolation have an error in warm_err (k_save_x)
    T* warm_err = nullptr;             // [kMaxWarm][nbC]
    int* warm_order_dev = nullptr;     // the order the last warm start took (diagnostics)
    // Coefficients of every extrapolation order (k_pack_x): row m-1 = (-1)^j C(m, j+1) a^(j+1), a = 1 - step.
    void warm_coefficients(WarmTerms<T>& w) const {
        const double a = 1.0 - step_scale();
        for (int m = 1; m <= kMaxWarm; ++m) {
            double binom = 1, apow = 1;
            for (int j = 1; j <= kMaxWarm; ++j) {
                if (j <= m) { binom = binom * (m - j + 1) / j; apow *= a; w.c[m - 1][j - 1] = (T)((j & 1 ? 1.0 : -1.0) * binom * apow); }
                else w.c[m - 1][j - 1] = T(0);
            }
        }
    }
    int coarse_sweeps = kCoarseSweeps;
    std::vector<int> sweeps_list;      // research: sweeps per side on levels 1, 2, ... (TSGO_SWEEPS_LIST="2,2,1"; the last entry repeats)
    int nu_at(size_t l) const {
        if (!sweeps_list.empty()) return sweeps_list[std::min(l - 1, sweeps_list.size() - 1)];
        return lv[l].n <= kSmallLevelRows ? kSmallLevelSweeps : coarse_sweeps;
    }
    bool low_cycle = true;     // f32 slot planes for the Schur products inside the multigrid cycle
    bool cy16 = true;          // the cycle-format copies of A_l, P_l, R_l as packed half floats (20 B per block) or f32 (36 B): tsgo_config.cycle_storage,
                               // until a solve on this structure shows that the graph is too ill-conditioned for 11-bit blocks (do_solve)
    int n_cycle_f32_switches = 0;
    size_t cyw() const { return cy16 ? (size_t)kCyWordsF16 : (size_t)kCyWordsF32; }

    // Environment, read ONCE per handle (tsgo_create).  Operational: verbosity / timing traces.  Research and test hooks
    // (TSGO_RESEARCH_ENV: only builds with -DTSGO_TESTING see them, host/knobs.h).
    bool say_env = false, solve_timing = false, stage_timing = false;
    bool hook_inject_amg_failure = false;      // TSGO_INJECT_AMG_FAILURE: the first multigrid solve of every tsgo_optimize is declared broken down
    bool hook_force_host_slow = false;         // TSGO_FORCE_HOST_SLOW: the handle believes its host thread is too slow for eager launches
    bool hook_force_paced = false;             // TSGO_FORCE_PACED: paced eager launches from the first solve on (no timing heuristics)
    int pace_lead = kPaceLead;                 // TSGO_PACE_LEAD: iterations enqueued ahead of the gate that has reported
    explicit Engine(const tsgo_config& c) : cfg(c) {
        say_env = c.verbose || getenv("TSGO_VERBOSE") != nullptr;
        solve_timing = getenv("TSGO_SOLVE_TIMING") != nullptr;
        stage_timing = getenv("TSGO_STAGE_TIMING") != nullptr;
        if (const char* e = TSGO_RESEARCH_ENV("TSGO_SWEEPS_LIST")) for (const char* q = e; *q;) { sweeps_list.push_back(std::max(1, std::min(4, atoi(q)))); while (*q && *q != ',') ++q; if (*q == ',') ++q; }
        explicit0 = c.cycle_level0 != 0;
        cy16 = c.cycle_storage != 32;
        if (const char* e = TSGO_RESEARCH_ENV("TSGO_HOST_PRODUCTS")) device_products = atoi(e) == 0;
        if (const char* e = TSGO_RESEARCH_ENV("TSGO_SYM_DECLINE")) sym_decline = atoi(e);
        if (const char* e = TSGO_RESEARCH_ENV("TSGO_HIER_MAX_AGE")) hier_max_age = std::max(1, atoi(e));
        if (const char* e = TSGO_RESEARCH_ENV("TSGO_HIER_SLACK")) hier_slack = std::max(0, atoi(e));
        if (const char* e = TSGO_RESEARCH_ENV("TSGO_PACE_LEAD")) pace_lead = std::max(1, atoi(e));
        hook_inject_amg_failure = TSGO_RESEARCH_ENV("TSGO_INJECT_AMG_FAILURE") != nullptr;
        hook_force_host_slow = TSGO_RESEARCH_ENV("TSGO_FORCE_HOST_SLOW") != nullptr;
        hook_force_paced = TSGO_RESEARCH_ENV("TSGO_FORCE_PACED") != nullptr;
    }

    ~Engine() override { release(); for (Slab& sl : slabs) (void)hipFree(sl.base); if (carry_dev) (void)hipFree(carry_dev); if (stage) (void)hipHostFree(stage); if (stream) (void)hipStreamDestroy(stream); if (stream2) (void)hipStreamDestroy(stream2); for (auto& e : ev) if (e) (void)hipEventDestroy(e); for (auto& m : prof) (void)hipEventDestroy(m.e); }

    void release() {
        if (amg_builder.joinable()) amg_builder.join();
        if (cg_graph) { (void)hipGraphExecDestroy(cg_graph); cg_graph = nullptr; }
        for (Slab& sl : slabs) sl.used = 0;
        if (h_state) { (void)hipHostFree(h_state); h_state = nullptr; }
        if (h_flag) { (void)hipHostFree(h_flag); h_flag = nullptr; }
        if (h_scratch) { (void)hipHostFree(h_scratch); h_scratch = nullptr; }
        if (h_rho) { (void)hipHostFree(h_rho); h_rho = nullptr; }
        have_graph_data = false;
    }
    // The staging buffer survives release(): it is sized by the largest table seen and reused across requests.
    int stage_reserve(size_t n) {
        if (n <= stage_cap) return 0;
        if (stage) { (void)hipHostFree(stage); stage = nullptr; stage_cap = 0; }
        n += n / 4;                               // headroom for a growing graph: pinning memory is the expensive part
        HIP_OK(hipHostMalloc((void**)&stage, n * sizeof(T)));
        stage_cap = n;
        return 0;
    }

    int init() {
        HIP_OK(hipSetDevice(cfg.device));
        HIP_OK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        HIP_OK(hipStreamCreateWithFlags(&stream2, hipStreamNonBlocking));
        for (auto& e : ev) HIP_OK(hipEventCreate(&e));
        return 0;
    }

    double ms_in_malloc = 0; int n_malloc = 0;
    // Every copy and fill goes through THIS engine's stream (never the legacy default stream): several engines serve
    // requests from different threads of one process, and a legacy-stream call in one thread is refused by the runtime
    // while another thread captures a graph.
    int copy_sync(void* dst, const void* src, size_t bytes, hipMemcpyKind kind) {
        HIP_OK(hipMemcpyAsync(dst, src, bytes, kind, cs()));
        HIP_OK(hipStreamSynchronize(cs()));
        return 0;
    }
    int fill_zero(void* dst, size_t bytes) { HIP_OK(hipMemsetAsync(dst, 0, bytes, cs())); return 0; }
    template <typename U> int dalloc(U** out, size_t n) {
        std::lock_guard<std::mutex> guard(slab_mu);
        const size_t bytes = (std::max<size_t>(n, 1) * sizeof(U) + 255) & ~size_t(255);
        for (Slab& sl : slabs)
            if (sl.used + bytes <= sl.cap) { *out = (U*)(sl.base + sl.used); sl.used += bytes; return 0; }
        const size_t cap = std::max(bytes, std::min(kSlabMax, std::max(kSlabMin, slab_total)));
        void* ptr = nullptr;
        const auto t0 = std::chrono::steady_clock::now();
        HIP_OK(hipMalloc(&ptr, cap));
        ms_in_malloc += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); ++n_malloc;
        slabs.push_back({(char*)ptr, cap, bytes}); slab_total += cap;
        *out = (U*)ptr;
        return 0;
    }
    int upload_T(T** out, const double* src, size_t n) {
        if (int rc = dalloc(out, n)) return rc;
        if (n == 0) return 0;
        std::vector<T> tmp(n);
        for (size_t k = 0; k < n; ++k) tmp[k] = (T)src[k];
        { if (int rc_ = copy_sync(*out, tmp.data(), n * sizeof(T), hipMemcpyHostToDevice)) return rc_; }
        return 0;
    }
    int upload_u32(const uint32_t** out, const std::vector<uint32_t>& v) {
        uint32_t* d = nullptr;
        if (int rc = dalloc(&d, v.size())) return rc;
        if (!v.empty()) { if (int rc_ = copy_sync(d, v.data(), v.size() * sizeof(uint32_t), hipMemcpyHostToDevice)) return rc_; }
        *out = d;
        return 0;
    }
    int upload_u32m(uint32_t** out, const std::vector<uint32_t>& v) {
        if (int rc = dalloc(out, v.size())) return rc;
        if (!v.empty()) { if (int rc_ = copy_sync(*out, v.data(), v.size() * sizeof(uint32_t), hipMemcpyHostToDevice)) return rc_; }
        return 0;
    }
    static std::vector<int> rows_of(const BlockCsr& m) {
        std::vector<int> r(m.col.size());
        for (int i = 0; i < m.n_rows; ++i) for (int a = m.ptr[i]; a < m.ptr[i + 1]; ++a) r[a] = i;
        return r;
    }
    // The hierarchy's patterns are built by a host thread (host/amg.cpp) while this thread uploads state and slot tables —
    // and then the finished parts of the hierarchy itself: the contribution lists of level 0 as soon as they exist, every
    // level as soon as it is complete (a level's gather lists are tens of MB of pageable memory: their copies hide behind
    // the symbolic work on the next level).  Sharded runs remap the lists after the build and upload everything at the end.
    std::thread amg_builder;
    std::string amg_builder_error;
    std::mutex amg_mu; std::condition_variable amg_cv;
    bool amg_schur_ready = false, amg_finished = false; int amg_levels_ready = 0;
    bool device_products = true;     // research switch TSGO_HOST_PRODUCTS=1: every pair list on the host, as rounds 1-2
    int sym_decline = 0;             // test hook TSGO_SYM_DECLINE (bits): the device builders behave as if a row had overflowed their LDS tables —
                                     // 1: level 0, 2: T = A P of every level, 4: A' = R T of every odd level — so that the hand-back to the host runs
    void start_amg_builder(const tsgo_graph& g) {
        amg_schur_ready = amg_finished = false; amg_levels_ready = 0;
        amg_builder = std::thread([this, &g] {      // g is borrowed for the whole tsgo_set_graph call, which joins this thread
            const auto t0 = std::chrono::steady_clock::now();
            (void)hipSetDevice(cfg.device);
            t_builder_stream = stream2;
            AmgProgress pg;
            pg.schur_ready = [this] { { std::lock_guard<std::mutex> l(amg_mu); amg_schur_ready = true; } amg_cv.notify_all(); };
            pg.level_ready = [this](int n) { { std::lock_guard<std::mutex> l(amg_mu); amg_levels_ready = n; } amg_cv.notify_all(); };
            if (device_products && pr.world == 1)
                pg.schur = [this](const SchurCsr& in, BlockCsr& A0, std::vector<int>& sc_ptr_h, std::vector<int>& sc_od_ptr_h, std::string& err) -> bool {
                    bool accepted = false;
                    if (run_device_schur(in, A0, sc_ptr_h, sc_od_ptr_h, &accepted)) { err = last_error(); return false; }
                    return accepted;
                };
            if (device_products && pr.world == 1)
                pg.products = [this](int level, AmgLevel& L, BlockCsr& A_next, std::string& err) -> bool {       // on THIS thread, on its own stream
                    bool accepted = false;
                    if (level < 0 || level >= (int)lv.size()) return false;
                    if (run_device_products(level, L, A_next, &accepted)) { err = last_error(); return false; }
                    return accepted;
                };
            amg_builder_error = pr.world > 1 ? build_amg_sharded(g, pr, amg) : build_amg(pr, amg, &pg);
            ms_amg_symbolic = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            t_builder_stream = nullptr;
            { std::lock_guard<std::mutex> l(amg_mu); amg_finished = true; }
            amg_cv.notify_all();
        });
    }
    template <typename A> int upload_i32(int** out, const std::vector<int, A>& v) {
        if (int rc = dalloc(out, v.size())) return rc;
        if (!v.empty()) { if (int rc_ = copy_sync(*out, v.data(), v.size() * sizeof(int), hipMemcpyHostToDevice)) return rc_; }
        return 0;
    }
#define UP(dst, vec) if (int rc = upload_i32(&dst, vec)) return rc
    bool schur_dev = false;      // level 0's pattern and contribution lists were built on the device (run_device_schur)
    int upload_schur_lists() {
        if (schur_dev) return 0;
        UP(sc_ptr, amg.schur.ptr); UP(sc_optr, amg.schur.od_ptr);
        if (int rc = upload_u32m(&sc_si, amg.schur.slot_i)) return rc;
        if (int rc = upload_u32m(&sc_sk, amg.schur.slot_k)) return rc;
        if (int rc = upload_u32m(&sc_os, amg.schur.od_slot)) return rc;
        return 0;
    }
    int upload_level(size_t l) {
        const AmgLevel& L = amg.levels[l]; DevLevel<T>& D = lv[l];
        D.n = L.n; D.n_agg = L.n_agg; D.nnzA = L.A.nnz(); D.nnzP = L.P.nnz();
        if (!D.patterns_up) { UP(D.A_ptr, L.A.ptr); UP(D.A_col, L.A.col); UP(D.P_ptr, L.P.ptr); UP(D.P_col, L.P.col); UP(D.R_ptr, L.R.ptr); UP(D.R_col, L.R.col); UP(D.r_to_p, L.r_to_p); }
        UP(D.A_row, rows_of(L.A)); UP(D.diag, L.diag);
        UP(D.P_row, rows_of(L.P)); UP(D.p_self, L.p_self);
        UP(D.ps_ptr, L.p_src.ptr); UP(D.ps_x, L.p_src.x); UP(D.ps_y, L.p_src.y);
        { std::vector<int> inv(L.r_to_p.size()); for (size_t k = 0; k < inv.size(); ++k) inv[L.r_to_p[k]] = (int)k; UP(D.p_to_r, inv); }
        if (!D.products_dev) {       // the host built the two pair-list products (sharded runs, TSGO_HOST_PRODUCTS, a row too dense for the device tables)
            D.nnzT = L.T.nnz(); D.nnzNext = (int)L.a_src.ptr.size() - 1;
            D.pairs_T = (double)L.t_src.x.size() / std::max(1, D.nnzT); D.pairs_A = (double)L.a_src.x.size() / std::max<double>(1, (double)std::count(L.a_mirror.begin(), L.a_mirror.end(), -1));
            UP(D.ts_ptr, L.t_src.ptr); UP(D.ts_x, L.t_src.x); UP(D.ts_y, L.t_src.y); UP(D.T_ptr, L.T.ptr); UP(D.T_col, L.T.col);
            UP(D.as_ptr, L.a_src.ptr); UP(D.as_x, L.a_src.x); UP(D.as_y, L.a_src.y); UP(D.as_mirror, L.a_mirror);
            { std::vector<int> up; for (int b = 0; b < (int)L.a_mirror.size(); ++b) if (L.a_mirror[b] < 0) up.push_back(b); D.n_upper = (int)up.size(); UP(D.as_upper, up); }
        }
        if (int rc = upload_T(&D.rel, L.rel.data(), L.rel.size())) return rc;
        if (int rc = dalloc(&D.A, (size_t)D.nnzA * 9)) return rc;
        if (int rc = dalloc(&D.Dinv, (size_t)D.n * 9)) return rc;
        if (int rc = dalloc(&D.P, (size_t)D.nnzP * 9)) return rc;
        if (int rc = dalloc(&D.Rv, (size_t)D.nnzP * 9)) return rc;
        if (int rc = dalloc(&D.Tv, (size_t)D.nnzT * 9)) return rc;
        // sized for the f32 form (9 words per block): a handle that finds its graph too ill-conditioned for the packed halves
        // switches the cycle's copies to f32 in place (do_solve)
        if (int rc = dalloc(&D.Ppm, (size_t)D.nnzP * kCyWordsF32)) return rc;
        if (int rc = dalloc(&D.Rpm, (size_t)D.nnzP * kCyWordsF32)) return rc;
        if (l == 0 && explicit0) { if (int rc = dalloc(&D.Apm, (size_t)D.nnzA * kCyWordsF32)) return rc; }
        if (l > 0) {
            if (int rc = dalloc(&D.Apm, (size_t)D.nnzA * kCyWordsF32)) return rc;
            if (int rc = dalloc(&D.r, (size_t)D.n * 3)) return rc;
            if (int rc = dalloc(&D.z, (size_t)D.n * 3)) return rc;
            if (int rc = dalloc(&D.res, (size_t)D.n * 3)) return rc;
            if (int rc = dalloc(&D.z2, (size_t)D.n * 3)) return rc;
        }
        return 0;
    }
    // Level 0 on the device (tsgo_sym_kernels.h: k_s0_count / k_s0_fill): the pattern of the explicit Schur complement and, per block, the
    // pairs of LM-edge slots and the odometry slots it is summed from.  The pattern and the list offsets go back to the host builder (the
    // aggregation weighs a coupling by its number of contributions); the lists stay here.
    int run_device_schur(const SchurCsr& in, BlockCsr& A0, std::vector<int>& sc_ptr_h, std::vector<int>& sc_od_ptr_h, bool* accepted) {
        *accepted = false;
        const auto t0 = std::chrono::steady_clock::now();
        const int P = in.P;
        int *pp_ptr, *pp_lm, *obs_ptr, *obs_pose, *od_ptr, *od_col; uint32_t *pp_slot, *obs_slot, *od_slot;
        UP(pp_ptr, in.pp_ptr); UP(pp_lm, in.pp_lm); UP(obs_ptr, in.obs_ptr); UP(obs_pose, in.obs_pose); UP(od_ptr, in.od_ptr); UP(od_col, in.od_col);
        if (int rc = upload_u32m(&pp_slot, in.pp_slot)) return rc;
        if (int rc = upload_u32m(&obs_slot, in.obs_slot)) return rc;
        if (int rc = upload_u32m(&od_slot, in.od_slot)) return rc;
        int *d, *m, *mo, *zptr, *poff, *ooff, *flags;
        if (int rc = dalloc(&d, (size_t)P)) return rc;
        if (int rc = dalloc(&m, (size_t)P)) return rc;
        if (int rc = dalloc(&mo, (size_t)P)) return rc;
        if (int rc = dalloc(&zptr, (size_t)P + 1)) return rc;
        if (int rc = dalloc(&poff, (size_t)P + 1)) return rc;
        if (int rc = dalloc(&ooff, (size_t)P + 1)) return rc;
        if (int rc = dalloc(&flags, 4)) return rc;
        if (int rc = fill_zero(flags, 4 * sizeof(int))) return rc;
        hipLaunchKernelGGL(k_s0_count, dim3(P), dim3(kSymWave), 0, cs(), P, (const int*)pp_ptr, (const int*)pp_lm, (const int*)obs_ptr, (const int*)obs_pose, (const int*)od_ptr,
                           (const int*)od_col, in.max_pair_degree, d, m, mo, flags);
        hipLaunchKernelGGL(k_sym_scan, dim3(1), dim3(1024), 0, cs(), P, (const int*)d, zptr);
        hipLaunchKernelGGL(k_sym_scan, dim3(1), dim3(1024), 0, cs(), P, (const int*)m, poff);
        hipLaunchKernelGGL(k_sym_scan, dim3(1), dim3(1024), 0, cs(), P, (const int*)mo, ooff);
        int h4[4];
        HIP_OK(hipMemcpyAsync(&h4[0], zptr + P, sizeof(int), hipMemcpyDeviceToHost, cs()));
        HIP_OK(hipMemcpyAsync(&h4[1], poff + P, sizeof(int), hipMemcpyDeviceToHost, cs()));
        HIP_OK(hipMemcpyAsync(&h4[2], ooff + P, sizeof(int), hipMemcpyDeviceToHost, cs()));
        HIP_OK(hipMemcpyAsync(&h4[3], flags, sizeof(int), hipMemcpyDeviceToHost, cs()));
        HIP_OK(hipStreamSynchronize(cs()));
        if (h4[3] || (sym_decline & 1)) return 0;      // a pose couples to too many others for the LDS tables: the host builds level 0
        const int nnz = h4[0], n_pairs = h4[1], n_od = h4[2];
        int* zcol = nullptr;
        if (int rc = dalloc(&zcol, (size_t)nnz)) return rc;
        if (int rc = dalloc(&sc_ptr, (size_t)nnz + 1)) return rc;
        if (int rc = dalloc(&sc_optr, (size_t)nnz + 1)) return rc;
        if (int rc = dalloc(&sc_si, (size_t)n_pairs)) return rc;
        if (int rc = dalloc(&sc_sk, (size_t)n_pairs)) return rc;
        if (int rc = dalloc(&sc_os, (size_t)n_od)) return rc;
        hipLaunchKernelGGL(k_s0_fill, dim3(P), dim3(kSymWave), 0, cs(), P, (const int*)pp_ptr, (const int*)pp_lm, (const uint32_t*)pp_slot, (const int*)obs_ptr, (const int*)obs_pose,
                           (const uint32_t*)obs_slot, (const int*)od_ptr, (const int*)od_col, (const uint32_t*)od_slot, in.max_pair_degree, (const int*)zptr, (const int*)poff,
                           (const int*)ooff, zcol, sc_ptr, sc_optr, sc_si, sc_sk, sc_os);
        HIP_OK(hipMemcpyAsync(sc_ptr + nnz, &n_pairs, sizeof(int), hipMemcpyHostToDevice, cs()));
        HIP_OK(hipMemcpyAsync(sc_optr + nnz, &n_od, sizeof(int), hipMemcpyHostToDevice, cs()));
        A0.n_rows = A0.n_cols = P;
        A0.ptr.resize((size_t)P + 1); A0.col.resize((size_t)nnz); sc_ptr_h.resize((size_t)nnz + 1); sc_od_ptr_h.resize((size_t)nnz + 1);
        HIP_OK(hipMemcpyAsync(A0.ptr.data(), zptr, ((size_t)P + 1) * sizeof(int), hipMemcpyDeviceToHost, cs()));
        if (nnz) HIP_OK(hipMemcpyAsync(A0.col.data(), zcol, (size_t)nnz * sizeof(int), hipMemcpyDeviceToHost, cs()));
        HIP_OK(hipMemcpyAsync(sc_ptr_h.data(), sc_ptr, ((size_t)nnz + 1) * sizeof(int), hipMemcpyDeviceToHost, cs()));
        HIP_OK(hipMemcpyAsync(sc_od_ptr_h.data(), sc_optr, ((size_t)nnz + 1) * sizeof(int), hipMemcpyDeviceToHost, cs()));
        HIP_OK(hipStreamSynchronize(cs()));
        schur_dev = true;
        *accepted = true;
        ms_device_products += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        return 0;
    }
    // One level's two pair-list products on the device (tsgo_sym_kernels.h): T = A P with its lists, A' = R T (upper blocks
    // listed, lower ones mirrored) with its lists; the pattern of A' goes back to the host builder, which needs it for the next
    // level.  *accepted = false: a row was too dense for the LDS tables — the host builds this level's lists itself.
    double ms_device_products = 0;
    int run_device_products(int l, AmgLevel& L, BlockCsr& A_next, bool* accepted) {
        *accepted = false;
        const auto t0 = std::chrono::steady_clock::now();
        DevLevel<T>& D = lv[l];
        UP(D.A_ptr, L.A.ptr); UP(D.A_col, L.A.col); UP(D.P_ptr, L.P.ptr); UP(D.P_col, L.P.col); UP(D.R_ptr, L.R.ptr); UP(D.R_col, L.R.col); UP(D.r_to_p, L.r_to_p);
        D.patterns_up = true;
        const int n = L.n, na = L.n_agg;
        int *d = nullptr, *m = nullptr, *tptr = nullptr, *tpoff = nullptr, *flags = nullptr;
        if (int rc = dalloc(&d, (size_t)std::max(n, na))) return rc;
        if (int rc = dalloc(&m, (size_t)std::max(n, na))) return rc;
        if (int rc = dalloc(&tptr, (size_t)n + 1)) return rc;
        if (int rc = dalloc(&tpoff, (size_t)n + 1)) return rc;
        if (int rc = dalloc(&flags, 4)) return rc;
        if (int rc = fill_zero(flags, 4 * sizeof(int))) return rc;
        int h3[3];
        // ---- T = A P
        hipLaunchKernelGGL((k_sym_count<0>), dim3(n), dim3(kSymWave), 0, cs(), n, (const int*)D.A_ptr, (const int*)D.A_col, (const int*)D.P_ptr, (const int*)D.P_col, d, m, flags);
        hipLaunchKernelGGL(k_sym_scan, dim3(1), dim3(1024), 0, cs(), n, (const int*)d, tptr);
        hipLaunchKernelGGL(k_sym_scan, dim3(1), dim3(1024), 0, cs(), n, (const int*)m, tpoff);
        HIP_OK(hipMemcpyAsync(&h3[0], tptr + n, sizeof(int), hipMemcpyDeviceToHost, cs()));
        HIP_OK(hipMemcpyAsync(&h3[1], tpoff + n, sizeof(int), hipMemcpyDeviceToHost, cs()));
        HIP_OK(hipMemcpyAsync(&h3[2], flags, sizeof(int), hipMemcpyDeviceToHost, cs()));
        HIP_OK(hipStreamSynchronize(cs()));
        if (h3[2] || (sym_decline & 2)) return 0;      // declined
        const int nnzT = h3[0], pairsT = h3[1];
        int* tcol = nullptr;
        if (int rc = dalloc(&tcol, (size_t)nnzT)) return rc;
        if (int rc = dalloc(&D.ts_ptr, (size_t)nnzT + 1)) return rc;
        if (int rc = dalloc(&D.ts_x, (size_t)pairsT)) return rc;
        if (int rc = dalloc(&D.ts_y, (size_t)pairsT)) return rc;
        hipLaunchKernelGGL((k_sym_fill<0>), dim3(n), dim3(kSymWave), 0, cs(), n, (const int*)D.A_ptr, (const int*)D.A_col, (const int*)nullptr, (const int*)D.P_ptr, (const int*)D.P_col,
                           (const int*)tptr, (const int*)tpoff, tcol, D.ts_ptr, D.ts_x, D.ts_y, (int*)nullptr);
        HIP_OK(hipMemcpyAsync(D.ts_ptr + nnzT, &pairsT, sizeof(int), hipMemcpyHostToDevice, cs()));
        // ---- A' = R T, upper blocks listed
        int *zptr = nullptr, *zpoff = nullptr, *nup = nullptr, *uoff = nullptr;
        if (int rc = dalloc(&zptr, (size_t)na + 1)) return rc;
        if (int rc = dalloc(&zpoff, (size_t)na + 1)) return rc;
        if (int rc = dalloc(&nup, (size_t)na)) return rc;
        if (int rc = dalloc(&uoff, (size_t)na + 1)) return rc;
        hipLaunchKernelGGL((k_sym_count<1>), dim3(na), dim3(kSymWave), 0, cs(), na, (const int*)D.R_ptr, (const int*)D.R_col, (const int*)tptr, (const int*)tcol, d, m, flags + 1);
        hipLaunchKernelGGL(k_sym_scan, dim3(1), dim3(1024), 0, cs(), na, (const int*)d, zptr);
        hipLaunchKernelGGL(k_sym_scan, dim3(1), dim3(1024), 0, cs(), na, (const int*)m, zpoff);
        HIP_OK(hipMemcpyAsync(&h3[0], zptr + na, sizeof(int), hipMemcpyDeviceToHost, cs()));
        HIP_OK(hipMemcpyAsync(&h3[1], zpoff + na, sizeof(int), hipMemcpyDeviceToHost, cs()));
        HIP_OK(hipMemcpyAsync(&h3[2], flags + 1, sizeof(int), hipMemcpyDeviceToHost, cs()));
        HIP_OK(hipStreamSynchronize(cs()));
        if (h3[2] || ((sym_decline & 4) && (l & 1))) { D.ts_ptr = D.ts_x = D.ts_y = nullptr; return 0; }       // declined (the slab bytes of T's lists are lost until the next structure)
        const int nnzN = h3[0], pairsA = h3[1];
        int* zcol = nullptr;
        if (int rc = dalloc(&zcol, (size_t)nnzN)) return rc;
        if (int rc = dalloc(&D.as_ptr, (size_t)nnzN + 1)) return rc;
        if (int rc = dalloc(&D.as_x, (size_t)pairsA)) return rc;
        if (int rc = dalloc(&D.as_y, (size_t)pairsA)) return rc;
        if (int rc = dalloc(&D.as_mirror, (size_t)nnzN)) return rc;
        if (int rc = dalloc(&D.as_upper, (size_t)nnzN)) return rc;
        hipLaunchKernelGGL((k_sym_fill<1>), dim3(na), dim3(kSymWave), 0, cs(), na, (const int*)D.R_ptr, (const int*)D.R_col, (const int*)D.r_to_p, (const int*)tptr, (const int*)tcol,
                           (const int*)zptr, (const int*)zpoff, zcol, D.as_ptr, D.as_x, D.as_y, nup);
        HIP_OK(hipMemcpyAsync(D.as_ptr + nnzN, &pairsA, sizeof(int), hipMemcpyHostToDevice, cs()));
        hipLaunchKernelGGL(k_sym_scan, dim3(1), dim3(1024), 0, cs(), na, (const int*)nup, uoff);
        hipLaunchKernelGGL(k_sym_mirror, dim3((na + 255) / 256), dim3(256), 0, cs(), na, (const int*)zptr, (const int*)zcol, (const int*)uoff, D.as_mirror, D.as_upper, flags + 2);
        A_next.n_rows = A_next.n_cols = na;
        A_next.ptr.resize((size_t)na + 1); A_next.col.resize((size_t)nnzN);
        HIP_OK(hipMemcpyAsync(A_next.ptr.data(), zptr, ((size_t)na + 1) * sizeof(int), hipMemcpyDeviceToHost, cs()));
        if (nnzN) HIP_OK(hipMemcpyAsync(A_next.col.data(), zcol, (size_t)nnzN * sizeof(int), hipMemcpyDeviceToHost, cs()));
        HIP_OK(hipMemcpyAsync(&h3[0], uoff + na, sizeof(int), hipMemcpyDeviceToHost, cs()));
        HIP_OK(hipMemcpyAsync(&h3[1], flags + 2, sizeof(int), hipMemcpyDeviceToHost, cs()));
        HIP_OK(hipStreamSynchronize(cs()));
        if (h3[1]) return set_error(-2, "tsgo_set_graph: the Galerkin pattern is not structurally symmetric");
        D.n_upper = h3[0]; D.nnzT = nnzT; D.nnzNext = nnzN; D.T_ptr = tptr; D.T_col = tcol;
        D.pairs_T = (double)pairsT / std::max(1, nnzT); D.pairs_A = (double)pairsA / std::max(1, D.n_upper);
        D.products_dev = true;
        *accepted = true;
        ms_device_products += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        return 0;
    }
    int upload_amg() {
        size_t done = 0; bool schur_done = false;
        if (pr.world == 1) {                     // consume what the builder has finished while it works on the rest
            for (;;) {
                int ready; bool schur, fin;
                {
                    std::unique_lock<std::mutex> l(amg_mu);
                    amg_cv.wait(l, [&] { return amg_finished || (int)done < amg_levels_ready || (!schur_done && amg_schur_ready); });
                    ready = amg_levels_ready; schur = amg_schur_ready; fin = amg_finished;
                }
                if (fin) break;                  // whatever is left is uploaded after the join (and errors are looked at there)
                if (schur && !schur_done) { if (int rc = upload_schur_lists()) return rc; schur_done = true; }
                for (; (int)done < ready; ++done) if (int rc = upload_level(done)) return rc;
            }
        }
        if (amg_builder.joinable()) amg_builder.join();
        if (!amg_builder_error.empty()) return set_error(-2, "tsgo_set_graph: " + amg_builder_error);
        if (amg.levels.size() > 16) return set_error(-2, "tsgo_set_graph: too many multigrid levels");
        if (!schur_done) { if (int rc = upload_schur_lists()) return rc; }
        for (; done < amg.levels.size(); ++done) if (int rc = upload_level(done)) return rc;
        lv.resize(amg.levels.size());
        UP(last_ptr, amg.A_last.ptr); UP(last_col, amg.A_last.col);
        nb_last = amg.A_last.n_rows; nnz_last = amg.A_last.nnz();
        if (nb_last * 3 > kDenseMax) return set_error(-2, "tsgo_set_graph: coarsest multigrid level too large");
        if (int rc = dalloc(&A_last, (size_t)nnz_last * 9)) return rc;
        if (int rc = dalloc(&inv_last, (size_t)nb_last * 3 * nb_last * 3)) return rc;
        if (int rc = dalloc(&r_last, (size_t)nb_last * 3)) return rc;
        if (int rc = dalloc(&z_last, (size_t)nb_last * 3)) return rc;
        if (int rc = dalloc(&rzpart, (size_t)nbP)) return rc;
        if (int rc = dalloc(&omega_dev, 16)) return rc;
        if (int rc = dalloc(&pw_a, (size_t)pr.P * 3)) return rc;
        if (int rc = dalloc(&pw_b, (size_t)pr.P * 3)) return rc;
        if (int rc = dalloc(&rho_part, 16 * 2 * kRhoBlocks)) return rc;
        if (h_rho) (void)hipHostFree(h_rho);
        HIP_OK(hipHostMalloc((void**)&h_rho, sizeof(T) * 16 * 2 * kRhoBlocks));
        lin_count = 0; hier_age = -1;
        {   // S is symmetric: k_schur_blocks sums the blocks on and above the diagonal, the others are mirrored (half of the set-up's most
            // expensive gather: 137 us per hierarchy build at 100k poses)
            DevLevel<T>& L0 = lv[0];
            int *nup = nullptr, *uoff = nullptr, *flag = nullptr;
            n_upper0 = -1;
            if (int rc = dalloc(&nup, (size_t)L0.n)) return rc;
            if (int rc = dalloc(&uoff, (size_t)L0.n + 1)) return rc;
            if (int rc = dalloc(&flag, 4)) return rc;
            if (int rc = dalloc(&mirror0, (size_t)L0.nnzA)) return rc;
            if (int rc = dalloc(&upper0, (size_t)L0.nnzA)) return rc;
            if (int rc = fill_zero(flag, 4 * sizeof(int))) return rc;
            hipLaunchKernelGGL(k_count_upper, dim3((L0.n + 255) / 256), dim3(256), 0, cs(), L0.n, (const int*)L0.A_ptr, (const int*)L0.A_col, nup);
            hipLaunchKernelGGL(k_sym_scan, dim3(1), dim3(1024), 0, cs(), L0.n, (const int*)nup, uoff);
            hipLaunchKernelGGL(k_sym_mirror, dim3((L0.n + 255) / 256), dim3(256), 0, cs(), L0.n, (const int*)L0.A_ptr, (const int*)L0.A_col, (const int*)uoff, mirror0, upper0, flag);
            if (int rc = dalloc(&lower0, (size_t)L0.nnzA)) return rc;
            HIP_OK(hipMemsetAsync(lower0, 0xff, sizeof(int) * (size_t)L0.nnzA, cs()));
            hipLaunchKernelGGL(k_invert_mirror, dim3((L0.nnzA + 255) / 256), dim3(256), 0, cs(), L0.nnzA, (const int*)mirror0, lower0);
            int h2[2] = {0, 0};
            HIP_OK(hipMemcpyAsync(&h2[0], uoff + L0.n, sizeof(int), hipMemcpyDeviceToHost, cs()));
            HIP_OK(hipMemcpyAsync(&h2[1], flag, sizeof(int), hipMemcpyDeviceToHost, cs()));
            HIP_OK(hipStreamSynchronize(cs()));
            if (!h2[1]) n_upper0 = h2[0];      // (a pattern that is not structurally symmetric — it always is — keeps every block summed)
        }
        // the bottom of the cycle as one dense operator (tsgo_amg_kernels.h: k_bottom_*): the last explicit level when it is small enough
        // and runs V(1,1); and the level above it in factored form (k_tail_*) when that one is small too
        const size_t nl = lv.size();
        bottom_dense = nl >= 2 && lv.back().n * 4 <= kDenseThreads && nb_last > 0 && nu_at(nl - 1) == 1;
        tail2 = bottom_dense && nl >= 3 && lv[nl - 2].n <= kSmallLevelRows && nu_at(nl - 2) == 1 && (size_t)lv[nl - 2].n * 3 * (size_t)lv.back().n * 3 <= (size_t(4) << 20);
        if (bottom_dense) {
            const size_t n3 = (size_t)lv.back().n * 3, nd = (size_t)nb_last * 3;
            if (int rc = dalloc(&bot_S, n3 * n3)) return rc;
            if (int rc = dalloc(&bot_B, n3 * n3)) return rc;
            if (int rc = dalloc(&bot_P, n3 * nd)) return rc;
            if (int rc = dalloc(&bot_E, n3 * nd)) return rc;
            if (int rc = dalloc(&bot_F, n3 * nd)) return rc;
            if (int rc = dalloc(&bot_Bf, n3 * n3)) return rc;
        }
        if (tail2) {
            const size_t n3 = (size_t)lv[nl - 2].n * 3, nd = (size_t)lv.back().n * 3;
            if (int rc = dalloc(&tail_Ef, n3 * nd)) return rc;
            if (int rc = dalloc(&tail_Etf, n3 * nd)) return rc;
            if (int rc = dalloc(&tail_Gf, n3 * nd)) return rc;
            if (int rc = dalloc(&tail_t, nd)) return rc;
        }
        return 0;
    }
    int n_upper0 = -1; int *mirror0 = nullptr, *upper0 = nullptr, *lower0 = nullptr;      // level 0: blocks on / above the diagonal, and the mirror of every block below it
    bool bottom_dense = false, tail2 = false;
    T *bot_S = nullptr, *bot_B = nullptr, *bot_P = nullptr, *bot_E = nullptr, *bot_F = nullptr; float* bot_Bf = nullptr;
    T* tail_t = nullptr; float *tail_Ef = nullptr, *tail_Etf = nullptr, *tail_Gf = nullptr;
    // B = W + S W + E C E^T of the last explicit level (after k_dense_inverse, and again whenever the level's damping changes), then
    // E = P - W (A P) and G = E B of the level above it
    int launch_bottom_setup() {
        if (!bottom_dense) return 0;
        auto tiles = [](int n) { return (unsigned)((n + 15) / 16); };
        {
            DevLevel<T>& L = lv.back();
            const int n3 = L.n * 3, nd = nb_last * 3;
            const T* om = omega_dev + (lv.size() - 1);
            HIP_OK(hipMemsetAsync(bot_S, 0, sizeof(T) * (size_t)n3 * n3, stream));
            HIP_OK(hipMemsetAsync(bot_P, 0, sizeof(T) * (size_t)n3 * nd, stream));
            hipLaunchKernelGGL((k_bottom_scatter<T>), dim3(grid_for(L.nnzA + L.nnzP)), dim3(kBlock), 0, stream, L.nnzA, (const int*)L.A_row, (const int*)L.A_col, (const H*)L.A, (const H*)L.Dinv, om,
                               L.nnzP, (const int*)L.P_row, (const int*)L.P_col, (const H*)L.P, n3, nd, bot_S, bot_P);
            hipLaunchKernelGGL((k_small_gemm<T, 0>), dim3(tiles(nd), tiles(n3)), dim3(256), 0, stream, n3, nd, n3, (const T*)bot_S, n3, (const T*)bot_P, nd, bot_E, nd);          // E = S P
            hipLaunchKernelGGL((k_small_gemm<T, 0>), dim3(tiles(nd), tiles(n3)), dim3(256), 0, stream, n3, nd, nd, (const T*)bot_E, nd, (const T*)inv_last, nd, bot_F, nd);      // F = E C
            hipLaunchKernelGGL((k_small_gemm<T, 1>), dim3(tiles(n3), tiles(n3)), dim3(256), 0, stream, n3, n3, nd, (const T*)bot_F, nd, (const T*)bot_E, nd, bot_B, n3);         // G = F E^T
            hipLaunchKernelGGL((k_bottom_finish<T>), dim3(grid_for(n3 * n3)), dim3(kBlock), 0, stream, n3, (const T*)bot_S, (const H*)L.Dinv, om, (const T*)bot_B, bot_Bf, (T*)nullptr);
        }
        if (tail2) {
            DevLevel<T>& L = lv[lv.size() - 2];
            const int n3 = L.n * 3, nd = lv.back().n * 3;
            const T* om = omega_dev + (lv.size() - 2);
            HIP_OK(hipMemsetAsync(tail_Ef, 0, sizeof(float) * (size_t)n3 * nd, stream));
            hipLaunchKernelGGL((k_scatter_blocks<T>), dim3(grid_for(L.nnzP)), dim3(kBlock), 0, stream, L.nnzP, (const int*)L.P_row, (const int*)L.P_col, (const H*)L.P, nd, tail_Ef);
            hipLaunchKernelGGL((k_tail_E<T>), dim3(grid_for(L.n, 64)), dim3(kBlock), 0, stream, L.n, (const int*)L.T_ptr, (const int*)L.T_col, (const H*)L.Tv, (const H*)L.Dinv, om, nd, tail_Ef);      // E = P - W (A P)
            hipLaunchKernelGGL(k_gemm_f32, dim3((nd + 63) / 64, (n3 + 31) / 32), dim3(256), 0, stream, n3, nd, nd, (const float*)tail_Ef, nd, (const float*)bot_Bf, nd, tail_Gf, nd);      // G = E B
            hipLaunchKernelGGL(k_transpose_f32, dim3(grid_for(n3 * nd)), dim3(kBlock), 0, stream, n3, nd, (const float*)tail_Ef, tail_Etf);
        }
        return 0;
    }
#undef UP

    // Device buffers of one slot table (indices uploaded, static planes allocated: they are staged separately).
    int alloc_table(Table<T>& t, T** st_out, const SellTable& h, int st_planes, int dyn_planes, bool pairs) {
        t.slots = h.slots(); t.n_slices = h.n_slices; t.n_vertices = h.n_vertices; t.xcd = cfg.xcd_map ? 1 : 0;
        if (int rc = upload_u32(&t.row_off, h.row_off)) return rc;
        if (int rc = upload_u32(&t.idx, h.idx)) return rc;
        if (int rc = dalloc(st_out, (size_t)st_planes * h.slots())) return rc;
        t.st = *st_out;
        if (int rc = dalloc(&t.dyn, (size_t)dyn_planes * h.slots())) return rc;
        t.dyn32 = nullptr;
        if (pairs) { if (int rc = dalloc(&t.dyn32, h.slots())) return rc; { if (int rc_ = fill_zero(t.dyn32, std::max<size_t>(h.slots(), 1) * sizeof(float4))) return rc_; } }
        { if (int rc_ = fill_zero(t.dyn, std::max<size_t>((size_t)dyn_planes * h.slots(), 1) * sizeof(T))) return rc_; }
        return 0;
    }

    // ---- what changes from request to request when the structure does not: vertex estimates, measurements, weights.
    // Everything is written ONCE, in the device's scalar type and plane order, into pinned memory and copied from there
    // (no vector<double> -> vector<T> -> pageable copy).  Used by the first request of a structure and by every refill,
    // so a refilled engine holds bit for bit what a fresh one would.
    //   LM tables  : pair-plane-major [(zx, zy) | (w0, w1)] per slot, padding 0          (host/problem.h: lm_static)
    //   ODOM table : nine planes, rows 0-1 of the inverse measurement (6) + weights (3)
    //   state      : ps (x, y, cos, sin), theta, lmrec (lx, ly, 0 ...)
    // host_state: also refresh the layout's own copy of the estimates (pr.pose_xyt / lm_xy).  Only a refill does: on the first
    // build build_problem has just stored the same values and the multigrid builder thread is reading them.
    int stage_values(const tsgo_graph& g, bool host_state) {
        const size_t Sp = pr.by_pose.slots(), Sl = pr.by_lm.slots(), So = pr.odom.slots();
        const size_t P = (size_t)pr.P, L = (size_t)pr.L;
        const size_t o_p = 0, o_l = o_p + 4 * Sp, o_o = o_l + 4 * Sl, o_ps = o_o + 9 * So, o_th = o_ps + 4 * P, o_lm = o_th + P, total = o_lm + (size_t)kLmRec * std::max<size_t>(L, 1);
        if (int rc = stage_reserve(total)) return rc;
        const size_t nE = (size_t)g.n_edges;
        const bool timing = stage_timing;
        auto t_last = std::chrono::steady_clock::now();
        auto lap = [&](const char* what) {
            if (!timing) return;
            const auto n = std::chrono::steady_clock::now();
            std::fprintf(stderr, "[stage] %-28s %7.2f ms\n", what, std::chrono::duration<double, std::milli>(n - t_last).count());
            t_last = n;
        };
        // per LM edge: its four static values, once (a cos and a sin each), then scattered into both groupings
        std::vector<double>& ev = lm_values; if (ev.size() < 4 * nE) ev.resize(4 * nE);      // grow-only across requests
        parallel_chunks((int)nE, [&](int, int b, int e) {
            for (int k = b; k < e; ++k) if (g.e_type[k] == 1) lm_static(g.e_meas + 9 * (size_t)k, g.e_inf + 3 * (size_t)k, &ev[4 * (size_t)k]);
        });
        auto lm_table = [&](const SellTable& h, T* dst) {
            const size_t S = h.slots();
            parallel_chunks((int)S, [&](int, int b, int e) {
                for (int k = b; k < e; ++k) {
                    const uint32_t ed = h.edge[k];
                    const double* v = ed == kNoEdge ? nullptr : &ev[4 * (size_t)ed];
                    dst[2 * (size_t)k] = v ? (T)v[LM_ZX] : T(0); dst[2 * (size_t)k + 1] = v ? (T)v[LM_ZY] : T(0);
                    dst[2 * (S + (size_t)k)] = v ? (T)v[LM_W0] : T(0); dst[2 * (S + (size_t)k) + 1] = v ? (T)v[LM_W1] : T(0);
                }
            });
        };
        lap("per-edge LM values");
        lm_table(pr.by_pose, stage + o_p);
        lm_table(pr.by_lm, stage + o_l);
        lap("LM tables into staging");
        std::atomic<int> bad_edge{INT32_MAX};
        parallel_chunks((int)So, [&](int, int b, int e) {
            T* dst = stage + o_o;
            for (size_t k = (size_t)b; k < (size_t)e; ++k) {
                const uint32_t ed = pr.odom.edge[k];
                if (ed == kNoEdge) { for (int m = 0; m < 9; ++m) dst[(size_t)m * So + k] = T(0); continue; }
                if (pr.odom.idx[k] & kVlmBit) {      // virtual landmark measurement: the slot's own and the neighbour's local point, two weights
                    double v9[9]; vlm_static(g.e_meas + 9 * (size_t)ed, g.e_inf + 3 * (size_t)ed, (pr.odom.idx[k] & kDirBit) ? 1 : 0, v9);
                    for (int m = 0; m < 9; ++m) dst[(size_t)m * So + k] = (T)v9[m];
                    continue;
                }
                double inv[9];
                if (!invert3(g.e_meas + 9 * (size_t)ed, inv)) { int seen = bad_edge.load(); while ((int)ed < seen && !bad_edge.compare_exchange_weak(seen, (int)ed)) {} continue; }
                for (int m = 0; m < 6; ++m) dst[(size_t)(OD_MI0 + m) * So + k] = (T)inv[m];
                for (int m = 0; m < 3; ++m) dst[(size_t)(OD_W0 + m) * So + k] = (T)g.e_inf[3 * (size_t)ed + m];
            }
        }, 4096);
        if (bad_edge.load() != INT32_MAX) return set_error(-2, "tsgo_set_graph: ODOM edge " + std::to_string(bad_edge.load()) + " has a singular measurement matrix");
        parallel_chunks((int)P, [&](int, int b, int e) {
            for (size_t i = (size_t)b; i < (size_t)e; ++i) {
                const double* v = g.v_pos + 3 * (size_t)pr.pose_vertex[i];
                if (host_state) { pr.pose_xyt[3 * i] = v[0]; pr.pose_xyt[3 * i + 1] = v[1]; pr.pose_xyt[3 * i + 2] = v[2]; }
                stage[o_ps + 4 * i] = (T)v[0]; stage[o_ps + 4 * i + 1] = (T)v[1]; stage[o_ps + 4 * i + 2] = (T)std::cos(v[2]); stage[o_ps + 4 * i + 3] = (T)std::sin(v[2]);
                stage[o_th + i] = (T)v[2];
            }
        }, 4096);
        if (L == 0) std::fill(stage + o_lm, stage + o_lm + (size_t)kLmRec, T(0));
        parallel_chunks((int)L, [&](int, int b, int e) {
            for (size_t l = (size_t)b; l < (size_t)e; ++l) {
                const double* v = g.v_pos + 3 * (size_t)pr.lm_vertex[l];
                if (host_state) { pr.lm_xy[2 * l] = v[0]; pr.lm_xy[2 * l + 1] = v[1]; }
                T* rec = stage + o_lm + l * kLmRec;
                rec[0] = (T)v[0]; rec[1] = (T)v[1];
                for (int m = 2; m < kLmRec; ++m) rec[m] = T(0);
            }
        }, 4096);
        lap("ODOM planes, state");
        auto put = [&](T* dst, size_t off, size_t n) -> int { if (n) HIP_OK(hipMemcpyAsync(dst, stage + off, n * sizeof(T), hipMemcpyHostToDevice, stream)); return 0; };
        if (int rc = put(st_p, o_p, 4 * Sp)) return rc;
        if (int rc = put(st_l, o_l, 4 * Sl)) return rc;
        if (int rc = put(st_o, o_o, 9 * So)) return rc;
        if (int rc = put(ps, o_ps, 4 * P)) return rc;
        if (int rc = put(theta, o_th, P)) return rc;
        if (int rc = put(lmrec, o_lm, (size_t)kLmRec * std::max<size_t>(L, 1))) return rc;
        if (timing) { HIP_OK(hipStreamSynchronize(stream)); lap("copies to the device (waited for)"); }
        return 0;       // the caller synchronises the stream before the staging buffer is touched again
    }

    // solver state a fresh engine starts from: whatever was learnt on the previous graph must not leak into this one
    int reset_solver_state() {
        have_prev = false; n_prev = 0; n_tested = 0; carried = false; predicted_cg = 0; n_decided = 0; n_slow_seen = 0; n_host_slow = 0; ref_us_per_iter = 0; n_paced_slow = 0; std::fill(iters_by_age, iters_by_age + kAgeSlots, 0); lin_count = 0; n_lins = 0; hier_age = -1; iters_fresh = 0; iters_last = 0;
        const T one = 1;
        { if (int rc_ = copy_sync(one_dev, &one, sizeof(T), hipMemcpyHostToDevice)) return rc_; }
        { if (int rc_ = copy_sync(gscale_dev, &one, sizeof(T), hipMemcpyHostToDevice)) return rc_; }
        if (amg_on) {
            std::vector<T> init(16, (T)kSmootherOmega); init[0] = (T)kSmoother0Omega;
            { if (int rc_ = copy_sync(omega_dev, init.data(), 16 * sizeof(T), hipMemcpyHostToDevice)) return rc_; }
            omega_host.assign(lv.size(), kSmootherOmega); if (!omega_host.empty()) omega_host[0] = kSmoother0Omega;
        }
        return 0;
    }

    // Same structure as the graph the tables were built for: refill values, keep everything else.
    // ---- solver history across requests (tsgo_config.warm_requests) --------------------------------------------------------------
    // The deltas of the last Gauss-Newton iterations outlive tsgo_set_graph: in place when the structure is the same, carried over by
    // vertex id when it is not (the slabs are reused by the new tables, so the vectors are parked in an allocation of their own).
    struct Carry { int n = 0, n_tested = 0, P = 0; std::vector<uint32_t> pose_id; double err[kMaxWarm] = {}; } carry;
    T* carry_dev = nullptr; size_t carry_cap = 0;
    bool carried = false;        // the history came from the previous request: the first warm start made from it is checked (do_solve)
    int n_carried = 0, n_carry_dropped = 0;
    int carry_out() {
        carry.n = 0;
        if (!cfg.warm_requests || !have_graph_data || !have_prev || n_prev <= 0 || pr.P <= 0) return 0;
        const size_t n = (size_t)pr.P * 3;
        if ((size_t)n_prev * n > carry_cap) {
            if (carry_dev) { (void)hipFree(carry_dev); carry_dev = nullptr; carry_cap = 0; }
            const size_t want = (size_t)kMaxWarm * (n + n / 4);      // a growing graph comes back a little larger every time
            HIP_OK(hipMalloc((void**)&carry_dev, want * sizeof(T)));
            carry_cap = want;
        }
        for (int j = 0; j < n_prev; ++j) HIP_OK(hipMemcpyAsync(carry_dev + (size_t)j * n, hist[j], n * sizeof(T), hipMemcpyDeviceToDevice, stream));
        std::vector<T> e((size_t)kMaxWarm * nbC);
        if (int rc = copy_sync(e.data(), warm_err, e.size() * sizeof(T), hipMemcpyDeviceToHost)) return rc;
        for (int m = 0; m < kMaxWarm; ++m) { double s = 0; for (int k = 0; k < nbC; ++k) s += (double)e[(size_t)m * nbC + k]; carry.err[m] = s; }
        carry.pose_id.resize((size_t)pr.P);
        for (int i = 0; i < pr.P; ++i) carry.pose_id[(size_t)i] = structure.v_id[(size_t)pr.pose_vertex[(size_t)i]];
        HIP_OK(hipStreamSynchronize(stream));
        carry.P = pr.P; carry.n = n_prev; carry.n_tested = n_tested;
        return 0;
    }
    // after the new tables exist (hist, warm_err allocated; solver state reset): the parked deltas into the new pose numbering
    int carry_in(const tsgo_graph& g) {
        if (carry.n <= 0) return 0;
        const int n_old = carry.n; carry.n = 0;
        uint32_t max_id = 0;
        for (uint32_t id : carry.pose_id) max_id = std::max(max_id, id);
        const bool flat = (uint64_t)max_id < 8ull * (uint64_t)carry.P + 1024;
        std::vector<int> table(flat ? (size_t)max_id + 1 : 0, -1);
        std::unordered_map<uint32_t, int> by_id;
        if (flat) for (int i = 0; i < carry.P; ++i) table[carry.pose_id[(size_t)i]] = i;
        else { by_id.reserve((size_t)carry.P * 2); for (int i = 0; i < carry.P; ++i) by_id.emplace(carry.pose_id[(size_t)i], i); }
        std::vector<int> src((size_t)pr.P);
        int found = 0;
        for (int i = 0; i < pr.P; ++i) {
            const uint32_t id = g.v_id[(size_t)pr.pose_vertex[(size_t)i]];
            int j = -1;
            if (flat) { if (id <= max_id) j = table[id]; } else { auto it = by_id.find(id); if (it != by_id.end()) j = it->second; }
            src[(size_t)i] = j; found += j >= 0;
        }
        if (2 * (int64_t)found < (int64_t)std::max(pr.P, carry.P)) return 0;      // another graph altogether (or one grown beyond recognition): nothing to continue
        int* src_dev = nullptr;
        if (int rc = upload_i32(&src_dev, src)) return rc;
        const size_t n = (size_t)carry.P * 3;
        for (int j = 0; j < n_old; ++j)
            hipLaunchKernelGGL((k_gather_hist<T>), dim3(nbC), dim3(kBlock), 0, stream, pr.P, (const int*)src_dev, (const T*)(carry_dev + (size_t)j * n), hist[j]);
        std::vector<T> e((size_t)kMaxWarm * nbC, T(0));
        for (int m = 0; m < kMaxWarm; ++m) e[(size_t)m * nbC] = (T)carry.err[m];
        if (int rc = copy_sync(warm_err, e.data(), e.size() * sizeof(T), hipMemcpyHostToDevice)) return rc;
        HIP_OK(hipStreamSynchronize(stream));
        have_prev = true; n_prev = n_old; n_tested = carry.n_tested; carried = true; ++n_carried;      // (a grown graph IS young where it grew: n_lins stays 0)
        return 0;
    }

    // tsgo_reset_history: the next tsgo_set_graph starts the solver from nothing, whatever warm_requests says (a pooled handle
    // changing hands: one client's deltas must not seed another client's solves)
    void reset_history() override { have_prev = false; n_prev = 0; n_tested = 0; carried = false; carry.n = 0; }

    int refill(const tsgo_graph& g) {
        const auto t0 = std::chrono::steady_clock::now();
        const int rc = refill_values(g);
        // any failure leaves tables, lever arms and solver state half-updated: the handle then holds no graph and the next
        // tsgo_set_graph rebuilds from scratch
        if (rc) { have_graph_data = false; return rc; }
        ++structure_reuses;
        ms_setup = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (say_env) std::fprintf(stderr, "[tsgo] set_graph: same structure as the previous graph: values refilled in %.1f ms\n", ms_setup);
        return 0;
    }
    int refill_values(const tsgo_graph& g) {
        // a refilled handle must do, bit for bit, what a fresh one does: the cycle starts from the configured storage again (a
        // structure that left the packed format leaves it again, at the same solve)
        if (cy16 != (cfg.cycle_storage != 32)) {
            cy16 = cfg.cycle_storage != 32;
            if (cg_graph) { (void)hipGraphExecDestroy(cg_graph); cg_graph = nullptr; }
        }
        if (int rc = stage_values(g, true)) return rc;
        if (amg_on) {           // the rigid-mode lever arms follow the new estimates (host/amg.h: refresh_amg_geometry)
            refresh_amg_geometry(pr.pose_xyt, amg);
            HIP_OK(hipStreamSynchronize(stream));
            for (size_t l = 0; l < amg.levels.size(); ++l) {
                const std::vector<double>& rel = amg.levels[l].rel;
                if (int rc = stage_reserve(rel.size())) return rc;
                for (size_t k = 0; k < rel.size(); ++k) stage[k] = (T)rel[k];
                if (!rel.empty()) { if (int rc_ = copy_sync(lv[l].rel, stage, rel.size() * sizeof(T), hipMemcpyHostToDevice)) return rc_; }
            }
        }
        const bool keep = cfg.warm_requests && have_prev && n_prev > 0;      // same structure, same numbering: the history stays where it is
        const int keep_prev = n_prev, keep_tested = n_tested;
        if (int rc = reset_solver_state()) return rc;
        if (keep) { have_prev = true; n_prev = keep_prev; n_tested = keep_tested; carried = true; ++n_carried; n_lins = kYoungLins; }      // a continued graph is not a young one
        HIP_OK(hipStreamSynchronize(stream));
        return 0;
    }

    int set_graph(const tsgo_graph& g) override {
        const auto t0 = std::chrono::steady_clock::now();
        HIP_OK(hipSetDevice(cfg.device));
        if (g.n_vertices < 0 || g.n_edges < 0 || g.n_fixed < 0) return set_error(-2, "tsgo_set_graph: negative count");
        last_set_reused = have_graph_data && cfg.reuse_structure && structure.same_as(g);
        if (last_set_reused) return refill(g);
        if (int rc = carry_out()) return rc;
        release();
        BuildOptions bo; bo.rank = cfg.rank; bo.world = cfg.world; bo.lanes_per_pose = cfg.lanes_per_pose; bo.lanes_per_lm = cfg.lanes_per_lm;
        bo.fill_planes = false;
        const std::string err = build_problem(g, bo, pr);
        od_live = -1;
        if (!err.empty()) return set_error(-2, "tsgo_set_graph: " + err);
        pr.odom_analytic = cfg.odom_jacobian == 1;
        const bool say = say_env;
        auto lap = [&, last = t0](const char* what) mutable {
            const auto n = std::chrono::steady_clock::now();
            if (say) std::fprintf(stderr, "[tsgo] set_graph: %-34s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(n - last).count());
            last = n;
        };
        lap("layout (build_problem)");
        const int P = pr.P, L = pr.L;
        if (P == 0) return set_error(-2, "tsgo_set_graph: the graph has no Se2 vertex");
        amg_on = cfg.preconditioner == 1 && pr.P > kCoarsestMax;
        lv.assign(32, DevLevel<T>());            // host/amg.cpp never builds more levels than this; trimmed in upload_amg (the builder thread fills entries)
        ms_device_products = 0; schur_dev = false;
        if (amg_on) start_amg_builder(g);
        struct Joiner { std::thread& t; ~Joiner() { if (t.joinable()) t.join(); } } joiner{amg_builder};   // on every error path too
        if (int rc = dalloc(&ps, (size_t)P * 4)) return rc;
        if (int rc = dalloc(&theta, (size_t)P)) return rc;
        if (int rc = dalloc(&lmrec, (size_t)std::max(L, 1) * kLmRec)) return rc;
        if (int rc = upload_T(&gauge_p, pr.gauge_p.data(), pr.gauge_p.size())) return rc;
        if (int rc = upload_T(&gauge_l, pr.gauge_l.data(), pr.gauge_l.size())) return rc;
        if (int rc = alloc_table(tp, &st_p, pr.by_pose, 4, 4, true)) return rc;
        if (int rc = alloc_table(tl, &st_l, pr.by_lm, 4, 4, true)) return rc;
        if (int rc = alloc_table(to, &st_o, pr.odom, 9, oj() ? (int)PP_PLANES : 3, false)) return rc;
        if (int rc = stage_values(g, false)) return rc;
        // table-kernel grids are multiples of 8 (one eighth of the slices per XCD, see xcd_block())
        nbP = 8 * (((tp.n_slices + kWavesPerBlock - 1) / kWavesPerBlock + 7) / 8);
        nbL = 8 * (((tl.n_slices + kWavesPerBlock - 1) / kWavesPerBlock + 7) / 8);
        nbC = (P + kBlock - 1) / kBlock;
        if (int rc = dalloc(&part, (size_t)P * 18 + nbP)) return rc;
        if (int rc = dalloc(&dp, (size_t)P * 6)) return rc;
        if (int rc = dalloc(&minv, (size_t)P * 6)) return rc;
        if (int rc = dalloc(&r, (size_t)P * 3)) return rc;
        if (int rc = dalloc(&p, (size_t)P * 3)) return rc;
        if (int rc = dalloc(&q, (size_t)P * 3)) return rc;
        if (int rc = dalloc(&x, (size_t)P * 3)) return rc;
        if (int rc = dalloc(&zc, (size_t)P * kPoseRec)) return rc;
        { if (int rc_ = fill_zero(zc, (size_t)P * kPoseRec * sizeof(T))) return rc_; }
        if (int rc = dalloc(&zc32, (size_t)P * kPoseRec)) return rc;
        { if (int rc_ = fill_zero(zc32, (size_t)P * kPoseRec * sizeof(float))) return rc_; }
        if (int rc = dalloc(&tvec32, (size_t)std::max(L, 1) * 2)) return rc;
        if (int rc = dalloc(&sbuf, (size_t)P * 3 + nbP)) return rc;
        if (int rc = dalloc(&tvec, (size_t)std::max(L, 1) * 2)) return rc;
        if (int rc = dalloc(&ninv, (size_t)std::max(L, 1) * kNinvRec)) return rc;
        { if (int rc_ = fill_zero(ninv, (size_t)std::max(L, 1) * kNinvRec * sizeof(T))) return rc_; }
        if (int rc = dalloc(&dl, (size_t)std::max(L, 1) * 2)) return rc;
        { if (int rc_ = fill_zero(dl, (size_t)std::max(L, 1) * 2 * sizeof(T))) return rc_; }
        for (int k = 0; k < 2; ++k) { if (int rc = dalloc(&gpart[k], nbC)) return rc; if (int rc = dalloc(&st[k], 1)) return rc; }
        if (int rc = dalloc(&npart, (size_t)nbC + std::max(nbL, 1))) return rc;
        if (int rc = dalloc(&one_dev, 1)) return rc;
        if (int rc = dalloc(&gscale_dev, 1)) return rc;
        for (int j = 0; j < kMaxWarm; ++j) if (int rc = dalloc(&hist[j], (size_t)P * 3)) return rc;
        if (int rc = dalloc(&warm_err, (size_t)kMaxWarm * nbC)) return rc;
        if (int rc = dalloc(&warm_order_dev, 1)) return rc;
        HIP_OK(hipHostMalloc((void**)&h_state, sizeof(CgState<T>)));
        HIP_OK(hipHostMalloc((void**)&h_flag, 16 * sizeof(int), hipHostMallocCoherent));
        std::memset(h_flag, 0, 16 * sizeof(int));
        HIP_OK(hipHostMalloc((void**)&h_scratch, sizeof(T) * (size_t)(std::max(nbP, 2 * nbC) + nbL + 8)));
        HIP_OK(hipStreamSynchronize(stream));
        lap("state + slot tables to the device");
        if (amg_on) { if (int rc = upload_amg()) return rc; }
        if (int rc = reset_solver_state()) return rc;
        if (int rc = carry_in(g)) return rc;
        HIP_OK(hipStreamSynchronize(stream));
        if (say) std::fprintf(stderr, "[tsgo] set_graph:   of which multigrid patterns on the host %8.1f ms (its pair-list products on the device: %.1f ms of that)\n", ms_amg_symbolic, ms_device_products);
        if (say) for (size_t l = 0; l < lv.size(); ++l)
            std::fprintf(stderr, "[tsgo] level %zu: %d rows, %d blocks; pairs per block: A*P %.1f (%d blocks), P^T(AP) %.1f (%d upper blocks)\n", l, lv[l].n, lv[l].nnzA,
                         lv[l].pairs_T, lv[l].nnzT, lv[l].pairs_A, lv[l].n_upper);
        lap("multigrid patterns + upload");
        have_graph_data = true;
        // The captured PCG iterations (hipGraph, 15 ms to capture and instantiate at 100k poses) are NOT made here: the first
        // tsgo_optimize on a new structure launches eagerly — measured as fast (4.28 against 4.33 ms per step, profiles/r03j_*) as
        // long as the host thread keeps up — and the capture happens at the second tsgo_optimize on the same tables (a bench's
        // second step, a connection's second request with this structure: Engine::optimize).  A front-end that grows its graph
        // sends a new structure every time and never pays it.
        optimize_calls_on_tables = 0;
        cy16 = cfg.cycle_storage != 32;
        if (say) std::fprintf(stderr, "[tsgo] set_graph: %d slabs (%.0f MB) hold the graph; hipMalloc calls of this handle so far: %d, %.1f ms\n", (int)slabs.size(), slab_total / 1048576.0, n_malloc, ms_in_malloc);
        structure.take(g);
        ms_setup = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        return 0;
    }

    // ---- in-situ profiler (tsgo_profile_iteration): an event before every launch of an eagerly launched iteration ----
    struct ProfMark { hipEvent_t e; char name[64]; char where[32]; double bytes; };
    std::vector<ProfMark> prof; size_t prof_n = 0; bool prof_on = false;
    static const char* tname() { return sizeof(T) == 8 ? "double" : "float"; }
    // name: the kernel symbol as rocprofv3 prints it, without arguments (printf-style), e.g. "k_schur_lm<double, 4, 0, 1>"
    // PF(...): the arguments (byte models, labels) are evaluated only while a profile is being taken
#define PF(...) do { if (prof_on) pf(__VA_ARGS__); } while (0)
    void pf(double bytes, const char* where, const char* fmt, ...) __attribute__((format(printf, 4, 5))) {
        if (!prof_on) return;
        if (prof_n == prof.size()) { ProfMark m{}; if (hipEventCreate(&m.e) != hipSuccess) { prof_on = false; return; } prof.push_back(m); }
        ProfMark& m = prof[prof_n++];
        va_list ap; va_start(ap, fmt); std::vsnprintf(m.name, sizeof(m.name), fmt, ap); va_end(ap);
        std::snprintf(m.where, sizeof(m.where), "%s", where); m.bytes = bytes;
        (void)hipEventRecord(m.e, stream);
    }
    std::string lvl(const char* role, size_t l) const { return std::string(role) + " L" + std::to_string(l); }
    // algorithmic bytes of the table kernels and of the block-row kernels of the cycle
    double od_live = -1;      // live ODOM slots of this shard (the byte models'), counted once per structure
    double od_slots_live() { if (od_live < 0) { od_live = 0; for (uint32_t e : pr.odom.edge) od_live += e != kNoEdge; } return od_live; }
    double bytes_schur_lm(bool low) const { const double s = low ? 4 : sizeof(T), v = sizeof(T); return (double)pr.n_lm_edges * (4 + 4 * s) + pr.P * 5.0 * v + pr.L * 5.0 * v; }
    double bytes_schur_pose(bool low) { const double s = low ? 4 : sizeof(T), v = sizeof(T); return (double)pr.n_lm_edges * (4 + 4 * s) + pr.L * 2.0 * v + pr.P * 14.0 * v + od_slots_live() * (4 + 6 * v); }
    double bytes_sweep(const DevLevel<T>& L) const { return (double)L.nnzA * (4.0 * cyw() + 4) + (double)L.n * (3 * 3 * sizeof(T) + 9 * sizeof(H) + 4); }
    double bytes_transfer(const DevLevel<T>& L, int vecs_fine) const { return (double)L.nnzP * (4.0 * cyw() + 4) + (double)L.n * 3 * sizeof(T) * vecs_fine + (double)L.n_agg * (3 * sizeof(T) + 4); }

    // ---- launches --------------------------------------------------------------------------------
    // damping of the current linearisation (rules = 1, graph_optimizer.py:24-43; 0 under the cpu/eigen rules) and the step the update takes
    double lambda = 0;
    bool py_rules() const { return cfg.rules == 1; }
    // pose-pose slots in general form (tsgo_math.h: eight dynamic planes per slot): analytic ODOM Jacobians, or a graph that holds
    // virtual landmark measurements (edge type 2) — the kernels' OJ = 1 instantiations
    bool oj() const { return cfg.odom_jacobian == 1 || pr.has_vlm; }
    int odom_analytic_flag() const { return cfg.odom_jacobian == 1 ? 1 : 0; }
    double step_scale() const { return py_rules() ? cfg.lr : kStepScale; }
    void launch_lin() {
        const int zf = py_rules() ? 1 : 0;
        if (tl.n_slices > 0) LAUNCH_G(pr.by_lm.G, k_lin_lm, nbL, stream, tl, ps, lmrec, gauge_l, ninv, (T)lambda, zf);
        if (oj()) LAUNCH_GM(pr.by_pose.G, k_lin_pose, 1, nbP, stream, tp, to, ps, lmrec, gauge_p, pr.pose_first, pr.pose_last, part, part + (size_t)pr.P * 18, (T)lambda, zf, odom_analytic_flag());
        else LAUNCH_G(pr.by_pose.G, k_lin_pose, nbP, stream, tp, to, ps, lmrec, gauge_p, pr.pose_first, pr.pose_last, part, part + (size_t)pr.P * 18, (T)lambda, zf);
    }
    void launch_lin_pose_only() {       // tsgo_time_kernel
        const int zf = py_rules() ? 1 : 0;
        if (oj()) LAUNCH_GM(pr.by_pose.G, k_lin_pose, 1, nbP, stream, tp, to, ps, lmrec, gauge_p, pr.pose_first, pr.pose_last, part, part + (size_t)pr.P * 18, (T)lambda, zf, odom_analytic_flag());
        else LAUNCH_G(pr.by_pose.G, k_lin_pose, nbP, stream, tp, to, ps, lmrec, gauge_p, pr.pose_first, pr.pose_last, part, part + (size_t)pr.P * 18, (T)lambda, zf);
    }
    void launch_finalize() {
        hipLaunchKernelGGL((k_pose_finalize<T>), dim3(nbC), dim3(kBlock), 0, stream, pr.P, part, ps, dp, minv, r, p, q, x, zc, gpart[0], st[0], (const T*)(amg_on ? omega_dev : one_dev), gscale_dev, amg_on && low_cycle ? zc32 : (float*)nullptr);
    }
    // S * (vector in zc) -> sbuf, dot partials behind it.  low: read the f32 copy of the slot planes (the two
    // products inside the multigrid cycle; never the product PCG itself takes).
    // Edge-sharded runs: every rank's passes cover its own landmarks (and the ODOM rows / diagonal blocks of its own
    // poses), so what lands in sbuf is a PARTIAL product and partial dots: one all-reduce of [3P | nbP] makes both whole
    // on every rank.  (r, z) partials are computed redundantly from replicated vectors and need no reduction.
    int launch_matvec(int slot, bool with_rz = false, bool low = false) {
        const char* wh = low ? "in-cycle product" : (with_rz ? "PCG product" : "product");
        if (low) {
            PF(bytes_schur_lm(true), wh, "k_schur_lm<%s, %d, 0, 1>", tname(), pr.by_lm.G);
            if (tl.n_slices > 0) LAUNCH_GML(pr.by_lm.G, k_schur_lm, 0, 1, nbL, stream, tl, zc, lmrec, (const T*)ninv, tvec, st[slot], T(0), dl, npart, (const float*)zc32, tvec32);
            PF(bytes_schur_pose(true), wh, "k_schur_pose<%s, %d, 1, %d>", tname(), pr.by_pose.G, oj() ? 1 : 0);
            if (oj()) LAUNCH_GML(pr.by_pose.G, k_schur_pose, 1, 1, nbP, stream, tp, to, zc, tvec, dp, pr.pose_first, pr.pose_last, sbuf, sbuf + (size_t)pr.P * 3, st[slot],
                                 (const T*)nullptr, rzpart, (const float*)zc32, (const float*)tvec32);
            else LAUNCH_GML1(pr.by_pose.G, k_schur_pose, 1, nbP, stream, tp, to, zc, tvec, dp, pr.pose_first, pr.pose_last, sbuf, sbuf + (size_t)pr.P * 3, st[slot],
                             (const T*)nullptr, rzpart, (const float*)zc32, (const float*)tvec32);
        } else {
            PF(bytes_schur_lm(false), wh, "k_schur_lm<%s, %d, 0, 0>", tname(), pr.by_lm.G);
            if (tl.n_slices > 0) LAUNCH_GM(pr.by_lm.G, k_schur_lm, 0, nbL, stream, tl, zc, lmrec, (const T*)ninv, tvec, st[slot], T(0), dl, npart);
            PF(bytes_schur_pose(false), wh, "k_schur_pose<%s, %d, 0, %d>", tname(), pr.by_pose.G, oj() ? 1 : 0);
            if (oj()) LAUNCH_GML(pr.by_pose.G, k_schur_pose, 0, 1, nbP, stream, tp, to, zc, tvec, dp, pr.pose_first, pr.pose_last, sbuf, sbuf + (size_t)pr.P * 3, st[slot],
                                 (const T*)(with_rz ? r : nullptr), rzpart);
            else LAUNCH_G(pr.by_pose.G, k_schur_pose, nbP, stream, tp, to, zc, tvec, dp, pr.pose_first, pr.pose_last, sbuf, sbuf + (size_t)pr.P * 3, st[slot],
                          (const T*)(with_rz ? r : nullptr), rzpart);
        }
        return allreduce(sbuf, (size_t)pr.P * 3 + nbP);
    }
    static int grid_for(int n, int per_thread_lanes = 1) { return std::max(1, (int)(((size_t)n * per_thread_lanes + kBlock - 1) / kBlock)); }

    // numeric multigrid setup for the current linearisation (after lin + finalize)
    int launch_amg_setup() {
        DevLevel<T>& L0 = lv[0];
        // sharded: off-diagonal blocks are partial sums over this rank's landmarks and ODOM rows; the diagonal (from the
        // all-reduced linearisation partials, identical everywhere) is contributed by rank 0 alone; one all-reduce
        // makes level 0 whole and identical on every rank, everything below it is then computed redundantly
        const int n_sum = n_upper0 >= 0 ? n_upper0 : L0.nnzA;
        hipLaunchKernelGGL((k_schur_blocks<T>), dim3(grid_for(n_sum)), dim3(kBlock), 0, stream, n_sum, L0.A_row, L0.A_col, sc_ptr, sc_si, sc_sk,
                           sc_optr, sc_os, tp, (const T*)to.dyn, to.slots, (const T*)lmrec, (const T*)ps, (const T*)part, L0.A, pr.rank == 0 ? 1 : 0,
                           to.idx, oj() ? 1 : 0, (const int*)(n_upper0 >= 0 ? upper0 : nullptr), (const int*)(n_upper0 >= 0 ? lower0 : nullptr));
        if (int rc = allreduce_h(L0.A, (size_t)L0.nnzA * 9)) return rc;
        if (explicit0) do { if (cy16) hipLaunchKernelGGL((k_to_planes<T, 1>), dim3(grid_for(L0.n, 8)), dim3(kBlock), 0, stream, L0.n, (const int*)L0.A_ptr, (const H*)L0.A, L0.Apm); else hipLaunchKernelGGL((k_to_planes<T, 0>), dim3(grid_for(L0.n, 8)), dim3(kBlock), 0, stream, L0.n, (const int*)L0.A_ptr, (const H*)L0.A, L0.Apm); } while (0);
        for (size_t l = 0; l < lv.size(); ++l) {
            DevLevel<T>& L = lv[l];
            H* Anext = l + 1 < lv.size() ? lv[l + 1].A : A_last;
            hipLaunchKernelGGL((k_block_inv<T>), dim3(grid_for(L.n)), dim3(kBlock), 0, stream, L.n, L.diag, (const H*)L.A, L.Dinv);
            hipLaunchKernelGGL((k_prolongator<T>), dim3(grid_for(L.nnzP)), dim3(kBlock), 0, stream, L.nnzP, L.P_row, L.p_self, L.ps_ptr, L.ps_x, L.ps_y,
                               (const H*)L.A, (const H*)L.Dinv, (const T*)L.rel, (T)kProlongOmega, L.P, L.p_to_r, L.Rv);
            if (L.pairs_T > kVeryLongPairList) hipLaunchKernelGGL((k_pair_gemm_wave<T, 0, 64>), dim3(grid_for(L.nnzT, 64)), dim3(kBlock), 0, stream, L.nnzT, L.ts_ptr, L.ts_x, L.ts_y, (const H*)L.A, (const H*)L.P, L.Tv, (const int*)nullptr);
            else if (L.pairs_T > kLongPairList) hipLaunchKernelGGL((k_pair_gemm_wave<T, 0, 16>), dim3(grid_for(L.nnzT, 16)), dim3(kBlock), 0, stream, L.nnzT, L.ts_ptr, L.ts_x, L.ts_y, (const H*)L.A, (const H*)L.P, L.Tv, (const int*)nullptr);
            else if (L.pairs_T > kMediumPairList) hipLaunchKernelGGL((k_pair_gemm_wave<T, 0, 8>), dim3(grid_for(L.nnzT, 8)), dim3(kBlock), 0, stream, L.nnzT, L.ts_ptr, L.ts_x, L.ts_y, (const H*)L.A, (const H*)L.P, L.Tv, (const int*)nullptr);
            else hipLaunchKernelGGL((k_pair_gemm<T, 0>), dim3(grid_for((L.nnzT + kPairBlocksPerWave - 1) / kPairBlocksPerWave, 64)), dim3(kBlock), 0, stream, L.nnzT, L.ts_ptr, L.ts_x, L.ts_y, (const H*)L.A, (const H*)L.P, L.Tv, (const int*)nullptr);
            if (L.pairs_A > kVeryLongPairList) hipLaunchKernelGGL((k_pair_gemm_wave<T, 1, 64>), dim3(grid_for(L.n_upper, 64)), dim3(kBlock), 0, stream, L.n_upper, L.as_ptr, L.as_x, L.as_y, (const H*)L.P, (const H*)L.Tv, Anext, (const int*)L.as_upper);
            else if (L.pairs_A > kLongPairList) hipLaunchKernelGGL((k_pair_gemm_wave<T, 1, 16>), dim3(grid_for(L.n_upper, 16)), dim3(kBlock), 0, stream, L.n_upper, L.as_ptr, L.as_x, L.as_y, (const H*)L.P, (const H*)L.Tv, Anext, (const int*)L.as_upper);
            else if (L.pairs_A > kMediumPairList) hipLaunchKernelGGL((k_pair_gemm_wave<T, 1, 8>), dim3(grid_for(L.n_upper, 8)), dim3(kBlock), 0, stream, L.n_upper, L.as_ptr, L.as_x, L.as_y, (const H*)L.P, (const H*)L.Tv, Anext, (const int*)L.as_upper);
            else hipLaunchKernelGGL((k_pair_gemm<T, 1>), dim3(grid_for((L.n_upper + kPairBlocksPerWave - 1) / kPairBlocksPerWave, 64)), dim3(kBlock), 0, stream, L.n_upper, L.as_ptr, L.as_x, L.as_y, (const H*)L.P, (const H*)L.Tv, Anext, (const int*)L.as_upper);
            hipLaunchKernelGGL((k_mirror_blocks<T>), dim3(grid_for(L.nnzNext, 9)), dim3(kBlock), 0, stream, L.nnzNext, (const int*)L.as_mirror, Anext);
            // cycle format of what the V-cycle reads on this level (level 0's matrix is only read by the setup)
            do { if (cy16) hipLaunchKernelGGL((k_to_planes<T, 1>), dim3(grid_for(L.n, 8)), dim3(kBlock), 0, stream, L.n, (const int*)L.P_ptr, (const H*)L.P, L.Ppm); else hipLaunchKernelGGL((k_to_planes<T, 0>), dim3(grid_for(L.n, 8)), dim3(kBlock), 0, stream, L.n, (const int*)L.P_ptr, (const H*)L.P, L.Ppm); } while (0);
            do { if (cy16) hipLaunchKernelGGL((k_to_planes<T, 1>), dim3(grid_for(L.n_agg, 8)), dim3(kBlock), 0, stream, L.n_agg, (const int*)L.R_ptr, (const H*)L.Rv, L.Rpm); else hipLaunchKernelGGL((k_to_planes<T, 0>), dim3(grid_for(L.n_agg, 8)), dim3(kBlock), 0, stream, L.n_agg, (const int*)L.R_ptr, (const H*)L.Rv, L.Rpm); } while (0);
            if (l + 1 < lv.size()) do { if (cy16) hipLaunchKernelGGL((k_to_planes<T, 1>), dim3(grid_for(lv[l + 1].n, 8)), dim3(kBlock), 0, stream, lv[l + 1].n, (const int*)lv[l + 1].A_ptr, (const H*)lv[l + 1].A, lv[l + 1].Apm); else hipLaunchKernelGGL((k_to_planes<T, 0>), dim3(grid_for(lv[l + 1].n, 8)), dim3(kBlock), 0, stream, lv[l + 1].n, (const int*)lv[l + 1].A_ptr, (const H*)lv[l + 1].A, lv[l + 1].Apm); } while (0);
        }
        hipLaunchKernelGGL((k_dense_inverse<T>), dim3(1), dim3(kDenseThreads), 0, stream, nb_last, last_ptr, last_col, (const H*)A_last, inv_last);
        return launch_bottom_setup();
    }

    static int lanes_for(double avg_row) {
        return avg_row <= 4 ? 4 : (avg_row <= 12 ? 8 : (avg_row <= 40 ? 32 : 64));
    }
    // Block-row sweeps (k_bcsr_residual): on the big levels (thousands of rows: every wave slot of the device is taken more
    // than once) a lane should carry two to four blocks, not one — 16 lanes per row at 28 and at 67 blocks per row measured
    // 8.7 / 5.6 us against 9.3 / 6.2 us with 32 / 64 lanes; the small levels are one wave round either way and want the
    // shortest chain, i.e. many lanes (profiles/r02h_lanes_per_row.txt).
    static int lanes_for_sweep(double avg_row, int n_rows) {
        if (n_rows >= 4096 && avg_row > 12) return 16;
        return lanes_for(avg_row);
    }
#define LAUNCH_LPR_(LPR, INST, n_rows, ...)                                                                              \
    do {                                                                                                                 \
        switch (LPR) {                                                                                                   \
            case 4: hipLaunchKernelGGL((INST(4)), dim3(grid_for(n_rows, 4)), dim3(kBlock), 0, stream, __VA_ARGS__); break;   \
            case 8: hipLaunchKernelGGL((INST(8)), dim3(grid_for(n_rows, 8)), dim3(kBlock), 0, stream, __VA_ARGS__); break;   \
            case 16: hipLaunchKernelGGL((INST(16)), dim3(grid_for(n_rows, 16)), dim3(kBlock), 0, stream, __VA_ARGS__); break; \
            case 32: hipLaunchKernelGGL((INST(32)), dim3(grid_for(n_rows, 32)), dim3(kBlock), 0, stream, __VA_ARGS__); break; \
            default: hipLaunchKernelGGL((INST(64)), dim3(grid_for(n_rows, 64)), dim3(kBlock), 0, stream, __VA_ARGS__); break; \
        }                                                                                                                \
    } while (0)
    // a block-row sweep over the cycle-format copy of a level's matrix (MODE 0 residual, 1 smoothing sweep), f32 or packed half
#define SWEEP_INST16(L_) k_bcsr_residual<T, L_, SWEEP_MODE, 1, 1>
#define SWEEP_INST32(L_) k_bcsr_residual<T, L_, SWEEP_MODE, 1, 0>
#define RESTRICT_INST16(L_) k_restrict<T, L_, SWEEP_MODE, 1>
#define RESTRICT_INST32(L_) k_restrict<T, L_, SWEEP_MODE, 0>
#define PROLONG_INST16(L_) k_prolong_add<T, L_, 1>
#define PROLONG_INST32(L_) k_prolong_add<T, L_, 0>
#define APPLY_INST16(L_) k_bcsr_apply<T, L_, 1>
#define APPLY_INST32(L_) k_bcsr_apply<T, L_, 0>
    template <int SWEEP_MODE> void launch_sweep(int lpr, DevLevel<T>& L, const T* rhs, const T* cur, T* out, const T* omega, const CgState<T>* s) {
        if (cy16) LAUNCH_LPR_(lpr, SWEEP_INST16, L.n, L.n, L.A_ptr, L.A_col, (const void*)L.Apm, rhs, cur, (const H*)L.Dinv, out, omega, s);
        else LAUNCH_LPR_(lpr, SWEEP_INST32, L.n, L.n, L.A_ptr, L.A_col, (const void*)L.Apm, rhs, cur, (const H*)L.Dinv, out, omega, s);
    }
    template <int SWEEP_MODE> void launch_restrict(int lpr, DevLevel<T>& L, const T* va, const T* vb, T* rc, const H* dinv_next, T* z_next, const T* omega, const CgState<T>* s) {
        if (cy16) LAUNCH_LPR_(lpr, RESTRICT_INST16, L.n_agg, L.n_agg, L.R_ptr, L.R_col, (const uint32_t*)L.Rpm, va, vb, rc, dinv_next, z_next, omega, s);
        else LAUNCH_LPR_(lpr, RESTRICT_INST32, L.n_agg, L.n_agg, L.R_ptr, L.R_col, (const uint32_t*)L.Rpm, va, vb, rc, dinv_next, z_next, omega, s);
    }
    void launch_prolong(DevLevel<T>& L, const T* e, T* z, int zs, const CgState<T>* s, size_t level) {
        PF(bytes_transfer(L, 2), lvl("prolong into", level).c_str(), "k_prolong_add<%s, %d, %d>", tname(), lanes_for((double)L.nnzP / std::max(1, L.n)), cy16 ? 1 : 0);
        const int lpr = lanes_for((double)L.nnzP / std::max(1, L.n));
        float* z32 = (level == 0 && low_cycle && !explicit0) ? zc32 : (float*)nullptr;      // level 0 prolongs into the pose records: keep their f32 copy current
        if (cy16) LAUNCH_LPR_(lpr, PROLONG_INST16, L.n, L.n, L.P_ptr, L.P_col, (const uint32_t*)L.Ppm, e, z, zs, s, z32);
        else LAUNCH_LPR_(lpr, PROLONG_INST32, L.n, L.n, L.P_ptr, L.P_col, (const uint32_t*)L.Ppm, e, z, zs, s, z32);
    }

    // Damping of the block-Jacobi smoother per level from a power iteration on D^-1 A (12 steps): the V-cycle
    // is a symmetric positive definite preconditioner only while omega * rho(D^-1 A) < 2, and smoothed Galerkin
    // matrices reach rho = 2.1 ... 3.4 (measured on the CPU twin).  omega = min(1, 1.6 / (1.05 rho)).
    int estimate_damping() {
        const size_t nl = lv.size();
        for (size_t l = 0; l < nl; ++l) {
            DevLevel<T>& L = lv[l];
            T* a = l == 0 ? pw_a : L.res; T* b = l == 0 ? pw_b : L.z2;
            const int n3 = L.n * 3;
            hipLaunchKernelGGL((k_seed_vector<T>), dim3(grid_for(n3)), dim3(kBlock), 0, stream, n3, a);
            const int lprA = lanes_for((double)L.nnzA / std::max(1, L.n));
            for (int it = 0; it < kRhoSteps; ++it) {
                switch (lprA) {     // the block-indexed matrix (PM = 0): level 0 has no cycle-format copy
                    case 4: hipLaunchKernelGGL((k_bcsr_residual<T, 4, 2, 0>), dim3(grid_for(L.n, 4)), dim3(kBlock), 0, stream, L.n, L.A_ptr, L.A_col, (const void*)L.A, (const T*)a, (const T*)a, (const H*)L.Dinv, b, (const T*)omega_dev, (const CgState<T>*)st[0]); break;
                    case 16: hipLaunchKernelGGL((k_bcsr_residual<T, 16, 2, 0>), dim3(grid_for(L.n, 16)), dim3(kBlock), 0, stream, L.n, L.A_ptr, L.A_col, (const void*)L.A, (const T*)a, (const T*)a, (const H*)L.Dinv, b, (const T*)omega_dev, (const CgState<T>*)st[0]); break;
                    case 8: hipLaunchKernelGGL((k_bcsr_residual<T, 8, 2, 0>), dim3(grid_for(L.n, 8)), dim3(kBlock), 0, stream, L.n, L.A_ptr, L.A_col, (const void*)L.A, (const T*)a, (const T*)a, (const H*)L.Dinv, b, (const T*)omega_dev, (const CgState<T>*)st[0]); break;
                    case 32: hipLaunchKernelGGL((k_bcsr_residual<T, 32, 2, 0>), dim3(grid_for(L.n, 32)), dim3(kBlock), 0, stream, L.n, L.A_ptr, L.A_col, (const void*)L.A, (const T*)a, (const T*)a, (const H*)L.Dinv, b, (const T*)omega_dev, (const CgState<T>*)st[0]); break;
                    default: hipLaunchKernelGGL((k_bcsr_residual<T, 64, 2, 0>), dim3(grid_for(L.n, 64)), dim3(kBlock), 0, stream, L.n, L.A_ptr, L.A_col, (const void*)L.A, (const T*)a, (const T*)a, (const H*)L.Dinv, b, (const T*)omega_dev, (const CgState<T>*)st[0]); break;
                }
                std::swap(a, b);
            }
            // a = v_K, b = v_{K-1}
            hipLaunchKernelGGL((k_norm2<T>), dim3(kRhoBlocks), dim3(kBlock), 0, stream, n3, (const T*)a, rho_part + (2 * l) * kRhoBlocks);
            hipLaunchKernelGGL((k_norm2<T>), dim3(kRhoBlocks), dim3(kBlock), 0, stream, n3, (const T*)b, rho_part + (2 * l + 1) * kRhoBlocks);
        }
        HIP_OK(hipMemcpyAsync(h_rho, rho_part, sizeof(T) * 2 * nl * kRhoBlocks, hipMemcpyDeviceToHost, stream));
        HIP_OK(hipStreamSynchronize(stream));
        std::vector<T> om(16, (T)kSmootherOmega);
        for (size_t l = 0; l < nl; ++l) {
            double nk = 0, nk1 = 0;
            for (int k = 0; k < kRhoBlocks; ++k) { nk += (double)h_rho[(2 * l) * kRhoBlocks + k]; nk1 += (double)h_rho[(2 * l + 1) * kRhoBlocks + k]; }
            double w = kSmootherOmega;
            if (nk1 > 0 && nk > 0 && std::isfinite(nk) && std::isfinite(nk1)) {
                const double rho = 1.05 * std::sqrt(nk / nk1);
                w = std::min(1.0, 1.6 / rho);
            }
            om[l] = (T)w; omega_host[l] = w;
        }
        if (say_env) { std::fprintf(stderr, "[tsgo] smoother damping per level:"); for (size_t l = 0; l < nl; ++l) std::fprintf(stderr, " %.3f", omega_host[l]); std::fprintf(stderr, "\n"); }
        HIP_OK(hipMemcpyAsync(omega_dev, om.data(), 16 * sizeof(T), hipMemcpyHostToDevice, stream));
        HIP_OK(hipStreamSynchronize(stream));
        return 0;
    }

    // zc[.][0..2] = V(1,1)-cycle(r).  On entry zc already holds the level-0 pre-smoothing Minv r
    // (written by pose_finalize / k_cg_step).  Level l >= 1 keeps r, z (pre-smoothed by the restriction
    // above it), res and the post-smoothed result z2.
    // tsgo_config.cycle_level0 = 1: the two products inside the cycle read the EXPLICIT level-0 matrix of the hierarchy
    // (lagged with it, hub landmarks truncated, f32) instead of the implicit Schur passes — no all-reduce in a sharded run
    bool explicit0 = false;
    int launch_cycle_product(int slot) {
        if (!explicit0) return launch_matvec(slot, false, low_cycle);
        DevLevel<T>& L = lv[0];
        PF(bytes_sweep(L), "in-cycle product (explicit)", "k_bcsr_apply<%s, %d, %d>", tname(), lanes_for_sweep((double)L.nnzA / std::max(1, L.n), L.n), cy16 ? 1 : 0);
        const int lpr = lanes_for_sweep((double)L.nnzA / std::max(1, L.n), L.n);
        if (cy16) LAUNCH_LPR_(lpr, APPLY_INST16, L.n, L.n, L.A_ptr, L.A_col, (const uint32_t*)L.Apm, (const T*)zc, kPoseRec, sbuf, (const CgState<T>*)st[slot]);
        else LAUNCH_LPR_(lpr, APPLY_INST32, L.n, L.n, L.A_ptr, L.A_col, (const uint32_t*)L.Apm, (const T*)zc, kPoseRec, sbuf, (const CgState<T>*)st[slot]);
        return 0;
    }
    int launch_vcycle(int slot) {
        const CgState<T>* s = st[slot];
        const size_t nl = lv.size();              // explicit levels 0 .. nl-1, dense level below
        if (int rc = launch_cycle_product(slot)) return rc;
        {
            DevLevel<T>& L = lv[0];
            const int lpr = lanes_for((double)L.nnzP / std::max(1, L.n_agg));
            if (nl > 1) PF(bytes_transfer(L, 2), "restrict from L0", "k_restrict<%s, %d, 1, %d>", tname(), lpr, cy16 ? 1 : 0);
            if (nl > 1) launch_restrict<1>(lpr, L, (const T*)r, (const T*)sbuf, lv[1].r, (bottom_dense && nl == 2) ? (const H*)nullptr : (const H*)lv[1].Dinv, lv[1].z, (const T*)(omega_dev + 1), s);      // (nl == 3 with the factored level: lv[1] keeps its pre-sweep)
        }
        // coarse levels: V(nu,nu) with nu = coarse_sweeps block-Jacobi sweeps (the first pre-sweep comes fused
        // with the restriction above).  The current iterate alternates between L.z and L.z2; it ends in L.z2.
        const bool dense_bottom = bottom_dense && nl > 1;      // the last explicit level's whole cycle is one dense product (k_bottom_apply) ...
        const bool dense_tail2 = dense_bottom && tail2;        // ... and the level above it two launches (t = E^T r, k_tail_up)
        const size_t first_dense = dense_tail2 ? nl - 2 : (dense_bottom ? nl - 1 : nl);      // levels from here down run no sweeps of their own
        for (size_t l = 1; l < std::min(nl, first_dense); ++l) {
            DevLevel<T>& L = lv[l];
            const int nu = nu_at(l);
            const int lprA = lanes_for_sweep((double)L.nnzA / std::max(1, L.n), L.n);
            T* cur = L.z; T* oth = L.z2;
            for (int sw = 1; sw < nu; ++sw) {
                PF(bytes_sweep(L), lvl("pre-sweep", l).c_str(), "k_bcsr_residual<%s, %d, 1, 1, %d>", tname(), lprA, cy16 ? 1 : 0);
                launch_sweep<1>(lprA, L, (const T*)L.r, (const T*)cur, oth, (const T*)(omega_dev + l), s);
                std::swap(cur, oth);
            }
            PF(bytes_sweep(L), lvl("residual", l).c_str(), "k_bcsr_residual<%s, %d, 0, 1, %d>", tname(), lprA, cy16 ? 1 : 0);
            launch_sweep<0>(lprA, L, (const T*)L.r, (const T*)cur, L.res, (const T*)(omega_dev + l), s);
            if (l + 1 < nl) {
                const int lpr = lanes_for((double)L.nnzP / std::max(1, L.n_agg));
                const bool no_presmooth = dense_bottom && !dense_tail2 && l + 2 == nl;      // the dense bottom operator pre-smooths by itself (the factored level wants z1 = W r)
                PF(bytes_transfer(L, 1), lvl("restrict from", l).c_str(), "k_restrict<%s, %d, 0, %d>", tname(), lpr, cy16 ? 1 : 0);
                launch_restrict<0>(lpr, L, (const T*)L.res, (const T*)L.res, lv[l + 1].r, no_presmooth ? (const H*)nullptr : (const H*)lv[l + 1].Dinv, lv[l + 1].z, (const T*)(omega_dev + l + 1), s);
            }
        }
        // iterate of level l after the down pass: L.z when nu is odd, L.z2 when even
        auto down_iter = [&](DevLevel<T>& L, int nu) { return (nu % 2) ? L.z : L.z2; };
        auto down_other = [&](DevLevel<T>& L, int nu) { return (nu % 2) ? L.z2 : L.z; };
        if (dense_tail2) {       // levels nl-2 and nl-1 at once: z2 = 2 z1 - W A z1 + G (E^T r), z1 = W r left by the restriction into nl-2
            DevLevel<T>& L = lv[nl - 2];
            const int n3 = L.n * 3, nd = lv[nl - 1].n * 3;
            PF((double)n3 * nd * sizeof(float) + (double)(n3 + nd) * sizeof(T), lvl("t = E^T r of", nl - 2).c_str(), "k_rowdot_wg<%s>", tname());
            hipLaunchKernelGGL((k_rowdot_wg<T>), dim3(nd), dim3(kBlock), 0, stream, nd, n3, (const float*)tail_Etf, (const T*)L.r, tail_t, s);
            PF(bytes_sweep(L) + (double)n3 * nd * sizeof(float), lvl("cycles of", nl - 2).c_str(), "k_tail_up<%s, %d>", tname(), cy16 ? 1 : 0);
            if (cy16) hipLaunchKernelGGL((k_tail_up<T, 1>), dim3(L.n), dim3(kBlock), 0, stream, L.n, (const int*)L.A_ptr, (const int*)L.A_col, (const uint32_t*)L.Apm, (const H*)L.Dinv, (const T*)(omega_dev + nl - 2), (const T*)L.z, nd, (const float*)tail_Gf, (const T*)tail_t, L.z2, s);
            else hipLaunchKernelGGL((k_tail_up<T, 0>), dim3(L.n), dim3(kBlock), 0, stream, L.n, (const int*)L.A_ptr, (const int*)L.A_col, (const uint32_t*)L.Apm, (const H*)L.Dinv, (const T*)(omega_dev + nl - 2), (const T*)L.z, nd, (const float*)tail_Gf, (const T*)tail_t, L.z2, s);
        } else if (dense_bottom) {      // z2 = B r: pre-sweep, coarse correction through the dense inverse and post-sweep of the last explicit level at once
            DevLevel<T>& L = lv[nl - 1];
            const int n3 = L.n * 3;
            PF((double)n3 * n3 * sizeof(float) + 2.0 * n3 * sizeof(T), lvl("whole cycle of", nl - 1).c_str(), "k_bottom_apply<%s>", tname());
            hipLaunchKernelGGL((k_bottom_apply<T>), dim3(grid_for(n3, 64)), dim3(kBlock), 0, stream, n3, n3, (const float*)bot_Bf, (const T*)L.r, L.z2, s);
        } else if (nl > 1 && lv[nl - 1].n * 4 <= kDenseThreads) {   // bottom: restrict + dense inverse + prolong in one workgroup, on the last explicit level
            DevLevel<T>& L = lv[nl - 1];
            PF(2.0 * L.nnzP * (9 * sizeof(H) + 4) + (double)nb_last * 3 * nb_last * 3 * sizeof(T) + L.n * 6.0 * sizeof(T), lvl("restrict + dense solve + prolong", nl - 1).c_str(), "k_coarse_tail<%s>", tname());
            hipLaunchKernelGGL((k_coarse_tail<T>), dim3(1), dim3(kDenseThreads), 0, stream, L.n, L.n_agg, L.R_ptr, L.R_col, (const H*)L.Rv, L.P_ptr, L.P_col, (const H*)L.P,
                               (const T*)L.res, (const T*)inv_last, down_iter(L, nu_at(nl - 1)), s);
        } else if (nl > 1) {   // a last explicit level too long for the one-workgroup kernel (4 lanes per row): the same three steps as launches
            DevLevel<T>& L = lv[nl - 1];
            PF(bytes_transfer(L, 1), lvl("restrict from", nl - 1).c_str(), "k_restrict<%s, 8, 0, %d>", tname(), cy16 ? 1 : 0);
            launch_restrict<0>(8, L, (const T*)L.res, (const T*)L.res, r_last, (const H*)nullptr, (T*)nullptr, (const T*)one_dev, s);
            PF((double)nb_last * 3 * nb_last * 3 * sizeof(T), "dense solve", "k_dense_apply<%s>", tname());
            hipLaunchKernelGGL((k_dense_apply<T>), dim3(1), dim3(kBlock), 0, stream, nb_last * 3, (const T*)inv_last, (const T*)r_last, z_last, s);
            launch_prolong(L, z_last, down_iter(L, nu_at(nl - 1)), 3, s, nl - 1);
        } else {        // only level 0 above the dense level: residual r - S z is restricted from (r, sbuf)
            DevLevel<T>& L = lv[0];
            PF(bytes_transfer(L, 2), "restrict from L0", "k_restrict<%s, 8, 1, %d>", tname(), cy16 ? 1 : 0);
            launch_restrict<1>(8, L, (const T*)r, (const T*)sbuf, r_last, (const H*)nullptr, (T*)nullptr, (const T*)one_dev, s);
            PF((double)nb_last * 3 * nb_last * 3 * sizeof(T), "dense solve", "k_dense_apply<%s>", tname());
            hipLaunchKernelGGL((k_dense_apply<T>), dim3(1), dim3(kBlock), 0, stream, nb_last * 3, (const T*)inv_last, (const T*)r_last, z_last, s);
        }
        for (size_t l = nl - 1; l >= 1; --l) {
            DevLevel<T>& L = lv[l];
            if (l >= first_dense) continue;      // a dense level's result is in its z2 already
            const int nu = nu_at(l);
            T* cur = down_iter(L, nu); T* oth = down_other(L, nu);
            if (l + 1 < nl) launch_prolong(L, lv[l + 1].z2, cur, 3, s, l);
            const int lprA = lanes_for_sweep((double)L.nnzA / std::max(1, L.n), L.n);
            for (int sw = 0; sw < nu; ++sw) {
                PF(bytes_sweep(L), lvl("post-sweep", l).c_str(), "k_bcsr_residual<%s, %d, 1, 1, %d>", tname(), lprA, cy16 ? 1 : 0);
                launch_sweep<1>(lprA, L, (const T*)L.r, (const T*)cur, oth, (const T*)(omega_dev + l), s);
                std::swap(cur, oth);
            }
            // nu post-sweeps after nu-1 pre-swaps: the result sits in L.z2 for every nu (odd+odd / even+even swaps)
        }
        launch_prolong(lv[0], nl > 1 ? (const T*)lv[1].z2 : (const T*)z_last, zc, kPoseRec, s, 0);
        if (int rc = launch_cycle_product(slot)) return rc;
        PF(pr.P * (6 + 3 + 3 + 3 + 3) * (double)sizeof(T), "post-smoothing L0", "k_smooth0<%s, 1>", tname());
        hipLaunchKernelGGL((k_smooth0<T, 1>), dim3(nbC), dim3(kBlock), 0, stream, pr.P, (const T*)minv, (const T*)r, (const T*)sbuf, zc, (const T*)omega_dev, s);
        return 0;
    }
    void launch_cg_step(int slot) {
        const T tol2 = (T)(cfg.pcg_rel_tol * cfg.pcg_rel_tol);
        PF(pr.P * (3 + 3 + 6 + 4 * 3 * 2) * (double)sizeof(T), "vector step + pre-smoothing L0", "k_cg_step<%s>", tname());
        hipLaunchKernelGGL((k_cg_step<T>), dim3(nbC), dim3(kBlock), 0, stream, pr.P, (const T*)sbuf, (const T*)(sbuf + (size_t)pr.P * 3), (const T*)rzpart, nbP,
                           (const CgState<T>*)st[slot], st[slot ^ 1], r, p, q, x, zc, (const T*)minv, (const T*)omega_dev, tol2, cfg.pcg_max_iters, (const T*)gscale_dev, kAmgStallIter, (T)kAmgStallRatio,
                           npart, low_cycle && !explicit0 ? zc32 : (float*)nullptr);
    }
    // one PCG iteration reading state slot `slot`, writing slot^1
    int launch_iteration(int slot, int seq = 0) {      // seq > 0: the gate reports to the host thread (do_solve_paced)
        if (amg_on) {
            PF(2.0 * nbC * sizeof(T), "stopping rule", "k_iter_gate<%s>", tname());
            hipLaunchKernelGGL((k_iter_gate<T>), dim3(1), dim3(kBlock), 0, stream, st[slot], (const T*)npart, (const T*)gpart[0], nbC, (T)(cfg.pcg_rel_tol * cfg.pcg_rel_tol),
                               seq > 0 ? h_flag : (int*)nullptr, seq);
            if (int rc = launch_vcycle(slot)) return rc;
            if (int rc = launch_matvec(slot, true)) return rc;
            launch_cg_step(slot);
        } else {
            if (int rc = launch_matvec(slot)) return rc;
            launch_cg_update(slot);
        }
        return 0;
    }
    int chunk() const {
        return amg_on ? kChunkAmg : kChunk;
    }
    void launch_cg_update(int slot) {
        const T tol2 = (T)(cfg.pcg_rel_tol * cfg.pcg_rel_tol);
        PF(pr.P * (3 + 3 + 6 + 4 * 3 * 2) * (double)sizeof(T), "vector step", "k_cg_update<%s>", tname());
        hipLaunchKernelGGL((k_cg_update<T>), dim3(nbC), dim3(kBlock), 0, stream, pr.P, sbuf, sbuf + (size_t)pr.P * 3, nbP, gpart[slot], nbC,
                           gpart[slot ^ 1], st[slot], st[slot ^ 1], minv, r, p, q, x, zc, tol2, cfg.pcg_max_iters, (const T*)gscale_dev);
    }
    // The collective path (eager launches, block-Jacobi PCG, all-reduces between kernels) is taken by every shard of a
    // split graph — and by a single shard that was given a communicator (tsgo_comm_init with world = 1), which is how
    // the RCCL plumbing is exercised on a one-GPU box.
    bool collective() const { return pr.world > 1 || comm != nullptr || lgroup != nullptr; }
#ifdef TSGO_TESTING
    template <typename U> int allreduce_local(U* buf, size_t n) {
        tsgo_local_group& G = *lgroup;
        std::vector<unsigned char>& mine = G.stage[cfg.rank];
        mine.resize(n * sizeof(U));
        HIP_OK(hipMemcpyAsync(mine.data(), buf, n * sizeof(U), hipMemcpyDeviceToHost, stream));
        HIP_OK(hipStreamSynchronize(stream));
        if (!G.barrier()) return set_error(-12, "in-process all-reduce: the other ranks did not arrive (ranks took different decisions?)");   // every rank's contribution is staged
        std::vector<U> sum(n, U(0));
        for (int r = 0; r < G.world; ++r) {
            if (G.stage[r].size() != n * sizeof(U)) return set_error(-12, "in-process all-reduce: ranks disagree on the buffer size");
            const U* src = (const U*)G.stage[r].data();
            for (size_t k = 0; k < n; ++k) sum[k] += src[k];
        }
        if (!G.barrier()) return set_error(-12, "in-process all-reduce: the other ranks did not arrive (ranks took different decisions?)");   // nobody restages while another rank still reads
        HIP_OK(hipMemcpyAsync(buf, sum.data(), n * sizeof(U), hipMemcpyHostToDevice, stream));
        HIP_OK(hipStreamSynchronize(stream));
        return 0;
    }
#else
    template <typename U> int allreduce_local(U*, size_t) { return set_error(-12, "the in-process all-reduce group exists in TSGO_TESTING builds only"); }
#endif
    int allreduce(T* buf, size_t n) {
        if (!collective()) return 0;
        if (lgroup) return allreduce_local(buf, n);
        if (!comm) return set_error(-12, "world > 1 but tsgo_comm_init was not called");
        NCCL_OK(ncclAllReduce(buf, buf, n, sizeof(T) == 8 ? ncclDouble : ncclFloat, ncclSum, comm, stream));
        return 0;
    }
    int allreduce_h(H* buf, size_t n) {       // hierarchy storage type (f32 unless TSGO_HIER_F64)
        if (!collective()) return 0;
        if (lgroup) return allreduce_local(buf, n);
        if (!comm) return set_error(-12, "world > 1 but tsgo_comm_init was not called");
        NCCL_OK(ncclAllReduce(buf, buf, n, sizeof(H) == 8 ? ncclDouble : ncclFloat, ncclSum, comm, stream));
        return 0;
    }
    // One element through the same all-reduce the solver uses, on the engine's stream: every rank contributes rank + 1 and
    // must read world (world + 1) / 2 back.  The first collective of a communicator is where a broken fabric or a missing
    // peer shows (as a hang: bench.py runs this under a watchdog); *ranks_out = what the communicator itself says its size is.
    int comm_selftest(int* ranks_out) override {
        HIP_OK(hipSetDevice(cfg.device));
        int n = 1;
        if (comm) NCCL_OK(ncclCommCount(comm, &n));
#ifdef TSGO_TESTING
        else if (lgroup) n = lgroup->world;
#endif
        if (ranks_out) *ranks_out = n;
        if (!collective()) return 0;
        T* d = nullptr;
        HIP_OK(hipMalloc((void**)&d, sizeof(T)));
        T v = (T)(cfg.rank + 1);
        int rc = 0;
        if (hipMemcpyAsync(d, &v, sizeof(T), hipMemcpyHostToDevice, stream) != hipSuccess) rc = set_error(-10, "tsgo_comm_selftest: copy to the device failed");
        if (!rc) rc = allreduce(d, 1);
        if (!rc && (hipMemcpyAsync(&v, d, sizeof(T), hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess)) rc = set_error(-10, "tsgo_comm_selftest: the all-reduce did not complete");
        (void)hipFree(d);
        if (rc) return rc;
        const double want = 0.5 * n * (n + 1.0);
        if (n != std::max(1, cfg.world) || std::fabs((double)v - want) > 1e-6) return set_error(-12, "tsgo_comm_selftest: " + std::to_string(n) + " ranks in the communicator, world " + std::to_string(cfg.world) + ", sum " + std::to_string((double)v) + " instead of " + std::to_string(want));
        return 0;
    }
    int capture_cg_graph() {
        hipGraph_t graph = nullptr;
        HIP_OK(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
        for (int j = 0; j < chunk(); ++j) if (int rc = launch_iteration(j & 1)) return rc;     // never collective: no RCCL call is captured
        HIP_OK(hipStreamEndCapture(stream, &graph));
        HIP_OK(hipGraphInstantiate(&cg_graph, graph, nullptr, nullptr, 0));
        HIP_OK(hipGraphDestroy(graph));
        return 0;
    }

    // one linearisation; chi2 on the host
    int do_linearize(double* chi2) {
        launch_lin();
        if (int rc = allreduce(part, (size_t)pr.P * 18 + nbP)) return rc;
        launch_finalize();
        if (amg_on) {
            // The Galerkin hierarchy is a preconditioner, not the operator: level 0 (the Schur products, its diagonal
            // inverse) is always the current linearisation, the coarse matrices may lag.  They are rebuilt when they
            // have served hier_max_age solves or the last solve took kHierSlack iterations more than the first one did.
            // ... and a young graph's linearisations move faster than a hierarchy ages (Huber weights switch by the thousand in the first
            // steps from a front-end's estimates: 22 / 23 / 30 iterations on one hierarchy where fresh ones take 22 / 19 / 18 at 100k poses):
            // the first kYoungLins linearisations of a graph share a hierarchy between two at most (profiles/r03z_early_iterations.txt)
            const int max_age = n_lins < kYoungLins ? std::min(hier_max_age, kYoungMaxAge) : hier_max_age;
            ++n_lins;
            const bool refresh = hier_age < 0 || hier_age >= max_age || iters_last > iters_fresh + hier_slack || iters_last > kHierFreshAbove;
            if (refresh) {
                if (int rc = launch_amg_setup()) return rc;
                hier_age = 0;
                if (lin_count++ % kRhoEvery == 0) {
                    if (int rc = estimate_damping()) return rc;
                    if (int rc = launch_bottom_setup()) return rc;      // the dense bottom operator holds the last level's damping
                    launch_finalize();        // zc = omega_0 Minv r with the fresh omega_0
                }
            }
        }
        HIP_OK(hipMemcpyAsync(h_scratch, part + (size_t)pr.P * 18, sizeof(T) * nbP, hipMemcpyDeviceToHost, stream));
        HIP_OK(hipStreamSynchronize(stream));
        double s = 0;
        for (int k = 0; k < nbP; ++k) s += (double)h_scratch[k];
        *chi2 = s;
        return 0;
    }

    // PCG; if the multigrid-preconditioned solve breaks down (indefinite preconditioner), the solve is
    // repeated from the same right-hand side with the block-Jacobi preconditioner.
    int n_fallbacks = 0, n_hier_retries = 0;
    bool inject_armed = false;
    // Warm start (cfg.warm_start): the Gauss-Newton update takes kStepScale of the solved delta, so (1 - kStepScale) of it
    // is still to go at the next linearisation.  x0 = that remainder, r = b~ - S x0 (one extra product), and the stopping
    // rule keeps measuring against the right-hand side: gamma0 is scaled by (b^T D^-1 b) / (r0^T D^-1 r0).
    int launch_warm() {
        // x0 = the un-taken remainder of the previous step, (1 - step) d1, at order 1; higher orders continue the trend of the last
        // deltas as well (order 2: (1 - step) (d1 + c1) with c1 = d1 - (1 - step) d2, what the last step added over ITS prediction).
        // cfg.warm_start caps the order; below the cap the kernel takes the order that would have predicted the last delta best
        // (k_save_x measured every order then): high orders win once the iteration is smooth (50 iterations at 100 k poses:
        // 849 PCG iterations at order 2, 724 at order 6) and lose while Huber weights still switch (profiles/r03w_warm_start_order.txt).
        WarmTerms<T> w{};
        warm_coefficients(w);
        for (int j = 0; j < kMaxWarm; ++j) w.v[j] = hist[j];
        w.n_max = std::max(1, std::min({n_prev, (int)cfg.warm_start, kMaxWarm}));
        w.n_tested = std::min(n_tested, w.n_max);
        w.errpart = warm_err; w.nb_err = nbC;
        hipLaunchKernelGGL((k_pack_x<T>), dim3(nbC), dim3(kBlock), 0, stream, pr.P, x, zc, w, warm_order_dev);
        if (int rc = launch_matvec(0)) return rc;
        hipLaunchKernelGGL((k_warm_residual<T>), dim3(nbC), dim3(kBlock), 0, stream, pr.P, (const T*)sbuf, (const T*)minv, r, zc, (const T*)(amg_on ? omega_dev : one_dev), npart, amg_on && low_cycle ? zc32 : (float*)nullptr);
        hipLaunchKernelGGL((k_warm_scale<T>), dim3(1), dim3(kBlock), 0, stream, nbC, (const T*)gpart[0], (const T*)npart, amg_on ? (T*)nullptr : gpart[0], gscale_dev,
                           amg_on ? st[0] : (CgState<T>*)nullptr, (T)(cfg.pcg_rel_tol * cfg.pcg_rel_tol));
        return 0;
    }
    int do_solve(int* iters, int* fail) {
        const bool warmed = cfg.warm_start && have_prev && step_scale() < 1.0;      // a full step (rules = 1, lr = 1) leaves no remainder to start from
        if (warmed) { if (int rc = launch_warm()) return rc; }
        if (warmed && carried) {
            // The history is the previous REQUEST's: it continues this one only if the client sent back the estimates it was
            // returned.  k_warm_scale leaves b'D^-1 b / r0'D^-1 r0, or 1 when the start is no better than zero: then the history
            // is dropped and the solve starts cold (every shard reads the same all-reduced numbers and decides alike).
            T gs = 0;
            if (int rc = copy_sync(&gs, gscale_dev, sizeof(T), hipMemcpyDeviceToHost)) return rc;
            if (!(gs > T(1))) { launch_finalize(); have_prev = false; n_prev = 0; n_tested = 0; ++n_carry_dropped; n_lins = 1; }      // ... and the graph is a young one after all
        }
        carried = false;
        if (int rc = do_solve_once(iters, fail)) return rc;
        const bool warm_trace = solve_timing;
        if (warm_trace && warmed) {
            int order = 0; T gs = 0; std::vector<T> e((size_t)kMaxWarm * nbC);
            if (int rc = copy_sync(&order, warm_order_dev, sizeof(int), hipMemcpyDeviceToHost)) return rc;
            if (int rc = copy_sync(&gs, gscale_dev, sizeof(T), hipMemcpyDeviceToHost)) return rc;
            if (int rc = copy_sync(e.data(), warm_err, e.size() * sizeof(T), hipMemcpyDeviceToHost)) return rc;
            std::fprintf(stderr, "[tsgo] warm start: order %d of %d tested, b'D^-1 b / r0'D^-1 r0 = %.3e, %d PCG iterations; prediction errors of the last delta:", order, n_tested, (double)gs, *iters);
            for (int m = 0; m < n_tested; ++m) { double s = 0; for (int k = 0; k < nbC; ++k) s += (double)e[(size_t)m * nbC + k]; std::fprintf(stderr, " %.3e", s); }
            std::fprintf(stderr, "\n");
        }
        if (*fail == 3 && amg_on) *fail = 1;          // stagnation under the multigrid cycle
        if (amg_on && cy16 && (*fail != 0 || *iters > kPackedCycleMaxIters)) {
            // Packed half floats round every block of the cycle's operators to 11 bits.  A coarse operator of a nearly singular
            // system (an odometry-only chain under the analytic Jacobians: a 24k-link beam) is a difference of large entries; at
            // that precision it stops being positive definite and the solve breaks down or crawls (20 000 iterations where the
            // f32 copies need 1 500; profiles/r03k_hard_chain.txt).  Such a graph shows itself by its iteration count: from here
            // on this structure's cycle reads f32 copies (sticky until a new structure arrives), and a solve that failed is
            // repeated with them.
            cy16 = false; ++n_cycle_f32_switches;
            if (cg_graph) { (void)hipGraphExecDestroy(cg_graph); cg_graph = nullptr; optimize_calls_on_tables = 1; }     // it holds the packed kernels; re-captured at the next tsgo_optimize
            if (say_env) std::fprintf(stderr, "[tsgo] %d PCG iterations (fail %d) with the packed cycle format: this structure's cycle switches to f32 copies\n", *iters, *fail);
            if (int rc = launch_amg_setup()) return rc;
            hier_age = 0; iters_fresh = 0;
            if (*fail != 0) {
                launch_finalize();
                if (int rc = do_solve_once(iters, fail)) return rc;
                if (*fail == 3) *fail = 1;
            }
        }
        const int age_used = hier_age;
        if (amg_on) { iters_last = *iters; if (hier_age == 0) iters_fresh = *iters; if (hier_age >= 0) ++hier_age; }
        if (*fail == 1 && amg_on && age_used > 0) {
            // The hierarchy that failed was built for an earlier linearisation (the lag rule).  Before giving the multigrid
            // cycle up for this solve, build it for THIS one and solve again from the plain right-hand side: on beam-like
            // odometry chains (analytic Jacobians) a lagged hierarchy can be indefinite where a fresh one takes 1 400 iterations
            // and block-Jacobi does not finish in 20 000.
            ++n_hier_retries;
            if (int rc = launch_amg_setup()) return rc;
            hier_age = 0;
            launch_finalize();
            if (int rc = do_solve_once(iters, fail)) return rc;
            if (*fail == 3) *fail = 1;
            iters_last = *iters; iters_fresh = *iters; hier_age = 1;
        }
        // test hook: TSGO_INJECT_AMG_FAILURE=1 treats the first multigrid solve of every tsgo_optimize call as broken down, so that the
        // block-Jacobi repeat below runs on a graph where the cycle is perfectly healthy (tests/test_gpu_parity.py)
        const bool inject = hook_inject_amg_failure;
        if (inject && amg_on && inject_armed) { inject_armed = false; *fail = 1; }
        if (*fail == 1 && amg_on) {
            ++n_fallbacks;
            hier_age = -1;                               // whatever went wrong, start from a fresh hierarchy next time
            const bool keep = amg_on; hipGraphExec_t g = cg_graph;
            amg_on = false; cg_graph = nullptr;          // eager block-Jacobi launches
            launch_finalize();
            const int rc = do_solve_once(iters, fail);
            amg_on = keep; cg_graph = g;
            return rc;
        }
        return 0;
    }
    // Eager launches, one device, multigrid cycle: the host thread stays ONE iteration ahead of the device instead of predicting a burst.
    // The gate (first kernel) of every iteration writes to pinned host memory that it has run and what it saw; the host enqueues
    // iteration j + 1 when the gate of iteration j has run (90 us of launches against the 217 us the device then spends on j) and
    // stops at the first gate that reports `done`.  What is wasted past convergence is the rest of that one iteration (29 kernels that
    // exit at once, 43 us) instead of the two or three a predicted burst over-provisions, an under-provisioned burst (the device idle
    // while the host enqueues more: 0.4 ms) cannot happen, and the iteration count and failure flag arrive with the report: the stream is
    // not drained at the end of the solve — the back-substitution queues up behind the last exits (profiles/r03z_paced_eager.txt).
    double ref_us_per_iter = 0;        // the device's time per iteration as the burst path measured it (the structure's first solves)
    int n_paced_slow = 0;
    uint32_t paced_base = 0;           // sequence numbers grow ACROSS solves: a gate of an earlier solve that is still queued when a retry
                                       // starts the next one (pace_lead > 1, hierarchy / block-Jacobi repeats) reports a number <= base and is ignored
    int do_solve_paced(int* iters, int* fail) {
        uint64_t* hw = reinterpret_cast<uint64_t*>(h_flag);
        if (paced_base > 0x70000000u) {      // (wrap-around: once per 4e9 iterations) nothing may be in flight when the numbering restarts
            HIP_OK(hipStreamSynchronize(stream));
            paced_base = 0; __atomic_store_n(hw, (uint64_t)0, __ATOMIC_SEQ_CST);
        }
        const uint32_t base = paced_base;
        const bool timing = solve_timing;
        const auto w0 = std::chrono::steady_clock::now();
        auto since = [&] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - w0).count(); };
        int launched = 0, seen_last = 0; long spins = 0;
        double t_last_gate = 0;        // when the newest gate was seen (or the last launch made): a healthy long solve keeps moving this
        uint64_t w = 0;
        for (;;) {
            w = __atomic_load_n(hw, __ATOMIC_ACQUIRE);      // seq << 32 | done << 31 | fail << 28 | iterations completed (k_iter_gate)
            const int32_t rel = (int32_t)((uint32_t)(w >> 32) - base);
            const int seen = rel > 0 ? (int)rel : 0;          // reports of earlier solves count as "nothing seen yet"
            if (seen != seen_last) { seen_last = seen; t_last_gate = since(); }
            if (seen > 0 && ((w >> 31) & 1)) break;
            if (launched - seen < pace_lead) {
                if (launched > cfg.pcg_max_iters + 4) { paced_base = base + (uint32_t)launched; return set_error(-20, "PCG did not terminate"); }
                if (int rc = launch_iteration(launched & 1, (int)(base + (uint32_t)launched + 1u))) { paced_base = base + (uint32_t)launched + 1u; return rc; }
                ++launched; spins = 0; t_last_gate = since();
            } else {
                __builtin_ia32_pause();
                if ((spins & 0x3f) == 0x3f) std::this_thread::yield();      // the device needs ~120 us before the next iteration has to be on its way: other threads may have the core
                if ((++spins & 0xfffff) == 0 && since() - t_last_gate > 30e6) { paced_base = base + (uint32_t)launched; return set_error(-20, "PCG: the device stopped reporting (30 s without a gate)"); }
            }
        }
        paced_base = base + (uint32_t)launched;
        *iters = (int)(w & 0x0fffffffu); *fail = (int)((w >> 28) & 7);
        const double wall = since();
        if (timing) std::fprintf(stderr, "[tsgo] solve (paced): %d iterations launched, done reported after %d at %.0f us\n", launched, *iters, wall);
        predicted_cg = *iters;
        if (amg_on && hier_age >= 0 && hier_age < kAgeSlots) iters_by_age[hier_age] = *fail ? 0 : *iters;
        // a host that cannot stay ahead shows as iterations that take longer than the burst path measured: then the handle goes over to replay
        if (cfg.use_graphs == 2 && *iters >= 8 && !*fail && ref_us_per_iter > 0 && !hook_force_paced) {
            if (wall / *iters > 1.3 * ref_us_per_iter) { if (++n_paced_slow >= 3) host_slow = true; } else n_paced_slow = 0;
        }
        return 0;
    }
    // (only where the structure's first solves showed a host with room to spare: on a 150-pose graph an iteration is 20 kernels at the floor, 74 us,
    // and waiting for a gate before enqueueing the next iteration would expose the host's 60 us every time: use_graphs = 0 keeps the predicted burst there)
    bool paced() const {
        if (!amg_on || cg_graph || collective() || cfg.use_graphs == 1 || prof_on || h_flag == nullptr) return false;
        if (hook_force_paced) return true;       // test hook (TSGO_TESTING builds): the paced path from the first solve on
        return n_decided >= kDecideSolves && 2 * n_slow_seen <= kDecideSolves && !host_slow;
    }

    // PCG until the device state says done.  The state ring is at slot 0 on entry and on exit.
    int do_solve_once(int* iters, int* fail) {
        if (paced()) return do_solve_paced(iters, fail);
        int launched = 0;
        const int ch = chunk();
        // chunks before the first look at the device state.  An iteration past convergence costs ~40 us (its kernels exit
        // at once), a look costs about as much plus an idle gap: under the multigrid cycle, where counts move by one or
        // two between solves, the burst aims one iteration past the prediction (a solve on a fresh hierarchy is predicted
        // by the last fresh one, an aged one by the previous solve + 1); block-Jacobi counts are in the thousands and
        // drift, so that burst stops at 90 %.
        // Under the lag rule the counts repeat from one hierarchy to the next (15 15 16 17 | 15 15 16 17 at 100k poses): the best
        // predictor of a solve is the solve of the same age on the previous hierarchy.
        const int by_age = (amg_on && hier_age >= 0 && hier_age < kAgeSlots) ? iters_by_age[hier_age] : 0;
        const int pred = amg_on ? (by_age > 0 ? by_age : (hier_age == 0 && iters_fresh > 0 ? iters_fresh : predicted_cg + 1)) : predicted_cg;
        int burst = amg_on ? std::max(1, (pred + 1 + ch - 1) / ch) : std::max(1, (int)(0.9 * pred) / ch);
        const bool timing = solve_timing;
        const auto w0 = std::chrono::steady_clock::now();
        auto since = [&] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - w0).count(); };
        for (;;) {
            for (int b = 0; b < burst; ++b) {
                if (cg_graph) HIP_OK(hipGraphLaunch(cg_graph, stream));
                else for (int j = 0; j < ch; ++j) if (int rc = launch_iteration(j & 1)) return rc;
                launched += ch;
            }
            const double t_enq = since();
            if (timing) std::fprintf(stderr, "[tsgo] solve: %d chunk(s) enqueued at %.0f us", burst, t_enq);
            if (!cg_graph && amg_on && burst >= 4 && dev_us_per_iter > 0) {      // an eager burst of >= 8 iterations: was the host well ahead of the device?
                const double host_us_per_iter = t_enq / (burst * ch);
                const double share = host_us_per_iter / dev_us_per_iter;
                if (n_decided < kDecideSolves) {      // the structure's first solves settle it (within a bench's warm-up, a request's first iterations): no flip in mid-run
                    n_slow_seen += share > kHostSlowFraction;
                    if (++n_decided == kDecideSolves && 2 * n_slow_seen > kDecideSolves) host_slow = true;
                } else if (share > 0.95) { if (++n_host_slow >= 3) host_slow = true; } else n_host_slow = 0;      // later only outright starvation (a profiler attached, cores taken away)
            }
            if (cg_graph) replayed = true;
            burst = 1;
            HIP_OK(hipMemcpyAsync(h_state, st[0], sizeof(CgState<T>), hipMemcpyDeviceToHost, stream));
            HIP_OK(hipStreamSynchronize(stream));
            if (timing) std::fprintf(stderr, ", drained at %.0f us (iters %d done %d)\n", since(), h_state->iters, h_state->done);
            if (h_state->done) break;
            if (launched > cfg.pcg_max_iters + 2 * ch) return set_error(-20, "PCG did not terminate");
        }
        *iters = h_state->iters; *fail = h_state->fail;
        predicted_cg = h_state->iters;
        if (amg_on && h_state->iters >= 4 && !h_state->fail) { dev_us_per_iter = since() / h_state->iters; if (!cg_graph) ref_us_per_iter = ref_us_per_iter > 0 ? std::min(ref_us_per_iter, dev_us_per_iter) : dev_us_per_iter; }      // (an upper bound: the solve's wall time over its iterations)
        if (amg_on && hier_age >= 0 && hier_age < kAgeSlots) iters_by_age[hier_age] = h_state->fail ? 0 : h_state->iters;
        return 0;
    }

    // landmarks: dl = u - Dl^-1 W^T x (+ optional update); poses: update.  Returns ||delta_p||^2 (identical on every
    // rank: pose vectors are replicated) and THIS rank's ||delta_l||^2 (landmark deltas are shard-local).
    int do_backsub_update(T step, double* np2_out, double* nl2_local_out) {
        const int P = pr.P;
        if (step != T(0)) {
            std::rotate(hist, hist + kMaxWarm - 1, hist + kMaxWarm);      // the oldest buffer takes this delta; hist[1..] are the deltas before it
            WarmTerms<T> w{};
            warm_coefficients(w);
            for (int j = 0; j + 1 < kMaxWarm; ++j) w.v[j] = hist[j + 1];
            w.n_max = std::min({n_prev, (int)cfg.warm_start, kMaxWarm - 1});      // orders that can be tested on this delta
            hipLaunchKernelGGL((k_save_x<T>), dim3(nbC), dim3(kBlock), 0, stream, P, (const T*)x, zc, hist[0], w, warm_err);
            n_tested = w.n_max; have_prev = true; n_prev = std::min(n_prev + 1, kMaxWarm);
        } else {      // a probe (step 0) leaves nothing to carry over
            hipLaunchKernelGGL((k_pack_x<T>), dim3(nbC), dim3(kBlock), 0, stream, P, x, zc, WarmTerms<T>{}, (int*)nullptr);
            have_prev = false; n_prev = 0; n_tested = 0;
        }
        if (tl.n_slices > 0) LAUNCH_GM(pr.by_lm.G, k_schur_lm, 1, nbL, stream, tl, zc, lmrec, (const T*)ninv, tvec, st[0], step, dl, npart + nbC);
        hipLaunchKernelGGL((k_pose_update<T>), dim3(nbC), dim3(kBlock), 0, stream, P, x, ps, theta, step, npart);
        const int nl = tl.n_slices > 0 ? nbL : 0;
        HIP_OK(hipMemcpyAsync(h_scratch, npart, sizeof(T) * (size_t)(nbC + nl), hipMemcpyDeviceToHost, stream));
        HIP_OK(hipStreamSynchronize(stream));
        double np2 = 0, nl2 = 0;
        for (int k = 0; k < nbC; ++k) np2 += (double)h_scratch[k];
        for (int k = 0; k < nl; ++k) nl2 += (double)h_scratch[nbC + k];
        *np2_out = np2; *nl2_local_out = nl2;
        return 0;
    }
    // Sum of the ranks' landmark-delta norms.  The stop rule ||delta|| < 1e-3 (OptimizerCpu.h:173) can only fire when the
    // pose part alone is already below the tolerance, and the pose part is known on every rank: this collective runs
    // when that happens and once at the end of tsgo_optimize for the reported norm, not once per iteration.
    int landmark_norm_allreduce(double* nl2) {
        if (!collective()) return 0;
        T* d = npart;         // one-element device scratch (its partials were consumed by do_backsub_update)
        T v = (T)*nl2;
        HIP_OK(hipMemcpyAsync(d, &v, sizeof(T), hipMemcpyHostToDevice, stream));
        if (int rc = allreduce(d, 1)) return rc;
        HIP_OK(hipMemcpyAsync(&v, d, sizeof(T), hipMemcpyDeviceToHost, stream));
        HIP_OK(hipStreamSynchronize(stream));
        *nl2 = (double)v;
        return 0;
    }

    int optimize_calls_on_tables = 0;       // tsgo_optimize calls since the tables were built (lazy hipGraph capture, see set_graph)
    int optimize(int iterations, tsgo_stats* out) override {
        if (!have_graph_data) return set_error(-3, "tsgo_optimize: no graph set");
        HIP_OK(hipSetDevice(cfg.device));
        // use_graphs 1: the PCG iterations are replayed from a captured hipGraph (from the second tsgo_optimize on these tables on).  2
        // (default): eager launches while the host thread enqueues an iteration in well under the time the device takes to run it
        // (3 us per launch against 7 on an EPYC 9575F: eager is then 1-3 % FASTER than the replay and steadier, profiles/r03z_eager_vs_graph.txt);
        // a host that cannot keep that distance (busy cores, a slow clock, a profiler; or a graph of 10k poses, whose iteration the device runs in 99 us)
        // is noticed by do_solve_once and the handle goes over to replay.
        if (hook_force_host_slow) host_slow = true;      // test hook (TSGO_TESTING builds)
        const bool want_graph = cfg.use_graphs == 1 || (cfg.use_graphs == 2 && (host_slow || !amg_on));      // (block-Jacobi PCG is two short kernels per iteration, thousands of times: always replayed)
        if (want_graph && !collective() && !cg_graph && optimize_calls_on_tables >= 1) { if (int rc = capture_cg_graph()) return rc; }
        inject_armed = true;
        ++optimize_calls_on_tables;
        tsgo_stats s; std::memset(&s, 0, sizeof(s));
        s.n_pose = pr.P; s.n_lm = pr.L_total; s.n_odom_edges = pr.n_odom_edges_total; s.n_lm_edges = pr.n_lm_edges_total;
        s.ms_setup = ms_setup; s.structure_reused = last_set_reused ? 1 : 0;
        double prevErr = -1; int penalty = 0;
        bool nl2_whole = true;        // sharded: last_delta_norm holds every rank's landmark part (see landmark_norm_allreduce)
        double np2_last = 0, nl2_last = 0;
        const int fallbacks0 = n_fallbacks, dropped0 = n_carry_dropped;
        const bool started_carried = carried;
        replayed = false;
        s.stop_reason = TSGO_STOP_CAP;
        const auto wall0 = std::chrono::steady_clock::now();
        // rules = 1 (graph_optimizer.py:24-31): lambda starts at 1e-3 on every call
        const double lam_max = 1e1, lam_min = 1e-6, lam_fac = 1.1;
        double lam = 1e-3;
        const double step = step_scale();
        for (int it = 0; it < iterations; ++it) {
            double err = 0; float ms = 0;
            HIP_OK(hipEventRecord(ev[0], stream));
            if (py_rules()) {
                // lambda follows the chi^2 of THIS linearisation (:41-42), which the linearisation itself needs: it is run with the
                // value a non-increasing chi^2 gives (the common case) and repeated with the other one when chi^2 did rise
                lambda = std::max(lam / lam_fac, lam_min);
                if (int rc = do_linearize(&err)) return rc;
                if (prevErr > -1 && err > prevErr) { lambda = std::min(lam * lam_fac, lam_max); if (int rc = do_linearize(&err)) return rc; }
                lam = lambda; s.lambda_last = lam;
            } else {
                lambda = 0;
                if (int rc = do_linearize(&err)) return rc;
            }
            HIP_OK(hipEventRecord(ev[1], stream));
            if (it < TSGO_MAX_TRACE) s.chi2[it] = err;
            s.chi2_last = err;
            s.iterations_run = it + 1; s.trace_len = std::min(it + 1, TSGO_MAX_TRACE);
            if (!py_rules()) {
                if (prevErr > 0 && err > prevErr) {                          // OptimizerCpu.h:140-153
                    if (++penalty > 2) { s.stop_reason = TSGO_STOP_WORSE; break; }
                } else penalty = 0;
            }
            int cg = 0, fail = 0;
            if (int rc = do_solve(&cg, &fail)) return rc;
            HIP_OK(hipEventRecord(ev[2], stream));
            if (it < TSGO_MAX_TRACE) s.pcg_iters[it] = cg;
            s.pcg_iters_total += cg;
            if (fail != 0) { s.stop_reason = TSGO_STOP_SOLVER; break; }       // breakdown, or pcg_max_iters reached without convergence: no step is taken
            double np2 = 0, nl2 = 0;
            if (int rc = do_backsub_update((T)step, &np2, &nl2)) return rc;   // :159-165 / graph_optimizer.py:66-75
            HIP_OK(hipEventRecord(ev[3], stream));
            HIP_OK(hipEventSynchronize(ev[3]));
            HIP_OK(hipEventElapsedTime(&ms, ev[0], ev[1])); s.ms_linearize += ms;
            HIP_OK(hipEventElapsedTime(&ms, ev[1], ev[2])); s.ms_solve += ms;
            HIP_OK(hipEventElapsedTime(&ms, ev[2], ev[3])); s.ms_update += ms;
            nl2_whole = !collective();
            const bool last = it + 1 == iterations;
            const bool plateau = !py_rules() && std::fabs(err - prevErr) < kPlateauTol;
            // the norm the stop rule looks at: ||delta|| (OptimizerCpu.h:173) or ||lr * dx|| (graph_optimizer.py:66,90)
            const double norm_scale = py_rules() ? step : 1.0;
            if (collective() && (last || plateau || norm_scale * norm_scale * np2 < kDeltaTol * kDeltaTol)) {      // every rank takes this branch or none does
                if (int rc = landmark_norm_allreduce(&nl2)) return rc;
                nl2_whole = true;
            }
            s.last_delta_norm = norm_scale * std::sqrt(np2 + nl2);
            np2_last = np2; nl2_last = nl2;
            if (plateau) { s.stop_reason = TSGO_STOP_PLATEAU; break; }                                  // :167-171
            if (nl2_whole && s.last_delta_norm < kDeltaTol) { s.stop_reason = TSGO_STOP_CONVERGED; break; }   // :173-177 / py :90-92
            prevErr = err;                                                                              // :179 / py :44
        }
        if (!nl2_whole) {             // the loop ended before its last update's landmark norm was summed (worse / solver stop)
            if (int rc = landmark_norm_allreduce(&nl2_last)) return rc;
            s.last_delta_norm = (py_rules() ? step_scale() : 1.0) * std::sqrt(np2_last + nl2_last);
        }
        s.pcg_fallbacks = n_fallbacks - fallbacks0;
        s.history_carried = started_carried ? (n_carry_dropped > dropped0 ? 2 : 1) : 0;
        s.graph_replay = replayed ? 1 : 0;
        s.cycle_storage_now = amg_on ? (cy16 ? 16 : 32) : 0;
        s.ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
        if (out) *out = s;
        return 0;
    }

    int get_vertices(double* out) override {
        if (!have_graph_data) return set_error(-3, "tsgo_get_vertices: no graph set");
        HIP_OK(hipSetDevice(cfg.device));
        const int P = pr.P, L = pr.L;
        std::vector<T> hp((size_t)P * 4), hl((size_t)std::max(L, 1) * kLmRec);
        { if (int rc_ = copy_sync(hp.data(), ps, hp.size() * sizeof(T), hipMemcpyDeviceToHost)) return rc_; }
        if (L) { if (int rc_ = copy_sync(hl.data(), lmrec, (size_t)L * kLmRec * sizeof(T), hipMemcpyDeviceToHost)) return rc_; }
        for (int i = 0; i < P; ++i) {
            const int v = pr.pose_vertex[i];
            out[3 * (size_t)v] = hp[4 * (size_t)i]; out[3 * (size_t)v + 1] = hp[4 * (size_t)i + 1];
            out[3 * (size_t)v + 2] = std::atan2((double)hp[4 * (size_t)i + 3], (double)hp[4 * (size_t)i + 2]);   // SerializeGraphFuncCpu.h:28
        }
        for (int l = 0; l < L; ++l) {
            const int v = pr.lm_vertex[l];
            out[3 * (size_t)v] = hl[(size_t)l * kLmRec]; out[3 * (size_t)v + 1] = hl[(size_t)l * kLmRec + 1]; out[3 * (size_t)v + 2] = 0;
        }
        return 0;
    }

    int linearize(double* diag, double* grad, double* chi2) override {
        if (!have_graph_data) return set_error(-3, "tsgo_linearize: no graph set");
        HIP_OK(hipSetDevice(cfg.device));
        if (int rc = do_linearize(chi2)) return rc;
        const int P = pr.P, L = pr.L;
        std::vector<T> hpart((size_t)P * 18), hl((size_t)std::max(L, 1) * kLmRec);
        { if (int rc_ = copy_sync(hpart.data(), part, hpart.size() * sizeof(T), hipMemcpyDeviceToHost)) return rc_; }
        if (L) { if (int rc_ = copy_sync(hl.data(), lmrec, (size_t)L * kLmRec * sizeof(T), hipMemcpyDeviceToHost)) return rc_; }
        std::memset(diag, 0, sizeof(double) * 9 * (size_t)pr.n_vertices);
        std::memset(grad, 0, sizeof(double) * 3 * (size_t)pr.n_vertices);
        for (int i = 0; i < P; ++i) {
            const T* o = &hpart[(size_t)i * 18];
            double* d = diag + 9 * (size_t)pr.pose_vertex[i]; double* g = grad + 3 * (size_t)pr.pose_vertex[i];
            d[0] = o[0]; d[1] = d[3] = o[1]; d[2] = d[6] = o[2]; d[4] = o[3]; d[5] = d[7] = o[4]; d[8] = o[5];
            g[0] = o[6]; g[1] = o[7]; g[2] = o[8];
        }
        for (int l = 0; l < L; ++l) {
            const T* o = &hl[(size_t)l * kLmRec];
            double ixx = o[2], ixy = o[3], iyy = o[4], dxx, dxy, dyy;
            inv_sym2<double>(ixx, ixy, iyy, dxx, dxy, dyy);       // Dl = (Dl^-1)^-1
            double* d = diag + 9 * (size_t)pr.lm_vertex[l]; double* g = grad + 3 * (size_t)pr.lm_vertex[l];
            d[0] = dxx; d[1] = d[3] = dxy; d[4] = dyy;
            g[0] = dxx * o[5] + dxy * o[6]; g[1] = dxy * o[5] + dyy * o[6];
        }
        return 0;
    }

    int solve_step(double* delta, double* chi2, int* iters) override {
        if (!have_graph_data) return set_error(-3, "tsgo_solve_step: no graph set");
        HIP_OK(hipSetDevice(cfg.device));
        if (int rc = do_linearize(chi2)) return rc;
        int cg = 0, fail = 0;
        if (int rc = do_solve(&cg, &fail)) return rc;
        if (iters) *iters = cg;
        // back-substitute with step 0: state untouched (theta is re-derived from the same cos/sin)
        double np2 = 0, nl2 = 0;
        if (int rc = do_backsub_update((T)0, &np2, &nl2)) return rc;
        const int P = pr.P, L = pr.L;
        std::vector<T> hx((size_t)P * 3), hd((size_t)std::max(L, 1) * 2);
        { if (int rc_ = copy_sync(hx.data(), x, hx.size() * sizeof(T), hipMemcpyDeviceToHost)) return rc_; }
        if (L) { if (int rc_ = copy_sync(hd.data(), dl, (size_t)L * 2 * sizeof(T), hipMemcpyDeviceToHost)) return rc_; }
        std::memset(delta, 0, sizeof(double) * 3 * (size_t)pr.n_vertices);
        for (int i = 0; i < P; ++i) for (int k = 0; k < 3; ++k) delta[3 * (size_t)pr.pose_vertex[i] + k] = hx[(size_t)i * 3 + k];
        for (int l = 0; l < L; ++l) for (int k = 0; k < 2; ++k) delta[3 * (size_t)pr.lm_vertex[l] + k] = hd[(size_t)l * 2 + k];
        return fail == 1 ? set_error(-21, "PCG breakdown") : (fail != 0 ? set_error(-22, "PCG did not converge within pcg_max_iters") : 0);
    }

    int cycle_probe(int reps, tsgo_cycle_level* out, int cap) override {
        if (!have_graph_data) return set_error(-3, "tsgo_cycle_probe: no graph set");
        HIP_OK(hipSetDevice(cfg.device));
        if (!amg_on) return 0;
        double chi2;
        if (int rc = do_linearize(&chi2)) return rc;      // a built hierarchy; state slot 0 says "not done"
        int n = 0;
        for (size_t l = 1; l < lv.size() && n < cap; ++l, ++n) {
            DevLevel<T>& L = lv[l];
            const int lprA = lanes_for_sweep((double)L.nnzA / std::max(1, L.n), L.n);
            for (int pass = 0; pass < 2; ++pass) {
                const int m = pass == 0 ? 3 : reps;
                HIP_OK(hipEventRecord(ev[0], stream));
                for (int k = 0; k < m; ++k)
                    launch_sweep<1>(lprA, L, (const T*)L.r, (const T*)L.z, L.z2, (const T*)(omega_dev + l), (const CgState<T>*)st[0]);
                HIP_OK(hipEventRecord(ev[1], stream));
                HIP_OK(hipEventSynchronize(ev[1]));
                if (pass == 1) { float ms = 0; HIP_OK(hipEventElapsedTime(&ms, ev[0], ev[1])); out[n].us_per_sweep = 1e3 * ms / m; }
            }
            out[n].rows = L.n; out[n].blocks = L.nnzA; out[n].lanes_per_row = lprA;
            out[n].sweeps_per_cycle = 2 * nu_at(l);         // (nu - 1) pre-sweeps + the residual + nu post-sweeps
            out[n].bytes_per_sweep = bytes_sweep(L);
        }
        return n;
    }

    // One PCG iteration kernel by kernel, in situ: `reps` iterations launched eagerly, the stopping test disabled, an event before
    // every launch (PF()).  Entry k of the result = the k-th launch of an iteration, averaged over the iterations.
    int profile_iteration(int reps, tsgo_prof_entry* out, int cap) override {
        if (!have_graph_data) return set_error(-3, "tsgo_profile_iteration: no graph set");
        HIP_OK(hipSetDevice(cfg.device));
        double chi2;
        if (int rc = do_linearize(&chi2)) return rc;      // valid operands, a built hierarchy; state slot 0 says "not done"
        struct TolGuard { double& tol; double keep; ~TolGuard() { tol = keep; } } tol_guard{cfg.pcg_rel_tol, cfg.pcg_rel_tol};
        cfg.pcg_rel_tol = 0;
        reps = std::max(2, reps + (reps & 1));           // whole pairs: the state ring has two slots
        for (int j = 0; j < 4; ++j) if (int rc = launch_iteration(j & 1)) return rc;        // warm caches and clocks
        // Eager launches + event records are host-bound (~7 us each against kernels of 4-15 us): the stream is first blocked by a
        // kernel that waits kProfBlockMs on the constant-rate clock, the host enqueues everything behind it, and the device then
        // runs the queue back to back — what a hipGraph replay of the same iterations does.
        hipLaunchKernelGGL(k_wait_ms, dim3(1), dim3(64), 0, stream, (int)(kProfBlockUsPerLaunch * 40.0 * reps / 1000.0) + 2);
        prof_n = 0; prof_on = true;
        int rc = 0;
        for (int j = 0; j < reps && rc == 0; ++j) rc = launch_iteration(j & 1);
        PF(0, "", "end");
        prof_on = false;
        HIP_OK(hipStreamSynchronize(stream));
        if (rc) return rc;
        const size_t marks = prof_n - 1;
        if (marks == 0 || marks % (size_t)reps != 0) return set_error(-30, "tsgo_profile_iteration: the iterations did not launch the same kernels");
        const size_t per = marks / (size_t)reps;
        const int n = (int)std::min<size_t>(per, (size_t)cap);
        for (int k = 0; k < n; ++k) {
            double sum = 0;
            for (int j = 0; j < reps; ++j) {
                float ms = 0;
                HIP_OK(hipEventElapsedTime(&ms, prof[(size_t)j * per + k].e, prof[(size_t)j * per + k + 1].e));
                sum += ms;
            }
            const ProfMark& m = prof[k];
            std::memset(&out[k], 0, sizeof(out[k]));
            std::snprintf(out[k].name, sizeof(out[k].name), "%s", m.name);
            std::snprintf(out[k].where, sizeof(out[k].where), "%s", m.where);
            out[k].launches_per_iteration = 1; out[k].us = 1e3 * sum / reps; out[k].bytes = m.bytes;
        }
        cfg.pcg_rel_tol = tol_guard.keep;
        if (int rc2 = do_linearize(&chi2)) return rc2;   // leave a consistent state behind
        return n;
    }

    // which: 0 schur_lm, 1 schur_pose, 2 cg_update, 3 lin_lm, 4 lin_pose, 5 one whole PCG iteration
    int time_kernel(int which, int reps, double* us, double* bytes) override {
        if (!have_graph_data) return set_error(-3, "tsgo_time_kernel: no graph set");
        HIP_OK(hipSetDevice(cfg.device));
        double chi2;
        if (int rc = do_linearize(&chi2)) return rc;      // valid operands; state slot 0 says "not done"
        const double s = sizeof(T);
        const double El = (double)pr.n_lm_edges, P = pr.P, L = pr.L;
        double od = 0; for (uint32_t e : pr.odom.edge) od += e != kNoEdge;
        const double b_lm = El * (4 + 4 * s) + P * 5 * s + L * 5 * s;
        const double b_pose = El * (4 + 4 * s) + L * 2 * s + P * (5 + 6 + 3) * s + od * (4 + 3 * s + 3 * s);
        const double b_upd = P * (3 + 3 + 6 + 4 * 3 * 2 - 3) * s;   // sz, z in; minv in; r p q x in+out (x,r,p,q), z out
        const double b_linlm = El * (4 + 4 * s + 4 * s) + P * 4 * s + L * (2 + 5 + 3) * s;
        const double b_linpose = El * (4 + 4 * s + 4 * s) + L * 7 * s + P * (4 + 18) * s + od * (4 + 9 * s + 3 * s);
        // whole iterations are timed with the stopping test disabled: a converged solve turns every kernel into an early exit
        struct TolGuard { double& tol; double keep; ~TolGuard() { tol = keep; } } tol_guard{cfg.pcg_rel_tol, cfg.pcg_rel_tol};
        if (which == 5) cfg.pcg_rel_tol = 0;
        for (int pass = 0; pass < 2; ++pass) {
            const int n = pass == 0 ? 3 : reps;
            HIP_OK(hipEventRecord(ev[0], stream));
            for (int k = 0; k < n; ++k) {
                switch (which) {
                    case 0: if (tl.n_slices > 0) LAUNCH_GM(pr.by_lm.G, k_schur_lm, 0, nbL, stream, tl, zc, lmrec, (const T*)ninv, tvec, st[0], T(0), dl, npart); break;
                    case 1: if (oj()) LAUNCH_GML(pr.by_pose.G, k_schur_pose, 0, 1, nbP, stream, tp, to, zc, tvec, dp, pr.pose_first, pr.pose_last, sbuf, sbuf + (size_t)pr.P * 3, st[0], (const T*)nullptr, (T*)nullptr);
                            else LAUNCH_G(pr.by_pose.G, k_schur_pose, nbP, stream, tp, to, zc, tvec, dp, pr.pose_first, pr.pose_last, sbuf, sbuf + (size_t)pr.P * 3, st[0], (const T*)nullptr, (T*)nullptr);
                            break;
                    case 2: {   // state slot 1 is never written here, slot 0 stays "iters = 0, not done"
                        const T tol2 = (T)0;
                        hipLaunchKernelGGL((k_cg_update<T>), dim3(nbC), dim3(kBlock), 0, stream, pr.P, sbuf, sbuf + (size_t)pr.P * 3, nbP, gpart[0], nbC,
                                           gpart[1], st[0], st[1], minv, r, p, q, x, zc, tol2, 1 << 30, (const T*)one_dev);
                        break;
                    }
                    case 3: if (tl.n_slices > 0) LAUNCH_G(pr.by_lm.G, k_lin_lm, nbL, stream, tl, ps, lmrec, gauge_l, ninv, (T)lambda, py_rules() ? 1 : 0); break;
                    case 4: launch_lin_pose_only(); break;
                    case 6: if (amg_on) { if (int rc = launch_amg_setup()) return rc; } break;
                    default: if (int rc = launch_iteration(0)) return rc; if (int rc = launch_iteration(1)) return rc; break;
                }
            }
            HIP_OK(hipEventRecord(ev[1], stream));
            HIP_OK(hipEventSynchronize(ev[1]));
            if (pass == 1) {
                float ms = 0;
                HIP_OK(hipEventElapsedTime(&ms, ev[0], ev[1]));
                const double per = which == 5 ? 2.0 * n : (double)n;
                *us = 1e3 * ms / per;
            }
        }
        cfg.pcg_rel_tol = tol_guard.keep;
        const double tab[7] = {b_lm, b_pose, b_upd, b_linlm, b_linpose, (amg_on ? 3.0 : 1.0) * (b_lm + b_pose) + b_upd, 0.0};
        *bytes = tab[std::min(std::max(which, 0), 6)];
        // leave a consistent state behind
        return do_linearize(&chi2);
    }
};

}  // namespace

struct tsgo_optimizer {
    tsgo_config cfg;
    IEngine* eng = nullptr;
};

extern "C" {

int tsgo_create(const tsgo_config* cfg, tsgo_optimizer** out) {
    if (!out) return tsgo::set_error(-1, "tsgo_create: null argument");
    tsgo_config c;
    if (cfg) c = *cfg; else tsgo_default_config(&c);
    if (c.precision != 32 && c.precision != 64) return tsgo::set_error(-1, "tsgo_create: precision must be 32 or 64");
    if (c.world < 1) c.world = 1;
    if (c.pcg_rel_tol <= 0) c.pcg_rel_tol = 1e-10;
    if (c.pcg_max_iters <= 0) c.pcg_max_iters = 20000;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
        return tsgo::set_error(-10, "tsgo_create: no HIP device is visible; this library has no CPU fallback");
    if (c.device < 0 || c.device >= n_dev) return tsgo::set_error(-10, "tsgo_create: device ordinal out of range");
    auto* o = new tsgo_optimizer();
    o->cfg = c;
    int rc;
    if (c.precision == 64) { auto* e = new Engine<double>(c); rc = e->init(); o->eng = e; }
    else { auto* e = new Engine<float>(c); rc = e->init(); o->eng = e; }
    if (rc) { delete o->eng; delete o; return rc; }
    *out = o;
    return 0;
}

void tsgo_destroy(tsgo_optimizer* o) {
    if (!o) return;
    if (o->eng && o->eng->comm) (void)ncclCommDestroy(o->eng->comm);
    delete o->eng;
    delete o;
}

int tsgo_set_graph(tsgo_optimizer* o, const tsgo_graph* g) {
    if (!o || !g) return tsgo::set_error(-1, "tsgo_set_graph: null argument");
    return o->eng->set_graph(*g);
}
int tsgo_optimize(tsgo_optimizer* o, int32_t iterations, tsgo_stats* st) {
    if (!o) return tsgo::set_error(-1, "tsgo_optimize: null argument");
    return o->eng->optimize(iterations, st);
}
int tsgo_get_vertices(tsgo_optimizer* o, double* out) {
    if (!o || !out) return tsgo::set_error(-1, "tsgo_get_vertices: null argument");
    return o->eng->get_vertices(out);
}
int tsgo_linearize(tsgo_optimizer* o, double* diag, double* grad, double* chi2) {
    if (!o || !diag || !grad || !chi2) return tsgo::set_error(-1, "tsgo_linearize: null argument");
    return o->eng->linearize(diag, grad, chi2);
}
int tsgo_solve_step(tsgo_optimizer* o, double* delta, double* chi2, int32_t* iters) {
    if (!o || !delta || !chi2) return tsgo::set_error(-1, "tsgo_solve_step: null argument");
    return o->eng->solve_step(delta, chi2, iters);
}
int tsgo_time_kernel(tsgo_optimizer* o, int32_t which, int32_t reps, double* us, double* bytes) {
    if (!o || !us || !bytes || reps <= 0) return tsgo::set_error(-1, "tsgo_time_kernel: bad argument");
    return o->eng->time_kernel(which, reps, us, bytes);
}
int tsgo_cycle_probe(tsgo_optimizer* o, int32_t reps, tsgo_cycle_level* out, int32_t cap) {
    if (!o || !out || reps <= 0 || cap <= 0) return tsgo::set_error(-1, "tsgo_cycle_probe: bad argument");
    return o->eng->cycle_probe(reps, out, cap);
}
int tsgo_profile_iteration(tsgo_optimizer* o, int32_t reps, tsgo_prof_entry* out, int32_t cap) {
    if (!o || !out || reps <= 0 || cap <= 0) return tsgo::set_error(-1, "tsgo_profile_iteration: bad argument");
    return o->eng->profile_iteration(reps, out, cap);
}
#ifdef TSGO_TESTING
int tsgo_local_group_create(int32_t world, tsgo_local_group** out) {
    if (!out || world < 1) return tsgo::set_error(-1, "tsgo_local_group_create: bad argument");
    auto* g = new tsgo_local_group(); g->world = world; g->stage.resize(world);
    *out = g;
    return 0;
}
void tsgo_local_group_destroy(tsgo_local_group* g) { delete g; }
int tsgo_comm_init_local(tsgo_optimizer* o, tsgo_local_group* g) {
    if (!o || !g) return tsgo::set_error(-1, "tsgo_comm_init_local: null argument");
    if (o->cfg.world != g->world || o->cfg.rank < 0 || o->cfg.rank >= g->world) return tsgo::set_error(-1, "tsgo_comm_init_local: the handle's rank / world do not fit the group");
    o->eng->lgroup = g;
    return 0;
}
#endif
void tsgo_reset_history(tsgo_optimizer* o) {
    if (o && o->eng) o->eng->reset_history();
}
int tsgo_comm_selftest(tsgo_optimizer* o, int32_t* ranks_out) {
    if (!o) return tsgo::set_error(-1, "tsgo_comm_selftest: null argument");
    int n = 0;
    const int rc = o->eng->comm_selftest(&n);
    if (ranks_out) *ranks_out = n;
    return rc;
}
int tsgo_comm_unique_id(uint8_t id_out[128]) {
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId size");
    ncclUniqueId id;
    NCCL_OK(ncclGetUniqueId(&id));
    std::memcpy(id_out, &id, 128);
    return 0;
}
int tsgo_comm_init(tsgo_optimizer* o, const uint8_t id_in[128]) {
    if (!o || !id_in) return tsgo::set_error(-1, "tsgo_comm_init: null argument");
    HIP_OK(hipSetDevice(o->cfg.device));
    ncclUniqueId id;
    std::memcpy(&id, id_in, 128);
    NCCL_OK(ncclCommInitRank(&o->eng->comm, o->cfg.world, id, o->cfg.rank));
    return 0;
}

}  // extern "C"
