// tsgo_hip.hip — device side of the C ABI in include/tsgo.h: buffers, launches, the Gauss-Newton
// loop with the reference's stop rules (remote/optimizer/OptimizerCpu.h:80-180), hipGraph replay of
// the PCG iteration, RCCL all-reduces for edge-sharded runs.  Kernels: tsgo_kernels.h, tsgo_amg_kernels.h, tsgo_sym_kernels.h.
//
// One translation unit; the engine class is laid out over this file and engine/*.inc (each included inside the class body):
//   this file                      members, configuration / environment (host/knobs.h), device slabs and upload helpers; the C ABI at the end
//   engine/engine_hierarchy.inc    the multigrid hierarchy on the device: patterns, symbolic products, dense bottom operators
//   engine/engine_graph.inc        tsgo_set_graph: tables, value staging, structure re-use, solver history across requests
//   engine/engine_launch.inc       every kernel launch of a linearisation / a hierarchy build / a PCG iteration
//   engine/engine_collective.inc   the all-reduces of an edge-sharded run
//   engine/engine_solve.inc        the Gauss-Newton loop, the PCG drivers, read-outs
//   engine/engine_probes.inc       timing probes (bench.py)
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
    virtual int comm_time_allreduce(int64_t n, int reps, double* us) = 0;
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
    bool fuse_post_smooth = true;      // the level-0 post-smoothing in the epilogue of the cycle's second product (research: TSGO_FUSE_POST=0 = k_smooth0)
    double hier_shift_cfg = 0;         // what a graph starts with (0; research: TSGO_HIER_SHIFT)
    double hier_shift = 0;             // relative raise of the diagonal of the hierarchy's level-0 matrix (research: TSGO_HIER_SHIFT; do_solve sets it after a breakdown)
    bool fold_gate = true;             // the stopping rule in workgroup 0 of the iteration's first product (research: TSGO_FOLD_GATE=0 = its own kernel)
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
    T *inv_last = nullptr, *r_last = nullptr, *z_last = nullptr, *rzpart = nullptr, *fold_part = nullptr;
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
        return coarse_sweeps != kCoarseSweeps ? (lv[l].n <= kSmallLevelRows ? kSmallLevelSweeps : coarse_sweeps) : sweeps_per_side(l, lv[l].n);
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
        if (const char* e = TSGO_RESEARCH_ENV("TSGO_LPR_XCD")) lpr_xcd = atoi(e) != 0;
        if (const char* e = TSGO_RESEARCH_ENV("TSGO_FOLD_GATE")) fold_gate = atoi(e) != 0;
        if (const char* e = TSGO_RESEARCH_ENV("TSGO_HIER_SHIFT")) hier_shift = hier_shift_cfg = atof(e);
        if (const char* e = TSGO_RESEARCH_ENV("TSGO_FUSE_POST")) fuse_post_smooth = atoi(e) != 0;
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
#include "engine/engine_hierarchy.inc"
#include "engine/engine_graph.inc"
#include "engine/engine_launch.inc"
#include "engine/engine_collective.inc"
#include "engine/engine_solve.inc"
#include "engine/engine_probes.inc"
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
int tsgo_device_count(void) {
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}
int tsgo_comm_time_allreduce(tsgo_optimizer* o, int64_t n_elements, int32_t reps, double* us_per_call) {
    if (!o || !us_per_call || n_elements <= 0 || reps <= 0) return tsgo::set_error(-1, "tsgo_comm_time_allreduce: bad argument");
    return o->eng->comm_time_allreduce(n_elements, reps, us_per_call);
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
