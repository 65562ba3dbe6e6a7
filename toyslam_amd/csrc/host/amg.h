// amg.h — symbolic (pattern) side of the smoothed-aggregation multigrid preconditioner for the
// reduced pose system S (built once per request on the host; all numeric work is on the device).
//
// Why: the reference solves H delta = b with a dense QR (remote/solver/SolverEigen.h:20).  The sparse
// replacement is PCG on S = Hpp - W Dl^-1 W^T; with a block-Jacobi preconditioner a 100k-pose graph
// needs ~6 000 iterations per Gauss-Newton step (profiles/r01a_*).  The slow modes of a landmark-SLAM
// pose system are rigid motions of groups of poses that see the same landmarks (consecutive poses, and
// revisits of the same place).  Aggregates are grown by heavy-edge matching on the co-observation graph
// (weight = shared landmarks + kAggOdomWeight per odometry edge; 4, 4, 8, 8 ... nodes per aggregate); the
// coarse space of an aggregate is its three rigid modes (tx, ty, rotation about its centroid), the
// prolongator is Jacobi-smoothed.  A V-cycle over that hierarchy brings PCG to ~17 iterations.
//
// Level l holds A_l (block CSR, 3x3 blocks; level 0 = explicit Schur complement), aggregates, the
// prolongator P_l = (I - w D^-1 A_l) Z_l and the Galerkin product A_{l+1} = P_l^T (A_l P_l).  Every
// numeric product is a gather over precomputed (x, y) block-pair lists: no atomics, fixed order.
#pragma once
#include <cstdint>
#include <functional>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "problem.h"

namespace tsgo {

constexpr int kAggSize = 4;          // poses per aggregate on level 0
constexpr int kAggSizesBelow[] = {4, 8};      // nodes per aggregate on levels 1, 2+ (the last repeats); powers of two
constexpr float kAggOdomWeight = 16.f;        // coupling weight of an odometry edge, in shared landmarks (see aggregate_by_matching)
constexpr int kSmoothLevels = 99;    // levels whose prolongator is Jacobi-smoothed (the rest use the tentative one)
constexpr int kCoarsestMax = 28;     // stop coarsening at <= this many block rows (dense inverse in LDS, <= 84 x 84)
constexpr int kMaxPairDegree = 64;    // landmarks observed from more poses than this do not add off-diagonal level-0 blocks
constexpr double kProlongOmega = 0.7;
constexpr double kSmoother0Omega = 1.0; // smoother damping before the first estimate (level 0)
constexpr double kSmootherOmega = 0.8;  // ... and on the coarse levels.  The working values are ESTIMATED per level from a power
                                        // iteration on D^-1 A (omega = min(1, 1.6 / (1.05 rho))): smoothed Galerkin matrices reach
                                        // rho = 2 ... 17, and omega * rho >= 2 makes the cycle indefinite (seen with fixed 0.8 / 1.0)
constexpr int kCoarseSweeps = 2;      // block-Jacobi sweeps per side on the coarse levels: V(1,1) on level 0, V(2,2) below ...
constexpr int kSmallLevelRows = 2048; // ... except on levels this small, which get
constexpr int kSmallLevelSweeps = 1;  // one sweep per side: their kernels are pure launch latency (5 us each) and the second
                                      // sweep buys no iterations there (100k poses, device: 19.6 -> 19.8 PCG iterations, 276 -> 265 us)
constexpr int kBigLevel1Rows = 8192;  // ... and LEVEL 1 of a graph this large gets one sweep per side as well: it is the most expensive coarse
                                      // level (two launches of 8 us at 100k poses, 50 us at 1M) and its second sweep buys 0.45 of 15 / 0.5 of 20
                                      // PCG iterations (round 4, profiles/r04m_sweeps_per_level.txt: 3.55 -> 3.48 ms, 32.2 -> 30.8 ms per step;
                                      // the same cut on level 2 costs 3.3 iterations; at 10k poses level 1 is small and the cut does not pay)
// block-Jacobi sweeps per side of the cycle on coarse level l >= 1 with n_rows block rows — ONE rule for the device engine and the CPU twin
inline int sweeps_per_side(size_t l, int n_rows) {
    if (n_rows <= kSmallLevelRows) return kSmallLevelSweeps;
    if (l == 1 && n_rows >= kBigLevel1Rows) return 1;
    if (l >= 3) return 1;       // a third coarse level above 2 048 rows exists from ~260k poses on: at a million poses its second sweep buys nothing (20.5 iterations
                                // either way, 30.4 -> 30.0 ms per step); level 2 is the one that needs two (one there: 18.3 instead of 15.4 iterations at 100k poses)
    return kCoarseSweeps;
}

struct BlockCsr {
    int n_rows = 0, n_cols = 0;
    std::vector<int> ptr, col;
    int nnz() const { return (int)col.size(); }
};

// std::vector<int> whose resize() leaves new elements uninitialised: the pair lists hold 43 M entries at 100k poses and
// every one of them is written by the fill pass; zero-filling them first costs more than filling them.
template <typename V> struct DefaultInit {
    using value_type = V;
    DefaultInit() = default;
    template <typename U> DefaultInit(const DefaultInit<U>&) {}
    V* allocate(size_t n) { return std::allocator<V>().allocate(n); }
    void deallocate(V* p, size_t n) { std::allocator<V>().deallocate(p, n); }
    template <typename U> void construct(U* p) { ::new ((void*)p) U; }
    template <typename U, typename A0, typename... A> void construct(U* p, A0&& a0, A&&... a) { ::new ((void*)p) U(std::forward<A0>(a0), std::forward<A>(a)...); }
    template <typename U> bool operator==(const DefaultInit<U>&) const { return true; }
    template <typename U> bool operator!=(const DefaultInit<U>&) const { return false; }
};
using ivec = std::vector<int, DefaultInit<int>>;

struct PairList {                    // output block o sums over pairs [ptr[o], ptr[o+1])
    ivec ptr, x, y;
};

struct AmgLevel {
    int n = 0, n_agg = 0;
    BlockCsr A;                      // pattern of A_l
    std::vector<int> diag;           // per row: index of its diagonal block
    std::vector<int> agg;            // node -> aggregate
    std::vector<double> rel;         // 2 per node: position relative to the aggregate centroid
    std::vector<uint8_t> rig;        // per node: rotation couples to translation here (a pose with landmark observations below it)
    BlockCsr P;                      // n x n_agg
    std::vector<int> p_self;         // per P block: 1 when col == agg(row)
    PairList p_src;                  // per P block (i,a): x = A block (i,k) with agg(k) == a, y = k
    BlockCsr R;                      // n_agg x n (pattern of P^T)
    std::vector<int> r_to_p;         // per R block: the P block it transposes
    BlockCsr T;                      // n x n_agg: pattern of A_l P_l
    PairList t_src;                  // per T block: x = A block, y = P block
    PairList a_src;                  // per A_{l+1} block on or above the diagonal: x = P block (transposed), y = T block
    std::vector<int> a_mirror;       // per A_{l+1} block: below the diagonal, the block it is the transpose of; else -1
};

struct SchurLists {                  // level 0: how each off-diagonal S block is summed
    std::vector<int> ptr;            // per S block
    std::vector<uint32_t> slot_i, slot_k;   // by_pose slots of the two LM edges sharing a landmark
    std::vector<int> od_ptr;         // per S block
    std::vector<uint32_t> od_slot;   // odom-table slots (row i side) joining i and k
};

struct AmgSym {
    std::vector<AmgLevel> levels;    // levels[l] coarsens A_l to A_{l+1}
    BlockCsr A_last;                 // pattern of the coarsest matrix
    std::vector<int> diag_last;
    SchurLists schur;
    std::vector<int> order;          // internal pose -> position along the trajectory (visiting order and numbering of the aggregates)
};

// Progress of build_amg, for a caller that wants to consume (upload) finished parts while the rest is being built:
// schur_ready() once out.schur and out.order are final; level_ready(n) when out.levels[0 .. n-1] are final (the vector
// never reallocates); both are called on the building thread.
// Level 0's inputs as three CSR lists (what a device builder of the Schur pattern reads): a pose's LM edges (landmark, by_pose slot),
// a landmark's observers (pose, by_pose slot of that edge), a pose's odometry slots (other pose, odom-table slot).
struct SchurCsr {
    int P = 0, L = 0, max_pair_degree = 0;
    std::vector<int> pp_ptr, pp_lm; std::vector<uint32_t> pp_slot;
    std::vector<int> obs_ptr, obs_pose; std::vector<uint32_t> obs_slot;
    std::vector<int> od_ptr, od_col; std::vector<uint32_t> od_slot;
};

struct AmgProgress {
    // Optional: level 0's pattern and contribution lists built by somebody else (the engine: tsgo_sym_kernels.h, k_s0_*).  True:
    // A0 holds the pattern of S, sc_ptr / sc_od_ptr the list offsets per block (their differences are the coupling weights the
    // aggregation needs); the lists themselves stay where they were built (SchurLists::slot_i, slot_k, od_slot remain empty).
    std::function<bool(const SchurCsr& in, BlockCsr& A0, std::vector<int>& sc_ptr, std::vector<int>& sc_od_ptr, std::string& err)> schur;
    std::function<void()> schur_ready;
    std::function<void(int)> level_ready;
    // Optional: the two pair-list products of a level (T = A P and A' = P^T T) built by somebody else — the engine builds them on
    // the device (tsgo_sym_kernels.h).  Called on the building thread with L.A, L.P, L.R and L.r_to_p final.  True: A_next holds
    // the pattern of the next matrix, and L.T, L.t_src, L.a_src, L.a_mirror stay EMPTY on the host (the lists live where they were
    // built).  False: declined (e.g. a row too dense for the device tables) — the host builds them itself.  err: a failure.
    std::function<bool(int level, AmgLevel& L, BlockCsr& A_next, std::string& err)> products;
};

// Builds the hierarchy from the layout of a WHOLE graph (world = 1) INTO `out` (cleared first).  Returns "" or an error text.
std::string build_amg(const Problem& pr, AmgSym& out, const AmgProgress* progress = nullptr);

// Edge-sharded runs (Problem::world > 1): the hierarchy is REPLICATED — every rank builds the same patterns from the
// whole graph — but the level-0 matrix is summed from per-rank partial blocks: `out.schur` lists, for every S block,
// only the landmark pairs / odometry rows THIS shard owns, as slots of `local`'s own tables; the ranks' partial
// blocks are then all-reduced (tsgo_hip.hip: launch_amg_setup).  The diagonal blocks come from the all-reduced
// linearisation partials and are contributed by rank 0 alone.
std::string build_amg_sharded(const tsgo_graph& g, const Problem& local, AmgSym& out);

// A request with the same structure but new pose estimates reuses every pattern; only the rigid-mode lever arms
// (AmgLevel::rel: node positions relative to their aggregate's centroid) depend on the estimates.  Recomputes them on
// every level from pose_xyt (3 per pose, internal numbering), exactly as build_amg does.
void refresh_amg_geometry(const std::vector<double>& pose_xyt, AmgSym& amg);

}  // namespace tsgo
