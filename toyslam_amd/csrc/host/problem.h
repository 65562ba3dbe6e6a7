// problem.h — host-side layout of one optimisation problem, built once per request.
//
// The reference keeps an unordered_map of heap vertices and a vector of heap edges
// (remote/graph/GraphCpu.h:55-57) and, on its CUDA path, a byte-packed arena plus AoS edge arrays
// (remote/cuda/graph/GraphGpu.h:80-187).  Here the graph becomes:
//   * dense internal vertex numbers per class (pose p, landmark l), degree-sorted inside windows;
//   * three SELL-64 slot tables (rows of 64 lanes, G lanes cooperating on one vertex):
//       by_pose : LM edges grouped by pose      (Schur pose pass, pose-side linearisation)
//       by_lm   : LM edges grouped by landmark  (Schur landmark pass, landmark-side linearisation)
//       odom    : ODOM edges listed at BOTH endpoints
//     so every device pass is "lane = vertex, loop over its rows" with coalesced plane loads and no
//     atomics.
// Edge sharding (multi-GPU): a shard owns a contiguous landmark range with ALL its LM edges, and a
// contiguous pose range for the ODOM rows and gauge terms.  Pose state is replicated.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "../../../include/tsgo.h"

namespace tsgo {

constexpr uint32_t kNoEdge = 0xFFFFFFFFu;
constexpr uint32_t kDirBit = 0x80000000u;   // odom slot: set when the row's pose is id2 of the edge
constexpr uint32_t kVlmBit = 0x40000000u;   // odom slot: the edge is a virtual landmark measurement (edge type 2), not an ODOM edge
constexpr uint32_t kPoseMask = 0x3FFFFFFFu; // odom slot: the neighbour's internal pose number
constexpr int kWave = 64;
constexpr double kGaugeTerm = 1e6;        // remote/optimizer/OptimizerCpu.h:136
constexpr int kSortWindow = 2048;           // vertices; degree sort happens inside such windows

struct SellTable {
    int G = 1;                       // lanes per vertex
    int n_vertices = 0;              // rows cover internal vertices [0, n_vertices)
    int n_slices = 0;                // ceil(n_vertices / (64 / G))
    std::vector<uint32_t> row_off;   // n_slices + 1, in rows
    size_t rows = 0;
    std::vector<uint32_t> idx;       // rows*64: neighbour's internal number (padding: 0)
    std::vector<uint32_t> edge;      // rows*64: index into tsgo_graph edges, kNoEdge for padding
    int n_planes = 0;
    std::vector<double> planes;      // n_planes * rows*64, plane-major: static per-slot inputs
    size_t slots() const { return rows * kWave; }
    double* plane(int k) { return planes.data() + (size_t)k * slots(); }
    const double* plane(int k) const { return planes.data() + (size_t)k * slots(); }
};

// LM tables: planes zx, zy, w0, w1.  ODOM table: planes mi[0..5], w[0..2]; a virtual-landmark slot (kVlmBit) keeps the local point of
// its OWN pose and of the neighbour in mi[0..3] (pox, poy, pnx, pny) and its two weights in w[0..1].
enum { LM_ZX = 0, LM_ZY, LM_W0, LM_W1, LM_PLANES };
enum { OD_MI0 = 0, OD_W0 = 6, OD_PLANES = 9 };

struct Problem {
    int rank = 0, world = 1;
    int P = 0;                 // poses (all, replicated)
    int L = 0;                 // landmarks owned by this shard
    int L_total = 0;
    int64_t n_lm_edges = 0;    // owned
    int64_t n_lm_edges_total = 0, n_odom_edges_total = 0;
    int lm_first = 0, lm_last = 0;      // owned range in landmark order of the input graph
    int pose_first = 0, pose_last = 0;  // owned internal pose range (ODOM rows + gauge)
    std::vector<int> pose_vertex;       // internal pose -> position in tsgo_graph vertex arrays
    std::vector<int> lm_vertex;         // internal (local) landmark -> position
    std::vector<double> pose_xyt;       // 3 per pose
    std::vector<double> lm_xy;          // 2 per owned landmark
    std::vector<double> gauge_p, gauge_l;   // 1e6 * multiplicity in the fixed list: every pose (applied by the shard that owns it), owned landmarks
    SellTable by_pose, by_lm, odom;
    int n_vertices = 0;                 // of the input graph
    bool has_vlm = false;               // the graph holds virtual landmark measurements (edge type 2): pose-pose slots in general form (tsgo_math.h)
    int64_t n_vlm_edges_total = 0;
    bool odom_analytic = false;         // analytic ODOM Jacobians (tsgo_config.odom_jacobian): odometry then couples heading to translation,
                                        // so every pose an edge touches carries a lever arm in the multigrid coarse space (host/amg.cpp)
};

struct BuildOptions {
    int rank = 0, world = 1;
    int lanes_per_pose = 0, lanes_per_lm = 0;   // 0 = auto
    bool fill_planes = true;    // false: leave SellTable::planes empty (the device engine stages them itself, in its own
                                // scalar type and plane order, straight into pinned memory: tsgo_hip.hip stage_static_*)
};

// Returns empty string on success, otherwise the error text.
std::string build_problem(const tsgo_graph& g, const BuildOptions& opt, Problem& out);

// general 3x3 inverse in double; false if singular
bool invert3(const double* m, double* out);

// The four static per-slot values of an LM edge (zx, zy, w0, w1) from its measurement (range, bearing) and information
// diagonal — ONE definition, shared by build_problem and by the engine's staging/refill path, so that a refilled table
// holds the same bits as a freshly built one.
void lm_static(const double* meas, const double* inf, double* out4);
// ... and the nine static plane values of a virtual-landmark slot at the edge's first (side = 0) or second endpoint:
// meas = (r1, phi1, r2, phi2), inf = (w0, w1, .)
void vlm_static(const double* meas, const double* inf, int side, double* out9);

}  // namespace tsgo
