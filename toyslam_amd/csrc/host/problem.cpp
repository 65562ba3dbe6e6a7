// problem.cpp — builds the device-ready layout (see problem.h) from a tsgo_graph.
#include "problem.h"
#include "knobs.h"
#include "parallel.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <climits>
#include <cstdint>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <unordered_map>

namespace tsgo {

void vlm_static(const double* meas, const double* inf, int side, double* out9) {
    const double p1x = meas[0] * std::cos(meas[1]), p1y = meas[0] * std::sin(meas[1]), p2x = meas[2] * std::cos(meas[3]), p2y = meas[2] * std::sin(meas[3]);
    out9[0] = side ? p2x : p1x; out9[1] = side ? p2y : p1y; out9[2] = side ? p1x : p2x; out9[3] = side ? p1y : p2y;
    out9[4] = out9[5] = 0; out9[6] = inf[0]; out9[7] = inf[1]; out9[8] = 0;
}

bool invert3(const double* m, double* out) {
    double a[3][6];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) { a[i][j] = m[i * 3 + j]; a[i][3 + j] = (i == j) ? 1.0 : 0.0; }
    for (int col = 0; col < 3; ++col) {
        int piv = col;
        for (int r = col + 1; r < 3; ++r) if (std::fabs(a[r][col]) > std::fabs(a[piv][col])) piv = r;
        if (!(std::fabs(a[piv][col]) > 0.0)) return false;
        if (piv != col) for (int j = 0; j < 6; ++j) std::swap(a[piv][j], a[col][j]);
        const double d = a[col][col];
        for (int j = 0; j < 6; ++j) a[col][j] /= d;
        for (int r = 0; r < 3; ++r) if (r != col) {
            const double f = a[r][col];
            for (int j = 0; j < 6; ++j) a[r][j] -= f * a[col][j];
        }
    }
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) out[i * 3 + j] = a[i][3 + j];
    return true;
}

void lm_static(const double* meas, const double* inf, double* out4) {
    out4[LM_ZX] = meas[0] * std::cos(meas[1]); out4[LM_ZY] = meas[0] * std::sin(meas[1]);   // EdgeSe2Point2d.h:34-35
    out4[LM_W0] = inf[0]; out4[LM_W1] = inf[1];
}

namespace {

// Lanes per vertex, from the sweep in profiles/r01c (100k poses, MI355X): the landmark passes want ~1-2
// rows per slice (k_schur_lm 24.1 -> 16.8 us going from 1 to 4 lanes at degree 5); the pose passes carry
// a longer per-vertex epilogue and the ODOM rows, and are best at 2 lanes for degree 10-12.
int auto_lanes_pose(double mean_degree) {
    if (mean_degree >= 40) return 8;
    if (mean_degree >= 20) return 4;
    if (mean_degree >= 6) return 2;
    return 1;
}
int auto_lanes_lm(double mean_degree) {
    if (mean_degree >= 16) return 8;
    if (mean_degree >= 3) return 4;
    if (mean_degree >= 1.5) return 2;
    return 1;
}

bool valid_lanes(int g) { return g == 1 || g == 2 || g == 4 || g == 8; }

// order[i] = class index placed at internal position i; degree-descending inside windows.
std::vector<int> window_sort(const std::vector<int>& degree, int window) {
    std::vector<int> order(degree.size());
    std::iota(order.begin(), order.end(), 0);
    if (window <= 1) return order;
    const int n_win = (int)((order.size() + (size_t)window - 1) / (size_t)window);
    parallel_chunks(n_win, [&](int, int wb, int we) {           // windows are independent
        for (int w = wb; w < we; ++w) {
            const size_t b = (size_t)w * (size_t)window, e = std::min(order.size(), b + (size_t)window);
            std::stable_sort(order.begin() + b, order.begin() + e, [&](int x, int y) { return degree[x] > degree[y]; });
        }
    }, 4);
    return order;
}

// Shapes a SELL table for per-vertex degrees (internal numbering) and returns, through `place`,
// a function-like table: slot_of(v, k) = position of the k-th entry of vertex v.
void shape_table(SellTable& t, int G, const std::vector<int>& degree, int n_planes, bool with_planes) {
    t.G = G; t.n_vertices = (int)degree.size(); t.n_planes = n_planes;
    const int vps = kWave / G;
    t.n_slices = (t.n_vertices + vps - 1) / vps;
    t.row_off.assign(t.n_slices + 1, 0);
    for (int s = 0; s < t.n_slices; ++s) {
        int w = 0;
        for (int v = s * vps; v < std::min(t.n_vertices, (s + 1) * vps); ++v) w = std::max(w, (degree[v] + G - 1) / G);
        t.row_off[s + 1] = t.row_off[s] + (uint32_t)w;
    }
    t.rows = t.row_off[t.n_slices];
    t.idx.assign(t.slots(), 0u);
    t.edge.assign(t.slots(), kNoEdge);
    if (with_planes) t.planes.assign((size_t)n_planes * t.slots(), 0.0); else t.planes.clear();
}

inline size_t slot_of(const SellTable& t, int v, int k) {
    const int vps = kWave / t.G;
    const int s = v / vps, lane = (v % vps) * t.G + (k % t.G);
    return ((size_t)t.row_off[s] + (size_t)(k / t.G)) * kWave + (size_t)lane;
}

}  // namespace

std::string build_problem(const tsgo_graph& g, const BuildOptions& opt, Problem& out) {
    static const bool timing = getenv("TSGO_LAYOUT_TIMING") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!timing) return;
        const auto n = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[layout] %-28s %7.1f ms\n", what, std::chrono::duration<double, std::milli>(n - t_last).count());
        t_last = n;
    };
    Problem pr;
    pr.rank = opt.rank; pr.world = std::max(1, opt.world);
    if (pr.rank < 0 || pr.rank >= pr.world) return "rank outside [0, world)";
    if (g.n_vertices < 0 || g.n_edges < 0 || g.n_fixed < 0) return "negative count";
    const int nV = g.n_vertices, nE = g.n_edges;
    pr.n_vertices = nV;

    // ---- classify vertices -----------------------------------------------------------------------
    std::vector<int> cls(nV);                 // class index within its type
    std::vector<int> pose_pos, lm_pos;        // class index -> vertex position
    // id -> vertex position: a flat table when the ids are (nearly) dense, as the reference's client sends them
    // (python/slam_main.py:161-185 numbers vertices 0..V-1); a hash map for anything else
    uint32_t max_id = 0;
    for (int i = 0; i < nV; ++i) max_id = std::max(max_id, g.v_id[i]);
    const bool flat = nV > 0 && (uint64_t)max_id < 4ull * (uint64_t)nV + 1024;
    std::vector<int> table(flat ? (size_t)max_id + 1 : 0, -1);
    std::unordered_map<uint32_t, int> by_id;
    if (!flat) by_id.reserve((size_t)nV * 2);
    auto lookup = [&](uint32_t id) -> int {
        if (flat) return id <= max_id ? table[id] : -1;
        auto it = by_id.find(id);
        return it == by_id.end() ? -1 : it->second;
    };
    for (int i = 0; i < nV; ++i) {
        const uint32_t t = g.v_type[i];
        if (t == 0) { cls[i] = (int)pose_pos.size(); pose_pos.push_back(i); }
        else if (t == 1) { cls[i] = (int)lm_pos.size(); lm_pos.push_back(i); }
        else return "unknown vertex type " + std::to_string(t);
        if (flat) { if (table[g.v_id[i]] >= 0) return "duplicate vertex id " + std::to_string(g.v_id[i]); table[g.v_id[i]] = i; }
        else if (!by_id.emplace(g.v_id[i], i).second) return "duplicate vertex id " + std::to_string(g.v_id[i]);
    }
    const int P = (int)pose_pos.size(), Lt = (int)lm_pos.size();
    pr.P = P; pr.L_total = Lt;

    lap("classify vertices");
    // ---- validate edges, degrees -------------------------------------------------------------------
    std::vector<int> ev1(nE), ev2(nE);        // vertex positions of the endpoints
    std::vector<int> deg_pose_lm(P, 0), deg_lm(Lt, 0), deg_pose_od(P, 0);
    for (int e = 0; e < nE; ++e) {
        const int a = lookup(g.e_ids[2 * (size_t)e]), b = lookup(g.e_ids[2 * (size_t)e + 1]);
        if (a < 0 || b < 0)
            return "edge " + std::to_string(e) + " refers to an unknown vertex id";
        ev1[e] = a; ev2[e] = b;
        const uint32_t t = g.e_type[e];
        if (t == 0) {
            if (g.v_type[ev1[e]] != 0 || g.v_type[ev2[e]] != 0) return "ODOM edge " + std::to_string(e) + " must join two Se2 vertices";
            ++deg_pose_od[cls[ev1[e]]]; ++deg_pose_od[cls[ev2[e]]];
            ++pr.n_odom_edges_total;
        } else if (t == 1) {
            if (g.v_type[ev1[e]] != 0 || g.v_type[ev2[e]] != 1) return "LM edge " + std::to_string(e) + " must join an Se2 vertex to a Point2 vertex";
            ++deg_pose_lm[cls[ev1[e]]]; ++deg_lm[cls[ev2[e]]];
            ++pr.n_lm_edges_total;
        } else if (t == 2) {      // virtual landmark measurement: a pose-pose edge in the ODOM table (tsgo_math.h: vlm_linearize)
            if (g.v_type[ev1[e]] != 0 || g.v_type[ev2[e]] != 0) return "virtual-landmark edge " + std::to_string(e) + " must join two Se2 vertices";
            ++deg_pose_od[cls[ev1[e]]]; ++deg_pose_od[cls[ev2[e]]];
            ++pr.n_vlm_edges_total; pr.has_vlm = true;
        } else return "unknown edge type " + std::to_string(t);
    }

    lap("validate edges, degrees");
    // ---- shard: contiguous landmark range balanced by LM-edge count; contiguous pose range --------
    {
        const int64_t total = pr.n_lm_edges_total;
        auto bound = [&](int r) -> int {         // first landmark class index of shard r
            if (r <= 0) return 0;
            if (r >= pr.world) return Lt;
            const int64_t target = (total * r) / pr.world;
            int64_t acc = 0; int j = 0;
            while (j < Lt && acc + deg_lm[j] <= target) { acc += deg_lm[j]; ++j; }
            // with no LM edges at all fall back to an even split of the landmarks
            if (total == 0) j = (int)(((int64_t)Lt * r) / pr.world);
            return j;
        };
        pr.lm_first = bound(pr.rank); pr.lm_last = bound(pr.rank + 1);
        pr.pose_first = (int)(((int64_t)P * pr.rank) / pr.world);
        pr.pose_last = (int)(((int64_t)P * (pr.rank + 1)) / pr.world);
    }
    const int L = pr.lm_last - pr.lm_first;
    pr.L = L;

    // ---- internal numbering --------------------------------------------------------------------------
    // poses: global (identical on every shard) -> sort by the FULL graph's LM degree
    static const int win_pose = TSGO_RESEARCH_INT("TSGO_SORT_WINDOW_POSE", kSortWindow);     // research
    static const int win_lm = TSGO_RESEARCH_INT("TSGO_SORT_WINDOW_LM", kSortWindow);
    const std::vector<int> pose_order = window_sort(deg_pose_lm, win_pose);       // internal -> class
    std::vector<int> pose_internal(P);
    for (int i = 0; i < P; ++i) pose_internal[pose_order[i]] = i;
    std::vector<int> deg_lm_local(deg_lm.begin() + pr.lm_first, deg_lm.begin() + pr.lm_last);
    const std::vector<int> lm_order = window_sort(deg_lm_local, win_lm);        // internal(local) -> class - lm_first
    std::vector<int> lm_internal(L);
    for (int i = 0; i < L; ++i) lm_internal[lm_order[i]] = i;

    pr.pose_vertex.resize(P); pr.pose_xyt.resize((size_t)P * 3); pr.gauge_p.assign(P, 0.0);
    for (int i = 0; i < P; ++i) {
        const int v = pose_pos[pose_order[i]];
        pr.pose_vertex[i] = v;
        for (int k = 0; k < 3; ++k) pr.pose_xyt[(size_t)i * 3 + k] = g.v_pos[(size_t)v * 3 + k];
    }
    pr.lm_vertex.resize(L); pr.lm_xy.resize((size_t)L * 2); pr.gauge_l.assign(L, 0.0);
    for (int i = 0; i < L; ++i) {
        const int v = lm_pos[pr.lm_first + lm_order[i]];
        pr.lm_vertex[i] = v;
        pr.lm_xy[(size_t)i * 2] = g.v_pos[(size_t)v * 3]; pr.lm_xy[(size_t)i * 2 + 1] = g.v_pos[(size_t)v * 3 + 1];
    }
    for (int i = 0; i < g.n_fixed; ++i) {
        const int v = lookup(g.fixed[i]);
        if (v < 0) return "fixed vertex id " + std::to_string(g.fixed[i]) + " is unknown";
        if (g.v_type[v] == 0) {
            const int p = pose_internal[cls[v]];
            pr.gauge_p[p] += kGaugeTerm;      // known on every shard (the Python rules zero a fixed pose's gradient everywhere); APPLIED by the owner only
        } else {
            const int c = cls[v];
            if (c >= pr.lm_first && c < pr.lm_last) pr.gauge_l[lm_internal[c - pr.lm_first]] += kGaugeTerm;
        }
    }

    lap("numbering, windows, state");
    // ---- per-vertex degrees in internal numbering (owned edges only) ---------------------------------
    // The k-th edge of a vertex IN INPUT ORDER takes the vertex's k-th slot (the order fixes every summation order on the
    // device).  Edge chunks are counted side by side: a chunk's count per vertex, an exclusive prefix over the chunks, and
    // the fill pass of a chunk starts every vertex at its prefix — the same slots as one walk over all edges.
    const int nt = std::max(1, std::min(host_threads(), nE / 65536));
    auto chunk_begin = [&](int c) { return (int)((int64_t)nE * c / nt); };
    std::vector<std::vector<int>> cP(nt), cL(nt), cO(nt);
    std::vector<int64_t> n_lm_chunk(nt, 0);
    parallel_chunks(nt, [&](int, int cb, int ce) {
        for (int c = cb; c < ce; ++c) {
            std::vector<int>& kp = cP[c]; std::vector<int>& kl = cL[c]; std::vector<int>& ko = cO[c];
            kp.assign(P, 0); kl.assign(L, 0); ko.assign(P, 0);
            for (int e = chunk_begin(c); e < chunk_begin(c + 1); ++e) {
                if (g.e_type[e] == 1) {
                    const int cl = cls[ev2[e]];
                    if (cl < pr.lm_first || cl >= pr.lm_last) continue;
                    ++kp[pose_internal[cls[ev1[e]]]]; ++kl[lm_internal[cl - pr.lm_first]]; ++n_lm_chunk[c];
                } else {
                    const int p1 = pose_internal[cls[ev1[e]]], p2 = pose_internal[cls[ev2[e]]];
                    if (p1 >= pr.pose_first && p1 < pr.pose_last) ++ko[p1];
                    if (p2 >= pr.pose_first && p2 < pr.pose_last) ++ko[p2];
                }
            }
        }
    }, 1);
    for (int c = 0; c < nt; ++c) pr.n_lm_edges += n_lm_chunk[c];
    std::vector<int> dP(P, 0), dL(L, 0), dO(P, 0);
    auto prefix = [&](std::vector<std::vector<int>>& cnt, std::vector<int>& total) {      // cnt[c][v] := edges of v before chunk c
        parallel_chunks((int)total.size(), [&](int, int b, int e) {
            for (int v = b; v < e; ++v) { int acc = 0; for (int c = 0; c < nt; ++c) { const int k = cnt[c][v]; cnt[c][v] = acc; acc += k; } total[v] = acc; }
        }, 4096);
    };
    prefix(cP, dP); prefix(cL, dL); prefix(cO, dO);
    int Gp = opt.lanes_per_pose, Gl = opt.lanes_per_lm;
    // From quantities every shard knows alike: the ranks all-reduce buffers whose length depends on the pose table's shape
    // ([3P | one partial per workgroup]), so they must choose the same lanes per pose — a shard's OWN mean degree can fall on
    // the other side of a threshold than its neighbour's (found by tests/research/soak_sharded_gpu.py: 2 shards, 6.0 edges per pose).
    if (Gp == 0) Gp = auto_lanes_pose(P ? (double)pr.n_lm_edges_total / pr.world / P : 0.0);
    if (Gl == 0) Gl = auto_lanes_lm(Lt ? (double)pr.n_lm_edges_total / Lt : 0.0);
    if (!valid_lanes(Gp) || !valid_lanes(Gl)) return "lanes per vertex must be 1, 2, 4 or 8";

    shape_table(pr.by_pose, Gp, dP, LM_PLANES, opt.fill_planes);
    shape_table(pr.by_lm, Gl, dL, LM_PLANES, opt.fill_planes);
    shape_table(pr.odom, Gp, dO, OD_PLANES, opt.fill_planes);

    lap("owned degrees, table shapes");
    // ---- fill ----------------------------------------------------------------------------------------
    std::atomic<int> bad_edge{INT32_MAX};
    parallel_chunks(nt, [&](int, int cb, int ce) {
        for (int c = cb; c < ce; ++c) {
            std::vector<int>& fillP = cP[c]; std::vector<int>& fillL = cL[c]; std::vector<int>& fillO = cO[c];
            for (int e = chunk_begin(c); e < chunk_begin(c + 1); ++e) {
                const double* m = g.e_meas + (size_t)e * 9;
                const double* w = g.e_inf + (size_t)e * 3;
                if (g.e_type[e] == 1) {
                    const int cl = cls[ev2[e]];
                    if (cl < pr.lm_first || cl >= pr.lm_last) continue;
                    const int p = pose_internal[cls[ev1[e]]], l = lm_internal[cl - pr.lm_first];
                    const size_t sp = slot_of(pr.by_pose, p, fillP[p]++), sl = slot_of(pr.by_lm, l, fillL[l]++);
                    pr.by_pose.idx[sp] = (uint32_t)l; pr.by_pose.edge[sp] = (uint32_t)e;
                    pr.by_lm.idx[sl] = (uint32_t)p; pr.by_lm.edge[sl] = (uint32_t)e;
                    if (opt.fill_planes) {
                        double vals[LM_PLANES];
                        lm_static(m, w, vals);
                        for (int k = 0; k < LM_PLANES; ++k) { pr.by_pose.plane(k)[sp] = vals[k]; pr.by_lm.plane(k)[sl] = vals[k]; }
                    }
                } else if (g.e_type[e] == 2) {
                    const int p1 = pose_internal[cls[ev1[e]]], p2 = pose_internal[cls[ev2[e]]];
                    for (int side = 0; side < 2; ++side) {
                        const int self = side ? p2 : p1, other = side ? p1 : p2;
                        if (self < pr.pose_first || self >= pr.pose_last) continue;
                        const size_t so = slot_of(pr.odom, self, fillO[self]++);
                        pr.odom.idx[so] = (uint32_t)other | (side ? kDirBit : 0u) | kVlmBit;
                        pr.odom.edge[so] = (uint32_t)e;
                        if (opt.fill_planes) {
                            double v9[9]; vlm_static(m, w, side, v9);
                            for (int k = 0; k < 9; ++k) pr.odom.plane(k)[so] = v9[k];
                        }
                    }
                } else {
                    double inv[9];
                    if (!invert3(m, inv)) { int seen = bad_edge.load(); while (e < seen && !bad_edge.compare_exchange_weak(seen, e)) {} continue; }
                    const int p1 = pose_internal[cls[ev1[e]]], p2 = pose_internal[cls[ev2[e]]];
                    for (int side = 0; side < 2; ++side) {
                        const int self = side ? p2 : p1, other = side ? p1 : p2;
                        if (self < pr.pose_first || self >= pr.pose_last) continue;
                        const size_t so = slot_of(pr.odom, self, fillO[self]++);
                        pr.odom.idx[so] = (uint32_t)other | (side ? kDirBit : 0u);
                        pr.odom.edge[so] = (uint32_t)e;
                        if (opt.fill_planes) {
                            for (int k = 0; k < 6; ++k) pr.odom.plane(OD_MI0 + k)[so] = inv[k];
                            for (int k = 0; k < 3; ++k) pr.odom.plane(OD_W0 + k)[so] = w[k];
                        }
                    }
                }
            }
        }
    }, 1);
    if (bad_edge.load() != INT32_MAX) return "ODOM edge " + std::to_string(bad_edge.load()) + " has a singular measurement matrix";
    lap("fill slots");
    out = std::move(pr);
    lap("hand over (frees the old)");
    return std::string();
}

}  // namespace tsgo
