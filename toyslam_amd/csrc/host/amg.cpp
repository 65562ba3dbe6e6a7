// amg.cpp — host-side symbolic setup of the smoothed-aggregation hierarchy (see amg.h).
#include "amg.h"
#include "knobs.h"
#include "parallel.h"

#include <sched.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <climits>
#include <cstdint>
#include <numeric>
#include <string>
#include <thread>

namespace tsgo {
namespace {

template <typename F> void for_slots(const SellTable& tb, int v, F f) {
    const int vps = kWave / tb.G, sl = v / vps, base = (v % vps) * tb.G;
    for (uint32_t row = tb.row_off[sl]; row < tb.row_off[sl + 1]; ++row)
        for (int sub = 0; sub < tb.G; ++sub) {
            const size_t slot = (size_t)row * kWave + base + sub;
            if (tb.edge[slot] != kNoEdge) f(slot);
        }
}

struct Stopwatch {
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    bool on = getenv("TSGO_AMG_TIMING") != nullptr;
    void lap(const char* what) {
        auto n = std::chrono::steady_clock::now();
        if (on) std::fprintf(stderr, "[amg symbolic] %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(n - t).count());
        t = n;
    }
};

struct Triple { int c, x, y; };
inline bool triple_less(const Triple& p, const Triple& q) { return p.c != q.c ? p.c < q.c : (p.x != q.x ? p.x < q.x : p.y < q.y); }

// Row-wise grouped output of a symbolic product, built per chunk and concatenated in row order.
struct RowsOut {
    std::vector<int> row_nnz, col, grp_len, x, y, flag;
};
// The chunks' outputs land at offsets known from their sizes alone, so they are copied side by side (the serial version of
// this spent more time than the chunks took to produce: 43 M list entries at 100k poses).
void concat(std::vector<RowsOut>& parts, int used, BlockCsr& Z, PairList& pl, std::vector<int>* flag) {
    std::vector<size_t> r0(used + 1, 0), c0(used + 1, 0), p0(used + 1, 0);
    for (int c = 0; c < used; ++c) { r0[c + 1] = r0[c] + parts[c].row_nnz.size(); c0[c + 1] = c0[c] + parts[c].col.size(); p0[c + 1] = p0[c] + parts[c].x.size(); }
    Z.ptr.resize(r0[used] + 1); Z.col.resize(c0[used]); pl.ptr.resize(c0[used] + 1); pl.x.resize(p0[used]); pl.y.resize(p0[used]);
    if (flag) flag->resize(c0[used]);
    Z.ptr[0] = 0; pl.ptr[0] = 0;
    parallel_chunks(used, [&](int, int cb, int ce) {
        for (int c = cb; c < ce; ++c) {
            RowsOut& p = parts[c];
            int at = (int)c0[c];
            for (size_t k = 0; k < p.row_nnz.size(); ++k) Z.ptr[r0[c] + k + 1] = (at += p.row_nnz[k]);
            std::copy(p.col.begin(), p.col.end(), Z.col.begin() + c0[c]);
            int pa = (int)p0[c];
            for (size_t k = 0; k < p.grp_len.size(); ++k) pl.ptr[c0[c] + k + 1] = (pa += p.grp_len[k]);
            std::copy(p.x.begin(), p.x.end(), pl.x.begin() + p0[c]); std::copy(p.y.begin(), p.y.end(), pl.y.begin() + p0[c]);
            if (flag) std::copy(p.flag.begin(), p.flag.end(), flag->begin() + c0[c]);
            p = RowsOut();
        }
    }, 1);
}
// sorted triples of one row -> grouped output
inline void emit_row(std::vector<Triple>& row, RowsOut& o) {
    std::sort(row.begin(), row.end(), triple_less);
    int nn = 0;
    for (size_t t = 0; t < row.size(); ++t) {
        if (t == 0 || row[t].c != row[t - 1].c) { o.col.push_back(row[t].c); o.grp_len.push_back(0); ++nn; }
        if (row[t].x >= 0) { o.x.push_back(row[t].x); o.y.push_back(row[t].y); ++o.grp_len.back(); }
    }
    o.row_nnz.push_back(nn);
}

// Z = X * Y (patterns), with the (x block, y block) pairs that sum into every Z block, row by row.
// Two passes over the rows, both parallel and without sorting the pairs: pass A counts the distinct columns and
// the pairs of every row (a per-thread marker array over the columns), a prefix sum fixes where every row lands
// in the final arrays, pass B sorts the row's distinct columns, lays out its groups and scatters the pairs into
// them in (x, y) walking order.  mirror (for the symmetric product Z = P^T (A P)): only blocks on or above the
// diagonal get a pair list; a block below it gets the index of its transpose instead (-1 elsewhere).
std::string spgemm_sym(const BlockCsr& X, const std::vector<int>* x_alias, const BlockCsr& Y, BlockCsr& Z, PairList& pl,
                       std::vector<int>* mirror = nullptr) {
    Z.n_rows = X.n_rows; Z.n_cols = Y.n_cols;
    const int n = X.n_rows, nc = Y.n_cols;
    const bool upper = mirror != nullptr;
    const auto t_begin = std::chrono::steady_clock::now();
    std::vector<int> d(n, 0);
    std::vector<int64_t> m(n, 0);
    parallel_chunks(n, [&](int, int b, int e) {
        std::vector<int> mark(nc, -1);
        for (int i = b; i < e; ++i) {
            int dd = 0; int64_t mm = 0;
            for (int a = X.ptr[i]; a < X.ptr[i + 1]; ++a) {
                const int k = X.col[a];
                for (int q = Y.ptr[k]; q < Y.ptr[k + 1]; ++q) {
                    const int c = Y.col[q];
                    if (mark[c] != i) { mark[c] = i; ++dd; }
                    if (!upper || c >= i) ++mm;
                }
            }
            d[i] = dd; m[i] = mm;
        }
    });
    Z.ptr.assign(n + 1, 0);
    std::vector<int64_t> poff(n + 1, 0);
    for (int i = 0; i < n; ++i) { Z.ptr[i + 1] = Z.ptr[i] + d[i]; poff[i + 1] = poff[i] + m[i]; }
    if (poff[n] > INT32_MAX) return "multigrid gather lists exceed 2^31 pairs";
    const int nnz = Z.ptr[n];
    Z.col.assign(nnz, 0);
    pl.ptr.resize((size_t)nnz + 1); pl.x.resize((size_t)poff[n]); pl.y.resize((size_t)poff[n]);      // uninitialised: the fill pass writes every entry
    pl.ptr[nnz] = (int)poff[n];
    Stopwatch sw3; sw3.t = t_begin; sw3.lap("    spgemm count pass");
    parallel_chunks(n, [&](int, int b, int e) {
        std::vector<int> mark(nc, -1), slot(nc, 0), cols, cnt, ord, rank, fill;
        for (int i = b; i < e; ++i) {
            cols.clear(); cnt.clear();
            for (int a = X.ptr[i]; a < X.ptr[i + 1]; ++a) {
                const int k = X.col[a];
                for (int q = Y.ptr[k]; q < Y.ptr[k + 1]; ++q) {
                    const int c = Y.col[q];
                    if (mark[c] != i) { mark[c] = i; slot[c] = (int)cols.size(); cols.push_back(c); cnt.push_back(0); }
                    if (!upper || c >= i) ++cnt[slot[c]];
                }
            }
            const int dd = (int)cols.size();
            ord.resize(dd); rank.resize(dd); fill.resize(dd);
            std::iota(ord.begin(), ord.end(), 0);
            std::sort(ord.begin(), ord.end(), [&](int p, int q) { return cols[p] < cols[q]; });
            int64_t at = poff[i];
            for (int r = 0; r < dd; ++r) {
                const int loc = ord[r];
                rank[loc] = r;
                Z.col[Z.ptr[i] + r] = cols[loc];
                pl.ptr[(size_t)Z.ptr[i] + r] = (int)at;
                fill[r] = (int)at;
                at += cnt[loc];
            }
            for (int a = X.ptr[i]; a < X.ptr[i + 1]; ++a) {
                const int k = X.col[a];
                const int xa = x_alias ? (*x_alias)[a] : a;
                for (int q = Y.ptr[k]; q < Y.ptr[k + 1]; ++q) {
                    const int c = Y.col[q];
                    if (upper && c < i) continue;
                    const int dst = fill[rank[slot[c]]]++;
                    pl.x[dst] = xa; pl.y[dst] = q;
                }
            }
        }
    });
    sw3.lap("    spgemm fill pass");
    if (upper) {
        mirror->assign(nnz, -1);
        std::atomic<bool> ok{true};
        parallel_chunks(n, [&](int, int b, int e) {
            for (int i = b; i < e; ++i)
                for (int z = Z.ptr[i]; z < Z.ptr[i + 1]; ++z) {
                    const int c = Z.col[z];
                    if (c >= i) continue;
                    const int* lo = Z.col.data() + Z.ptr[c]; const int* hi = Z.col.data() + Z.ptr[c + 1];
                    const int* f = std::lower_bound(lo, hi, i);
                    if (f == hi || *f != i) { ok = false; continue; }
                    (*mirror)[z] = (int)(f - Z.col.data());
                }
        });
        if (!ok) return "the Galerkin pattern is not structurally symmetric";
    }
    return std::string();
}

void transpose_pattern(const BlockCsr& X, BlockCsr& Xt, std::vector<int>& to_src) {
    Xt.n_rows = X.n_cols; Xt.n_cols = X.n_rows;
    Xt.ptr.assign(Xt.n_rows + 1, 0);
    for (int c : X.col) ++Xt.ptr[c + 1];
    for (int r = 0; r < Xt.n_rows; ++r) Xt.ptr[r + 1] += Xt.ptr[r];
    Xt.col.resize(X.col.size()); to_src.resize(X.col.size());
    std::vector<int> cur(Xt.ptr.begin(), Xt.ptr.end() - 1);
    for (int i = 0; i < X.n_rows; ++i)
        for (int a = X.ptr[i]; a < X.ptr[i + 1]; ++a) { const int d = cur[X.col[a]]++; Xt.col[d] = i; to_src[d] = a; }
}

std::vector<int> find_diag(const BlockCsr& A) {
    std::vector<int> d(A.n_rows, -1);
    parallel_chunks(A.n_rows, [&](int, int b, int e) {
        for (int i = b; i < e; ++i)
            for (int a = A.ptr[i]; a < A.ptr[i + 1]; ++a) if (A.col[a] == i) d[i] = a;
    }, 4096);
    return d;
}

// One coarsening step: level.A, level.agg, level.n_agg and xy are given; fills the rest and the
// pattern of the next matrix.
// rigid[i]: node i contains a pose with landmark observations.  Only those couple rotation to translation (an LM
// edge's Jacobian carries the lever arm; the reference's ODOM Jacobians are -I / +I, EdgeSe2.h:35-37), so only
// for them is "rotation about the aggregate's centroid" a slow mode; for the others it is the heading alone.
std::string coarsen(AmgLevel& L, const std::vector<double>& xy, std::vector<char>& rigid, BlockCsr& A_next, std::vector<double>& xy_next, bool smooth_p,
                    const AmgProgress* progress = nullptr, int level = 0) {
    const int n = L.n, na = L.n_agg;
    L.diag = find_diag(L.A);
    // centroids, relative coordinates
    xy_next.assign((size_t)na * 2, 0.0);
    std::vector<int> cnt(na, 0);
    for (int i = 0; i < n; ++i) { xy_next[2 * (size_t)L.agg[i]] += xy[2 * (size_t)i]; xy_next[2 * (size_t)L.agg[i] + 1] += xy[2 * (size_t)i + 1]; ++cnt[L.agg[i]]; }
    for (int a = 0; a < na; ++a) if (cnt[a]) { xy_next[2 * (size_t)a] /= cnt[a]; xy_next[2 * (size_t)a + 1] /= cnt[a]; }
    L.rel.resize((size_t)n * 2);
    for (int i = 0; i < n; ++i) {
        L.rel[2 * (size_t)i] = rigid[i] ? xy[2 * (size_t)i] - xy_next[2 * (size_t)L.agg[i]] : 0.0;
        L.rel[2 * (size_t)i + 1] = rigid[i] ? xy[2 * (size_t)i + 1] - xy_next[2 * (size_t)L.agg[i] + 1] : 0.0;
    }
    L.rig.assign(rigid.begin(), rigid.end());
    {
        std::vector<char> next(na, 0);
        for (int i = 0; i < n; ++i) next[L.agg[i]] |= rigid[i];
        rigid.swap(next);
    }
    // P pattern: aggregates of the row's neighbours
    L.P.n_rows = n; L.P.n_cols = na;
    {
        std::vector<RowsOut> parts(kMaxHostThreads);
        const int used = parallel_chunks(n, [&](int c, int b, int e) {
            std::vector<Triple> row;
            RowsOut o;
            for (int i = b; i < e; ++i) {
                row.clear();
                bool has_self = false;
                // smoothed prolongator: one block per aggregate the row touches, fed by the row's A blocks;
                // plain (tentative) prolongator on the deep levels: the row's own aggregate only, no sources
                if (smooth_p) for (int a = L.A.ptr[i]; a < L.A.ptr[i + 1]; ++a) { row.push_back({L.agg[L.A.col[a]], a, L.A.col[a]}); has_self |= L.agg[L.A.col[a]] == L.agg[i]; }
                if (!has_self) row.push_back({L.agg[i], -1, -1});      // structurally missing diagonal / tentative prolongator
                const size_t before = o.col.size();
                emit_row(row, o);
                for (size_t t = before; t < o.col.size(); ++t) o.flag.push_back(o.col[t] == L.agg[i] ? 1 : 0);
            }
            parts[c] = std::move(o);
        });
        concat(parts, used, L.P, L.p_src, &L.p_self);
    }
    Stopwatch sw2;
    sw2.lap("  (P pattern)");
    transpose_pattern(L.P, L.R, L.r_to_p);
    sw2.lap("  transpose");
    std::string err;
    if (progress && progress->products && progress->products(level, L, A_next, err)) {       // both products, elsewhere (the device)
        sw2.lap("  T = A P, A' = R T (device)");
        return err;
    }
    if (!err.empty()) return err;
    err = spgemm_sym(L.A, nullptr, L.P, L.T, L.t_src);
    sw2.lap("  T = A P");
    if (err.empty()) err = spgemm_sym(L.R, &L.r_to_p, L.T, A_next, L.a_src, &L.a_mirror);
    sw2.lap("  A' = R T");
    return err;
}


// ---- aggregation by heavy-edge matching ------------------------------------------------------------------
// Weighted graph of the couplings (no diagonal).  Level 0: weight = number of landmarks two poses share
// (+ their odometry edges); below: the sums over the merged groups.
struct WGraph {
    int n = 0;
    std::vector<int> ptr, col;
    std::vector<float> w;
};

// Groups of `g` (cmap: node -> group, nc groups) become the nodes of the returned graph.  Row a of the result lists the groups that
// a's members couple to, in the order the members' edges meet them, weights summed in that order: the rows are independent, so
// chunks of them are built side by side (every chunk with its own marker arrays) and laid out by a prefix over the chunks' sizes —
// the same rows whatever the thread count.  (Serial, this ran four times in the level-0 matching: 8 of its 14 ms at 100k poses.)
WGraph contract(const WGraph& g, const std::vector<int>& cmap, int nc) {
    std::vector<int> mptr(nc + 1, 0), mem(g.n);
    for (int i = 0; i < g.n; ++i) ++mptr[cmap[i] + 1];
    for (int a = 0; a < nc; ++a) mptr[a + 1] += mptr[a];
    { std::vector<int> cur(mptr.begin(), mptr.end() - 1); for (int i = 0; i < g.n; ++i) mem[cur[cmap[i]]++] = i; }
    WGraph c; c.n = nc; c.ptr.assign(nc + 1, 0);
    struct Part { std::vector<int> col; std::vector<float> w; };
    std::vector<Part> parts(kMaxHostThreads);
    std::vector<int> first(kMaxHostThreads + 1, 0);
    const int used = parallel_chunks(nc, [&](int ch, int b, int e) {
        std::vector<int> mark(nc, -1), pos(nc, 0);
        Part& o = parts[ch];
        first[ch] = b;
        for (int a = b; a < e; ++a) {
            const size_t row0 = o.col.size();
            for (int m = mptr[a]; m < mptr[a + 1]; ++m) {
                const int i = mem[m];
                for (int ed = g.ptr[i]; ed < g.ptr[i + 1]; ++ed) {
                    const int bb = cmap[g.col[ed]];
                    if (bb == a) continue;
                    if (mark[bb] != a) { mark[bb] = a; pos[bb] = (int)o.col.size(); o.col.push_back(bb); o.w.push_back(g.w[ed]); }
                    else o.w[pos[bb]] += g.w[ed];
                }
            }
            c.ptr[a + 1] = (int)(o.col.size() - row0);        // the row's length; prefixed below
        }
    }, 2048);
    for (int a = 0; a < nc; ++a) c.ptr[a + 1] += c.ptr[a];
    c.col.resize((size_t)c.ptr[nc]); c.w.resize((size_t)c.ptr[nc]);
    parallel_chunks(used, [&](int, int cb, int ce) {
        for (int ch = cb; ch < ce; ++ch) {
            std::copy(parts[ch].col.begin(), parts[ch].col.end(), c.col.begin() + c.ptr[first[ch]]);
            std::copy(parts[ch].w.begin(), parts[ch].w.end(), c.w.begin() + c.ptr[first[ch]]);
        }
    }, 1);
    return c;
}

// Aggregates of up to `target` nodes (a power of two) by log2(target) passes of heavy-edge matching: every
// pass visits the groups in `key` order and merges each unmatched group with the unmatched neighbour it is
// most strongly coupled to.  Groups left small at the end join their strongest neighbour.  The groups are
// numbered by their smallest key (memory locality along the trajectory).  g is replaced by the coarse graph.
int aggregate_by_matching(WGraph& g, std::vector<int>& key, int target, std::vector<int>& agg) {
    const int n0 = g.n;
    agg.resize(n0);
    std::iota(agg.begin(), agg.end(), 0);
    std::vector<int> size(n0, 1);
    for (int span = 2; span <= target; span *= 2) {
        const int n = g.n;
        std::vector<int> visit(n);
        std::iota(visit.begin(), visit.end(), 0);
        std::sort(visit.begin(), visit.end(), [&](int a, int b) { return key[a] < key[b]; });
        std::vector<int> mate(n, -1), cmap(n, -1);
        int nc = 0;
        for (int i : visit) {
            if (cmap[i] >= 0) continue;
            int best = -1; float bw = 0;
            for (int e = g.ptr[i]; e < g.ptr[i + 1]; ++e) {
                const int k = g.col[e];
                if (cmap[k] >= 0 || size[i] + size[k] > span) continue;
                if (g.w[e] > bw || (g.w[e] == bw && best >= 0 && key[k] < key[best])) { bw = g.w[e]; best = k; }
            }
            cmap[i] = nc;
            if (best >= 0) cmap[best] = nc;
            ++nc;
        }
        std::vector<int> nsize(nc, 0), nkey(nc, INT32_MAX);
        for (int i = 0; i < n; ++i) { nsize[cmap[i]] += size[i]; nkey[cmap[i]] = std::min(nkey[cmap[i]], key[i]); }
        for (int v = 0; v < n0; ++v) agg[v] = cmap[agg[v]];
        g = contract(g, cmap, nc);
        size.swap(nsize); key.swap(nkey);
    }
    // groups below a quarter of the target join the neighbour they are most strongly coupled to
    {
        const int n = g.n;
        std::vector<int> cmap(n);
        std::iota(cmap.begin(), cmap.end(), 0);
        bool any = false;
        for (int i = 0; i < n; ++i) {
            if (size[i] * 4 > target) continue;
            int best = -1; float bw = 0;
            for (int e = g.ptr[i]; e < g.ptr[i + 1]; ++e) { const int k = g.col[e]; if (size[k] * 4 > target && g.w[e] > bw) { bw = g.w[e]; best = k; } }
            if (best >= 0) { cmap[i] = best; any = true; }
        }
        if (any) {
            std::vector<int> dense(n, -1); int nc = 0;
            for (int i = 0; i < n; ++i) if (cmap[i] == i) dense[i] = nc++;
            for (int i = 0; i < n; ++i) cmap[i] = dense[cmap[i]];
            std::vector<int> nsize(nc, 0), nkey(nc, INT32_MAX);
            for (int i = 0; i < n; ++i) { nsize[cmap[i]] += size[i]; nkey[cmap[i]] = std::min(nkey[cmap[i]], key[i]); }
            for (int v = 0; v < n0; ++v) agg[v] = cmap[agg[v]];
            g = contract(g, cmap, nc);
            size.swap(nsize); key.swap(nkey);
        }
    }
    // number the groups by key
    {
        const int n = g.n;
        std::vector<int> ord(n), rank(n);
        std::iota(ord.begin(), ord.end(), 0);
        std::sort(ord.begin(), ord.end(), [&](int a, int b) { return key[a] < key[b]; });
        for (int r = 0; r < n; ++r) rank[ord[r]] = r;
        for (int v = 0; v < n0; ++v) agg[v] = rank[agg[v]];
        g = contract(g, rank, n);
        std::vector<int> nkey(n);
        for (int i = 0; i < n; ++i) nkey[rank[i]] = key[i];
        key.swap(nkey);
    }
    return g.n;
}

}  // namespace

std::string build_amg(const Problem& pr, AmgSym& out, const AmgProgress* progress) {
    if (pr.world != 1) return "build_amg takes the WHOLE graph's layout (a shard goes through build_amg_sharded, which builds the whole graph's patterns and keeps its own contribution lists)";
    Stopwatch sw;
    out = AmgSym();
    AmgSym& S = out;
    S.levels.reserve(32);        // never reallocates: a consumer may read finished levels while later ones are built
    const int P = pr.P;
    // ---- trajectory order: follow ODOM edges id1 -> id2 where that is a simple chain ------------------
    {
        std::vector<int> next(P, -1), indeg(P, 0), nout(P, 0);
        for (int i = 0; i < P; ++i)
            for_slots(pr.odom, i, [&](size_t k) {
                const uint32_t raw = pr.odom.idx[k];
                if (raw & (kDirBit | kVlmBit)) return;        // this row is id2 / not an odometry edge (a virtual landmark measurement joins any two poses)
                const int j = (int)(raw & kPoseMask);
                if (j == i) return;
                if (nout[i]++ == 0) { next[i] = j; ++indeg[j]; }
            });
        std::vector<char> seen(P, 0);
        S.order.assign(P, 0);
        int pos = 0;
        auto walk = [&](int s) { for (int v = s; v >= 0 && !seen[v]; v = next[v]) { seen[v] = 1; S.order[v] = pos++; } };
        std::vector<int> by_vertex(P);
        std::iota(by_vertex.begin(), by_vertex.end(), 0);
        std::sort(by_vertex.begin(), by_vertex.end(), [&](int a, int b) { return pr.pose_vertex[a] < pr.pose_vertex[b]; });
        for (int v : by_vertex) if (indeg[v] == 0) walk(v);   // chain heads, in input order
        for (int v : by_vertex) walk(v);                      // cycles / leftovers
    }
    sw.lap("trajectory order");
    // ---- level 0: pattern of S and the contribution lists --------------------------------------------
    uint32_t max_edge = 0;
    for (uint32_t e : pr.by_pose.edge) if (e != kNoEdge) max_edge = std::max(max_edge, e);
    std::vector<uint32_t> epos((size_t)max_edge + 1, 0);
    parallel_chunks((int)pr.by_pose.edge.size(), [&](int, int b, int e) {
        for (int s = b; s < e; ++s) if (pr.by_pose.edge[s] != kNoEdge) epos[pr.by_pose.edge[s]] = (uint32_t)s;      // an edge sits in one slot
    }, 1 << 16);
    // observers of every landmark: (pose, by_pose slot of that edge), in the slot order of the landmark's row (count, prefix,
    // parallel fill — the serial walk took 32 ms of the 145 ms critical path at 100k poses)
    std::vector<int> obs_ptr(pr.L + 1, 0);
    parallel_chunks(pr.L, [&](int, int b, int e) {
        for (int l = b; l < e; ++l) { int n = 0; for_slots(pr.by_lm, l, [&](size_t) { ++n; }); obs_ptr[l + 1] = n; }
    }, 4096);
    for (int l = 0; l < pr.L; ++l) obs_ptr[l + 1] += obs_ptr[l];
    std::vector<int> obs_pose((size_t)obs_ptr[pr.L]); std::vector<uint32_t> obs_slot((size_t)obs_ptr[pr.L]);
    parallel_chunks(pr.L, [&](int, int b, int e) {
        for (int l = b; l < e; ++l) {
            int at = obs_ptr[l];
            for_slots(pr.by_lm, l, [&](size_t k) { obs_pose[at] = (int)pr.by_lm.idx[k]; obs_slot[at] = epos[pr.by_lm.edge[k]]; ++at; });
        }
    }, 4096);
    sw.lap("landmark observers");
    AmgLevel L0;
    L0.n = P;
    L0.A.n_rows = L0.A.n_cols = P;
    bool schur_elsewhere = false;
    if (progress && progress->schur) {
        // the same inputs as CSR lists, laid out in parallel, for a builder that is not this one (the device)
        SchurCsr in; in.P = P; in.L = pr.L; in.max_pair_degree = kMaxPairDegree;
        in.pp_ptr.assign(P + 1, 0); in.od_ptr.assign(P + 1, 0);
        parallel_chunks(P, [&](int, int b, int e) {
            for (int i = b; i < e; ++i) {
                int n = 0, no = 0;
                for_slots(pr.by_pose, i, [&](size_t) { ++n; });
                for_slots(pr.odom, i, [&](size_t k) { if ((int)(pr.odom.idx[k] & kPoseMask) != i) ++no; });
                in.pp_ptr[i + 1] = n; in.od_ptr[i + 1] = no;
            }
        }, 4096);
        for (int i = 0; i < P; ++i) { in.pp_ptr[i + 1] += in.pp_ptr[i]; in.od_ptr[i + 1] += in.od_ptr[i]; }
        in.pp_lm.resize((size_t)in.pp_ptr[P]); in.pp_slot.resize((size_t)in.pp_ptr[P]);
        in.od_col.resize((size_t)in.od_ptr[P]); in.od_slot.resize((size_t)in.od_ptr[P]);
        parallel_chunks(P, [&](int, int b, int e) {
            for (int i = b; i < e; ++i) {
                int at = in.pp_ptr[i], ao = in.od_ptr[i];
                for_slots(pr.by_pose, i, [&](size_t k) { in.pp_lm[at] = (int)pr.by_pose.idx[k]; in.pp_slot[at] = (uint32_t)k; ++at; });
                for_slots(pr.odom, i, [&](size_t k) { const int j = (int)(pr.odom.idx[k] & kPoseMask); if (j != i) { in.od_col[ao] = j; in.od_slot[ao] = (uint32_t)k; ++ao; } });
            }
        }, 4096);
        in.obs_ptr = std::move(obs_ptr); in.obs_pose = std::move(obs_pose); in.obs_slot = std::move(obs_slot);
        std::string herr;
        schur_elsewhere = progress->schur(in, L0.A, S.schur.ptr, S.schur.od_ptr, herr);
        if (!herr.empty()) return herr;
        if (!schur_elsewhere) { obs_ptr = std::move(in.obs_ptr); obs_pose = std::move(in.obs_pose); obs_slot = std::move(in.obs_slot); }
        else { L0.A.n_rows = L0.A.n_cols = P; sw.lap("S pattern + lists (device)"); }
    }
    struct Tup { int k; uint32_t a, b; int kind; };            // kind 0: landmark pair, 1: odom slot
    struct SOut { std::vector<int> row_nnz, col, n_pair, n_od; std::vector<uint32_t> si, sk, os; };
    if (!schur_elsewhere) {
        std::vector<SOut> parts(kMaxHostThreads);
        const int used = parallel_chunks(P, [&](int c, int b, int e) {
            std::vector<Tup> row;
            SOut o;
            for (int i = b; i < e; ++i) {
                row.clear();
                for_slots(pr.by_pose, i, [&](size_t k) {
                    const int l = (int)pr.by_pose.idx[k];
                    // a landmark seen from d poses couples d^2 pose pairs; beyond kMaxPairDegree its couplings are
                    // left out of the PRECONDITIONER's explicit matrix (its diagonal part stays; the Schur
                    // product itself is exact either way), which bounds the list memory on hub landmarks
                    if (obs_ptr[l + 1] - obs_ptr[l] > kMaxPairDegree) return;
                    for (int q = obs_ptr[l]; q < obs_ptr[l + 1]; ++q)
                        if (obs_pose[q] != i) row.push_back({obs_pose[q], (uint32_t)k, obs_slot[q], 0});
                });
                for_slots(pr.odom, i, [&](size_t k) {
                    const int j = (int)(pr.odom.idx[k] & kPoseMask);
                    if (j != i) row.push_back({j, (uint32_t)k, 0u, 1});
                });
                std::sort(row.begin(), row.end(), [](const Tup& p, const Tup& q) {
                    return p.k != q.k ? p.k < q.k : (p.kind != q.kind ? p.kind < q.kind : (p.a != q.a ? p.a < q.a : p.b < q.b)); });
                int nn = 0; bool diag_done = false;
                auto emit_diag = [&] { o.col.push_back(i); o.n_pair.push_back(0); o.n_od.push_back(0); diag_done = true; ++nn; };
                for (size_t t = 0; t < row.size();) {
                    const int k = row[t].k;
                    if (!diag_done && k > i) emit_diag();
                    o.col.push_back(k); o.n_pair.push_back(0); o.n_od.push_back(0); ++nn;
                    for (; t < row.size() && row[t].k == k; ++t) {
                        if (row[t].kind == 0) { o.si.push_back(row[t].a); o.sk.push_back(row[t].b); ++o.n_pair.back(); }
                        else { o.os.push_back(row[t].a); ++o.n_od.back(); }
                    }
                }
                if (!diag_done) emit_diag();
                o.row_nnz.push_back(nn);
            }
            parts[c] = std::move(o);
        });
        // the chunks' outputs, side by side (offsets from their sizes)
        std::vector<size_t> r0(used + 1, 0), c0(used + 1, 0), q0(used + 1, 0), d0(used + 1, 0);
        for (int c = 0; c < used; ++c) {
            r0[c + 1] = r0[c] + parts[c].row_nnz.size(); c0[c + 1] = c0[c] + parts[c].col.size();
            q0[c + 1] = q0[c] + parts[c].si.size(); d0[c + 1] = d0[c] + parts[c].os.size();
        }
        L0.A.ptr.resize(r0[used] + 1); L0.A.col.resize(c0[used]);
        S.schur.ptr.resize(c0[used] + 1); S.schur.od_ptr.resize(c0[used] + 1);
        S.schur.slot_i.resize(q0[used]); S.schur.slot_k.resize(q0[used]); S.schur.od_slot.resize(d0[used]);
        L0.A.ptr[0] = 0; S.schur.ptr[0] = 0; S.schur.od_ptr[0] = 0;
        parallel_chunks(used, [&](int, int cb, int ce) {
            for (int c = cb; c < ce; ++c) {
                SOut& o = parts[c];
                int at = (int)c0[c];
                for (size_t k = 0; k < o.row_nnz.size(); ++k) L0.A.ptr[r0[c] + k + 1] = (at += o.row_nnz[k]);
                std::copy(o.col.begin(), o.col.end(), L0.A.col.begin() + c0[c]);
                int qa = (int)q0[c], da = (int)d0[c];
                for (size_t k = 0; k < o.n_pair.size(); ++k) { S.schur.ptr[c0[c] + k + 1] = (qa += o.n_pair[k]); S.schur.od_ptr[c0[c] + k + 1] = (da += o.n_od[k]); }
                std::copy(o.si.begin(), o.si.end(), S.schur.slot_i.begin() + q0[c]); std::copy(o.sk.begin(), o.sk.end(), S.schur.slot_k.begin() + q0[c]);
                std::copy(o.os.begin(), o.os.end(), S.schur.od_slot.begin() + d0[c]);
                o = SOut();
            }
        }, 1);
    }
    if (!schur_elsewhere) sw.lap("S pattern + lists");
    if (progress && progress->schur_ready) progress->schur_ready();

    // ---- hierarchy ----------------------------------------------------------------------------------------
    std::vector<double> xy((size_t)P * 2);
    for (int i = 0; i < P; ++i) { xy[2 * (size_t)i] = pr.pose_xyt[3 * (size_t)i]; xy[2 * (size_t)i + 1] = pr.pose_xyt[3 * (size_t)i + 1]; }
    std::vector<char> rigid(P, 0);
    for (int i = 0; i < P; ++i) for_slots(pr.by_pose, i, [&](size_t) { rigid[i] = 1; });
    if (pr.odom_analytic) for (int i = 0; i < P; ++i) for_slots(pr.odom, i, [&](size_t) { rigid[i] = 1; });   // A = [[-M, q], ..]: q is the lever arm
    if (pr.has_vlm) for (int i = 0; i < P; ++i) for_slots(pr.odom, i, [&](size_t k) { if (pr.odom.idx[k] & kVlmBit) rigid[i] = 1; });      // A = [I | dR/dth p]: a lever arm too
    L0.agg.resize(P);
    // aggregate size per level; research override: TSGO_AGG_LIST="8,4,4,8" (last entry repeats) or TSGO_AGG0 / TSGO_AGGC
    std::vector<int> agg_list;
    if (const char* e = TSGO_RESEARCH_ENV("TSGO_AGG_LIST")) { for (const char* q = e; *q;) { agg_list.push_back(std::max(2, atoi(q))); while (*q && *q != ',') ++q; if (*q == ',') ++q; } }
    if (agg_list.empty()) {
        agg_list.push_back(TSGO_RESEARCH_INT("TSGO_AGG0", kAggSize));
        if (const char* e = TSGO_RESEARCH_ENV("TSGO_AGGC")) agg_list.push_back(atoi(e));
        else for (int m : kAggSizesBelow) agg_list.push_back(m);
    }
    auto agg_at = [&](size_t l) { return agg_list[std::min(l, agg_list.size() - 1)]; };
    const int agg0 = agg_at(0);
    const int smooth_levels = TSGO_RESEARCH_INT("TSGO_SMOOTH_LEVELS", kSmoothLevels);
    const int smooth_from = TSGO_RESEARCH_INT("TSGO_SMOOTH_FROM", 0);     // research: tentative prolongators above this level
    // research override: TSGO_AGG_MODE=traj cuts the trajectory into runs of consecutive poses instead (the first version)
    const char* agg_mode = TSGO_RESEARCH_ENV("TSGO_AGG_MODE");
    const bool matching = !(agg_mode && std::string(agg_mode) == "traj");
    WGraph wg; std::vector<int> wkey;
    if (matching) {
        const float w_od = (float)TSGO_RESEARCH_FLOAT("TSGO_AGG_WOD", kAggOdomWeight);
        wg.n = P; wg.ptr.assign(P + 1, 0);
        for (int i = 0; i < P; ++i) wg.ptr[i + 1] = wg.ptr[i] + (L0.A.ptr[i + 1] - L0.A.ptr[i] - 1);      // every row holds its diagonal block (emit_diag)
        wg.col.resize(wg.ptr[P]); wg.w.resize(wg.ptr[P]);
        parallel_chunks(P, [&](int, int b, int e) {
            for (int i = b; i < e; ++i) {
                int at = wg.ptr[i];
                for (int a = L0.A.ptr[i]; a < L0.A.ptr[i + 1]; ++a) {
                    if (L0.A.col[a] == i) continue;
                    wg.col[at] = L0.A.col[a];
                    wg.w[at++] = (float)(S.schur.ptr[a + 1] - S.schur.ptr[a]) + w_od * (float)(S.schur.od_ptr[a + 1] - S.schur.od_ptr[a]);
                }
            }
        }, 4096);
        wkey = S.order;
        L0.n_agg = aggregate_by_matching(wg, wkey, agg0, L0.agg);
    } else {
        for (int i = 0; i < P; ++i) L0.agg[i] = S.order[i] / agg0;
        L0.n_agg = (P + agg0 - 1) / agg0;
    }
    sw.lap("aggregation");
    AmgLevel cur = std::move(L0);
    for (;;) {
        if (cur.n <= kCoarsestMax) { S.A_last = cur.A; S.diag_last = find_diag(cur.A); break; }
        // The NEXT level's aggregates depend on the contracted coupling graph alone (aggregate_by_matching leaves it in wg),
        // not on the Galerkin pattern: they are matched by a helper thread while this level's products are laid out.
        const int na = cur.n_agg;
        const int aggc = agg_at(S.levels.size() + 1);
        const bool match_next = matching && na > kCoarsestMax;
        std::vector<int> next_agg; int next_n_agg = 0;
        std::thread helper;
        if (match_next) helper = std::thread([&] { next_n_agg = aggregate_by_matching(wg, wkey, aggc, next_agg); });
        BlockCsr A_next; std::vector<double> xy_next;
        const std::string cerr = coarsen(cur, xy, rigid, A_next, xy_next, (int)S.levels.size() < smooth_levels && (int)S.levels.size() >= smooth_from, progress, (int)S.levels.size());
        if (helper.joinable()) helper.join();
        if (!cerr.empty()) return cerr;
        sw.lap("coarsen level");
        S.levels.push_back(std::move(cur));
        if (S.levels.size() >= 32) return "too many multigrid levels";
        if (progress && progress->level_ready) progress->level_ready((int)S.levels.size());
        cur = AmgLevel();
        cur.n = na; cur.A = std::move(A_next);
        if (match_next) { cur.agg = std::move(next_agg); cur.n_agg = next_n_agg; }
        else {
            cur.agg.resize(na);
            for (int a = 0; a < na; ++a) cur.agg[a] = a / aggc;     // aggregates are numbered along the trajectory
            cur.n_agg = (na + aggc - 1) / aggc;
        }
        xy = std::move(xy_next);
    }
    return std::string();
}

std::string build_amg_sharded(const tsgo_graph& g, const Problem& local, AmgSym& out) {
    Problem full;
    BuildOptions bo; bo.lanes_per_pose = local.by_pose.G; bo.lanes_per_lm = local.by_lm.G;
    std::string err = build_problem(g, bo, full);
    if (!err.empty()) return err;
    full.odom_analytic = local.odom_analytic; full.has_vlm = local.has_vlm;
    if (full.P != local.P || full.pose_vertex != local.pose_vertex) return "shard and whole-graph pose numbering differ";
    AmgSym S;
    err = build_amg(full, S);
    if (!err.empty()) return err;
    // edge -> slot of the LOCAL tables (LM edges: by_pose slot; ODOM edges: one slot per owned endpoint)
    const size_t nE = (size_t)std::max(0, g.n_edges);
    std::vector<uint32_t> lm_slot(nE, kNoEdge), od_slot(2 * nE, kNoEdge);
    for (size_t s = 0; s < local.by_pose.edge.size(); ++s) if (local.by_pose.edge[s] != kNoEdge) lm_slot[local.by_pose.edge[s]] = (uint32_t)s;
    for (size_t s = 0; s < local.odom.edge.size(); ++s)
        if (local.odom.edge[s] != kNoEdge) od_slot[2 * (size_t)local.odom.edge[s] + ((local.odom.idx[s] & kDirBit) ? 1 : 0)] = (uint32_t)s;
    SchurLists mine;
    const size_t nb = S.schur.ptr.size() - 1;
    mine.ptr.assign(1, 0); mine.od_ptr.assign(1, 0);
    mine.ptr.reserve(nb + 1); mine.od_ptr.reserve(nb + 1);
    for (size_t b = 0; b < nb; ++b) {
        for (int q = S.schur.ptr[b]; q < S.schur.ptr[b + 1]; ++q) {
            const uint32_t ei = full.by_pose.edge[S.schur.slot_i[q]], ek = full.by_pose.edge[S.schur.slot_k[q]];
            const uint32_t si = lm_slot[ei], sk = lm_slot[ek];
            if (si == kNoEdge) continue;                       // the shared landmark belongs to another shard
            if (sk == kNoEdge) return "a landmark's edges are split across shards";
            mine.slot_i.push_back(si); mine.slot_k.push_back(sk);
        }
        mine.ptr.push_back((int)mine.slot_i.size());
        for (int q = S.schur.od_ptr[b]; q < S.schur.od_ptr[b + 1]; ++q) {
            const size_t fs = S.schur.od_slot[q];
            const uint32_t s = od_slot[2 * (size_t)full.odom.edge[fs] + ((full.odom.idx[fs] & kDirBit) ? 1 : 0)];
            if (s != kNoEdge) mine.od_slot.push_back(s);       // the row pose is owned by this shard
        }
        mine.od_ptr.push_back((int)mine.od_slot.size());
    }
    S.schur = std::move(mine);
    out = std::move(S);
    return std::string();
}

void refresh_amg_geometry(const std::vector<double>& pose_xyt, AmgSym& amg) {
    if (amg.levels.empty()) return;
    const int P = amg.levels[0].n;
    std::vector<double> xy((size_t)P * 2), next;
    for (int i = 0; i < P; ++i) { xy[2 * (size_t)i] = pose_xyt[3 * (size_t)i]; xy[2 * (size_t)i + 1] = pose_xyt[3 * (size_t)i + 1]; }
    for (AmgLevel& L : amg.levels) {          // the same arithmetic, in the same order, as coarsen()
        const int n = L.n, na = L.n_agg;
        next.assign((size_t)na * 2, 0.0);
        std::vector<int> cnt(na, 0);
        for (int i = 0; i < n; ++i) { next[2 * (size_t)L.agg[i]] += xy[2 * (size_t)i]; next[2 * (size_t)L.agg[i] + 1] += xy[2 * (size_t)i + 1]; ++cnt[L.agg[i]]; }
        for (int a = 0; a < na; ++a) if (cnt[a]) { next[2 * (size_t)a] /= cnt[a]; next[2 * (size_t)a + 1] /= cnt[a]; }
        for (int i = 0; i < n; ++i) {
            L.rel[2 * (size_t)i] = L.rig[i] ? xy[2 * (size_t)i] - next[2 * (size_t)L.agg[i]] : 0.0;
            L.rel[2 * (size_t)i + 1] = L.rig[i] ? xy[2 * (size_t)i + 1] - next[2 * (size_t)L.agg[i] + 1] : 0.0;
        }
        xy.swap(next);
    }
}

}  // namespace tsgo
