// amg.cpp — host-side symbolic setup of the smoothed-aggregation hierarchy (see amg.h).
#include "amg.h"

#include <algorithm>
#include <cstdlib>
#include <numeric>

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

struct Triple { int c, x, y; };

// Z = X * Y (patterns), with the (x block, y block) pairs that sum into every Z block, row by row.
void spgemm_sym(const BlockCsr& X, const std::vector<int>* x_alias, const BlockCsr& Y, BlockCsr& Z, PairList& pl) {
    Z.n_rows = X.n_rows; Z.n_cols = Y.n_cols; Z.ptr.assign(1, 0); Z.col.clear();
    pl.ptr.assign(1, 0); pl.x.clear(); pl.y.clear();
    std::vector<Triple> row;
    for (int i = 0; i < X.n_rows; ++i) {
        row.clear();
        for (int a = X.ptr[i]; a < X.ptr[i + 1]; ++a) {
            const int k = X.col[a];
            for (int b = Y.ptr[k]; b < Y.ptr[k + 1]; ++b) row.push_back({Y.col[b], x_alias ? (*x_alias)[a] : a, b});
        }
        std::stable_sort(row.begin(), row.end(), [](const Triple& p, const Triple& q) { return p.c < q.c; });
        for (size_t t = 0; t < row.size(); ++t) {
            if (t == 0 || row[t].c != row[t - 1].c) {
                if (t) pl.ptr.push_back((int)pl.x.size());
                Z.col.push_back(row[t].c);
            }
            pl.x.push_back(row[t].x); pl.y.push_back(row[t].y);
        }
        if (!row.empty()) pl.ptr.push_back((int)pl.x.size());
        Z.ptr.push_back((int)Z.col.size());
    }
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
    for (int i = 0; i < A.n_rows; ++i)
        for (int a = A.ptr[i]; a < A.ptr[i + 1]; ++a) if (A.col[a] == i) d[i] = a;
    return d;
}

// One coarsening step: level.A, level.agg, level.n_agg and xy are given; fills the rest and the
// pattern of the next matrix.
void coarsen(AmgLevel& L, const std::vector<double>& xy, BlockCsr& A_next, std::vector<double>& xy_next) {
    const int n = L.n, na = L.n_agg;
    L.diag = find_diag(L.A);
    // centroids, relative coordinates
    xy_next.assign((size_t)na * 2, 0.0);
    std::vector<int> cnt(na, 0);
    for (int i = 0; i < n; ++i) { xy_next[2 * (size_t)L.agg[i]] += xy[2 * (size_t)i]; xy_next[2 * (size_t)L.agg[i] + 1] += xy[2 * (size_t)i + 1]; ++cnt[L.agg[i]]; }
    for (int a = 0; a < na; ++a) if (cnt[a]) { xy_next[2 * (size_t)a] /= cnt[a]; xy_next[2 * (size_t)a + 1] /= cnt[a]; }
    L.rel.resize((size_t)n * 2);
    for (int i = 0; i < n; ++i) { L.rel[2 * (size_t)i] = xy[2 * (size_t)i] - xy_next[2 * (size_t)L.agg[i]]; L.rel[2 * (size_t)i + 1] = xy[2 * (size_t)i + 1] - xy_next[2 * (size_t)L.agg[i] + 1]; }
    // P pattern: aggregates of the row's neighbours
    L.P.n_rows = n; L.P.n_cols = na; L.P.ptr.assign(1, 0); L.P.col.clear(); L.p_self.clear();
    L.p_src.ptr.assign(1, 0); L.p_src.x.clear(); L.p_src.y.clear();
    std::vector<Triple> row;
    for (int i = 0; i < n; ++i) {
        row.clear();
        bool has_self = false;
        for (int a = L.A.ptr[i]; a < L.A.ptr[i + 1]; ++a) { row.push_back({L.agg[L.A.col[a]], a, L.A.col[a]}); has_self |= L.agg[L.A.col[a]] == L.agg[i]; }
        if (!has_self) row.push_back({L.agg[i], -1, -1});      // structurally missing diagonal
        std::stable_sort(row.begin(), row.end(), [](const Triple& p, const Triple& q) { return p.c < q.c; });
        for (size_t t = 0; t < row.size(); ++t) {
            if (t == 0 || row[t].c != row[t - 1].c) {
                if (t) L.p_src.ptr.push_back((int)L.p_src.x.size());
                L.P.col.push_back(row[t].c); L.p_self.push_back(row[t].c == L.agg[i] ? 1 : 0);
            }
            if (row[t].x >= 0) { L.p_src.x.push_back(row[t].x); L.p_src.y.push_back(row[t].y); }
        }
        L.p_src.ptr.push_back((int)L.p_src.x.size());
        L.P.ptr.push_back((int)L.P.col.size());
    }
    transpose_pattern(L.P, L.R, L.r_to_p);
    spgemm_sym(L.A, nullptr, L.P, L.T, L.t_src);
    spgemm_sym(L.R, &L.r_to_p, L.T, A_next, L.a_src);
}

}  // namespace

std::string build_amg(const Problem& pr, AmgSym& out) {
    if (pr.world != 1) return "the multigrid preconditioner is single-shard";
    AmgSym S;
    const int P = pr.P;
    // ---- trajectory order: follow ODOM edges id1 -> id2 where that is a simple chain ------------------
    {
        std::vector<int> next(P, -1), indeg(P, 0), nout(P, 0);
        for (int i = 0; i < P; ++i)
            for_slots(pr.odom, i, [&](size_t k) {
                const uint32_t raw = pr.odom.idx[k];
                if (raw & kDirBit) return;                    // this row is id2
                const int j = (int)(raw & ~kDirBit);
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
    // ---- level 0: pattern of S and the contribution lists --------------------------------------------
    uint32_t max_edge = 0;
    for (uint32_t e : pr.by_pose.edge) if (e != kNoEdge) max_edge = std::max(max_edge, e);
    std::vector<uint32_t> epos((size_t)max_edge + 1, 0);
    for (size_t s = 0; s < pr.by_pose.edge.size(); ++s) if (pr.by_pose.edge[s] != kNoEdge) epos[pr.by_pose.edge[s]] = (uint32_t)s;

    struct Tup { int i, k; uint32_t a, b; int kind; };          // kind 0: landmark pair, 1: odom slot
    std::vector<int> row_count(P + 1, 0);
    std::vector<Tup> tup;
    {
        std::vector<std::pair<int, uint32_t>> obs;
        for (int l = 0; l < pr.L; ++l) {
            obs.clear();
            for_slots(pr.by_lm, l, [&](size_t k) { obs.emplace_back((int)pr.by_lm.idx[k], epos[pr.by_lm.edge[k]]); });
            for (size_t x = 0; x < obs.size(); ++x)
                for (size_t y = 0; y < obs.size(); ++y)
                    if (obs[x].first != obs[y].first) tup.push_back({obs[x].first, obs[y].first, obs[x].second, obs[y].second, 0});
        }
        for (int i = 0; i < P; ++i)
            for_slots(pr.odom, i, [&](size_t k) {
                const int j = (int)(pr.odom.idx[k] & ~kDirBit);
                if (j != i) tup.push_back({i, j, (uint32_t)k, 0u, 1});
            });
    }
    for (const Tup& t : tup) ++row_count[t.i + 1];
    for (int i = 0; i < P; ++i) row_count[i + 1] += row_count[i];
    std::vector<Tup> sorted(tup.size());
    {
        std::vector<int> cur(row_count.begin(), row_count.end() - 1);
        for (const Tup& t : tup) sorted[cur[t.i]++] = t;
    }
    tup.clear(); tup.shrink_to_fit();
    AmgLevel L0;
    L0.n = P;
    L0.A.n_rows = L0.A.n_cols = P; L0.A.ptr.assign(1, 0);
    S.schur.ptr.assign(1, 0); S.schur.od_ptr.assign(1, 0);
    for (int i = 0; i < P; ++i) {
        auto b = sorted.begin() + row_count[i], e = sorted.begin() + row_count[i + 1];
        std::stable_sort(b, e, [](const Tup& p, const Tup& q) { return p.k < q.k; });
        bool diag_done = false;
        auto emit_diag = [&] { L0.A.col.push_back(i); S.schur.ptr.push_back((int)S.schur.slot_i.size()); S.schur.od_ptr.push_back((int)S.schur.od_slot.size()); diag_done = true; };
        for (auto it = b; it != e;) {
            const int k = it->k;
            if (!diag_done && k > i) emit_diag();
            L0.A.col.push_back(k);
            for (; it != e && it->k == k; ++it) {
                if (it->kind == 0) { S.schur.slot_i.push_back(it->a); S.schur.slot_k.push_back(it->b); }
                else S.schur.od_slot.push_back(it->a);
            }
            S.schur.ptr.push_back((int)S.schur.slot_i.size()); S.schur.od_ptr.push_back((int)S.schur.od_slot.size());
        }
        if (!diag_done) emit_diag();
        L0.A.ptr.push_back((int)L0.A.col.size());
    }
    sorted.clear(); sorted.shrink_to_fit();

    // ---- hierarchy ----------------------------------------------------------------------------------------
    std::vector<double> xy((size_t)P * 2);
    for (int i = 0; i < P; ++i) { xy[2 * (size_t)i] = pr.pose_xyt[3 * (size_t)i]; xy[2 * (size_t)i + 1] = pr.pose_xyt[3 * (size_t)i + 1]; }
    L0.agg.resize(P);
    const int agg0 = getenv("TSGO_AGG0") ? atoi(getenv("TSGO_AGG0")) : kAggSize;
    const int aggc = getenv("TSGO_AGGC") ? atoi(getenv("TSGO_AGGC")) : kAggSize;
    for (int i = 0; i < P; ++i) L0.agg[i] = S.order[i] / agg0;
    L0.n_agg = (P + agg0 - 1) / agg0;
    AmgLevel cur = std::move(L0);
    for (;;) {
        if (cur.n <= kCoarsestMax) { S.A_last = cur.A; S.diag_last = find_diag(cur.A); break; }
        BlockCsr A_next; std::vector<double> xy_next;
        coarsen(cur, xy, A_next, xy_next);
        const int na = cur.n_agg;
        S.levels.push_back(std::move(cur));
        cur = AmgLevel();
        cur.n = na; cur.A = std::move(A_next);
        cur.agg.resize(na);
        for (int a = 0; a < na; ++a) cur.agg[a] = a / aggc;     // aggregates are numbered along the trajectory
        cur.n_agg = (na + aggc - 1) / aggc;
        xy = std::move(xy_next);
    }
    out = std::move(S);
    return std::string();
}

}  // namespace tsgo
