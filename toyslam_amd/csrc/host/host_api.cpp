// host_api.cpp — host-only C-ABI entry points that need no GPU (see include/tsgo.h).
#include "../../../include/tsgo.h"
#include "errors.h"
#include "problem.h"
#include "amg.h"
#include <algorithm>
#include <chrono>
#include <cstring>

extern "C" void tsgo_default_config(tsgo_config* c) {
    if (!c) return;
    c->device = 0; c->precision = 64; c->pcg_rel_tol = 1e-10; c->pcg_max_iters = 20000;
    c->lanes_per_pose = 0; c->lanes_per_lm = 0; c->use_graphs = 2; c->rank = 0; c->world = 1; c->verbose = 0; c->preconditioner = 1; c->xcd_map = 1; c->warm_start = 6; c->odom_jacobian = 0; c->reuse_structure = 1; c->rules = 0; c->lr = 0.2; c->cycle_level0 = 0; c->cycle_storage = 16; c->warm_requests = 0;
}

extern "C" int tsgo_layout_probe(const tsgo_graph* g, int32_t rank, int32_t world, int32_t lanes_per_pose,
                                 int32_t lanes_per_lm, tsgo_layout_info* out) {
    if (!g || !out) return tsgo::set_error(-1, "tsgo_layout_probe: null argument");
    tsgo::Problem pr;
    tsgo::BuildOptions bo; bo.rank = rank; bo.world = world; bo.lanes_per_pose = lanes_per_pose; bo.lanes_per_lm = lanes_per_lm;
    const std::string err = tsgo::build_problem(*g, bo, pr);
    if (!err.empty()) return tsgo::set_error(-2, err);
    out->n_pose = pr.P; out->n_lm_local = pr.L; out->n_lm_total = pr.L_total; out->n_lm_edges_local = pr.n_lm_edges;
    int64_t od = 0; for (uint32_t e : pr.odom.edge) od += e != tsgo::kNoEdge;
    out->n_odom_slots = od;
    out->rows_by_pose = (int64_t)pr.by_pose.rows; out->rows_by_lm = (int64_t)pr.by_lm.rows; out->rows_odom = (int64_t)pr.odom.rows;
    out->lanes_per_pose = pr.by_pose.G; out->lanes_per_lm = pr.by_lm.G;
    out->lm_first = pr.lm_first; out->lm_last = pr.lm_last; out->pose_first = pr.pose_first; out->pose_last = pr.pose_last;
    return 0;
}

namespace {
struct Fnv {
    uint64_t h = 1469598103934665603ull;
    void bytes(const void* p, size_t n) { const unsigned char* b = (const unsigned char*)p; for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; } }
    template <typename V> void vec(const V& v) { const uint64_t n = v.size(); bytes(&n, sizeof(n)); if (n) bytes(v.data(), n * sizeof(v[0])); }
    void table(const tsgo::SellTable& t) { vec(t.row_off); vec(t.idx); vec(t.edge); }
    void csr(const tsgo::BlockCsr& m) { vec(m.ptr); vec(m.col); }
    void pairs(const tsgo::PairList& p) { vec(p.ptr); vec(p.x); vec(p.y); }
};
}  // namespace

static int amg_probe(const tsgo_graph* g, int rank, int world, tsgo_amg_info* out, int64_t* odom_out) {
    std::memset(out, 0, sizeof(*out));
    tsgo::Problem pr; tsgo::BuildOptions bo; bo.rank = rank; bo.world = world;
    auto t0 = std::chrono::steady_clock::now();
    std::string err = tsgo::build_problem(*g, bo, pr);
    if (!err.empty()) return tsgo::set_error(-2, err);
    auto t1 = std::chrono::steady_clock::now();
    tsgo::AmgSym amg;
    err = world > 1 ? tsgo::build_amg_sharded(*g, pr, amg) : tsgo::build_amg(pr, amg);
    if (!err.empty()) return tsgo::set_error(-2, err);
    auto t2 = std::chrono::steady_clock::now();
    out->ms_layout = std::chrono::duration<double, std::milli>(t1 - t0).count();
    out->ms_symbolic = std::chrono::duration<double, std::milli>(t2 - t1).count();
    int n = 0;
    for (const auto& L : amg.levels) if (n < 8) {
        out->rows[n] = L.n; out->blocks[n] = L.A.nnz(); out->p_blocks[n] = L.P.nnz();
        std::vector<int> cnt(L.n_agg, 0);
        for (int a : L.agg) ++cnt[a];
        out->agg_min[n] = cnt.empty() ? 0 : *std::min_element(cnt.begin(), cnt.end());
        out->agg_max[n] = cnt.empty() ? 0 : *std::max_element(cnt.begin(), cnt.end());
        ++n;
    }
    if (n < 8) { out->rows[n] = amg.A_last.n_rows; out->blocks[n] = amg.A_last.nnz(); ++n; }
    out->n_levels = n;
    out->schur_contribs = (int64_t)amg.schur.slot_i.size();
    if (odom_out) *odom_out = (int64_t)amg.schur.od_slot.size();
    Fnv f;
    f.table(pr.by_pose); f.table(pr.by_lm); f.table(pr.odom); f.vec(pr.pose_vertex); f.vec(pr.lm_vertex); f.vec(pr.gauge_p); f.vec(pr.gauge_l);
    f.vec(amg.order); f.vec(amg.schur.ptr); f.vec(amg.schur.slot_i); f.vec(amg.schur.slot_k); f.vec(amg.schur.od_ptr); f.vec(amg.schur.od_slot);
    for (const auto& L : amg.levels) {
        f.csr(L.A); f.vec(L.diag); f.vec(L.agg); f.vec(L.rel); f.vec(L.rig); f.csr(L.P); f.vec(L.p_self); f.pairs(L.p_src); f.csr(L.R); f.vec(L.r_to_p);
        f.csr(L.T); f.pairs(L.t_src); f.pairs(L.a_src); f.vec(L.a_mirror);
    }
    f.csr(amg.A_last); f.vec(amg.diag_last);
    out->checksum = f.h;
    return 0;
}

extern "C" int tsgo_amg_probe(const tsgo_graph* g, tsgo_amg_info* out) {
    if (!g || !out) return tsgo::set_error(-1, "tsgo_amg_probe: null argument");
    return amg_probe(g, 0, 1, out, nullptr);
}

extern "C" int tsgo_amg_probe_shard(const tsgo_graph* g, int32_t rank, int32_t world, tsgo_amg_info* out, int64_t* odom_contribs_out) {
    if (!g || !out || world < 1 || rank < 0 || rank >= world) return tsgo::set_error(-1, "tsgo_amg_probe_shard: bad argument");
    return amg_probe(g, rank, world, out, odom_contribs_out);
}
