// synth.cpp — seeded synthetic 2-D SLAM graphs for BASELINE.json configs 2-5.
//
// The reference has no generator beyond its 150-step simulator; this one is modelled on it:
//   * odometry noise and information: python/slam_main.py:33-51,137-142 (the sim passes VARIANCES as
//     standard deviations: sigma_xy = 0.25, sigma_theta = 0.01536 rad; ODOM_INF = diag(4, 4, 65.12));
//   * lidar noise on the local x, y of each observation, sigma = 0.0225, LIDAR_INF = diag(44.4, 44.4):
//     python/slam/slam_helper.py:8-14, slam_main.py:42-50;
//   * landmark initialisation = first observation projected from the (noisy) pose estimate:
//     slam_helper.py:12-13;
//   * ids: poses 0..P-1, landmarks P.. in first-seen order, ODOM edges then LM edges, vertex 0 fixed:
//     slam_main.py:157-187.
// What differs (documented in DESIGN.md): the trajectory is a reflected random-curvature walk in a
// square sized so that landmarks are re-observed on revisits; the initial pose guess is ground truth
// plus a mean-reverting (Ornstein-Uhlenbeck) drift instead of open-loop dead reckoning, which at 10^5
// steps would leave every pose kilometres from the truth; every number is rounded to f32 because
// that is all the wire carries (python/remote/graph_to_bytes.py:6-7).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <random>
#include <vector>

#include "../../../include/tsgo.h"
#include "errors.h"

struct tsgo_synth {
    std::vector<uint32_t> v_id, v_type, e_type, e_ids, fixed;
    std::vector<double> v_pos, e_meas, e_inf, truth;
};

namespace {
inline double f32r(double v) { return (double)(float)v; }
const double kPi = 3.14159265358979323846;
inline double wrap(double a) { return std::atan2(std::sin(a), std::cos(a)); }
}  // namespace

extern "C" int tsgo_synth_create(const tsgo_synth_config* cfg, tsgo_synth** out) {
    if (!cfg || !out) return tsgo::set_error(-1, "tsgo_synth_create: null argument");
    const int64_t P = cfg->n_poses;
    const int k = cfg->lm_per_pose;
    if (P < 2 || P > 50000000 || k < 0 || k > 64) return tsgo::set_error(-1, "tsgo_synth_create: bad sizes");
    const double obs = cfg->lm_obs_target > 0 ? cfg->lm_obs_target : 5.0;
    std::mt19937_64 rng(cfg->seed);
    std::normal_distribution<double> N01(0.0, 1.0);
    std::uniform_real_distribution<double> U01(0.0, 1.0);

    const double sig_xy = 0.25, sig_th = std::pow(7.1 * kPi / 180.0, 2), sig_lidar = 0.15 * 0.15;
    const double inf_xy = 1.0 / 0.25, inf_th = 1.0 / sig_th, inf_lidar = 1.0 / sig_lidar;

    // landmark density such that the k-nearest corridor swept by the walk holds ~P*k/obs landmarks
    const double rho = k > 0 ? kPi * k / (4.0 * obs * obs) / 0.40 : 0.0;
    const double rk = k > 0 ? std::sqrt(k / (kPi * rho)) : 1.0;
    const double R = 3.0 * rk;
    const double side = std::max(20.0, std::sqrt((double)P * 2.0 * rk));

    // ---- ground-truth trajectory -------------------------------------------------------------------
    std::vector<double> gx(P), gy(P), gth(P);
    gx[0] = 0.5 * side; gy[0] = 0.5 * side; gth[0] = 2 * kPi * U01(rng) - kPi;
    for (int64_t i = 1; i < P; ++i) {
        double th = gth[i - 1] + (3.0 * kPi / 180.0) * N01(rng);
        double nx = gx[i - 1] + std::cos(th), ny = gy[i - 1] + std::sin(th);
        if (nx < 0 || nx > side) { th = kPi - th; nx = gx[i - 1] + std::cos(th); }
        if (ny < 0 || ny > side) { th = -th; ny = gy[i - 1] + std::sin(th); }
        nx = std::min(std::max(nx, 0.0), side); ny = std::min(std::max(ny, 0.0), side);
        gx[i] = nx; gy[i] = ny; gth[i] = wrap(th);
    }
    // ---- initial guess: truth + Ornstein-Uhlenbeck drift (pose 0 exact: it is the gauge) ------------
    std::vector<double> ex(P), ey(P), eth(P);
    {
        double dx = 0, dy = 0, dt = 0; const double keep = 0.99;
        ex[0] = gx[0]; ey[0] = gy[0]; eth[0] = gth[0];
        for (int64_t i = 1; i < P; ++i) {
            dx = keep * dx + sig_xy * N01(rng); dy = keep * dy + sig_xy * N01(rng); dt = keep * dt + sig_th * N01(rng);
            ex[i] = gx[i] + dx; ey[i] = gy[i] + dy; eth[i] = wrap(gth[i] + dt);
        }
    }
    // ---- landmarks on a uniform grid hash ------------------------------------------------------------
    const int64_t Lgen = k > 0 ? (int64_t)(rho * side * side) : 0;
    std::vector<double> lx(Lgen), ly(Lgen);
    const int cells = std::max(1, (int)(side / R));
    const double cw = side / cells;
    std::vector<int> cell_count((size_t)cells * cells + 1, 0);
    std::vector<int> lcell(Lgen);
    for (int64_t j = 0; j < Lgen; ++j) {
        lx[j] = side * U01(rng); ly[j] = side * U01(rng);
        const int cx = std::min(cells - 1, (int)(lx[j] / cw)), cy = std::min(cells - 1, (int)(ly[j] / cw));
        lcell[j] = cy * cells + cx; ++cell_count[lcell[j] + 1];
    }
    for (size_t c = 0; c < (size_t)cells * cells; ++c) cell_count[c + 1] += cell_count[c];
    std::vector<int> cell_items(Lgen), cursor(cell_count.begin(), cell_count.end() - 1);
    for (int64_t j = 0; j < Lgen; ++j) cell_items[cursor[lcell[j]]++] = (int)j;

    auto* s = new tsgo_synth();
    // ---- vertices: poses ------------------------------------------------------------------------------
    s->v_id.reserve(P); s->v_type.reserve(P);
    for (int64_t i = 0; i < P; ++i) {
        s->v_id.push_back((uint32_t)i); s->v_type.push_back(0);
        s->v_pos.push_back(f32r(ex[i])); s->v_pos.push_back(f32r(ey[i])); s->v_pos.push_back(f32r(eth[i]));
        s->truth.push_back(gx[i]); s->truth.push_back(gy[i]); s->truth.push_back(gth[i]);
    }
    // ---- ODOM edges i -> i+1, then loop closures ------------------------------------------------------
    auto add_odom = [&](int64_t a, int64_t b) {
        const double ca = std::cos(gth[a]), sa = std::sin(gth[a]);
        const double dx = gx[b] - gx[a], dy = gy[b] - gy[a];
        const double mx = ca * dx + sa * dy + sig_xy * N01(rng);
        const double my = -sa * dx + ca * dy + sig_xy * N01(rng);
        const double mt = wrap(gth[b] - gth[a]) + sig_th * N01(rng);
        const double c = std::cos(mt), sn = std::sin(mt);
        const double m[9] = {c, -sn, mx, sn, c, my, 0, 0, 1};
        s->e_type.push_back(0); s->e_ids.push_back((uint32_t)a); s->e_ids.push_back((uint32_t)b);
        for (double v : m) s->e_meas.push_back(f32r(v));
        s->e_inf.push_back(f32r(inf_xy)); s->e_inf.push_back(f32r(inf_xy)); s->e_inf.push_back(f32r(inf_th));
    };
    for (int64_t i = 0; i + 1 < P; ++i) add_odom(i, i + 1);
    if (cfg->loop_closures > 0) {
        // poses hashed on a 2-unit grid; a closure joins poses closer than 2 with index gap > 1000
        const double cw2 = 2.0; const int c2 = std::max(1, (int)(side / cw2) + 1);
        std::vector<std::vector<int>> bucket((size_t)c2 * c2);
        int made = 0;
        for (int64_t i = 0; i < P && made < cfg->loop_closures; ++i) {
            const int cx = std::min(c2 - 1, (int)(gx[i] / cw2)), cy = std::min(c2 - 1, (int)(gy[i] / cw2));
            int found = -1;
            for (int yy = std::max(0, cy - 1); yy <= std::min(c2 - 1, cy + 1) && found < 0; ++yy)
                for (int xx = std::max(0, cx - 1); xx <= std::min(c2 - 1, cx + 1) && found < 0; ++xx)
                    for (int j : bucket[(size_t)yy * c2 + xx])
                        if (i - j > 1000 && std::hypot(gx[i] - gx[j], gy[i] - gy[j]) < 2.0) { found = j; break; }
            if (found >= 0 && (i % 3) == 0) { add_odom(found, i); ++made; }
            bucket[(size_t)cy * c2 + cx].push_back((int)i);
        }
    }
    // ---- LM edges: each pose observes its k nearest landmarks within R --------------------------------
    std::vector<int> lm_new_id(Lgen, -1);
    std::vector<double> lm_init;      // 2 per observed landmark (first observation from the noisy pose)
    std::vector<double> lm_truth;
    std::vector<std::pair<double, int>> cand;
    int64_t L = 0;
    for (int64_t i = 0; i < P && k > 0; ++i) {
        cand.clear();
        const int cx = std::min(cells - 1, (int)(gx[i] / cw)), cy = std::min(cells - 1, (int)(gy[i] / cw));
        for (int yy = std::max(0, cy - 1); yy <= std::min(cells - 1, cy + 1); ++yy)
            for (int xx = std::max(0, cx - 1); xx <= std::min(cells - 1, cx + 1); ++xx) {
                const int c = yy * cells + xx;
                for (int q = cell_count[c]; q < cell_count[c + 1]; ++q) {
                    const int j = cell_items[q];
                    const double d2 = (lx[j] - gx[i]) * (lx[j] - gx[i]) + (ly[j] - gy[i]) * (ly[j] - gy[i]);
                    if (d2 <= R * R) cand.emplace_back(d2, j);
                }
            }
        const size_t take = std::min<size_t>((size_t)k, cand.size());
        std::partial_sort(cand.begin(), cand.begin() + take, cand.end());
        const double c = std::cos(gth[i]), sn = std::sin(gth[i]);
        for (size_t q = 0; q < take; ++q) {
            const int j = cand[q].second;
            const double dx = lx[j] - gx[i], dy = ly[j] - gy[i];
            const double mx = c * dx + sn * dy + sig_lidar * N01(rng);
            const double my = -sn * dx + c * dy + sig_lidar * N01(rng);
            const double range = std::hypot(mx, my), bearing = std::atan2(my, mx);
            if (lm_new_id[j] < 0) {
                lm_new_id[j] = (int)L++;
                const double ce = std::cos(eth[i]), se = std::sin(eth[i]);
                lm_init.push_back(ex[i] + ce * mx - se * my); lm_init.push_back(ey[i] + se * mx + ce * my);
                lm_truth.push_back(lx[j]); lm_truth.push_back(ly[j]);
            }
            s->e_type.push_back(1); s->e_ids.push_back((uint32_t)i); s->e_ids.push_back((uint32_t)(P + lm_new_id[j]));
            const double m[9] = {range, bearing, 0, 0, 0, 0, 0, 0, 0};
            for (double v : m) s->e_meas.push_back(f32r(v));
            s->e_inf.push_back(f32r(inf_lidar)); s->e_inf.push_back(f32r(inf_lidar)); s->e_inf.push_back(0.0);
        }
    }
    for (int64_t j = 0; j < L; ++j) {
        s->v_id.push_back((uint32_t)(P + j)); s->v_type.push_back(1);
        s->v_pos.push_back(f32r(lm_init[2 * j])); s->v_pos.push_back(f32r(lm_init[2 * j + 1])); s->v_pos.push_back(0.0);
        s->truth.push_back(lm_truth[2 * j]); s->truth.push_back(lm_truth[2 * j + 1]); s->truth.push_back(0.0);
    }
    s->fixed.push_back(0);
    *out = s;
    return 0;
}

extern "C" void tsgo_synth_view(const tsgo_synth* s, tsgo_graph* g) {
    g->n_vertices = (int32_t)s->v_id.size(); g->v_id = s->v_id.data(); g->v_type = s->v_type.data(); g->v_pos = s->v_pos.data();
    g->n_edges = (int32_t)s->e_type.size(); g->e_type = s->e_type.data(); g->e_ids = s->e_ids.data();
    g->e_meas = s->e_meas.data(); g->e_inf = s->e_inf.data();
    g->n_fixed = (int32_t)s->fixed.size(); g->fixed = s->fixed.data();
}
extern "C" const double* tsgo_synth_truth(const tsgo_synth* s) { return s->truth.data(); }
extern "C" void tsgo_synth_free(tsgo_synth* s) { delete s; }
