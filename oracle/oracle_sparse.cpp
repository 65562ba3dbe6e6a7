// oracle_sparse.cpp — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// Scalar CPU twin of the sparse Gauss-Newton path the HIP library runs (same layout, same implicit
// Schur-complement PCG, same stop rules), used
//   * by tests/ as the checker at sizes where the dense restatement (oracle_dense.cpp) cannot run:
//     it is itself validated against the dense restatement on the golden config-1 graph and on small
//     synthetic graphs (tests/test_sparse_twin.py), and the HIP path is then compared with it;
//   * by the world_size-2 gloo tests: every place the HIP path calls an RCCL all-reduce, this twin
//     calls a user hook, so the sharded algorithm can run across CPU processes;
//   * by bench.py as the `cpu_baseline` ("port": same math, NOT the reference's dense algorithm, which
//     needs O(n^2) memory and cannot run configs 2-5: SURVEY.md section 0, finding 1).
// It reuses the product's layout builder (toyslam_amd/csrc/host/problem.cpp) and per-edge arithmetic
// (toyslam_amd/csrc/tsgo_math.h); nothing in the product links this file.
//
// GN loop rules: remote/optimizer/OptimizerCpu.h:80-180.  Per-edge math: tsgo_math.h (which cites
// EdgeSe2Point2d.h / EdgeSe2.h).  Vertex update: VertexSe2.h:16-27, Vertex2d.h:16-19.
#include <omp.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <deque>
#include <vector>

#include "amg_twin.h"
#include "host/amg.h"
#include "host/problem.h"
#include "tsgo_math.h"

static int g_cycle_level0 = 0;     // oracle_set_cycle_level0: 0 implicit Schur products inside the multigrid cycle, 1 the explicit level-0 matrix
extern "C" int oracle_get_odom_jacobian(void);      // oracle_dense.cpp: 0 = the reference's constants, 1 = analytic (extension)

namespace {

using tsgo::kNoEdge; using tsgo::kWave; using tsgo::kDirBit; using tsgo::kVlmBit; using tsgo::kPoseMask;
typedef void (*allreduce_fn)(double* buf, int64_t n, void* ctx);

struct Twin {
    tsgo::Problem pr;
    allreduce_fn hook = nullptr; void* hook_ctx = nullptr;
    int P = 0, L = 0;
    std::vector<double> ps, th, ls;                 // pose (x,y,c,s), theta, landmark (x,y)
    std::vector<double> pa, la;                     // 4 planes each: a0, a1, ppx, ppy (pose-major / lm-major)
    std::vector<double> oa;                         // 3 planes (odom)
    std::vector<double> dlinv, u, t;                // per landmark: 3, 2, 2
    std::vector<double> part;                       // per pose 18: Dp(6) g(3) Sd(6) Wu(3); + 1 chi2 at the end
    std::vector<double> dp, minv, r, z, p, q, x, s; // per pose
    double chi2 = 0;
    std::vector<double> gscratch;
    bool use_amg = false;
    bool analytic = false;                          // analytic ODOM Jacobians (extension; tsgo_math.h: odom_blocks)
    bool general = false;                           // eight planes per pose-pose slot: analytic ODOM Jacobians or virtual landmark measurements in the graph
    double lambda = 0; bool zero_fixed = false;     // the Python optimizer's rules (graph_optimizer.py:42,150), as Engine::launch_lin; 0 / false: cpu eigen
    double step = tsgo::kStepScale;
    int n_lin = 0;
    int hier_age = -1, iters_fresh = 0, iters_last = 0;   // hierarchy refresh policy, as Engine::do_linearize (tsgo_hip.hip)
    tsgo::AmgSym amg;
    amgtwin::Hierarchy hier;
    std::vector<double> res0, s0;

    void allreduce(double* b, int64_t n) { if (hook && pr.world > 1) hook(b, n, hook_ctx); }

    std::string init(const tsgo_graph& g, int rank, int world) {
        tsgo::BuildOptions bo; bo.rank = rank; bo.world = world; bo.lanes_per_pose = 1; bo.lanes_per_lm = 1;
        std::string e = tsgo::build_problem(g, bo, pr);
        if (!e.empty()) return e;
        P = pr.P; L = pr.L;
        ps.resize((size_t)P * 4); th.resize(P); ls = pr.lm_xy;
        for (int i = 0; i < P; ++i) {
            th[i] = pr.pose_xyt[3 * (size_t)i + 2];
            ps[4 * (size_t)i] = pr.pose_xyt[3 * (size_t)i]; ps[4 * (size_t)i + 1] = pr.pose_xyt[3 * (size_t)i + 1];
            ps[4 * (size_t)i + 2] = std::cos(th[i]); ps[4 * (size_t)i + 3] = std::sin(th[i]);
        }
        analytic = oracle_get_odom_jacobian() != 0;
        pr.odom_analytic = analytic;
        general = analytic || pr.has_vlm;      // pose-pose slots in general form (tsgo_math.h), as Engine::oj()
        pa.assign(4 * pr.by_pose.slots(), 0); la.assign(4 * pr.by_lm.slots(), 0); oa.assign((general ? (int)tsgo::PP_PLANES : 3) * pr.odom.slots(), 0);
        dlinv.assign((size_t)L * 3, 0); u.assign((size_t)L * 2, 0); t.assign((size_t)L * 2, 0);
        part.assign((size_t)P * 18 + 1, 0);
        dp.assign((size_t)P * 6, 0); minv.assign((size_t)P * 6, 0);
        for (auto* v : {&r, &z, &p, &q, &x, &s}) v->assign((size_t)P * 3, 0);
        return std::string();
    }

    // world > 1: replicated hierarchy, level-0 blocks summed from per-shard partials (host/amg.h: build_amg_sharded)
    std::string enable_amg(const tsgo_graph& g) {
        std::string e = pr.world > 1 ? tsgo::build_amg_sharded(g, pr, amg) : tsgo::build_amg(pr, amg);
        if (!e.empty()) return e;
        hier.alloc(amg);
        res0.assign((size_t)P * 3, 0); s0.assign((size_t)P * 3, 0);
        use_amg = true;
        return std::string();
    }

    // level 0 of the hierarchy: explicit Schur complement blocks (twin of k_schur_blocks)
    void build_schur_blocks() {
        const tsgo::AmgLevel& L = amg.levels[0];
        std::vector<double>& A = hier.A[0];
        const tsgo::SellTable& tb = pr.by_pose; const size_t S = tb.slots();
        const size_t SO = pr.odom.slots();
        #pragma omp parallel for schedule(static)
        for (int i = 0; i < P; ++i) {
            const double ci = ps[4 * (size_t)i + 2], si = ps[4 * (size_t)i + 3];
            for (int b = L.A.ptr[i]; b < L.A.ptr[i + 1]; ++b) {
                const int k = L.A.col[b];
                double* o = &A[(size_t)b * 9];
                if (k == i) {
                    if (pr.rank != 0) { for (int m = 0; m < 9; ++m) o[m] = 0; continue; }     // the all-reduced diagonal: one rank contributes it
                    const double* o18 = &part[(size_t)i * 18];
                    const double m[6] = {o18[0] - o18[9], o18[1] - o18[10], o18[2] - o18[11], o18[3] - o18[12], o18[4] - o18[13], o18[5] - o18[14]};
                    o[0] = m[0]; o[1] = m[1]; o[2] = m[2]; o[3] = m[1]; o[4] = m[3]; o[5] = m[4]; o[6] = m[2]; o[7] = m[4]; o[8] = m[5];
                    continue;
                }
                const double ck = ps[4 * (size_t)k + 2], sk = ps[4 * (size_t)k + 3];
                double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
                for (int q = amg.schur.ptr[b]; q < amg.schur.ptr[b + 1]; ++q) {
                    const size_t e1 = amg.schur.slot_i[q], e2 = amg.schur.slot_k[q];
                    const uint32_t l = tb.idx[e1];
                    const double a0i = pa[e1], a1i = pa[S + e1], vi0 = pa[3 * S + e1], vi1 = -pa[2 * S + e1];
                    const double a0k = pa[e2], a1k = pa[S + e2], vk0 = pa[3 * S + e2], vk1 = -pa[2 * S + e2];
                    const double nxx = dlinv[3 * (size_t)l], nxy = dlinv[3 * (size_t)l + 1], nyy = dlinv[3 * (size_t)l + 2];
                    // G = Ri^T N Rk
                    const double m00 = nxx * ck + nxy * sk, m01 = -nxx * sk + nxy * ck, m10 = nxy * ck + nyy * sk, m11 = -nxy * sk + nyy * ck;
                    const double g00 = ci * m00 + si * m10, g01 = ci * m01 + si * m11, g10 = -si * m00 + ci * m10, g11 = -si * m01 + ci * m11;
                    const double q00 = a0i * g00 * a0k, q01 = a0i * g01 * a1k, q10 = a1i * g10 * a0k, q11 = a1i * g11 * a1k;
                    // Ri Q Rk^T
                    const double u00 = ci * q00 - si * q10, u01 = ci * q01 - si * q11, u10 = si * q00 + ci * q10, u11 = si * q01 + ci * q11;   // Ri Q
                    acc[0] += u00 * ck - u01 * sk; acc[1] += u00 * sk + u01 * ck; acc[3] += u10 * ck - u11 * sk; acc[4] += u10 * sk + u11 * ck;
                    const double qv0 = q00 * vk0 + q01 * vk1, qv1 = q10 * vk0 + q11 * vk1;      // Q vk
                    acc[2] -= ci * qv0 - si * qv1; acc[5] -= si * qv0 + ci * qv1;
                    const double vq0 = vi0 * q00 + vi1 * q10, vq1 = vi0 * q01 + vi1 * q11;      // vi^T Q
                    acc[6] -= vq0 * ck - vq1 * sk; acc[7] -= vq0 * sk + vq1 * ck;
                    acc[8] += vi0 * qv0 + vi1 * qv1;
                }
                for (int m = 0; m < 9; ++m) o[m] = -acc[m];
                for (int q = amg.schur.od_ptr[b]; q < amg.schur.od_ptr[b + 1]; ++q) {
                    const size_t e = amg.schur.od_slot[q];
                    if (general) {      // the slot's own row block [[-K, c], [r^T, -kappa]] (tsgo_math.h; k_schur_blocks)
                        double h[tsgo::PP_PLANES];
                        for (int m = 0; m < tsgo::PP_PLANES; ++m) h[m] = oa[(size_t)m * SO + e];
                        o[0] -= h[tsgo::PP_K00]; o[1] -= h[tsgo::PP_K01]; o[3] -= h[tsgo::PP_K01]; o[4] -= h[tsgo::PP_K11];
                        o[2] += h[tsgo::PP_C0]; o[5] += h[tsgo::PP_C1]; o[6] += h[tsgo::PP_R0]; o[7] += h[tsgo::PP_R1]; o[8] -= h[tsgo::PP_KAPPA];
                    } else { o[0] -= oa[e]; o[4] -= oa[SO + e]; o[8] -= oa[2 * SO + e]; }
                }
            }
        }
        allreduce(A.data(), (int64_t)((size_t)L.A.nnz() * 9));     // tsgo_hip.hip: launch_amg_setup
    }

    // z = M^-1 r by one V(1,1) cycle (level 0 uses the implicit Schur product)
    void amg_apply() {
        const double w0 = hier.omega.empty() ? 1.0 : hier.omega[0];
        for (int i = 0; i < P; ++i) {
            tsgo::sym3_mul(&minv[6 * (size_t)i], r[3 * (size_t)i], r[3 * (size_t)i + 1], r[3 * (size_t)i + 2], z[3 * (size_t)i], z[3 * (size_t)i + 1], z[3 * (size_t)i + 2]);
            if (!amg.levels.empty()) for (int k = 0; k < 3; ++k) z[3 * (size_t)i + k] *= w0;
        }
        if (amg.levels.empty()) return;
        // cycle_level0 = 1 (oracle_set_cycle_level0; the engine's tsgo_config.cycle_level0, tsgo_hip.hip: launch_cycle_product): inside
        // the cycle the product with the replicated EXPLICIT level-0 matrix — as old as the hierarchy — and no all-reduce
        const bool explicit0 = g_cycle_level0 != 0;
        auto cycle_product = [&] {
            if (explicit0) { amgtwin::Hierarchy::spmv(amg.levels[0].A, hier.A[0], z, s0); return; }
            schur_lm(z); schur_pose(z, s0); allreduce(s0.data(), (int64_t)s0.size());
        };
        static const bool additive0 = getenv("TSGO_TWIN_ADDITIVE0") != nullptr;      // research: M^-1 = w D^-1 + P Mc^-1 P^T, no level-0 product in the cycle
        if (additive0) {
            hier.restrict_to(0, r, hier.r[1]);
            hier.cycle(1);
            hier.prolong_add(0, hier.z[1], z);
            return;
        }
        cycle_product();
        for (size_t k = 0; k < s0.size(); ++k) res0[k] = r[k] - s0[k];
        hier.restrict_to(0, res0, hier.r[1]);
        hier.cycle(1);
        hier.prolong_add(0, hier.z[1], z);
        cycle_product();
        for (int i = 0; i < P; ++i) {
            double d0, d1, d2;
            tsgo::sym3_mul(&minv[6 * (size_t)i], r[3 * (size_t)i] - s0[3 * (size_t)i], r[3 * (size_t)i + 1] - s0[3 * (size_t)i + 1], r[3 * (size_t)i + 2] - s0[3 * (size_t)i + 2], d0, d1, d2);
            z[3 * (size_t)i] += w0 * d0; z[3 * (size_t)i + 1] += w0 * d1; z[3 * (size_t)i + 2] += w0 * d2;
        }
    }

    template <typename F> static void for_slots(const tsgo::SellTable& tb, int v, F f) {
        const int vps = kWave / tb.G, sl = v / vps, base = (v % vps) * tb.G;
        for (uint32_t row = tb.row_off[sl]; row < tb.row_off[sl + 1]; ++row)
            for (int sub = 0; sub < tb.G; ++sub) {
                const size_t slot = (size_t)row * kWave + base + sub;
                if (tb.edge[slot] != kNoEdge) f(slot);
            }
    }

    // K1: landmark side of the linearisation (B^T W B, B^T W e), block inverse, u = Dl^-1 g_l
    void lin_lm() {
        const tsgo::SellTable& tb = pr.by_lm; const size_t S = tb.slots();
        #pragma omp parallel for schedule(static)
        for (int l = 0; l < L; ++l) {
            double dxx = pr.gauge_l[l] + lambda, dxy = 0, dyy = pr.gauge_l[l] + lambda, g0 = 0, g1 = 0;
            const double lx = ls[2 * (size_t)l], ly = ls[2 * (size_t)l + 1];
            for_slots(tb, l, [&](size_t k) {
                const double* q4 = &ps[4 * (size_t)tb.idx[k]];
                const double c = q4[2], sn = q4[3];
                auto o = tsgo::lm_linearize<double>(q4[0], q4[1], c, sn, lx, ly, tb.plane(tsgo::LM_ZX)[k], tb.plane(tsgo::LM_ZY)[k],
                                                   tb.plane(tsgo::LM_W0)[k], tb.plane(tsgo::LM_W1)[k]);
                la[k] = o.a0; la[S + k] = o.a1; la[2 * S + k] = o.ppx; la[3 * S + k] = o.ppy;
                dxx += o.a0 * c * c + o.a1 * sn * sn; dxy += (o.a0 - o.a1) * c * sn; dyy += o.a0 * sn * sn + o.a1 * c * c;
                const double f0 = o.a0 * o.e0, f1 = o.a1 * o.e1;          // g_l = -B^T W e = -R (f0, f1)
                g0 -= c * f0 - sn * f1; g1 -= sn * f0 + c * f1;
            });
            if (zero_fixed && pr.gauge_l[l] > 0) { g0 = 0; g1 = 0; }
            double ixx, ixy, iyy; tsgo::inv_sym2(dxx, dxy, dyy, ixx, ixy, iyy);
            dlinv[3 * (size_t)l] = ixx; dlinv[3 * (size_t)l + 1] = ixy; dlinv[3 * (size_t)l + 2] = iyy;
            u[2 * (size_t)l] = ixx * g0 + ixy * g1; u[2 * (size_t)l + 1] = ixy * g0 + iyy * g1;
        }
    }

    // K2: pose side (A^T W A, A^T W e, Schur diagonal W Dl^-1 W^T, W u), ODOM rows, chi^2
    void lin_pose() {
        const tsgo::SellTable& tb = pr.by_pose; const size_t S = tb.slots();
        const tsgo::SellTable& od = pr.odom; const size_t SO = od.slots();
        double chi = 0;
        #pragma omp parallel for schedule(static) reduction(+ : chi)
        for (int i = 0; i < P; ++i) {
            const double x0 = ps[4 * (size_t)i], y0 = ps[4 * (size_t)i + 1], c = ps[4 * (size_t)i + 2], sn = ps[4 * (size_t)i + 3];
            double sA0 = 0, sA1 = 0, sAv0 = 0, sAv1 = 0, sVV = 0, ge0 = 0, ge1 = 0, get = 0;
            double K00 = 0, K01 = 0, K11 = 0, Kv0 = 0, Kv1 = 0, vKv = 0, wu0 = 0, wu1 = 0, wut = 0;
            for_slots(tb, i, [&](size_t k) {
                const uint32_t l = tb.idx[k];
                auto o = tsgo::lm_linearize<double>(x0, y0, c, sn, ls[2 * (size_t)l], ls[2 * (size_t)l + 1], tb.plane(tsgo::LM_ZX)[k],
                                                   tb.plane(tsgo::LM_ZY)[k], tb.plane(tsgo::LM_W0)[k], tb.plane(tsgo::LM_W1)[k]);
                pa[k] = o.a0; pa[S + k] = o.a1; pa[2 * S + k] = o.ppx; pa[3 * S + k] = o.ppy;
                chi += o.rho;
                const double v0 = o.ppy, v1 = -o.ppx;                      // A = [-R^T | v]
                sA0 += o.a0; sA1 += o.a1; sAv0 += o.a0 * v0; sAv1 += o.a1 * v1; sVV += o.a0 * v0 * v0 + o.a1 * v1 * v1;
                ge0 += o.a0 * o.e0; ge1 += o.a1 * o.e1; get += o.a0 * o.e0 * v0 + o.a1 * o.e1 * v1;
                // N~ = R^T Dl^-1 R in the pose frame
                const double nxx = dlinv[3 * (size_t)l], nxy = dlinv[3 * (size_t)l + 1], nyy = dlinv[3 * (size_t)l + 2];
                const double n00 = c * c * nxx + 2 * c * sn * nxy + sn * sn * nyy;
                const double n01 = c * sn * (nyy - nxx) + (c * c - sn * sn) * nxy;
                const double n11 = sn * sn * nxx - 2 * c * sn * nxy + c * c * nyy;
                const double k00 = o.a0 * n00 * o.a0, k01 = o.a0 * n01 * o.a1, k11 = o.a1 * n11 * o.a1;   // K = W~ N~ W~
                K00 += k00; K01 += k01; K11 += k11;
                const double kv0 = k00 * v0 + k01 * v1, kv1 = k01 * v0 + k11 * v1;
                Kv0 += kv0; Kv1 += kv1; vKv += v0 * kv0 + v1 * kv1;
                const double ux = u[2 * (size_t)l], uy = u[2 * (size_t)l + 1];
                const double t0 = c * ux + sn * uy, t1 = c * uy - sn * ux;                               // tau = R^T u
                wu0 += o.a0 * t0; wu1 += o.a1 * t1; wut += o.a0 * t0 * v0 + o.a1 * t1 * v1;
            });
            double* o18 = &part[(size_t)i * 18];
            // rotate the pose-frame sums to the world frame
            // Dp_tt = R diag(sA) R^T ; Dp_t,th = -R sAv ; Dp_th,th = sVV
            o18[0] = c * c * sA0 + sn * sn * sA1; o18[1] = c * sn * (sA0 - sA1); o18[3] = sn * sn * sA0 + c * c * sA1;
            o18[2] = -(c * sAv0 - sn * sAv1); o18[4] = -(sn * sAv0 + c * sAv1); o18[5] = sVV;
            // g = -A^T W e = [R (ge0, ge1) ; -get]
            o18[6] = c * ge0 - sn * ge1; o18[7] = sn * ge0 + c * ge1; o18[8] = -get;
            // Sd = W Dl^-1 W^T = [[R K R^T, -R Kv], [., vKv]]
            o18[9] = c * c * K00 - 2 * c * sn * K01 + sn * sn * K11;
            o18[10] = c * sn * (K00 - K11) + (c * c - sn * sn) * K01;
            o18[12] = sn * sn * K00 + 2 * c * sn * K01 + c * c * K11;
            o18[11] = -(c * Kv0 - sn * Kv1); o18[13] = -(sn * Kv0 + c * Kv1); o18[14] = vKv;
            // W u = [-R (wu0, wu1) ; wut]
            o18[15] = -(c * wu0 - sn * wu1); o18[16] = -(sn * wu0 + c * wu1); o18[17] = wut;
            // ODOM rows (both directions are listed; chi^2 counted at id1 only)
            for_slots(od, i, [&](size_t k) {
                const uint32_t raw = od.idx[k]; const bool second = raw & kDirBit; const uint32_t j = raw & kPoseMask;
                const double* me = &ps[4 * (size_t)i]; const double* ot = &ps[4 * (size_t)j];
                const double* a = second ? ot : me; const double* b = second ? me : ot;
                double mi[6], w[3];
                for (int m = 0; m < 6; ++m) mi[m] = od.plane(tsgo::OD_MI0 + m)[k];
                for (int m = 0; m < 3; ++m) w[m] = od.plane(tsgo::OD_W0 + m)[k];
                if (raw & kVlmBit) {      // virtual landmark measurement (as k_lin_pose<.., 1>): mi = (pox, poy, pnx, pny), w = (w0, w1)
                    const auto v = tsgo::vlm_linearize<double>(me[0], me[1], me[2], me[3], ot[0], ot[1], ot[2], ot[3], mi[0], mi[1], mi[2], mi[3], w[0], w[1]);
                    double h[tsgo::PP_PLANES]; tsgo::vlm_slot<double>(v, h);
                    for (int m = 0; m < tsgo::PP_PLANES; ++m) oa[(size_t)m * SO + k] = h[m];
                    o18[0] += v.om0; o18[3] += v.om1; o18[2] += v.om0 * v.u0; o18[4] += v.om1 * v.u1; o18[5] += v.om0 * v.u0 * v.u0 + v.om1 * v.u1 * v.u1;
                    o18[6] -= v.om0 * v.d0; o18[7] -= v.om1 * v.d1; o18[8] -= v.om0 * v.u0 * v.d0 + v.om1 * v.u1 * v.d1;
                    if (!second) chi += v.rho;
                    return;
                }
                auto o = tsgo::odom_linearize<double>(a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3], mi, w);
                if (analytic) {                                              // tsgo_math.h: odom_blocks (as k_lin_pose<.., OJ = 1>)
                    const auto ob = tsgo::odom_blocks<double>(o, a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3], mi);
                    double h[tsgo::PP_PLANES] = {ob.k00, ob.k01, ob.k11, second ? ob.g0 : 0.0, second ? ob.g1 : 0.0, second ? 0.0 : ob.g0, second ? 0.0 : ob.g1, ob.w};
                    for (int m = 0; m < tsgo::PP_PLANES; ++m) oa[(size_t)m * SO + k] = h[m];
                    o18[0] += ob.k00; o18[1] += ob.k01; o18[3] += ob.k11;
                    if (!second) { o18[2] -= ob.g0; o18[4] -= ob.g1; o18[5] += ob.s + ob.w; o18[6] += ob.h0; o18[7] += ob.h1; o18[8] += ob.kt - ob.ht; }
                    else { o18[5] += ob.w; o18[6] -= ob.h0; o18[7] -= ob.h1; o18[8] -= ob.kt; }
                } else {
                    if (general) {
                        const double h[tsgo::PP_PLANES] = {o.a[0], 0.0, o.a[1], 0.0, 0.0, 0.0, 0.0, o.a[2]};
                        for (int m = 0; m < tsgo::PP_PLANES; ++m) oa[(size_t)m * SO + k] = h[m];
                    } else for (int m = 0; m < 3; ++m) oa[m * SO + k] = o.a[m];
                    o18[0] += o.a[0]; o18[3] += o.a[1]; o18[5] += o.a[2];
                    const double sg = second ? -1.0 : 1.0;                       // b1 += W e, b2 -= W e
                    for (int m = 0; m < 3; ++m) o18[6 + m] += sg * o.a[m] * o.e[m];
                }
                if (!second) chi += o.rho;
            });
            if (zero_fixed && pr.gauge_p[i] > 0) { o18[6] = 0; o18[7] = 0; o18[8] = 0; }
        }
        part[(size_t)P * 18] = chi;
    }

    // K3: M = Dp + gauge - Sd, M^-1, reduced rhs, CG start vectors.  Returns gamma0 = r^T M^-1 r.
    double finalize() {
        chi2 = part[(size_t)P * 18];
        std::vector<double> gi(P, 0.0);     // summed serially below: every shard must get the same bits
        #pragma omp parallel for schedule(static)
        for (int i = 0; i < P; ++i) {
            const double* o18 = &part[(size_t)i * 18];
            double m[6];
            for (int k = 0; k < 6; ++k) { dp[6 * (size_t)i + k] = o18[k]; m[k] = o18[k] - o18[9 + k]; }
            // NOTE: the gauge term of poses owned by other shards arrives through the all-reduce
            tsgo::inv_sym3(m, &minv[6 * (size_t)i]);
            for (int k = 0; k < 3; ++k) { r[3 * (size_t)i + k] = o18[6 + k] - o18[15 + k]; x[3 * (size_t)i + k] = 0; p[3 * (size_t)i + k] = 0; q[3 * (size_t)i + k] = 0; }
            tsgo::sym3_mul(&minv[6 * (size_t)i], r[3 * (size_t)i], r[3 * (size_t)i + 1], r[3 * (size_t)i + 2], z[3 * (size_t)i], z[3 * (size_t)i + 1], z[3 * (size_t)i + 2]);
            for (int k = 0; k < 3; ++k) gi[i] += r[3 * (size_t)i + k] * z[3 * (size_t)i + k];
        }
        double gamma = 0;
        for (int i = 0; i < P; ++i) gamma += gi[i];
        return gamma;
    }

    // KA: t = Dl^-1 W^T v   (v per pose, 3)
    void schur_lm(const std::vector<double>& v) {
        const tsgo::SellTable& tb = pr.by_lm; const size_t S = tb.slots();
        #pragma omp parallel for schedule(static)
        for (int l = 0; l < L; ++l) {
            double a0s = 0, a1s = 0;
            for_slots(tb, l, [&](size_t k) {
                const uint32_t i = tb.idx[k];
                const double c = ps[4 * (size_t)i + 2], sn = ps[4 * (size_t)i + 3];
                const double v0 = v[3 * (size_t)i], v1 = v[3 * (size_t)i + 1], v2 = v[3 * (size_t)i + 2];
                const double vt0 = c * v0 + sn * v1, vt1 = c * v1 - sn * v0;
                const double m0 = la[k] * (-vt0 + la[3 * S + k] * v2), m1 = la[S + k] * (-vt1 - la[2 * S + k] * v2);
                a0s += c * m0 - sn * m1; a1s += sn * m0 + c * m1;              // R m
            });
            const double* n = &dlinv[3 * (size_t)l];
            t[2 * (size_t)l] = n[0] * a0s + n[1] * a1s; t[2 * (size_t)l + 1] = n[1] * a0s + n[2] * a1s;
        }
    }

    // KB: out = Hpp v - W t  (partial over this shard's edges); returns partial dot (out, v)
    double schur_pose(const std::vector<double>& v, std::vector<double>& out) {
        const tsgo::SellTable& tb = pr.by_pose; const size_t S = tb.slots();
        const tsgo::SellTable& od = pr.odom; const size_t SO = od.slots();
        double dot = 0;
        #pragma omp parallel for schedule(static) reduction(+ : dot)
        for (int i = 0; i < P; ++i) {
            const double c = ps[4 * (size_t)i + 2], sn = ps[4 * (size_t)i + 3];
            double acc0 = 0, acc1 = 0, acc2 = 0;
            for_slots(tb, i, [&](size_t k) {
                const uint32_t l = tb.idx[k];
                const double tx = t[2 * (size_t)l], ty = t[2 * (size_t)l + 1];
                const double t0 = pa[k] * (c * tx + sn * ty), t1 = pa[S + k] * (c * ty - sn * tx);
                acc0 += t0; acc1 += t1; acc2 += t0 * pa[3 * S + k] - t1 * pa[2 * S + k];
            });
            // -W t = [R (acc0, acc1) ; -acc2]
            double o0 = c * acc0 - sn * acc1, o1 = sn * acc0 + c * acc1, o2 = -acc2;
            const bool own = i >= pr.pose_first && i < pr.pose_last;
            const double v0 = v[3 * (size_t)i], v1 = v[3 * (size_t)i + 1], v2 = v[3 * (size_t)i + 2];
            if (own) {
                // this shard's share of the diagonal block: LM part of every shard is in dp (all-reduced),
                // so the FULL diagonal block is applied by the owner of the pose only
                double d0, d1, d2; tsgo::sym3_mul(&dp[6 * (size_t)i], v0, v1, v2, d0, d1, d2);
                o0 += d0; o1 += d1; o2 += d2;
                for_slots(od, i, [&](size_t k) {
                    const uint32_t j = od.idx[k] & kPoseMask;
                    const double* vj = &v[3 * (size_t)j];
                    if (general) {
                        double h[tsgo::PP_PLANES];
                        for (int m = 0; m < tsgo::PP_PLANES; ++m) h[m] = oa[(size_t)m * SO + k];
                        tsgo::pair_apply<double>(h, vj[0], vj[1], vj[2], o0, o1, o2);
                    } else { o0 -= oa[k] * vj[0]; o1 -= oa[SO + k] * vj[1]; o2 -= oa[2 * SO + k] * vj[2]; }
                });
            }
            out[3 * (size_t)i] = o0; out[3 * (size_t)i + 1] = o1; out[3 * (size_t)i + 2] = o2;
            dot += o0 * v0 + o1 * v1 + o2 * v2;
        }
        return dot;
    }

    // one linearisation: fills everything CG needs; returns gamma = r^T M^-1 r of the first residual and leaves
    // in gamma_ref the value the stopping rule compares against.  warm (optional): pose delta of the previous
    // Gauss-Newton iteration; PCG then starts from x0 = (1 - step) * warm, the un-taken remainder of that step
    // (r = b~ - S x0), and gamma_ref = gamma * (b^T D^-1 b) / (r0^T D^-1 r0) keeps the rule relative to b~.
    double gamma_ref = 0, bdb = 0, rdr_start = 0;
    const std::deque<std::vector<double>>* gal_hist = nullptr; int gal_m = 0;      // research: TSGO_TWIN_GALERKIN
    double linearize(const std::vector<double>* warm = nullptr) {
        lin_lm(); lin_pose();
        // gauge of owned poses goes into the partial so that it is summed exactly once across shards
        for (int i = pr.pose_first; i < pr.pose_last; ++i) { const double ga = pr.gauge_p[i] + lambda; part[(size_t)i * 18] += ga; part[(size_t)i * 18 + 3] += ga; part[(size_t)i * 18 + 5] += ga; }
        allreduce(part.data(), (int64_t)part.size());
        double g0 = finalize();
        bdb = rdr_start = g0;           // b~^T D^-1 b~: what the multigrid PCG's stopping rule measures against (k_cg_step)
        double scale = 1;
        // research (TSGO_TWIN_GALERKIN=m): x0 = the S-norm-optimal combination of the last m deltas (m products with THIS linearisation's S,
        // an m x m system) instead of the polynomial continuation; profiles/r03z_galerkin_warm_start_twin.txt
        std::vector<double> xg;
        if (warm && gal_hist && gal_m > 0 && !gal_hist->empty()) {
            // basis: the polynomial prediction itself (TSGO_TWIN_GALERKIN_POLY=1) + the newest deltas
            static const bool with_poly = getenv("TSGO_TWIN_GALERKIN_POLY") != nullptr;
            std::vector<std::vector<double>> basis;
            if (with_poly) { basis.push_back(*warm); for (double& v : basis.back()) v *= (1.0 - step); }
            for (size_t j = 0; j < gal_hist->size() && (int)basis.size() < gal_m; ++j) basis.push_back((*gal_hist)[j]);
            const std::vector<std::vector<double>>* gal_basis = &basis;
            const int m = (int)basis.size();
            std::vector<std::vector<double>> q(m, std::vector<double>((size_t)P * 3 + 1));
            for (int j = 0; j < m; ++j) { const std::vector<double>& d = basis[j]; schur_lm(d); q[j][(size_t)P * 3] = schur_pose(d, q[j]); allreduce(q[j].data(), (int64_t)q[j].size()); }
            std::vector<double> G((size_t)m * m), rhs(m), c(m, 0.0);
            for (int i = 0; i < m; ++i) {
                const std::vector<double>& d = (*gal_basis)[i];
                double t = 0; for (int k = 0; k < 3 * P; ++k) t += d[k] * r[k];
                rhs[i] = t;
                for (int j = 0; j < m; ++j) { double g = 0; for (int k = 0; k < 3 * P; ++k) g += d[k] * q[j][k]; G[(size_t)i * m + j] = g; }
            }
            // Gaussian elimination with partial pivoting (m <= 6); a nearly dependent basis is caught by the pivot test
            std::vector<double> A = G, bb = rhs; bool ok = true;
            for (int col = 0; col < m && ok; ++col) {
                int piv = col; for (int rw = col + 1; rw < m; ++rw) if (std::fabs(A[(size_t)rw * m + col]) > std::fabs(A[(size_t)piv * m + col])) piv = rw;
                if (std::fabs(A[(size_t)piv * m + col]) < 1e-14 * std::fabs(G[0])) { ok = false; break; }
                if (piv != col) { for (int k = 0; k < m; ++k) std::swap(A[(size_t)piv * m + k], A[(size_t)col * m + k]); std::swap(bb[piv], bb[col]); }
                for (int rw = col + 1; rw < m; ++rw) { const double f = A[(size_t)rw * m + col] / A[(size_t)col * m + col]; for (int k = col; k < m; ++k) A[(size_t)rw * m + k] -= f * A[(size_t)col * m + k]; bb[rw] -= f * bb[col]; }
            }
            if (ok) {
                for (int rw = m - 1; rw >= 0; --rw) { double t = bb[rw]; for (int k = rw + 1; k < m; ++k) t -= A[(size_t)rw * m + k] * c[k]; c[rw] = t / A[(size_t)rw * m + rw]; }
                xg.assign((size_t)P * 3, 0.0);
                for (int j = 0; j < m; ++j) for (int k = 0; k < 3 * P; ++k) xg[k] += c[j] * (*gal_basis)[j][k] / (1.0 - step);
                warm = &xg;
            }
        }
        if (warm && warm->size() == x.size()) {
            for (size_t k = 0; k < x.size(); ++k) x[k] = (1.0 - step) * (*warm)[k];
            std::vector<double> buf((size_t)P * 3 + 1);
            schur_lm(x);
            buf[(size_t)P * 3] = schur_pose(x, buf);
            allreduce(buf.data(), (int64_t)buf.size());
            double nr = 0;
            for (int i = 0; i < P; ++i) {
                for (int k = 0; k < 3; ++k) r[3 * (size_t)i + k] -= buf[3 * (size_t)i + k];
                tsgo::sym3_mul(&minv[6 * (size_t)i], r[3 * (size_t)i], r[3 * (size_t)i + 1], r[3 * (size_t)i + 2], z[3 * (size_t)i], z[3 * (size_t)i + 1], z[3 * (size_t)i + 2]);
                for (int k = 0; k < 3; ++k) nr += r[3 * (size_t)i + k] * z[3 * (size_t)i + k];
            }
            if (nr > 0 && g0 > nr) scale = g0 / nr;
            g0 = nr; rdr_start = nr;
        }
        if (use_amg) {
            // the coarse matrices may lag the linearisation (level 0 never does): rebuilt after kHierMaxAge solves or
            // when the last solve took kHierSlack iterations more than the first one on this hierarchy
            static const int max_age = getenv("TSGO_HIER_MAX_AGE") ? std::max(1, atoi(getenv("TSGO_HIER_MAX_AGE"))) : 4;      // = kHierMaxAge (tsgo_hip.hip)
            const bool refresh = hier_age < 0 || hier_age >= (n_lin < 6 ? std::min(max_age, 2) : max_age) || iters_last > iters_fresh + 2;      // kYoungLins, kYoungMaxAge (tsgo_hip.hip)
            if (!amg.levels.empty() && refresh) { build_schur_blocks(); hier.setup_from_level0(); hier_age = 0; }
            ++n_lin;
            amg_apply();
            double g = 0;
            for (int i = 0; i < 3 * P; ++i) g += r[i] * z[i];
            gamma_ref = g * scale;
            return g;
        }
        gamma_ref = g0 * scale;
        return g0;
    }

    // Chronopoulos-Gear PCG on the reduced (pose) system S x = b~.  Returns iterations.
    // Stopping rule, both preconditioners: sqrt(r^T D^-1 r) <= tol sqrt(b~^T D^-1 b~), D = the 3x3 block diagonal of S.  With
    // block-Jacobi that is gamma itself; under the multigrid cycle it is evaluated on the residual each step produces, BEFORE the
    // cycle is applied to it (as k_cg_step does on the device: a finished solve does not pay one more cycle and product).
    int solve(double gamma_first, double tol, int max_it, bool* ok) {
        std::vector<double> buf((size_t)P * 3 + 1);
        const double gamma0 = gamma_ref;
        double gamma = gamma_first, gamma_old = 0, alpha_old = 0;
        *ok = true;
        if (!(gamma0 > 0)) return 0;
        int it = 0;
        if (use_amg && rdr_start <= tol * tol * bdb) return 0;           // a warm start that already meets the rule (k_warm_scale)
        for (; it < max_it; ++it) {
            if (gamma < 0 || gamma != gamma) { *ok = false; break; }     // indefinite preconditioner: breakdown, not convergence
            if (!use_amg && gamma <= tol * tol * gamma0) break;
            if (use_amg && !(gamma > 0)) break;                          // M^-1 r vanished
            schur_lm(z);
            buf[(size_t)P * 3] = schur_pose(z, buf);
            allreduce(buf.data(), (int64_t)buf.size());
            const double delta = buf[(size_t)P * 3];
            double alpha, beta;
            if (it == 0) { beta = 0; alpha = gamma / delta; }
            else { beta = gamma / gamma_old; alpha = gamma / (delta - beta * gamma / alpha_old); }
            if (!(alpha > 0) || !std::isfinite(alpha)) { *ok = false; break; }
            std::vector<double>& gi = gscratch; gi.assign(P, 0.0);
            #pragma omp parallel for schedule(static)
            for (int i = 0; i < P; ++i) {
                for (int k = 0; k < 3; ++k) {
                    const size_t j = 3 * (size_t)i + k;
                    p[j] = z[j] + beta * p[j]; q[j] = buf[j] + beta * q[j];
                    x[j] += alpha * p[j]; r[j] -= alpha * q[j];
                }
                double z0, z1, z2;
                tsgo::sym3_mul(&minv[6 * (size_t)i], r[3 * (size_t)i], r[3 * (size_t)i + 1], r[3 * (size_t)i + 2], z0, z1, z2);
                gi[i] = r[3 * (size_t)i] * z0 + r[3 * (size_t)i + 1] * z1 + r[3 * (size_t)i + 2] * z2;      // r^T D^-1 r
                if (!use_amg) { z[3 * (size_t)i] = z0; z[3 * (size_t)i + 1] = z1; z[3 * (size_t)i + 2] = z2; }
            }
            double gnew = 0;
            for (int i = 0; i < P; ++i) gnew += gi[i];       // serial: identical on every shard
            if (use_amg) {
                if (!(gnew > tol * tol * bdb)) { ++it; if (gnew != gnew) *ok = false; break; }      // this step's residual meets the rule: x is final
                amg_apply();
                gnew = 0;
                for (int i = 0; i < P; ++i) { double g = 0; for (int k = 0; k < 3; ++k) g += r[3 * (size_t)i + k] * z[3 * (size_t)i + k]; gi[i] = g; }
                for (int i = 0; i < P; ++i) gnew += gi[i];
            }
            gamma_old = gamma; alpha_old = alpha; gamma = gnew;
        }
        if (use_amg) { iters_last = it; if (hier_age == 0) iters_fresh = it; if (hier_age >= 0) ++hier_age; if (!*ok) hier_age = -1; }
        return it;
    }

    // As Engine::do_solve (tsgo_hip.hip): a multigrid-preconditioned solve that breaks down (indefinite or singular
    // coarse operator, e.g. a graph without any fixed vertex) is repeated from the same right-hand side with block-Jacobi.
    int n_fallbacks = 0;
    int solve_with_fallback(double gamma_first, double tol, int max_it, bool* ok) {
        int it = solve(gamma_first, tol, max_it, ok);
        if (!*ok && use_amg) {
            ++n_fallbacks;
            use_amg = false;
            const double g0 = finalize();          // r = b~, x = 0, z = D^-1 r
            gamma_ref = g0;
            it = solve(g0, tol, max_it, ok);
            use_amg = true;
            hier_age = -1;
        }
        return it;
    }

    // back-substitution for the landmarks: delta_l = u - Dl^-1 W^T delta_p
    void backsub(std::vector<double>& dl) {
        schur_lm(x);
        dl.resize((size_t)L * 2);
        for (size_t k = 0; k < dl.size(); ++k) dl[k] = u[k] - t[k];
    }

    // VertexSe2.h:16-27, Vertex2d.h:16-19 with the 0.2 step of OptimizerCpu.h:164
    double update(const std::vector<double>& dl) {
        double n2 = 0;
        for (int i = 0; i < P; ++i) {
            const double* d = &x[3 * (size_t)i];
            n2 += d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
            ps[4 * (size_t)i] += step * d[0]; ps[4 * (size_t)i + 1] += step * d[1];
            const double nt = std::atan2(ps[4 * (size_t)i + 3], ps[4 * (size_t)i + 2]) + step * d[2];
            th[i] = nt; ps[4 * (size_t)i + 2] = std::cos(nt); ps[4 * (size_t)i + 3] = std::sin(nt);
        }
        double l2 = 0;
        for (int l = 0; l < L; ++l) {
            l2 += dl[2 * (size_t)l] * dl[2 * (size_t)l] + dl[2 * (size_t)l + 1] * dl[2 * (size_t)l + 1];
            ls[2 * (size_t)l] += step * dl[2 * (size_t)l]; ls[2 * (size_t)l + 1] += step * dl[2 * (size_t)l + 1];
        }
        // landmark deltas are shard-local: sum their squared norm across shards
        double b[1] = {l2}; allreduce(b, 1);
        return std::sqrt(n2 + b[0]);
    }

    void store(double* v_pos, double* delta_or_null, const std::vector<double>* dl) const {
        for (int i = 0; i < P; ++i) {
            const int v = pr.pose_vertex[i];
            if (v_pos) { v_pos[3 * (size_t)v] = ps[4 * (size_t)i]; v_pos[3 * (size_t)v + 1] = ps[4 * (size_t)i + 1]; v_pos[3 * (size_t)v + 2] = std::atan2(ps[4 * (size_t)i + 3], ps[4 * (size_t)i + 2]); }
            if (delta_or_null) for (int k = 0; k < 3; ++k) delta_or_null[3 * (size_t)v + k] = x[3 * (size_t)i + k];
        }
        for (int l = 0; l < L; ++l) {
            const int v = pr.lm_vertex[l];
            if (v_pos) { v_pos[3 * (size_t)v] = ls[2 * (size_t)l]; v_pos[3 * (size_t)v + 1] = ls[2 * (size_t)l + 1]; v_pos[3 * (size_t)v + 2] = 0; }
            if (delta_or_null && dl) { delta_or_null[3 * (size_t)v] = (*dl)[2 * (size_t)l]; delta_or_null[3 * (size_t)v + 1] = (*dl)[2 * (size_t)l + 1]; delta_or_null[3 * (size_t)v + 2] = 0; }
        }
    }
};

tsgo_graph make_view(int nV, const uint32_t* v_id, const uint32_t* v_type, const double* v_pos, int nE, const uint32_t* e_type,
                     const uint32_t* e_ids, const double* e_meas, const double* e_inf, int nF, const uint32_t* fixed) {
    tsgo_graph g; g.n_vertices = nV; g.v_id = v_id; g.v_type = v_type; g.v_pos = v_pos; g.n_edges = nE; g.e_type = e_type;
    g.e_ids = e_ids; g.e_meas = e_meas; g.e_inf = e_inf; g.n_fixed = nF; g.fixed = fixed; return g;
}

}  // namespace

#define GRAPH_ARGS int nV, const uint32_t* v_id, const uint32_t* v_type, const double* v_pos, int nE, \
    const uint32_t* e_type, const uint32_t* e_ids, const double* e_meas, const double* e_inf, int nF, const uint32_t* fixed
#define GRAPH_PASS nV, v_id, v_type, v_pos, nE, e_type, e_ids, e_meas, e_inf, nF, fixed

extern "C" {

// OpenMP threads used by the twin's loops (a GPU box exposes far more logical CPUs than its share).
void oracle_set_threads(int n) { omp_set_num_threads(n < 1 ? 1 : n); }
void oracle_set_cycle_level0(int explicit_matrix) { g_cycle_level0 = explicit_matrix ? 1 : 0; }

// One Gauss-Newton step at the given state: delta (3 per vertex, graph order; landmarks of other
// shards are left 0), chi2, PCG iterations.  hook/ctx: all-reduce(sum) over shards, may be NULL.
int oracle_sparse_step(GRAPH_ARGS, double pcg_tol, int max_cg, int precond, int rank, int world, allreduce_fn hook, void* ctx,
                       double* delta_out, double* chi2_out, int* cg_iters) {
    Twin tw; tw.hook = hook; tw.hook_ctx = ctx;
    const tsgo_graph g = make_view(GRAPH_PASS);
    if (!tw.init(g, rank, world).empty()) return -2;
    if (precond == 1 && !tw.enable_amg(g).empty()) return -5;
    const double gamma0 = tw.linearize();
    bool ok; *cg_iters = tw.solve_with_fallback(gamma0, pcg_tol, max_cg, &ok);
    std::vector<double> dl; tw.backsub(dl);
    std::memset(delta_out, 0, sizeof(double) * 3 * (size_t)nV);
    tw.store(nullptr, delta_out, &dl);
    *chi2_out = tw.chi2;
    return ok ? 0 : -4;
}

// Full loop with the reference's rules (OptimizerCpu.h:80-180).  v_pos_out: 3 per vertex (landmarks of
// other shards are left at their input value).  stop_reason as in oracle_dense.cpp.
// rules 0: OptimizerCpu.h:80-180.  rules 1: graph_optimizer.py:20-92 (lambda * I, step lr, b zeroed at fixed vertices, stop on ||lr dx||).
int oracle_sparse_optimize(GRAPH_ARGS, double* v_pos_out, int iterations, double pcg_tol, int max_cg, int precond, int rank, int world,
                           allreduce_fn hook, void* ctx, double* chi2_trace, int* iters_run, int* stop_reason,
                           int* cg_trace, double* last_delta_norm, double* seconds_lin, double* seconds_solve, int rules, double lr) {
    Twin tw; tw.hook = hook; tw.hook_ctx = ctx;
    const bool py = rules == 1;
    if (py) { tw.zero_fixed = true; tw.step = lr; }
    double lam = 1e-3;
    const tsgo_graph g = make_view(GRAPH_PASS);
    if (!tw.init(g, rank, world).empty()) return -2;
    if (precond == 1 && !tw.enable_amg(g).empty()) return -5;
    std::memcpy(v_pos_out, v_pos, sizeof(double) * 3 * (size_t)nV);
    double prevErr = -1; int penalty = 0;
    static const int galerkin_m = getenv("TSGO_TWIN_GALERKIN") ? atoi(getenv("TSGO_TWIN_GALERKIN")) : 0;
    constexpr int kTwinMaxWarm = 6; constexpr double kTwinWarmMargin = 4;      // kMaxWarm, kWarmMargin of tsgo_kernels.h
    std::deque<std::vector<double>> hist;      // pose deltas of the last solves, newest first
    double warm_err[kTwinMaxWarm] = {}; int n_tested = 0;
    *stop_reason = 0; *iters_run = 0; *last_delta_norm = 0;
    if (seconds_lin) *seconds_lin = 0;
    if (seconds_solve) *seconds_solve = 0;
    auto now = [] { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; };
    // research: a PCG tolerance per Gauss-Newton iteration, TSGO_TWIN_TOL_SCHED="1e-7,1e-7,1e-10" (the last entry repeats)
    std::vector<double> tol_sched;
    if (const char* e = getenv("TSGO_TWIN_TOL_SCHED")) for (const char* q = e; *q;) { tol_sched.push_back(atof(q)); while (*q && *q != ',') ++q; if (*q == ',') ++q; }
    for (int it = 0; it < iterations; ++it) {
        double t0 = now();
        static const bool cold = getenv("TSGO_TWIN_COLD") != nullptr;
        // as Engine::launch_warm / k_pack_x (tsgo_hip.hip, tsgo_kernels.h): x0 = sum_j c[m][j] d_j, the continuation of the degree-(m-1)
        // trend of d_j / a^j, at the order m (<= TSGO_TWIN_WARM, default 6) that would have predicted the last delta best
        static const int warm_cap = getenv("TSGO_TWIN_WARM") ? atoi(getenv("TSGO_TWIN_WARM")) : 6;
        std::vector<double> xw;
        if (it > 0 && !cold && warm_cap > 0 && tw.step < 1.0) {      // a full step (lr = 1) leaves no remainder to start from
            const double a = 1.0 - tw.step;
            auto coeff = [&](int m, int j) { double binom = 1, apow = 1; for (int q = 1; q <= j; ++q) { binom = binom * (m - q + 1) / q; apow *= a; } return (j & 1 ? 1.0 : -1.0) * binom * apow; };   // j = 1..m
            int m = 1;
            const int n_max = std::max(1, std::min({(int)hist.size(), warm_cap, kTwinMaxWarm}));
            if (n_tested > 0) {
                double best = warm_err[0];
                for (int j = 1; j < n_tested && j < n_max; ++j) if (warm_err[j] * kTwinWarmMargin < best) { best = warm_err[j]; m = j + 1; }
            }
            xw.assign(hist[0].size(), 0.0);
            for (int j = 1; j <= m; ++j) { const double c = coeff(m, j) / a; for (size_t k = 0; k < xw.size(); ++k) xw[k] += c * hist[j - 1][k]; }   // linearize() scales by a
        }
        const std::vector<double>* warm = xw.empty() ? nullptr : &xw;
        tw.gal_hist = &hist; tw.gal_m = galerkin_m;
        double gamma0;
        if (py) {          // as Engine::optimize: linearise with the lambda a non-increasing chi^2 gives, repeat when it did rise
            tw.lambda = std::max(lam / 1.1, 1e-6);
            gamma0 = tw.linearize(warm);
            if (prevErr > -1 && tw.chi2 > prevErr) { tw.lambda = std::min(lam * 1.1, 1e1); gamma0 = tw.linearize(warm); }
            lam = tw.lambda;
        } else gamma0 = tw.linearize(warm);
        double t1 = now(); if (seconds_lin) *seconds_lin += t1 - t0;
        const double err = tw.chi2;
        chi2_trace[it] = err; *iters_run = it + 1;
        if (!py) { if (prevErr > 0 && err > prevErr) { if (++penalty > 2) { *stop_reason = 1; break; } } else penalty = 0; }
        const double tol_it = tol_sched.empty() ? pcg_tol : tol_sched[std::min((size_t)it, tol_sched.size() - 1)];
        bool ok; cg_trace[it] = tw.solve_with_fallback(gamma0, tol_it, max_cg, &ok);
        if (!ok) { *stop_reason = 4; break; }
        {   // k_save_x: what every order would have made of predicting this delta from the ones before it
            const double a = 1.0 - tw.step;
            const int n_test = std::min({(int)hist.size(), warm_cap, kTwinMaxWarm - 1});
            for (int m = 1; m <= n_test; ++m) {
                double binom = 1, apow = 1; double c[kTwinMaxWarm];
                for (int j = 1; j <= m; ++j) { binom = binom * (m - j + 1) / j; apow *= a; c[j - 1] = (j & 1 ? 1.0 : -1.0) * binom * apow; }
                double e = 0;
                for (size_t k = 0; k < tw.x.size(); ++k) { double p = 0; for (int j = 0; j < m; ++j) p += c[j] * hist[j][k]; e += (tw.x[k] - p) * (tw.x[k] - p); }
                warm_err[m - 1] = e;
            }
            n_tested = n_test;
            hist.push_front(tw.x);
            if ((int)hist.size() > kTwinMaxWarm) hist.pop_back();
        }
        std::vector<double> dl; tw.backsub(dl);
        const double nrm = (py ? lr : 1.0) * tw.update(dl);
        if (seconds_solve) *seconds_solve += now() - t1;
        *last_delta_norm = nrm;
        if (!py && std::fabs(err - prevErr) < tsgo::kPlateauTol) { *stop_reason = 2; break; }
        if (nrm < tsgo::kDeltaTol) { *stop_reason = 3; break; }
        prevErr = err;
    }
    tw.store(v_pos_out, nullptr, nullptr);
    return 0;
}

}  // extern "C"
