// tsgo_math.h — per-edge arithmetic of the Gauss-Newton hot path, shared by the HIP kernels
// (tsgo_kernels.hip) and by host code.  Everything is expressed in the POSE FRAME of the edge's
// first vertex, which is what lets one LM edge be stored as four numbers (a0, a1, ppx, ppy).
//
// Reference behaviour restated here (paths relative to the ToySlam tree):
//   LM edge   remote/graph/edge/EdgeSe2Point2d.h:27-70   e = R^T (l - t) - z,  A = [-R^T | v],  B = R^T
//             with v = (ppy, -ppx), (ppx, ppy) = R^T (l - t)   (A(0,2) = ppy, A(1,2) = -ppx, :58,:61)
//   ODOM edge remote/graph/edge/EdgeSe2.h:23-38          e = (D02, D12, atan2(D10, D00)),
//             D = meas^-1 (T1^-1 T2),  A = -I, B = +I (constant, :35-37)
//   Huber     remote/optimizer/OptimizerCpu.h:36-46      on chi^2, delta = 1.5
//   blocks    remote/optimizer/OptimizerCpu.h:88-119     Omega_w = w_huber * Omega (diagonal on the wire,
//             remote/serialization/DeserializeGraph.h:123-147)
#pragma once

#if defined(__HIPCC__)
#define TSGO_HD __host__ __device__ __forceinline__
#else
#define TSGO_HD inline
#endif

#include <cmath>

namespace tsgo {

constexpr double kHuberDelta = 1.5;      // OptimizerCpu.h:92
constexpr double kStepScale = 0.2;       // OptimizerCpu.h:164
constexpr double kGauge = 1e6;           // OptimizerCpu.h:136
constexpr double kPlateauTol = 1e-3;     // OptimizerCpu.h:167
constexpr double kDeltaTol = 1e-3;       // OptimizerCpu.h:173

template <typename T> TSGO_HD void huber(T chi2, T& rho, T& w) {
    const T d = T(kHuberDelta), d2 = d * d;
    if (chi2 <= d2) { rho = chi2; w = T(1); }
    else { const T sq = sqrt(chi2); rho = T(2) * sq * d - d2; w = d / sq; }
}

// One LM edge seen from pose (x, y, c = cos th, s = sin th) and landmark (lx, ly); z = (r cos phi,
// r sin phi) is precomputed on the host.  Outputs the four numbers every later pass needs.
template <typename T> struct LmLin { T a0, a1, ppx, ppy, e0, e1, rho; };

template <typename T>
TSGO_HD LmLin<T> lm_linearize(T x, T y, T c, T s, T lx, T ly, T zx, T zy, T w0, T w1) {
    LmLin<T> o;
    const T dx = lx - x, dy = ly - y;
    o.ppx = c * dx + s * dy;
    o.ppy = c * dy - s * dx;
    o.e0 = o.ppx - zx;
    o.e1 = o.ppy - zy;
    const T chi2 = w0 * o.e0 * o.e0 + w1 * o.e1 * o.e1;
    T hw;
    huber(chi2, o.rho, hw);
    o.a0 = hw * w0;
    o.a1 = hw * w1;
    return o;
}

// ODOM edge: pose1 (x1,y1,c1,s1) -> pose2; mi = top two rows of meas^-1 (row-major 2x3).
template <typename T> struct OdomLin { T a[3], e[3], rho; };

// Analytic Jacobians of that residual (tsgo_config.odom_jacobian = 1; an extension: the reference's are the constants
// A = -I, B = +I, EdgeSe2.h:35-37) under the reference's own vertex update (theta and the world-frame translation are
// added to, VertexSe2.h:16-27).  With N = the 2x2 corner of meas^-1 and M = N R1^T:
//   A = [[-M, q], [0 0 -kappa]],  B = [[M, 0], [0 0 kappa]],  q = N (dR1^T/dth1)(t2 - t1) = N (py, -px),  kappa = d e_th / d(th2 - th1).
// What the passes need of the edge's blocks (Omega_t = diag(a0, a1) Huber-scaled):
//   K = M^T Omega_t M (k00 k01 k11),  g = M^T Omega_t q,  s = q^T Omega_t q,  w = a2 kappa^2
//   H_11 += [[K, -g], [., s + w]],  H_22 += [[K, 0], [0, w]],  H_12 = [[-K, 0], [g^T, -w]]
//   b_1 = [h; -ht + kt],  b_2 = [-h; -kt]   with h = M^T Omega_t e_t, ht = q^T Omega_t e_t, kt = kappa a2 e_th
// (M = I, q = 0, kappa = 1 gives back the constant-Jacobian blocks: K = diag(a0, a1), w = a2, b = +-a e.)
template <typename T> struct OdomBlocks { T k00, k01, k11, g0, g1, s, w, h0, h1, ht, kt; };

template <typename T>
TSGO_HD OdomBlocks<T> odom_blocks(const OdomLin<T>& o, T x1, T y1, T c1, T s1, T x2, T y2, T c2, T s2, const T* mi) {
    OdomBlocks<T> b;
    const T dx = x2 - x1, dy = y2 - y1;
    const T px = c1 * dx + s1 * dy, py = c1 * dy - s1 * dx;
    const T m00 = mi[0] * c1 - mi[1] * s1, m01 = mi[0] * s1 + mi[1] * c1, m10 = mi[3] * c1 - mi[4] * s1, m11 = mi[3] * s1 + mi[4] * c1;
    const T q0 = mi[0] * py - mi[1] * px, q1 = mi[3] * py - mi[4] * px;
    const T cc = c1 * c2 + s1 * s2, ss = c1 * s2 - s1 * c2;
    const T d00 = mi[0] * cc + mi[1] * ss, d10 = mi[3] * cc + mi[4] * ss;
    const T e00 = -mi[0] * ss + mi[1] * cc, e10 = -mi[3] * ss + mi[4] * cc;
    const T den = d00 * d00 + d10 * d10;
    const T kappa = den > T(0) ? (d00 * e10 - d10 * e00) / den : T(0);      // a padding slot of the device table has mi = 0 (and zero weights)
    const T a0 = o.a[0], a1 = o.a[1], a2 = o.a[2];
    b.k00 = m00 * m00 * a0 + m10 * m10 * a1; b.k01 = m00 * m01 * a0 + m10 * m11 * a1; b.k11 = m01 * m01 * a0 + m11 * m11 * a1;
    b.g0 = m00 * a0 * q0 + m10 * a1 * q1; b.g1 = m01 * a0 * q0 + m11 * a1 * q1;
    b.s = a0 * q0 * q0 + a1 * q1 * q1; b.w = a2 * kappa * kappa;
    b.h0 = m00 * a0 * o.e[0] + m10 * a1 * o.e[1]; b.h1 = m01 * a0 * o.e[0] + m11 * a1 * o.e[1];
    b.ht = a0 * q0 * o.e[0] + a1 * q1 * o.e[1]; b.kt = kappa * a2 * o.e[2];
    return b;
}

template <typename T>
TSGO_HD OdomLin<T> odom_linearize(T x1, T y1, T c1, T s1, T x2, T y2, T c2, T s2, const T* mi, const T* w) {
    OdomLin<T> o;
    const T dx = x2 - x1, dy = y2 - y1;
    const T px = c1 * dx + s1 * dy, py = c1 * dy - s1 * dx;       // R1^T (t2 - t1)
    const T cc = c1 * c2 + s1 * s2, ss = c1 * s2 - s1 * c2;       // R1^T R2 = [[cc,-ss],[ss,cc]]
    o.e[0] = mi[0] * px + mi[1] * py + mi[2];
    o.e[1] = mi[3] * px + mi[4] * py + mi[5];
    o.e[2] = atan2(mi[3] * cc + mi[4] * ss, mi[0] * cc + mi[1] * ss);
    const T chi2 = w[0] * o.e[0] * o.e[0] + w[1] * o.e[1] * o.e[1] + w[2] * o.e[2] * o.e[2];
    T hw;
    huber(chi2, o.rho, hw);
    o.a[0] = hw * w[0]; o.a[1] = hw * w[1]; o.a[2] = hw * w[2];
    return o;
}

// ---- pose-pose slots in general form -------------------------------------------------------------------------------------------
// A pose-pose edge is listed at BOTH endpoints; the slot at pose p (neighbour n) carries ITS row block of the Hessian,
//     H_pn = [[-K, c], [r^T, -kappa]]        K symmetric 2x2 (k00 k01 k11), c and r 2-vectors, kappa scalar: 8 numbers,
// already oriented for p, so that the passes that use it (the Schur product of a pose row, the explicit level-0 block) need not
// know what kind of edge it came from:
//     constant ODOM Jacobians (EdgeSe2.h:35-37)   K = diag(a0, a1), c = r = 0, kappa = a2
//     analytic ODOM Jacobians, p the first pose   K, c = 0, r = g, kappa = w          (H_12 = [[-K, 0], [g^T, -w]], above)
//                              p the second pose  K, c = g, r = 0, kappa = w          (H_21 = H_12^T)
//     virtual landmark measurement                K = Omega, c = -Omega v, r = -Omega u, kappa = u^T Omega v     (below)
// The 8-plane layout is what kernels built for `general pairs` (k_lin_pose<.., 1>, k_schur_pose<.., 1>, k_schur_blocks) use.
enum { PP_K00 = 0, PP_K01, PP_K11, PP_C0, PP_C1, PP_R0, PP_R1, PP_KAPPA, PP_PLANES };
template <typename T> struct PairSlot { T v[PP_PLANES]; };

// out_p += H_pn z_n
template <typename T> TSGO_HD void pair_apply(const T* h, T z0, T z1, T zt, T& o0, T& o1, T& o2) {
    o0 += -(h[PP_K00] * z0 + h[PP_K01] * z1) + h[PP_C0] * zt;
    o1 += -(h[PP_K01] * z0 + h[PP_K11] * z1) + h[PP_C1] * zt;
    o2 += h[PP_R0] * z0 + h[PP_R1] * z1 - h[PP_KAPPA] * zt;
}

// Virtual landmark measurement (edge type 2; python/optimizer/edges2d.py:83-121, restated in oracle/oracle_dense.cpp: vlm_edge): the
// same physical point seen from two poses, e = T1 p1 - T2 p2, A = [I | dR1/dth p1], B = -[I | dR2/dth p2].  Seen from the slot's OWN
// pose (x, y, c, s; local point (pox, poy)) with the neighbour (xn, yn, cn, sn; (pnx, pny)):
//   d = own world point - neighbour's world point  (= e at the first endpoint, -e at the second),  u = dR_own/dth p_own,  v = dR_n/dth p_n
//   J_own^T Omega (signed residual) = [I | u]^T Omega d,   H_pp += [I | u]^T Omega [I | u],   H_pn = -[I | u]^T Omega [I | v]
// with Omega = Huber weight * diag(w0, w1) — no direction bit needed beyond counting chi^2 once.
template <typename T> struct VlmLin { T om0, om1, u0, u1, v0, v1, d0, d1, rho; };
template <typename T>
TSGO_HD VlmLin<T> vlm_linearize(T x, T y, T c, T s, T xn, T yn, T cn, T sn, T pox, T poy, T pnx, T pny, T w0, T w1) {
    VlmLin<T> o;
    o.d0 = (x + c * pox - s * poy) - (xn + cn * pnx - sn * pny);
    o.d1 = (y + s * pox + c * poy) - (yn + sn * pnx + cn * pny);
    o.u0 = -s * pox - c * poy; o.u1 = c * pox - s * poy;
    o.v0 = -sn * pnx - cn * pny; o.v1 = cn * pnx - sn * pny;
    const T chi2 = w0 * o.d0 * o.d0 + w1 * o.d1 * o.d1;
    T hw;
    huber(chi2, o.rho, hw);
    o.om0 = hw * w0; o.om1 = hw * w1;
    return o;
}
template <typename T> TSGO_HD void vlm_slot(const VlmLin<T>& o, T* h) {
    h[PP_K00] = o.om0; h[PP_K01] = T(0); h[PP_K11] = o.om1;
    h[PP_C0] = -o.om0 * o.v0; h[PP_C1] = -o.om1 * o.v1;
    h[PP_R0] = -o.om0 * o.u0; h[PP_R1] = -o.om1 * o.u1;
    h[PP_KAPPA] = o.om0 * o.u0 * o.v0 + o.om1 * o.u1 * o.v1;
}

// Symmetric 2x2 inverse (xx, xy, yy).  A singular block (vertex without edges) maps to zero, which
// leaves that vertex where it is (the reference's rank-revealing QR does the same).
template <typename T> TSGO_HD void inv_sym2(T xx, T xy, T yy, T& ixx, T& ixy, T& iyy) {
    const T det = xx * yy - xy * xy;
    if (!(fabs(det) > T(0))) { ixx = ixy = iyy = T(0); return; }
    const T r = T(1) / det;
    ixx = yy * r; ixy = -xy * r; iyy = xx * r;
}

// Symmetric 3x3 inverse, storage (00, 01, 02, 11, 12, 22).
template <typename T> TSGO_HD void inv_sym3(const T* m, T* o) {
    const T c00 = m[3] * m[5] - m[4] * m[4];
    const T c01 = m[2] * m[4] - m[1] * m[5];
    const T c02 = m[1] * m[4] - m[2] * m[3];
    const T det = m[0] * c00 + m[1] * c01 + m[2] * c02;
    if (!(fabs(det) > T(0))) { for (int k = 0; k < 6; ++k) o[k] = T(0); return; }
    const T r = T(1) / det;
    o[0] = c00 * r; o[1] = c01 * r; o[2] = c02 * r;
    o[3] = (m[0] * m[5] - m[2] * m[2]) * r;
    o[4] = (m[1] * m[2] - m[0] * m[4]) * r;
    o[5] = (m[0] * m[3] - m[1] * m[1]) * r;
}

template <typename T> TSGO_HD void sym3_mul(const T* m, T v0, T v1, T v2, T& o0, T& o1, T& o2) {
    o0 = m[0] * v0 + m[1] * v1 + m[2] * v2;
    o1 = m[1] * v0 + m[3] * v1 + m[4] * v2;
    o2 = m[2] * v0 + m[4] * v1 + m[5] * v2;
}

}  // namespace tsgo
