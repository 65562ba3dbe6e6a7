// tsgo_kernels.h — the hand-written gfx950 kernels of the Gauss-Newton hot path.
//
// Execution shape shared by every "table" kernel: one 64-lane wavefront per SELL slice, G lanes
// (1, 2, 4 or 8) cooperating on one vertex, a workgroup of 256 threads = 4 slices.  A lane walks the
// rows of its slice; every per-slot plane load is a fully coalesced 64-lane row (512 B in f64), the
// only irregular access is ONE small gathered record per slot (a 64-B pose record or a 16-B
// landmark vector).  Sums over a vertex's edges stay in registers and finish with G-lane xor
// shuffles: no atomics anywhere, results are bitwise reproducible.  MFMA is not used: there is no
// dense contraction on this path (blocks are 3x3 / 3x2 and the arithmetic intensity is < 1 flop/B).
//
// What each kernel replaces in the reference's CUDA pipeline (function, not code):
//   lin_lm / lin_pose   ProcessSe2Point2s + ProcessSe2s  (remote/cuda/optimizer/kernels/
//                       KernelSe2Point2.cu:46-155, KernelSe2.cu:37-112): residual, Jacobian, Huber,
//                       J^T W J / J^T W r — but into block-sparse storage by segmented register sums
//                       instead of 25-36 atomicAdds per edge into a dense n x n H
//   pose_finalize       FixVertices + Negativate (KernelCommon.cu:6-25,62-71) + preconditioner
//   schur_lm/schur_pose + cg_update   the linear solve (cuSOLVER geqrf/ormqr + cuBLAS trsm,
//                       remote/cuda/solver/SolverCudaQr.h:51-79) as implicit-Schur PCG
//   pose_update / schur_lm<BACKSUB>   Update (KernelCommon.cu:27-60)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "tsgo_math.h"

namespace tsgo {

constexpr int kBlock = 256;             // threads per workgroup = 4 wavefronts
constexpr int kWavesPerBlock = kBlock / 64;
constexpr uint32_t kDirMask = 0x80000000u;      // odom slot: the row's pose is the edge's second vertex
constexpr uint32_t kVlmMask = 0x40000000u;      // odom slot: a virtual landmark measurement (edge type 2), general-pairs kernels only
constexpr uint32_t kPoseIdxMask = 0x3FFFFFFFu;  // odom slot: the neighbour's internal pose number

// record layouts (in units of T)
constexpr int kPoseRec = 8;   // zc: v0 v1 v2 c s . . .      (the vector CG multiplies + the pose's cos/sin)
constexpr int kNinvRec = 4;   // ninv: ixx ixy iyy .          (Dl^-1 alone, what k_schur_lm's product mode reads)
constexpr int kLmRec = 8;     // lmrec: lx ly ixx ixy iyy ux uy .   (landmark, Dl^-1, u = Dl^-1 g_l)

// 16-byte (f64) / 8-byte (f32) pair loads for the gathered records: one gathered 64-B pose record costs
// three load instructions instead of five (each wave-instruction of a gather touches 64 lines).
template <typename T> struct Pair;
template <> struct Pair<double> { using type = double2; };
template <> struct Pair<float> { using type = float2; };
template <typename T> __device__ __forceinline__ typename Pair<T>::type ld2(const T* p) {
    return *reinterpret_cast<const typename Pair<T>::type*>(p);
}

template <typename T> __device__ __forceinline__ void st2(T* p, T a, T b) {
    typename Pair<T>::type v; v.x = a; v.y = b;
    *reinterpret_cast<typename Pair<T>::type*>(p) = v;
}

// Keeps the compiler from sinking a load below an early exit: the value must exist when this statement is reached, so the load
// that produces it is issued (and waited for) before the test that follows — beside the flag's own load instead of behind it.
template <typename V> __device__ __forceinline__ void issue_before_exit(V& v) { asm volatile("" : "+v"(v)); }

template <typename T> struct CgState { T gamma_old, alpha_old, gamma0, pad; int iters, done, fail, pad2; };

// LM tables keep their per-slot values as 16-byte PAIRS, pair-plane-major: st = [(zx,zy) | (w0,w1)],
// dyn = [(a0,a1) | (ppx,ppy)], element k of pair-plane q at (q*slots + k)*2: one dwordx4 per lane and
// plane instead of two dwordx2 (8-byte lanes stream at 0.54-0.70x the 16-byte rate, MI355X_MICROARCH.md).
// The ODOM table keeps single-value planes.  dyn32 = the same four numbers of every LM slot as ONE float4: the two
// Schur products INSIDE the multigrid cycle read it instead (a preconditioner tolerates f32 operands; the
// product PCG itself takes stays f64).
template <typename T> struct Table {       // one SELL table on the device
    const uint32_t* row_off; const uint32_t* idx; const T* st; T* dyn; float4* dyn32; size_t slots; int n_slices; int n_vertices; int xcd;
};

// XCD-aware workgroup -> slice-group map (guide T1).  Workgroups are dealt round-robin over the 8 XCDs,
// each with a private L2; giving XCD x the x-th contiguous eighth of the slices makes every gathered
// record (pose records, landmark vectors) live in ONE L2 instead of being pulled into all eight
// (profiles/r01a: 1.7x over-fetch on k_schur_lm).  Grids are rounded up to a multiple of 8; only speed
// depends on the placement guess, never correctness.
__device__ __forceinline__ int xcd_block() { return (int)((blockIdx.x % 8u) * (gridDim.x / 8u) + blockIdx.x / 8u); }

template <typename T, int G> __device__ __forceinline__ T group_sum(T v) {
#pragma unroll
    for (int m = 1; m < G; m <<= 1) v += __shfl_xor(v, m);
    return v;
}

template <typename T> __device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}

// Sum over the workgroup; every thread gets the result.  `red` = kWavesPerBlock T's of LDS.
template <typename T> __device__ __forceinline__ T block_sum(T v, T* red) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    T s = red[0];
#pragma unroll
    for (int k = 1; k < kWavesPerBlock; ++k) s += red[k];
    return s;
}

// Fixed-order sum of n partials by a whole workgroup (n is a few hundred).
template <typename T> __device__ __forceinline__ T block_sum_array(const T* a, int n, T* red) {
    T v = 0;
    for (int k = threadIdx.x; k < n; k += kBlock) v += a[k];
    return block_sum(v, red);
}

// The stopping rule of the multigrid-preconditioned PCG, evaluated by ONE workgroup at the start of an iteration (tsgo_amg_kernels.h:
// k_iter_gate, or workgroup 0 of the iteration's first product kernel): sums the partials of r^T D^-1 r (left by k_cg_step) and of
// b^T D^-1 b, sets st->done when the residual meets the tolerance, and — eager launches — tells the host thread in ONE aligned 8-byte
// store that the seq-th launched iteration has started and what it saw (seq | done | fail | iterations completed; Engine::do_solve_paced).
template <typename T> struct GateArgs { CgState<T>* st; const T* rdr_part; const T* bpart; int n; T tol2; int* host_flag; int seq; };
template <typename T> __device__ __forceinline__ void iter_gate_body(const GateArgs<T> g, T* red) {
    const int done = g.st->done, iters = g.st->iters, fail0 = g.st->fail;
    T a = 0, b = 0;                                  // the partials are on their way while the state is looked at
    for (int k = threadIdx.x; k < g.n; k += kBlock) { a += g.rdr_part[k]; b += g.bpart[k]; }
    issue_before_exit(a);
    int done_now = done, fail_now = fail0;
    if (!(done || iters == 0)) {                     // iteration 0: no step has been taken yet (a warm start is judged by k_warm_scale)
        const T rdr = block_sum<T>(a, red);
        const T bdb = block_sum<T>(b, red);
        if (threadIdx.x == 0 && !(rdr > g.tol2 * bdb)) { fail_now = (rdr != rdr) ? 1 : 0; done_now = 1; g.st->done = 1; g.st->fail = fail_now; }      // NaN: breakdown
    }
    if (g.host_flag && threadIdx.x == 0)      // ONE aligned 8-byte store: the host never sees the fields of two gates mixed
        *reinterpret_cast<volatile unsigned long long*>(g.host_flag) =
            ((unsigned long long)(unsigned)g.seq << 32) | ((unsigned long long)(done_now ? 1u : 0u) << 31) | ((unsigned long long)((unsigned)fail_now & 7u) << 28) | (unsigned long long)((unsigned)iters & 0x0fffffffu);
}

// ------------------------------------------------------------------------------------------------
// K1 lin_lm: per landmark — landmark-side linearisation of its LM edges.
//   reads : slot planes zx zy w0 w1 + pose index (coalesced), pose state ps[i] = (x,y,c,s) (gather)
//   writes: slot planes a0 a1 ppx ppy (lm-major copy), lmrec[l][2..6] = Dl^-1, u
template <typename T, int G>
__global__ __launch_bounds__(kBlock) void k_lin_lm(Table<T> tb, const T* __restrict__ ps, T* __restrict__ lmrec,
                                                   const T* __restrict__ gauge_l, T* __restrict__ ninv, T lambda, int zero_fixed) {
    const int slice = (tb.xcd ? xcd_block() : (int)blockIdx.x) * kWavesPerBlock + (threadIdx.x >> 6);
    if (slice >= tb.n_slices) return;
    const int lane = threadIdx.x & 63;
    constexpr int VPS = 64 / G;
    const int l = slice * VPS + lane / G;
    const bool valid = l < tb.n_vertices;
    const int lc = valid ? l : tb.n_vertices - 1;
    const T lx = lmrec[(size_t)lc * kLmRec], ly = lmrec[(size_t)lc * kLmRec + 1];
    T dxx = 0, dxy = 0, dyy = 0, g0 = 0, g1 = 0;
    const size_t S = tb.slots;
    const uint32_t r0 = tb.row_off[slice], r1 = tb.row_off[slice + 1];
    for (uint32_t row = r0; row < r1; ++row) {
        const size_t k = (size_t)row * 64 + lane;
        const uint32_t i = tb.idx[k];
        const auto zz = ld2<T>(tb.st + 2 * k), ww = ld2<T>(tb.st + 2 * (S + k));
        const T zx = zz.x, zy = zz.y, w0 = ww.x, w1 = ww.y;
        const T* q = ps + (size_t)i * 4;
        const auto q01 = ld2<T>(q), q23 = ld2<T>(q + 2);
        const T x = q01.x, y = q01.y, c = q23.x, s = q23.y;
        const LmLin<T> o = lm_linearize<T>(x, y, c, s, lx, ly, zx, zy, w0, w1);
        st2<T>(tb.dyn + 2 * k, o.a0, o.a1); st2<T>(tb.dyn + 2 * (S + k), o.ppx, o.ppy);
        tb.dyn32[k] = make_float4((float)o.a0, (float)o.a1, (float)o.ppx, (float)o.ppy);
        dxx += o.a0 * c * c + o.a1 * s * s; dxy += (o.a0 - o.a1) * c * s; dyy += o.a0 * s * s + o.a1 * c * c;
        const T f0 = o.a0 * o.e0, f1 = o.a1 * o.e1;
        g0 -= c * f0 - s * f1; g1 -= s * f0 + c * f1;
    }
    dxx = group_sum<T, G>(dxx); dxy = group_sum<T, G>(dxy); dyy = group_sum<T, G>(dyy);
    g0 = group_sum<T, G>(g0); g1 = group_sum<T, G>(g1);
    if (valid && (lane % G) == 0) {
        // lambda: the LM-style damping of the reference's Python optimizer (H + lambda I, graph_optimizer.py:42), 0 under the
        // cpu/eigen rules; zero_fixed: that optimizer also zeroes b at fixed vertices (:150), OptimizerCpu.h does not
        const T ga = gauge_l[l];
        if (zero_fixed && ga > T(0)) { g0 = 0; g1 = 0; }
        T ixx, ixy, iyy;
        inv_sym2<T>(dxx + ga + lambda, dxy, dyy + ga + lambda, ixx, ixy, iyy);
        T* o = lmrec + (size_t)l * kLmRec;
        o[2] = ixx; o[3] = ixy; o[4] = iyy; o[5] = ixx * g0 + ixy * g1; o[6] = ixy * g0 + iyy * g1;
        // compact copy of Dl^-1 for the Schur product's epilogue: 32 B per landmark instead of a 64-B record
        T* nq = ninv + (size_t)l * kNinvRec;
        nq[0] = ixx; nq[1] = ixy; nq[2] = iyy;
    }
}

// ------------------------------------------------------------------------------------------------
// K2 lin_pose: per pose — pose-side linearisation of its LM edges + its ODOM rows.
//   writes: slot planes (pose-major copy), ODOM weights, part[i][18] = Dp(6) g(3) Sd(6) Wu(3) in the
//           world frame, one chi^2 partial per workgroup
// OJ = 1: pose-pose slots in GENERAL form (tsgo_math.h: eight dynamic planes per slot, the slot's own row block of the Hessian) — what
// analytic ODOM Jacobians (odom_analytic, tsgo_config.odom_jacobian) and virtual landmark measurements (edge type 2, kVlmMask) need;
// ODOM edges under the reference's constant Jacobians are written in that form too when the graph holds the other kind.
template <typename T, int G, int OJ = 0>
__global__ __launch_bounds__(kBlock) void k_lin_pose(Table<T> tb, Table<T> od, const T* __restrict__ ps,
                                                     const T* __restrict__ lmrec, const T* __restrict__ gauge_p,
                                                     int pose_first, int pose_last, T* __restrict__ part,
                                                     T* __restrict__ chi_part, T lambda, int zero_fixed, int odom_analytic = 0) {
    __shared__ T red[kWavesPerBlock];
    const int slice = (tb.xcd ? xcd_block() : (int)blockIdx.x) * kWavesPerBlock + (threadIdx.x >> 6);
    const bool live = slice < tb.n_slices;
    const int lane = threadIdx.x & 63;
    constexpr int VPS = 64 / G;
    const int i = slice * VPS + lane / G;
    const bool valid = live && i < tb.n_vertices;
    T chi = 0;
    if (live) {
        const int ic = valid ? i : tb.n_vertices - 1;
        const T x0 = ps[(size_t)ic * 4], y0 = ps[(size_t)ic * 4 + 1], c = ps[(size_t)ic * 4 + 2], s = ps[(size_t)ic * 4 + 3];
        T sA0 = 0, sA1 = 0, sAv0 = 0, sAv1 = 0, sVV = 0, ge0 = 0, ge1 = 0, get = 0;
        T K00 = 0, K01 = 0, K11 = 0, Kv0 = 0, Kv1 = 0, vKv = 0, wu0 = 0, wu1 = 0, wut = 0;
        {
            const size_t S = tb.slots;
            const uint32_t r0 = tb.row_off[slice], r1 = tb.row_off[slice + 1];
            for (uint32_t row = r0; row < r1; ++row) {
                const size_t k = (size_t)row * 64 + lane;
                const uint32_t l = tb.idx[k];
                const auto zz = ld2<T>(tb.st + 2 * k), ww = ld2<T>(tb.st + 2 * (S + k));
                const T zx = zz.x, zy = zz.y, w0 = ww.x, w1 = ww.y;
                const T* lr = lmrec + (size_t)l * kLmRec;
                const auto l01 = ld2<T>(lr), l23 = ld2<T>(lr + 2), l45 = ld2<T>(lr + 4);
                const T lx = l01.x, ly = l01.y, nxx = l23.x, nxy = l23.y, nyy = l45.x, ux = l45.y, uy = lr[6];
                const LmLin<T> o = lm_linearize<T>(x0, y0, c, s, lx, ly, zx, zy, w0, w1);
                st2<T>(tb.dyn + 2 * k, o.a0, o.a1); st2<T>(tb.dyn + 2 * (S + k), o.ppx, o.ppy);
                tb.dyn32[k] = make_float4((float)o.a0, (float)o.a1, (float)o.ppx, (float)o.ppy);
                chi += o.rho;
                const T v0 = o.ppy, v1 = -o.ppx;
                sA0 += o.a0; sA1 += o.a1; sAv0 += o.a0 * v0; sAv1 += o.a1 * v1; sVV += o.a0 * v0 * v0 + o.a1 * v1 * v1;
                ge0 += o.a0 * o.e0; ge1 += o.a1 * o.e1; get += o.a0 * o.e0 * v0 + o.a1 * o.e1 * v1;
                const T n00 = c * c * nxx + 2 * c * s * nxy + s * s * nyy;
                const T n01 = c * s * (nyy - nxx) + (c * c - s * s) * nxy;
                const T n11 = s * s * nxx - 2 * c * s * nxy + c * c * nyy;
                const T k00 = o.a0 * n00 * o.a0, k01 = o.a0 * n01 * o.a1, k11 = o.a1 * n11 * o.a1;
                K00 += k00; K01 += k01; K11 += k11;
                const T kv0 = k00 * v0 + k01 * v1, kv1 = k01 * v0 + k11 * v1;
                Kv0 += kv0; Kv1 += kv1; vKv += v0 * kv0 + v1 * kv1;
                const T t0 = c * ux + s * uy, t1 = c * uy - s * ux;
                wu0 += o.a0 * t0; wu1 += o.a1 * t1; wut += o.a0 * t0 * v0 + o.a1 * t1 * v1;
            }
        }
        // ODOM rows (listed at both endpoints; chi^2 counted at id1)
        T od0 = 0, od1 = 0, od2 = 0, og0 = 0, og1 = 0, og2 = 0;
        T od01 = 0, od02 = 0, od12 = 0;        // off-diagonal entries of the pose's ODOM diagonal block: analytic Jacobians only
        {
            const size_t S = od.slots;
            const uint32_t r0 = od.row_off[slice], r1 = od.row_off[slice + 1];
            for (uint32_t row = r0; row < r1; ++row) {
                const size_t k = (size_t)row * 64 + lane;
                const uint32_t raw = od.idx[k];
                const bool second = (raw & kDirMask) != 0;
                const uint32_t j = raw & kPoseIdxMask;
                T mi[6], w[3];
#pragma unroll
                for (int m = 0; m < 6; ++m) mi[m] = od.st[(size_t)m * S + k];
#pragma unroll
                for (int m = 0; m < 3; ++m) w[m] = od.st[(size_t)(6 + m) * S + k];
                const T* oq = ps + (size_t)j * 4;
                const T xj = oq[0], yj = oq[1], cj = oq[2], sj = oq[3];
                if (OJ && (raw & kVlmMask)) {      // virtual landmark measurement: mi = (pox, poy, pnx, pny, ., .), w = (w0, w1, .)
                    const VlmLin<T> v = vlm_linearize<T>(x0, y0, c, s, xj, yj, cj, sj, mi[0], mi[1], mi[2], mi[3], w[0], w[1]);
                    T h[PP_PLANES];
                    vlm_slot<T>(v, h);
#pragma unroll
                    for (int m = 0; m < PP_PLANES; ++m) od.dyn[(size_t)m * S + k] = h[m];
                    od0 += v.om0; od1 += v.om1; od02 += v.om0 * v.u0; od12 += v.om1 * v.u1; od2 += v.om0 * v.u0 * v.u0 + v.om1 * v.u1 * v.u1;
                    og0 -= v.om0 * v.d0; og1 -= v.om1 * v.d1; og2 -= v.om0 * v.u0 * v.d0 + v.om1 * v.u1 * v.d1;
                    if (!second) chi += v.rho;
                    continue;
                }
                const OdomLin<T> o = second ? odom_linearize<T>(xj, yj, cj, sj, x0, y0, c, s, mi, w)
                                            : odom_linearize<T>(x0, y0, c, s, xj, yj, cj, sj, mi, w);
                if (OJ && odom_analytic) {
                    const OdomBlocks<T> ob = second ? odom_blocks<T>(o, xj, yj, cj, sj, x0, y0, c, s, mi) : odom_blocks<T>(o, x0, y0, c, s, xj, yj, cj, sj, mi);
                    // H_12 = [[-K, 0], [g^T, -w]] at the first endpoint, its transpose at the second
                    od.dyn[(size_t)PP_K00 * S + k] = ob.k00; od.dyn[(size_t)PP_K01 * S + k] = ob.k01; od.dyn[(size_t)PP_K11 * S + k] = ob.k11;
                    od.dyn[(size_t)PP_C0 * S + k] = second ? ob.g0 : T(0); od.dyn[(size_t)PP_C1 * S + k] = second ? ob.g1 : T(0);
                    od.dyn[(size_t)PP_R0 * S + k] = second ? T(0) : ob.g0; od.dyn[(size_t)PP_R1 * S + k] = second ? T(0) : ob.g1;
                    od.dyn[(size_t)PP_KAPPA * S + k] = ob.w;
                    od0 += ob.k00; od01 += ob.k01; od1 += ob.k11;
                    if (!second) { od02 -= ob.g0; od12 -= ob.g1; od2 += ob.s + ob.w; og0 += ob.h0; og1 += ob.h1; og2 += ob.kt - ob.ht; }
                    else { od2 += ob.w; og0 -= ob.h0; og1 -= ob.h1; og2 -= ob.kt; }
                } else {
                    if (OJ) {      // the reference's constants in the general form: K = diag(a0, a1), c = r = 0, kappa = a2
                        od.dyn[(size_t)PP_K00 * S + k] = o.a[0]; od.dyn[(size_t)PP_K01 * S + k] = T(0); od.dyn[(size_t)PP_K11 * S + k] = o.a[1];
                        od.dyn[(size_t)PP_C0 * S + k] = T(0); od.dyn[(size_t)PP_C1 * S + k] = T(0); od.dyn[(size_t)PP_R0 * S + k] = T(0); od.dyn[(size_t)PP_R1 * S + k] = T(0);
                        od.dyn[(size_t)PP_KAPPA * S + k] = o.a[2];
                    } else { od.dyn[k] = o.a[0]; od.dyn[S + k] = o.a[1]; od.dyn[2 * S + k] = o.a[2]; }
                    od0 += o.a[0]; od1 += o.a[1]; od2 += o.a[2];
                    const T sg = second ? T(-1) : T(1);
                    og0 += sg * o.a[0] * o.e[0]; og1 += sg * o.a[1] * o.e[1]; og2 += sg * o.a[2] * o.e[2];
                }
                // a padding slot has w = 0: rho = 0 there
                if (!second) chi += o.rho;
            }
        }
#define GS(v) v = group_sum<T, G>(v)
        GS(sA0); GS(sA1); GS(sAv0); GS(sAv1); GS(sVV); GS(ge0); GS(ge1); GS(get);
        GS(K00); GS(K01); GS(K11); GS(Kv0); GS(Kv1); GS(vKv); GS(wu0); GS(wu1); GS(wut);
        GS(od0); GS(od1); GS(od2); GS(og0); GS(og1); GS(og2);
        if (OJ) { GS(od01); GS(od02); GS(od12); }
#undef GS
        if (valid && (lane % G) == 0) {
            // gauge and damping enter once: through the shard that owns the pose
            const bool own = i >= pose_first && i < pose_last;
            const T ga = own ? gauge_p[i] + lambda : T(0);
            const bool fixed_here = zero_fixed && gauge_p[i] > T(0);      // gauge_p holds the owned poses' terms; others see 0 and the owner's partial decides
            T* o = part + (size_t)i * 18;
            o[0] = c * c * sA0 + s * s * sA1 + od0 + ga; o[1] = c * s * (sA0 - sA1) + od01; o[3] = s * s * sA0 + c * c * sA1 + od1 + ga;
            o[2] = -(c * sAv0 - s * sAv1) + od02; o[4] = -(s * sAv0 + c * sAv1) + od12; o[5] = sVV + od2 + ga;
            o[6] = c * ge0 - s * ge1 + og0; o[7] = s * ge0 + c * ge1 + og1; o[8] = -get + og2;
            if (fixed_here) { o[6] = 0; o[7] = 0; o[8] = 0; }
            o[9] = c * c * K00 - 2 * c * s * K01 + s * s * K11;
            o[10] = c * s * (K00 - K11) + (c * c - s * s) * K01;
            o[12] = s * s * K00 + 2 * c * s * K01 + c * c * K11;
            o[11] = -(c * Kv0 - s * Kv1); o[13] = -(s * Kv0 + c * Kv1); o[14] = vKv;
            o[15] = -(c * wu0 - s * wu1); o[16] = -(s * wu0 + c * wu1); o[17] = wut;
        }
    }
    const T total = block_sum<T>(chi, red);
    if (threadIdx.x == 0) chi_part[blockIdx.x] = total;
}

// ------------------------------------------------------------------------------------------------
// K3 pose_finalize: per pose (thread = pose) — M = Dp - Sd (gauge already inside Dp), M^-1, reduced
// right-hand side, CG start (x = p = q = 0, r = b~, z = M^-1 r) and gamma_0 partials.
template <typename T>
__global__ __launch_bounds__(kBlock) void k_pose_finalize(int P, const T* __restrict__ part, const T* __restrict__ ps,
                                                          T* __restrict__ dp, T* __restrict__ minv, T* __restrict__ r,
                                                          T* __restrict__ p, T* __restrict__ q, T* __restrict__ x,
                                                          T* __restrict__ zc, T* __restrict__ gpart,
                                                          CgState<T>* __restrict__ st0, const T* __restrict__ omega_ptr,
                                                          T* __restrict__ gamma0_scale, float* __restrict__ zc32) {
    __shared__ T red[kWavesPerBlock];
    const int i = blockIdx.x * kBlock + threadIdx.x;
    T g = 0;
    if (i < P) {
        const T* o = part + (size_t)i * 18;
        T m[6], mi[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) { dp[(size_t)i * 6 + k] = o[k]; m[k] = o[k] - o[9 + k]; }
        inv_sym3<T>(m, mi);
#pragma unroll
        for (int k = 0; k < 6; ++k) minv[(size_t)i * 6 + k] = mi[k];
        const T r0 = o[6] - o[15], r1 = o[7] - o[16], r2 = o[8] - o[17];
        T z0, z1, z2;
        sym3_mul<T>(mi, r0, r1, r2, z0, z1, z2);
        r[(size_t)i * 3] = r0; r[(size_t)i * 3 + 1] = r1; r[(size_t)i * 3 + 2] = r2;
#pragma unroll
        for (int k = 0; k < 3; ++k) { p[(size_t)i * 3 + k] = 0; q[(size_t)i * 3 + k] = 0; x[(size_t)i * 3 + k] = 0; }
        T* zr = zc + (size_t)i * kPoseRec;
        const T w = *omega_ptr;    // 1 for block-Jacobi PCG; the level-0 smoother damping under the multigrid cycle
        zr[0] = w * z0; zr[1] = w * z1; zr[2] = w * z2; zr[3] = ps[(size_t)i * 4 + 2]; zr[4] = ps[(size_t)i * 4 + 3];
        if (zc32) {     // f32 copy of the record for the products inside the multigrid cycle (k_schur_lm / k_schur_pose, LOW)
            float* zq = zc32 + (size_t)i * kPoseRec;
            *reinterpret_cast<float4*>(zq) = make_float4((float)(w * z0), (float)(w * z1), (float)(w * z2), (float)zr[3]);
            zq[4] = (float)zr[4];
        }
        g = r0 * z0 + r1 * z1 + r2 * z2;
    }
    const T total = block_sum<T>(g, red);
    if (threadIdx.x == 0) gpart[blockIdx.x] = total;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        CgState<T> s; s.gamma_old = 0; s.alpha_old = 0; s.gamma0 = 0; s.pad = 0; s.iters = 0; s.done = 0; s.fail = 0; s.pad2 = 0;
        *st0 = s;
        *gamma0_scale = 1;       // cold start; the warm-start kernels overwrite it
    }
}

// ------------------------------------------------------------------------------------------------
// KA schur_lm: per landmark — t = Dl^-1 W^T v, v being the vector held in zc[.][0..2].
//   MODE 0: write t.   MODE 1 (back-substitution): dl = u - t, landmark += step * dl, |dl|^2 partial.
// HOT KERNEL 1 of the PCG iteration.
template <typename T, int G, int MODE, int LOW = 0>
__global__ __launch_bounds__(kBlock) void k_schur_lm(Table<T> tb, const T* __restrict__ zc, T* __restrict__ lmrec,
                                                     const T* __restrict__ ninv, T* __restrict__ t, const CgState<T>* __restrict__ st,
                                                     T step, T* __restrict__ dl_out, T* __restrict__ norm_part,
                                                     const float* __restrict__ zc32 = nullptr, float* __restrict__ t32 = nullptr,
                                                     const GateArgs<T> gate = GateArgs<T>{nullptr, nullptr, nullptr, 0, T(0), nullptr, 0}) {
    __shared__ T red[kWavesPerBlock];
    // The iteration's stopping rule rides in workgroup 0 of its first product (gate.st set): one launch fewer per iteration.  The other
    // workgroups do not wait for the verdict: a solve that has just converged runs this one pass for nothing (its output is scratch) and
    // every later kernel of the iteration sees st->done.
    if (MODE == 0 && gate.st != nullptr && blockIdx.x == 0) iter_gate_body<T>(gate, red);
    // the flag of a finished solve is requested here and tested after the loads that depend on the arguments alone (row bounds,
    // the vertex's inverse block) are on their way: tested first, it adds a scalar round trip in front of the first vector load
    const int done = MODE == 0 ? st->done : 0;
    const int slice = (tb.xcd ? xcd_block() : (int)blockIdx.x) * kWavesPerBlock + (threadIdx.x >> 6);
    const bool live = slice < tb.n_slices;
    const int lane = threadIdx.x & 63;
    constexpr int VPS = 64 / G;
    const int l = slice * VPS + lane / G;
    T nrm = 0;
    if (live) {
        T acc0 = 0, acc1 = 0;
        const size_t S = tb.slots;
        const uint32_t r0 = tb.row_off[slice], r1 = tb.row_off[slice + 1];
        // the inverse block this vertex needs at the very end is requested first: its latency hides behind the rows
        const int lq = (l < tb.n_vertices) ? l : tb.n_vertices - 1;
        const auto n01 = ld2<T>(ninv + (size_t)lq * kNinvRec);
        T n2 = ninv[(size_t)lq * kNinvRec + 2];
        issue_before_exit(n2);
        if (done) return;
        // UB rows are walked at a time with every load of the batch issued before any use: a wave's time is
        // the depth of its dependent-load chain (row -> index -> gathered pose record), so memory-level
        // parallelism is what buys time.  Rows past the end are clamped (in bounds) and masked out of the sums.
        constexpr int UB = (G >= 4) ? 2 : 4;
        for (uint32_t base = r0; base < r1; base += UB) {
            uint32_t i[UB]; T a0[UB], a1[UB], ppx[UB], ppy[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const uint32_t row = min(base + u, r1 - 1);
                const size_t k = (size_t)row * 64 + lane;
                i[u] = tb.idx[k];
                if (LOW) { const float4 f = tb.dyn32[k]; a0[u] = f.x; a1[u] = f.y; ppx[u] = f.z; ppy[u] = f.w; }
                else {
                    const auto aa = ld2<T>(tb.dyn + 2 * k), pp = ld2<T>(tb.dyn + 2 * (S + k));
                    a0[u] = aa.x; a1[u] = aa.y; ppx[u] = pp.x; ppy[u] = pp.y;
                }
            }
            T v0[UB], v1[UB], v2[UB], c[UB], s[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                if (LOW) {      // the f32 copy of the pose records (32 B each): half the bytes gathered through L2
                    const float* zr = zc32 + (size_t)i[u] * kPoseRec;
                    const float4 q = *reinterpret_cast<const float4*>(zr);
                    v0[u] = q.x; v1[u] = q.y; v2[u] = q.z; c[u] = q.w; s[u] = zr[4];
                } else {
                    const T* zr = zc + (size_t)i[u] * kPoseRec;
                    const auto z01 = ld2<T>(zr), z23 = ld2<T>(zr + 2);
                    v0[u] = z01.x; v1[u] = z01.y; v2[u] = z23.x; c[u] = z23.y; s[u] = zr[4];
                }
            }
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                if (base + u < r1) {
                    const T vt0 = c[u] * v0[u] + s[u] * v1[u], vt1 = c[u] * v1[u] - s[u] * v0[u];
                    const T m0 = a0[u] * (ppy[u] * v2[u] - vt0), m1 = a1[u] * (-vt1 - ppx[u] * v2[u]);
                    acc0 += c[u] * m0 - s[u] * m1; acc1 += s[u] * m0 + c[u] * m1;
                }
            }
        }
        acc0 = group_sum<T, G>(acc0); acc1 = group_sum<T, G>(acc1);
        if (l < tb.n_vertices && (lane % G) == 0) {
            T* lr = lmrec + (size_t)l * kLmRec;
            const T ixx = n01.x, ixy = n01.y, iyy = n2;
            const T t0 = ixx * acc0 + ixy * acc1, t1 = ixy * acc0 + iyy * acc1;
            if (MODE == 0) {
                if (LOW) *reinterpret_cast<float2*>(t32 + (size_t)l * 2) = make_float2((float)t0, (float)t1);
                else { t[(size_t)l * 2] = t0; t[(size_t)l * 2 + 1] = t1; }
            }
            else {
                const T d0 = lr[5] - t0, d1 = lr[6] - t1;
                dl_out[(size_t)l * 2] = d0; dl_out[(size_t)l * 2 + 1] = d1;
                lr[0] += step * d0; lr[1] += step * d1;
                nrm = d0 * d0 + d1 * d1;
            }
        }
    }
    else if (done) return;
    if (MODE == 1) {
        const T total = block_sum<T>(nrm, red);
        if (threadIdx.x == 0) norm_part[blockIdx.x] = total;
    }
}

// ------------------------------------------------------------------------------------------------
// KB schur_pose: per pose — out = Hpp v - W t (this shard's share), partial dot (out, v).
// HOT KERNEL 2 of the PCG iteration.
template <typename T, int G, int LOW = 0, int OJ = 0>
__global__ __launch_bounds__(kBlock) void k_schur_pose(Table<T> tb, Table<T> od, const T* __restrict__ zc,
                                                       const T* __restrict__ t, const T* __restrict__ dp,
                                                       int pose_first, int pose_last, T* __restrict__ out,
                                                       T* __restrict__ dot_part, const CgState<T>* __restrict__ st,
                                                       const T* __restrict__ rvec, T* __restrict__ rz_part,
                                                       const float* __restrict__ zc32 = nullptr, const float* __restrict__ t32 = nullptr,
                                                       const T* __restrict__ post_minv = nullptr, const T* __restrict__ post_r = nullptr,
                                                       const T* __restrict__ post_omega = nullptr, T* __restrict__ zc_post = nullptr) {
    __shared__ T red[kWavesPerBlock];
    // post_minv (the SECOND product inside a multigrid cycle, one shard, f32-copy operands): the level-0 post-smoothing
    // zc_i += omega Minv_i (r_i - (S z)_i) in this kernel's epilogue instead of a launch of its own (k_smooth0<1>).  The pass gathers
    // its operands from the f32 copies (zc32), so writing the f64 records it does not read is no race; the f32 copies are rewritten by
    // k_cg_step before anything reads them again.
    const int done = st->done;      // requested now, tested after the first argument-only loads are in flight (see k_schur_lm)
    const int slice = (tb.xcd ? xcd_block() : (int)blockIdx.x) * kWavesPerBlock + (threadIdx.x >> 6);
    const bool live = slice < tb.n_slices;
    const int lane = threadIdx.x & 63;
    constexpr int VPS = 64 / G;
    const int i = slice * VPS + lane / G;
    T dot = 0, rz = 0;
    if (!live && done) return;      // workgroup-uniform: done is, and the other waves of the group leave below
    if (live) {
        const bool valid = i < tb.n_vertices;
        const int ic = valid ? i : tb.n_vertices - 1;
        T v0, v1, v2, c, s;
        if (LOW) { const float* zr = zc32 + (size_t)ic * kPoseRec; v0 = zr[0]; v1 = zr[1]; v2 = zr[2]; c = zr[3]; s = zr[4]; }
        else { const T* zr = zc + (size_t)ic * kPoseRec; v0 = zr[0]; v1 = zr[1]; v2 = zr[2]; c = zr[3]; s = zr[4]; }
        const uint32_t lm_r0 = tb.row_off[slice], lm_r1 = tb.row_off[slice + 1];
        issue_before_exit(v0);
        if (done) return;
        // the pose's own diagonal block and residual entry are used at the very end: asked for first
        const bool head = valid && (lane % G) == 0;
        const bool own = head && i >= pose_first && i < pose_last;
        T dpi[6] = {0, 0, 0, 0, 0, 0}, rv0 = 0, rv1 = 0, rv2 = 0;
        if (own) {
#pragma unroll
            for (int m = 0; m < 6; ++m) dpi[m] = dp[(size_t)i * 6 + m];
        }
        if (head && rvec) { rv0 = rvec[(size_t)i * 3]; rv1 = rvec[(size_t)i * 3 + 1]; rv2 = rvec[(size_t)i * 3 + 2]; }
        T pm[6] = {0, 0, 0, 0, 0, 0}, pr0 = 0, pr1 = 0, pr2 = 0, pz0 = 0, pz1 = 0, pz2 = 0, pw = 0;      // post-smoothing operands: asked for before the rows are walked
        if (LOW && post_r && head) {      // post_r alone (the FIRST product of a cycle): `out` receives the residual r - S z the restriction wants
            pr0 = post_r[(size_t)i * 3]; pr1 = post_r[(size_t)i * 3 + 1]; pr2 = post_r[(size_t)i * 3 + 2];
            if (post_minv) {
#pragma unroll
                for (int m = 0; m < 6; ++m) pm[m] = post_minv[(size_t)i * 6 + m];
                const T* zr = zc_post + (size_t)i * kPoseRec;
                pz0 = zr[0]; pz1 = zr[1]; pz2 = zr[2]; pw = *post_omega;
            }
        }
        T acc0 = 0, acc1 = 0, acc2 = 0;
        {
            const size_t S = tb.slots;
            const uint32_t r0 = lm_r0, r1 = lm_r1;
#pragma unroll 2
            for (uint32_t row = r0; row < r1; ++row) {
                const size_t k = (size_t)row * 64 + lane;
                const uint32_t l = tb.idx[k];
                T a0, a1, ppx, ppy;
                if (LOW) { const float4 f = tb.dyn32[k]; a0 = f.x; a1 = f.y; ppx = f.z; ppy = f.w; }
                else {
                    const auto aa = ld2<T>(tb.dyn + 2 * k), pp = ld2<T>(tb.dyn + 2 * (S + k));
                    a0 = aa.x; a1 = aa.y; ppx = pp.x; ppy = pp.y;
                }
                T tx, ty;
                if (LOW) { const float2 txy = *reinterpret_cast<const float2*>(t32 + (size_t)l * 2); tx = txy.x; ty = txy.y; }
                else { const auto txy = ld2<T>(t + (size_t)l * 2); tx = txy.x; ty = txy.y; }
                const T t0 = a0 * (c * tx + s * ty), t1 = a1 * (c * ty - s * tx);
                acc0 += t0; acc1 += t1; acc2 += t0 * ppy - t1 * ppx;
            }
        }
        T o0 = 0, o1 = 0, o2 = 0;
        {
            const size_t S = od.slots;
            const uint32_t r0 = od.row_off[slice], r1 = od.row_off[slice + 1];
            for (uint32_t row = r0; row < r1; ++row) {
                const size_t k = (size_t)row * 64 + lane;
                const uint32_t raw = od.idx[k];
                const uint32_t j = raw & kPoseIdxMask;
                T zj[3];
                if (LOW) { const float* q = zc32 + (size_t)j * kPoseRec; zj[0] = q[0]; zj[1] = q[1]; zj[2] = q[2]; }
                else { const T* q = zc + (size_t)j * kPoseRec; zj[0] = q[0]; zj[1] = q[1]; zj[2] = q[2]; }
                if (OJ) {           // the slot's own row block H_pn = [[-K, c], [r^T, -kappa]] (tsgo_math.h), whatever edge it came from
                    T h[PP_PLANES];
#pragma unroll
                    for (int m = 0; m < PP_PLANES; ++m) h[m] = od.dyn[(size_t)m * S + k];
                    pair_apply<T>(h, zj[0], zj[1], zj[2], o0, o1, o2);
                } else { o0 -= od.dyn[k] * zj[0]; o1 -= od.dyn[S + k] * zj[1]; o2 -= od.dyn[2 * S + k] * zj[2]; }
            }
        }
        acc0 = group_sum<T, G>(acc0); acc1 = group_sum<T, G>(acc1); acc2 = group_sum<T, G>(acc2);
        o0 = group_sum<T, G>(o0); o1 = group_sum<T, G>(o1); o2 = group_sum<T, G>(o2);
        if (head) {
            o0 += c * acc0 - s * acc1; o1 += s * acc0 + c * acc1; o2 -= acc2;
            if (own) {
                T d0, d1, d2;
                sym3_mul<T>(dpi, v0, v1, v2, d0, d1, d2);
                o0 += d0; o1 += d1; o2 += d2;
            }
            if (LOW && post_r && !post_minv) { out[(size_t)i * 3] = pr0 - o0; out[(size_t)i * 3 + 1] = pr1 - o1; out[(size_t)i * 3 + 2] = pr2 - o2; }
            else { out[(size_t)i * 3] = o0; out[(size_t)i * 3 + 1] = o1; out[(size_t)i * 3 + 2] = o2; }
            dot = o0 * v0 + o1 * v1 + o2 * v2;
            if (rvec) rz = rv0 * v0 + rv1 * v1 + rv2 * v2;
            if (LOW && post_minv) {      // k_smooth0<1>: zc += omega Minv (r - S zc)
                T z0, z1, z2;
                sym3_mul<T>(pm, pr0 - o0, pr1 - o1, pr2 - o2, z0, z1, z2);
                T* zr = zc_post + (size_t)i * kPoseRec;
                zr[0] = pz0 + pw * z0; zr[1] = pz1 + pw * z1; zr[2] = pz2 + pw * z2;
            }
        }
    }
    const T total = block_sum<T>(dot, red);
    if (threadIdx.x == 0) dot_part[blockIdx.x] = total;
    if (rvec) {        // wave-uniform: a kernel argument
        const T trz = block_sum<T>(rz, red);
        if (threadIdx.x == 0) rz_part[blockIdx.x] = trz;
    }
}

// ------------------------------------------------------------------------------------------------
// KC cg_update: per pose — the vector half of a Chronopoulos-Gear PCG iteration (one global
// reduction point per iteration instead of two).  Every workgroup first sums the partials of
// delta = (S z, z) (from KB) and gamma = (r, z) (from the previous KC / pose_finalize) in a fixed
// order, then:  beta = gamma/gamma_old, alpha = gamma/(delta - beta*gamma/alpha_old),
//   p = z + beta p,  q = S z + beta q,  x += alpha p,  r -= alpha q,  z = M^-1 r.
template <typename T>
__global__ __launch_bounds__(kBlock) void k_cg_update(int P, const T* __restrict__ sz, const T* __restrict__ dot_part,
                                                      int n_dot, const T* __restrict__ gpart_in, int n_g,
                                                      T* __restrict__ gpart_out, const CgState<T>* __restrict__ st_in,
                                                      CgState<T>* __restrict__ st_out, const T* __restrict__ minv,
                                                      T* __restrict__ r, T* __restrict__ p, T* __restrict__ q,
                                                      T* __restrict__ x, T* __restrict__ zc, T tol2, int max_iters,
                                                      const T* __restrict__ gamma0_scale) {
    __shared__ T red[kWavesPerBlock];
    const CgState<T> s = *st_in;
    const bool writer = blockIdx.x == 0 && threadIdx.x == 0;
    if (s.done) { if (writer) *st_out = s; return; }
    const T delta = block_sum_array<T>(dot_part, n_dot, red);
    const T gamma = block_sum_array<T>(gpart_in, n_g, red);
    // warm-started solves measure convergence against the right-hand side, not against the (already small) first residual
    const T gamma0 = s.iters == 0 ? gamma * (*gamma0_scale) : s.gamma0;
    CgState<T> n = s; n.gamma0 = gamma0;
    if (!(gamma > tol2 * gamma0) || s.iters >= max_iters) {      // converged (or cap, or NaN): x is final
        n.done = 1; n.fail = (gamma != gamma) ? 1 : ((gamma > tol2 * gamma0) ? 2 : 0);
        if (writer) *st_out = n;
        return;
    }
    T beta, alpha;
    if (s.iters == 0) { beta = 0; alpha = gamma / delta; }
    else { beta = gamma / s.gamma_old; alpha = gamma / (delta - beta * gamma / s.alpha_old); }
    if (!(alpha > 0) || !(alpha < T(1e300))) {                    // breakdown: keep x, flag it
        n.done = 1; n.fail = 1;
        if (writer) *st_out = n;
        return;
    }
    const int i = blockIdx.x * kBlock + threadIdx.x;
    T g = 0;
    if (i < P) {
        T rr[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const size_t j = (size_t)i * 3 + k;
            const T zk = zc[(size_t)i * kPoseRec + k];
            const T pk = zk + beta * p[j];
            const T qk = sz[j] + beta * q[j];
            p[j] = pk; q[j] = qk;
            x[j] += alpha * pk;
            rr[k] = r[j] - alpha * qk;
            r[j] = rr[k];
        }
        T z0, z1, z2;
        sym3_mul<T>(minv + (size_t)i * 6, rr[0], rr[1], rr[2], z0, z1, z2);
        T* zr = zc + (size_t)i * kPoseRec;
        zr[0] = z0; zr[1] = z1; zr[2] = z2;
        g = rr[0] * z0 + rr[1] * z1 + rr[2] * z2;
    }
    const T total = block_sum<T>(g, red);
    if (threadIdx.x == 0) gpart_out[blockIdx.x] = total;
    if (writer) { n.gamma_old = gamma; n.alpha_old = alpha; n.iters = s.iters + 1; *st_out = n; }
}

// Warm start, first half.  d_k / a^k (a = 1 - step: what a damped step leaves of its delta) is smooth in k, so the next delta is
// predicted by continuing the polynomial of degree m - 1 through the last m deltas: x0 = sum_j c[m-1][j] v[j],
// c[m-1][j] = (-1)^j C(m, j+1) a^(j+1).  Which order m: the one that would have predicted the LAST delta best from the
// deltas before it — k_save_x leaves ||d - prediction_m||^2 for every order it could test in errpart[m-1][block].
constexpr int kMaxWarm = 6;      // deltas of earlier Gauss-Newton iterations the warm start can combine
constexpr int kWarmMargin = 4;   // a higher order is taken when its squared prediction error is this many times smaller: its coefficients
                                 // (order 4 at step 0.2: 3.2, -3.8, 2.0, -0.4) amplify whatever in the deltas is not a smooth trend
template <typename T> struct WarmTerms {
    const T* v[kMaxWarm];          // deltas, newest first
    T c[kMaxWarm][kMaxWarm];       // row m-1: the coefficients of order m
    int n_max;                     // highest order allowed (0: no warm start — plain copy)
    int n_tested;                  // orders 1..n_tested have an error in errpart
    const T* errpart; int nb_err;  // [kMaxWarm][nb_err]
};

// x -> zc[.][0..2] (the landmark pass reads its vector from zc).  w.n_max = 0: plain copy of x into zc.  Otherwise x = the
// prediction of the order with the smallest tested error (order 1 when nothing has been tested yet) first.
template <typename T>
__global__ __launch_bounds__(kBlock) void k_pack_x(int P, T* __restrict__ x, T* __restrict__ zc, const WarmTerms<T> w, int* __restrict__ order_out) {
    __shared__ T red[kWavesPerBlock];
    const int i = blockIdx.x * kBlock + threadIdx.x;
    int m = w.n_max > 0 ? 1 : 0;
    if (w.n_max > 0 && w.n_tested > 0) {          // every workgroup sums the same partials in the same order: one choice
        T best = block_sum_array<T>(w.errpart, w.nb_err, red);
        for (int j = 1; j < w.n_tested && j < w.n_max; ++j) {
            const T e = block_sum_array<T>(w.errpart + (size_t)j * w.nb_err, w.nb_err, red);
            if (e * kWarmMargin < best) { best = e; m = j + 1; }
        }
    }
    if (order_out && blockIdx.x == 0 && threadIdx.x == 0) *order_out = m;
    if (i >= P) return;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        T v = x[(size_t)i * 3 + k];
        if (m > 0) {
            v = w.c[m - 1][0] * w.v[0][(size_t)i * 3 + k];
            for (int j = 1; j < m; ++j) v += w.c[m - 1][j] * w.v[j][(size_t)i * 3 + k];
            x[(size_t)i * 3 + k] = v;
        }
        zc[(size_t)i * kPoseRec + k] = v;
    }
}

// The solved delta x -> zc[.][0..2] and -> xsave (the newest entry of the warm start's history), and what every order would
// have made of predicting it from the deltas before it (w.v, newest first; orders 1..w.n_max): errpart[m-1][block].
template <typename T>
__global__ __launch_bounds__(kBlock) void k_save_x(int P, const T* __restrict__ x, T* __restrict__ zc, T* __restrict__ xsave, const WarmTerms<T> w, T* __restrict__ errpart) {
    __shared__ T red[kWavesPerBlock];
    const int i = blockIdx.x * kBlock + threadIdx.x;
    T err[kMaxWarm];
#pragma unroll
    for (int m = 0; m < kMaxWarm; ++m) err[m] = 0;
    if (i < P) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const T v = x[(size_t)i * 3 + k];
            zc[(size_t)i * kPoseRec + k] = v;
            xsave[(size_t)i * 3 + k] = v;
            T u[kMaxWarm];
#pragma unroll
            for (int j = 0; j < kMaxWarm; ++j) u[j] = j < w.n_max ? w.v[j][(size_t)i * 3 + k] : T(0);
#pragma unroll
            for (int m = 0; m < kMaxWarm; ++m) {
                if (m < w.n_max) {
                    T p = 0;
#pragma unroll
                    for (int j = 0; j <= m; ++j) p += w.c[m][j] * u[j];
                    err[m] += (v - p) * (v - p);
                }
            }
        }
    }
    for (int m = 0; m < w.n_max; ++m) {
        const T total = block_sum<T>(err[m], red);
        if (threadIdx.x == 0) errpart[(size_t)m * gridDim.x + blockIdx.x] = total;
    }
}

// History across requests (tsgo_config.warm_requests): a delta of the previous graph, in ITS pose numbering, into this graph's —
// src[i] = the old index of pose i, or -1 for a pose the old graph did not hold.
template <typename T>
__global__ __launch_bounds__(kBlock) void k_gather_hist(int P, const int* __restrict__ src, const T* __restrict__ old, T* __restrict__ out) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= P) return;
    const int j = src[i];
#pragma unroll
    for (int k = 0; k < 3; ++k) out[(size_t)i * 3 + k] = j >= 0 ? old[(size_t)j * 3 + k] : T(0);
}

// Warm start, second half: r = b~ - S x0 (S x0 in sx), zc = omega Minv r, partials of r^T Minv r.
template <typename T>
__global__ __launch_bounds__(kBlock) void k_warm_residual(int P, const T* __restrict__ sx, const T* __restrict__ minv, T* __restrict__ r,
                                                          T* __restrict__ zc, const T* __restrict__ omega_ptr, T* __restrict__ part, float* __restrict__ zc32) {
    __shared__ T red[kWavesPerBlock];
    const int i = blockIdx.x * kBlock + threadIdx.x;
    T g = 0;
    if (i < P) {
        const T r0 = r[(size_t)i * 3] - sx[(size_t)i * 3], r1 = r[(size_t)i * 3 + 1] - sx[(size_t)i * 3 + 1], r2 = r[(size_t)i * 3 + 2] - sx[(size_t)i * 3 + 2];
        r[(size_t)i * 3] = r0; r[(size_t)i * 3 + 1] = r1; r[(size_t)i * 3 + 2] = r2;
        T z0, z1, z2;
        sym3_mul<T>(minv + (size_t)i * 6, r0, r1, r2, z0, z1, z2);
        const T w = *omega_ptr;
        T* zr = zc + (size_t)i * kPoseRec;
        zr[0] = w * z0; zr[1] = w * z1; zr[2] = w * z2;
        if (zc32) { float* zq = zc32 + (size_t)i * kPoseRec; zq[0] = (float)(w * z0); zq[1] = (float)(w * z1); zq[2] = (float)(w * z2); }
        g = r0 * z0 + r1 * z1 + r2 * z2;
    }
    const T total = block_sum<T>(g, red);
    if (threadIdx.x == 0) part[blockIdx.x] = total;
}

// scale = max(1, (b^T D^-1 b) / (r0^T D^-1 r0)) from the two partial arrays; one workgroup.
// st0 (multigrid PCG, whose stopping rule is r^T D^-1 r <= tol^2 b^T D^-1 b, k_cg_step): a warm start that already meets the
// rule ends the solve before its first iteration.
template <typename T>
__global__ __launch_bounds__(kBlock) void k_warm_scale(int n, const T* b_part, const T* __restrict__ r_part, T* gpart_out, T* __restrict__ scale,
                                                       CgState<T>* __restrict__ st0, T tol2) {
    __shared__ T red[kWavesPerBlock];
    const T nb = block_sum_array<T>(b_part, n, red);
    const T nr = block_sum_array<T>(r_part, n, red);
    // block-Jacobi PCG reads gamma = r^T Minv r from gpart: hand it the warm-started value
    if (gpart_out) for (int k = threadIdx.x; k < n; k += kBlock) gpart_out[k] = r_part[k];
    if (threadIdx.x == 0) {
        *scale = (nr > T(0) && nb > nr) ? nb / nr : T(1);
        if (st0 && !(nr > tol2 * nb)) st0->done = 1;
    }
}

// pose (+)= step * delta: VertexSe2::Update (remote/graph/vertex/VertexSe2.h:16-27) with the 0.2 of
// OptimizerCpu.h:164 — theta = atan2(sin, cos) + step*dth, translation added in the world frame.
template <typename T>
__global__ __launch_bounds__(kBlock) void k_pose_update(int P, const T* __restrict__ x, T* __restrict__ ps,
                                                        T* __restrict__ theta, T step, T* __restrict__ norm_part) {
    __shared__ T red[kWavesPerBlock];
    const int i = blockIdx.x * kBlock + threadIdx.x;
    T nrm = 0;
    if (i < P) {
        const T d0 = x[(size_t)i * 3], d1 = x[(size_t)i * 3 + 1], d2 = x[(size_t)i * 3 + 2];
        nrm = d0 * d0 + d1 * d1 + d2 * d2;
        if (step != T(0)) {
            T* q = ps + (size_t)i * 4;
            const T th = atan2(q[3], q[2]) + step * d2;
            q[0] += step * d0; q[1] += step * d1; q[2] = cos(th); q[3] = sin(th);
            theta[i] = th;
        }
    }
    const T total = block_sum<T>(nrm, red);
    if (threadIdx.x == 0) norm_part[blockIdx.x] = total;
}

}  // namespace tsgo
