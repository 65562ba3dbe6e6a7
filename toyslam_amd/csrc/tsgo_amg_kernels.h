// tsgo_amg_kernels.h — device side of the smoothed-aggregation multigrid preconditioner (host/amg.h):
// numeric setup once per Gauss-Newton iteration, V(1,1) cycle once per PCG iteration.
//
// All blocks are 3x3 row-major.  Every sum is a gather over a precomputed list in fixed order: no
// atomics, bitwise reproducible.  The coarse levels are tiny (12.5k / 1.5k / 196 / 25 block rows at
// 100k poses): those kernels are latency-bound by design and kept deliberately simple; the bytes are
// in level 0, whose residuals reuse the implicit Schur passes of tsgo_kernels.h.
#pragma once
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>

#include "tsgo_kernels.h"

namespace tsgo {

// Storage type of the multigrid hierarchy (A_l, D^-1, P, R, A P).  The hierarchy only preconditions: PCG
// converges to the same f64 solution with operators rounded to f32, and every setup/cycle kernel that is
// bound by gathered 72-byte blocks moves half the bytes.  Sums are accumulated in the vector type T.
#ifdef TSGO_HIER_F64
template <typename T> struct Hier { using type = T; };
#else
template <typename T> struct Hier { using type = float; };
#endif
template <typename T> using HT = typename Hier<T>::type;

constexpr int kPairBlocksPerWave = 21;   // k_pair_gemm: three lanes per 3x3 output block
constexpr int kDenseMax = 84;            // coarsest matrix is at most 84 x 84 (host/amg.h: kCoarsestMax * 3)

template <typename T, typename A, typename B> __device__ __forceinline__ void m3_mul_acc(const A* a, const B* b, T* c) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) c[3 * i + j] += T(a[3 * i]) * T(b[j]) + T(a[3 * i + 1]) * T(b[3 + j]) + T(a[3 * i + 2]) * T(b[6 + j]);
}
template <typename T, typename A, typename B> __device__ __forceinline__ void m3_tmul_acc(const A* a, const B* b, T* c) {   // c += a^T b
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) c[3 * i + j] += T(a[i]) * T(b[j]) + T(a[3 + i]) * T(b[3 + j]) + T(a[6 + i]) * T(b[6 + j]);
}
template <typename T> __device__ __forceinline__ void m3_inv(const T* m, T* o) {
    const T c00 = m[4] * m[8] - m[5] * m[7], c01 = m[5] * m[6] - m[3] * m[8], c02 = m[3] * m[7] - m[4] * m[6];
    const T det = m[0] * c00 + m[1] * c01 + m[2] * c02;
    if (!(fabs(det) > T(0))) { for (int k = 0; k < 9; ++k) o[k] = T(0); return; }
    const T r = T(1) / det;
    o[0] = c00 * r; o[1] = (m[2] * m[7] - m[1] * m[8]) * r; o[2] = (m[1] * m[5] - m[2] * m[4]) * r;
    o[3] = c01 * r; o[4] = (m[0] * m[8] - m[2] * m[6]) * r; o[5] = (m[2] * m[3] - m[0] * m[5]) * r;
    o[6] = c02 * r; o[7] = (m[1] * m[6] - m[0] * m[7]) * r; o[8] = (m[0] * m[4] - m[1] * m[3]) * r;
}

// cycle format (described further down, "cycle format"): words per 3x3 block, and the packing of one block
constexpr int kCyWordsF32 = 9, kCyWordsF16 = 5;
template <int PK> __host__ __device__ constexpr int cy_words() { return PK ? kCyWordsF16 : kCyWordsF32; }
// the nine values of one block into its cycle-format words (the j-th block of a row of `len` blocks whose words start at `o` = base + j)
template <int PK> __device__ __forceinline__ void cy_store(const float* v, uint32_t* __restrict__ o, size_t len) {
    if (PK) {
        float mx = 0;
#pragma unroll
        for (int m = 0; m < 9; ++m) mx = fmaxf(mx, fabsf(v[m]));
        int e = 0;
        if (mx > 0.f && mx < 3.0e38f) {
            e = ilogbf(mx) - 14;                                  // largest entry -> [2^14, 2^15)
            e = e < -126 ? -126 : (e > 112 ? 112 : e);            // 2^e and 2^-e stay normal floats
        }
        const float inv = __uint_as_float((uint32_t)(127 - e) << 23);
        unsigned short h[9];
#pragma unroll
        for (int m = 0; m < 9; ++m) h[m] = __half_as_ushort(__float2half_rn(v[m] * inv));
#pragma unroll
        for (int m = 0; m < 4; ++m) o[(size_t)m * len] = (uint32_t)h[2 * m] | ((uint32_t)h[2 * m + 1] << 16);
        o[(size_t)4 * len] = (uint32_t)h[8] | ((uint32_t)(unsigned short)(short)e << 16);
    } else {
#pragma unroll
        for (int m = 0; m < 9; ++m) o[(size_t)m * len] = __float_as_uint(v[m]);
    }
}

// ---- numeric setup ---------------------------------------------------------------------------------
// Explicit Schur complement blocks S_ik (level 0 of the hierarchy), one thread per block on or above the diagonal.
//   diagonal: Dp - Sd from the linearisation partials; off-diagonal: -sum_j W_ij N_j W_kj^T - odom.
template <typename T>
__global__ __launch_bounds__(kBlock) void k_schur_blocks(int nnz, const int* __restrict__ blk_row, const int* __restrict__ blk_col,
                                                         const int* __restrict__ sptr, const uint32_t* __restrict__ slot_i,
                                                         const uint32_t* __restrict__ slot_k, const int* __restrict__ optr,
                                                         const uint32_t* __restrict__ oslot, Table<T> tb, const T* __restrict__ od_dyn,
                                                         size_t od_slots, const T* __restrict__ lmrec, const T* __restrict__ ps,
                                                         const T* __restrict__ part, HT<T>* __restrict__ A, int diag_on,
                                                         const uint32_t* __restrict__ od_idx, int odom_analytic, const int* __restrict__ which,
                                                         const int* __restrict__ lower_of, T diag_shift = T(0)) {
    const int t = blockIdx.x * kBlock + threadIdx.x;
    if (t >= nnz) return;
    const int b = which ? which[t] : t;      // which: the blocks on or above the diagonal; S is symmetric: each also writes its mirror (lower_of)
    const int i = blk_row[b], k = blk_col[b];
    HT<T>* o = A + (size_t)b * 9;
    if (i == k) {
        if (!diag_on) {       // edge-sharded: the (already all-reduced) diagonal is contributed by one rank only
#pragma unroll
            for (int m = 0; m < 9; ++m) o[m] = 0;
            return;
        }
        const T* p = part + (size_t)i * 18;
        const T m0 = p[0] - p[9], m1 = p[1] - p[10], m2 = p[2] - p[11], m3 = p[3] - p[12], m4 = p[4] - p[13], m5 = p[5] - p[14];
        const T up = T(1) + diag_shift;      // (engine: hier_shift) the hierarchy's copy of S with its diagonal raised: positive definite again where f32 rounding left it indefinite
        o[0] = m0 * up; o[1] = m1; o[2] = m2; o[3] = m1; o[4] = m3 * up; o[5] = m4; o[6] = m2; o[7] = m4; o[8] = m5 * up;
        return;
    }
    const T ci = ps[(size_t)i * 4 + 2], si = ps[(size_t)i * 4 + 3], ck = ps[(size_t)k * 4 + 2], sk = ps[(size_t)k * 4 + 3];
    const size_t S = tb.slots;
    T acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int q = sptr[b]; q < sptr[b + 1]; ++q) {
        const size_t e1 = slot_i[q], e2 = slot_k[q];
        const uint32_t l = tb.idx[e1];
        const auto ai = ld2<T>(tb.dyn + 2 * e1), pi = ld2<T>(tb.dyn + 2 * (S + e1));
        const auto ak = ld2<T>(tb.dyn + 2 * e2), pk = ld2<T>(tb.dyn + 2 * (S + e2));
        const T a0i = ai.x, a1i = ai.y, vi0 = pi.y, vi1 = -pi.x;
        const T a0k = ak.x, a1k = ak.y, vk0 = pk.y, vk1 = -pk.x;
        const T* lr = lmrec + (size_t)l * kLmRec;
        const T nxx = lr[2], nxy = lr[3], nyy = lr[4];
        const T m00 = nxx * ck + nxy * sk, m01 = nxy * ck - nxx * sk, m10 = nxy * ck + nyy * sk, m11 = nyy * ck - nxy * sk;
        const T g00 = ci * m00 + si * m10, g01 = ci * m01 + si * m11, g10 = ci * m10 - si * m00, g11 = ci * m11 - si * m01;
        const T q00 = a0i * g00 * a0k, q01 = a0i * g01 * a1k, q10 = a1i * g10 * a0k, q11 = a1i * g11 * a1k;
        const T u00 = ci * q00 - si * q10, u01 = ci * q01 - si * q11, u10 = si * q00 + ci * q10, u11 = si * q01 + ci * q11;
        acc[0] += u00 * ck - u01 * sk; acc[1] += u00 * sk + u01 * ck; acc[3] += u10 * ck - u11 * sk; acc[4] += u10 * sk + u11 * ck;
        const T qv0 = q00 * vk0 + q01 * vk1, qv1 = q10 * vk0 + q11 * vk1;
        acc[2] -= ci * qv0 - si * qv1; acc[5] -= si * qv0 + ci * qv1;
        const T vq0 = vi0 * q00 + vi1 * q10, vq1 = vi0 * q01 + vi1 * q11;
        acc[6] -= vq0 * ck - vq1 * sk; acc[7] -= vq0 * sk + vq1 * ck;
        acc[8] += vi0 * qv0 + vi1 * qv1;
    }
    T d[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};      // the odometry edges joining i and k: -diag(a) each, or the full H_12 / H_21 of the analytic Jacobians
    for (int q = optr[b]; q < optr[b + 1]; ++q) {
        const size_t e = oslot[q];
        if (odom_analytic) {      // general pose-pose slots (tsgo_math.h): the slot's own row block [[-K, c], [r^T, -kappa]]
            T h[PP_PLANES];
#pragma unroll
            for (int m = 0; m < PP_PLANES; ++m) h[m] = od_dyn[(size_t)m * od_slots + e];
            d[0] -= h[PP_K00]; d[1] -= h[PP_K01]; d[3] -= h[PP_K01]; d[4] -= h[PP_K11]; d[2] += h[PP_C0]; d[5] += h[PP_C1]; d[6] += h[PP_R0]; d[7] += h[PP_R1]; d[8] -= h[PP_KAPPA];
        } else { d[0] -= od_dyn[e]; d[4] -= od_dyn[od_slots + e]; d[8] -= od_dyn[2 * od_slots + e]; }
    }
#pragma unroll
    for (int m = 0; m < 9; ++m) o[m] = -acc[m] + d[m];
    if (lower_of) {
        const int lo = lower_of[b];
        if (lo >= 0) {
            HT<T>* q = A + (size_t)lo * 9;
#pragma unroll
            for (int x = 0; x < 3; ++x)
#pragma unroll
                for (int y = 0; y < 3; ++y) q[3 * x + y] = o[3 * y + x];
        }
    }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void k_block_inv(int n, const int* __restrict__ diag, const HT<T>* __restrict__ A, HT<T>* __restrict__ Dinv) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    T m[9], o[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) m[k] = A[(size_t)diag[i] * 9 + k];
    m3_inv<T>(m, o);
#pragma unroll
    for (int k = 0; k < 9; ++k) Dinv[(size_t)i * 9 + k] = o[k];
}

// P = Z - w Dinv (A Z), one thread per P block; Z_k = [[1,0,-ry],[0,1,rx],[0,0,1]] (rigid modes).
// PK: also writes the cycle-format words of the block in P's rows (Ppm) and, transposed, in R's rows (Rpm) — what k_to_planes did in two more
// launches per level.
template <typename T, int PK>
__global__ __launch_bounds__(kBlock) void k_prolongator(int nnzP, const int* __restrict__ p_row, const int* __restrict__ p_self,
                                                        const int* __restrict__ sptr, const int* __restrict__ sx, const int* __restrict__ sy,
                                                        const HT<T>* __restrict__ A, const HT<T>* __restrict__ Dinv, const T* __restrict__ rel,
                                                        T omega, HT<T>* __restrict__ P, const int* __restrict__ p_to_r, HT<T>* __restrict__ Rv,
                                                        const int* __restrict__ p_ptr, const int* __restrict__ p_col, const int* __restrict__ r_ptr,
                                                        uint32_t* __restrict__ Ppm, uint32_t* __restrict__ Rpm) {
    const int pb = blockIdx.x * kBlock + threadIdx.x;
    if (pb >= nnzP) return;
    const int i = p_row[pb];
    T acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int q = sptr[pb]; q < sptr[pb + 1]; ++q) {
        const int k = sy[q];
        const T zk[9] = {T(1), T(0), -rel[(size_t)k * 2 + 1], T(0), T(1), rel[(size_t)k * 2], T(0), T(0), T(1)};
        T a[9];
#pragma unroll
        for (int m = 0; m < 9; ++m) a[m] = A[(size_t)sx[q] * 9 + m];
        m3_mul_acc<T>(a, zk, acc);
    }
    T d[9], da[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int m = 0; m < 9; ++m) d[m] = Dinv[(size_t)i * 9 + m];
    m3_mul_acc<T>(d, acc, da);
    T o[9];
#pragma unroll
    for (int m = 0; m < 9; ++m) o[m] = -omega * da[m];
    // a node whose diagonal block is singular (a vertex no edge touches: Dinv was set to 0) stays out of the
    // coarse space, otherwise the cycle would inject a null-space component that nothing removes
    const bool dead = d[0] == T(0) && d[4] == T(0) && d[8] == T(0);
    if (p_self[pb] && !dead) { o[0] += T(1); o[4] += T(1); o[8] += T(1); o[2] -= rel[(size_t)i * 2 + 1]; o[5] += rel[(size_t)i * 2]; }
#pragma unroll
    for (int m = 0; m < 9; ++m) P[(size_t)pb * 9 + m] = o[m];
    // the same block transposed, stored in the order the restriction walks it (rows of R = P^T):
    // reading P through an index there cost 2.7x the bytes (profiles/r01c: 88 MB for a 33 MB operator)
    const int rb = p_to_r[pb];
    HT<T>* rt = Rv + (size_t)rb * 9;
    float vp[9], vr[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) { rt[3 * i + j] = o[3 * j + i]; vp[3 * i + j] = (float)(HT<T>)o[3 * i + j]; vr[3 * i + j] = (float)(HT<T>)o[3 * j + i]; }
    {
        const int p0 = p_ptr[i];
        cy_store<PK>(vp, Ppm + (size_t)p0 * cy_words<PK>() + (pb - p0), (size_t)(p_ptr[i + 1] - p0));
        const int a = p_col[pb], q0 = r_ptr[a];
        cy_store<PK>(vr, Rpm + (size_t)q0 * cy_words<PK>() + (rb - q0), (size_t)(r_ptr[a + 1] - q0));
    }
}

// out[o] = sum over its pair list of X[x] * Y[y]  (TRANS: X[x]^T * Y[y]).  Three lanes per output block, one per
// output row: per pair a lane reads one row (column when TRANS) of X and the whole 36-byte Y block — 14 loads for
// 9 multiply-adds, against 8 loads for 3 with a lane per element — and 21 output blocks share a wavefront.
template <typename T, int TRANS>
__global__ __launch_bounds__(kBlock) void k_pair_gemm(int n_out, const int* __restrict__ ptr, const int* __restrict__ px,
                                                      const int* __restrict__ py, const HT<T>* __restrict__ X, const HT<T>* __restrict__ Y,
                                                      HT<T>* __restrict__ out, const int* __restrict__ which) {
    const int lane = threadIdx.x & 63;
    if (lane >= 63) return;
    const int wave = (blockIdx.x * kBlock + threadIdx.x) >> 6;
    const int oo = wave * kPairBlocksPerWave + lane / 3, i = lane % 3;
    if (oo >= n_out) return;
    const int o = which ? which[oo] : oo;        // which: the output blocks that have a pair list (the others are mirrored)
    T acc0 = 0, acc1 = 0, acc2 = 0;
    // two pairs per trip, index loads first, then value loads, then arithmetic
    constexpr int U = 2;
    const int qe = ptr[o + 1];
    for (int q = ptr[o]; q < qe; q += U) {
        int xi[U], yi[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const int qq = q + u < qe ? q + u : qe - 1; xi[u] = px[qq]; yi[u] = py[qq]; }
        HT<T> a[U][3], b[U][9];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const HT<T>* pa = X + (size_t)xi[u] * 9; const HT<T>* pb = Y + (size_t)yi[u] * 9;
            if (TRANS) { a[u][0] = pa[i]; a[u][1] = pa[3 + i]; a[u][2] = pa[6 + i]; }
            else { a[u][0] = pa[3 * i]; a[u][1] = pa[3 * i + 1]; a[u][2] = pa[3 * i + 2]; }
#pragma unroll
            for (int m = 0; m < 9; ++m) b[u][m] = pb[m];
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (q + u < qe) {
                acc0 += T(a[u][0]) * T(b[u][0]) + T(a[u][1]) * T(b[u][3]) + T(a[u][2]) * T(b[u][6]);
                acc1 += T(a[u][0]) * T(b[u][1]) + T(a[u][1]) * T(b[u][4]) + T(a[u][2]) * T(b[u][7]);
                acc2 += T(a[u][0]) * T(b[u][2]) + T(a[u][1]) * T(b[u][5]) + T(a[u][2]) * T(b[u][8]);
            }
    }
    HT<T>* po = out + (size_t)o * 9 + 3 * i;
    po[0] = acc0; po[1] = acc1; po[2] = acc2;
}

// Same product for levels whose pair lists are long (smoothed Galerkin matrices get dense: hundreds of pairs
// per output block): one wavefront per output block, lanes stride over the pairs and keep nine partial
// sums each, xor-shuffle reduction at the end.  The nine-lane kernel above would walk such a list serially.
template <typename T, int TRANS, int LPB = 64>
__global__ __launch_bounds__(kBlock) void k_pair_gemm_wave(int n_out, const int* __restrict__ ptr, const int* __restrict__ px,
                                                           const int* __restrict__ py, const HT<T>* __restrict__ X, const HT<T>* __restrict__ Y,
                                                           HT<T>* __restrict__ out, const int* __restrict__ which) {
    // LPB lanes (64 or 16) share one output block
    const int sub = threadIdx.x % LPB;
    const int oo = (blockIdx.x * kBlock + threadIdx.x) / LPB;
    const bool live = oo < n_out;
    const int o = live ? (which ? which[oo] : oo) : 0;
    T acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (live)
        for (int q = ptr[o] + sub; q < ptr[o + 1]; q += LPB) {
            T a[9], b[9];
#pragma unroll
            for (int m = 0; m < 9; ++m) { a[m] = X[(size_t)px[q] * 9 + m]; b[m] = Y[(size_t)py[q] * 9 + m]; }
            if (TRANS) m3_tmul_acc<T>(a, b, acc); else m3_mul_acc<T>(a, b, acc);
        }
#pragma unroll
    for (int m = 0; m < 9; ++m) acc[m] = group_sum<T, LPB>(acc[m]);
    if (live && sub == 0) {
#pragma unroll
        for (int m = 0; m < 9; ++m) out[(size_t)o * 9 + m] = acc[m];
    }
}

// The Galerkin product P^T (A P) is symmetric: only its blocks on and above the diagonal are summed from pair
// lists; a block below it is the transpose of its mirror (host/amg.h: a_mirror).  One thread per element.
template <typename T>
__global__ __launch_bounds__(kBlock) void k_mirror_blocks(int n_blocks, const int* __restrict__ mirror, HT<T>* A) {
    const int t = blockIdx.x * kBlock + threadIdx.x;
    const int b = t / 9, e = t % 9;
    if (b >= n_blocks) return;
    const int m = mirror[b];
    if (m >= 0) A[(size_t)b * 9 + e] = A[(size_t)m * 9 + (e % 3) * 3 + e / 3];
}

// Dense inverse of the coarsest matrix in LDS (n <= 84), in-place Gauss-Jordan; SPD so no pivoting.
// One workgroup of 1024 threads as a 32 x 32 tile walking the matrix.
constexpr int kDenseThreads = 1024;
template <typename T>
__global__ __launch_bounds__(kDenseThreads) void k_dense_inverse(int nb, const int* __restrict__ ptr, const int* __restrict__ col,
                                                                 const HT<T>* __restrict__ A, T* __restrict__ inv) {
    __shared__ T M[kDenseMax * kDenseMax];
    __shared__ T colk[kDenseMax];
    const int n = nb * 3;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int e = threadIdx.x; e < n * n; e += kDenseThreads) M[e] = T(0);
    __syncthreads();
    for (int i = threadIdx.x; i < nb; i += kDenseThreads)
        for (int a = ptr[i]; a < ptr[i + 1]; ++a)
            for (int x = 0; x < 3; ++x)
                for (int y = 0; y < 3; ++y) M[(3 * i + x) * n + 3 * col[a] + y] = A[(size_t)a * 9 + 3 * x + y];
    __syncthreads();
    for (int k = 0; k < n; ++k) {
        const T piv = M[k * n + k];
        const T ip = (fabs(piv) > T(0)) ? T(1) / piv : T(0);
        if ((int)threadIdx.x < n) colk[threadIdx.x] = M[threadIdx.x * n + k];
        __syncthreads();
        if ((int)threadIdx.x < n) M[k * n + threadIdx.x] = ((int)threadIdx.x == k ? T(1) : M[k * n + threadIdx.x]) * ip;
        __syncthreads();
        for (int i = ty; i < n; i += 32) {
            if (i == k) continue;
            const T f = colk[i];
            for (int j = tx; j < n; j += 32) M[i * n + j] = (j == k ? T(0) : M[i * n + j]) - f * M[k * n + j];
        }
        __syncthreads();
    }
    for (int e = threadIdx.x; e < n * n; e += kDenseThreads) inv[e] = M[e];
}

// ---- cycle format --------------------------------------------------------------------------------------------
// The setup kernels address 3x3 blocks by block index (36 contiguous bytes, "AoS").  Read that way by the cycle kernels —
// a lane per block, nine loads — every load instruction of a wavefront touches 36-byte-strided words: 18+ cache lines
// for 256 useful bytes, and the texture addressers, not the memory system, set the pace (profiles/r01e: L1 tag pipes at
// 40 %).  The cycle therefore reads a COPY in which the blocks of one row are stored plane-major: word m of the j-th
// block of row i lives at W * ptr[i] + m * len_i + j, so that lanes j, j+1, ... load consecutive words.  One or two lines per
// load instruction.  The copy is written once per hierarchy build (k_to_planes, 8 lanes per row).  Two encodings:
//   PK = 0: W = 9 words, the block's nine f32 values.
//   PK = 1 (tsgo_config.cycle_storage = 16, default): W = 5 words = nine IEEE half floats + a power-of-two exponent (int16) common
//           to the block: value_k = half_k * 2^e, e chosen so that the block's largest entry sits in [2^14, 2^15) — eleven
//           significant bits relative to the block's own maximum whatever its magnitude (gauge blocks of 1e6 next to 1e-3).
//           20 bytes per block instead of 36 and five loads instead of nine: the sweeps of the big levels are byte-bound at
//           1 M poses and half byte-, half latency-bound at 100 k.  A block and its mirror (A_ij, A_ji^T) have the same maximum,
//           hence the same exponent and element-wise the same rounding: the rounded matrix is still symmetric, and because
//           every use inside the cycle (pre-sweeps, residual, post-sweeps; restriction and prolongation through R = P^T copies of
//           the same rounded blocks) reads THIS copy, the V-cycle stays a symmetric operator.  The setup (Galerkin products,
//           diagonal inverses) keeps the f32 blocks.
// (kCyWordsF32 / kCyWordsF16, cy_words<PK>() and cy_store<PK>() are defined at the top of this file: the set-up kernels write the format too)

// the j-th block of a row whose cycle-format words start at `base` (row length len): its words as stored (cy_fetch: loads only, so
// that several blocks' words can be requested before any is used), and the words as nine values of type T (cy_decode)
template <int PK> __device__ __forceinline__ void cy_fetch(const uint32_t* __restrict__ base, size_t len, size_t j, uint32_t* w) {
    const uint32_t* q = base + j;
#pragma unroll
    for (int m = 0; m < cy_words<PK>(); ++m) w[m] = q[(size_t)m * len];
}
template <typename T, int PK> __device__ __forceinline__ void cy_decode(const uint32_t* w, T* b) {
    if (PK) {
        const int e = (int)(short)(w[4] >> 16);
        const float sc = __uint_as_float((uint32_t)(e + 127) << 23);          // 2^e, e in [-126, 127] by construction
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            b[2 * m] = T(__half2float(__ushort_as_half((unsigned short)(w[m] & 0xffffu))) * sc);
            b[2 * m + 1] = T(__half2float(__ushort_as_half((unsigned short)(w[m] >> 16))) * sc);
        }
        b[8] = T(__half2float(__ushort_as_half((unsigned short)(w[4] & 0xffffu))) * sc);
    } else {
#pragma unroll
        for (int m = 0; m < 9; ++m) b[m] = T(__uint_as_float(w[m]));
    }
}
template <typename T, int PK> __device__ __forceinline__ void cy_load(const uint32_t* __restrict__ base, size_t len, size_t j, T* b) {
    uint32_t w[cy_words<PK>()];
    cy_fetch<PK>(base, len, j, w);
    cy_decode<T, PK>(w, b);
}

template <typename T, int PK>
__global__ __launch_bounds__(kBlock) void k_to_planes(int n_rows, const int* __restrict__ ptr, const HT<T>* __restrict__ src, uint32_t* __restrict__ dst) {
    const int g = (blockIdx.x * kBlock + threadIdx.x) / 8, sub = threadIdx.x % 8;
    if (g >= n_rows) return;
    const int p0 = ptr[g], len = ptr[g + 1] - p0;
    constexpr int W = cy_words<PK>();
    for (int j = sub; j < len; j += 8) {
        const HT<T>* b = src + (size_t)(p0 + j) * 9;
        uint32_t* o = dst + (size_t)p0 * W + j;
        float v[9];
#pragma unroll
        for (int m = 0; m < 9; ++m) v[m] = (float)b[m];
        cy_store<PK>(v, o, (size_t)len);
    }
}

// Round 4: the cycle-format copies are written by the kernels that PRODUCE the blocks — k_prolongator (P and R = P^T) and k_mirror_pack
// (the Galerkin matrix of the next level) — instead of by fourteen k_to_planes launches per hierarchy build (137 us at 100k poses).
// k_to_planes remains for the explicit level-0 matrix of tsgo_config.cycle_level0 = 1.
//
// The Galerkin product P^T (A P) is symmetric: only its blocks on and above the diagonal are summed from pair lists; a block below it is
// the transpose of its mirror (host/amg.h: a_mirror).  One thread per block: fills the mirrored block and writes the block's
// cycle-format words (row = the block's row, ptr = the matrix's row pointers).
template <typename T, int PK>
__global__ __launch_bounds__(kBlock) void k_mirror_pack(int n_blocks, const int* __restrict__ mirror, HT<T>* __restrict__ A, const int* __restrict__ row,
                                                        const int* __restrict__ ptr, uint32_t* __restrict__ Apm) {
    const int b = blockIdx.x * kBlock + threadIdx.x;
    if (b >= n_blocks) return;
    const int m = mirror[b];
    float v[9];
    if (m >= 0) {
#pragma unroll
        for (int x = 0; x < 3; ++x)
#pragma unroll
            for (int y = 0; y < 3; ++y) v[3 * x + y] = (float)A[(size_t)m * 9 + 3 * y + x];
#pragma unroll
        for (int k = 0; k < 9; ++k) A[(size_t)b * 9 + k] = v[k];
    } else {
#pragma unroll
        for (int k = 0; k < 9; ++k) v[k] = (float)A[(size_t)b * 9 + k];
    }
    if (Apm) {
        const int i = row[b], p0 = ptr[i];
        cy_store<PK>(v, Apm + (size_t)p0 * cy_words<PK>() + (b - p0), (size_t)(ptr[i + 1] - p0));
    }
}

// ---- V-cycle ---------------------------------------------------------------------------------------
// Early exit of a finished solve: every cycle kernel reads st->done.  The flag is REQUESTED first and TESTED only after the loads that
// depend on the kernel arguments alone (row bounds, the row's own entries) have been issued: tested at the very top it put one
// more scalar round trip (st -> done) in front of the first vector load of kernels whose whole life is four such trips.
// The coarse levels hold little work (12.5k / 1.5k / 196 block rows at 100k poses): what matters is the
// length of the dependent-load chain, so LPR lanes share one block row (one 3x3 block per lane per
// trip, 72 contiguous bytes per lane) and finish with an LPR-lane xor-shuffle sum.

// Block-row kernels of the big levels take the XCD-aware workgroup map of the table kernels (xcd_block(), tsgo_kernels.h): an XCD walks a
// contiguous eighth of the rows, so the vector entries a row gathers (its neighbours': nearby rows) are fetched into ONE L2 instead of all
// eight (PMC, round 3: k_restrict from level 0 fetched 1.74x its algorithmic bytes, k_prolong_add 1.48x).  xcd = 0: round-robin, as before.
__device__ __forceinline__ int lpr_block(int xcd) { return xcd ? xcd_block() : (int)blockIdx.x; }

// MODE 0: out = r - A z.   MODE 1: out = z + omega Dinv (r - A z)  (smoothing sweep).
// MODE 2: out = Dinv A z  (power iteration for the smoother's damping).
// PM: A is the cycle-format copy (above; PK = its encoding); otherwise the block-indexed f32 / HT matrix.
template <typename T, int LPR, int MODE, int PM = 1, int PK = 0>
__global__ __launch_bounds__(kBlock) void k_bcsr_residual(int n, const int* __restrict__ ptr, const int* __restrict__ col,
                                                          const void* __restrict__ Av, const T* __restrict__ r, const T* __restrict__ z,
                                                          const HT<T>* __restrict__ Dinv, T* __restrict__ out,
                                                          const T* __restrict__ omega_ptr, const CgState<T>* __restrict__ st, int xcd = 0) {
    const int done = MODE != 2 ? st->done : 0;
    const int g = (lpr_block(xcd) * kBlock + threadIdx.x) / LPR, sub = threadIdx.x % LPR;
    // LPR == 64: the row is wave-uniform, its bounds come through the scalar cache (one dependent round trip shorter)
    const int i = LPR == 64 ? __builtin_amdgcn_readfirstlane(g < n ? g : n - 1) : (g < n ? g : n - 1);
    T s0 = 0, s1 = 0, s2 = 0;
    int p0 = ptr[i];
    const int p1 = ptr[i + 1];
    issue_before_exit(p0);
    if (done) return;
    // What the row's epilogue needs (its right-hand side, its own entry of z, its diagonal inverse) depends on the row
    // alone: requested now, it arrives while the blocks are walked instead of adding a fourth dependent trip at the end.
    const bool head = g < n && sub == 0;
    T ri0 = 0, ri1 = 0, ri2 = 0, zi0 = 0, zi1 = 0, zi2 = 0;
    HT<T> d[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (head) {
        if (MODE != 2) { ri0 = r[(size_t)i * 3]; ri1 = r[(size_t)i * 3 + 1]; ri2 = r[(size_t)i * 3 + 2]; }
        if (MODE == 1) { zi0 = z[(size_t)i * 3]; zi1 = z[(size_t)i * 3 + 1]; zi2 = z[(size_t)i * 3 + 2]; }
        if (MODE != 0) {
#pragma unroll
            for (int m = 0; m < 9; ++m) d[m] = Dinv[(size_t)i * 9 + m];
        }
    }
    const size_t len = (size_t)(p1 - p0);
    const uint32_t* base = (const uint32_t*)Av + (size_t)p0 * cy_words<PK>();
    // (Two blocks per trip with every load of the pair issued before either is used — half the dependent round trips of a lane with
    // two to five blocks — was measured in round 3: 78 instead of 62 VGPRs, six resident waves per SIMD instead of eight, and the
    // sweeps of levels 1-2 got slower, 7.7 -> 8.4 us and 7.9 -> 8.7 us.  One block per trip it stays.)
    for (int a = p0 + sub; a < p1; a += LPR) {
        const T* v = z + (size_t)col[a] * 3;
        T b[9];
        if (PM) cy_load<T, PK>(base, len, (size_t)(a - p0), b);
        else {
            const HT<T>* q = (const HT<T>*)Av + (size_t)a * 9;
#pragma unroll
            for (int m = 0; m < 9; ++m) b[m] = q[m];
        }
        const T v0 = v[0], v1 = v[1], v2 = v[2];
        s0 += b[0] * v0 + b[1] * v1 + b[2] * v2; s1 += b[3] * v0 + b[4] * v1 + b[5] * v2; s2 += b[6] * v0 + b[7] * v1 + b[8] * v2;
    }
    s0 = group_sum<T, LPR>(s0); s1 = group_sum<T, LPR>(s1); s2 = group_sum<T, LPR>(s2);
    if (head) {
        if (MODE == 2) {
            out[(size_t)i * 3] = d[0] * s0 + d[1] * s1 + d[2] * s2; out[(size_t)i * 3 + 1] = d[3] * s0 + d[4] * s1 + d[5] * s2;
            out[(size_t)i * 3 + 2] = d[6] * s0 + d[7] * s1 + d[8] * s2;
            return;
        }
        const T e0 = ri0 - s0, e1 = ri1 - s1, e2 = ri2 - s2;
        if (MODE == 0) { out[(size_t)i * 3] = e0; out[(size_t)i * 3 + 1] = e1; out[(size_t)i * 3 + 2] = e2; }
        else {
            const T omega = *omega_ptr;
            out[(size_t)i * 3] = zi0 + omega * (d[0] * e0 + d[1] * e1 + d[2] * e2);
            out[(size_t)i * 3 + 1] = zi1 + omega * (d[3] * e0 + d[4] * e1 + d[5] * e2);
            out[(size_t)i * 3 + 2] = zi2 + omega * (d[6] * e0 + d[7] * e1 + d[8] * e2);
        }
    }
}

// out = A z with z stored at row stride `zs` (research, TSGO_CYCLE_EXPLICIT0: the explicit level-0 matrix in place of the
// implicit Schur product inside the cycle; z is the pose-record array zc, stride kPoseRec).  A plane-major (cycle format).
template <typename T, int LPR, int PK>
__global__ __launch_bounds__(kBlock) void k_bcsr_apply(int n, const int* __restrict__ ptr, const int* __restrict__ col, const uint32_t* __restrict__ A,
                                                       const T* __restrict__ z, int zs, T* __restrict__ out, const CgState<T>* __restrict__ st, int xcd = 0) {
    const int done = st->done;
    const int g = (lpr_block(xcd) * kBlock + threadIdx.x) / LPR, sub = threadIdx.x % LPR;
    const int i = g < n ? g : n - 1;
    T s0 = 0, s1 = 0, s2 = 0;
    int p0 = ptr[i];
    const int p1 = ptr[i + 1];
    issue_before_exit(p0);
    if (done) return;
    const size_t len = (size_t)(p1 - p0);
    const uint32_t* base = A + (size_t)p0 * cy_words<PK>();
    for (int a = p0 + sub; a < p1; a += LPR) {
        const T* v = z + (size_t)col[a] * zs;
        T b[9];
        cy_load<T, PK>(base, len, (size_t)(a - p0), b);
        const T v0 = v[0], v1 = v[1], v2 = v[2];
        s0 += b[0] * v0 + b[1] * v1 + b[2] * v2; s1 += b[3] * v0 + b[4] * v1 + b[5] * v2; s2 += b[6] * v0 + b[7] * v1 + b[8] * v2;
    }
    s0 = group_sum<T, LPR>(s0); s1 = group_sum<T, LPR>(s1); s2 = group_sum<T, LPR>(s2);
    if (g < n && sub == 0) { out[(size_t)i * 3] = s0; out[(size_t)i * 3 + 1] = s1; out[(size_t)i * 3 + 2] = s2; }
}

// rc = P^T v over the rows of R = P^T, and (when dinv_next is given) the next level's pre-smoothing
// z_next = Dinv_next rc in the same pass.  SUB: v = a - b (level 0: r - S z, never materialised).
template <typename T, int LPR, int SUB, int PK = 0>
__global__ __launch_bounds__(kBlock) void k_restrict(int n_agg, const int* __restrict__ rptr, const int* __restrict__ rcol,
                                                     const uint32_t* __restrict__ Rv, const T* __restrict__ va,
                                                     const T* __restrict__ vb, T* __restrict__ rc, const HT<T>* __restrict__ dinv_next,
                                                     T* __restrict__ z_next, const T* __restrict__ omega_ptr, const CgState<T>* __restrict__ st, int xcd = 0) {
    const int done = st->done;
    const int g = (lpr_block(xcd) * kBlock + threadIdx.x) / LPR, sub = threadIdx.x % LPR;
    const int a = g < n_agg ? g : n_agg - 1;
    T s0 = 0, s1 = 0, s2 = 0;
    int p0 = rptr[a];
    const int p1 = rptr[a + 1];
    issue_before_exit(p0);
    if (done) return;
    const bool head = g < n_agg && sub == 0;
    HT<T> dn[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};      // the next level's diagonal inverse: asked for before the walk, used after it
    T omega = 0;
    if (head && dinv_next) {
        omega = *omega_ptr;
#pragma unroll
        for (int m = 0; m < 9; ++m) dn[m] = dinv_next[(size_t)a * 9 + m];
    }
    const size_t len = (size_t)(p1 - p0);
    const uint32_t* base = Rv + (size_t)p0 * cy_words<PK>();          // cycle format
    for (int rb = p0 + sub; rb < p1; rb += LPR) {
        const size_t i = (size_t)rcol[rb] * 3;
        T b[9];
        cy_load<T, PK>(base, len, (size_t)(rb - p0), b);
        T x0 = va[i], x1 = va[i + 1], x2 = va[i + 2];
        if (SUB) { x0 -= vb[i]; x1 -= vb[i + 1]; x2 -= vb[i + 2]; }
        s0 += b[0] * x0 + b[1] * x1 + b[2] * x2; s1 += b[3] * x0 + b[4] * x1 + b[5] * x2; s2 += b[6] * x0 + b[7] * x1 + b[8] * x2;
    }
    s0 = group_sum<T, LPR>(s0); s1 = group_sum<T, LPR>(s1); s2 = group_sum<T, LPR>(s2);
    if (head) {
        rc[(size_t)a * 3] = s0; rc[(size_t)a * 3 + 1] = s1; rc[(size_t)a * 3 + 2] = s2;
        if (dinv_next) {
            z_next[(size_t)a * 3] = omega * (dn[0] * s0 + dn[1] * s1 + dn[2] * s2);
            z_next[(size_t)a * 3 + 1] = omega * (dn[3] * s0 + dn[4] * s1 + dn[5] * s2);
            z_next[(size_t)a * 3 + 2] = omega * (dn[6] * s0 + dn[7] * s1 + dn[8] * s2);
        }
    }
}

// z_i += sum_a P_ia e_a; z has row stride `zs` (3 on coarse levels, kPoseRec for zc)
template <typename T, int LPR, int PK = 0>
__global__ __launch_bounds__(kBlock) void k_prolong_add(int n, const int* __restrict__ pptr, const int* __restrict__ pcol,
                                                        const uint32_t* __restrict__ P, const T* __restrict__ e, T* __restrict__ z, int zs,
                                                        const CgState<T>* __restrict__ st, float* __restrict__ z32 = nullptr, int xcd = 0) {
    const int done = st->done;
    const int g = (lpr_block(xcd) * kBlock + threadIdx.x) / LPR, sub = threadIdx.x % LPR;
    const int i = g < n ? g : n - 1;
    T s0 = 0, s1 = 0, s2 = 0;
    int p0 = pptr[i];
    const int p1 = pptr[i + 1];
    issue_before_exit(p0);
    if (done) return;
    const bool head = g < n && sub == 0;
    T z0 = 0, z1 = 0, z2 = 0;                          // the entry this row adds to: read before the walk
    if (head) { z0 = z[(size_t)i * zs]; z1 = z[(size_t)i * zs + 1]; z2 = z[(size_t)i * zs + 2]; }
    const size_t len = (size_t)(p1 - p0);
    const uint32_t* base = P + (size_t)p0 * cy_words<PK>();           // cycle format
    for (int pb = p0 + sub; pb < p1; pb += LPR) {
        const T* v = e + (size_t)pcol[pb] * 3;
        T b[9];
        cy_load<T, PK>(base, len, (size_t)(pb - p0), b);
        const T v0 = v[0], v1 = v[1], v2 = v[2];
        s0 += b[0] * v0 + b[1] * v1 + b[2] * v2; s1 += b[3] * v0 + b[4] * v1 + b[5] * v2; s2 += b[6] * v0 + b[7] * v1 + b[8] * v2;
    }
    s0 = group_sum<T, LPR>(s0); s1 = group_sum<T, LPR>(s1); s2 = group_sum<T, LPR>(s2);
    if (head) {
        z[(size_t)i * zs] = z0 + s0; z[(size_t)i * zs + 1] = z1 + s1; z[(size_t)i * zs + 2] = z2 + s2;
        if (z32) { float* q = z32 + (size_t)i * zs; q[0] = (float)(z0 + s0); q[1] = (float)(z1 + s1); q[2] = (float)(z2 + s2); }     // level 0: the pose records' f32 copy
    }
}

// Bottom of the V-cycle in ONE workgroup of 1024 threads: restrict the last explicit level's residual
// (n <= 224 rows) to the dense level (<= 28 aggregates, 32 lanes each), apply the dense inverse,
// prolong the correction back (4 lanes per row).
template <typename T>
__global__ __launch_bounds__(kDenseThreads) void k_coarse_tail(int n, int n_agg, const int* __restrict__ rptr, const int* __restrict__ rcol,
                                                               const HT<T>* __restrict__ Rv, const int* __restrict__ pptr,
                                                               const int* __restrict__ pcol, const HT<T>* __restrict__ P, const T* __restrict__ res,
                                                               const T* __restrict__ inv, T* __restrict__ z, const CgState<T>* __restrict__ st) {
    const int done = st->done;
    __shared__ T rc[kDenseMax], zc_[kDenseMax];
    {
        const int a = threadIdx.x / 32, sub = threadIdx.x % 32;
        T s0 = 0, s1 = 0, s2 = 0;
        const int ac = a < n_agg ? a : n_agg - 1;
        int rb0 = rptr[ac];
        const int rb1 = rptr[ac + 1];
        issue_before_exit(rb0);
        if (done) return;
        if (a < n_agg)
            for (int rb = rb0 + sub; rb < rb1; rb += 32) {
                const HT<T>* b = Rv + (size_t)rb * 9; const size_t i = (size_t)rcol[rb] * 3;
                const T x0 = res[i], x1 = res[i + 1], x2 = res[i + 2];
                s0 += b[0] * x0 + b[1] * x1 + b[2] * x2; s1 += b[3] * x0 + b[4] * x1 + b[5] * x2; s2 += b[6] * x0 + b[7] * x1 + b[8] * x2;
            }
        s0 = group_sum<T, 32>(s0); s1 = group_sum<T, 32>(s1); s2 = group_sum<T, 32>(s2);
        if (a < n_agg && sub == 0) { rc[3 * a] = s0; rc[3 * a + 1] = s1; rc[3 * a + 2] = s2; }
    }
    __syncthreads();
    const int nd = n_agg * 3;
    if ((int)threadIdx.x < nd) { T s = 0; for (int j = 0; j < nd; ++j) s += inv[(size_t)threadIdx.x * nd + j] * rc[j]; zc_[threadIdx.x] = s; }
    __syncthreads();
    {
        const int i = threadIdx.x / 4, sub = threadIdx.x % 4;
        T s0 = 0, s1 = 0, s2 = 0;
        if (i < n)
            for (int pb = pptr[i] + sub; pb < pptr[i + 1]; pb += 4) {
                const HT<T>* b = P + (size_t)pb * 9; const T* v = zc_ + pcol[pb] * 3;
                s0 += b[0] * v[0] + b[1] * v[1] + b[2] * v[2]; s1 += b[3] * v[0] + b[4] * v[1] + b[5] * v[2]; s2 += b[6] * v[0] + b[7] * v[1] + b[8] * v[2];
            }
        s0 = group_sum<T, 4>(s0); s1 = group_sum<T, 4>(s1); s2 = group_sum<T, 4>(s2);
        if (i < n && sub == 0) { z[(size_t)i * 3] += s0; z[(size_t)i * 3 + 1] += s1; z[(size_t)i * 3 + 2] += s2; }
    }
}

// ---- the bottom of the V-cycle as ONE dense operator (round 4) ---------------------------------------------------------------------
// On the last explicit level (n <= 256 block rows; 122 at 100k poses) the cycle is: pre-sweep z1 = W r (W = omega D^-1), coarse
// correction through the dense inverse C of the level below, z2 = z1 + P C P^T (r - A z1), post-sweep z3 = z2 + W (r - A z2).  That is a
// LINEAR map of r, and with S = I - W A, E = S P it reads
//     z3 = B r,   B = W + S W + E C E^T                      (symmetric: W A W, E C E^T and W are)
// so it is formed once per hierarchy build (five tiny launches, dense arithmetic in T, 0.5 (B + B^T) stored as f32: exactly symmetric)
// and APPLIED as one dense matrix-vector product per PCG iteration — instead of residual, restrict + dense solve + prolong (one workgroup,
// ~10 us) and post-sweep: three dependent launches, 17-19 us of the 217 an iteration takes at 100k poses, for the same operator.
// The level's matrix has 92 blocks per row of 122: it IS dense.
template <typename T>
__global__ __launch_bounds__(kBlock) void k_bottom_scatter(int nnzA, const int* __restrict__ a_row, const int* __restrict__ a_col, const HT<T>* __restrict__ A,
                                                           const HT<T>* __restrict__ Dinv, const T* __restrict__ omega_ptr, int nnzP, const int* __restrict__ p_row,
                                                           const int* __restrict__ p_col, const HT<T>* __restrict__ P, int n3, int nd, T* __restrict__ Sd, T* __restrict__ Pd) {
    const int t = blockIdx.x * kBlock + threadIdx.x;
    if (t < nnzA) {           // S = I - omega D^-1 A, block (i, j)
        const int i = a_row[t], j = a_col[t];
        const T w = *omega_ptr;
        T d[9], a[9], m[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < 9; ++k) { d[k] = Dinv[(size_t)i * 9 + k]; a[k] = A[(size_t)t * 9 + k]; }
        m3_mul_acc<T>(d, a, m);
#pragma unroll
        for (int x = 0; x < 3; ++x)
#pragma unroll
            for (int y = 0; y < 3; ++y) Sd[(size_t)(3 * i + x) * n3 + 3 * j + y] = (i == j && x == y ? T(1) : T(0)) - w * m[3 * x + y];
    } else if (t - nnzA < nnzP) {
        const int pb = t - nnzA, i = p_row[pb], c = p_col[pb];
#pragma unroll
        for (int x = 0; x < 3; ++x)
#pragma unroll
            for (int y = 0; y < 3; ++y) Pd[(size_t)(3 * i + x) * nd + 3 * c + y] = T(P[(size_t)pb * 9 + 3 * x + y]);
    }
}

// C[M x N] = A[M x K] * B, B given as [K x N] (TB = 0) or as [N x K] (TB = 1: C = A B^T); row-major, 16 x 16 tiles through LDS.
template <typename T, int TB>
__global__ __launch_bounds__(256) void k_small_gemm(int M, int N, int K, const T* __restrict__ A, int lda, const T* __restrict__ B, int ldb, T* __restrict__ C, int ldc) {
    __shared__ T sa[16][17], sb[16][17];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int row = blockIdx.y * 16 + ty, col = blockIdx.x * 16 + tx;
    T acc = 0;
    for (int k0 = 0; k0 < K; k0 += 16) {
        sa[ty][tx] = (row < M && k0 + tx < K) ? A[(size_t)row * lda + k0 + tx] : T(0);
        if (TB) { const int n = blockIdx.x * 16 + ty; sb[tx][ty] = (n < N && k0 + tx < K) ? B[(size_t)n * ldb + k0 + tx] : T(0); }      // sb[k][n]
        else sb[ty][tx] = (k0 + ty < K && col < N) ? B[(size_t)(k0 + ty) * ldb + col] : T(0);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; ++k) acc += sa[ty][k] * sb[k][tx];
        __syncthreads();
    }
    if (row < M && col < N) C[(size_t)row * ldc + col] = acc;
}

// Bf = f32(0.5 (B + B^T)), B = W + S W + G with G = E C E^T already in Bd; W = omega D^-1 block diagonal.  One thread per entry; the
// entry and its mirror are both evaluated by the thread (and again, in the other order, by the mirror's thread: a + b == b + a).
template <typename T>
__global__ __launch_bounds__(kBlock) void k_bottom_finish(int n3, const T* __restrict__ Sd, const HT<T>* __restrict__ Dinv, const T* __restrict__ omega_ptr,
                                                          const T* __restrict__ Bd, float* __restrict__ Bf, T* __restrict__ Bs) {
    const int t = blockIdx.x * kBlock + threadIdx.x;
    if (t >= n3 * n3) return;
    const int r = t / n3, c = t % n3;
    const T w = *omega_ptr;
    auto entry = [&](int rr, int cc) {
        const int jb = cc / 3, y = cc % 3;
        T v = Bd[(size_t)rr * n3 + cc];
#pragma unroll
        for (int k = 0; k < 3; ++k) v += Sd[(size_t)rr * n3 + 3 * jb + k] * (w * T(Dinv[(size_t)jb * 9 + 3 * k + y]));
        if (rr / 3 == jb) v += w * T(Dinv[(size_t)jb * 9 + 3 * (rr % 3) + y]);
        return v;
    };
    const T v = T(0.5) * (entry(r, c) + entry(c, r));
    Bf[(size_t)r * n3 + c] = (float)v;
    if (Bs) Bs[(size_t)r * n3 + c] = v;
}

// z = B r for a dense f32 matrix B [n_rows x n_cols], one wavefront per row (every row a few coalesced passes): the last level's
// whole cycle in one launch (B = the level's operator), or t = E^T r of the level above it (B = E^T, below)
template <typename T>
__global__ __launch_bounds__(kBlock) void k_bottom_apply(int n_rows, int n_cols, const float* __restrict__ Bf, const T* __restrict__ r, T* __restrict__ z, const CgState<T>* __restrict__ st) {
    const int done = st->done;
    const int row = (blockIdx.x * kBlock + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const int rc = row < n_rows ? row : n_rows - 1;
    const float* b = Bf + (size_t)rc * n_cols;
    T acc = 0;
    float b0 = lane < n_cols ? b[lane] : 0.f;
    issue_before_exit(b0);
    if (done) return;
    if (lane < n_cols) acc = T(b0) * r[lane];
    for (int c = lane + 64; c < n_cols; c += 64) acc += T(b[c]) * r[c];
    acc = wave_sum<T>(acc);
    if (row < n_rows && lane == 0) z[row] = acc;
}

// ---- ... and the level ABOVE it in factored form -----------------------------------------------------------------------------------
// With B_b the dense operator of the bottom level b, the V(1,1) cycle of the level q above it is the linear map
//     z = W r + S W r + E B_b E^T r,      W = omega D^-1,  S = I - W A,  E = S P            (q's own W, A, P)
// and with z1 = W r (the pre-sweep the restriction INTO q already leaves) and t = E^T r it is applied as
//     z = 2 z1 - W (A z1) + G t,          G = E B_b    (dense [3 n_q x 3 n_b], formed once per hierarchy build)
// — two launches (t = E^T r: k_bottom_apply on E^T;  k_tail_up) for what was residual, restrict, the bottom's cycle, prolong and
// post-sweep: five launches with B_b, seven before it.  E = P - W (A P) needs no pattern of A P: the level's block rows times dense P.
template <typename T>
__global__ __launch_bounds__(kBlock) void k_scatter_blocks(int nnz, const int* __restrict__ row, const int* __restrict__ col, const HT<T>* __restrict__ blk, int ld, float* __restrict__ out) {
    const int b = blockIdx.x * kBlock + threadIdx.x;
    if (b >= nnz) return;
    const int i = row[b], c = col[b];
#pragma unroll
    for (int x = 0; x < 3; ++x)
#pragma unroll
        for (int y = 0; y < 3; ++y) out[(size_t)(3 * i + x) * ld + 3 * c + y] = (float)blk[(size_t)b * 9 + 3 * x + y];
}

// E -= omega D^-1 T over the blocks of T = A P (computed by the Galerkin setup: Tv), E holding the scattered P on entry; one wavefront
// per block row, a lane per block of the row
template <typename T>
__global__ __launch_bounds__(kBlock) void k_tail_E(int n, const int* __restrict__ tptr, const int* __restrict__ tcol, const HT<T>* __restrict__ Tv, const HT<T>* __restrict__ Dinv,
                                                   const T* __restrict__ omega_ptr, int nd, float* __restrict__ E) {
    const int i = (blockIdx.x * kBlock + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (i >= n) return;
    const T w = *omega_ptr;
    T d[9];
#pragma unroll
    for (int m = 0; m < 9; ++m) d[m] = T(Dinv[(size_t)i * 9 + m]);
    for (int tb = tptr[i] + lane; tb < tptr[i + 1]; tb += 64) {
        const int c = tcol[tb];
        T t[9], o[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int m = 0; m < 9; ++m) t[m] = T(Tv[(size_t)tb * 9 + m]);
        m3_mul_acc<T>(d, t, o);
#pragma unroll
        for (int x = 0; x < 3; ++x)
#pragma unroll
            for (int y = 0; y < 3; ++y) { float* e = E + (size_t)(3 * i + x) * nd + 3 * c + y; *e = (float)(T(*e) - w * o[3 * x + y]); }
    }
}

// C[M x N] = A[M x K] B[K x N], row-major f32: 32 x 64 tile per workgroup (2 x 4 outputs per thread), K in steps of 16 through LDS, the
// next step's operands in registers while this one is multiplied.  G = E B of the factored level: 2733 x 366 x 366 at 100k poses
// (516 workgroups); the f64 64 x 64 version without the prefetch took 115 us, a tenth of a hierarchy build.
__global__ __launch_bounds__(256) void k_gemm_f32(int M, int N, int K, const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb, float* __restrict__ C, int ldc) {
    __shared__ float sa[16][33], sb[16][65];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int row0 = blockIdx.y * 32, col0 = blockIdx.x * 64;
    float acc[2][4] = {};
    float pa[2], pb[4];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int q = 0; q < 2; ++q) { const int e = threadIdx.x + 256 * q, ar = e >> 4, ak = e & 15; pa[q] = (row0 + ar < M && k0 + ak < K) ? A[(size_t)(row0 + ar) * lda + k0 + ak] : 0.f; }
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int e = threadIdx.x + 256 * q, bk = e >> 6, bc = e & 63; pb[q] = (k0 + bk < K && col0 + bc < N) ? B[(size_t)(k0 + bk) * ldb + col0 + bc] : 0.f; }
    };
    fetch(0);
    for (int k0 = 0; k0 < K; k0 += 16) {
#pragma unroll
        for (int q = 0; q < 2; ++q) { const int e = threadIdx.x + 256 * q; sa[e & 15][e >> 4] = pa[q]; }
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int e = threadIdx.x + 256 * q; sb[e >> 6][e & 63] = pb[q]; }
        __syncthreads();
        if (k0 + 16 < K) fetch(k0 + 16);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const float a0 = sa[k][ty * 2], a1 = sa[k][ty * 2 + 1];
            float b[4];
#pragma unroll
            for (int v = 0; v < 4; ++v) b[v] = sb[k][tx * 4 + v];
#pragma unroll
            for (int v = 0; v < 4; ++v) { acc[0][v] += a0 * b[v]; acc[1][v] += a1 * b[v]; }
        }
        __syncthreads();
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int r = row0 + ty * 2 + u, c = col0 + tx * 4 + v;
            if (r < M && c < N) C[(size_t)r * ldc + c] = acc[u][v];
        }
}

// the run-time copy of E^T: Etf [nd x n3] = E^T
__global__ __launch_bounds__(kBlock) void k_transpose_f32(int n3, int nd, const float* __restrict__ E, float* __restrict__ Etf) {
    const int t = blockIdx.x * kBlock + threadIdx.x;
    if (t >= n3 * nd) return;
    const int i = t / nd, m = t % nd;
    Etf[(size_t)m * n3 + i] = E[t];
}

// z = B r for a dense f32 matrix with LONG rows (n_cols in the thousands): one workgroup per row, every load of a thread issued
// before its first use, block reduction.  (One wavefront per row, k_bottom_apply, walks such a row in 43 dependent trips: 21 us.)
template <typename T>
__global__ __launch_bounds__(kBlock) void k_rowdot_wg(int n_rows, int n_cols, const float* __restrict__ Bf, const T* __restrict__ r, T* __restrict__ z, const CgState<T>* __restrict__ st) {
    __shared__ T red[kWavesPerBlock];
    const int done = st->done;
    const int row = blockIdx.x;
    const float* b = Bf + (size_t)row * n_cols;
    constexpr int U = 12;                                    // 3 072 columns per trip: a level of <= 1 024 block rows is ONE trip
    T acc = 0;
    float b0 = (int)threadIdx.x < n_cols ? b[threadIdx.x] : 0.f;
    issue_before_exit(b0);
    if (done) return;                                        // workgroup-uniform
    for (int c0 = threadIdx.x; c0 < n_cols; c0 += U * kBlock) {
        float bv[U]; T rv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const int c = c0 + u * kBlock; const bool in = c < n_cols; bv[u] = in ? b[c] : 0.f; rv[u] = in ? r[c] : T(0); }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += T(bv[u]) * rv[u];
    }
    const T total = block_sum<T>(acc, red);
    if (threadIdx.x == 0) z[row] = total;
}

// z_i = 2 z1_i - omega D_i^-1 sum_j A_ij z1_j + sum_m G[3i..3i+2][m] t[m]: one WORKGROUP per block row (A in the cycle format): the
// row's blocks a lane each, the three dense rows of G spread over all 256 threads
template <typename T, int PK>
__global__ __launch_bounds__(kBlock) void k_tail_up(int n, const int* __restrict__ ptr, const int* __restrict__ col, const uint32_t* __restrict__ Apm, const HT<T>* __restrict__ Dinv,
                                                    const T* __restrict__ omega_ptr, const T* __restrict__ z1, int nd, const float* __restrict__ Gf, const T* __restrict__ t,
                                                    T* __restrict__ z, const CgState<T>* __restrict__ st) {
    __shared__ T red6[kWavesPerBlock * 6];
    const int done = st->done;
    const int i = blockIdx.x;
    int p0 = ptr[i];
    const int p1 = ptr[i + 1];
    issue_before_exit(p0);
    if (done) return;                                        // workgroup-uniform
    // the dense part first: its operands depend on the arguments alone
    const float* g0 = Gf + (size_t)i * 3 * nd;
    T d0 = 0, d1 = 0, d2 = 0;
    for (int m = threadIdx.x; m < nd; m += kBlock) { const T tm = t[m]; d0 += T(g0[m]) * tm; d1 += T(g0[nd + m]) * tm; d2 += T(g0[2 * (size_t)nd + m]) * tm; }
    T s0 = 0, s1 = 0, s2 = 0;
    const size_t len = (size_t)(p1 - p0);
    const uint32_t* base = Apm + (size_t)p0 * cy_words<PK>();
    for (int a = p0 + threadIdx.x; a < p1; a += kBlock) {
        const T* v = z1 + (size_t)col[a] * 3;
        T b[9];
        cy_load<T, PK>(base, len, (size_t)(a - p0), b);
        const T v0 = v[0], v1 = v[1], v2 = v[2];
        s0 += b[0] * v0 + b[1] * v1 + b[2] * v2; s1 += b[3] * v0 + b[4] * v1 + b[5] * v2; s2 += b[6] * v0 + b[7] * v1 + b[8] * v2;
    }
    // six sums, ONE barrier: per-wave shuffles, then thread 0 adds the four waves' partials
    s0 = wave_sum<T>(s0); s1 = wave_sum<T>(s1); s2 = wave_sum<T>(s2); d0 = wave_sum<T>(d0); d1 = wave_sum<T>(d1); d2 = wave_sum<T>(d2);
    if ((threadIdx.x & 63) == 0) { T* q = red6 + (threadIdx.x >> 6) * 6; q[0] = s0; q[1] = s1; q[2] = s2; q[3] = d0; q[4] = d1; q[5] = d2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        T v[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) { v[k] = red6[k]; for (int wv = 1; wv < kWavesPerBlock; ++wv) v[k] += red6[wv * 6 + k]; }
        const T w = *omega_ptr;
        const HT<T>* d = Dinv + (size_t)i * 9;
        const T* zi = z1 + (size_t)i * 3;
        z[(size_t)i * 3] = T(2) * zi[0] - w * (T(d[0]) * v[0] + T(d[1]) * v[1] + T(d[2]) * v[2]) + v[3];
        z[(size_t)i * 3 + 1] = T(2) * zi[1] - w * (T(d[3]) * v[0] + T(d[4]) * v[1] + T(d[5]) * v[2]) + v[4];
        z[(size_t)i * 3 + 2] = T(2) * zi[2] - w * (T(d[6]) * v[0] + T(d[7]) * v[1] + T(d[8]) * v[2]) + v[5];
    }
}

// coarsest level alone (graphs with a single explicit level): z = inv r, single workgroup
template <typename T>
__global__ __launch_bounds__(kBlock) void k_dense_apply(int n, const T* __restrict__ inv, const T* __restrict__ r, T* __restrict__ z,
                                                        const CgState<T>* __restrict__ st) {
    if (st->done) return;
    __shared__ T rv[kDenseMax];
    for (int j = threadIdx.x; j < n; j += kBlock) rv[j] = r[j];
    __syncthreads();
    const int i = threadIdx.x;
    if (i < n) { T s = 0; for (int j = 0; j < n; ++j) s += inv[(size_t)i * n + j] * rv[j]; z[i] = s; }
}

// level 0 smoothing with the symmetric 3x3 inverse blocks of the Schur diagonal (minv, 6 per pose):
//   MODE 0: zc = Minv r              (pre-smoothing, zero guess)
//   MODE 1: zc += Minv (r - s)       (post-smoothing; s = S zc from the implicit passes)
template <typename T, int MODE>
__global__ __launch_bounds__(kBlock) void k_smooth0(int P, const T* __restrict__ minv, const T* __restrict__ r, const T* __restrict__ s,
                                                    T* __restrict__ zc, const T* __restrict__ omega_ptr, const CgState<T>* __restrict__ st) {
    const int done = st->done;
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= P) return;
    T e0 = r[(size_t)i * 3], e1 = r[(size_t)i * 3 + 1], e2 = r[(size_t)i * 3 + 2];
    issue_before_exit(e0);
    if (done) return;
    if (MODE == 1) { e0 -= s[(size_t)i * 3]; e1 -= s[(size_t)i * 3 + 1]; e2 -= s[(size_t)i * 3 + 2]; }
    T z0, z1, z2;
    sym3_mul<T>(minv + (size_t)i * 6, e0, e1, e2, z0, z1, z2);
    const T w = *omega_ptr;
    T* zr = zc + (size_t)i * kPoseRec;
    if (MODE == 0) { zr[0] = w * z0; zr[1] = w * z1; zr[2] = w * z2; } else { zr[0] += w * z0; zr[1] += w * z1; zr[2] += w * z2; }
}

// power iteration helpers: a fixed pseudo-random start vector and ||v||^2 partials
template <typename T>
__global__ __launch_bounds__(kBlock) void k_seed_vector(int n3, T* __restrict__ v) {
    const int k = blockIdx.x * kBlock + threadIdx.x;
    if (k < n3) v[k] = sin(T(0.37) * T(k)) + T(0.1);
}
template <typename T>
__global__ __launch_bounds__(kBlock) void k_norm2(int n3, const T* __restrict__ v, T* __restrict__ part) {
    __shared__ T red[kWavesPerBlock];
    T g = 0;
    for (int k = blockIdx.x * kBlock + threadIdx.x; k < n3; k += gridDim.x * kBlock) g += v[k] * v[k];
    const T total = block_sum<T>(g, red);
    if (threadIdx.x == 0) part[blockIdx.x] = total;
}

// PCG vector step when the preconditioner is applied by separate kernels (the V-cycle):
//   gamma = (r, z), delta = (S z, z) arrive as partials; p = z + beta p, q = S z + beta q,
//   x += alpha p, r -= alpha q.  z for the next iteration comes from the next V-cycle.
// Stopping rule: sqrt(r^T D^-1 r) <= tol * sqrt(b^T D^-1 b), D = the 3x3 block diagonal of S (minv = D^-1; the b-side partials
// were left in bpart by k_pose_finalize) — the rule block-Jacobi PCG applies as well, and a norm the multigrid operator has no
// part in.  It is evaluated on the residual THIS step produces: the level-0 pre-smoothing below computes D^-1 r anyway, every
// workgroup leaves its partial of r^T D^-1 r, and k_iter_gate — one workgroup, first kernel of the next iteration — sums them
// and sets `done`: a converged solve does not run one more V-cycle and product just to learn that it was finished (the rule on
// r^T M^-1 r could only be tested after applying M^-1).  (The sum inside this kernel by the last-arriving workgroup was measured:
// its agent-scope release drains the 24 MB this kernel has just written, 12 -> 30 us; profiles/r03b_*.)
// Large graphs: every workgroup of k_cg_step sums ALL the partials of the two dot products (3 900 each at a million poses: 62 KB per
// workgroup, 244 MB over the launch — as much as the vectors it updates).  Above kFoldAbove partials they are first folded to kFoldOut
// per sum, in fixed order (chunk by chunk), by this kernel: out[w * kFoldOut + b] = sum of chunk b of array w.
constexpr int kFoldAbove = 1024, kFoldOut = 64;
template <typename T>
__global__ __launch_bounds__(kBlock) void k_fold_partials(int n, const T* __restrict__ a0, const T* __restrict__ a1, T* __restrict__ out) {
    __shared__ T red[kWavesPerBlock];
    const T* a = blockIdx.y == 0 ? a0 : a1;
    const int chunk = (n + kFoldOut - 1) / kFoldOut;
    const int b0 = min(n, (int)blockIdx.x * chunk), b1 = min(n, b0 + chunk);
    const T total = block_sum_array<T>(a + b0, b1 - b0, red);
    if (threadIdx.x == 0) out[blockIdx.y * kFoldOut + blockIdx.x] = total;
}

template <typename T>
__global__ __launch_bounds__(kBlock) void k_cg_step(int P, const T* __restrict__ sz, const T* __restrict__ dot_part,
                                                    const T* __restrict__ rz_part, int n_part, const CgState<T>* __restrict__ st_in,
                                                    CgState<T>* __restrict__ st_out, T* __restrict__ r, T* __restrict__ p,
                                                    T* __restrict__ q, T* __restrict__ x, T* __restrict__ zc,
                                                    const T* __restrict__ minv, const T* __restrict__ omega_ptr, T tol2, int max_iters,
                                                    const T* __restrict__ gamma0_scale, int stall_iter, T stall_ratio,
                                                    T* __restrict__ rdr_part, float* __restrict__ zc32) {
    __shared__ T red[kWavesPerBlock];
    const CgState<T> s = *st_in;
    const bool writer = blockIdx.x == 0 && threadIdx.x == 0;
    if (s.done) { if (writer) *st_out = s; return; }
    const T delta = block_sum_array<T>(dot_part, n_part, red);
    const T gamma = block_sum_array<T>(rz_part, n_part, red);
    const T gamma0 = s.iters == 0 ? gamma * (*gamma0_scale) : s.gamma0;
    CgState<T> n = s; n.gamma0 = gamma0;
    // gamma = r^T M^-1 r < 0 (or NaN) means the preconditioner is not positive definite: breakdown, NOT convergence (the host then
    // repeats the solve with block-Jacobi); gamma == 0: M^-1 r vanished, nothing left to correct; the cap: unconverged
    if (!(gamma > T(0)) || s.iters >= max_iters) {
        n.done = 1; n.fail = (gamma != gamma || gamma < T(0)) ? 1 : (gamma > T(0) ? 2 : 0);
        if (writer) *st_out = n;
        return;
    }
    // stagnation: by stall_iter the multigrid-preconditioned iteration has not reduced r^T M^-1 r below stall_ratio of its start
    if (s.iters == stall_iter && gamma > stall_ratio * gamma0) { n.done = 1; n.fail = 3; if (writer) *st_out = n; return; }
    T beta, alpha;
    if (s.iters == 0) { beta = 0; alpha = gamma / delta; }
    else { beta = gamma / s.gamma_old; alpha = gamma / (delta - beta * gamma / s.alpha_old); }
    if (!(alpha > 0) || !(alpha < T(1e300))) { n.done = 1; n.fail = 1; if (writer) *st_out = n; return; }
    // The vector updates element by element (lane = element: every load and store of a wavefront is one contiguous 512 bytes; lane = pose
    // made them 24-byte-strided triples), the residual and the diagonal inverses handed to the per-pose part through LDS.
    __shared__ T s_r[3 * kBlock];
    __shared__ T s_m[6 * kBlock];
    const int i0 = blockIdx.x * kBlock;
    const int n_here = min(kBlock, P - i0);
    {
        const size_t e0 = (size_t)i0 * 3;
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            const int le = (int)threadIdx.x + m * kBlock;
            if (le < 3 * n_here) {
                const size_t j = e0 + le;
                const int ip = le / 3, k = le - 3 * ip;
                const T pk = zc[(size_t)(i0 + ip) * kPoseRec + k] + beta * p[j];
                const T qk = sz[j] + beta * q[j];
                p[j] = pk; q[j] = qk; x[j] += alpha * pk;
                const T rk = r[j] - alpha * qk;
                r[j] = rk; s_r[le] = rk;
            }
        }
        const T* mi = minv + (size_t)i0 * 6;
#pragma unroll
        for (int m = 0; m < 6; ++m) {
            const int le = (int)threadIdx.x + m * kBlock;
            if (le < 6 * n_here) s_m[le] = mi[le];
        }
    }
    __syncthreads();
    const int i = i0 + threadIdx.x;
    T g = 0;
    if (i < P) {
        const T rr[3] = {s_r[3 * threadIdx.x], s_r[3 * threadIdx.x + 1], s_r[3 * threadIdx.x + 2]};
        // level-0 pre-smoothing of the NEXT V-cycle (zero initial guess): zc = Minv r
        T z0, z1, z2;
        sym3_mul<T>(s_m + 6 * threadIdx.x, rr[0], rr[1], rr[2], z0, z1, z2);
        const T w = *omega_ptr;
        T* zr = zc + (size_t)i * kPoseRec;
        zr[0] = w * z0; zr[1] = w * z1; zr[2] = w * z2;
        if (zc32) { float* zq = zc32 + (size_t)i * kPoseRec; zq[0] = (float)(w * z0); zq[1] = (float)(w * z1); zq[2] = (float)(w * z2); }
        g = rr[0] * z0 + rr[1] * z1 + rr[2] * z2;
    }
    const T total = block_sum<T>(g, red);
    if (threadIdx.x == 0) rdr_part[blockIdx.x] = total;
    if (writer) { n.gamma_old = gamma; n.alpha_old = alpha; n.iters = s.iters + 1; *st_out = n; }
}

// First kernel of every multigrid-preconditioned PCG iteration, one workgroup: has the residual the last k_cg_step produced met the
// stopping rule?  (Round 4: on the default path the test rides in workgroup 0 of the iteration's first product kernel instead —
// iter_gate_body in tsgo_kernels.h, k_schur_lm's `gate` argument — and this kernel runs only where that product is not the first launch.)  Then this and every later kernel of the solve exits at once (they all read st->done).
// host_flag (pinned host memory, eager launches only): the gate of the seq-th launched iteration tells the host thread that it has
// run and what it saw — one 64-bit word: seq | done | fail | iterations completed — so that the host
// neither predicts how many iterations a solve will take nor drains the stream to find out (Engine::do_solve_paced).
template <typename T>
__global__ __launch_bounds__(kBlock) void k_iter_gate(CgState<T>* __restrict__ st, const T* __restrict__ rdr_part, const T* __restrict__ bpart, int n, T tol2,
                                                      int* __restrict__ host_flag, int seq) {
    __shared__ T red[kWavesPerBlock];
    iter_gate_body<T>(GateArgs<T>{st, rdr_part, bpart, n, tol2, host_flag, seq}, red);
}

}  // namespace tsgo
