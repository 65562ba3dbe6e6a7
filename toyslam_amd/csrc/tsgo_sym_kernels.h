// tsgo_sym_kernels.h — the multigrid hierarchy's SYMBOLIC products on the device (round 3).
//
// Every numeric product of the hierarchy build is a gather over a precomputed list of (x block, y block) pairs per output
// block (host/amg.h: PairList).  Rounds 1-2 built those lists on the host (host/amg.cpp: spgemm_sym, count pass + fill pass on
// 16 threads): 43 M entries and 350 MB of uploads at 100k poses, 53 of the 140 ms that tsgo_set_graph spends on the critical
// path of a new structure.  Here the same lists are built where they are used:
//
//   Z = X * Y (patterns), X: n rows, Y: any;  optionally UPPER: only blocks on or above the diagonal get a pair list (the
//   symmetric Galerkin product P^T (A P)), a block below it gets the index of its mirror instead.
//
//   k_sym_count   one wavefront per row: the distinct columns (a hash set in LDS) and the number of pairs of the row
//   (exclusive scans of both: k_sym_scan)
//   k_sym_fill    one wavefront per row: distinct columns again, sorted ascending -> Z.col; pairs per column, scanned -> the
//                 list offsets; then the pairs themselves, X blocks one after the other (a Y row has every column once, so the
//                 lanes that share an X block never meet in one output block): within an output block the pairs are in (x, y)
//                 walking order — the order the host builder produces, so the two builders' lists are IDENTICAL, entry by
//                 entry (tests compare them), and everything computed from them is bit for bit the same.
//   k_sym_mirror  UPPER: mirror index of every block below the diagonal (binary search in the mirror's row), -1 elsewhere
//
// Rows whose distinct columns overflow the LDS table (kSymTable / 2) raise a flag: the caller falls back to the host builder.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tsgo {

constexpr int kSymWave = 64;            // one wavefront per row, one row per workgroup
constexpr int kSymBits = 11;
constexpr int kSymTable = 1 << kSymBits;         // hash slots per row (keys + values + sort buffers: 24 KB of LDS per row in flight)
constexpr int kSymMaxDistinct = kSymTable / 2;   // distinct columns a row may have (also the sort buffer); the hierarchy's rows have a few dozen to a few hundred

__device__ __forceinline__ uint32_t sym_hash(int c) { return ((uint32_t)c * 2654435761u) >> (32 - kSymBits); }

// inserts c; returns its slot (-1: the table is full — the caller's row is beyond what this builder takes).  *fresh = 1 when this
// call created the entry.
__device__ __forceinline__ int sym_insert(int* keys, int c, int* fresh) {
    uint32_t h = sym_hash(c);
    for (int probe = 0; probe < kSymTable; ++probe) {
        const int old = atomicCAS(&keys[h], -1, c);
        if (old == -1) { *fresh = 1; return (int)h; }
        if (old == c) { *fresh = 0; return (int)h; }
        h = (h + 1) & (kSymTable - 1);
    }
    *fresh = 0;
    return -1;
}
// slot of a column that IS in the table (every caller looks up what it inserted); bounded all the same: a wave must never spin
__device__ __forceinline__ int sym_find(const int* keys, int c) {
    uint32_t h = sym_hash(c);
    for (int probe = 0; probe < kSymTable && keys[h] != c; ++probe) h = (h + 1) & (kSymTable - 1);
    return (int)h;
}
__device__ __forceinline__ int sym_wave_sum(int v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}

// d[i] = distinct columns of row i of X*Y, m[i] = pairs of the row that get listed (UPPER: those with column >= i)
template <int UPPER>
__global__ __launch_bounds__(kSymWave) void k_sym_count(int n, const int* __restrict__ xptr, const int* __restrict__ xcol, const int* __restrict__ yptr,
                                                        const int* __restrict__ ycol, int* __restrict__ d, int* __restrict__ m, int* __restrict__ overflow) {
    __shared__ int keys[kSymTable];
    const int i = blockIdx.x, lane = threadIdx.x;
    if (i >= n) return;
    for (int k = lane; k < kSymTable; k += kSymWave) keys[k] = -1;
    __syncthreads();
    int distinct = 0, pairs = 0;
    for (int a = xptr[i]; a < xptr[i + 1]; ++a) {
        const int k = xcol[a];
        for (int q = yptr[k] + lane; q < yptr[k + 1]; q += kSymWave) {
            const int c = ycol[q];
            int fresh;
            if (sym_insert(keys, c, &fresh) < 0) *overflow = 1;
            distinct += fresh;
            pairs += (!UPPER || c >= i) ? 1 : 0;
        }
        // a table more than half full stops being a table: give up on this row (the caller falls back to the host builder)
        if (sym_wave_sum(distinct) > kSymMaxDistinct) { if (lane == 0) *overflow = 1; return; }
    }
    distinct = sym_wave_sum(distinct); pairs = sym_wave_sum(pairs);
    if (lane == 0) { d[i] = distinct; m[i] = pairs; }
}

// exclusive scan of n ints by ONE workgroup (n is at most a few million: a row count or a block count); out has n + 1 entries
__global__ __launch_bounds__(1024) void k_sym_scan(int n, const int* __restrict__ in, int* __restrict__ out) {
    __shared__ long long part[1024];
    const int t = threadIdx.x;
    const long long per = ((long long)n + 1023) / 1024;
    const long long b = (long long)t * per, e = b + per < n ? b + per : n;
    long long s = 0;
    for (long long k = b; k < e; ++k) s += in[k];
    part[t] = s;
    __syncthreads();
    if (t == 0) { long long run = 0; for (int k = 0; k < 1024; ++k) { const long long v = part[k]; part[k] = run; run += v; } }
    __syncthreads();
    long long run = part[t];
    for (long long k = b; k < e; ++k) { out[k] = (int)run; run += in[k]; }
    if (t == 1023) out[n] = (int)run;       // the last thread's chunk ends at n (possibly empty): run = the total
}

// in-LDS bitonic sort of v[0 .. size) (size a power of two) by one wavefront
__device__ __forceinline__ void sym_sort(int* v, int size, int lane) {
    for (int k = 2; k <= size; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = lane; t < size; t += kSymWave) {
                const int p = t ^ j;
                if (p > t) {
                    const int a = v[t], b = v[p];
                    const bool up = (t & k) == 0;
                    if ((a > b) == up) { v[t] = b; v[p] = a; }
                }
            }
            __syncthreads();
        }
}

// Z.col, the list offsets pl_ptr (per Z block), the lists (px, py); n_upper[i] = blocks of row i on or above the diagonal (UPPER)
template <int UPPER>
__global__ __launch_bounds__(kSymWave) void k_sym_fill(int n, const int* __restrict__ xptr, const int* __restrict__ xcol, const int* __restrict__ x_alias,
                                                       const int* __restrict__ yptr, const int* __restrict__ ycol, const int* __restrict__ zptr,
                                                       const int* __restrict__ poff, int* __restrict__ zcol, int* __restrict__ pl_ptr,
                                                       int* __restrict__ px, int* __restrict__ py, int* __restrict__ n_upper) {
    __shared__ int keys[kSymTable], vals[kSymTable];
    __shared__ int cols[kSymMaxDistinct], cnt[kSymMaxDistinct];
    __shared__ int n_cols;
    const int i = blockIdx.x, lane = threadIdx.x;
    if (i >= n) return;
    for (int k = lane; k < kSymTable; k += kSymWave) keys[k] = -1;
    if (lane == 0) n_cols = 0;
    __syncthreads();
    // distinct columns -> cols[]
    for (int a = xptr[i]; a < xptr[i + 1]; ++a) {
        const int k = xcol[a];
        for (int q = yptr[k] + lane; q < yptr[k + 1]; q += kSymWave) {
            const int c = ycol[q];
            int fresh;
            (void)sym_insert(keys, c, &fresh);
            if (fresh) cols[atomicAdd(&n_cols, 1)] = c;
        }
    }
    __syncthreads();
    const int dd = n_cols;
    int size = 1; while (size < dd) size <<= 1;
    for (int t = dd + lane; t < size; t += kSymWave) cols[t] = 0x7fffffff;
    __syncthreads();
    sym_sort(cols, size, lane);
    // rank of every column, its pair count
    const int z0 = zptr[i];
    for (int r = lane; r < dd; r += kSymWave) { vals[sym_find(keys, cols[r])] = r; zcol[z0 + r] = cols[r]; cnt[r] = 0; }
    __syncthreads();
    for (int a = xptr[i]; a < xptr[i + 1]; ++a) {
        const int k = xcol[a];
        for (int q = yptr[k] + lane; q < yptr[k + 1]; q += kSymWave) {
            const int c = ycol[q];
            if (!UPPER || c >= i) atomicAdd(&cnt[vals[sym_find(keys, c)]], 1);
        }
    }
    __syncthreads();
    // exclusive scan of cnt over the row's blocks (serial per wave: dd is a few dozen) -> cnt becomes the running fill position
    if (lane == 0) {
        int run = poff[i], first_upper = dd;
        for (int r = 0; r < dd; ++r) {
            const int v = cnt[r];
            pl_ptr[z0 + r] = run; cnt[r] = run; run += v;
            if (UPPER && first_upper == dd && cols[r] >= i) first_upper = r;
        }
        if (UPPER) n_upper[i] = dd - first_upper;
    }
    __syncthreads();
    // the pairs, X blocks in order: within one X block every column occurs once, so no two lanes share an output block
    for (int a = xptr[i]; a < xptr[i + 1]; ++a) {
        const int k = xcol[a];
        const int xa = x_alias ? x_alias[a] : a;
        for (int q = yptr[k] + lane; q < yptr[k + 1]; q += kSymWave) {
            const int c = ycol[q];
            if (UPPER && c < i) continue;
            const int r = vals[sym_find(keys, c)];
            const int dst = cnt[r];
            cnt[r] = dst + 1;
            px[dst] = xa; py[dst] = q;
        }
        __syncthreads();
    }
}

// blocks on or above the diagonal per row of a pattern with sorted rows (level 0 of the hierarchy: k_schur_blocks sums those only)
__global__ __launch_bounds__(256) void k_count_upper(int n, const int* __restrict__ zptr, const int* __restrict__ zcol, int* __restrict__ nup) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int u = 0;
    for (int z = zptr[i]; z < zptr[i + 1]; ++z) u += zcol[z] >= i;
    nup[i] = u;
}

// lower_of[u] = the block below the diagonal whose mirror u is (-1: a diagonal block), from mirror[] of k_sym_mirror
__global__ __launch_bounds__(256) void k_invert_mirror(int nnz, const int* __restrict__ mirror, int* __restrict__ lower_of) {
    const int z = blockIdx.x * 256 + threadIdx.x;
    if (z >= nnz) return;
    const int m = mirror[z];
    if (m >= 0) lower_of[m] = z;
}

// UPPER products: mirror[z] = the block (c, i) for a block z = (i, c) below the diagonal, -1 on or above it; upper[u] = the
// blocks on or above the diagonal in block order (uoff = exclusive scan of n_upper)
__global__ __launch_bounds__(256) void k_sym_mirror(int n, const int* __restrict__ zptr, const int* __restrict__ zcol, const int* __restrict__ uoff,
                                                    int* __restrict__ mirror, int* __restrict__ upper, int* __restrict__ bad) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int u = uoff[i];
    for (int z = zptr[i]; z < zptr[i + 1]; ++z) {
        const int c = zcol[z];
        if (c >= i) { mirror[z] = -1; upper[u++] = z; continue; }
        int lo = zptr[c], hi = zptr[c + 1];
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (zcol[mid] < i) lo = mid + 1; else hi = mid; }
        if (lo < zptr[c + 1] && zcol[lo] == i) mirror[z] = lo; else { mirror[z] = -1; *bad = 1; }      // not structurally symmetric
    }
}

}  // namespace tsgo

// ---- level 0: pattern of the explicit Schur complement S and its contribution lists (host/amg.cpp: "S pattern + lists") ------------
// Row i = pose i.  A block (i, k), k != i, exists when the poses share a landmark (seen from at most `max_deg` poses: hub landmarks
// stay out of the preconditioner's explicit matrix) or an odometry edge; it is summed from pairs (slot of i's edge, slot of k's edge)
// per shared landmark, and from the odometry slots of row i that lead to k.  The diagonal block always exists, with empty lists.
// Inputs are three CSR lists the host lays out in parallel: a pose's LM edges (landmark, by_pose slot; slots ascending), a
// landmark's observers (pose, by_pose slot of that edge), a pose's odometry slots (other pose, slot; slots ascending).
// Lists come out sorted as the host sorts them: pairs by (slot of i, slot of k), odometry slots ascending.
namespace tsgo {

__global__ __launch_bounds__(kSymWave) void k_s0_count(int P, const int* __restrict__ pp_ptr, const int* __restrict__ pp_lm, const int* __restrict__ obs_ptr,
                                                       const int* __restrict__ obs_pose, const int* __restrict__ od_ptr, const int* __restrict__ od_col,
                                                       int max_deg, int* __restrict__ d, int* __restrict__ m, int* __restrict__ mo, int* __restrict__ overflow) {
    __shared__ int keys[kSymTable];
    const int i = blockIdx.x, lane = threadIdx.x;
    if (i >= P) return;
    for (int k = lane; k < kSymTable; k += kSymWave) keys[k] = -1;
    __syncthreads();
    int distinct = 0, pairs = 0, odn = 0, fresh;
    if (lane == 0) { (void)sym_insert(keys, i, &fresh); distinct += fresh; }
    for (int a = pp_ptr[i]; a < pp_ptr[i + 1]; ++a) {
        const int l = pp_lm[a];
        const int q0 = obs_ptr[l], q1 = obs_ptr[l + 1];
        if (q1 - q0 > max_deg) continue;
        for (int q = q0 + lane; q < q1; q += kSymWave) {
            const int c = obs_pose[q];
            if (c == i) continue;
            if (sym_insert(keys, c, &fresh) < 0) *overflow = 1;
            distinct += fresh; ++pairs;
        }
        if (sym_wave_sum(distinct) > kSymMaxDistinct) { if (lane == 0) *overflow = 1; return; }
    }
    for (int e = od_ptr[i] + lane; e < od_ptr[i + 1]; e += kSymWave) {
        if (sym_insert(keys, od_col[e], &fresh) < 0) *overflow = 1;
        distinct += fresh; ++odn;
    }
    distinct = sym_wave_sum(distinct); pairs = sym_wave_sum(pairs); odn = sym_wave_sum(odn);
    // the odometry neighbours count too: k_s0_fill's cols / cnt / cnt_od hold kSymMaxDistinct entries (a pose with ~1024 co-observers
    // plus odometry or loop-closure neighbours that are not among them): the host builder takes such a graph
    if (distinct > kSymMaxDistinct) { if (lane == 0) *overflow = 1; return; }
    if (lane == 0) { d[i] = distinct; m[i] = pairs; mo[i] = odn; }
}

__global__ __launch_bounds__(kSymWave) void k_s0_fill(int P, const int* __restrict__ pp_ptr, const int* __restrict__ pp_lm, const uint32_t* __restrict__ pp_slot,
                                                      const int* __restrict__ obs_ptr, const int* __restrict__ obs_pose, const uint32_t* __restrict__ obs_slot,
                                                      const int* __restrict__ od_ptr, const int* __restrict__ od_col, const uint32_t* __restrict__ od_slot, int max_deg,
                                                      const int* __restrict__ zptr, const int* __restrict__ poff, const int* __restrict__ ooff, int* __restrict__ zcol,
                                                      int* __restrict__ sc_ptr, int* __restrict__ sc_optr, uint32_t* __restrict__ slot_i, uint32_t* __restrict__ slot_k,
                                                      uint32_t* __restrict__ os) {
    __shared__ int keys[kSymTable], vals[kSymTable];
    __shared__ int cols[kSymMaxDistinct], cnt[kSymMaxDistinct], cnt_od[kSymMaxDistinct];
    __shared__ int n_cols;
    const int i = blockIdx.x, lane = threadIdx.x;
    if (i >= P) return;
    for (int k = lane; k < kSymTable; k += kSymWave) keys[k] = -1;
    if (lane == 0) n_cols = 0;
    __syncthreads();
    int fresh;
    if (lane == 0) { (void)sym_insert(keys, i, &fresh); if (fresh) cols[atomicAdd(&n_cols, 1)] = i; }
    for (int a = pp_ptr[i]; a < pp_ptr[i + 1]; ++a) {
        const int l = pp_lm[a];
        const int q0 = obs_ptr[l], q1 = obs_ptr[l + 1];
        if (q1 - q0 > max_deg) continue;
        for (int q = q0 + lane; q < q1; q += kSymWave) {
            const int c = obs_pose[q];
            if (c == i) continue;
            (void)sym_insert(keys, c, &fresh);
            if (fresh) cols[atomicAdd(&n_cols, 1)] = c;
        }
    }
    for (int e = od_ptr[i] + lane; e < od_ptr[i + 1]; e += kSymWave) {
        (void)sym_insert(keys, od_col[e], &fresh);
        if (fresh) cols[atomicAdd(&n_cols, 1)] = od_col[e];
    }
    __syncthreads();
    const int dd = n_cols;
    int size = 1; while (size < dd) size <<= 1;
    for (int t = dd + lane; t < size; t += kSymWave) cols[t] = 0x7fffffff;
    __syncthreads();
    sym_sort(cols, size, lane);
    const int z0 = zptr[i];
    for (int r = lane; r < dd; r += kSymWave) { vals[sym_find(keys, cols[r])] = r; zcol[z0 + r] = cols[r]; cnt[r] = 0; cnt_od[r] = 0; }
    __syncthreads();
    for (int a = pp_ptr[i]; a < pp_ptr[i + 1]; ++a) {
        const int l = pp_lm[a];
        const int q0 = obs_ptr[l], q1 = obs_ptr[l + 1];
        if (q1 - q0 > max_deg) continue;
        for (int q = q0 + lane; q < q1; q += kSymWave) {
            const int c = obs_pose[q];
            if (c != i) atomicAdd(&cnt[vals[sym_find(keys, c)]], 1);
        }
    }
    for (int e = od_ptr[i] + lane; e < od_ptr[i + 1]; e += kSymWave) atomicAdd(&cnt_od[vals[sym_find(keys, od_col[e])]], 1);
    __syncthreads();
    if (lane == 0) {
        int run = poff[i], run_od = ooff[i];
        for (int r = 0; r < dd; ++r) {
            const int v = cnt[r], w = cnt_od[r];
            sc_ptr[z0 + r] = run; cnt[r] = run; run += v;
            sc_optr[z0 + r] = run_od; cnt_od[r] = run_od; run_od += w;
        }
    }
    __syncthreads();
    // pairs: one LM edge of pose i after the other (slots ascending).  Within one edge the observers that are the same pose (an edge
    // given twice) must keep their order, so a lane that meets a column another lane of this trip also has waits for its turn:
    // the trip is serialised over the lanes only in that (rare) case.
    for (int a = pp_ptr[i]; a < pp_ptr[i + 1]; ++a) {
        const int l = pp_lm[a];
        const int q0 = obs_ptr[l], q1 = obs_ptr[l + 1];
        if (q1 - q0 > max_deg) continue;
        const uint32_t sa = pp_slot[a];
        for (int base = q0; base < q1; base += kSymWave) {
            const int q = base + lane;
            const int c = q < q1 ? obs_pose[q] : i;
            const int r = c != i ? vals[sym_find(keys, c)] : -1;
            // rank of this lane among the earlier lanes of the trip with the same output block (0 unless the edge list repeats an edge)
            int before = 0;
            for (int o = 0; o < kSymWave; ++o) { const int ro = __shfl(r, o); if (o < lane && ro == r) ++before; }
            if (r >= 0) { const int dst = cnt[r] + before; slot_i[dst] = sa; slot_k[dst] = obs_slot[q]; }
            __syncthreads();
            if (r >= 0) atomicAdd(&cnt[r], 1);
            __syncthreads();
        }
    }
    if (lane == 0)
        for (int e = od_ptr[i]; e < od_ptr[i + 1]; ++e) { const int r = vals[sym_find(keys, od_col[e])]; os[cnt_od[r]++] = od_slot[e]; }
}

}  // namespace tsgo
