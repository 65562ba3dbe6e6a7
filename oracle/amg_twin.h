// amg_twin.h — TEST INFRASTRUCTURE: scalar CPU twin of the device-side numeric multigrid
// (toyslam_amd/csrc/tsgo_amg_kernels.h), operating on the product's symbolic hierarchy (host/amg.h).
#pragma once
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "host/amg.h"

namespace amgtwin {

using tsgo::AmgLevel; using tsgo::AmgSym;

inline void mat3_mul(const double* a, const double* b, double* c) {       // c += a b
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { double s = 0; for (int k = 0; k < 3; ++k) s += a[3 * i + k] * b[3 * k + j]; c[3 * i + j] += s; }
}
inline void mat3_tmul(const double* a, const double* b, double* c) {      // c += a^T b
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { double s = 0; for (int k = 0; k < 3; ++k) s += a[3 * k + i] * b[3 * k + j]; c[3 * i + j] += s; }
}
inline bool mat3_inv(const double* m, double* o) {
    const double c00 = m[4] * m[8] - m[5] * m[7], c01 = m[5] * m[6] - m[3] * m[8], c02 = m[3] * m[7] - m[4] * m[6];
    const double det = m[0] * c00 + m[1] * c01 + m[2] * c02;
    if (!(std::fabs(det) > 0)) { std::memset(o, 0, 72); return false; }
    const double r = 1.0 / det;
    o[0] = c00 * r; o[1] = (m[2] * m[7] - m[1] * m[8]) * r; o[2] = (m[1] * m[5] - m[2] * m[4]) * r;
    o[3] = c01 * r; o[4] = (m[0] * m[8] - m[2] * m[6]) * r; o[5] = (m[2] * m[3] - m[0] * m[5]) * r;
    o[6] = c02 * r; o[7] = (m[1] * m[6] - m[0] * m[7]) * r; o[8] = (m[0] * m[4] - m[1] * m[3]) * r;
    return true;
}

struct Hierarchy {
    const AmgSym* sym = nullptr;
    std::vector<std::vector<double>> A, Dinv, P, T;     // per level: 9 doubles per block / row
    std::vector<double> A_last, inv_last;               // coarsest: block values, dense inverse (n x n)
    int n_last = 0;
    std::vector<std::vector<double>> r, z, res, z2;     // per level work vectors (3 per node), level 0 unused for r
    std::vector<double> omega;                          // smoother damping per level (power-iteration estimate)

    void alloc(const AmgSym& s) {
        sym = &s;
        const size_t nl = s.levels.size();
        A.resize(nl); Dinv.resize(nl); P.resize(nl); T.resize(nl); r.resize(nl + 1); z.resize(nl + 1); res.resize(nl + 1); z2.resize(nl + 1);
        for (size_t l = 0; l < nl; ++l) {
            const AmgLevel& L = s.levels[l];
            A[l].assign((size_t)L.A.nnz() * 9, 0); Dinv[l].assign((size_t)L.n * 9, 0);
            P[l].assign((size_t)L.P.nnz() * 9, 0); T[l].assign((size_t)L.T.nnz() * 9, 0);
            r[l].assign((size_t)L.n * 3, 0); z[l].assign((size_t)L.n * 3, 0); res[l].assign((size_t)L.n * 3, 0); z2[l].assign((size_t)L.n * 3, 0);
        }
        n_last = s.A_last.n_rows * 3;
        A_last.assign((size_t)s.A_last.nnz() * 9, 0); inv_last.assign((size_t)n_last * n_last, 0);
        r[nl].assign(n_last, 0); z[nl].assign(n_last, 0);
    }

    // everything above level 0's A: P, T, Galerkin products, block inverses, coarsest inverse
    void setup_from_level0() {
        const size_t nl = sym->levels.size();
        for (size_t l = 0; l < nl; ++l) {
            const AmgLevel& L = sym->levels[l];
            std::vector<double>& Al = A[l];
            for (int i = 0; i < L.n; ++i) mat3_inv(&Al[(size_t)L.diag[i] * 9], &Dinv[l][(size_t)i * 9]);
            // P = Z - w Dinv (A Z)
            for (int i = 0; i < L.n; ++i)
                for (int pb = L.P.ptr[i]; pb < L.P.ptr[i + 1]; ++pb) {
                    double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
                    for (int q = L.p_src.ptr[pb]; q < L.p_src.ptr[pb + 1]; ++q) {
                        const int k = L.p_src.y[q];
                        const double zk[9] = {1, 0, -L.rel[2 * (size_t)k + 1], 0, 1, L.rel[2 * (size_t)k], 0, 0, 1};
                        mat3_mul(&Al[(size_t)L.p_src.x[q] * 9], zk, acc);
                    }
                    double da[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
                    mat3_mul(&Dinv[l][(size_t)i * 9], acc, da);
                    double* o = &P[l][(size_t)pb * 9];
                    static const double omega = getenv("TSGO_TWIN_OMEGA") ? atof(getenv("TSGO_TWIN_OMEGA")) : tsgo::kProlongOmega;
                    for (int m = 0; m < 9; ++m) o[m] = -omega * da[m];
                    const double* dd = &Dinv[l][(size_t)i * 9];
                    const bool dead = dd[0] == 0 && dd[4] == 0 && dd[8] == 0;       // vertex without edges: keep it out of the coarse space
                    if (L.p_self[pb] && !dead) { o[0] += 1; o[4] += 1; o[8] += 1; o[2] += -L.rel[2 * (size_t)i + 1]; o[5] += L.rel[2 * (size_t)i]; }
                }
            // T = A P
            for (int tb = 0; tb < L.T.nnz(); ++tb) {
                double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
                for (int q = L.t_src.ptr[tb]; q < L.t_src.ptr[tb + 1]; ++q) mat3_mul(&Al[(size_t)L.t_src.x[q] * 9], &P[l][(size_t)L.t_src.y[q] * 9], acc);
                std::memcpy(&T[l][(size_t)tb * 9], acc, 72);
            }
            // A_next = P^T T
            std::vector<double>& An = (l + 1 < nl) ? A[l + 1] : A_last;
            const int nb = (int)L.a_src.ptr.size() - 1;
            for (int ab = 0; ab < nb; ++ab) {
                double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
                for (int q = L.a_src.ptr[ab]; q < L.a_src.ptr[ab + 1]; ++q) mat3_tmul(&P[l][(size_t)L.a_src.x[q] * 9], &T[l][(size_t)L.a_src.y[q] * 9], acc);
                std::memcpy(&An[(size_t)ab * 9], acc, 72);
            }
            // blocks below the diagonal are the transposes of their mirrors (host/amg.h: a_mirror)
            for (int ab = 0; ab < nb; ++ab) {
                const int mb = L.a_mirror[ab];
                if (mb < 0) continue;
                for (int x = 0; x < 3; ++x) for (int y = 0; y < 3; ++y) An[(size_t)ab * 9 + 3 * x + y] = An[(size_t)mb * 9 + 3 * y + x];
            }
        }
        // smoother damping per level: 16 power steps on D^-1 A from a fixed start vector, exactly as the device
        // does (tsgo_hip.hip, estimate_damping): omega = min(1, 1.6 / (1.05 rho))
        omega.assign(nl, tsgo::kSmootherOmega);
        for (size_t l = 0; l < nl; ++l) {
            const AmgLevel& L = sym->levels[l];
            std::vector<double> v((size_t)L.n * 3), w((size_t)L.n * 3), u((size_t)L.n * 3);
            for (size_t k = 0; k < v.size(); ++k) v[k] = std::sin(0.37 * (double)k) + 0.1;
            double nk = 0, nk1 = 0;
            for (int it = 0; it < 16; ++it) {
                spmv(L.A, A[l], v, w);
                for (int i = 0; i < L.n; ++i) { const double* d = &Dinv[l][(size_t)i * 9]; const double* x = &w[(size_t)i * 3]; for (int c = 0; c < 3; ++c) u[(size_t)i * 3 + c] = d[3 * c] * x[0] + d[3 * c + 1] * x[1] + d[3 * c + 2] * x[2]; }
                nk1 = 0; for (double x : v) nk1 += x * x;
                nk = 0; for (double x : u) nk += x * x;
                v.swap(u);
            }
            if (nk > 0 && nk1 > 0) omega[l] = std::min(1.0, 1.6 / (1.05 * std::sqrt(nk / nk1)));
            if (getenv("TSGO_TWIN_RHO")) std::fprintf(stderr, "[twin] level %zu: n = %d, blocks/row = %.1f, rho ~ %.3f, omega = %.3f\n", l, L.n, (double)L.A.nnz() / L.n, std::sqrt(nk / nk1), omega[l]);
        }
        // coarsest: dense inverse by Gauss-Jordan (SPD, no pivoting needed; partial pivoting kept for safety)
        const int n = n_last;
        std::vector<double> M((size_t)n * n, 0.0);
        const tsgo::BlockCsr& Ac = sym->A_last;
        for (int i = 0; i < Ac.n_rows; ++i)
            for (int a = Ac.ptr[i]; a < Ac.ptr[i + 1]; ++a)
                for (int x = 0; x < 3; ++x) for (int y = 0; y < 3; ++y) M[(size_t)(3 * i + x) * n + 3 * Ac.col[a] + y] = A_last[(size_t)a * 9 + 3 * x + y];
        std::vector<double>& I = inv_last;
        std::fill(I.begin(), I.end(), 0.0);
        for (int i = 0; i < n; ++i) I[(size_t)i * n + i] = 1;
        for (int c = 0; c < n; ++c) {
            int piv = c;
            for (int rr = c + 1; rr < n; ++rr) if (std::fabs(M[(size_t)rr * n + c]) > std::fabs(M[(size_t)piv * n + c])) piv = rr;
            if (piv != c) for (int j = 0; j < n; ++j) { std::swap(M[(size_t)piv * n + j], M[(size_t)c * n + j]); std::swap(I[(size_t)piv * n + j], I[(size_t)c * n + j]); }
            const double d = M[(size_t)c * n + c];
            if (!(std::fabs(d) > 0)) continue;
            for (int j = 0; j < n; ++j) { M[(size_t)c * n + j] /= d; I[(size_t)c * n + j] /= d; }
            for (int rr = 0; rr < n; ++rr) if (rr != c) {
                const double f = M[(size_t)rr * n + c];
                if (f != 0) for (int j = 0; j < n; ++j) { M[(size_t)rr * n + j] -= f * M[(size_t)c * n + j]; I[(size_t)rr * n + j] -= f * I[(size_t)c * n + j]; }
            }
        }
    }

    static void spmv(const tsgo::BlockCsr& Ap, const std::vector<double>& Av, const std::vector<double>& x, std::vector<double>& y) {
        #pragma omp parallel for schedule(static)
        for (int i = 0; i < Ap.n_rows; ++i) {
            double s0 = 0, s1 = 0, s2 = 0;
            for (int a = Ap.ptr[i]; a < Ap.ptr[i + 1]; ++a) {
                const double* b = &Av[(size_t)a * 9]; const double* v = &x[(size_t)Ap.col[a] * 3];
                s0 += b[0] * v[0] + b[1] * v[1] + b[2] * v[2]; s1 += b[3] * v[0] + b[4] * v[1] + b[5] * v[2]; s2 += b[6] * v[0] + b[7] * v[1] + b[8] * v[2];
            }
            y[(size_t)i * 3] = s0; y[(size_t)i * 3 + 1] = s1; y[(size_t)i * 3 + 2] = s2;
        }
    }
    static void dinv_apply(const std::vector<double>& D, const std::vector<double>& r_, std::vector<double>& z_, int n, bool add, double ws) {
        #pragma omp parallel for schedule(static)
        for (int i = 0; i < n; ++i) {
            const double* d = &D[(size_t)i * 9]; const double* v = &r_[(size_t)i * 3];
            for (int x = 0; x < 3; ++x) { const double s = ws * (d[3 * x] * v[0] + d[3 * x + 1] * v[1] + d[3 * x + 2] * v[2]); if (add) z_[(size_t)i * 3 + x] += s; else z_[(size_t)i * 3 + x] = s; }
        }
    }
    // rc = P^T v   (over R rows)
    void restrict_to(size_t l, const std::vector<double>& v, std::vector<double>& rc) const {
        const AmgLevel& L = sym->levels[l];
        #pragma omp parallel for schedule(static)
        for (int a = 0; a < L.R.n_rows; ++a) {
            double s[3] = {0, 0, 0};
            for (int rb = L.R.ptr[a]; rb < L.R.ptr[a + 1]; ++rb) {
                const double* b = &P[l][(size_t)L.r_to_p[rb] * 9]; const double* x = &v[(size_t)L.R.col[rb] * 3];
                for (int c = 0; c < 3; ++c) s[c] += b[c] * x[0] + b[3 + c] * x[1] + b[6 + c] * x[2];
            }
            rc[(size_t)a * 3] = s[0]; rc[(size_t)a * 3 + 1] = s[1]; rc[(size_t)a * 3 + 2] = s[2];
        }
    }
    // z += P e
    void prolong_add(size_t l, const std::vector<double>& e, std::vector<double>& z_) const {
        const AmgLevel& L = sym->levels[l];
        #pragma omp parallel for schedule(static)
        for (int i = 0; i < L.n; ++i)
            for (int pb = L.P.ptr[i]; pb < L.P.ptr[i + 1]; ++pb) {
                const double* b = &P[l][(size_t)pb * 9]; const double* x = &e[(size_t)L.P.col[pb] * 3];
                for (int c = 0; c < 3; ++c) z_[(size_t)i * 3 + c] += b[3 * c] * x[0] + b[3 * c + 1] * x[1] + b[3 * c + 2] * x[2];
            }
    }
    // levels >= 1 of the V(1,1) cycle: z[l] = cycle(r[l])
    void cycle(size_t l) {
        const size_t nl = sym->levels.size();
        if (l == nl) {
            const int n = n_last;
            for (int i = 0; i < n; ++i) { double s = 0; for (int j = 0; j < n; ++j) s += inv_last[(size_t)i * n + j] * r[l][j]; z[l][i] = s; }
            return;
        }
        const AmgLevel& L = sym->levels[l];
        static const int nu_env = getenv("TSGO_TWIN_NU") ? atoi(getenv("TSGO_TWIN_NU")) : 0;
        const int nu = nu_env ? nu_env : tsgo::sweeps_per_side((size_t)l, L.n);
        static const int gam = getenv("TSGO_TWIN_GAMMA") ? atoi(getenv("TSGO_TWIN_GAMMA")) : 1;
        dinv_apply(Dinv[l], r[l], z[l], L.n, false, omega[l]);
        for (int s = 1; s < nu; ++s) {
            spmv(L.A, A[l], z[l], res[l]);
            for (size_t k = 0; k < res[l].size(); ++k) res[l][k] = r[l][k] - res[l][k];
            dinv_apply(Dinv[l], res[l], z[l], L.n, true, omega[l]);
        }
        for (int g = 0; g < (l >= 2 ? gam : 1); ++g) {
            spmv(L.A, A[l], z[l], res[l]);
            for (size_t k = 0; k < res[l].size(); ++k) res[l][k] = r[l][k] - res[l][k];
            restrict_to(l, res[l], r[l + 1]);
            cycle(l + 1);
            prolong_add(l, z[l + 1], z[l]);
        }
        for (int s = 0; s < nu; ++s) {
            spmv(L.A, A[l], z[l], res[l]);
            for (size_t k = 0; k < res[l].size(); ++k) res[l][k] = r[l][k] - res[l][k];
            dinv_apply(Dinv[l], res[l], z[l], L.n, true, omega[l]);
        }
    }
};

}  // namespace amgtwin
