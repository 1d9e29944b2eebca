// kkt.hpp -- per-candidate KKT solves on LDS-resident blocks, one wavefront per active set (gfx950).
//
// Reference: MPQP_Program.optimal_control_law (mpqp_program.py:146-198) solves
//     [A_as  0    ] [x]   [b_as + F_as theta]
//     [Q     A_as'] [l] = [-c   - H theta   ]
// with numpy.linalg.solve.  Two device paths:
//   mode 0 (Q > 0)  Schur complement on blocks precomputed once per program:
//          W = A Q^-1 A', UV = [A Q^-1 c + b | A Q^-1 H + F]:   S = W[as,as] (gathered),  S L = -UV[as]
//          by Cholesky; a non-positive pivot <=> A_as rank deficient (is_full_rank, constraint_utilities.py:222).
//   mode 1 (mpLP / PSD Q)  the dense (n_x+k) KKT matrix itself, LU with partial pivoting (== LAPACK gesv).
#pragma once
#include "lp_engine.hpp"

namespace mpc {

constexpr double RANK_TOL_CHOL = 1e-13;  // pivot / original diagonal (= sin^2 of the angle to the span)
constexpr double KKT_SING_TOL = 1e-12;   // LU pivot relative to the largest entry of the KKT matrix

// In-place lower Cholesky of the k x k LDS matrix S (row stride k) and solution of S X = R for the k x nr LDS
// block R.  diag0[i] holds the original diagonal.  Returns false when a pivot falls below RANK_TOL_CHOL.
// ill (optional): set when a pivot is below CHOL_ILL_TOL of its diagonal -- cond(S) >~ 1e9, the multipliers carry a relative
// error >~ 1e-7 (S = A_as Q^-1 A_as' squares the conditioning of A_as) and decisions at the 1e-7 tolerance are not reliable.
constexpr double CHOL_ILL_TOL = 1e-9;
__device__ inline bool chol_solve(double *S, int k, double *R, int nr, const double *diag0, bool *ill = nullptr) {
    const int lane = lane_id();
    for (int j = 0; j < k; ++j) {
        const double d = S[j * k + j];
        if (!(d > RANK_TOL_CHOL * diag0[j])) return false;
        if (ill && !(d > CHOL_ILL_TOL * diag0[j])) *ill = true;
        const double l = sqrt(d), inv = 1.0 / l;
        wave_sync();
        for (int i = j + 1 + lane; i < k; i += 64) S[i * k + j] = S[i * k + j] * inv;
        if (lane == 0) S[j * k + j] = l;
        wave_sync();
        const int cnt = k - j - 1;
        for (int idx = lane; idx < cnt * cnt; idx += 64) {
            const int i = j + 1 + idx / cnt, c = j + 1 + idx % cnt;
            if (c <= i) S[i * k + c] = fma(-S[i * k + j], S[c * k + j], S[i * k + c]);
        }
        wave_sync();
    }
    // forward substitution L Y = R (column oriented)
    for (int j = 0; j < k; ++j) {
        const double inv = 1.0 / S[j * k + j];
        wave_sync();
        for (int t = lane; t < nr; t += 64) R[j * nr + t] = R[j * nr + t] * inv;
        wave_sync();
        const int cnt = k - j - 1;
        for (int idx = lane; idx < cnt * nr; idx += 64) {
            const int i = j + 1 + idx / nr, t = idx % nr;
            R[i * nr + t] = fma(-S[i * k + j], R[j * nr + t], R[i * nr + t]);
        }
        wave_sync();
    }
    // back substitution L' X = Y
    for (int j = k - 1; j >= 0; --j) {
        const double inv = 1.0 / S[j * k + j];
        wave_sync();
        for (int t = lane; t < nr; t += 64) R[j * nr + t] = R[j * nr + t] * inv;
        wave_sync();
        for (int idx = lane; idx < j * nr; idx += 64) {
            const int i = idx / nr, t = idx % nr;
            R[i * nr + t] = fma(-S[j * k + i], R[j * nr + t], R[i * nr + t]);
        }
        wave_sync();
    }
    return true;
}

// LU with partial pivoting of the n x n LDS matrix M (row stride n), applied to the n x nr block B.
// Returns false if a pivot is <= sing_tol * scale (scale = largest |entry| of M).
__device__ inline bool lu_solve(double *M, int n, double *B, int nr, double sing_tol) {
    const int lane = lane_id();
    double scale = 0.0;
    for (int idx = lane; idx < n * n; idx += 64) scale = fmax(scale, fabs(M[idx]));
    for (int off = 32; off > 0; off >>= 1) scale = fmax(scale, __shfl_xor(scale, off));
    for (int col = 0; col < n; ++col) {
        double best = -1.0;
        int piv = -1;
        for (int i = col + lane; i < n; i += 64) {
            const double a = fabs(M[i * n + col]);
            if (a > best) { best = a; piv = i; }
        }
        reduce_max_first(best, piv);
        if (!(best > sing_tol * scale)) return false;
        wave_sync();
        if (piv != col) {
            for (int j = lane; j < n; j += 64) { const double t = M[col * n + j]; M[col * n + j] = M[piv * n + j]; M[piv * n + j] = t; }
            for (int j = lane; j < nr; j += 64) { const double t = B[col * nr + j]; B[col * nr + j] = B[piv * nr + j]; B[piv * nr + j] = t; }
            wave_sync();
        }
        const double inv = 1.0 / M[col * n + col];
        wave_sync();
        for (int i = col + 1 + lane; i < n; i += 64) M[i * n + col] = M[i * n + col] * inv;
        wave_sync();
        const int rows = n - col - 1, cols = n - col - 1 + nr;
        for (int idx = lane; idx < rows * cols; idx += 64) {
            const int i = col + 1 + idx / cols, c = idx % cols;
            const double f = M[i * n + col];
            if (c < n - col - 1) {
                const int j = col + 1 + c;
                M[i * n + j] = fma(-f, M[col * n + j], M[i * n + j]);
            } else {
                const int t = c - (n - col - 1);
                B[i * nr + t] = fma(-f, B[col * nr + t], B[i * nr + t]);
            }
        }
        wave_sync();
    }
    for (int j = n - 1; j >= 0; --j) {
        const double inv = 1.0 / M[j * n + j];
        wave_sync();
        for (int t = lane; t < nr; t += 64) B[j * nr + t] = B[j * nr + t] * inv;
        wave_sync();
        for (int idx = lane; idx < j * nr; idx += 64) {
            const int i = idx / nr, t = idx % nr;
            B[i * nr + t] = fma(-M[i * n + j], B[j * nr + t], B[i * nr + t]);
        }
        wave_sync();
    }
    return true;
}

// ---- is_full_rank(A, as)  (utils/constraint_utilities.py:222-236: numpy.linalg.matrix_rank(A[as]) == len(as)) ----------------------
// numpy's rule: singular values by SVD, rank = #{sigma > sigma_max * max(k, n) * eps}.  Two steps here:
//   screen   Gaussian elimination with complete pivoting on the k x n LDS matrix M (row stride n, destroyed).  Every pivot above
//            RANK_SCREEN_FULL of the largest entry: the rows are independent by a margin of seven decades over numpy's threshold
//            (complete pivoting keeps the growth small) -> full rank, no SVD.  Anything else is AMBIGUOUS;
//   exact    the singular values themselves by one-sided Jacobi (Hestenes) on the (re-loaded) rows, high relative accuracy also for
//            the small ones, and numpy's rule word for word.  (Round 3 stopped at the elimination with
//            a relative threshold of 1e-11: sets whose smallest singular value lies between 4e-15 and 1e-11 of the largest were
//            called rank deficient where the reference goes on.)
constexpr double RANK_SCREEN_FULL = 1e-8;
constexpr double RANK_SCREEN_ZERO = 2.220446049250313e-16;
// 1 full rank, 0 certainly deficient (k > n, a zero matrix, an exact dependency), 2 ambiguous (M destroyed: reload it and ask svd_full_row_rank)
__device__ inline int rank_screen(double *M, int k, int n) {
    const int lane = lane_id();
    if (k > n) return 0;
    double scale = 0.0;
    for (int idx = lane; idx < k * n; idx += 64) scale = fmax(scale, fabs(M[idx]));
    for (int off = 32; off > 0; off >>= 1) scale = fmax(scale, __shfl_xor(scale, off));
    if (!(scale > 0.0)) return k == 0 ? 1 : 0;
    for (int s = 0; s < k; ++s) {
        double best = -1.0;
        int pos = -1;
        const int rows = k - s, cols = n - s;
        for (int idx = lane; idx < rows * cols; idx += 64) {
            const int i = s + idx / cols, j = s + idx % cols;
            const double a = fabs(M[i * n + j]);
            if (a > best) { best = a; pos = i * n + j; }
        }
        reduce_max_first(best, pos);
        // what is left is below one unit in the last place of the largest entry: an exact dependency (both bounds of a variable active, a
        // repeated row: half of the candidates of a deep level) -- rank deficient by numpy's rule too (sigma_min <~ k * pivot, threshold
        // sigma_max * max(k, n) * eps), without paying for the singular values
        if (!(best > RANK_SCREEN_ZERO * scale)) return 0;
        if (!(best > RANK_SCREEN_FULL * scale)) return 2;
        const int pi = pos / n, pj = pos % n;
        wave_sync();
        if (pi != s) {
            for (int j = lane; j < n; j += 64) { const double t = M[s * n + j]; M[s * n + j] = M[pi * n + j]; M[pi * n + j] = t; }
            wave_sync();
        }
        if (pj != s) {
            for (int i = lane; i < k; i += 64) { const double t = M[i * n + s]; M[i * n + s] = M[i * n + pj]; M[i * n + pj] = t; }
            wave_sync();
        }
        const double inv = 1.0 / M[s * n + s];
        for (int idx = lane; idx < (rows - 1) * (cols - 1); idx += 64) {
            const int i = s + 1 + idx / (cols - 1), j = s + 1 + idx % (cols - 1);
            M[i * n + j] = fma(-(M[i * n + s] * inv), M[s * n + j], M[i * n + j]);
        }
        wave_sync();
    }
    return 1;
}

__device__ __forceinline__ double wave_sum(double v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// numpy.linalg.matrix_rank(M) == k for the k x n LDS matrix M (row stride n, k <= n, destroyed): the rows are orthogonalised by
// plane rotations (Hestenes); their norms are then the singular values.
__device__ inline bool svd_full_row_rank(double *M, int k, int n) {
    const int lane = lane_id();
    if (k > n) return false;
    if (k == 0) return true;
    for (int sweep = 0; sweep < 60; ++sweep) {
        bool rotated = false;
        for (int p = 0; p < k - 1; ++p)
            for (int q = p + 1; q < k; ++q) {
                double app = 0.0, aqq = 0.0, apq = 0.0;
                for (int j = lane; j < n; j += 64) { const double a = M[p * n + j], b = M[q * n + j]; app = fma(a, a, app); aqq = fma(b, b, aqq); apq = fma(a, b, apq); }
                app = wave_sum(app); aqq = wave_sum(aqq); apq = wave_sum(apq);      // (identical in every lane)
                if (apq == 0.0 || fabs(apq) <= 1e-15 * sqrt(app * aqq)) continue;
                rotated = true;
                const double zeta = (aqq - app) / (2.0 * apq);
                const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double cs = 1.0 / sqrt(1.0 + t * t), sn = cs * t;
                for (int j = lane; j < n; j += 64) {
                    const double a = M[p * n + j], b = M[q * n + j];
                    M[p * n + j] = cs * a - sn * b; M[q * n + j] = sn * a + cs * b;
                }
                wave_sync();
            }
        if (!rotated) break;
    }
    double smax = 0.0;
    for (int i = 0; i < k; ++i) {
        double s2 = 0.0;
        for (int j = lane; j < n; j += 64) s2 = fma(M[i * n + j], M[i * n + j], s2);
        smax = fmax(smax, sqrt(wave_sum(s2)));
    }
    const double tol = smax * (double)(k > n ? k : n) * 2.220446049250313e-16;
    int rank = 0;
    for (int i = 0; i < k; ++i) {
        double s2 = 0.0;
        for (int j = lane; j < n; j += 64) s2 = fma(M[i * n + j], M[i * n + j], s2);
        if (sqrt(wave_sum(s2)) > tol) ++rank;
    }
    return rank == k;
}

// is_full_rank on the k x n block `load` writes into M (called again when the screen was not conclusive)
template <class Load>
__device__ inline bool full_row_rank(double *M, int k, int n, Load load) {
    load();
    wave_sync();
    const int r = rank_screen(M, k, n);
    wave_sync();
    if (r != 2) return r == 1;
    load();
    wave_sync();
    const bool full = svd_full_row_rank(M, k, n);
    wave_sync();
    return full;
}

}  // namespace mpc
