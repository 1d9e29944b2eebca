/*
 * mpcombi_oracle.c -- CPU oracle (TEST INFRASTRUCTURE; see mpcombi_oracle.h).
 *
 * Plain C99 restatement of the reference's combinatorial hot path.  Each function cites the
 * reference lines it follows (paths relative to /root/reference/src/ppopt).  Build:
 *   gcc -O2 -std=c99 -ffp-contract=off -fopenmp -fPIC -shared mpcombi_oracle.c -o libmpcombi_oracle.so -lm
 */
#include "mpcombi_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------------
 * Per-thread scratch for orc_check_level.  Every candidate allocates its LP matrices afresh (the reference builds its
 * numpy blocks per call); those of the optimality LP are 150-500 KB, above glibc's mmap threshold, so with 256 threads
 * every candidate paid mmap / munmap system calls under one process-wide lock -- the OpenMP loop scaled 9.8x on 256 threads
 * (round 3).  Inside the loop malloc / calloc / free below are served from a thread-local block that is rewound per
 * candidate; anywhere else they are the C library's.
 * ---------------------------------------------------------------------------------------------- */
static __thread char *t_arena = NULL;
static __thread size_t t_top = 0, t_cap = 0;
static __thread int t_arena_on = 0;
static void *orc_malloc(size_t n) {
    if (t_arena_on) {
        const size_t need = (n + 63) & ~(size_t)63;
        if (t_top + need <= t_cap) { void *p = t_arena + t_top; t_top += need; return p; }
    }
    return (malloc)(n);
}
static void *orc_calloc(size_t a, size_t b) {
    void *p = orc_malloc(a * b);
    if (p) memset(p, 0, a * b);
    return p;
}
static void orc_free(void *p) {
    if (p && t_arena && (char *)p >= t_arena && (char *)p < t_arena + t_cap) return;   /* rewound with the candidate */
    (free)(p);
}
#define malloc(n) orc_malloc(n)
#define calloc(a, b) orc_calloc(a, b)
#define free(p) orc_free(p)
#define ORC_ARENA_BYTES ((size_t)24 << 20)

/* ------------------------------------------------------------------------------------------------
 * Dense two-phase simplex (stands in for GLPK behind solver_interface/cvxopt_interface.py:153-208).
 *
 * Dictionary form.  Row i:  basic_i = T[i][0] - sum_j T[i][j] * nonbasic_j   (j = 1..na)
 * Cost row m:               -z      = T[m][0] - sum_j (-d_j) ...  stored so that the same
 *                           elimination rule applies: T[m][j] = d_j, T[m][0] = -z.
 * Variables: structural x_j (free), one slack per row (>= 0, or fixed 0 for an equality row) and the
 * phase-1 artificial x0 >= 0.
 *   stage A  every free structural variable is pivoted into the basis (equality rows first, largest
 *            |coefficient|); a fixed slack that becomes nonbasic has its column deleted
 *   stage B  equality rows still basic are pivoted out (or found redundant / inconsistent)
 *   phase 1  x0 method with Dantzig pricing, Bland's rule while stalled at a degenerate vertex
 *   phase 2  primal simplex on the real objective
 * Tolerances mirror the 1e-7 primal feasibility tolerance of GLPK/HiGHS on rows scaled to O(1).
 * ---------------------------------------------------------------------------------------------- */
static double g_tol_feas = 1e-7; /* primal feasibility tolerance; orc_set_feas_tol() changes it for the
                                    * knife-edge classification in the tests (decisions that flip with it) */
#define TOL_FEAS g_tol_feas
#define TOL_PIV 1e-9
#define TOL_COST 1e-9
#define HARRIS_DELTA 1e-9
#define DEG_SWITCH 12
#define GROWTH_SAFE 1e3   /* largest tolerated |column max / pivot| before the basis is refactored */
#define MAX_REFACTOR 2

enum { RK_INEQ = 0, RK_EQ = 1, RK_FREE = 2, RK_DEAD = 3, RK_X0 = 4, RK_PRI = 5 };

typedef struct {
    int m, n, ld, na;
    double *T;
    int *colvar; /* 1-based positions */
    int *rowvar;
    unsigned char *rowkind;
    unsigned char *is_eq;
    int iters, max_iter;
    double growth; /* max over ratio-test pivots of (largest |entry| of the pivot column) / |pivot| */
} lp_t;

#define TT(lp, i, j) ((lp)->T[(size_t)(i) * (lp)->ld + (j)])

static void lp_pivot(lp_t *lp, int r, int q) {
    const int na = lp->na, m = lp->m;
    const double inv = 1.0 / TT(lp, r, q);
    for (int j = 0; j <= na; ++j)
        if (j != q) TT(lp, r, j) = TT(lp, r, j) * inv;
    TT(lp, r, q) = inv;
    for (int i = 0; i <= m; ++i) {
        if (i == r) continue;
        const double f = TT(lp, i, q);
        if (f == 0.0) continue;
        for (int j = 0; j <= na; ++j)
            if (j != q) TT(lp, i, j) = fma(-f, TT(lp, r, j), TT(lp, i, j));
        TT(lp, i, q) = -f * inv;
    }
    const int t = lp->rowvar[r];
    lp->rowvar[r] = lp->colvar[q];
    lp->colvar[q] = t;
    lp->iters++;
}

static void lp_drop_col(lp_t *lp, int q) {
    const int na = lp->na;
    if (q != na) {
        for (int i = 0; i <= lp->m; ++i) TT(lp, i, q) = TT(lp, i, na);
        lp->colvar[q] = lp->colvar[na];
    }
    lp->na = na - 1;
}

/* primal simplex iterations on cost row `crow` (== m for the real objective, or the x0 row in phase 1,
 * where the "reduced cost" of column j is -T[crow][j]).  Returns 0 optimal, 2 unbounded, 3 limit,
 * 4 (phase 1 only) x0 left the basis.
 * Ratio test: Harris two-pass (pass 1 bounds the step with beta + HARRIS_DELTA, pass 2 takes the largest pivot
 * among the rows whose own ratio does not exceed that bound); textbook rule with Bland's tie-break while stalled. */
static int lp_primal(lp_t *lp, int phase1_row) {
    const int m = lp->m;
    int deg = 0;
    for (;;) {
        if (lp->iters > lp->max_iter) return 3;
        const int bland = deg > DEG_SWITCH;
        const int crow = phase1_row >= 0 ? phase1_row : m;
        const double sgn = phase1_row >= 0 ? -1.0 : 1.0;
        if (phase1_row >= 0 && TT(lp, phase1_row, 0) <= TOL_FEAS) return 0;
        /* pricing */
        int q = -1;
        double best = -TOL_COST;
        int best_var = 0;
        for (int j = 1; j <= lp->na; ++j) {
            const double d = sgn * TT(lp, crow, j);
            if (d < -TOL_COST) {
                if (bland) {
                    if (q < 0 || lp->colvar[j] < best_var) { q = j; best_var = lp->colvar[j]; }
                } else if (d < best) { best = d; q = j; }
            }
        }
        if (q < 0) return 0;
        /* ratio test */
        double colmax = 0.0, tmax = 0.0;
        int any = 0;
        for (int i = 0; i < m; ++i) {
            const int kind = lp->rowkind[i];
            if (kind == RK_DEAD) continue;
            const double a = TT(lp, i, q);
            if (fabs(a) > colmax) colmax = fabs(a);
            if ((kind != RK_INEQ && kind != RK_X0) || a <= TOL_PIV) continue;
            double beta = TT(lp, i, 0);
            if (beta < 0.0) beta = 0.0;
            const double t = (beta + HARRIS_DELTA) / a;
            if (!any || t < tmax) { tmax = t; any = 1; }
        }
        if (!any) return 2;
        int r = -1;
        double rmin = 0.0, rpiv = 0.0;
        for (int i = 0; i < m; ++i) {
            const int kind = lp->rowkind[i];
            if (kind != RK_INEQ && kind != RK_X0) continue;
            const double a = TT(lp, i, q);
            if (a <= TOL_PIV) continue;
            double beta = TT(lp, i, 0);
            if (beta < 0.0) beta = 0.0;
            const double ratio = beta / a;
            int take = 0;
            if (bland) {
                if (r < 0 || ratio < rmin) take = 1;
                else if (ratio == rmin) {
                    if (kind == RK_X0) take = 1;
                    else if (lp->rowkind[r] == RK_X0) take = 0;
                    else take = lp->rowvar[i] < lp->rowvar[r];
                }
            } else {
                if (ratio > tmax) continue;
                if (r < 0) take = 1;
                else if (kind == RK_X0) take = 1;
                else if (lp->rowkind[r] == RK_X0) take = 0;
                else take = a > rpiv;
            }
            if (take) { r = i; rmin = ratio; rpiv = a; }
        }
        if (r < 0) return 2;
        if (colmax / rpiv > lp->growth) lp->growth = colmax / rpiv;
        deg = (rmin <= 0.0) ? deg + 1 : 0;
        const int leaving_x0 = lp->rowkind[r] == RK_X0;
        lp_pivot(lp, r, q);
        if (leaving_x0) {
            lp->rowkind[r] = RK_INEQ;
            lp_drop_col(lp, q);
            return 4;
        }
    }
}

static int lp_best_row(const lp_t *lp, int q, int kind) {
    int r = -1; double best = TOL_PIV;
    for (int i = 0; i < lp->m; ++i)
        if (lp->rowkind[i] == kind) { const double a = fabs(TT(lp, i, q)); if (a > best) { best = a; r = i; } }
    return r;
}

/* one pass: stage A, stage B, phase 1, phase 2 on the tableau as loaded */
static int lp_run(lp_t *lp, int has_cost) {
    const int m = lp->m, n = lp->n;
    int unbounded_if_feasible = 0;
    /* stage A: structural (free) variables enter the basis */
    for (int v = 0; v < n; ++v) {
        int q = -1;
        for (int j = 1; j <= lp->na; ++j) if (lp->colvar[j] == v) { q = j; break; }
        int r = lp_best_row(lp, q, RK_EQ);
        const int was_eq = r >= 0;
        if (r < 0) r = lp_best_row(lp, q, RK_PRI);
        if (r < 0) r = lp_best_row(lp, q, RK_INEQ);
        if (r < 0) { /* variable does not appear in any usable row */
            if (has_cost && fabs(TT(lp, m, q)) > TOL_COST) unbounded_if_feasible = 1;
            lp_drop_col(lp, q);
            continue;
        }
        lp_pivot(lp, r, q);
        lp->rowkind[r] = RK_FREE;
        if (was_eq) lp_drop_col(lp, q);
    }
    for (int i = 0; i < m; ++i) if (lp->rowkind[i] == RK_PRI) lp->rowkind[i] = RK_INEQ;
    /* stage B: remaining equality rows leave the basis */
    for (int i = 0; i < m; ++i) {
        if (lp->rowkind[i] != RK_EQ) continue;
        int q = -1; double best = TOL_PIV;
        for (int j = 1; j <= lp->na; ++j) { const double a = fabs(TT(lp, i, j)); if (a > best) { best = a; q = j; } }
        if (q < 0) {
            if (fabs(TT(lp, i, 0)) > TOL_FEAS) return ORC_LP_INFEASIBLE;
            lp->rowkind[i] = RK_DEAD;
            continue;
        }
        lp_pivot(lp, i, q);
        lp->rowkind[i] = RK_INEQ;
        lp_drop_col(lp, q);
    }
    /* phase 1 */
    {
        int r = -1; double mn = -TOL_FEAS;
        for (int i = 0; i < m; ++i)
            if (lp->rowkind[i] == RK_INEQ && TT(lp, i, 0) < mn) { mn = TT(lp, i, 0); r = i; }
        if (r >= 0) {
            const int q = ++lp->na;
            lp->colvar[q] = n + m;
            for (int i = 0; i <= m; ++i) TT(lp, i, q) = (i < m && lp->rowkind[i] == RK_INEQ) ? -1.0 : 0.0;
            lp_pivot(lp, r, q);
            lp->rowkind[r] = RK_X0;
            const int st = lp_primal(lp, r);
            if (st == 3) return ORC_LP_ITERLIMIT;
            if (st != 4) {
                /* x0 still basic */
                if (TT(lp, r, 0) > TOL_FEAS) return ORC_LP_INFEASIBLE;
                int qq = -1; double best = TOL_PIV;
                for (int j = 1; j <= lp->na; ++j) { const double a = fabs(TT(lp, r, j)); if (a > best) { best = a; qq = j; } }
                if (qq < 0) lp->rowkind[r] = RK_DEAD;
                else { lp_pivot(lp, r, qq); lp->rowkind[r] = RK_INEQ; lp_drop_col(lp, qq); }
            }
        }
    }
    /* phase 2 */
    if (has_cost) {
        if (unbounded_if_feasible) return ORC_LP_UNBOUNDED;
        const int st = lp_primal(lp, -1);
        if (st == 2) return ORC_LP_UNBOUNDED;
        if (st == 3) return ORC_LP_ITERLIMIT;
    }
    return ORC_LP_OPTIMAL;
}

/* (re)load the tableau from the problem data: power-of-two row scaling, zero rows, bookkeeping.
 * pri[i] != 0 marks an inequality row whose slack should be made nonbasic by stage A (basis refactorisation). */
static int lp_load(lp_t *lp, const double *A, const double *b, const double *c, const unsigned char *pri) {
    const int m = lp->m, n = lp->n;
    int consistent = 1;
    memset(lp->T, 0, sizeof(double) * (size_t)(m + 1) * lp->ld);
    for (int j = 1; j <= n; ++j) lp->colvar[j] = j - 1;
    lp->na = n;
    lp->growth = 0.0;
    for (int i = 0; i < m; ++i) {
        double mx = 0.0;
        for (int j = 0; j < n; ++j) { const double a = fabs(A[(size_t)i * n + j]); if (a > mx) mx = a; }
        lp->rowvar[i] = n + i;
        lp->rowkind[i] = lp->is_eq[i] ? RK_EQ : (pri && pri[i] ? RK_PRI : RK_INEQ);
        if (!(mx > 0.0)) {
            /* 0 <= b_i (or 0 == b_i) */
            if (lp->is_eq[i] ? fabs(b[i]) > TOL_FEAS : b[i] < -TOL_FEAS) consistent = 0;
            lp->rowkind[i] = RK_DEAD;
            continue;
        }
        int e; frexp(mx, &e);
        const double s = ldexp(1.0, -e);
        TT(lp, i, 0) = b[i] * s;
        for (int j = 0; j < n; ++j) TT(lp, i, j + 1) = A[(size_t)i * n + j] * s;
    }
    if (c) for (int j = 0; j < n; ++j) TT(lp, m, j + 1) = c[j];
    return consistent;
}

void orc_set_feas_tol(double tol) { g_tol_feas = tol; }

int orc_lp_solve(int m, int n, const double *A, const double *b, const double *c, int neq, const int32_t *eq,
                 double *x, double *obj, int32_t *iters) {
    lp_t lp;
    int status = ORC_LP_OPTIMAL;
    lp.m = m; lp.n = n; lp.ld = n + 3; lp.na = n; lp.iters = 0;
    lp.T = (double *)calloc((size_t)(m + 1) * lp.ld, sizeof(double));
    lp.colvar = (int *)calloc((size_t)n + 3, sizeof(int));
    lp.rowvar = (int *)calloc((size_t)m + 1, sizeof(int));
    lp.rowkind = (unsigned char *)calloc((size_t)m + 1, 1);
    lp.is_eq = (unsigned char *)calloc((size_t)m + 1, 1);
    unsigned char *pri = (unsigned char *)calloc((size_t)m + 1, 1);
    lp.max_iter = 50 * (m + n) + 100;
    for (int i = 0; i < neq; ++i)
        if (eq[i] >= 0 && eq[i] < m) lp.is_eq[eq[i]] = 1;
    for (int attempt = 0;; ++attempt) {
        if (!lp_load(&lp, A, b, c, attempt ? pri : NULL)) { status = ORC_LP_INFEASIBLE; break; }
        status = lp_run(&lp, c != NULL);
        if (status == ORC_LP_ITERLIMIT || !(lp.growth > GROWTH_SAFE) || attempt >= MAX_REFACTOR) break;
        /* numerically doubtful pivot sequence: rebuild the final basis from the original data and continue */
        memset(pri, 0, (size_t)m + 1);
        for (int j = 1; j <= lp.na; ++j) { const int v = lp.colvar[j]; if (v >= n && v < n + m) pri[v - n] = 1; }
    }
    if (status == ORC_LP_OPTIMAL) {
        if (x) {
            for (int j = 0; j < n; ++j) x[j] = 0.0;
            for (int i = 0; i < m; ++i)
                if (lp.rowkind[i] == RK_FREE) x[lp.rowvar[i]] = TT(&lp, i, 0);
        }
        if (obj) *obj = -TT(&lp, m, 0);
    }
    if (iters) *iters = lp.iters;
    free(lp.T); free(lp.colvar); free(lp.rowvar); free(lp.rowkind); free(lp.is_eq); free(pri);
    return status;
}

/* ------------------------------------------------------------------------------------------------
 * Dense linear algebra helpers
 * ---------------------------------------------------------------------------------------------- */

/* One-sided (Hestenes) Jacobi on the columns of G (rows x cols, column j at G[j*rows..]).  On return the
 * columns are mutually orthogonal; W (cols x cols, column major, may be NULL) accumulates the rotations. */
static void jacobi_orthogonalise(double *G, int rows, int cols, double *W) {
    if (W) { memset(W, 0, sizeof(double) * cols * cols); for (int i = 0; i < cols; ++i) W[i * cols + i] = 1.0; }
    for (int sweep = 0; sweep < 60; ++sweep) {
        int rotated = 0;
        for (int p = 0; p < cols - 1; ++p)
            for (int q = p + 1; q < cols; ++q) {
                double *gp = G + (size_t)p * rows, *gq = G + (size_t)q * rows;
                double app = 0, aqq = 0, apq = 0;
                for (int i = 0; i < rows; ++i) { app += gp[i] * gp[i]; aqq += gq[i] * gq[i]; apq += gp[i] * gq[i]; }
                if (apq == 0.0 || fabs(apq) <= 1e-15 * sqrt(app * aqq)) continue;
                rotated = 1;
                const double zeta = (aqq - app) / (2.0 * apq);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double cs = 1.0 / sqrt(1.0 + t * t), sn = cs * t;
                for (int i = 0; i < rows; ++i) {
                    const double a = gp[i], bq = gq[i];
                    gp[i] = cs * a - sn * bq; gq[i] = sn * a + cs * bq;
                }
                if (W) {
                    double *wp = W + (size_t)p * cols, *wq = W + (size_t)q * cols;
                    for (int i = 0; i < cols; ++i) {
                        const double a = wp[i], bq = wq[i];
                        wp[i] = cs * a - sn * bq; wq[i] = sn * a + cs * bq;
                    }
                }
            }
        if (!rotated) break;
    }
}

int orc_singular_values(const double *M, int m, int n, double *sv) {
    /* orthogonalise the shorter side: singular values of M == those of M^T */
    const int tall = m >= n;
    const int rows = tall ? m : n, cols = tall ? n : m;
    double *G = (double *)malloc(sizeof(double) * rows * cols);
    for (int j = 0; j < cols; ++j)
        for (int i = 0; i < rows; ++i) G[(size_t)j * rows + i] = tall ? M[(size_t)i * n + j] : M[(size_t)j * n + i];
    jacobi_orthogonalise(G, rows, cols, NULL);
    for (int j = 0; j < cols; ++j) {
        double s = 0; for (int i = 0; i < rows; ++i) s += G[(size_t)j * rows + i] * G[(size_t)j * rows + i];
        sv[j] = sqrt(s);
    }
    for (int i = 0; i < cols; ++i) /* descending */
        for (int j = i + 1; j < cols; ++j) if (sv[j] > sv[i]) { const double t = sv[i]; sv[i] = sv[j]; sv[j] = t; }
    free(G);
    return cols;
}

/* constraint_utilities.py:222-236 -> numpy.linalg.matrix_rank: count(S > S.max() * max(M,N) * eps) */
int orc_is_full_rank(const double *A, int n_cols, const int32_t *rows, int k) {
    if (k == 0) return 1;
    double *M = (double *)malloc(sizeof(double) * k * n_cols);
    double *sv = (double *)malloc(sizeof(double) * (k > n_cols ? k : n_cols));
    for (int i = 0; i < k; ++i) memcpy(M + (size_t)i * n_cols, A + (size_t)rows[i] * n_cols, sizeof(double) * n_cols);
    const int ns = orc_singular_values(M, k, n_cols, sv);
    const double tol = sv[0] * (double)(k > n_cols ? k : n_cols) * DBL_EPSILON;
    int rank = 0;
    for (int i = 0; i < ns; ++i) if (sv[i] > tol) rank++;
    free(M); free(sv);
    return rank == k;
}

/* LU with partial pivoting, nrhs right-hand sides (numpy.linalg.solve -> LAPACK gesv). 1 if exactly singular. */
static int lu_solve(double *M, int n, double *B, int nrhs) {
    for (int col = 0; col < n; ++col) {
        int piv = col; double best = fabs(M[(size_t)col * n + col]);
        for (int i = col + 1; i < n; ++i) { const double a = fabs(M[(size_t)i * n + col]); if (a > best) { best = a; piv = i; } }
        if (best == 0.0) return 1;
        if (piv != col) {
            for (int j = 0; j < n; ++j) { const double t = M[(size_t)col * n + j]; M[(size_t)col * n + j] = M[(size_t)piv * n + j]; M[(size_t)piv * n + j] = t; }
            for (int j = 0; j < nrhs; ++j) { const double t = B[(size_t)col * nrhs + j]; B[(size_t)col * nrhs + j] = B[(size_t)piv * nrhs + j]; B[(size_t)piv * nrhs + j] = t; }
        }
        const double d = M[(size_t)col * n + col];
        for (int i = col + 1; i < n; ++i) {
            const double f = M[(size_t)i * n + col] / d;
            if (f == 0.0) continue;
            for (int j = col + 1; j < n; ++j) M[(size_t)i * n + j] -= f * M[(size_t)col * n + j];
            for (int j = 0; j < nrhs; ++j) B[(size_t)i * nrhs + j] -= f * B[(size_t)col * nrhs + j];
        }
    }
    for (int i = n - 1; i >= 0; --i)
        for (int j = 0; j < nrhs; ++j) {
            double s = B[(size_t)i * nrhs + j];
            for (int l = i + 1; l < n; ++l) s -= M[(size_t)i * n + l] * B[(size_t)l * nrhs + j];
            B[(size_t)i * nrhs + j] = s / M[(size_t)i * n + i];
        }
    return 0;
}

/* numpy.linalg.pinv(M) for a k x n matrix (default rcond 1e-15); out is n x k row major */
static void pinv(const double *M, int k, int n, double *out) {
    /* orthogonalise the columns of M^T (n x k): M^T W = B  =>  pinv(M) = sum_i b_i w_i^T / sigma_i^2 */
    double *G = (double *)malloc(sizeof(double) * n * k), *W = (double *)malloc(sizeof(double) * k * k);
    double *s2 = (double *)malloc(sizeof(double) * k);
    for (int j = 0; j < k; ++j) for (int i = 0; i < n; ++i) G[(size_t)j * n + i] = M[(size_t)j * n + i];
    jacobi_orthogonalise(G, n, k, W);
    double smax = 0;
    for (int j = 0; j < k; ++j) { double s = 0; for (int i = 0; i < n; ++i) s += G[(size_t)j * n + i] * G[(size_t)j * n + i]; s2[j] = s; if (s > smax) smax = s; }
    memset(out, 0, sizeof(double) * n * k);
    for (int j = 0; j < k; ++j) {
        if (!(sqrt(s2[j]) > 1e-15 * sqrt(smax))) continue;
        for (int i = 0; i < n; ++i) for (int l = 0; l < k; ++l) out[(size_t)i * k + l] += G[(size_t)j * n + i] * W[(size_t)j * k + l] / s2[j];
    }
    free(G); free(W); free(s2);
}

/* ------------------------------------------------------------------------------------------------
 * Program primitives
 * ---------------------------------------------------------------------------------------------- */

/* mplp_program.py:411-444 */
int orc_check_feasibility(const orc_problem *p, const int32_t *as, int k, int check_rank) {
    if (check_rank && !orc_is_full_rank(p->A, p->n_x, as, k)) return 0;
    const int m = p->n_c + p->n_tc, n = p->n_x + p->n_t;
    double *A = (double *)calloc((size_t)m * n, sizeof(double)), *b = (double *)malloc(sizeof(double) * m);
    for (int i = 0; i < p->n_c; ++i) {
        for (int j = 0; j < p->n_x; ++j) A[(size_t)i * n + j] = p->A[(size_t)i * p->n_x + j];
        for (int j = 0; j < p->n_t; ++j) A[(size_t)i * n + p->n_x + j] = -p->F[(size_t)i * p->n_t + j];
        b[i] = p->b[i];
    }
    for (int i = 0; i < p->n_tc; ++i) {
        for (int j = 0; j < p->n_t; ++j) A[(size_t)(p->n_c + i) * n + p->n_x + j] = p->A_t[(size_t)i * p->n_t + j];
        b[p->n_c + i] = p->b_t[i];
    }
    const int st = orc_lp_solve(m, n, A, b, NULL, k, as, NULL, NULL, NULL);
    free(A); free(b);
    if (st == ORC_LP_ITERLIMIT) return -1;
    return st == ORC_LP_OPTIMAL;
}

/* mpqp_program.py:203-322 (mpQP) and mplp_program.py:446-569 (mpLP: zero block instead of Q, and
 * "not optimal" unless |active set| == n_x, line 472-473) */
int orc_check_optimality(const orc_problem *p, const int32_t *as, int k) {
    const int nx = p->n_x, nt = p->n_t, nc = p->n_c, ntc = p->n_tc, e = p->n_eq;
    if (!p->is_qp && k != nx) return 0;
    const int nact = k - e, nin = nc - k;
    const int n = nx + nt + nc + 1; /* x | theta | lambda(k) | slack(nin) | t */
    const int m = nx + k + nin + (nact > 0 ? nact : 0) + nin + 1 + (nact > 0 ? nact : 0) + nin + ntc;
    char *in_as = (char *)calloc((size_t)nc + 1, 1);
    for (int i = 0; i < k; ++i) in_as[as[i]] = 1;
    int *inact = (int *)malloc(sizeof(int) * (nc + 1));
    { int c = 0; for (int i = 0; i < nc; ++i) if (!in_as[i]) inact[c++] = i; }
    double *A = (double *)calloc((size_t)m * n, sizeof(double)), *b = (double *)calloc((size_t)m, sizeof(double));
    double *c = (double *)calloc((size_t)n, sizeof(double));
    const int oT = nx, oL = nx + nt, oS = nx + nt + k, ot = n - 1;
    int row = 0;
    /* 1) Q x + H theta + A_as^T lambda = -c */
    for (int i = 0; i < nx; ++i, ++row) {
        if (p->is_qp) for (int j = 0; j < nx; ++j) A[(size_t)row * n + j] = p->Q[(size_t)i * nx + j];
        for (int j = 0; j < nt; ++j) A[(size_t)row * n + oT + j] = p->H[(size_t)i * nt + j];
        for (int j = 0; j < k; ++j) A[(size_t)row * n + oL + j] = p->A[(size_t)as[j] * nx + i];
        b[row] = -p->c[i];
    }
    /* 2) A_as x - F_as theta = b_as */
    for (int i = 0; i < k; ++i, ++row) {
        for (int j = 0; j < nx; ++j) A[(size_t)row * n + j] = p->A[(size_t)as[i] * nx + j];
        for (int j = 0; j < nt; ++j) A[(size_t)row * n + oT + j] = -p->F[(size_t)as[i] * nt + j];
        b[row] = p->b[as[i]];
    }
    /* 3) A_J x - F_J theta + s = b_J */
    for (int i = 0; i < nin; ++i, ++row) {
        for (int j = 0; j < nx; ++j) A[(size_t)row * n + j] = p->A[(size_t)inact[i] * nx + j];
        for (int j = 0; j < nt; ++j) A[(size_t)row * n + oT + j] = -p->F[(size_t)inact[i] * nt + j];
        A[(size_t)row * n + oS + i] = 1.0;
        b[row] = p->b[inact[i]];
    }
    /* 4) t <= lambda_i, activated (non-equality) multipliers */
    for (int i = 0; i < nact; ++i, ++row) { A[(size_t)row * n + oL + e + i] = -1.0; A[(size_t)row * n + ot] = 1.0; }
    /* 5) t <= s_j */
    for (int i = 0; i < nin; ++i, ++row) { A[(size_t)row * n + oS + i] = -1.0; A[(size_t)row * n + ot] = 1.0; }
    /* 6) t >= 0 */
    A[(size_t)row * n + ot] = -1.0; ++row;
    /* 7) lambda >= 0 */
    for (int i = 0; i < nact; ++i, ++row) A[(size_t)row * n + oL + e + i] = -1.0;
    /* 8) s >= 0 */
    for (int i = 0; i < nin; ++i, ++row) A[(size_t)row * n + oS + i] = -1.0;
    /* 9) A_t theta <= b_t */
    for (int i = 0; i < ntc; ++i, ++row) {
        for (int j = 0; j < nt; ++j) A[(size_t)row * n + oT + j] = p->A_t[(size_t)i * nt + j];
        b[row] = p->b_t[i];
    }
    c[ot] = -1.0;
    int neq = nx + nc; /* lp_active_limit, lines 301-306 */
    if (k == 0) neq = nc;
    int32_t *eq = (int32_t *)malloc(sizeof(int32_t) * (neq + 1));
    for (int i = 0; i < neq; ++i) eq[i] = i;
    const int st = orc_lp_solve(row, n, A, b, c, neq, eq, NULL, NULL, NULL);
    free(in_as); free(inact); free(A); free(b); free(c); free(eq);
    if (st == ORC_LP_ITERLIMIT) return -1;
    return st == ORC_LP_OPTIMAL;
}

/* mpqp_program.py:146-198 (KKT solve, two numpy.linalg.solve calls on the same matrix) and
 * mplp_program.py:372-395 (pinv form) */
int orc_optimal_control_law(const orc_problem *p, const int32_t *as, int k, double *A_x, double *b_x, double *A_l,
                            double *b_l) {
    const int nx = p->n_x, nt = p->n_t;
    if (p->is_qp) {
        const int n = nx + k, nr = nt + 1;
        double *M = (double *)calloc((size_t)n * n, sizeof(double)), *B = (double *)calloc((size_t)n * nr, sizeof(double));
        for (int i = 0; i < k; ++i) {
            for (int j = 0; j < nx; ++j) M[(size_t)i * n + j] = p->A[(size_t)as[i] * nx + j];
            B[(size_t)i * nr] = p->b[as[i]];
            for (int j = 0; j < nt; ++j) B[(size_t)i * nr + 1 + j] = p->F[(size_t)as[i] * nt + j];
        }
        for (int i = 0; i < nx; ++i) {
            for (int j = 0; j < nx; ++j) M[(size_t)(k + i) * n + j] = p->Q[(size_t)i * nx + j];
            for (int j = 0; j < k; ++j) M[(size_t)(k + i) * n + nx + j] = p->A[(size_t)as[j] * nx + i];
            B[(size_t)(k + i) * nr] = -p->c[i];
            for (int j = 0; j < nt; ++j) B[(size_t)(k + i) * nr + 1 + j] = -p->H[(size_t)i * nt + j];
        }
        const int sing = lu_solve(M, n, B, nr);
        if (!sing) {
            for (int i = 0; i < nx; ++i) { b_x[i] = B[(size_t)i * nr]; for (int j = 0; j < nt; ++j) A_x[(size_t)i * nt + j] = B[(size_t)i * nr + 1 + j]; }
            for (int i = 0; i < k; ++i) { b_l[i] = B[(size_t)(nx + i) * nr]; for (int j = 0; j < nt; ++j) A_l[(size_t)i * nt + j] = B[(size_t)(nx + i) * nr + 1 + j]; }
        }
        free(M); free(B);
        return sing;
    }
    double *M = (double *)malloc(sizeof(double) * (k > 0 ? k : 1) * nx), *P = (double *)malloc(sizeof(double) * nx * (k > 0 ? k : 1));
    for (int i = 0; i < k; ++i) memcpy(M + (size_t)i * nx, p->A + (size_t)as[i] * nx, sizeof(double) * nx);
    pinv(M, k, nx, P); /* nx x k */
    for (int i = 0; i < nx; ++i) {
        double s = 0; for (int l = 0; l < k; ++l) s += P[(size_t)i * k + l] * p->b[as[l]];
        b_x[i] = s;
        for (int j = 0; j < nt; ++j) { s = 0; for (int l = 0; l < k; ++l) s += P[(size_t)i * k + l] * p->F[(size_t)as[l] * nt + j]; A_x[(size_t)i * nt + j] = s; }
    }
    for (int l = 0; l < k; ++l) {
        double s = 0; for (int i = 0; i < nx; ++i) s += P[(size_t)i * k + l] * p->c[i];
        b_l[l] = -s;
        for (int j = 0; j < nt; ++j) { s = 0; for (int i = 0; i < nx; ++i) s += P[(size_t)i * k + l] * p->H[(size_t)i * nt + j]; A_l[(size_t)l * nt + j] = -s; }
    }
    free(M); free(P);
    return 0;
}

int64_t orc_region_doubles(const orc_problem *p) {
    const int64_t kmax = p->n_c, emax = p->n_c + p->n_tc;
    return (int64_t)p->n_x * p->n_t + p->n_x + kmax * p->n_t + kmax + emax * p->n_t + emax;
}
int64_t orc_region_ints(const orc_problem *p) { return 5 + (int64_t)p->n_c + p->n_tc + p->n_c + p->n_c + p->n_c; }

/* mpqp_utils.py:89-195 (general) and :198-320 (one parameter) */
int orc_gen_cr(const orc_problem *p, const int32_t *as, int k, double *rec_d, int32_t *rec_i) {
    const int nx = p->n_x, nt = p->n_t, nc = p->n_c, ntc = p->n_tc, e = p->n_eq;
    const int kmax = nc, emax = nc + ntc;
    double *A_x = rec_d, *b_x = A_x + (size_t)nx * nt, *A_l = b_x + nx, *b_l = A_l + (size_t)kmax * nt;
    double *E = b_l + kmax, *f = E + (size_t)emax * nt;
    int32_t *hdr = rec_i, *act = rec_i + 5, *omega = act + kmax, *lam = omega + ntc, *ridx = lam + kmax, *rcon = ridx + nc;
    memset(rec_d, 0, sizeof(double) * orc_region_doubles(p));
    for (int64_t i = 0; i < orc_region_ints(p); ++i) rec_i[i] = -1;
    if (orc_optimal_control_law(p, as, k, A_x, b_x, A_l, b_l)) return ORC_SINGULAR_KKT;

    char *in_as = (char *)calloc((size_t)nc + 1, 1);
    for (int i = 0; i < k; ++i) in_as[as[i]] = 1;
    int *inact = (int *)malloc(sizeof(int) * (nc + 1));
    int nin = 0; for (int i = 0; i < nc; ++i) if (!in_as[i]) inact[nin++] = i;
    const int nlam = k - e, nrows = nlam + nin + ntc;
    double *CA = (double *)calloc((size_t)(nrows + 1) * (nt + 1), sizeof(double)), *Cb = (double *)calloc((size_t)nrows + 1, sizeof(double));
    int *kept = (int *)malloc(sizeof(int) * (nrows + 1)); /* original row index of kept row */
    /* rows: -A_l[e:] theta <= b_l[e:] ; (A_J A_x - F_J) theta <= b_J - A_J b_x ; A_t theta <= b_t   (lines 111-121) */
    for (int i = 0; i < nlam; ++i) { for (int j = 0; j < nt; ++j) CA[(size_t)i * nt + j] = -A_l[(size_t)(e + i) * nt + j]; Cb[i] = b_l[e + i]; }
    for (int i = 0; i < nin; ++i) {
        const int r = nlam + i, ci = inact[i];
        for (int j = 0; j < nt; ++j) { double s = 0; for (int l = 0; l < nx; ++l) s += p->A[(size_t)ci * nx + l] * A_x[(size_t)l * nt + j]; CA[(size_t)r * nt + j] = s - p->F[(size_t)ci * nt + j]; }
        double s = 0; for (int l = 0; l < nx; ++l) s += p->A[(size_t)ci * nx + l] * b_x[l];
        Cb[r] = p->b[ci] - s;
    }
    for (int i = 0; i < ntc; ++i) { const int r = nlam + nin + i; for (int j = 0; j < nt; ++j) CA[(size_t)r * nt + j] = p->A_t[(size_t)i * nt + j]; Cb[r] = p->b_t[i]; }
    /* numerically_nonzero_rows (constraint_utilities.py:469-470), then scale_constraint (:25-35) */
    int nk = 0;
    for (int r = 0; r < nrows; ++r) {
        int nz = 0; for (int j = 0; j < nt; ++j) if (!(fabs(CA[(size_t)r * nt + j]) <= 1e-8)) nz = 1;
        if (!nz) continue;
        double s = 0; for (int j = 0; j < nt; ++j) s += CA[(size_t)r * nt + j] * CA[(size_t)r * nt + j];
        const double inv = 1.0 / sqrt(s);
        for (int j = 0; j < nt; ++j) CA[(size_t)nk * nt + j] = CA[(size_t)r * nt + j] * inv;
        Cb[nk] = Cb[r] * inv;
        kept[nk++] = r;
    }
    int verdict = ORC_REGION;
    int n_omega = 0, n_lambda = 0, n_reg = 0, nE = 0;
    if (nt == 1) {
        /* get_bounds_1d / is_full_dimensional_1d (lines 304-320) */
        double mn = -INFINITY, mx = INFINITY;
        for (int r = 0; r < nk; ++r) { const double v = Cb[r] / CA[r]; if (CA[r] > 0) { if (v < mx) mx = v; } else { if (v > mn) mn = v; } }
        if (!(mn + 1e-8 <= mx)) verdict = ORC_OPTIMAL_NO_REGION;
        else {
            for (int r = 0; r < nk; ++r) {
                const double v = Cb[r] / CA[r];
                if (!(mn <= v && v <= mx)) continue;
                const int o = kept[r];
                if (o < nlam) lam[n_lambda++] = as[e + o];
                else if (o < nlam + nin) { ridx[n_reg] = o - nlam; rcon[n_reg] = inact[o - nlam]; n_reg++; }
                else omega[n_omega++] = o - nlam - nin;
            }
            E[0] = 1.0; f[0] = mx; E[1] = -1.0; f[1] = -mn; nE = 2;
        }
    } else {
        /* is_full_dimensional -> chebyshev_ball (mpqp_utils.py:323-344, chebyshev_ball.py:10-63) */
        const int nv = nt + 1;
        double *BA = (double *)calloc((size_t)(nk + 1) * nv, sizeof(double)), *Bb = (double *)calloc((size_t)nk + 1, sizeof(double));
        double *cc = (double *)calloc((size_t)nv, sizeof(double)), *xs = (double *)calloc((size_t)nv, sizeof(double));
        for (int r = 0; r < nk; ++r) {
            double s = 0; for (int j = 0; j < nt; ++j) { BA[(size_t)r * nv + j] = CA[(size_t)r * nt + j]; s += CA[(size_t)r * nt + j] * CA[(size_t)r * nt + j]; }
            BA[(size_t)r * nv + nt] = sqrt(s); Bb[r] = Cb[r];
        }
        BA[(size_t)nk * nv + nt] = -1.0; cc[nt] = -1.0;
        int st = orc_lp_solve(nk + 1, nv, BA, Bb, cc, 0, NULL, xs, NULL, NULL);
        if (st == ORC_LP_ITERLIMIT) verdict = ORC_LP_LIMIT;
        else if (st != ORC_LP_OPTIMAL || !(xs[nt] > 1e-8)) verdict = ORC_OPTIMAL_NO_REGION;
        free(BA); free(Bb); free(cc); free(xs);
        if (verdict == ORC_REGION) {
            /* one LP per kept row with that row as an equality (lines 143-178) */
            for (int r = 0; r < nk && verdict == ORC_REGION; ++r) {
                const int32_t eqr = r;
                st = orc_lp_solve(nk, nt, CA, Cb, NULL, 1, &eqr, NULL, NULL, NULL);
                if (st == ORC_LP_ITERLIMIT) { verdict = ORC_LP_LIMIT; break; }
                if (st != ORC_LP_OPTIMAL) continue;
                const int o = kept[r];
                if (o < nlam) lam[n_lambda++] = as[e + o];
                else if (o < nlam + nin) { ridx[n_reg] = o - nlam; rcon[n_reg] = inact[o - nlam]; n_reg++; }
                else omega[n_omega++] = o - nlam - nin;
                for (int j = 0; j < nt; ++j) E[(size_t)nE * nt + j] = CA[(size_t)r * nt + j];
                f[nE++] = Cb[r];
            }
        }
    }
    hdr[0] = k; hdr[1] = nE; hdr[2] = n_omega; hdr[3] = n_lambda; hdr[4] = n_reg;
    for (int i = 0; i < k; ++i) act[i] = as[i];
    free(in_as); free(inact); free(CA); free(Cb); free(kept);
    return verdict;
}

/* mpqp_parrallel_combinatorial.py:17-64 */
int orc_full_process(const orc_problem *p, const int32_t *as, int k, double *rec_d, int32_t *rec_i) {
    const int feas = orc_check_feasibility(p, as, k, 1);
    if (feas < 0) return ORC_LP_LIMIT;
    if (!feas) return ORC_INFEASIBLE;
    const int opt = orc_check_optimality(p, as, k);
    if (opt < 0) return ORC_LP_LIMIT;
    if (!opt) return ORC_FEASIBLE;
    double *d = rec_d; int32_t *ii = rec_i;
    if (!d) d = (double *)malloc(sizeof(double) * orc_region_doubles(p));
    if (!ii) ii = (int32_t *)malloc(sizeof(int32_t) * orc_region_ints(p));
    const int v = orc_gen_cr(p, as, k, d, ii);
    if (!rec_d) free(d);
    if (!rec_i) free(ii);
    return v;
}

int orc_check_level(const orc_problem *p, const int32_t *cands, int64_t n, int k, int threads, uint8_t *status,
                    double *rec_d, int32_t *rec_i) {
    const int64_t sd = orc_region_doubles(p), si = orc_region_ints(p);
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
#pragma omp parallel
    {
        if (!t_arena) { t_arena = (char *)(malloc)(ORC_ARENA_BYTES); t_cap = t_arena ? ORC_ARENA_BYTES : 0; }   /* kept for the life of the thread */
        t_arena_on = 1;
#pragma omp for schedule(dynamic, 4)
        for (int64_t i = 0; i < n; ++i) {
            t_top = 0;
            status[i] = (uint8_t)orc_full_process(p, cands + i * k, k, rec_d ? rec_d + i * sd : NULL, rec_i ? rec_i + i * si : NULL);
        }
        t_arena_on = 0;
    }
    return 0;
}

/* solver_utils.py:15-55 (CombinationTester.check) and :154-166 (generate_children_sets) */
int64_t orc_generate_children(const orc_problem *p, const int32_t *cands, int64_t n, int k, const uint8_t *status,
                              const int32_t *pruned, const int32_t *pruned_k, int64_t n_pruned, int pruned_stride,
                              int mplp_filter, int32_t *out, int64_t cap) {
    const int nc = p->n_c;
    int64_t count = 0;
    char *member = (char *)calloc((size_t)nc + 1, 1);
    for (int64_t c = 0; c < n; ++c) {
        if (status[c] != ORC_FEASIBLE && status[c] != ORC_REGION) continue;
        const int32_t *as = cands + c * k;
        memset(member, 0, (size_t)nc + 1);
        for (int i = 0; i < k; ++i) member[as[i]] = 1;
        const int start = k > 0 ? as[k - 1] + 1 : 0;
        for (int i = start; i < nc; ++i) {
            member[i] = 1;
            int ok = 1;
            for (int64_t j = 0; j < n_pruned && ok; ++j) {
                const int32_t *ps = pruned + j * pruned_stride;
                int sub = 1;
                for (int l = 0; l < pruned_k[j]; ++l) if (!member[ps[l]]) { sub = 0; break; }
                if (sub) ok = 0;
            }
            member[i] = 0;
            if (!ok) continue;
            /* mpLP child filter (driver lines 49-51): only for parents that were not optimal */
            if (mplp_filter && status[c] == ORC_FEASIBLE && i >= (k + 1) + nc - p->n_x) continue;
            if (out && count < cap) { memcpy(out + count * (k + 1), as, sizeof(int32_t) * k); out[count * (k + 1) + k] = i; }
            count++;
        }
    }
    free(member);
    return count;
}
