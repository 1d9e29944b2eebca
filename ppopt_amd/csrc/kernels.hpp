// kernels.hpp -- the batched per-candidate kernels of the combinatorial mpLP/mpQP path (gfx950).
//
// One 64-lane wavefront (= one 64-thread workgroup) owns one candidate active set from start to finish;
// work is pulled from an atomic queue so that wavefronts whose LPs terminate early pick up the next candidate.
// Everything a candidate touches after the initial gathers lives in that wavefront's LDS slice:
//   ints   active set, inactive list, simplex bookkeeping
//   KKT    gathered Schur block S = W[as,as] (+ right-hand sides) or the dense KKT matrix
//   T      one simplex tableau, reused by every LP of the candidate
// Shared, read-only problem blocks (A|b|F, W, UV, G', the base tableau) are read from HBM/L2 with coalesced
// or broadcast loads.
//
// Reference functions replaced (relative to /root/reference/src/ppopt):
//   k_verdict   full_process up to the optimal/feasible decision: is_full_rank + check_feasibility
//               (mplp_program.py:411-444) and check_optimality (mpqp_program.py:203-322,
//               mplp_program.py:446-569) -- the latter in its reduced theta-space form (KKT solve, then an LP over
//               theta only; same rows as mpqp_utils.py:111-121, same test as mpqp_combi_graph.py:48-66)
//   k_region    gen_cr_from_active_set (utils/mpqp_utils.py:89-320): optimal_control_law, region rows, zero-row
//               filter, scaling, Chebyshev-ball full-dimension test (chebyshev_ball.py:10-63), one LP per facet
//   k_children* generate_children_sets + CombinationTester.check (mp_solvers/solver_utils.py:15-55,154-166)
#pragma once
#include <stdint.h>

#include "kkt.hpp"

// Every kernel of kernels.hpp / kernels2.hpp is declared MPC_GLOBAL void MPC_LB(bounds) name(...).  In the library's main
// translation unit these are ordinary kernels.  batch_level.hip defines MPC_GLOBAL as an inlined device function before it
// includes the headers: there the bodies are called from kernels that take one argument table per MEMBER PROGRAM of a batch
// (blockIdx.y = member), so one launch runs a stage of a level for many programs -- the same code, statement for statement.
#ifndef MPC_GLOBAL
#define MPC_GLOBAL __global__
#define MPC_LB(...) __launch_bounds__(__VA_ARGS__)
#endif

namespace mpc {

constexpr int ST_INFEASIBLE = 0, ST_FEASIBLE = 1, ST_OPT_NO_REGION = 2, ST_REGION = 3, ST_SINGULAR = 4, ST_LP_LIMIT = 5;
constexpr int ST_OPT_PENDING = 6;  // internal: optimal, waiting for k_region
constexpr double ZERO_ROW_ATOL = 1e-8;  // numerically_nonzero_rows, constraint_utilities.py:469-470
constexpr double FULL_DIM_RADIUS = 1e-8;  // is_full_dimensional, mpqp_utils.py:342-344

struct DevProblem {
    int n_x, n_t, n_c, n_eq, n_tc, is_qp, kkt_mode;
    const double *A, *b, *F, *c, *H, *Q, *A_t, *b_t;
    const double *W;    // n_c x n_c            A Q^-1 A'
    const double *UV;   // n_c x (n_t+1)        [A Q^-1 c + b | A Q^-1 H + F]
    const double *Gt;   // n_c x n_x            row i = (Q^-1 A_i')'
    const double *X0H;  // n_x x (n_t+1)        [-Q^-1 c | -Q^-1 H]
    const double *AAT;  // n_c x n_c            A A'  (Gram matrix of the rows: fast rank screen, kernels2.hpp)
    const double *AT;   // n_x x n_c            A' (column l of A contiguous: k_region2's lanes, one inactive row each, read it coalesced)
    const double *tvp;  // k_region2's view only: the zero-padded [tv_minv (NT x NT) | tv_theta | box lo | box hi] block of ThetaArgs (staged in LDS)
    const double *base; // (n_c+n_tc) x (1+n_x+n_t)   [b | A | -F ; b_t | 0 | A_t]
    // pre-crashed dictionary of the (x,theta) LP at a feasible vertex of the base polytope (built once per program):
    // basic slack rows d0_rows (n_d0r) in terms of the nonbasic inequality slacks d0_cols (n_d0c); program
    // equalities are already eliminated.  d0: n_d0r x (1 + n_d0c), column 0 = value at the vertex.
    const double *d0; const int *d0_rows; const int *d0_cols; int n_d0r, n_d0c, has_d0;
    const double *d0T;  // the same dictionary stored column-major ((1 + n_d0c) x n_d0r): lane i reads row i coalesced
    // vertex of the parameter polytope {A_t theta <= b_t} (kernels2.hpp): theta = tv_theta - tv_minv * sigma, sigma = slacks
    // of its n_t tight rows; tv_rows = the remaining n_tpre rows of A_t already expressed in sigma ([value | coefficients])
    const double *tv_theta, *tv_minv, *tv_rows; int n_tpre, has_tv;
    const int *tv_tight;  // the n_t rows of A_t that are tight at the vertex (sigma_j is the slack of row tv_tight[j])
    // LDS layout (offsets in doubles from the start of dynamic LDS; ints follow the doubles)
    int kmax;           // largest cardinality with a solvable KKT (= min(n_c, n_x))
    int ld_x, ld_t;     // odd tableau strides of the (x,theta) LP and of the theta-space LPs
    int off_T, off_K, off_L, off_E, off_X, n_doubles;  // T tableau, K kkt matrix, L multipliers, E region rows, X x-law
    int off_as, off_inact, off_colvar, off_rowvar, off_rowkind, off_kept, off_pri, off_stored, n_ints;
};

struct LevelCounters {
    unsigned long long status[8];
    unsigned long long pivots;
    unsigned long long xtheta_lps;  // candidates that needed the large (x,theta) LP
    unsigned long long xtheta_fallbacks;  // ... of which the warm start from the pre-crashed vertex was abandoned
    unsigned long long x_cached;    // (x,theta) solves that started from the parent's cached dictionary
    unsigned long long rcycles[6];  // k_region2: rows, chebyshev, facets, record wave-cycles; refactors; facet pivots
    unsigned long long cycles[8];   // wave-cycles (s_memtime) spent in: KKT solve, theta LP, (x,theta) LP, region build; [4] theta rows, [5] theta stage 2; [6],[7] candidates decided by the box screen (stage 1 / multiplier row)
    unsigned int work_verdict, work_region, n_opt, n_pruned_new, work_retry, work_x, e_rows, work_r2, work_q, n_retry_theta, n_rretry;
    unsigned long long r_box;       // k_region2: region rows removed by the bounding-box screen
    unsigned long long r2_not_t0, r2_t1;  // k_region2: ~(wall clock of the first wavefront's start), wall clock of the last one's end
    unsigned long long xq_pivots;   // k_xq / k_xq_grouped: product-form iterations executed (each reads one column and one row of the parent's record)
    unsigned int xq_thread, pad_xq;  // candidates of the last level's quick test decided by k_xq_thread (round 5); pad_xq: ... of which from another parent's record
    // Round 6: the queue between the theta stage and the region stage of a large last level.  k_theta2 appends every candidate it finds
    // optimal (q_tail; the entries live in the level's opt_list, -1 until written), region wavefronts claim positions (work_r2 is the
    // head); q_closed is raised by a one-thread launch behind the theta kernel: the tail is final.
    unsigned int q_tail, q_closed, q_fault, q_early;   // q_fault: a claimed entry never arrived (never observed); q_early: regions built by the early launch
    unsigned int x_second, pad_x2;   // k_x2 (round 6): doubtful cached runs repeated from D0 inside the kernel (DictCache::second_max)
};

struct Smem {
    double *T, *K, *L, *E, *X;
    int *as, *inact, *colvar, *rowvar, *rowkind, *kept, *pri, *stored;
};

__device__ __forceinline__ Smem carve(const DevProblem &P, double *base) {
    Smem s;
    s.T = base + P.off_T;
    s.K = base + P.off_K;
    s.L = base + P.off_L;
    s.E = base + P.off_E;
    s.X = base + P.off_X;
    int *ib = reinterpret_cast<int *>(base + P.n_doubles);
    s.as = ib + P.off_as;
    s.inact = ib + P.off_inact;
    s.colvar = ib + P.off_colvar;
    s.rowvar = ib + P.off_rowvar;
    s.rowkind = ib + P.off_rowkind;
    s.kept = ib + P.off_kept;
    s.pri = ib + P.off_pri;
    s.stored = ib + P.off_stored;
    return s;
}

// active set -> LDS, complement -> inact[] (ascending).  Returns the number of inactive rows.
__device__ inline int load_active_set(const DevProblem &P, const int32_t *cand, int k, Smem &s) {
    const int lane = lane_id();
    wave_sync();
    for (int i = lane; i < k; i += 64) s.as[i] = cand[i];
    wave_sync();
    int base = 0;
    for (int j0 = 0; j0 < P.n_c; j0 += 64) {
        const int j = j0 + lane;
        bool inactive = j < P.n_c;
        if (inactive)
            for (int i = 0; i < k; ++i) inactive = inactive && (s.as[i] != j);
        const unsigned long long bal = __ballot(inactive);
        if (inactive) s.inact[base + __popcll(bal & ((1ull << lane) - 1ull))] = j;
        base += __popcll(bal);
    }
    wave_sync();
    return base;
}

// KKT solve.  On success L (k x (n_t+1)) holds [b_l | A_l]; in mode 1 X (n_x x (n_t+1)) also holds [b_x | A_x].
// returns 0 ok, 1 rank deficient active set, 2 singular KKT, 3 mpLP active set that is not a vertex
__device__ inline int kkt_solve(const DevProblem &P, int k, Smem &s, bool *ill = nullptr) {
    const int lane = lane_id(), nr = P.n_t + 1, nx = P.n_x, nc = P.n_c;
    if (k > nx) return 1;
    if (k == 0) {
        if (P.kkt_mode == 1) {
            if (!P.is_qp) return 3;
            // x = -Q^-1 (c + H theta): solve Q X = -[c | H]
            for (int idx = lane; idx < nx * nx; idx += 64) s.K[idx] = P.Q[idx];
            for (int idx = lane; idx < nx * nr; idx += 64) {
                const int i = idx / nr, t = idx % nr;
                s.X[idx] = t == 0 ? -P.c[i] : -P.H[i * P.n_t + t - 1];
            }
            wave_sync();
            if (!lu_solve(s.K, nx, s.X, nr, KKT_SING_TOL)) return 2;
        }
        return 0;
    }
    if (P.kkt_mode == 0) {
        {
            // is_full_rank(A, as) on the rows themselves (exact structural dependencies show up as ~1e-16 pivots)
            double *M = s.K;
            if (!full_row_rank(M, k, nx, [&] { for (int idx = lane; idx < k * nx; idx += 64) M[idx] = P.A[s.as[idx / nx] * nx + idx % nx]; })) return 1;
            wave_sync();
        }
        double *S = s.K, *diag = s.K + k * k;
        for (int idx = lane; idx < k * k; idx += 64) {
            const int i = idx / k, j = idx % k;
            S[idx] = P.W[s.as[i] * nc + s.as[j]];
        }
        for (int i = lane; i < k; i += 64) diag[i] = P.W[s.as[i] * nc + s.as[i]];
        for (int idx = lane; idx < k * nr; idx += 64) {
            const int i = idx / nr, t = idx % nr;
            s.L[idx] = -P.UV[s.as[i] * nr + t];
        }
        wave_sync();
        return chol_solve(S, k, s.L, nr, diag, ill) ? 0 : 2;
    }
    // mode 1: rank test on A_as, then the dense KKT system
    {
        double *M = s.K;
        if (!full_row_rank(M, k, nx, [&] { for (int idx = lane; idx < k * nx; idx += 64) M[idx] = P.A[s.as[idx / nx] * nx + idx % nx]; })) return 1;
    }
    if (!P.is_qp && k != nx) return 3;  // mpLP: only a vertex (n_x active rows) can be optimal (mplp_program.py:472-473)
    const int n = nx + k;
    double *M = s.K, *B = s.K + n * n;
    for (int idx = lane; idx < n * n; idx += 64) {
        const int i = idx / n, j = idx % n;
        double v = 0.0;
        if (i < k) { if (j < nx) v = P.A[s.as[i] * nx + j]; }
        else if (j < nx) { if (P.is_qp) v = P.Q[(i - k) * nx + j]; }
        else v = P.A[s.as[j - nx] * nx + (i - k)];
        M[idx] = v;
    }
    for (int idx = lane; idx < n * nr; idx += 64) {
        const int i = idx / nr, t = idx % nr;
        double v;
        if (i < k) v = t == 0 ? P.b[s.as[i]] : P.F[s.as[i] * P.n_t + t - 1];
        else v = t == 0 ? -P.c[i - k] : -P.H[(i - k) * P.n_t + t - 1];
        B[idx] = v;
    }
    wave_sync();
    if (!lu_solve(M, n, B, nr, KKT_SING_TOL)) return 2;
    for (int idx = lane; idx < nx * nr; idx += 64) s.X[idx] = B[idx];
    for (int idx = lane; idx < k * nr; idx += 64) s.L[idx] = B[nx * nr + idx];
    wave_sync();
    return 0;
}

// x*(theta) = X[:,0] + X[:,1:] theta for mode 0:  [b_x | A_x] = X0H - sum_a G'[as[a]] (x) L[a]
__device__ inline void x_law_schur(const DevProblem &P, int k, Smem &s) {
    const int lane = lane_id(), nr = P.n_t + 1, nx = P.n_x;
    for (int idx = lane; idx < nx * nr; idx += 64) {
        const int i = idx / nr, t = idx % nr;
        double acc = P.X0H[idx];
        for (int a = 0; a < k; ++a) acc = fma(-P.Gt[s.as[a] * nx + i], s.L[a * nr + t], acc);
        s.X[idx] = acc;
    }
    wave_sync();
}

// Rows of the critical-region polytope in the order of mpqp_utils.py:111-121:
//   lambda rows  -A_l[e:] theta <= b_l[e:]  |  inactive rows (A_J A_x - F_J) theta <= b_J - A_J b_x  |  A_t theta <= b_t
// written to dst (row stride ld, column 0 = rhs, columns 1..n_t = coefficients).  use_x: form the inactive rows
// from the x-law (reference arithmetic); otherwise from the Schur blocks (no x-law needed: verdict kernel).
__device__ inline int build_theta_rows(const DevProblem &P, int k, int nin, Smem &s, double *dst, int ld, bool use_x) {
    const int lane = lane_id(), nt = P.n_t, nr = nt + 1, nx = P.n_x, e = P.n_eq, nlam = k - e;
    for (int idx = lane; idx < nlam * nr; idx += 64) {
        const int i = idx / nr, t = idx % nr;
        const double v = s.L[(e + i) * nr + t];
        dst[i * ld + t] = t == 0 ? v : -v;
    }
    for (int idx = lane; idx < nin * nr; idx += 64) {
        const int i = idx / nr, t = idx % nr, ci = s.inact[i];
        double v;
        if (use_x) {
            // t == 0: b_J - A_J b_x ;  t > 0: A_J A_x - F_J
            double acc = 0.0;
            for (int l = 0; l < nx; ++l) acc = fma(P.A[ci * nx + l], s.X[l * nr + t], acc);
            v = t == 0 ? P.b[ci] - acc : acc - P.F[ci * nt + t - 1];
        } else {
            // slack_j(theta) = UV[j] + W[j,as] L  >= 0   <=>   -(coef) theta <= const
            double acc = P.UV[ci * nr + t];
            for (int a = 0; a < k; ++a) acc = fma(P.W[ci * P.n_c + s.as[a]], s.L[a * nr + t], acc);
            v = t == 0 ? acc : -acc;
        }
        dst[(nlam + i) * ld + t] = v;
    }
    for (int idx = lane; idx < P.n_tc * nr; idx += 64) {
        const int i = idx / nr, t = idx % nr;
        dst[(nlam + nin + i) * ld + t] = t == 0 ? P.b_t[i] : P.A_t[i * nt + t - 1];
    }
    wave_sync();
    return nlam + nin + P.n_tc;
}

// (x,theta) feasibility with the rows `as` active, started from the program's pre-crashed dictionary D0 (a feasible
// vertex of the base polytope): the slack of every active row is fixed at zero -- its column is deleted if it is
// nonbasic, otherwise it is pivoted out of the basis first -- and phase 1 repairs what that broke.  k - n_eq pivots
// replace the n_x + n_t crash pivots of the from-scratch LP, on a tableau without the free-variable rows.
// Returns LP_OPTIMAL / LP_INFEASIBLE / LP_ITERLIMIT, or -1 when a pivot was numerically doubtful (caller falls back).
__device__ inline int xtheta_from_vertex(const DevProblem &P, int k, Smem &s, Lp &lp) {
    const int lane = lane_id(), nv = P.n_x + P.n_t, mr = P.n_d0r, nc0 = P.n_d0c, cols = 1 + nc0, ld = P.ld_x;
    lp.T = s.T; lp.ld = ld; lp.colvar = s.colvar; lp.rowvar = s.rowvar; lp.rowkind = s.rowkind;
    lp.n = nv; lp.m = mr; lp.na = nc0; lp.growth = 0.0; lp.iters = 0; lp.max_iter = 50 * (mr + nc0) + 100;
    double *T = s.T;
    wave_sync();
    for (int idx = lane; idx < mr * cols; idx += 64) T[(idx / cols) * ld + idx % cols] = P.d0[idx];
    for (int j = lane; j <= nc0 + 1; j += 64) T[mr * ld + j] = 0.0;
    for (int i = lane; i < mr; i += 64) { s.rowkind[i] = RK_INEQ; s.rowvar[i] = nv + P.d0_rows[i]; }
    for (int j = lane; j < nc0; j += 64) s.colvar[1 + j] = nv + P.d0_cols[j];
    wave_sync();
    for (int a = P.n_eq; a < k; ++a) {
        const int v = nv + s.as[a];
        int q = -1;
        for (int j = 1 + lane; j <= lp.na; j += 64) if (s.colvar[j] == v) q = j;
        { double dummy = q >= 0 ? 1.0 : 0.0; reduce_max_first(dummy, q); }
        if (q >= 0) { lp_drop_col(lp, q); continue; }
        int r = -1;
        for (int i = lane; i < mr; i += 64) if (s.rowvar[i] == v && s.rowkind[i] == RK_INEQ) r = i;
        { double dummy = r >= 0 ? 1.0 : 0.0; reduce_max_first(dummy, r); }
        if (r < 0) return -1;
        q = lp_best_col(lp, r);
        if (q < 0) {
            if (fabs(T[r * ld]) > TOL_FEAS) return LP_INFEASIBLE;
            wave_sync();
            if (lane == 0) s.rowkind[r] = RK_DEAD;
            wave_sync();
            continue;
        }
        double colmax = 0.0;
        for (int i = lane; i < mr; i += 64) if (s.rowkind[i] != RK_DEAD) colmax = fmax(colmax, fabs(T[i * ld + q]));
        colmax = wave_max(colmax);
        lp.growth = fmax(lp.growth, colmax / fabs(T[r * ld + q]));
        lp_pivot(lp, r, q);
        lp_drop_col(lp, q);
    }
    const int st = lp_phase1(lp);
    if (st != LP_ITERLIMIT && lp.growth > GROWTH_SAFE) return -1;
    return st;
}

// ------------------------------------------------------------------------------------------------------------
// k_recession: "max t unbounded  =>  not optimal" for programs whose parameter set is open in some direction.
// ------------------------------------------------------------------------------------------------------------
// The reference's optimality LP MAXIMISES t subject to t <= lambda_i(theta) (activated rows), t <= slack_j(theta) (inactive rows),
// A_t theta <= b_t (mpqp_program.py:203-322, mplp_program.py:446-569) and its solver adapter returns None for every status but
// 'optimal' (solver_interface/cvxopt_interface.py:19-23): when the parameter set lets every one of those rows grow without bound the
// LP is unbounded and the active set counts as NOT optimal although its critical region is non-empty.  The verdict kernels decide
// "optimal" as "the rows' theta set is non-empty"; for the candidates they call optimal this kernel poses the recession question
//        exists d :   g_i' d >= 1  (every multiplier / slack row  h_i + g_i' theta),    A_t d <= 0
// -- the same coefficient rows as the theta LP, constants replaced by -1 / 0 -- and turns ST_OPT_PENDING into ST_FEASIBLE when it has a
// solution (also when there is no multiplier / slack row at all: t is then bounded by nothing).  A bounded parameter set has no such
// d, so the kernel is launched only for handles with mpc_handle::theta_open (config 1-5 never run it).
MPC_GLOBAL void MPC_LB(64) k_recession(DevProblem P, const int32_t *__restrict__ cands, long long n, int k, uint8_t *__restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    Smem s = carve(P, smem);
    const int lane = lane_id(), nt = P.n_t;
    for (long long c = blockIdx.x; c < n; c += gridDim.x) {
        if (status[c] != ST_OPT_PENDING) continue;      // (wave-uniform)
        const int nin = load_active_set(P, cands + (size_t)c * k, k, s);
        if (kkt_solve(P, k, s) != 0) continue;           // cannot happen: the verdict stage solved the same system
        Lp lp;
        lp.T = s.T; lp.ld = P.ld_t; lp.colvar = s.colvar; lp.rowvar = s.rowvar; lp.rowkind = s.rowkind;
        const int nlam = k - P.n_eq;
        lp.n = nt; lp.m = nlam + nin + P.n_tc; lp.iters = 0;
        const int m = lp.m;
        const int r = lp_solve(lp, false, s.pri, [&](const int *pri) {
            build_theta_rows(P, k, nin, s, s.T, P.ld_t, P.kkt_mode == 1);
            for (int i = lane; i <= m; i += 64) {
                double *Ti = s.T + i * P.ld_t;
                if (i == m) { for (int j = 0; j <= nt + 1; ++j) Ti[j] = 0.0; continue; }
                double mx = 0.0;
                for (int j = 1; j <= nt; ++j) mx = fmax(mx, fabs(Ti[j]));
                if (!(mx > ZERO_ROW_ATOL)) for (int j = 1; j <= nt; ++j) Ti[j] = 0.0;   // the theta LP's zero-row rule
                Ti[0] = i < nlam + nin ? -1.0 : 0.0;
                s.rowkind[i] = (pri && pri[i]) ? RK_PRI : RK_INEQ;
            }
        });
        if (r == LP_OPTIMAL && lane == 0) status[c] = (uint8_t)ST_FEASIBLE;
        wave_sync();
    }
}

// ------------------------------------------------------------------------------------------------------------
// k_verdict: status per candidate: INFEASIBLE / FEASIBLE / SINGULAR / LP_LIMIT / OPT_PENDING
// ------------------------------------------------------------------------------------------------------------
// list != nullptr: process only the candidates list[0..n) (the retry list of k_verdict2), work counter ctr->work_region
MPC_GLOBAL void MPC_LB(64) k_verdict(DevProblem P, const int32_t *__restrict__ cands, long long n, int k,
                                                uint8_t *__restrict__ status, LevelCounters *__restrict__ ctr,
                                                const int32_t *__restrict__ list, const int32_t *__restrict__ n_dev) {
    // n_dev != nullptr: the length of `list` is read from device memory (the level runs without host round trips)
    if (n_dev) { n = *n_dev; if ((long long)blockIdx.x >= n) return; }   // surplus block of a launch sized by a bound
    extern __shared__ __attribute__((aligned(16))) double smem[];
    Smem s = carve(P, smem);
    const int lane = lane_id(), nt = P.n_t, nx = P.n_x;
    unsigned long long pivots = 0, n_xlp = 0, n_fallback = 0;
    long long cyc_kkt = 0, cyc_theta = 0, cyc_x = 0;
    for (;;) {
        unsigned int c = 0;
        if (lane == 0) c = atomicAdd(list ? &ctr->work_retry : &ctr->work_verdict, 1u);
        c = (unsigned)__builtin_amdgcn_readfirstlane((int)c);
        if (c >= n) break;
        if (list) c = (unsigned)list[c];
        const int nin = load_active_set(P, cands + (size_t)c * k, k, s);
        int st = -1;
        const long long t0 = clock64();
        const int kk = kkt_solve(P, k, s);
        const long long t1 = clock64();
        cyc_kkt += t1 - t0;
        bool singular = false;
        if (kk == 1) st = ST_INFEASIBLE;
        else if (kk == 2) singular = true;
        else if (kk == 0) {
            // Two-stage LP over theta only (rows of mpqp_utils.py:111-121 evaluated at the KKT point x*(theta), l*(theta)):
            //   stage 1  {slack(theta) >= 0, A_t theta <= b_t} non-empty  =>  (x*(theta), theta) is a point of the (x,theta)
            //            polytope with the rows `as` active: the candidate is FEASIBLE without the large LP
            //   stage 2  the multiplier rows lambda(theta) >= 0, carried passively through stage 1, are switched on and
            //            phase 1 continues from the same dictionary: non-empty  <=>  optimal (check_optimality)
            Lp lp;
            lp.noscale = true;
            lp.T = s.T; lp.ld = P.ld_t; lp.colvar = s.colvar; lp.rowvar = s.rowvar; lp.rowkind = s.rowkind;
            const int nlam = k - P.n_eq;
            lp.n = nt; lp.m = nlam + nin + P.n_tc; lp.iters = 0;
            const int m = lp.m;
            auto load = [&](const int *pri, int lam_kind) {
                build_theta_rows(P, k, nin, s, s.T, P.ld_t, P.kkt_mode == 1);
                for (int i = lane; i <= m; i += 64) {
                    double *Ti = s.T + i * P.ld_t;
                    if (i == m) { for (int j = 0; j <= nt + 1; ++j) Ti[j] = 0.0; continue; }
                    double mx = 0.0;
                    for (int j = 1; j <= nt; ++j) mx = fmax(mx, fabs(Ti[j]));
                    if (!(mx > ZERO_ROW_ATOL)) for (int j = 1; j <= nt; ++j) Ti[j] = 0.0;
                    s.rowkind[i] = (pri && pri[i]) ? RK_PRI : (i < nlam ? lam_kind : RK_INEQ);
                }
            };
            // (Stage A of the engine may have to pivot a free theta_j into the basis on a PASSIVE multiplier row -- when the enforced rows do
            // not span every parameter direction: open parameter sets with few inactive rows; lp_engine.hpp, lp_run -- so only rows that
            // are still passive are switched on below.)
            const int r1 = lp_solve(lp, false, s.pri, [&](const int *pri) { load(pri, RK_PASSIVE); });
            if (r1 == LP_ITERLIMIT) st = ST_LP_LIMIT;
            else if (r1 == LP_OPTIMAL) {
                wave_sync();
                for (int i = lane; i < nlam; i += 64) if (s.rowkind[i] == RK_PASSIVE) s.rowkind[i] = RK_INEQ;
                wave_sync();
                lp.growth = 0.0;
                int r2 = lp_phase1(lp);
                if (r2 != LP_ITERLIMIT && lp.growth > GROWTH_SAFE)  // doubtful pivots: decide from a fresh solve over all rows
                    r2 = lp_solve(lp, false, s.pri, [&](const int *pri) { load(pri, RK_INEQ); });
                st = r2 == LP_OPTIMAL ? ST_OPT_PENDING : (r2 == LP_ITERLIMIT ? ST_LP_LIMIT : ST_FEASIBLE);
            }
            pivots += lp.iters;
        }
        const long long t2 = clock64();
        cyc_theta += t2 - t1;
        if (st < 0) {
            // feasibility of {A x - F theta <= b, A_t theta <= b_t, rows `as` active}   (mplp_program.py:439-444)
            Lp lp;
            n_xlp++;
            int r = -1;
            if (P.has_d0) {
                r = xtheta_from_vertex(P, k, s, lp);
                pivots += lp.iters;
                if (r < 0) n_fallback++;
            }
            if (r < 0) {
                lp.T = s.T; lp.ld = P.ld_x; lp.colvar = s.colvar; lp.rowvar = s.rowvar; lp.rowkind = s.rowkind;
                lp.n = nx + nt; lp.m = P.n_c + P.n_tc; lp.iters = 0;
                const int cols = 1 + nx + nt, mm = lp.m;
                r = lp_solve(lp, false, s.pri, [&](const int *pri) {
                    wave_sync();
                    for (int i = 0; i < mm; ++i)
                        for (int j = lane; j < cols; j += 64) s.T[i * P.ld_x + j] = P.base[i * cols + j];
                    for (int j = lane; j <= cols; j += 64) s.T[mm * P.ld_x + j] = 0.0;
                    for (int i = lane; i < mm; i += 64) s.rowkind[i] = (pri && pri[i]) ? RK_PRI : RK_INEQ;
                    wave_sync();
                    for (int i = lane; i < k; i += 64) s.rowkind[s.as[i]] = RK_EQ;
                });
                pivots += lp.iters;
            }
            if (r == LP_OPTIMAL) st = singular ? ST_SINGULAR : ST_FEASIBLE;
            else if (r == LP_ITERLIMIT) st = ST_LP_LIMIT;
            else st = ST_INFEASIBLE;
        }
        cyc_x += clock64() - t2;
        if (lane == 0) status[c] = (uint8_t)st;
    }
    if (lane == 0) { atomicAdd(&ctr->cycles[0], (unsigned long long)cyc_kkt); atomicAdd(&ctr->cycles[1], (unsigned long long)cyc_theta); atomicAdd(&ctr->cycles[2], (unsigned long long)cyc_x);
        atomicAdd(&ctr->pivots, pivots); atomicAdd(&ctr->xtheta_lps, n_xlp); atomicAdd(&ctr->xtheta_fallbacks, n_fallback); }
}

// ------------------------------------------------------------------------------------------------------------
// k_region: one optimal candidate per wavefront -> region record or OPT_NO_REGION
// ------------------------------------------------------------------------------------------------------------
// MODE 0 (RG_FULL)      one wavefront does everything for a candidate (throughput form)
// MODE 1 (RG_FACET)     one wavefront per (candidate, region row): rebuilds the rows and solves only that row's facet LP,
//                       result -> facet_flags[slot * rows_t + row] (1 kept, 0 redundant, 2 LP limit)   (latency form, step 1)
// MODE 2 (RG_ASSEMBLE)  one wavefront per candidate: rows + Chebyshev LP, facet decisions read from facet_flags (step 2)
constexpr int RG_FULL = 0, RG_FACET = 1, RG_ASSEMBLE = 2;
template <int MODE>
MPC_GLOBAL void MPC_LB(64) k_region(DevProblem P, const int32_t *__restrict__ cands, int k,
                                               const int32_t *__restrict__ opt_list, int n_opt,
                                               uint8_t *__restrict__ status, double *__restrict__ rec_d,
                                               int32_t *__restrict__ rec_i, long long sd, long long si,
                                               LevelCounters *__restrict__ ctr, uint8_t *__restrict__ facet_flags) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    Smem s = carve(P, smem);
    const int lane = lane_id(), nt = P.n_t, nr = nt + 1, nx = P.n_x, nc = P.n_c, ntc = P.n_tc, e = P.n_eq;
    unsigned long long pivots = 0;
    for (;;) {
        unsigned int w = 0;
        if (lane == 0) w = atomicAdd(&ctr->work_region, 1u);
        w = (unsigned)__builtin_amdgcn_readfirstlane((int)w);
        const int rows_all = P.n_c - P.n_eq + P.n_tc;
        if (w >= (unsigned)(MODE == RG_FACET ? n_opt * rows_all : n_opt)) break;
        const int my_row = MODE == RG_FACET ? (int)(w % (unsigned)rows_all) : -1;
        if (MODE == RG_FACET) w = w / (unsigned)rows_all;
        const int c = opt_list[w];
        const int nin = load_active_set(P, cands + (size_t)c * k, k, s);
        double *rd = rec_d + (size_t)w * sd;
        int32_t *ri = rec_i + (size_t)w * si;
        if (MODE != RG_FACET) {
            for (long long i = lane; i < sd; i += 64) rd[i] = 0.0;
            for (long long i = lane; i < si; i += 64) ri[i] = -1;
        }
        int st = ST_REGION;
        bool ill = false;
        const int kk = kkt_solve(P, k, s, &ill);
        if (kk != 0) st = kk == 1 ? ST_INFEASIBLE : ST_SINGULAR;  // cannot happen after k_verdict said optimal
        int nE = 0, n_om = 0, n_la = 0, n_re = 0;
        if (st == ST_REGION) {
            if (P.kkt_mode == 0) x_law_schur(P, k, s);
            // raw region rows -> T (scratch, row stride nr: column 0 = rhs), then the zero-row filter and unit L2
            // scaling (remove_numerically_zero_rows + scale_constraint, mpqp_utils.py:123-126) compact them into the
            // master copy E | f (row stride nr, column 0 = f); kept[] = index of the original row
            const int ldE = nr;
            const int nrows = build_theta_rows(P, k, nin, s, s.T, ldE, true);
            const int nlam = k - e;
            int nk = 0;
            for (int r0 = 0; r0 < nrows; r0 += 64) {
                const int r = r0 + lane;
                bool keep = false;
                double ss = 0.0;
                if (r < nrows)
                    for (int j = 0; j < nt; ++j) {
                        const double v = s.T[r * ldE + 1 + j];
                        if (!(fabs(v) <= ZERO_ROW_ATOL)) keep = true;
                        ss = fma(v, v, ss);
                    }
                const unsigned long long bal = __ballot(keep);
                if (keep) {
                    const int dstrow = nk + __popcll(bal & ((1ull << lane) - 1ull));
                    const double inv = 1.0 / sqrt(ss);
                    s.kept[dstrow] = r;
                    for (int j = 0; j <= nt; ++j) s.E[dstrow * ldE + j] = s.T[r * ldE + j] * inv;
                }
                nk += __popcll(bal);
            }
            wave_sync();
            if (nt == 1 && MODE == RG_FACET) {
                // nothing to do: the one-parameter variant has no LPs
            } else if (nt == 1) {
                // one parameter: interval arithmetic, no LPs (mpqp_utils.py:198-320)
                double mn = -INFINITY, mx = INFINITY;
                for (int r = lane; r < nk; r += 64) {
                    const double a = s.E[r * ldE + 1], v = s.E[r * ldE] / a;
                    if (a > 0) mx = fmin(mx, v); else mn = fmax(mn, v);
                }
                for (int off = 32; off > 0; off >>= 1) { mx = fmin(mx, __shfl_xor(mx, off)); mn = fmax(mn, __shfl_xor(mn, off)); }
                if (!(mn + 1e-8 <= mx)) st = ST_OPT_NO_REGION;
                else {
                    for (int r0 = 0; r0 < nk; r0 += 64) {
                        const int r = r0 + lane;
                        bool keep = false;
                        int cls = 0, val = 0, val2 = 0;
                        if (r < nk) {
                            const double v = s.E[r * ldE] / s.E[r * ldE + 1];
                            keep = (mn <= v && v <= mx);
                            const int o = s.kept[r];
                            if (o < nlam) { cls = 0; val = s.as[e + o]; }
                            else if (o < nlam + nin) { cls = 1; val = o - nlam; val2 = s.inact[o - nlam]; }
                            else { cls = 2; val = o - nlam - nin; }
                        }
                        const unsigned long long b0 = __ballot(keep && cls == 0), b1 = __ballot(keep && cls == 1), b2 = __ballot(keep && cls == 2);
                        const unsigned long long below = (1ull << lane) - 1ull;
                        if (keep && cls == 0) ri[5 + nc + ntc + n_la + __popcll(b0 & below)] = val;
                        if (keep && cls == 1) { const int p = n_re + __popcll(b1 & below); ri[5 + nc + ntc + nc + p] = val; ri[5 + nc + ntc + nc + nc + p] = val2; }
                        if (keep && cls == 2) ri[5 + nc + n_om + __popcll(b2 & below)] = val;
                        n_la += __popcll(b0); n_re += __popcll(b1); n_om += __popcll(b2);
                    }
                    double *Eo = rd + nx * nt + nx + nc * nt + nc, *fo = Eo + (nc + ntc) * nt;
                    if (lane == 0) { Eo[0] = 1.0; fo[0] = mx; Eo[1] = -1.0; fo[1] = -mn; }
                    nE = 2;
                }
            } else {
                // Chebyshev ball: min -r s.t. E theta + ||E_i|| r <= f, -r <= 0   (chebyshev_ball.py:41-60)
                Lp lp;
                lp.T = s.T; lp.ld = P.ld_t; lp.colvar = s.colvar; lp.rowvar = s.rowvar; lp.rowkind = s.rowkind;
                lp.n = nt + 1; lp.m = nk + 1; lp.iters = 0;
                int r = LP_OPTIMAL;
                if (MODE != RG_FACET) r = lp_solve(lp, true, s.pri, [&](const int *pri) {
                    wave_sync();
                    for (int i = lane; i <= nk + 1; i += 64) {
                        double *Ti = s.T + i * P.ld_t;
                        if (i < nk) {
                            double ss = 0.0;
                            Ti[0] = s.E[i * ldE];
                            for (int j = 0; j < nt; ++j) { const double v = s.E[i * ldE + 1 + j]; Ti[1 + j] = v; ss = fma(v, v, ss); }
                            Ti[1 + nt] = sqrt(ss);
                            Ti[2 + nt] = 0.0;
                        } else {
                            for (int j = 0; j <= nt + 2; ++j) Ti[j] = 0.0;
                            Ti[1 + nt] = -1.0;  // row nk: -r <= 0 ; row nk+1 (cost): min -r
                        }
                        s.rowkind[i] = (pri && i <= nk && pri[i]) ? RK_PRI : RK_INEQ;
                    }
                });
                pivots += lp.iters;
                double radius = 0.0;
                if (r == LP_OPTIMAL) {
                    int found = -1;
                    for (int i = lane; i < lp.m; i += 64)
                        if (s.rowkind[i] == RK_FREE && s.rowvar[i] == nt) found = i;
                    double dummy = found >= 0 ? 1.0 : 0.0;
                    reduce_max_first(dummy, found);
                    radius = found >= 0 ? s.T[found * P.ld_t] : 0.0;
                }
                if (MODE == RG_FACET) radius = 1.0;
                if (r == LP_ITERLIMIT) st = ST_LP_LIMIT;
                else if (r != LP_OPTIMAL || !(radius > FULL_DIM_RADIUS)) st = ST_OPT_NO_REGION;
                // one feasibility LP per kept row with that row as an equality (mpqp_utils.py:143-178)
                double *Eo = rd + nx * nt + nx + nc * nt + nc, *fo = Eo + (nc + ntc) * nt;
                for (int row = 0; row < nk && st == ST_REGION; ++row) {
                    if (MODE == RG_FACET && row != my_row) continue;
                    lp.n = nt; lp.m = nk; lp.iters = 0;
                    int rr;
                    if (MODE == RG_ASSEMBLE) {
                        const int fl = facet_flags[(size_t)w * rows_all + row];
                        rr = fl == 1 ? LP_OPTIMAL : (fl == 2 ? LP_ITERLIMIT : LP_INFEASIBLE);
                    } else rr = lp_solve(lp, false, s.pri, [&](const int *pri) {
                        wave_sync();
                        for (int i = lane; i <= nk; i += 64) {
                            double *Ti = s.T + i * P.ld_t;
                            if (i < nk) { for (int j = 0; j <= nt; ++j) Ti[j] = s.E[i * ldE + j]; Ti[nt + 1] = 0.0; }
                            else for (int j = 0; j <= nt + 1; ++j) Ti[j] = 0.0;
                            s.rowkind[i] = (i == row) ? RK_EQ : ((pri && i < nk && pri[i]) ? RK_PRI : RK_INEQ);
                        }
                    });
                    pivots += lp.iters;
                    if (MODE == RG_FACET) {
                        if (lane == 0) facet_flags[(size_t)w * rows_all + row] = rr == LP_OPTIMAL ? 1 : (rr == LP_ITERLIMIT ? 2 : 0);
                        break;
                    }
                    if (rr == LP_ITERLIMIT) { st = ST_LP_LIMIT; break; }
                    if (rr != LP_OPTIMAL) continue;
                    const int o = s.kept[row];
                    if (lane == 0) {
                        if (o < nlam) ri[5 + nc + ntc + n_la] = s.as[e + o];
                        else if (o < nlam + nin) { ri[5 + nc + ntc + nc + n_re] = o - nlam; ri[5 + nc + ntc + nc + nc + n_re] = s.inact[o - nlam]; }
                        else ri[5 + nc + n_om] = o - nlam - nin;
                    }
                    if (o < nlam) n_la++; else if (o < nlam + nin) n_re++; else n_om++;
                    // exact duplicates of an earlier kept row are not stored again (remove_duplicate_rows, mpqp_utils.py:191;
                    // the index sets above are unaffected, as in the reference); stored[0..nE) = master rows already written
                    bool dup = false;
                    for (int j = lane; j < nE; j += 64) {
                        const int pr = s.stored[j];
                        bool same = true;
                        for (int t = 0; t <= nt; ++t) same = same && (s.E[pr * ldE + t] == s.E[row * ldE + t]);
                        dup = dup || same;
                    }
                    if (__any(dup)) continue;
                    wave_sync();
                    if (lane == 0) { fo[nE] = s.E[row * ldE]; s.stored[nE] = row; }
                    for (int j = lane; j < nt; j += 64) Eo[nE * nt + j] = s.E[row * ldE + 1 + j];
                    nE++;
                    wave_sync();
                }
            }
        }
        if (MODE == RG_FACET) continue;
        // "optimal but lower-dimensional" from an ill-conditioned Schur system is not trusted: the candidate is expanded
        // like a feasible, non-optimal one instead of being pruned together with its supersets
        if (ill && st == ST_OPT_NO_REGION) st = ST_FEASIBLE;
        if (st == ST_REGION) {
            // x-law, multipliers, header
            for (int idx = lane; idx < nx * nt; idx += 64) rd[idx] = s.X[(idx / nt) * nr + 1 + idx % nt];
            for (int i = lane; i < nx; i += 64) rd[nx * nt + i] = s.X[i * nr];
            double *Al = rd + nx * nt + nx, *bl = Al + nc * nt;
            for (int idx = lane; idx < k * nt; idx += 64) Al[idx] = s.L[(idx / nt) * nr + 1 + idx % nt];
            for (int i = lane; i < k; i += 64) bl[i] = s.L[i * nr];
            for (int i = lane; i < k; i += 64) ri[5 + i] = s.as[i];
            if (lane == 0) { ri[0] = k; ri[1] = nE; ri[2] = n_om; ri[3] = n_la; ri[4] = n_re; }
        }
        if (lane == 0) status[c] = (uint8_t)st;
    }
    if (lane == 0) atomicAdd(&ctr->pivots, pivots);
}

// ------------------------------------------------------------------------------------------------------------
// small utility kernels: flags, scan, compaction, children, pruned masks
// ------------------------------------------------------------------------------------------------------------
MPC_GLOBAL void k_histogram(const uint8_t *__restrict__ status, long long n, LevelCounters *ctr) {
    __shared__ unsigned int h[8];
    if (threadIdx.x < 8) h[threadIdx.x] = 0;
    __syncthreads();
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        atomicAdd(&h[status[i] & 7], 1u);
    __syncthreads();
    if (threadIdx.x < 8 && h[threadIdx.x]) atomicAdd(&ctr->status[threadIdx.x], (unsigned long long)h[threadIdx.x]);
}

// Exclusive scan of int32 values in three launches: per-block sums (1024 items per block), a single-block scan of
// the block sums, and the per-block scan with its offset.  total = sum of all values.
constexpr int SCAN_BLOCK = 1024;
__device__ __forceinline__ int block_exclusive_scan_1024(int v, int *total_out) {
    __shared__ int wsum[16];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(incl, off); if (lane >= off) incl += o; }
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    if (w == 0) {
        int t = lane < 16 ? wsum[lane] : 0, ti = t;
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) { const int o = __shfl_up(ti, off); if (lane >= off) ti += o; }
        if (lane < 16) wsum[lane] = ti - t;
        if (lane == 15 && total_out) *total_out = ti;
    }
    __syncthreads();
    return incl - v + wsum[w];
}
// The same scan of 1024 items by 1024 / IT threads, IT consecutive items per thread (IT = 4: four-wavefront workgroups, which find room
// on a compute unit that a persistent kernel's wavefronts occupy; a 16-wavefront workgroup waits there until a whole unit drains).
template <int IT>
__device__ __forceinline__ void block_exclusive_scan_it(const int (&v)[IT], int (&ex)[IT], int *total_out) {
    constexpr int NW = 16 / IT;
    __shared__ int wsum_it[NW];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int s = 0;
#pragma unroll
    for (int k = 0; k < IT; ++k) s += v[k];
    int incl = s;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(incl, off); if (lane >= off) incl += o; }
    if (lane == 63) wsum_it[w] = incl;
    __syncthreads();
    int base = 0;
#pragma unroll
    for (int k = 0; k < NW; ++k) base += k < w ? wsum_it[k] : 0;
    if (total_out && threadIdx.x == 64 * NW - 1) *total_out = base + incl;
    int run = base + incl - s;
#pragma unroll
    for (int k = 0; k < IT; ++k) { ex[k] = run; run += v[k]; }
    __syncthreads();
}
template <int IT>
MPC_GLOBAL void MPC_LB(1024 / IT) k_scan_block_sums(const int32_t *__restrict__ in, long long n, int32_t *__restrict__ sums) {
    __shared__ int tot;
    const long long i0 = blockIdx.x * (long long)SCAN_BLOCK + (long long)threadIdx.x * IT;
    int v[IT], ex[IT];
#pragma unroll
    for (int k = 0; k < IT; ++k) v[k] = i0 + k < n ? in[i0 + k] : 0;
    block_exclusive_scan_it<IT>(v, ex, &tot);
    if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}
// one block: exclusive scan of up to nb block sums in place (chunked), total -> *total
template <int IT>
MPC_GLOBAL void MPC_LB(1024 / IT) k_scan_sums(int32_t *__restrict__ sums, int nb, int32_t *__restrict__ total) {
    __shared__ int tot;
    int carry = 0;
    for (int base = 0; base < nb; base += SCAN_BLOCK) {
        const int i0 = base + (int)threadIdx.x * IT;
        int v[IT], ex[IT];
#pragma unroll
        for (int k = 0; k < IT; ++k) v[k] = i0 + k < nb ? sums[i0 + k] : 0;
        block_exclusive_scan_it<IT>(v, ex, &tot);
#pragma unroll
        for (int k = 0; k < IT; ++k) if (i0 + k < nb) sums[i0 + k] = ex[k] + carry;
        carry += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry;
}
template <int IT>
MPC_GLOBAL void MPC_LB(1024 / IT) k_scan_apply(const int32_t *__restrict__ in, int32_t *__restrict__ out, long long n,
                                                    const int32_t *__restrict__ sums) {
    const long long i0 = blockIdx.x * (long long)SCAN_BLOCK + (long long)threadIdx.x * IT;
    int v[IT], ex[IT];
#pragma unroll
    for (int k = 0; k < IT; ++k) v[k] = i0 + k < n ? in[i0 + k] : 0;
    block_exclusive_scan_it<IT>(v, ex, nullptr);
    const int add = sums[blockIdx.x];
#pragma unroll
    for (int k = 0; k < IT; ++k) if (i0 + k < n) out[i0 + k] = ex[k] + add;
}

// ---- deterministic multi-class partition by status -----------------------------------------------------------------------
// spec: 16 nibbles, nibble s = class (0..3) of status s, 15 = not listed.  Three launches produce up to four index lists in
// frontier order (lists + c*n) and their lengths, instead of one flag/scan/scatter round per class.
constexpr int PART_CLASSES = 4;
__device__ __forceinline__ int part_class(unsigned long long spec, int st) { return (int)((spec >> (4 * (st & 15))) & 15ull); }
// 1024 statuses per workgroup of 1024 / IT threads: sub-chunk j of the workgroup is the 1024 / IT statuses from j * (1024 / IT)
template <int IT>
MPC_GLOBAL void MPC_LB(1024 / IT) k_part_count(const uint8_t *__restrict__ status, long long n, unsigned long long spec,
                                                     int32_t *__restrict__ blockcounts, int nb) {
    constexpr int BT = 1024 / IT, NW = BT / 64;
    __shared__ int wc[16][PART_CLASSES];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int j = 0; j < IT; ++j) {
        const long long i = blockIdx.x * 1024LL + j * BT + threadIdx.x;
        const int cls = i < n ? part_class(spec, status[i]) : 15;
#pragma unroll
        for (int c = 0; c < PART_CLASSES; ++c) {
            const unsigned long long m = __ballot(cls == c);
            if (lane == 0) wc[j * NW + wave][c] = __popcll(m);
        }
    }
    __syncthreads();
    if (threadIdx.x < PART_CLASSES) {
        int tot = 0;
        for (int w = 0; w < 16; ++w) tot += wc[w][threadIdx.x];
        blockcounts[threadIdx.x * nb + blockIdx.x] = tot;
    }
}
// grid = PART_CLASSES blocks: exclusive scan of each class's block counts in place, totals[c] = list length
template <int IT>
MPC_GLOBAL void MPC_LB(1024 / IT) k_part_sums(int32_t *__restrict__ blockcounts, int nb, int32_t *__restrict__ totals) {
    __shared__ int tot;
    int32_t *sums = blockcounts + (size_t)blockIdx.x * nb;
    int carry = 0;
    for (int base = 0; base < nb; base += SCAN_BLOCK) {
        const int i0 = base + (int)threadIdx.x * IT;
        int v[IT], ex[IT];
#pragma unroll
        for (int k = 0; k < IT; ++k) v[k] = i0 + k < nb ? sums[i0 + k] : 0;
        block_exclusive_scan_it<IT>(v, ex, &tot);
#pragma unroll
        for (int k = 0; k < IT; ++k) if (i0 + k < nb) sums[i0 + k] = ex[k] + carry;
        carry += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) totals[blockIdx.x] = carry;
}
template <int IT>
MPC_GLOBAL void MPC_LB(1024 / IT) k_part_scatter(const uint8_t *__restrict__ status, long long n, unsigned long long spec,
                                                       const int32_t *__restrict__ blockbase, int nb, int32_t *__restrict__ lists) {
    constexpr int BT = 1024 / IT, NW = BT / 64;
    __shared__ int wc[16][PART_CLASSES];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int cls[IT], within[IT];
#pragma unroll
    for (int j = 0; j < IT; ++j) {
        const long long i = blockIdx.x * 1024LL + j * BT + threadIdx.x;
        cls[j] = i < n ? part_class(spec, status[i]) : 15;
        within[j] = 0;
#pragma unroll
        for (int c = 0; c < PART_CLASSES; ++c) {
            const unsigned long long m = __ballot(cls[j] == c);
            if (lane == 0) wc[j * NW + wave][c] = __popcll(m);
            if (cls[j] == c) within[j] = __popcll(m & ((1ull << lane) - 1ull));
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < IT; ++j) {
        if (cls[j] < PART_CLASSES) {
            const long long i = blockIdx.x * 1024LL + j * BT + threadIdx.x;
            int before = 0;
            for (int w = 0; w < j * NW + wave; ++w) before += wc[w][cls[j]];
            lists[(size_t)cls[j] * n + blockbase[cls[j] * nb + blockIdx.x] + before + within[j]] = (int32_t)i;
        }
    }
}

MPC_GLOBAL void k_flag_status(const uint8_t *__restrict__ status, long long n, int lo, int hi, int32_t *__restrict__ flag) {
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i < n) flag[i] = status[i] >= lo && status[i] <= hi;
}
MPC_GLOBAL void k_scatter_index(const int32_t *__restrict__ flag, const int32_t *__restrict__ pos, long long n,
                                int32_t *__restrict__ list) {
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i < n && flag[i]) list[pos[i]] = (int32_t)i;
}

// clears up to four buffers in ONE launch (a level's counters, list lengths and flags: a hipMemsetAsync each costs the small levels
// a ~4 us dispatch of its own).  Sizes in bytes, multiples of 8 with 8-byte-aligned pointers take the word path.
struct ZeroBufs { void *p[4]; unsigned long long bytes[4]; };
MPC_GLOBAL void MPC_LB(256) k_zero_bufs(ZeroBufs z) {
    for (int j = 0; j < 4; ++j) {
        const unsigned long long bytes = z.bytes[j];
        if (!z.p[j] || bytes == 0) continue;
        if ((reinterpret_cast<unsigned long long>(z.p[j]) & 7ull) == 0 && (bytes & 7ull) == 0) {
            unsigned long long *p = reinterpret_cast<unsigned long long *>(z.p[j]);
            for (unsigned long long i = blockIdx.x * 256ull + threadIdx.x; i < bytes / 8; i += (unsigned long long)gridDim.x * 256ull) p[i] = 0ull;
        } else {
            unsigned char *p = reinterpret_cast<unsigned char *>(z.p[j]);
            for (unsigned long long i = blockIdx.x * 256ull + threadIdx.x; i < bytes; i += (unsigned long long)gridDim.x * 256ull) p[i] = 0;
        }
    }
}

// Everything a region launch needs cleared / copied / initialised, as ONE launch (each hipMemsetAsync / hipMemcpyAsync costs the host
// ~16 us and the device a dependent 5-10 us step: five of them sat between the theta stage and the region kernel of every large level):
// up to four buffers zeroed, one int32 list copied (the optimal candidates, out of the partition buffer the next partition reuses),
// the spare record slots marked empty (what k_init_slots does).
struct RegionPrep { ZeroBufs z; const int32_t *copy_src; int32_t *copy_dst; long long copy_n; int32_t *head_i; int fi, first, extra; };
MPC_GLOBAL void MPC_LB(256) k_region_prep(RegionPrep a) {
    for (int j = 0; j < 4; ++j) {
        const unsigned long long bytes = a.z.bytes[j];
        if (!a.z.p[j] || bytes == 0) continue;
        if ((reinterpret_cast<unsigned long long>(a.z.p[j]) & 7ull) == 0 && (bytes & 7ull) == 0) {
            unsigned long long *p = reinterpret_cast<unsigned long long *>(a.z.p[j]);
            for (unsigned long long i = blockIdx.x * 256ull + threadIdx.x; i < bytes / 8; i += (unsigned long long)gridDim.x * 256ull) p[i] = 0ull;
        } else {
            unsigned char *p = reinterpret_cast<unsigned char *>(a.z.p[j]);
            for (unsigned long long i = blockIdx.x * 256ull + threadIdx.x; i < bytes; i += (unsigned long long)gridDim.x * 256ull) p[i] = 0;
        }
    }
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < a.copy_n; i += (long long)gridDim.x * 256ll) a.copy_dst[i] = a.copy_src[i];
    for (int j = blockIdx.x * 256 + threadIdx.x; j < a.extra; j += gridDim.x * 256) { int32_t *hi = a.head_i + (size_t)(a.first + j) * a.fi; hi[0] = 0; hi[1] = -1; }
}

// The slot arrays of a level that did not stream (head_d, head_i, rows) from their device buffers into page-locked host blocks, by
// stores of this kernel instead of three copy commands (each ~16 us of host time) and a synchronisation: the workgroup that finishes
// last raises `flag` (host memory, system-scope release), which is all the consumer waits for -- the level loop goes on at once.
struct FetchCopy { const void *src[3]; void *dst[3]; unsigned long long bytes[3]; unsigned int *counter; int32_t *flag; };
MPC_GLOBAL void MPC_LB(256) k_fetch_slots(FetchCopy a) {
    for (int j = 0; j < 3; ++j) {
        const unsigned long long words = a.bytes[j] / 4;      // every array is a whole number of 4-byte words
        const unsigned int *src = static_cast<const unsigned int *>(a.src[j]);
        unsigned int *dst = static_cast<unsigned int *>(a.dst[j]);
        for (unsigned long long i = blockIdx.x * 256ull + threadIdx.x; i < words; i += (unsigned long long)gridDim.x * 256ull) dst[i] = src[i];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    __syncthreads();
    if (threadIdx.x == 0) {
        if (atomicAdd(a.counter, 1u) + 1u == gridDim.x) {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "");
            atomicExch(a.counter, 0u);
            __hip_atomic_store(a.flag, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// The records of MANY handles copied to page-locked host memory by ONE launch (round 5: the shared launches of many programs paid
// three copy commands per member and level -- ~16 us of host time each on this runtime, 1.8 ms per level for 128 members).
// blockIdx.y = table entry (one array of one member), blockIdx.x strides over its 4-byte words.
struct FetchEntry { const void *src; void *dst; unsigned long long bytes; };
MPC_GLOBAL void MPC_LB(256) k_fetch_many(const FetchEntry *__restrict__ tab) {
    const FetchEntry e = tab[blockIdx.y];
    const unsigned long long words = e.bytes / 4;
    const unsigned int *src = static_cast<const unsigned int *>(e.src);
    unsigned int *dst = static_cast<unsigned int *>(e.dst);
    for (unsigned long long i = blockIdx.x * 256ull + threadIdx.x; i < words; i += (unsigned long long)gridDim.x * 256ull) dst[i] = src[i];
}

// copies a few words of device memory into pinned host memory (a read-back without a copy command: on this runtime a
// hipMemcpyAsync costs the host ~16 us, a launch ~3 us)
MPC_GLOBAL void k_publish_words(const unsigned int *__restrict__ src, unsigned int *__restrict__ dst, int n) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) dst[i] = src[i];
}
MPC_GLOBAL void MPC_LB(256) k_fill_i32(int32_t *__restrict__ dst, long long n, int32_t v) {
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256ll) dst[i] = v;
}
MPC_GLOBAL void k_set_u32(unsigned int *__restrict__ dst, unsigned int v) { if (threadIdx.x == 0 && blockIdx.x == 0) __hip_atomic_store(dst, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }
// two sources, one after the other in dst (a level's counters and its list lengths: one launch)
MPC_GLOBAL void k_publish_words2(const unsigned int *__restrict__ src1, int n1, const unsigned int *__restrict__ src2, int n2, unsigned int *__restrict__ dst) {
    for (int i = threadIdx.x; i < n1 + n2; i += blockDim.x) dst[i] = i < n1 ? src1[i] : src2[i - n1];
}

// ---- small levels: the same bookkeeping in ONE single-block launch each -------------------------------------------------------
// A level of a few thousand candidates is launch-latency bound (about 45 dependent device operations of 5-10 us each);
// below SMALL_LEVEL_N candidates compaction, partition and scan run as one block of 1024 threads that walks the array
// in chunks of 1024 (order preserved, so the lists are identical to the multi-block versions).
constexpr int SMALL_LEVEL_N = 16384;
MPC_GLOBAL void MPC_LB(1024) k_compact_small(const uint8_t *__restrict__ status, int n, int lo, int hi,
                                                        int32_t *__restrict__ list, int32_t *__restrict__ total) {
    __shared__ int wc[16];
    __shared__ int base;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (threadIdx.x == 0) base = 0;
    __syncthreads();
    for (int start = 0; start < n; start += 1024) {
        const int i = start + (int)threadIdx.x;
        const bool f = i < n && status[i] >= lo && status[i] <= hi;
        const unsigned long long m = __ballot(f);
        if (lane == 0) wc[wave] = __popcll(m);
        __syncthreads();
        int before = base;
        for (int w = 0; w < wave; ++w) before += wc[w];
        if (f) list[before + __popcll(m & ((1ull << lane) - 1ull))] = i;
        __syncthreads();
        if (threadIdx.x == 0) { int t = 0; for (int w = 0; w < 16; ++w) t += wc[w]; base += t; }
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = base;
}
MPC_GLOBAL void MPC_LB(1024) k_partition_small(const uint8_t *__restrict__ status, int n, unsigned long long spec,
                                                          int32_t *__restrict__ lists, long long stride, int32_t *__restrict__ totals) {
    __shared__ int wc[16][PART_CLASSES];
    __shared__ int base[PART_CLASSES];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (threadIdx.x < PART_CLASSES) base[threadIdx.x] = 0;
    __syncthreads();
    for (int start = 0; start < n; start += 1024) {
        const int i = start + (int)threadIdx.x;
        const int cls = i < n ? part_class(spec, status[i]) : 15;
        int within = 0;
#pragma unroll
        for (int c = 0; c < PART_CLASSES; ++c) {
            const unsigned long long m = __ballot(cls == c);
            if (lane == 0) wc[wave][c] = __popcll(m);
            if (cls == c) within = __popcll(m & ((1ull << lane) - 1ull));
        }
        __syncthreads();
        if (cls < PART_CLASSES) {
            int before = base[cls];
            for (int w = 0; w < wave; ++w) before += wc[w][cls];
            lists[(size_t)cls * stride + before + within] = i;
        }
        __syncthreads();
        if (threadIdx.x < PART_CLASSES) { int t = 0; for (int w = 0; w < 16; ++w) t += wc[w][threadIdx.x]; base[threadIdx.x] += t; }
        __syncthreads();
    }
    if (threadIdx.x < PART_CLASSES) totals[threadIdx.x] = base[threadIdx.x];
}
MPC_GLOBAL void MPC_LB(1024) k_scan_small(const int32_t *__restrict__ in, int32_t *__restrict__ out, int n,
                                                     int32_t *__restrict__ total) {
    __shared__ int tot;
    int carry = 0;
    for (int start = 0; start < n; start += SCAN_BLOCK) {
        const int i = start + (int)threadIdx.x;
        const int ex = block_exclusive_scan_1024(i < n ? in[i] : 0, &tot);
        __syncthreads();
        if (i < n) out[i] = ex + carry;
        carry += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry;
}

// keep_lowdim: the serial driver's rule (mp_solvers/mpqp_combinatorial.py:44-61, also mpqp_parallel_combinatorial_exp.py:38-52) --
// every feasible set is expanded, also one that is optimal with a lower-dimensional region; the parallel driver prunes such
// a set together with its supersets (mpqp_parrallel_combinatorial.py:57-59)
__device__ __forceinline__ bool expands(int st, int keep_lowdim = 0) {
    return st == ST_FEASIBLE || st == ST_REGION || st == ST_SINGULAR || st == ST_LP_LIMIT || (keep_lowdim && st == ST_OPT_NO_REGION);
}

// Active sets as bit masks of MW 64-bit words (MW = 2: n_c <= 128, MW = 4: n_c <= 256; mpc_mask_words).
template <int MW>
__device__ __forceinline__ void set_mask(const int32_t *as, int k, unsigned long long (&p)[MW]) {
#pragma unroll
    for (int w = 0; w < MW; ++w) p[w] = 0;
    for (int i = 0; i < k; ++i) {
        const int v = as[i];
#pragma unroll
        for (int w = 0; w < MW; ++w) if ((v >> 6) == w) p[w] |= 1ull << (v & 63);
    }
}

// children of one parent per wavefront: bit i of childmask = [as + {i}] survives CombinationTester.check and the
// mpLP filter (driver lines 49-51); count[c] = popcount.
// The superset test is organised by pruned set, not by child: for a pruned mask p, d = p & ~parent is empty (every
// child contains p: all culled) or a single bit i > last(parent) (exactly child i contains p) or irrelevant.  Lanes
// stride over the pruned list, each accumulating the bit set of culled children, then the wave ORs them together.
template <int MW>
MPC_GLOBAL void MPC_LB(64) k_children_count(DevProblem P, const int32_t *__restrict__ cands, long long n, int k,
                                                       const uint8_t *__restrict__ status,
                                                       const unsigned long long *__restrict__ pruned, long long n_pruned,
                                                       unsigned long long *__restrict__ childmask, int32_t *__restrict__ count, int keep_lowdim) {
    const long long c = blockIdx.x;
    const int lane = lane_id();
    if (c >= n) return;
    const int st = status[c];
    unsigned long long m[MW];
#pragma unroll
    for (int w = 0; w < MW; ++w) m[w] = 0;
    if (expands(st, keep_lowdim)) {
        const int32_t *as = cands + (size_t)c * k;
        unsigned long long p[MW], a[MW], kill[MW];
        set_mask<MW>(as, k, p);
        const int start = k > 0 ? as[k - 1] + 1 : 0;
        // candidates for children: indices start .. n_c-1 (mpLP filter for non-optimal parents)
        int stop = P.n_c;
        if (!P.is_qp && st == ST_FEASIBLE) stop = min(stop, (k + 1) + P.n_c - P.n_x);
#pragma unroll
        for (int w = 0; w < MW; ++w) {   // allowed children: bits start .. stop-1
            const int lo = max(start - 64 * w, 0), hi = min(stop - 64 * w, 64);
            a[w] = hi > lo ? ((hi >= 64 ? ~0ull : ((1ull << hi) - 1ull)) & ~((1ull << lo) - 1ull)) : 0ull;
            kill[w] = 0;
        }
        for (long long j = lane; j < n_pruned; j += 64) {
            unsigned long long d[MW];
            int bits = 0;
#pragma unroll
            for (int w = 0; w < MW; ++w) { d[w] = pruned[MW * j + w] & ~p[w]; bits += __popcll(d[w]); }
            if (bits == 0) {
#pragma unroll
                for (int w = 0; w < MW; ++w) kill[w] = ~0ull;
            } else if (bits == 1) {
#pragma unroll
                for (int w = 0; w < MW; ++w) kill[w] |= d[w];
            }
        }
#pragma unroll
        for (int w = 0; w < MW; ++w) {
            for (int off = 32; off > 0; off >>= 1) kill[w] |= __shfl_xor(kill[w], off);
            m[w] = a[w] & ~kill[w];
        }
    }
    if (lane == 0) {
        int cnt = 0;
#pragma unroll
        for (int w = 0; w < MW; ++w) { childmask[MW * c + w] = m[w]; cnt += __popcll(m[w]); }
        count[c] = cnt;
    }
}

// ---- round 6: the pruned list bucketed by each set's SMALLEST non-equality member ---------------------------------------------------
// A pruned set p matters to a parent P only if p \ P is a single index i beyond P's last member -- then every other member of p, its
// smallest in particular, is a member of P (sets of one non-equality member aside: they are ORed into `singles` once).  A parent therefore
// scans the buckets of ITS OWN members only -- k of n_c buckets -- instead of the whole list: the children stage of a deep tree is
// O(parents x pruned sets) (generate_mpqp_data(10,2,30): 640 k parents x 30 k sets, half of the solve).
// bucket_head: [0..256) counts, then offsets [256..513), then cursors [513..769), then `singles` as MW 64-bit words at [776..)
constexpr int PB_OFF = 256, PB_CUR = 513, PB_SINGLES = 776, PB_WORDS = PB_SINGLES + 2 * 4;
template <int MW>
__device__ __forceinline__ int pruned_min_member(const unsigned long long *p, int ne, int *n_members) {
    int first = -1, cnt = 0;
#pragma unroll
    for (int w = 0; w < MW; ++w) {
        unsigned long long v = p[w];
        const int lo = ne - 64 * w;                      // equality rows are members of every set: left out
        if (lo >= 64) v = 0ull; else if (lo > 0) v &= ~((1ull << lo) - 1ull);
        cnt += __popcll(v);
        if (first < 0 && v) first = 64 * w + __ffsll((long long)v) - 1;
    }
    *n_members = cnt;
    return first;
}
template <int MW>
MPC_GLOBAL void MPC_LB(256) k_pruned_bucket_count(const unsigned long long *__restrict__ pruned, long long n_pruned, int ne, int32_t *__restrict__ head) {
    const long long j = (long long)blockIdx.x * 256 + threadIdx.x;
    if (j >= n_pruned) return;
    int cnt = 0;
    const int b = pruned_min_member<MW>(pruned + (size_t)MW * j, ne, &cnt);
    if (cnt <= 1) {
        // one non-equality member: ORed into `singles`; none (the base set itself is pruned): every index is "pruned"
        unsigned long long *singles = reinterpret_cast<unsigned long long *>(head + PB_SINGLES);
#pragma unroll
        for (int w = 0; w < MW; ++w) { const unsigned long long v = cnt == 0 ? ~0ull : pruned[(size_t)MW * j + w]; if (v) atomicOr(&singles[w], v); }
    } else if (b >= 0) atomicAdd(&head[b], 1);
}
MPC_GLOBAL void MPC_LB(256) k_pruned_bucket_scan(int32_t *__restrict__ head) {
    __shared__ int sh[256];
    const int t = threadIdx.x;
    sh[t] = head[t];
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) { const int v = t >= off ? sh[t - off] : 0; __syncthreads(); sh[t] += v; __syncthreads(); }
    head[PB_OFF + t + 1] = sh[t];
    if (t == 0) head[PB_OFF] = 0;
    head[PB_CUR + t] = 0;
}
template <int MW>
MPC_GLOBAL void MPC_LB(256) k_pruned_bucket_scatter(const unsigned long long *__restrict__ pruned, long long n_pruned, int ne, int32_t *__restrict__ head,
                                                            unsigned long long *__restrict__ out) {
    const long long j = (long long)blockIdx.x * 256 + threadIdx.x;
    if (j >= n_pruned) return;
    int cnt = 0;
    const int b = pruned_min_member<MW>(pruned + (size_t)MW * j, ne, &cnt);
    if (cnt <= 1 || b < 0) return;
    const int pos = head[PB_OFF + b] + atomicAdd(&head[PB_CUR + b], 1);
#pragma unroll
    for (int w = 0; w < MW; ++w) out[(size_t)MW * pos + w] = pruned[(size_t)MW * j + w];
}
// k_children_count over the bucketed list: the same test on every pruned set whose smallest non-equality member is a member of the parent
template <int MW>
MPC_GLOBAL void MPC_LB(64) k_children_count_b(DevProblem P, const int32_t *__restrict__ cands, long long n, int k,
                                                         const uint8_t *__restrict__ status,
                                                         const unsigned long long *__restrict__ bucketed, const int32_t *__restrict__ head,
                                                         unsigned long long *__restrict__ childmask, int32_t *__restrict__ count, int keep_lowdim) {
    const long long c = blockIdx.x;
    const int lane = lane_id();
    if (c >= n) return;
    const int st = status[c];
    unsigned long long m[MW];
#pragma unroll
    for (int w = 0; w < MW; ++w) m[w] = 0;
    if (expands(st, keep_lowdim)) {
        const int32_t *as = cands + (size_t)c * k;
        unsigned long long p[MW], a[MW], kill[MW];
        set_mask<MW>(as, k, p);
        const int start = k > 0 ? as[k - 1] + 1 : 0;
        int stop = P.n_c;
        if (!P.is_qp && st == ST_FEASIBLE) stop = min(stop, (k + 1) + P.n_c - P.n_x);
        const unsigned long long *singles = reinterpret_cast<const unsigned long long *>(head + PB_SINGLES);
#pragma unroll
        for (int w = 0; w < MW; ++w) {   // allowed children: bits start .. stop-1
            const int lo = max(start - 64 * w, 0), hi = min(stop - 64 * w, 64);
            a[w] = hi > lo ? ((hi >= 64 ? ~0ull : ((1ull << hi) - 1ull)) & ~((1ull << lo) - 1ull)) : 0ull;
            kill[w] = singles[w] & ~p[w];          // a pruned set of ONE non-equality member i: exactly child i contains it
            // (... and if the parent itself holds such a member every child does -- cannot be for a consistent list, as in k_children_count)
            const int lo_e = P.n_eq - 64 * w;
            const unsigned long long emask = lo_e >= 64 ? ~0ull : (lo_e > 0 ? ((1ull << lo_e) - 1ull) : 0ull);
            if (singles[w] & p[w] & ~emask) kill[w] = ~0ull;
        }
        for (int mi = P.n_eq; mi < k; ++mi) {
            const int b = as[mi];
            const int j0 = head[PB_OFF + b], j1 = head[PB_OFF + b + 1];
            for (int j = j0 + lane; j < j1; j += 64) {
                unsigned long long d[MW];
                int bits = 0;
#pragma unroll
                for (int w = 0; w < MW; ++w) { d[w] = bucketed[(size_t)MW * j + w] & ~p[w]; bits += __popcll(d[w]); }
                if (bits == 0) {
#pragma unroll
                    for (int w = 0; w < MW; ++w) kill[w] = ~0ull;
                } else if (bits == 1) {
#pragma unroll
                    for (int w = 0; w < MW; ++w) kill[w] |= d[w];
                }
            }
        }
#pragma unroll
        for (int w = 0; w < MW; ++w) {
            for (int off = 32; off > 0; off >>= 1) kill[w] |= __shfl_xor(kill[w], off);
            m[w] = a[w] & ~kill[w];
        }
    }
    if (lane == 0) {
        int cnt = 0;
#pragma unroll
        for (int w = 0; w < MW; ++w) { childmask[MW * c + w] = m[w]; cnt += __popcll(m[w]); }
        count[c] = cnt;
    }
}

// stored / parent_slot (both optional): child -> index of its parent when the parent left its (x,theta) dictionary in the
// dictionary cache (k_x2), else -1
MPC_GLOBAL void MPC_LB(64) k_children_write(const int32_t *__restrict__ cands, long long n, int k, int mw,
                                                       const unsigned long long *__restrict__ childmask,
                                                       const int32_t *__restrict__ offset, int32_t *__restrict__ out,
                                                       const uint8_t *__restrict__ stored, int32_t *__restrict__ parent_slot) {
    const long long c = blockIdx.x;
    const int lane = lane_id();
    if (c >= n) return;
    const int32_t *as = cands + (size_t)c * k;
    int base = offset[c];
    const int ps = (stored && stored[c]) ? (int)c : -1;
    for (int half = 0; half < mw; ++half) {
        const unsigned long long mk = childmask[(size_t)mw * c + half];
        if (!mk) continue;
        const bool mine = (mk >> lane) & 1ull;
        const int pos = base + __popcll(mk & ((1ull << lane) - 1ull));
        if (mine) {
            int32_t *o = out + (size_t)pos * (k + 1);
            for (int i = 0; i < k; ++i) o[i] = as[i];
            o[k] = half * 64 + lane;
            if (parent_slot) parent_slot[pos] = ps;
        }
        base += __popcll(mk);
    }
}

// masks of the candidates pruned by this level (INFEASIBLE, OPT_NO_REGION), appended at pruned[n_pruned_old + ...]
template <int MW>
MPC_GLOBAL void k_pruned_append(const int32_t *__restrict__ cands, long long n, int k, const uint8_t *__restrict__ status,
                                unsigned long long *__restrict__ out, LevelCounters *ctr, int keep_lowdim) {
    // (round 6) ONE atomic per 4,096 candidates instead of one per pruned candidate.  Config 3's last level prunes 16 k of its 994 k candidates --
    // about one per wavefront, so a wave-level aggregation changes nothing -- and as many additions to one address, each waiting for its
    // answer, took 85 us (config 4: 6 us).  A workgroup counts its 16 x 256 candidates first (each thread keeps the verdicts of its sixteen as
    // a bit mask), takes the workgroup's range with one atomic and writes.  The list is a set: its order was the atomics' before.  The launch
    // keeps its grid of one workgroup per 256 candidates: the surplus workgroups leave at once.
    constexpr int PER = 16;
    __shared__ unsigned int wsum[4], base_s;
    const long long c0 = (long long)blockIdx.x * 256 * PER;
    if (c0 >= n) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned int bits = 0;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const long long c = c0 + (long long)j * 256 + threadIdx.x;
        const int st = c < n ? status[c] : 0;
        if (c < n && (st == ST_INFEASIBLE || (st == ST_OPT_NO_REGION && !keep_lowdim))) bits |= 1u << j;
    }
    const unsigned int cnt = (unsigned int)__popc(bits);
    unsigned int incl = cnt;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const unsigned int o = (unsigned int)__shfl_up((int)incl, off); if (lane >= off) incl += o; }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        base_s = total ? atomicAdd(&ctr->n_pruned_new, total) : 0u;
    }
    __syncthreads();
    unsigned int pos = base_s + incl - cnt;
    for (int w = 0; w < wave; ++w) pos += wsum[w];
    while (bits) {
        const int j = __builtin_ctz(bits);
        bits &= bits - 1;
        const long long c = c0 + (long long)j * 256 + threadIdx.x;
        unsigned long long p[MW];
        set_mask<MW>(cands + (size_t)c * k, k, p);
#pragma unroll
        for (int w = 0; w < MW; ++w) out[MW * (size_t)pos + w] = p[w];
        ++pos;
    }
}

// ---- small levels, round 5: the end of a level in ONE single-block launch instead of five -------------------------------------------
// k_small_end = k_histogram + k_pruned_append + the two counts the host needs before it accepts the level (candidates the (x,theta)
// stage left doubtful: ST_RETRY -> *n_retry; optimal candidates that missed the region launch: ST_OPT_PENDING -> *n_late); with pub_dst
// (a level without children) also k_publish_words2.  A level of <= 4096 candidates: four trips of 1024 threads.
__device__ __forceinline__ unsigned int load_word_agent(const unsigned int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <int MW>
MPC_GLOBAL void MPC_LB(1024) k_small_end(const int32_t *__restrict__ cands, int n, int k, const uint8_t *__restrict__ status,
                                         unsigned long long *__restrict__ pruned_out, LevelCounters *ctr, int keep_lowdim,
                                         int32_t *__restrict__ n_retry, int32_t *__restrict__ n_late,
                                         const unsigned int *pub_src2, int pub_n2, unsigned int *pub_dst) {
    __shared__ unsigned int hist[16];
    const int lane = threadIdx.x & 63;
    if (threadIdx.x < 16) hist[threadIdx.x] = 0;
    __syncthreads();
    for (int start = 0; start < n; start += 1024) {
        const int i = start + (int)threadIdx.x;
        const int st = i < n ? status[i] : 15;
#pragma unroll
        for (int b = 0; b < 12; ++b) {
            const unsigned long long m = __ballot(st == b);
            if (lane == 0 && m) atomicAdd(&hist[b], (unsigned int)__popcll(m));
        }
        const bool cut = st == ST_INFEASIBLE || (st == ST_OPT_NO_REGION && !keep_lowdim);
        const unsigned long long mc = __ballot(cut);
        if (mc) {
            unsigned int base = 0;
            if (lane == 0) base = atomicAdd(&ctr->n_pruned_new, (unsigned int)__popcll(mc));
            base = (unsigned int)__builtin_amdgcn_readfirstlane((int)base);
            if (cut) {
                unsigned long long pm[MW];
                set_mask<MW>(cands + (size_t)i * k, k, pm);
                const size_t pos = base + (unsigned int)__popcll(mc & ((1ull << lane) - 1ull));
#pragma unroll
                for (int w = 0; w < MW; ++w) pruned_out[MW * pos + w] = pm[w];
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < 8) {     // (k_histogram's bins: status & 7)
        const unsigned int v = hist[threadIdx.x] + (threadIdx.x < 4 ? hist[threadIdx.x + 8] : 0u);
        if (v) atomicAdd(&ctr->status[threadIdx.x], (unsigned long long)v);
    }
    if (threadIdx.x == 0) { *n_retry = (int32_t)hist[7];   /* 7 = ST_RETRY (kernels2.hpp) */ *n_late = (int32_t)hist[ST_OPT_PENDING]; }
    if (pub_dst) {
        __threadfence();
        __syncthreads();
        const unsigned int *src1 = reinterpret_cast<const unsigned int *>(ctr);
        const int n1 = (int)(sizeof(LevelCounters) / 4);
        for (int i = threadIdx.x; i < n1 + pub_n2; i += 1024) pub_dst[i] = load_word_agent(i < n1 ? src1 + i : pub_src2 + (i - n1));
    }
}
// k_scan_small + k_publish_words2 (a level with children: the scan's total is the last count the host waits for)
MPC_GLOBAL void MPC_LB(1024) k_scan_publish(const int32_t *__restrict__ in, int32_t *__restrict__ out, int n, int32_t *total,
                                            const unsigned int *src1, int n1, const unsigned int *src2, int n2, unsigned int *dst) {
    __shared__ int tot;
    int carry = 0;
    for (int start = 0; start < n; start += SCAN_BLOCK) {
        const int i = start + (int)threadIdx.x;
        const int ex = block_exclusive_scan_1024(i < n ? in[i] : 0, &tot);
        __syncthreads();
        if (i < n) out[i] = ex + carry;
        carry += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry;
    __threadfence();
    __syncthreads();
    for (int i = threadIdx.x; i < n1 + n2; i += 1024) dst[i] = load_word_agent(i < n1 ? src1 + i : src2 + (i - n1));
}

// ---- records of the candidates the register region kernel gave up on, into their slots ON THE DEVICE (round 5) -------------------------
// k_region2 marks such a slot ST_RETRY, the LDS-engine kernel k_region re-solves the candidate into a FIXED-stride record (recd / reci,
// in the order of retry_list).  Until round 5 the host merged those into the slot arrays after the copy (level_regions_slots_impl:
// eight copy commands and a synchronisation per member -- 2-3 ms per level of 128 programs); this kernel writes them into the slot
// arrays where they are, the rows [f | E] behind the pooled ones (row0 on), so the member's records leave with the shared copy launch.
// One block per member; same arithmetic-free moves as the host loop, same slot contents.
constexpr int RRETRY_MERGE_MAX = 1024;
struct RretryMerge {
    const int32_t *opt_list; const int32_t *rlist; const uint8_t *status; const double *recd; const int32_t *reci;
    double *headd; int32_t *headi; double *epool;
    int n_opt, n_rretry, rec_d, rec_i, fd, fi, row0, nx, nt, nc, ntc, k;
};
MPC_GLOBAL void MPC_LB(256) k_rretry_merge(RretryMerge a) {
    __shared__ int slots[RRETRY_MERGE_MAX], src[RRETRY_MERGE_MAX], rowoff[RRETRY_MERGE_MAX];
    __shared__ int wc[4];
    __shared__ int m;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) m = 0;
    __syncthreads();
    for (int start = 0; start < a.n_opt; start += 256) {     // the slots marked "given up", in slot order
        const int w = start + tid;
        const bool f = w < a.n_opt && a.headi[(size_t)w * a.fi] == 7;   // 7 = ST_RETRY (kernels2.hpp)
        const unsigned long long mk = __ballot(f);
        if (lane == 0) wc[wave] = __popcll(mk);
        __syncthreads();
        int before = m;
        for (int v = 0; v < wave; ++v) before += wc[v];
        const int pos = before + __popcll(mk & ((1ull << lane) - 1ull));
        if (f && pos < RRETRY_MERGE_MAX) slots[pos] = w;
        __syncthreads();
        if (tid == 0) m += wc[0] + wc[1] + wc[2] + wc[3];
        __syncthreads();
    }
    const int mm = min(m, RRETRY_MERGE_MAX);
    for (int j = tid; j < mm; j += 256) {
        const int cand = a.opt_list[slots[j]];
        int r = -1;
        for (int q = 0; q < a.n_rretry; ++q) if (a.rlist[q] == cand) { r = q; break; }
        src[j] = r;
        rowoff[j] = (r >= 0 && a.status[cand] == ST_REGION) ? a.reci[(size_t)r * a.rec_i + 1] : 0;
    }
    __syncthreads();
    if (tid == 0) { int run = a.row0; for (int j = 0; j < mm; ++j) { const int ne = rowoff[j]; rowoff[j] = run; run += ne; } }
    __syncthreads();
    const int nx = a.nx, nt = a.nt, nc = a.nc, ntc = a.ntc, k = a.k, nr = nt + 1;
    for (int j = 0; j < mm; ++j) {
        const int w = slots[j], cand = a.opt_list[w], r = src[j];
        const int st = a.status[cand];
        int32_t *oi = a.headi + (size_t)w * a.fi;
        double *od = a.headd + (size_t)w * a.fd;
        for (int i = tid; i < a.fi; i += 256) oi[i] = i == 0 ? st : (i == 1 ? cand : (i < 8 ? 0 : -1));
        if (r < 0 || st != ST_REGION) continue;      // (uniform)
        for (int i = tid; i < a.fd; i += 256) od[i] = 0.0;
        __syncthreads();
        const double *rd = a.recd + (size_t)r * a.rec_d;
        const int32_t *ri = a.reci + (size_t)r * a.rec_i;
        const int kk = ri[0], nE = ri[1], n_om = ri[2], n_la = ri[3], n_re = ri[4];
        const double *Al = rd + nx * nt + nx, *bl = Al + (size_t)nc * nt, *E = bl + nc, *f = E + (size_t)(nc + ntc) * nt;
        for (int i = tid; i < nx * nt + nx; i += 256) od[i] = rd[i];
        for (int i = tid; i < kk * nt; i += 256) od[nx * nt + nx + i] = Al[i];
        for (int i = tid; i < kk; i += 256) od[nx * nt + nx + k * nt + i] = bl[i];
        if (tid == 0) { oi[2] = nE; oi[3] = n_om; oi[4] = n_la; oi[5] = n_re; oi[6] = rowoff[j]; }
        int32_t *act = oi + 8, *om = act + k, *la = om + ntc, *ridx = la + k, *rcon = ridx + (nc - k);
        for (int i = tid; i < kk; i += 256) act[i] = ri[5 + i];
        for (int i = tid; i < n_om; i += 256) om[i] = ri[5 + nc + i];
        for (int i = tid; i < n_la; i += 256) la[i] = ri[5 + nc + ntc + i];
        for (int i = tid; i < n_re; i += 256) { ridx[i] = ri[5 + nc + ntc + nc + i]; rcon[i] = ri[5 + nc + ntc + nc + nc + i]; }
        for (int i = tid; i < nE * nr; i += 256) {
            const int rr = i / nr, t = i - rr * nr;
            a.epool[(size_t)(rowoff[j] + rr) * nr + t] = t == 0 ? f[rr] : E[(size_t)rr * nt + (t - 1)];
        }
        __syncthreads();
    }
}

// graph mode (MPC_LEVEL_GRAPH): the (x,theta) feasibility question is not posed -- "feasibility open" becomes "no region"
MPC_GLOBAL void k_close_open(const int32_t *__restrict__ list, int n_list, uint8_t *__restrict__ status) {
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w < n_list) { const int c = list[w]; status[c] = (uint8_t)(status[c] == 9 ? ST_SINGULAR : ST_FEASIBLE); }   // 9 = ST_NEEDX_SING
}

// status[list[w]] = tmp[list[w]]  (results of a retry kernel that ran on a side stream)
MPC_GLOBAL void k_apply_status(const int32_t *__restrict__ list, int n_list, const uint8_t *__restrict__ tmp, uint8_t *__restrict__ status) {
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w < n_list) { const int c = list[w]; status[c] = tmp[c]; }
}

// Spare region slots behind the slots of the overlapped region launch (level_run_impl): slot first + j gets status word `st`
// and candidate list[j] (or -1).  st = 0: unused slot; st = 7 (ST_RETRY): a candidate that turned out optimal after the launch,
// its record comes from the LDS-engine kernel like that of a candidate k_region2 gave up on.
MPC_GLOBAL void k_init_slots(int32_t *__restrict__ head_i, int fi, int first, int count, int st, const int32_t *__restrict__ list) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < count) { int32_t *hi = head_i + (size_t)(first + j) * fi; hi[0] = st; hi[1] = list ? list[j] : -1; }
}
// keeps rows start, start+stride, ... of a row-major int matrix (frontier sharding, mpc_frontier_shard)
MPC_GLOBAL void k_take_rows(const int32_t *__restrict__ src, long long n_new, int width, long long start, long long stride,
                            int32_t *__restrict__ dst) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_new * width) return;
    const long long r = idx / width;
    dst[idx] = src[(start + r * stride) * width + (idx - r * width)];
}

// the base active set: the equality rows alone (one candidate of n_eq indices)
MPC_GLOBAL void k_base_frontier(int n_eq, int32_t *out) { for (int i = threadIdx.x; i < n_eq; i += blockDim.x) out[i] = i; }
MPC_GLOBAL void k_root_frontier(int n_eq, int n_c, int32_t *out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int cnt = n_c - n_eq;
    if (i >= cnt) return;
    int32_t *o = out + (size_t)i * (n_eq + 1);
    for (int j = 0; j < n_eq; ++j) o[j] = j;
    o[n_eq] = n_eq + i;
}

// ------------------------------------------------------------------------------------------------------------
// k_lp_batch: generic LPs, one wavefront each (deterministic-solver plug)
// ------------------------------------------------------------------------------------------------------------
MPC_GLOBAL void MPC_LB(64) k_lp_batch(long long n_lp, int m, int n, int ld, const double *__restrict__ A, int shared_A,
                                                 const double *__restrict__ b, int shared_b, const double *__restrict__ c,
                                                 int shared_c, const uint8_t *__restrict__ eq, int32_t *__restrict__ status,
                                                 double *__restrict__ x, double *__restrict__ obj, int32_t *__restrict__ iters,
                                                 int32_t *__restrict__ tight, unsigned int *work) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int lane = lane_id();
    double *T = smem;
    int *ib = reinterpret_cast<int *>(smem + (size_t)(m + 1) * ld);
    Lp lp;
    lp.T = T; lp.ld = ld; lp.colvar = ib; lp.rowvar = ib + ld + 1; lp.rowkind = ib + ld + 1 + m + 2;
    for (;;) {
        unsigned int w = 0;
        if (lane == 0) w = atomicAdd(work, 1u);
        w = (unsigned)__builtin_amdgcn_readfirstlane((int)w);
        if (w >= n_lp) break;
        const double *Aw = A + (shared_A ? 0 : (size_t)w * m * n);
        const double *bw = b + (shared_b ? 0 : (size_t)w * m);
        const double *cw = c ? c + (shared_c ? 0 : (size_t)w * n) : nullptr;
        lp.m = m; lp.n = n; lp.iters = 0;
        int *pri_buf = lp.rowkind + m + 2;
        const int r = lp_solve(lp, cw != nullptr, pri_buf, [&](const int *pri) {
            wave_sync();
            for (int i = 0; i < m; ++i)
                for (int j = lane; j < n; j += 64) T[i * ld + 1 + j] = Aw[(size_t)i * n + j];
            for (int i = lane; i < m; i += 64) {
                T[i * ld] = bw[i];
                T[i * ld + n + 1] = 0.0;
                lp.rowkind[i] = eq[(size_t)w * m + i] ? RK_EQ : ((pri && pri[i]) ? RK_PRI : RK_INEQ);
            }
            for (int j = lane; j <= n + 1; j += 64) T[m * ld + j] = (cw && j >= 1 && j <= n) ? cw[j - 1] : 0.0;
        });
        if (x) {
            for (int j = lane; j < n; j += 64) x[(size_t)w * n + j] = 0.0;
            wave_sync();
            if (r == LP_OPTIMAL)
                for (int i = lane; i < m; i += 64)
                    if (lp.rowkind[i] == RK_FREE) x[(size_t)w * n + lp.rowvar[i]] = T[i * ld];
        }
        if (tight) {
            // rows whose slack is nonbasic in the final dictionary (the vertex's active rows, equalities included)
            for (int i = lane; i < m; i += 64) tight[(size_t)w * m + i] = 1;
            wave_sync();
            for (int i = lane; i < m; i += 64) {
                const int v = lp.rowvar[i];
                if (v >= n && v < n + m) tight[(size_t)w * m + (v - n)] = 0;
            }
        }
        if (lane == 0) {
            status[w] = r;
            if (obj) obj[w] = r == LP_OPTIMAL ? -T[m * ld] : 0.0;
            if (iters) iters[w] = lp.iters;
        }
        wave_sync();
    }
}

// ------------------------------------------------------------------------------------------------------------
// k_facet_centres: the Chebyshev centre and radius of EVERY FACET of a batch of polytopes, one wavefront per facet
// (the geometric algorithm's get_facet_centers, solver_utils.py:204-250 -> chebyshev_ball, utils/chebyshev_ball.py:10-63).
// ef: stacked rows [f | E] of all regions, row_off[r] .. row_off[r+1] the rows of region r; facet q = global row q.
// LP over (theta, r):  max r  s.t.  E_j theta + ||E_j|| r <= f_j (j != i),  E_i theta = f_i,  -r <= 0.
// The tableau is filled in LDS straight from the region's rows: nothing but the rows themselves crosses PCIe.
// ------------------------------------------------------------------------------------------------------------
MPC_GLOBAL void MPC_LB(64) k_facet_centres(long long n_facets, int nt, int m_max, int ld, const double *__restrict__ ef,
                                                      const long long *__restrict__ row_off, const int32_t *__restrict__ region_of_row,
                                                      double *__restrict__ centre, double *__restrict__ radius, int32_t *__restrict__ status,
                                                      unsigned int *work) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int lane = lane_id(), n = nt + 1, nr = nt + 1;
    double *T = smem;
    int *ib = reinterpret_cast<int *>(smem + (size_t)(m_max + 2) * ld);
    Lp lp;
    lp.T = T; lp.ld = ld; lp.colvar = ib; lp.rowvar = ib + ld + 1; lp.rowkind = ib + ld + 1 + (m_max + 1) + 2;
    for (;;) {
        unsigned int w = 0;
        if (lane == 0) w = atomicAdd(work, 1u);
        w = (unsigned)__builtin_amdgcn_readfirstlane((int)w);
        if (w >= n_facets) break;
        const int reg = region_of_row[w];
        const long long r0 = row_off[reg];
        const int rows = (int)(row_off[reg + 1] - r0), mine = (int)((long long)w - r0);
        const int m = rows + 1;                      // the region's rows + the row -r <= 0
        lp.m = m; lp.n = n; lp.iters = 0;
        int *pri_buf = lp.rowkind + (m_max + 1) + 2;
        const int r = lp_solve(lp, true, pri_buf, [&](const int *pri) {
            wave_sync();
            for (int i = lane; i < m; i += 64) {
                double *Ti = T + (size_t)i * ld;
                if (i < rows) {
                    const double *row = ef + (size_t)(r0 + i) * nr;
                    double ss = 0.0;
                    for (int t = 0; t < nt; ++t) { Ti[1 + t] = row[1 + t]; ss = fma(row[1 + t], row[1 + t], ss); }
                    Ti[0] = row[0];
                    Ti[1 + nt] = i == mine ? 0.0 : sqrt(ss);           // the facet's own row: an equality, no radius term
                    lp.rowkind[i] = i == mine ? RK_EQ : ((pri && pri[i]) ? RK_PRI : RK_INEQ);
                } else {
                    for (int t = 0; t < nt; ++t) Ti[1 + t] = 0.0;
                    Ti[0] = 0.0; Ti[1 + nt] = -1.0;                     // -r <= 0
                    lp.rowkind[i] = (pri && pri[i]) ? RK_PRI : RK_INEQ;
                }
                Ti[n + 1] = 0.0;
            }
            for (int j = lane; j <= n + 1; j += 64) T[(size_t)m * ld + j] = j == n ? -1.0 : 0.0;   // minimise -r
        });
        for (int j = lane; j < nt; j += 64) centre[(size_t)w * nt + j] = 0.0;
        if (lane == 0) radius[w] = 0.0;
        wave_sync();
        if (r == LP_OPTIMAL)
            for (int i = lane; i < m; i += 64)
                if (lp.rowkind[i] == RK_FREE) {
                    const int v = lp.rowvar[i];
                    if (v < nt) centre[(size_t)w * nt + v] = T[(size_t)i * ld];
                    else if (v == nt) radius[w] = T[(size_t)i * ld];
                }
        if (lane == 0) status[w] = r;
        wave_sync();
    }
}

}  // namespace mpc
