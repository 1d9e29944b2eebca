// qp.hpp -- the strictly convex QP of an mpQP at fixed parameter points, batched: one wavefront per point (gfx950).
//
// Reference: MPQP_Program.solve_theta (mpqp_program.py:109-143) -> Solver.solve_qp -> quadprog / gurobi / daqp
// (solver_interface/quad_prog_interface.py:16-89), one QP per call.  Callers: sample_theta_space (mplp_program.py:632-664,
// the seeds of the graph algorithms), Solution.verify_theta / verify_solution (solution.py:114-174).
//
//     min 1/2 x'Qx + (c + H theta)'x   s.t.  A x <= b + F theta  (first n_eq rows equalities),   Q > 0
//
// With x = -Q^-1 (c + H theta + A' lambda) the KKT conditions are a linear complementarity problem in the multipliers alone,
//     s = q(theta) + W lambda,   s >= 0, lambda >= 0, s'lambda = 0      (s_e = 0, lambda_e free on equality rows)
// with W = A Q^-1 A' and q = UV [1; theta] -- the blocks the combinatorial kernels already keep on the device (mpc_create).
// The LCP is solved by Lemke's complementary pivoting on an LDS dictionary (lp_engine.hpp's pivot): the multipliers of the
// equality rows enter first and their slacks are deleted (a principal block pivot), the covering variable z0 enters on the
// most negative q_i, then the complement of whatever left enters until z0 leaves (solved) or no row limits the entering
// variable (ray termination: the QP is infeasible at this theta).  W is positive semidefinite, so Lemke terminates in one
// of the two.  Ratio ties are broken in favour of z0's row, then by the lowest variable id.
#pragma once
#include "lp_engine.hpp"

namespace mpc {

enum : int { QP_OPTIMAL = 0, QP_INFEASIBLE = 1, QP_ITERLIMIT = 3 };

__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off));
    return v;
}

__global__ void __launch_bounds__(64) k_qp_batch(long long n_qp, int nc, int n_eq, int nt, int nx, int ld, const double *__restrict__ W,
                                                 const double *__restrict__ UV, const double *__restrict__ X0H, const double *__restrict__ Gt,
                                                 const double *__restrict__ theta, int32_t *__restrict__ status, double *__restrict__ x,
                                                 double *__restrict__ lam, uint8_t *__restrict__ active, int32_t *__restrict__ iters,
                                                 unsigned int *work) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int lane = lane_id(), nr = nt + 1;
    double *T = smem;
    int *ib = reinterpret_cast<int *>(smem + (size_t)(nc + 1) * ld);
    Lp lp;
    lp.T = T; lp.ld = ld; lp.colvar = ib; lp.rowvar = ib + ld + 1; lp.rowkind = ib + ld + 1 + nc + 2;
    // (ld + 1) + 2 (nc + 2) ints precede it and ld is odd: the int block has an even length, lamv is 8-byte aligned as it stands
    double *lamv = reinterpret_cast<double *>(lp.rowkind + nc + 2);   // multipliers of the solved point
    const int ID_Z0 = 2 * nc;
    for (;;) {
        unsigned int w = 0;
        if (lane == 0) w = atomicAdd(work, 1u);
        w = (unsigned)__builtin_amdgcn_readfirstlane((int)w);
        if (w >= n_qp) break;
        const double *th = theta + (size_t)w * nt;
        lp.m = nc; lp.n = 0; lp.na = nc + 1; lp.iters = 0; lp.max_iter = 50 * nc + 100; lp.growth = 0.0;
        wave_sync();
        // dictionary: s_i = q_i - sum_j (-W_ij) lambda_j - (-1) z0 ; the cost row (index nc) is unused
        for (int i = lane; i <= nc; i += 64) {
            double *Ti = T + (size_t)i * ld;
            double q = 0.0;
            if (i < nc) { q = UV[(size_t)i * nr]; for (int t = 0; t < nt; ++t) q = fma(UV[(size_t)i * nr + 1 + t], th[t], q); }
            Ti[0] = q;
            for (int j = 0; j < nc; ++j) Ti[1 + j] = i < nc ? -W[(size_t)i * nc + j] : 0.0;
            Ti[nc + 1] = (i >= n_eq && i < nc) ? -1.0 : 0.0;
            if (i < nc) { lp.rowvar[i] = nc + i; lp.rowkind[i] = i < n_eq ? RK_EQ : RK_INEQ; }
        }
        for (int j = lane; j <= nc + 1; j += 64) lp.colvar[j] = j == 0 ? -1 : (j <= nc ? j - 1 : ID_Z0);
        wave_sync();
        int st = QP_OPTIMAL;
        // equality rows: lambda_e enters on its own row (largest remaining diagonal first would be safer; W_ee > 0 for
        // independent equality rows), the slack s_e is fixed at zero and its column deleted
        for (int e = 0; e < n_eq && st == QP_OPTIMAL; ++e) {
            int q = -1;
            for (int j = 1 + lane; j <= lp.na; j += 64) if (lp.colvar[j] == e) q = j;
            q = wave_max_i(q);
            if (q < 0 || !(fabs(T[(size_t)e * ld + q]) > 1e-12)) { st = QP_INFEASIBLE; break; }   // dependent equality rows
            lp_pivot(lp, e, q);
            lp_drop_col(lp, q);
            if (lane == 0) lp.rowkind[e] = RK_FREE;
            wave_sync();
        }
        if (st == QP_OPTIMAL) {
            // most negative value among the inequality rows
            double vmin = 0.0; int key = 0, r = -1;
            for (int i = lane; i < nc; i += 64)
                if (lp.rowkind[i] == RK_INEQ) { const double v = T[(size_t)i * ld]; if (r < 0 || v < vmin) { vmin = v; r = i; } }
            reduce_min_first(vmin, key, r);
            if (r >= 0 && vmin < -1e-12) {
                int qz = -1;
                for (int j = 1 + lane; j <= lp.na; j += 64) if (lp.colvar[j] == ID_Z0) qz = j;
                qz = wave_max_i(qz);
                int left = lp.rowvar[r];
                lp_pivot(lp, r, qz);
                for (;;) {
                    if (lp.iters > lp.max_iter) { st = QP_ITERLIMIT; break; }
                    const int enter = left < nc ? left + nc : left - nc;   // the complement of the variable that left
                    int q = -1;
                    for (int j = 1 + lane; j <= lp.na; j += 64) if (lp.colvar[j] == enter) q = j;
                    q = wave_max_i(q);
                    if (q < 0) { st = QP_INFEASIBLE; break; }           // its column was deleted: cannot happen for inequality rows
                    // ratio test over the sign-restricted rows (the free multipliers of the equality rows never leave)
                    double best = INFINITY; int bz = 0, bvar = 0, br = -1;
                    for (int i = lane; i < nc; i += 64) {
                        if (lp.rowkind[i] != RK_INEQ) continue;
                        const double a = T[(size_t)i * ld + q];
                        if (!(a > TOL_PIV)) continue;
                        const double ratio = fmax(T[(size_t)i * ld], 0.0) / a;
                        const int isz = lp.rowvar[i] == ID_Z0 ? 1 : 0, var = lp.rowvar[i];
                        if (bland_better(ratio, isz, var, best, bz, bvar, br)) { best = ratio; bz = isz; bvar = var; br = i; }
                    }
                    double piv = 0.0;
                    reduce_ratio(best, piv, bz, bvar, br, true);
                    if (br < 0) { st = QP_INFEASIBLE; break; }          // ray termination
                    left = lp.rowvar[br];
                    lp_pivot(lp, br, q);
                    if (left == ID_Z0) break;                             // z0 left the basis: complementary solution
                }
            }
        }
        // outputs
        for (int i = lane; i < nc; i += 64) lamv[i] = 0.0;
        wave_sync();
        if (st == QP_OPTIMAL)
            for (int i = lane; i < nc; i += 64) { const int v = lp.rowvar[i]; if (v < nc) lamv[v] = T[(size_t)i * ld]; }
        wave_sync();
        if (lam) for (int i = lane; i < nc; i += 64) lam[(size_t)w * nc + i] = lamv[i];
        if (active) {
            // a constraint is active when its slack is nonbasic (zero) in the final dictionary
            for (int i = lane; i < nc; i += 64) active[(size_t)w * nc + i] = st == QP_OPTIMAL ? 1 : 0;
            wave_sync();
            if (st == QP_OPTIMAL)
                for (int i = lane; i < nc; i += 64) { const int v = lp.rowvar[i]; if (v >= nc && v < 2 * nc) active[(size_t)w * nc + (v - nc)] = 0; }
        }
        if (x) {
            for (int a = lane; a < nx; a += 64) {
                double v = X0H[(size_t)a * nr];
                for (int t = 0; t < nt; ++t) v = fma(X0H[(size_t)a * nr + 1 + t], th[t], v);
                for (int i = 0; i < nc; ++i) v = fma(-lamv[i], Gt[(size_t)i * nx + a], v);
                x[(size_t)w * nx + a] = st == QP_OPTIMAL ? v : __longlong_as_double(0x7ff8000000000000ll);
            }
        }
        if (lane == 0) { status[w] = st; if (iters) iters[w] = lp.iters; }
        wave_sync();
    }
}

}  // namespace mpc
