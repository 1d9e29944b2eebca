// lp_reg.hpp -- register-resident simplex for the hot kernels (gfx950): one wavefront per LP, NO LDS, NO barriers.
//
// Lane i owns tableau row i (and row i+64 when SLOTS == 2) in VGPRs: t[s][0] = value of the basic variable,
// t[s][1..NC-1] = coefficients of the nonbasic columns (NC-1 slots for at most NC-2 variables: phase 1 puts its
// artificial x0 into whichever slot is free).  All column loops
// are unrolled at compile time (NC is a template parameter) so every tableau entry is a named register.
//   * the pivot row is broadcast column by column with v_readlane (uniform lane index) -> SGPRs
//   * the pivot column of each lane's row is extracted with a uniform select chain
//   * pricing reads the cost row out of ONE lane with v_readlane: no cross-lane reduction at all
//   * the ratio test is the only cross-lane step: three DPP (row_shr) reductions finished with v_readlane
//   * a deleted column is a cleared bit in a uniform mask; the var id of column j lives in lane j of `cv`
// The LPs solved here have no free variables: the caller supplies dictionaries that are already expressed at a vertex
// (theta-space LPs at a vertex of {A_t theta <= b_t}; the (x,theta) LP at the program's pre-crashed vertex), so only
// "equality rows leave the basis", phase 1 (x0 method) and phase 2 are needed.  Pivot rules, tolerances and the
// arithmetic of a pivot are those of lp_engine.hpp; a pivot smaller than 1e-3 of its column marks the LP as doubtful
// (growth > GROWTH_SAFE) and the caller re-solves that candidate with the LDS engine, which can refactorise.
#pragma once
#include "lp_engine.hpp"

namespace mpc {

constexpr int RK_COST = 7;  // the objective row: updated by pivots, never in a ratio test

__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

__device__ __forceinline__ double readlane_f64(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

template <int CTRL> __device__ __forceinline__ int dpp_i32(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false); }
template <int CTRL> __device__ __forceinline__ double dpp_f64(double v) {
    return __hiloint2double(dpp_i32<CTRL>(__double2hiint(v)), dpp_i32<CTRL>(__double2loint(v)));
}
// row_shr:1,2,4,8 leave the reduction of each 16-lane row in its last lane; four v_readlane finish it
__device__ __forceinline__ double dpp_wave_min(double v) {
    v = fmin(v, dpp_f64<0x111>(v)); v = fmin(v, dpp_f64<0x112>(v)); v = fmin(v, dpp_f64<0x114>(v)); v = fmin(v, dpp_f64<0x118>(v));
    return fmin(fmin(readlane_f64(v, 15), readlane_f64(v, 31)), fmin(readlane_f64(v, 47), readlane_f64(v, 63)));
}
__device__ __forceinline__ double dpp_wave_max(double v) {
    v = fmax(v, dpp_f64<0x111>(v)); v = fmax(v, dpp_f64<0x112>(v)); v = fmax(v, dpp_f64<0x114>(v)); v = fmax(v, dpp_f64<0x118>(v));
    return fmax(fmax(readlane_f64(v, 15), readlane_f64(v, 31)), fmax(readlane_f64(v, 47), readlane_f64(v, 63)));
}
// argmax of a 64-bit key, ties -> lowest index; idx < 0 = no candidate.  Result uniform.
__device__ __forceinline__ bool key_better(unsigned long long ok, int oi, unsigned long long k, int i) {
    return oi >= 0 && (i < 0 || ok > k || (ok == k && oi < i));
}
template <int CTRL> __device__ __forceinline__ void argmax_step(unsigned long long &key, int &idx) {
    const unsigned lo = (unsigned)dpp_i32<CTRL>((int)(unsigned)key), hi = (unsigned)dpp_i32<CTRL>((int)(unsigned)(key >> 32));
    const int oi = dpp_i32<CTRL>(idx);
    const unsigned long long ok = ((unsigned long long)hi << 32) | lo;
    if (key_better(ok, oi, key, idx)) { key = ok; idx = oi; }
}
__device__ __forceinline__ int dpp_wave_argmax(unsigned long long key, int idx) {
    argmax_step<0x111>(key, idx); argmax_step<0x112>(key, idx); argmax_step<0x114>(key, idx); argmax_step<0x118>(key, idx);
    unsigned long long bk = 0; int bi = -1;
#pragma unroll
    for (int l = 15; l < 64; l += 16) {
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)key, l), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(key >> 32), l);
        const int oi = __builtin_amdgcn_readlane(idx, l);
        const unsigned long long ok = ((unsigned long long)hi << 32) | lo;
        if (key_better(ok, oi, bk, bi)) { bk = ok; bi = oi; }
    }
    return bi;
}
// order-preserving map double -> u64 (larger double <=> larger key)
__device__ __forceinline__ unsigned long long f64_key(double v) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    return (b & 0x8000000000000000ull) ? ~b : (b | 0x8000000000000000ull);
}

template <int NC, int SLOTS>
struct RegLp {
    static constexpr int XC = NC - 1;  // column reserved for x0
    double t[SLOTS][NC];
    int kind[SLOTS];
    int var[SLOTS];
    int cv;           // lane j holds the variable id of column j
    unsigned alive;   // uniform bit mask of live columns (bits 1..NC-1)
    int m;            // rows in use (row index = lane + 64*slot)
    int iters, max_iter;
    double growth;

    __device__ __forceinline__ double col(int s, int q) const {
        double f = 0.0;
#pragma unroll
        for (int j = 1; j < NC; ++j) f = (j == q) ? t[s][j] : f;
        return f;
    }
    __device__ __forceinline__ double row_entry(int r, const double (&c)[SLOTS]) const {
        const int rl = r & 63;
        if (SLOTS == 1) return readlane_f64(c[0], rl);
        return readlane_f64((r >> 6) ? c[SLOTS - 1] : c[0], rl);
    }

    __device__ __forceinline__ void pivot(int r_in, int q_in) {
        const int lane = lane_id(), r = uni(r_in), q = uni(q_in);
        double f[SLOTS];
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) f[s] = col(s, q);
        const double inv = 1.0 / row_entry(r, f);
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            double cj[SLOTS];
#pragma unroll
            for (int s = 0; s < SLOTS; ++s) cj[s] = t[s][j];
            const double trj = row_entry(r, cj) * inv;
#pragma unroll
            for (int s = 0; s < SLOTS; ++s) {
                const bool is_r = (lane + 64 * s) == r;
                double nv;
                if (j == q) nv = is_r ? inv : -f[s] * inv;
                else nv = is_r ? trj : fma(-f[s], trj, t[s][j]);
                t[s][j] = nv;
            }
        }
        const int vq = __builtin_amdgcn_readlane(cv, q);
        int vr = 0;
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) if ((r >> 6) == s) vr = __builtin_amdgcn_readlane(var[s], r & 63);
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) if ((lane + 64 * s) == r) var[s] = vq;
        if (lane == q) cv = vr;
        iters++;
    }

    __device__ __forceinline__ void drop_col(int q) { alive = (unsigned)uni((int)(alive & ~(1u << q))); }

    __device__ __forceinline__ void set_kind(int r, int k) {
        const int lane = lane_id();
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) if ((lane + 64 * s) == r) kind[s] = k;
    }
    __device__ __forceinline__ int get_kind(int r) const {
        int k = 0;
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) if ((r >> 6) == s) k = __builtin_amdgcn_readlane(kind[s], r & 63);
        return k;
    }
    __device__ __forceinline__ double beta(int r) const {
        double c0[SLOTS];
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) c0[s] = t[s][0];
        return row_entry(r, c0);
    }

    // largest |entry| > TOL_PIV of row r over the live columns (ties: lowest column); -1 if none
    __device__ __forceinline__ int best_col(int r_in) const {
        const int r = uni(r_in);
        const unsigned alive = (unsigned)uni((int)this->alive);
        int q = -1;
        double best = TOL_PIV;
#pragma unroll
        for (int j = 1; j < NC; ++j) {
            if (!((alive >> j) & 1u)) continue;
            double cj[SLOTS];
#pragma unroll
            for (int s = 0; s < SLOTS; ++s) cj[s] = t[s][j];
            const double a = fabs(row_entry(r, cj));
            if (a > best) { best = a; q = j; }
        }
        return q;
    }

    // primal simplex; phase1_row >= 0: minimise x0 (basic in that row), else minimise the RK_COST row `cost_row`.
    // 0 optimal, 2 unbounded, 3 iteration limit, 4 x0 left the basis
    // drop_on_leave: when the minimised variable leaves the basis its column is deleted (it stays fixed at zero);
    // false = it stays a regular nonbasic variable (facet tests in k_region2).
    __device__ __forceinline__ int primal(int phase1_row, int cost_row, bool drop_on_leave = true) {
        const int lane = lane_id();
        int deg = 0;
        for (;;) {
            if (iters > max_iter) return 3;
            const bool bland = deg > DEG_SWITCH;
            const unsigned alive = (unsigned)uni((int)this->alive);
            const int crow = uni(phase1_row >= 0 ? phase1_row : cost_row);
            const double sgn = phase1_row >= 0 ? -1.0 : 1.0;
            if (phase1_row >= 0 && beta(phase1_row) <= TOL_FEAS) return 0;
            // pricing: the cost row lives in one lane -> uniform scan with v_readlane
            int q = -1, best_var = 0;
            double best = -TOL_COST;
#pragma unroll
            for (int j = 1; j < NC; ++j) {
                if (!((alive >> j) & 1u)) continue;
                double cj[SLOTS];
#pragma unroll
                for (int s = 0; s < SLOTS; ++s) cj[s] = t[s][j];
                const double d = sgn * row_entry(crow, cj);
                if (d < -TOL_COST) {
                    if (bland) {
                        const int v = __builtin_amdgcn_readlane(cv, j);
                        if (q < 0 || v < best_var) { q = j; best_var = v; }
                    } else if (d < best) { best = d; q = j; }
                }
            }
            if (q < 0) return 0;
            // ratio test (Harris two-pass; textbook + Bland while stalled)
            double a[SLOTS], ratio[SLOTS];
            bool elig[SLOTS];
            double colmax = 0.0, tmax = INFINITY;
#pragma unroll
            for (int s = 0; s < SLOTS; ++s) {
                const int i = lane + 64 * s;
                a[s] = col(s, q);
                const bool used = i < m && kind[s] != RK_DEAD;
                if (used) colmax = fmax(colmax, fabs(a[s]));
                elig[s] = used && (kind[s] == RK_INEQ || kind[s] == RK_X0) && a[s] > TOL_PIV;
                ratio[s] = 0.0;
                if (elig[s]) {
                    const double b0 = fmax(t[s][0], 0.0);
                    ratio[s] = b0 / a[s];
                    tmax = fmin(tmax, (b0 + HARRIS_DELTA) / a[s]);
                }
            }
            colmax = dpp_wave_max(colmax);
            tmax = dpp_wave_min(tmax);
            if (tmax == INFINITY) return 2;
            int r;
            if (bland) {
                double rm = INFINITY;
#pragma unroll
                for (int s = 0; s < SLOTS; ++s) if (elig[s]) rm = fmin(rm, ratio[s]);
                rm = dpp_wave_min(rm);
                unsigned long long key = 0; int idx = -1;
#pragma unroll
                for (int s = 0; s < SLOTS; ++s)
                    if (elig[s] && ratio[s] == rm) {
                        const unsigned long long kk = ((unsigned long long)(kind[s] == RK_X0) << 40) | (unsigned long long)(0x7fffffff - var[s]);
                        if (key_better(kk, lane + 64 * s, key, idx)) { key = kk; idx = lane + 64 * s; }
                    }
                r = dpp_wave_argmax(key, idx);
            } else {
                unsigned long long key = 0; int idx = -1;
#pragma unroll
                for (int s = 0; s < SLOTS; ++s)
                    if (elig[s] && !(ratio[s] > tmax)) {
                        // a > 0: its bit pattern orders like the value; x0 row first
                        const unsigned long long kk = (unsigned long long)__double_as_longlong(a[s]) | ((unsigned long long)(kind[s] == RK_X0) << 63);
                        if (key_better(kk, lane + 64 * s, key, idx)) { key = kk; idx = lane + 64 * s; }
                    }
                r = dpp_wave_argmax(key, idx);
            }
            if (r < 0) return 2;
            const double rpiv = row_entry(r, a), rmin = row_entry(r, ratio);
            growth = fmax(growth, colmax / rpiv);
            deg = (rmin <= 0.0) ? deg + 1 : 0;
            const bool leaving_x0 = get_kind(r) == RK_X0;
            pivot(r, q);
            if (leaving_x0) { set_kind(r, RK_INEQ); if (drop_on_leave) drop_col(q); return 4; }
        }
    }

    // Drives the basic variable of row r to zero and out of the basis while every other row stays feasible: a primal
    // simplex run whose objective is that variable (the x0 machinery with the row's own slack in the role of x0).
    // Used to activate a constraint at a feasible vertex.  LP_OPTIMAL (done, column deleted) / LP_INFEASIBLE (its
    // minimum over the polytope is positive) / LP_ITERLIMIT.
    __device__ __forceinline__ int drive_to_zero(int r) {
        set_kind(r, RK_X0);
        const int st = primal(r, -1);
        if (st == 3) return LP_ITERLIMIT;
        if (st != 4) {
            if (beta(r) > TOL_FEAS) return LP_INFEASIBLE;
            const int qq = best_col(r);
            if (qq < 0) set_kind(r, RK_DEAD);
            else { pivot(r, qq); set_kind(r, RK_INEQ); drop_col(qq); }
        }
        return LP_OPTIMAL;
    }

    // Phase 1 on the current dictionary (re-entrant).  LP_OPTIMAL (feasible) / LP_INFEASIBLE / LP_ITERLIMIT.
    __device__ __forceinline__ int phase1() {
        const int lane = lane_id();
        unsigned long long key = 0; int idx = -1;
#pragma unroll
        for (int s = 0; s < SLOTS; ++s)
            if (lane + 64 * s < m && kind[s] == RK_INEQ && t[s][0] < -TOL_FEAS) {
                const unsigned long long kk = f64_key(-t[s][0]);  // most negative value = largest key
                if (key_better(kk, lane + 64 * s, key, idx)) { key = kk; idx = lane + 64 * s; }
            }
        const int r = dpp_wave_argmax(key, idx);
        if (r < 0) return LP_OPTIMAL;
        // x0 takes a free column slot.  (Not a fixed one: when x0 left the basis in an earlier phase 1 it was deleted from
        // the column it had moved to, and the slot it first entered through now holds a live slack.)
        const unsigned al = (unsigned)uni((int)alive);
        const int xc = __ffs((int)(~al & ~1u & ((NC >= 32 ? 0u : (1u << NC)) - 1u))) - 1;
        alive = al | (1u << xc);
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            const double v = (lane + 64 * s < m && kind[s] == RK_INEQ) ? -1.0 : 0.0;
#pragma unroll
            for (int j = 1; j < NC; ++j) t[s][j] = (j == xc) ? v : t[s][j];
        }
        if (lane == xc) cv = X0_VAR;
        pivot(r, xc);
        return drive_to_zero(r);
    }
};

}  // namespace mpc
