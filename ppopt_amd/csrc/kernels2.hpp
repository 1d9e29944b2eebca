// kernels2.hpp -- fast verdict kernels on the register-resident simplex (lp_reg.hpp): k_theta2 (KKT + two-stage theta LP)
// and k_x2 ((x,theta) feasibility for the candidates k_theta2 left open).
//
// Same decisions as k_verdict (kernels.hpp) for the common case; a candidate whose pivot sequence was numerically
// doubtful gets status ST_RETRY and is re-solved by k_verdict (LDS engine with basis refactorisation).  Requirements,
// checked at mpc_create: Q > 0 or mode-1 KKT as before; a vertex of the parameter polytope {A_t theta <= b_t}
// (`tv_*` blocks: every theta-space row is expressed in the slacks of that vertex's n_t tight rows, so the theta LPs
// need no crash pivots) and the pre-crashed dictionary D0 of the (x,theta) LP; n_t <= NT, D0 columns <= NXC - 2,
// rows <= 64 * SLOTS.
#pragma once
#include "kernels.hpp"
#include "lp_reg.hpp"

namespace mpc {

constexpr int ST_RETRY = 7;       // numerically doubtful: re-solve with the LDS engine (k_verdict)
constexpr int ST_NEEDX = 8;       // theta stage could not show feasibility: (x,theta) LP needed (k_x2)
constexpr int ST_NEEDX_SING = 9;  // the same, and the KKT matrix was singular (a feasible outcome is ST_SINGULAR)

template <int NT, int SLOTS>
__global__ void __launch_bounds__(64, (NT * SLOTS >= 20 ? 3 : 4)) k_theta2(const DevProblem *__restrict__ Pg, const int32_t *__restrict__ cands, long long n, int k,
                                                    uint8_t *__restrict__ status, LevelCounters *__restrict__ ctr) {
    // the program descriptor stays in memory (scalar loads on demand) instead of ~90 live SGPRs
    const DevProblem &P = *Pg;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    Smem s = carve(P, smem);
    const int lane = lane_id(), nt = P.n_t, nr = nt + 1, nv = P.n_x + P.n_t, e = P.n_eq;
    unsigned long long pivots = 0, n_retry = 0;
    long long cyc_kkt = 0, cyc_theta = 0;
    for (;;) {
        unsigned int c = 0;
        if (lane == 0) c = atomicAdd(&ctr->work_verdict, 1u);
        c = (unsigned)__builtin_amdgcn_readfirstlane((int)c);
        if (c >= n) break;
        const int nin = load_active_set(P, cands + (size_t)c * k, k, s);
        int st = -1;
        const long long t0 = clock64();
        const int kk = kkt_solve(P, k, s);
        const long long t1 = clock64();
        cyc_kkt += t1 - t0;
        bool singular = false, retry = false;
        if (kk == 1) st = ST_INFEASIBLE;
        else if (kk == 2) singular = true;
        else if (kk == 0) {
            // ---- theta-space two-stage LP, rows expressed at the vertex of {A_t theta <= b_t} ----------------------
            RegLp<NT + 2, SLOTS> lp;
            const int nlam = k - e, npre = P.n_tpre, m = nlam + nin + npre;
            lp.m = m; lp.iters = 0; lp.max_iter = 50 * (m + nt) + 100; lp.growth = 0.0;
            lp.alive = (nt >= 31 ? 0xfffffffeu : ((1u << (nt + 1)) - 2u));
            lp.cv = nt + m + lane - 1;
#pragma unroll
            for (int sl = 0; sl < SLOTS; ++sl) {
                const int i = lane + 64 * sl;
                double h = 0.0, g[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) g[t] = 0.0;
                lp.var[sl] = nt + i;
                lp.kind[sl] = i < nlam ? RK_PASSIVE : (i < m ? RK_INEQ : RK_DEAD);
                bool pre = false;
                if (i < nlam) {
                    h = s.L[(e + i) * nr];
#pragma unroll
                    for (int t = 0; t < NT; ++t) if (t < nt) g[t] = -s.L[(e + i) * nr + 1 + t];
                } else if (i < nlam + nin) {
                    const int ci = s.inact[i - nlam];
                    if (P.kkt_mode == 0) {
                        double acc[NT + 1];
#pragma unroll
                        for (int t = 0; t <= NT; ++t) acc[t] = t <= nt ? P.UV[ci * nr + t] : 0.0;
                        for (int a = 0; a < k; ++a) {
                            const double w = P.W[ci * P.n_c + s.as[a]];
#pragma unroll
                            for (int t = 0; t <= NT; ++t) if (t <= nt) acc[t] = fma(w, s.L[a * nr + t], acc[t]);
                        }
                        h = acc[0];
#pragma unroll
                        for (int t = 0; t < NT; ++t) g[t] = -acc[1 + t];
                    } else {
                        double acc[NT + 1];
#pragma unroll
                        for (int t = 0; t <= NT; ++t) acc[t] = 0.0;
                        for (int l = 0; l < P.n_x; ++l) {
                            const double w = P.A[ci * P.n_x + l];
#pragma unroll
                            for (int t = 0; t <= NT; ++t) if (t <= nt) acc[t] = fma(w, s.X[l * nr + t], acc[t]);
                        }
                        h = P.b[ci] - acc[0];
#pragma unroll
                        for (int t = 0; t < NT; ++t) if (t < nt) g[t] = acc[1 + t] - P.F[ci * nt + t];
                    }
                } else if (i < m) {
                    pre = true;
                    const double *row = P.tv_rows + (size_t)(i - nlam - nin) * nr;
                    lp.t[sl][0] = row[0];
#pragma unroll
                    for (int t = 0; t < NT; ++t) lp.t[sl][1 + t] = t < nt ? row[1 + t] : 0.0;
                }
                if (!pre) {
                    double mx = 0.0;
#pragma unroll
                    for (int t = 0; t < NT; ++t) mx = fmax(mx, fabs(g[t]));
                    if (!(mx > ZERO_ROW_ATOL)) {
#pragma unroll
                        for (int t = 0; t < NT; ++t) g[t] = 0.0;
                        mx = 0.0;
                    }
                    if (mx > 0.0) {
                        int ex;
                        (void)frexp(mx, &ex);
                        const double sc = ldexp(1.0, -ex);
                        h *= sc;
#pragma unroll
                        for (int t = 0; t < NT; ++t) g[t] *= sc;
                    }
                    // theta = theta_v - Minv sigma:  value at the vertex and coefficients of the tight-row slacks sigma
                    double b0 = h;
#pragma unroll
                    for (int t = 0; t < NT; ++t) if (t < nt) b0 = fma(-g[t], P.tv_theta[t], b0);
                    lp.t[sl][0] = b0;
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        double acc = 0.0;
#pragma unroll
                        for (int t = 0; t < NT; ++t) if (t < nt && j < nt) acc = fma(g[t], P.tv_minv[t * nt + j], acc);
                        lp.t[sl][1 + j] = -acc;
                    }
                }
                lp.t[sl][NT + 1] = 0.0;
            }
            const int r1 = lp.phase1();
            if (r1 == LP_ITERLIMIT) st = ST_LP_LIMIT;
            else if (lp.growth > GROWTH_SAFE) retry = true;
            else if (r1 == LP_OPTIMAL) {
#pragma unroll
                for (int sl = 0; sl < SLOTS; ++sl) if (lp.kind[sl] == RK_PASSIVE) lp.kind[sl] = RK_INEQ;
                const int r2 = lp.phase1();
                if (r2 == LP_ITERLIMIT) st = ST_LP_LIMIT;
                else if (lp.growth > GROWTH_SAFE) retry = true;
                else st = r2 == LP_OPTIMAL ? ST_OPT_PENDING : ST_FEASIBLE;
            }
            pivots += lp.iters;
        }
        const long long t2 = clock64();
        cyc_theta += t2 - t1;
        if (st < 0 && !retry) st = singular ? ST_NEEDX_SING : ST_NEEDX;  // feasibility still open: (x,theta) LP, k_x2
        if (retry) { st = ST_RETRY; n_retry++; }
        if (lane == 0) status[c] = (uint8_t)st;
    }
    if (lane == 0) {
        atomicAdd(&ctr->cycles[0], (unsigned long long)cyc_kkt); atomicAdd(&ctr->cycles[1], (unsigned long long)cyc_theta);
        atomicAdd(&ctr->pivots, pivots); atomicAdd(&ctr->xtheta_fallbacks, n_retry);
    }
}


// (x,theta) feasibility of the candidates list[0..n_list) from the pre-crashed vertex dictionary, in registers.
// No LDS: the kernel needs only the active set and D0.
template <int NXC, int SLOTS>
__global__ void __launch_bounds__(64, (NXC * SLOTS >= 64 ? 2 : 3)) k_x2(const DevProblem *__restrict__ Pg, const int32_t *__restrict__ cands, int k,
                                              const int32_t *__restrict__ list, int n_list, uint8_t *__restrict__ status,
                                              LevelCounters *__restrict__ ctr) {
    const DevProblem &P = *Pg;
    const int lane = lane_id(), nv = P.n_x + P.n_t, e = P.n_eq;
    unsigned long long pivots = 0, n_retry = 0;
    long long cyc_x = 0;
    for (;;) {
        unsigned int w = 0;
        if (lane == 0) w = atomicAdd(&ctr->work_x, 1u);
        w = (unsigned)__builtin_amdgcn_readfirstlane((int)w);
        if (w >= (unsigned)n_list) break;
        const int c = list[w];
        const int32_t *as = cands + (size_t)c * k;
        const bool singular = status[c] == ST_NEEDX_SING;
        bool retry = false;
        int st = -1;
        const long long t2 = clock64();
        {
            // ---- (x,theta) feasibility from the pre-crashed vertex dictionary ------------------------------------------
            RegLp<NXC, SLOTS> lx;
            const int mr = P.n_d0r, nc0 = P.n_d0c;
            lx.m = mr; lx.iters = 0; lx.max_iter = 50 * (mr + nc0) + 100; lx.growth = 0.0;
            lx.alive = (nc0 >= 31 ? 0xfffffffeu : ((1u << (nc0 + 1)) - 2u));
            lx.cv = (lane >= 1 && lane <= nc0) ? nv + P.d0_cols[lane - 1] : -1;
#pragma unroll
            for (int sl = 0; sl < SLOTS; ++sl) {
                const int i = lane + 64 * sl;
                lx.kind[sl] = i < mr ? RK_INEQ : RK_DEAD;
                lx.var[sl] = i < mr ? nv + P.d0_rows[i] : -1;
#pragma unroll
                for (int j = 0; j < NXC; ++j) lx.t[sl][j] = (i < mr && j <= nc0) ? P.d0T[(size_t)j * mr + i] : 0.0;
            }
            // every active row is switched on at the feasible vertex: a nonbasic slack is simply fixed at zero (column
            // deleted); a basic one is driven to zero by a primal simplex run that keeps all other rows feasible, so no
            // phase 1 is needed afterwards.  "Its minimum is positive" <=> the candidate is infeasible.
            int r = LP_OPTIMAL;
            for (int a = e; a < k && r == LP_OPTIMAL; ++a) {
                const int v = nv + as[a];
                const unsigned long long bc = __ballot(lx.cv == v && lane >= 1 && lane <= nc0 && ((lx.alive >> lane) & 1u));
                if (bc) lx.drop_col(__ffsll((long long)bc) - 1);
                int row = -1;
#pragma unroll
                for (int sl = SLOTS - 1; sl >= 0; --sl) {
                    const unsigned long long br = __ballot(lx.var[sl] == v && lx.kind[sl] == RK_INEQ);
                    if (br) row = __ffsll((long long)br) - 1 + 64 * sl;
                }
                if (row >= 0) r = lx.drive_to_zero(row);
                else if (!bc) retry = true;
            }
            pivots += lx.iters;
            if (!retry && r != LP_ITERLIMIT && lx.growth > GROWTH_SAFE) retry = true;
            if (!retry) {
                if (r == LP_OPTIMAL) st = singular ? ST_SINGULAR : ST_FEASIBLE;
                else if (r == LP_ITERLIMIT) st = ST_LP_LIMIT;
                else st = ST_INFEASIBLE;
            }
        }
        cyc_x += clock64() - t2;
        if (retry) { st = ST_RETRY; n_retry++; }
        if (lane == 0) status[c] = (uint8_t)st;
    }
    if (lane == 0) {
        atomicAdd(&ctr->cycles[2], (unsigned long long)cyc_x);
        atomicAdd(&ctr->pivots, pivots); atomicAdd(&ctr->xtheta_fallbacks, n_retry);
    }
}

}  // namespace mpc
