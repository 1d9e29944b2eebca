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
#include <type_traits>
#include "kernels.hpp"
#include "lp_reg.hpp"

namespace mpc {

// The theta LP of a candidate is a short run (a handful of pivots) from rows that were just built from the program data:
// nothing has accumulated, so the error of its final dictionary is bounded by eps * growth * pivots.  It is handed to the
// refactorising LDS engine only beyond 1e6 (error <= 1e-9, two decades below the feasibility tolerance) -- the LDS engine
// accepts a freshly refactorised run under the same reasoning.  Dictionaries that live across levels (k_x2's cache) and the
// long facet walks of k_region2 keep the 1e3 threshold of lp_engine.hpp.
constexpr double GROWTH_FRESH = 1e6;
#ifndef THETA_SCALE_MODE
#define THETA_SCALE_MODE 0
#endif
// power-of-two scale of a theta-space row whose largest coefficient has binary exponent ex
__device__ __forceinline__ double theta_row_scale(int ex) {
    if (THETA_SCALE_MODE == 1) return 1.0;
    if (THETA_SCALE_MODE == 2 && ex > 0) return 1.0;
    return ldexp(1.0, -ex);
}
constexpr int ST_RETRY = 7;       // numerically doubtful: re-solve with the LDS engine (k_verdict)
constexpr int ST_NEEDX = 8;       // theta stage could not show feasibility: (x,theta) LP needed (k_x2)
constexpr int ST_NEEDX_SING = 9;  // the same, and the KKT matrix was singular (a feasible outcome is ST_SINGULAR)


// hot read-only blocks of k_theta2, passed BY VALUE: the pointers are known to be global memory (global_load / s_load
// instead of flat_load) and cost no descriptor reload inside the candidate loop.  All blocks are zero padded to the
// kernel's compile-time NT so that the row build has no `t < n_t` guards:
//   UVp      n_c x (NT+1)      [A Q^-1 c + b | A Q^-1 H + F | 0]
//   tvp      NT*NT + 3*NT      tv_minv (row stride NT), tv_theta, then the bounding box of {A_t theta <= b_t}: lo, hi
//   tv_rows  n_tpre x (NT+1)   the non-tight rows of A_t at the theta vertex
struct ThetaArgs {
    const double *W, *UVp, *tvp, *tv_rows;
    int chunk;   // candidates taken from the work queue per atomic
    int cbuf_off;   // offset (in doubles, from the start of dynamic LDS) of the row-compaction scratch of k_theta2<.,2>; 0 = none
    const int32_t *n_dev;   // != nullptr: the number of work items is read from device memory (the level runs without host round trips)
    // k_kkt_thread works on the blocks with the program's ne equality rows eliminated (setup_mfma.hpp): Wr, UVrp (padded like UVp),
    // AATr; Me (ne x (n_t+1)), Ne (ne x n_c) give the equality rows' multipliers back; gE: ne Gram pivots of the equality rows, then
    // their ne diagonal entries.  Programs without equality rows: Wr == W, UVrp == UVp, AATr == the Gram matrix, ne == 0.
    const double *Wr, *UVrp, *AATr, *Me, *Ne, *gE;
    int ne;
    // k_theta2: at most wave_max wavefronts take work, and no more than one per wave_div work items (0: no limit).  The kernel is a
    // tail of few long LPs: measured on config 4 (round 4, MPC_TH_CAP) an item takes 29 us with one wavefront per SIMD, 48 us with two
    // and 130 us with four -- the launch was fastest with HALF (level 5: 15 k items) or a QUARTER (level 4: 5 k items) of its wave slots.
    int wave_div, wave_max;
    // k_kkt_thread: every side of the parameter set's bounding box is finite (mpc_create): the box screen takes the largest value of a row
    // over the box as h + sum_t max(a_t blo_t, a_t bhi_t) -- four instructions per parameter instead of nine, the same number bit for bit
    int box_finite;
    // k_kkt_thread (round 5): when set, the kernel lists its own output -- the candidates it leaves to k_theta2 (status ST_TODO) in
    // kt_list / *kt_n, those its box screen sends to the (x,theta) question (ST_NEEDX) in kx_list / *kx_n -- with one atomic per workgroup
    // and list, instead of a five-launch compaction of the status array per list behind it.  The lists are index-ordered within
    // a workgroup's piece only; they are work lists (every candidate's result is written by candidate index), so the order changes nothing.
    int32_t *kt_list, *kt_n, *kx_list, *kx_n;
    // k_theta2 (round 6): != nullptr: every candidate found optimal is appended to optq (position from *q_tail) at once, so that region
    // wavefronts of a launch that is ALREADY running can start on it while the rest of the theta stage is still being solved
    int32_t *optq; unsigned int *q_tail;
    int prio;   // > 0: the kernel's wavefronts raise their issue priority (s_setprio): the theta stage is the critical path beside the early region launch
};

// ------------------------------------------------------------------------------------------------------------------
// k_kkt_thread: the mode-0 KKT solve AND the box screen of the theta stage with ONE THREAD per candidate (K = cardinality
// and NT >= n_theta are compile-time, everything lives in VGPRs).
//
// A k x k Cholesky with k <= 8 has no work for 64 lanes; done by a wavefront it is ~45 dependent LDS phases.  Here every
// lane solves its own candidate with fully unrolled register code: same operations in the same order as chol_solve
// (kkt.hpp), so L is bit-identical to what kkt_solve leaves in LDS.
//   rank test   is_full_rank(A, as) (constraint_utilities.py:222): the Gram matrix A_as A_as' (gathered from AAT) is
//               eliminated; if the product of its relative pivots (= prod sin^2 of the angles between each row and the
//               span of the earlier ones) is > 1e-10 and every pivot > 1e-8 of its diagonal, the rows are independent
//               with a 1e6 margin over the exact test's 1e-11 threshold -> code 0/2.  Anything closer is left to the exact
//               complete-pivoting elimination of kkt_solve (code KK_UNDECIDED).
//   multipliers L = -S^-1 UV[as], S = W[as,as]
//   box screen  the theta-space rows of the inactive constraints, h - g theta >= 0 with [h | -g] = UV[ci] + sum_a
//               W[ci,as_a] L[a], are formed one after the other (the row index is wave-uniform: UV by scalar loads, W
//               gathers hit one 376-byte row) and tested against the bounding box of the parameter set exactly as
//               k_theta2 does: the first row that cannot hold anywhere in the box proves stage 1 of the theta LP
//               infeasible -> status ST_NEEDX, the candidate never reaches k_theta2.
//   output      code[c] in {0 ok, 2 singular, KK_UNDECIDED};  status[c] = ST_NEEDX (screened) or ST_TODO;
//               Lout[c*K*nr + i*nr + t] = [b_l | A_l] for the candidates that go on to k_theta2
#ifndef KKT_RU56
#define KKT_RU56 1   // rows per trip of k_kkt_thread's box screen at five and six active rows (1: round 5)
#endif
constexpr int KK_UNDECIDED = 255;
constexpr int KK_ILL = 64;        // solved, but the Schur system is ill-conditioned (CHOL_ILL_TOL, kkt.hpp)
constexpr int ST_TODO = 10;       // internal: waiting for k_theta2
constexpr int ST_RRETRY = 11;     // status[] of a candidate k_region2 gave up on (its slot carries ST_RETRY): re-solved by k_region
// SP > 1 (small levels, round 6): SP neighbouring lanes share a candidate.  Each repeats the factorisation and the multipliers (the same
// registers, the same values) and screens every SP-th group of inactive rows; the verdict is the OR of theirs, lane 0 of the group writes.
// The screen is a chain of dependent L2 round trips -- n_c / RU of them, 40 of the kernel's 50 us on a level of thirty candidates -- and a
// level that does not fill the chip has the lanes to cut the chain SP-fold: same rows, same arithmetic per row, the same verdict.
template <int K, int NT, int SP = 1>
MPC_GLOBAL void MPC_LB(256) k_kkt_thread(const DevProblem *__restrict__ Pg, const int32_t *__restrict__ cands, long long n,
                                                     uint8_t *__restrict__ code, double *__restrict__ Lout, uint8_t *__restrict__ status,
                                                     ThetaArgs ta, LevelCounters *__restrict__ ctr) {
    // K = the INEQUALITY rows of the active set; the program's ne equality rows lead every candidate (cardinality ne + K) and are
    // eliminated from the systems below (ThetaArgs: Wr, UVrp, AATr) -- with ne == 0 these are the original blocks.
    static_assert(SP >= 1 && SP <= 64 && (SP & (SP - 1)) == 0, "lanes per candidate: a power of two within a wavefront");
    const DevProblem &P = *Pg;
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long c = gid / SP;
    const int sub = (int)(gid & (SP - 1));
    const bool lead = sub == 0;
    int kind = 0;      // 1: left to the theta stage (ST_TODO), 2: sent to the (x,theta) question by the screen (ST_NEEDX)
  do {
    if (c >= n) break;
    constexpr int LS = NT + 1;
    const int nc = P.n_c, nt = P.n_t, nr = nt + 1, ne = ta.ne, kf = ne + K;
    int as[K];
#pragma unroll
    for (int i = 0; i < K; ++i) as[i] = cands[c * kf + ne + i];
    double S[K][K];
    // ---- Gram test ----------------------------------------------------------------------------------------------
    {
        double gmax = 0.0, g0[K];
        bool clear = true;
        for (int j = 0; j < ne; ++j) { gmax = fmax(gmax, ta.gE[ne + j]); clear = clear && (ta.gE[j] > 1e-8 * ta.gE[ne + j]); }
#pragma unroll
        for (int i = 0; i < K; ++i) {
#pragma unroll
            for (int j = 0; j <= i; ++j) S[i][j] = ta.AATr[as[i] * nc + as[j]];
            g0[i] = P.AAT[as[i] * nc + as[i]];
            gmax = fmax(gmax, g0[i]);
        }
        clear = clear && gmax > 0.0;
        double vol = 1.0;
        const double ginv = 1.0 / gmax;
        for (int j = 0; j < ne; ++j) vol *= ta.gE[j] * ginv;
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const double d = S[j][j];
            clear = clear && (d > 1e-8 * g0[j]);
            vol *= d * ginv;
            const double dinv = 1.0 / d;
#pragma unroll
            for (int i = j + 1; i < K; ++i) {
                const double f = S[i][j] * dinv;
#pragma unroll
                for (int cc = j + 1; cc <= i; ++cc) S[i][cc] = fma(-f, S[cc][j], S[i][cc]);
            }
        }
        clear = clear && (vol > 1e-10);
        if (!clear) { if (lead) { code[c] = (uint8_t)KK_UNDECIDED; status[c] = (uint8_t)ST_TODO; } kind = 1; break; }
    }
    // ---- S = Wr[as,as] = L L'  (chol_solve arithmetic; the pivots are those of the full Schur matrix behind its equality block) ----
    double diag0[K], invd[K];
#pragma unroll
    for (int i = 0; i < K; ++i) {
#pragma unroll
        for (int j = 0; j <= i; ++j) S[i][j] = ta.Wr[as[i] * nc + as[j]];
        diag0[i] = ta.W[as[i] * nc + as[i]];
    }
    bool ok = true, ill = false;
#pragma unroll
    for (int j = 0; j < K; ++j) {
        const double d = S[j][j];
        ok = ok && (d > RANK_TOL_CHOL * diag0[j]);
        ill = ill || !(d > CHOL_ILL_TOL * diag0[j]);
        const double l = sqrt(d), inv = 1.0 / l;
#pragma unroll
        for (int i = j + 1; i < K; ++i) S[i][j] = S[i][j] * inv;
        S[j][j] = l;
        invd[j] = 1.0 / l;
#pragma unroll
        for (int i = j + 1; i < K; ++i) {
#pragma unroll
            for (int cc = j + 1; cc <= i; ++cc) S[i][cc] = fma(-S[i][j], S[cc][j], S[i][cc]);
        }
    }
    if (!ok) { if (lead) { code[c] = 2; status[c] = (uint8_t)ST_TODO; } kind = 1; break; }
    // ---- multipliers of the inequality rows, all n_t + 1 right-hand sides (zero beyond n_t) -----------------------------------------
    double Lr[K][LS];
#pragma unroll
    for (int t = 0; t < LS; ++t) {
        double R[K];
#pragma unroll
        for (int i = 0; i < K; ++i) R[i] = -ta.UVrp[as[i] * LS + t];
#pragma unroll
        for (int j = 0; j < K; ++j) {
            R[j] = R[j] * invd[j];
#pragma unroll
            for (int i = j + 1; i < K; ++i) R[i] = fma(-S[i][j], R[j], R[i]);
        }
#pragma unroll
        for (int j = K - 1; j >= 0; --j) {
            R[j] = R[j] * invd[j];
#pragma unroll
            for (int i = 0; i < j; ++i) R[i] = fma(-S[j][i], R[j], R[i]);
        }
#pragma unroll
        for (int i = 0; i < K; ++i) Lr[i][t] = R[i];
    }
    // ---- box screen over the inactive rows (k_theta2's test, same arithmetic; the equality rows are never inactive) ---------------
    const double *blo = ta.tvp + NT * NT + NT, *bhi = blo + NT;
    bool fired = false;
    // Rows are taken RU at a time with all their W gathers issued before the first is used: one row per trip made the loop a chain of
    // n_c dependent L2 round trips -- 55 us for ONE candidate, the floor of this kernel on every level however small (round 4,
    // tools/timeline.sh).  Same rows, same order, same arithmetic per row.
    // (no early exit inside a group of rows: a group is ONE basic block, so the scheduler can issue all its loads first; from five rows of
    // active set on the kernel is bound by registers and throughput -- config 4's level of 10^6 candidates -- and keeps one row per trip)
    constexpr int RU = K <= 2 ? 8 : (K <= 4 ? 4 : (K <= 6 ? KKT_RU56 : 1));
    for (int c0 = ne + sub * RU; c0 < nc; c0 += RU * SP) {
      double wpre[RU][K];
#pragma unroll
      for (int u = 0; u < RU; ++u) {
        const double *Wrow = ta.Wr + (size_t)min(c0 + u, nc - 1) * nc;
#pragma unroll
        for (int a = 0; a < K; ++a) wpre[u][a] = Wrow[as[a]];
      }
#pragma unroll
      for (int u = 0; u < RU; ++u) {
        const int ci = min(c0 + u, nc - 1);
        bool active = c0 + u >= nc;      // rows beyond the last one repeat it and never fire
#pragma unroll
        for (int a = 0; a < K; ++a) active = active || (as[a] == ci);
        double acc[LS];
#pragma unroll
        for (int t = 0; t < LS; ++t) acc[t] = ta.UVrp[ci * LS + t];
#pragma unroll
        for (int a = 0; a < K; ++a) {
            const double w = wpre[u][a];
#pragma unroll
            for (int t = 0; t < LS; ++t) acc[t] = fma(w, Lr[a][t], acc[t]);
        }
        double h = acc[0], g[NT], mx = 0.0;
#pragma unroll
        for (int t = 0; t < NT; ++t) { g[t] = -acc[1 + t]; mx = fmax(mx, fabs(g[t])); }
        if (ta.box_finite) {
            // (round 6) -g theta over the box is largest at blo where g > 0 and at bhi where g < 0: max(a blo, a bhi) with a = -g picks that
            // product (rounded exactly as g blo / g bhi are, up to the sign); a row of all-zero coefficients keeps h alone
            double sm = h;
#pragma unroll
            for (int t = 0; t < NT; ++t) sm += fmax(acc[1 + t] * blo[t], acc[1 + t] * bhi[t]);
            const double smax = mx > ZERO_ROW_ATOL ? sm : h;
            if (!active && smax < -10 * TOL_FEAS) fired = true;
            continue;
        }
        if (!(mx > ZERO_ROW_ATOL)) {
#pragma unroll
            for (int t = 0; t < NT; ++t) g[t] = 0.0;
            mx = 0.0;
        }
        // k_theta2 scales the row by a power of two sc (theta_row_scale) and tests smax_scaled < -10 TOL sc.  A power-of-two scale commutes
        // with every rounding of the sum below, so the unscaled test is the same decision bit for bit -- without frexp / ldexp and NT + 2
        // multiplications per row (round 6)
        double smax = h;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const double term = g[t] > 0.0 ? g[t] * blo[t] : (g[t] < 0.0 ? g[t] * bhi[t] : 0.0);
            smax -= term;
        }
        if (!active && smax < -10 * TOL_FEAS) fired = true;   // the margin is meant in the row's own units (scaled: both sides times sc)
      }
      if (__all(fired)) break;
    }
    if (SP > 1) {     // (every lane of a candidate is here together: they took the same exits above)
        const unsigned long long fm = __ballot(fired);
        const int l0 = (int)(threadIdx.x & 63) & ~(SP - 1);
        fired = ((fm >> l0) & (SP >= 64 ? ~0ull : ((1ull << SP) - 1ull))) != 0ull;
    }
    if (fired) {
        if (lead) { code[c] = ill ? KK_ILL : 0; status[c] = (uint8_t)ST_NEEDX; }
        kind = 2;
        break;
    }
    if (!lead) { kind = 1; break; }
    double *out = Lout + (size_t)c * kf * nr;
    // multipliers of the equality rows:  lambda_E = -(Me + Ne[:, as] lambda_a)
    for (int i = 0; i < ne; ++i) {
        double v[LS];
#pragma unroll
        for (int t = 0; t < LS; ++t) v[t] = t < nr ? ta.Me[i * nr + t] : 0.0;
#pragma unroll
        for (int a = 0; a < K; ++a) {
            const double w = ta.Ne[(size_t)i * nc + as[a]];
#pragma unroll
            for (int t = 0; t < LS; ++t) v[t] = fma(w, Lr[a][t], v[t]);
        }
#pragma unroll
        for (int t = 0; t < LS; ++t) if (t < nr) out[i * nr + t] = -v[t];
    }
#pragma unroll
    for (int i = 0; i < K; ++i) {
#pragma unroll
        for (int t = 0; t < LS; ++t) if (t < nr) out[(ne + i) * nr + t] = Lr[i][t];
    }
    code[c] = ill ? KK_ILL : 0;
    status[c] = (uint8_t)ST_TODO;
    kind = 1;
  } while (false);
    if (!lead) kind = 0;
    // the kernel's own work lists (ThetaArgs::kt_list / kx_list): ONE atomic per workgroup and list -- per wavefront and exit point they cost
    // the last level of config 4 (15 k wavefronts, 98.5 % of the candidates on one list) 0.11 ms of atomics on one address
    if (ta.kt_list) {     // (uniform: every thread of the workgroup is here, whatever it decided)
        __shared__ int wcnt[2][4];
        __shared__ int bbase[2];
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const unsigned long long m1 = __ballot(kind == 1), m2 = __ballot(kind == 2);
        if (lane == 0) { wcnt[0][wave] = __popcll(m1); wcnt[1][wave] = __popcll(m2); }
        __syncthreads();
        if (threadIdx.x < 2) {
            const int tot = wcnt[threadIdx.x][0] + wcnt[threadIdx.x][1] + wcnt[threadIdx.x][2] + wcnt[threadIdx.x][3];
            bbase[threadIdx.x] = tot ? atomicAdd(threadIdx.x == 0 ? ta.kt_n : ta.kx_n, tot) : 0;
        }
        __syncthreads();
        if (kind) {
            const int li = kind - 1;
            int pos = bbase[li] + __popcll((li ? m2 : m1) & ((1ull << lane) - 1ull));
            for (int w = 0; w < wave; ++w) pos += wcnt[li][w];
            (li ? ta.kx_list : ta.kt_list)[pos] = (int)c;
        }
    }
}

// the KKT result of candidate c: from k_kkt_thread's output when it decided, else solved here
__device__ inline int kkt_fetch_or_solve(const DevProblem &P, int k, Smem &s, const uint8_t *kkcode, const double *Lin, size_t c,
                                         bool *ill = nullptr) {
    if (kkcode) {
        int code = kkcode[c];
        if (code != KK_UNDECIDED) {
            if (code == KK_ILL) { if (ill) *ill = true; code = 0; }
            if (code == 0) {
                const int cnt = k * (P.n_t + 1);
                const double *src = Lin + c * (size_t)cnt;
                for (int idx = lane_id(); idx < cnt; idx += 64) s.L[idx] = src[idx];
                wave_sync();
            }
            return code;
        }
    }
    return kkt_solve(P, k, s, ill);
}

#ifndef TH_WAVES_S2
#define TH_WAVES_S2 3   // two rows per lane, n_theta <= 4: 168 registers, no scratch (four wavefronts: 128 registers and 164 bytes); n_theta >= 8: two wavefronts, 231 registers
#endif
template <int NT, int SLOTS>
MPC_GLOBAL void MPC_LB(64, (SLOTS >= 2 ? (NT <= 4 ? TH_WAVES_S2 : 2) : (NT >= 8 ? 3 : 4))) k_theta2(const DevProblem *__restrict__ Pg, const int32_t *__restrict__ cands, long long n, int k,
                                                    uint8_t *__restrict__ status, LevelCounters *__restrict__ ctr,
                                                    const uint8_t *__restrict__ kkcode, const double *__restrict__ Lin, ThetaArgs ta,
                                                    const int32_t *__restrict__ list) {
    // list != nullptr: the candidates list[0..n) that k_kkt_thread's screen left open; else all candidates 0..n
    // the program descriptor stays in memory (scalar loads on demand) instead of ~90 live SGPRs
    const DevProblem &P = *Pg;
    // (a block beyond the number of work items has nothing to take from the queue: it leaves before touching the queue's counter --
    //  with device-resident lengths the launch is sized by a bound, and thousands of surplus blocks would otherwise serialise on
    //  that one address)
    if (ta.n_dev) {
        n = *ta.n_dev;
        // chunk <= 0: the host does not know n either; the rule it would have applied (candidates per queue atomic so that every
        // block of the persistent grid gets about eight turns, at most 16) is applied here
        long long active = gridDim.x;
        if (ta.wave_max > 0) active = min(active, (long long)ta.wave_max);
        if (ta.wave_div > 0) active = min(active, max(256ll, n / ta.wave_div));
        if (ta.chunk <= 0) ta.chunk = (int)max(1ll, min(16ll, n / (active * 8)));
        if ((long long)blockIdx.x >= active || (long long)blockIdx.x * ta.chunk >= n) return;
    }
    if (ta.prio >= 3) __builtin_amdgcn_s_setprio(3); else if (ta.prio == 2) __builtin_amdgcn_s_setprio(2); else if (ta.prio == 1) __builtin_amdgcn_s_setprio(1);
    extern __shared__ __attribute__((aligned(16))) double smem[];
    Smem s = carve(P, smem);
    constexpr int LS = NT + 1;
    const int lane = lane_id(), nt = P.n_t, nr = nt + 1, e = P.n_eq, nc = P.n_c, mode = P.kkt_mode, npre = P.n_tpre;
    // LDS: tv_minv | tv_theta | multipliers with padded row stride (zeros beyond n_t)
    double *tvm = s.T, *tvt = s.T + NT * NT, *blo = tvt + NT, *bhi = blo + NT, *Lp = s.T + NT * NT + 3 * NT;
    for (int idx = lane; idx < NT * NT + 3 * NT; idx += 64) s.T[idx] = ta.tvp[idx];
    wave_sync();
    unsigned long long pivots = 0, n_retry = 0;
    long long cyc_kkt = 0, cyc_theta = 0, cyc_rows = 0, cyc_s2 = 0;
    unsigned long long n_box1 = 0, n_box2 = 0;
    for (;;) {
        unsigned int c0 = 0;
        if (lane == 0) c0 = atomicAdd(&ctr->work_verdict, (unsigned)ta.chunk);
        c0 = (unsigned)__builtin_amdgcn_readfirstlane((int)c0);
        if (c0 >= n) break;
        const unsigned c1 = (unsigned)min((long long)c0 + ta.chunk, n);
      for (unsigned w = c0; w < c1; ++w) {
        const unsigned c = list ? (unsigned)list[w] : w;
        const int nin = load_active_set(P, cands + (size_t)c * k, k, s);
        int st = -1;
        const long long t0 = clock64();
        const int kk = kkt_fetch_or_solve(P, k, s, kkcode, Lin, c);
        const long long t1 = clock64();
        cyc_kkt += t1 - t0;
        bool singular = false, retry = false;
        if (kk == 1) st = ST_INFEASIBLE;
        else if (kk == 2) singular = true;
        else if (kk == 0) {
            // ---- theta-space two-stage LP, rows expressed at the vertex of {A_t theta <= b_t} ----------------------
            for (int idx = lane; idx < k * LS; idx += 64) {
                const int a = idx / LS, t = idx - a * LS;
                Lp[idx] = t <= nt ? s.L[a * nr + t] : 0.0;
            }
            wave_sync();
            RegLp<NT + 2, SLOTS, true> lp;
            const int nlam = k - e, m = nlam + nin + npre;
            lp.m = m; lp.iters = 0; lp.max_iter = 50 * (m + nt) + 100; lp.growth = 0.0;
            lp.alive = (nt >= 31 ? 0xfffffffeu : ((1u << (nt + 1)) - 2u));
            lp.cv = nt + m + lane - 1;
            bool box_lam = false, box_slack = false;
            bool live[SLOTS];   // rows the LP needs: present and not satisfied everywhere in the bounding box
#pragma unroll
            for (int sl = 0; sl < SLOTS; ++sl) live[sl] = (lane + 64 * sl) < (k - e) + nin + npre;
#pragma unroll
            for (int sl = 0; sl < SLOTS; ++sl) {
                const int i = lane + 64 * sl;
                double h = 0.0, g[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) g[t] = 0.0;
                lp.var[sl] = nt + i;
                lp.kind[sl] = i < nlam ? RK_PASSIVE : (i < m ? RK_INEQ : RK_DEAD);
                lp.w[sl] = 1.0; lp.winv[sl] = 1.0;
                bool pre = false;
                if (i < nlam) {
                    h = Lp[(e + i) * LS];
#pragma unroll
                    for (int t = 0; t < NT; ++t) g[t] = -Lp[(e + i) * LS + 1 + t];
                } else if (i < nlam + nin) {
                    const int ci = s.inact[i - nlam];
                    if (mode == 0) {
                        // row = UV[ci] + sum_a W[ci, as[a]] * L[a]   (all loads issued up front)
                        double acc[NT + 1], w[8];
                        const double *Wrow = ta.W + (size_t)ci * nc;
#pragma unroll
                        for (int a = 0; a < 8; ++a) { const int asv = s.as[a]; w[a] = Wrow[a < k ? asv : 0]; }
#pragma unroll
                        for (int t = 0; t <= NT; ++t) acc[t] = ta.UVp[ci * LS + t];
#pragma unroll
                        for (int a = 0; a < 8; ++a) {
                            if (a < k) {
#pragma unroll
                                for (int t = 0; t <= NT; ++t) acc[t] = fma(w[a], Lp[a * LS + t], acc[t]);
                            }
                        }
                        for (int a = 8; a < k; ++a) {
                            const double wa = Wrow[s.as[a]];
#pragma unroll
                            for (int t = 0; t <= NT; ++t) acc[t] = fma(wa, Lp[a * LS + t], acc[t]);
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
                    const double *row = ta.tv_rows + (size_t)(i - nlam - nin) * LS;
#pragma unroll
                    for (int t = 0; t <= NT; ++t) lp.t[sl][t] = row[t];
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
                    double sc = 1.0;
                    if (mx > 0.0) {
                        int ex;
                        (void)frexp(mx, &ex);
                        sc = theta_row_scale(ex);
                        lp.w[sl] = sc; lp.winv[sl] = 1.0 / sc;
                        h *= sc;
#pragma unroll
                        for (int t = 0; t < NT; ++t) g[t] *= sc;
                    }
                    // the largest slack h - g theta this row can have over the bounding box of the parameter polytope: if even
                    // that is negative the row alone makes the theta LP infeasible (margin 10 x the LP's tolerance)
                    if (SLOTS == 1 && ta.box_finite) {      // (two rows per lane: the second form tips k_theta2<4,2> into scratch under its register cap)
                        // (round 6, as k_kkt_thread) under a finite box the extremes of h - g theta are h + sum max / min (-g blo, -g bhi): the
                        // two products serve both bounds, the numbers are those of the select form below bit for bit
                        double smax = h, smin = h;
#pragma unroll
                        for (int t = 0; t < NT; ++t) {
                            const double p0 = -g[t] * blo[t], p1 = -g[t] * bhi[t];
                            smax += fmax(p0, p1);
                            if (SLOTS >= 2) smin += fmin(p0, p1);
                        }
                        if (i < m && smax < -10 * TOL_FEAS * sc) { if (i < nlam) box_lam = true; else box_slack = true; }
                        if (SLOTS >= 2 && smin > 10 * TOL_FEAS * sc * 10) live[sl] = false;
                    } else {
                        double smax = h;
#pragma unroll
                        for (int t = 0; t < NT; ++t) {
                            const double term = g[t] > 0.0 ? g[t] * blo[t] : (g[t] < 0.0 ? g[t] * bhi[t] : 0.0);
                            smax -= term;
                        }
                        if (i < m && smax < -10 * TOL_FEAS * sc) { if (i < nlam) box_lam = true; else box_slack = true; }
                        if (SLOTS >= 2) {
                            // ... and the smallest: a row whose slack stays above 1e-6 (row units) over the whole box can never
                            // limit a step of a walk that stays inside the parameter polytope
                            double smin = h;
#pragma unroll
                            for (int t = 0; t < NT; ++t) smin -= g[t] > 0.0 ? g[t] * bhi[t] : (g[t] < 0.0 ? g[t] * blo[t] : 0.0);
                            if (smin > 10 * TOL_FEAS * sc * 10) live[sl] = false;
                        }
                    }
                    // theta = theta_v - Minv sigma:  value at the vertex and coefficients of the tight-row slacks sigma
                    double b0 = h, cf[NT];
#pragma unroll
                    for (int t = 0; t < NT; ++t) b0 = fma(-g[t], tvt[t], b0);
#pragma unroll
                    for (int j = 0; j < NT; ++j) cf[j] = 0.0;
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
#pragma unroll
                        for (int j = 0; j < NT; ++j) cf[j] = fma(g[t], tvm[t * NT + j], cf[j]);
                    }
                    lp.t[sl][0] = b0;
#pragma unroll
                    for (int j = 0; j < NT; ++j) lp.t[sl][1 + j] = -cf[j];
                }
                lp.t[sl][NT + 1] = 0.0;
            }
            cyc_rows += clock64() - t1;
            const bool inf1 = __any(box_slack), inf2 = __any(box_lam);
            n_box1 += inf1; n_box2 += (!inf1 && inf2);
            // inf1: a stage-1 row cannot hold anywhere in the parameter set -> stage 1 infeasible, no LP needed
            long long t15 = 0;
            int lp_iters = 0;
            auto stages = [&](auto &q, auto slots_c) {
                constexpr int S = decltype(slots_c)::value;
                const int r1 = inf1 ? LP_INFEASIBLE : q.phase1();
                t15 = clock64();
                if (r1 == LP_ITERLIMIT) st = ST_LP_LIMIT;
                else if (q.growth > GROWTH_FRESH) retry = true;
                else if (r1 == LP_OPTIMAL && inf2) st = ST_FEASIBLE;   // feasible, and a multiplier row rules out optimality
                else if (r1 == LP_OPTIMAL) {
#pragma unroll
                    for (int sl = 0; sl < S; ++sl) if (q.kind[sl] == RK_PASSIVE) q.kind[sl] = RK_INEQ;
                    const int r2 = q.phase1();
                    if (r2 == LP_ITERLIMIT) st = ST_LP_LIMIT;
                    else if (q.growth > GROWTH_FRESH) retry = true;
                    else st = r2 == LP_OPTIMAL ? ST_OPT_PENDING : ST_FEASIBLE;
                }
                lp_iters = q.iters;
            };
            bool compacted = false;
            if constexpr (SLOTS == 2) {
                // Two tableau rows per lane cost twice the per-pivot work of one.  Rows that are satisfied over the whole
                // bounding box never take part in a ratio test, so when the others fit into 64 lanes (98 % of the open
                // candidates at config 3) they are packed, in their order, into a one-row-per-lane LP through LDS.
                const unsigned long long b0 = __ballot(live[0]), b1 = __ballot(live[1]);
                const int n_live = __popcll(b0) + __popcll(b1);
                if (ta.cbuf_off > 0 && n_live <= 64 && !inf1) {
                    constexpr int CS = NT + 4;
                    double *cb = smem + ta.cbuf_off;
                    int *cbi = reinterpret_cast<int *>(cb + 64 * CS);
                    const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
                    for (int sl = 0; sl < 2; ++sl) {
                        if (live[sl]) {
                            const int pos = sl == 0 ? __popcll(b0 & below) : __popcll(b0) + __popcll(b1 & below);
#pragma unroll
                            for (int j = 0; j < NT + 2; ++j) cb[pos * CS + j] = lp.t[sl][j];
                            cb[pos * CS + NT + 2] = lp.w[sl];
                            cb[pos * CS + NT + 3] = lp.winv[sl];
                            cbi[2 * pos] = lp.var[sl];
                            cbi[2 * pos + 1] = lp.kind[sl];
                        }
                    }
                    wave_sync();
                    RegLp<NT + 2, 1, true> lp1;
                    lp1.m = n_live; lp1.iters = 0; lp1.max_iter = lp.max_iter; lp1.growth = 0.0;
                    lp1.alive = lp.alive; lp1.cv = lp.cv;
                    const bool has = lane < n_live;
#pragma unroll
                    for (int j = 0; j < NT + 2; ++j) lp1.t[0][j] = has ? cb[lane * CS + j] : 0.0;
                    lp1.w[0] = has ? cb[lane * CS + NT + 2] : 1.0;
                    lp1.winv[0] = has ? cb[lane * CS + NT + 3] : 1.0;
                    lp1.var[0] = has ? cbi[2 * lane] : -1;
                    lp1.kind[0] = has ? cbi[2 * lane + 1] : RK_DEAD;
                    wave_sync();
                    stages(lp1, std::integral_constant<int, 1>{});
                    compacted = true;
                }
            }
            if (!compacted) stages(lp, std::integral_constant<int, SLOTS>{});
            pivots += lp_iters;
            cyc_s2 += clock64() - t15;
        }
        const long long t2 = clock64();
        cyc_theta += t2 - t1;
        if (st < 0 && !retry) st = singular ? ST_NEEDX_SING : ST_NEEDX;  // feasibility still open: (x,theta) LP, k_x2
        if (retry) { st = ST_RETRY; n_retry++; }
        if (lane == 0) {
            status[c] = (uint8_t)st;
            if (st == ST_OPT_PENDING && ta.optq) {
                // (the entry is written after its position has been counted: a consumer that claims the position waits for the entry)
                const unsigned int pos = atomicAdd(ta.q_tail, 1u);
                // (relaxed: a consumer needs nothing but the entry itself -- a release at agent scope is a write-back of the L2, and 6,238 of
                //  them made the theta kernel of config 4's last level 0.36 -> 0.50 ms; the queue is filled with -1 at the level's start)
                __hip_atomic_store(&ta.optq[pos], (int32_t)c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
      }
    }
    if (lane == 0) {
        atomicAdd(&ctr->cycles[0], (unsigned long long)cyc_kkt); atomicAdd(&ctr->cycles[1], (unsigned long long)cyc_theta);
        atomicAdd(&ctr->cycles[4], (unsigned long long)cyc_rows); atomicAdd(&ctr->cycles[5], (unsigned long long)cyc_s2);
        atomicAdd(&ctr->cycles[6], n_box1); atomicAdd(&ctr->cycles[7], n_box2); atomicAdd(&ctr->n_retry_theta, (unsigned)n_retry);
        atomicAdd(&ctr->pivots, pivots); atomicAdd(&ctr->xtheta_fallbacks, n_retry);
    }
}


// (x,theta) feasibility of the candidates list[0..n_list), in registers, no LDS.
//
// Dictionary cache (the 288 GB of HBM put to work): a candidate that is feasible leaves its final dictionary -- the
// program's pre-crashed vertex dictionary with the candidate's rows activated -- in HBM (dict_cur, slot = its index in the
// level).  A child (= parent + one larger row) whose parent has a slot loads that dictionary with coalesced column
// reads and activates ONE row instead of starting from D0 and activating all k.  The kernel then streams ~2 x 9 KB per
// candidate (config 4) and is bound by HBM bandwidth rather than by pivots.
// dict layout per slot: doubles [NXC][mr] (column j of row i at j*mr + i);
// ints var[mr] kind[mr] cv[NXC] alive growth_hi growth_lo pad | ineq[4] | pos bytes[n_c]   (round 5: the last two)
//   ineq  bit i = row i takes part in ratio tests (kind RK_INEQ)
//   pos   where the slack of program constraint c sits: 0..127 basic in that row, 128 + j nonbasic in column j; an entry is only
//         meaningful if the row / column it names really holds that variable (entries of absent variables are stale and fail that check)
__host__ __device__ inline int dict_ints_head(int mr, int NXC) { return 2 * mr + NXC + 4; }
__host__ __device__ inline long long dict_ints(int mr, int NXC, int n_c) { return dict_ints_head(mr, NXC) + 4 + (n_c + 3) / 4; }
struct DictCache {
    const int32_t *parent_slot;   // per candidate of this level (nullptr: no cache to read)
    const double *prev_d; const int32_t *prev_i;
    double *cur_d; int32_t *cur_i; uint8_t *stored;   // cur_d == nullptr: do not store
    long long stride_d, stride_i;
    int dict_only;   // 1: the candidates are already decided (theta stage); only their dictionary is wanted for the children
    // two optional list segments that are processed BEFORE `list` in the same launch, dictionary-only (the candidates the
    // theta stage found feasible / optimal: status untouched, they only leave a dictionary for their children)
    const int32_t *pre1, *pre2;
    int n_pre1, n_pre2;
    int chunk;       // work items per queue atomic (<= 64)
    // != nullptr: the three list lengths are read from device memory (the level runs without host round trips)
    const int32_t *n_list_dev, *n_pre1_dev, *n_pre2_dev;
    int max_blocks;  // > 0 (with device-resident lengths): only that many blocks of the launch work, the others leave (a member's share of a shared launch)
    int flag_retry;  // k_xq / k_xq_grouped: a candidate whose run meets a doubtful pivot gets ST_RETRY at once (k_x2 would only repeat the run to find the same)
    // k_x2 (round 6): a run that started from the program's own dictionary D0 (no cached parent) is a short fresh run -- k drives, nothing
    // accumulated before them -- and its DECISION stands up to this growth (GROWTH_FRESH, the theta LP's rule; 0: GROWTH_SAFE as for a cached
    // run).  Its dictionary is still only stored below GROWTH_SAFE: what the children inherit keeps the strict bound.
    double fresh_limit;
    int skip_below;  // k_xq with a device-resident list length: a list shorter than this is left to k_x2 (the kernel's comment)
    int second_max;  // k_x2: repeats from D0 of doubtful cached runs the level may spend (0: none; the kernel's comment)
};
// k_xq: the last level's quick (x,theta) test.  No dictionary is stored on the last level, so a candidate only needs a
// DECISION: up to XQ_ITERS simplex iterations from the parent's dictionary in product form (revised simplex with an eta
// file), reading only the vectors they touch -- the values, the new row, and per iteration one entering column and one
// pivot row (~0.3 KB each) -- instead of the whole tableau (~9 KB per candidate).  Exactly the decision of k_x2 when the run
// ends within those iterations (slack already nonbasic / already zero / no improving column / the new row leaves the
// basis); otherwise the candidate keeps its NEEDX status and goes to k_x2.  Few registers -> 8 waves per SIMD hide the
// dependent HBM round trips.
// 6 rather than 8: the eta file of 16 iterations does not fit 64 registers (292 bytes of scratch per lane at 8 waves per
// SIMD, 220 at 6); measured on config 4 / config 3, verdict stage: 8 waves 4.50 / 7.47 ms, 6 waves 4.39 / --, 4 waves 4.46 / 6.88
#ifndef XQ_WAVES
#define XQ_WAVES 6
#endif
#ifndef XQG_WAVES
#define XQG_WAVES 4   // k_xq_grouped (config 3): spill-free at 128 registers
#endif
#ifndef XQ_ITERS_N
#define XQ_ITERS_N 16
#endif
constexpr int XQ_ITERS = XQ_ITERS_N;
// The decision of one candidate of k_xq: `pd` / `pi` are its parent's dictionary record (doubles / ints), in HBM (k_xq) or
// staged in LDS (k_xq_grouped); v is the variable id of the candidate's new row.  Returns 1 feasible, 0 infeasible,
// -1 undecided (left to k_x2); piv_local counts the pivots.
template <int SLOTS, class PD, class PI>
__device__ __forceinline__ int xq_decide(PD pd, PI pi, int mr, int NXC, int ncol, int v, int lane, int &piv_local) {
    // NXC: column slots of the record layout; ncol <= NXC: the slots that are ever used (value column + D0 columns) -- only those
    // are stored and read
    int qvar[SLOTS], qkind[SLOTS], qhint[SLOTS];
    double qb[SLOTS];
#pragma unroll
    for (int sl = 0; sl < SLOTS; ++sl) {
        const int i = lane + 64 * sl;
        qvar[sl] = i < mr ? pi[i] : -1;
        const int kraw = i < mr ? pi[mr + i] : RK_DEAD;
        qkind[sl] = kraw & 0xff;
        qhint[sl] = kraw >> 8;      // the row's entering column at the parent's final dictionary (k_x2's store), 0 = none improves
        qb[sl] = i < mr ? pd[i] : 0.0;
    }
    const int qcv = lane < NXC + 3 ? pi[2 * mr + lane] : -1;
    const unsigned al = (unsigned)__builtin_amdgcn_readlane(qcv, NXC);
    const double growth0 = __hiloint2double(__builtin_amdgcn_readlane(qcv, NXC + 1), __builtin_amdgcn_readlane(qcv, NXC + 2));
    const unsigned long long bc = __ballot(qcv == v && lane >= 1 && lane < ncol && ((al >> lane) & 1u));
    int feas = -1;   // 1 feasible, 0 infeasible, -1 undecided
    piv_local = 0;
    if (bc) feas = 1;
    else {
        int row = -1;
#pragma unroll
        for (int sl = SLOTS - 1; sl >= 0; --sl) {
            const unsigned long long br = __ballot(qvar[sl] == v && qkind[sl] == RK_INEQ);
            if (br) row = __ffsll((long long)br) - 1 + 64 * sl;
        }
        if (row >= 0) {
            row = uni(row);
            // value of entry `r` of a vector whose element i lives in lane i & 63, slot i >> 6
            auto at = [&](const double (&vec)[SLOTS], int r) -> double {
                if (SLOTS == 1) return readlane_f64(vec[0], r & 63);
                return readlane_f64(r < 64 ? opaque(vec[0]) : opaque(vec[SLOTS - 1]), r & 63);   // (opaque: lp_reg.hpp -- the select must not become an indexed load)
            };
            // Up to XQ_ITERS simplex iterations in product form: the tableau is never formed.  Per pivot p the entering
            // column as it was (E[p], one entry per row) and the scaled pivot row (R[p], one entry per column) are kept;
            // a column or row needed later is read from the parent's dictionary and brought up to date through them,
            // with the operations pivot_core (lp_reg.hpp) would have applied to it -- bit for bit k_x2's arithmetic.
            double E[XQ_ITERS][SLOTS], R[XQ_ITERS];
            int rp[XQ_ITERS], qp[XQ_ITERS];
            double invp[XQ_ITERS];
            double xrow = (lane < ncol) ? pd[(size_t)lane * mr + row] : 0.0;   // the new row, entry j in lane j
            // first iteration: the column comes from the hint stored with the parent's record -- exactly the column the pricing of
            // `xrow` below would choose (same rule, same numbers) -- so the column is asked for together with the row, not after it
            const int q0 = SLOTS == 1 ? __builtin_amdgcn_readlane(qhint[0], row & 63) : __builtin_amdgcn_readlane(row < 64 ? opaque(qhint[0]) : opaque(qhint[SLOTS - 1]), row & 63);
            double growth = growth0;
#pragma unroll
            for (int it = 0; it < XQ_ITERS; ++it) {
                if (at(qb, row) <= TOL_FEAS) { feas = 1; break; }
                int q;
                if (it == 0) {
                    if (q0 <= 0) { feas = 0; break; }
                    q = q0;
                } else {
                    const double g = (lane >= 1 && lane < ncol) ? xrow : 0.0;
                    const double gm = dpp_wave_max(g > TOL_COST ? g : 0.0);
                    if (!(gm > TOL_COST)) { feas = 0; break; }
                    q = uni(__ffsll((long long)__ballot(g == gm && lane >= 1 && lane < ncol)) - 1);
                }
                // entering column at the current time
                double a[SLOTS], ratio[SLOTS];
#pragma unroll
                for (int sl = 0; sl < SLOTS; ++sl) { const int i = lane + 64 * sl; a[sl] = i < mr ? pd[(size_t)q * mr + i] : 0.0; }
#pragma unroll
                for (int p_ = 0; p_ < it; ++p_) {
                    if (q == qp[p_]) {
#pragma unroll
                        for (int sl = 0; sl < SLOTS; ++sl) a[sl] = (lane + 64 * sl == rp[p_]) ? 1.0 : 0.0;
                    }
                    const double x = at(a, rp[p_]) * invp[p_];
#pragma unroll
                    for (int sl = 0; sl < SLOTS; ++sl) a[sl] = (lane + 64 * sl == rp[p_]) ? x : fma(-E[p_][sl], x, a[sl]);
                }
                bool elig[SLOTS];
                float cmf = 0.0f;
                double tmax = INFINITY;
#pragma unroll
                for (int sl = 0; sl < SLOTS; ++sl) {
                    const int i = lane + 64 * sl;
                    const bool used = i < mr && qkind[sl] != RK_DEAD;
                    if (used) cmf = fmaxf(cmf, fabsf((float)a[sl]));
                    elig[sl] = used && qkind[sl] == RK_INEQ && a[sl] > TOL_PIV;
                    ratio[sl] = 0.0;
                    if (elig[sl]) {
                        const double b0 = fmax(qb[sl], 0.0), ia = fast_rcp(a[sl]);
                        ratio[sl] = b0 * ia;
                        tmax = fmin(tmax, (b0 + HARRIS_DELTA) * ia);
                    }
                }
                const float colmax = dpp_wave_max_f32(cmf);
                tmax = dpp_wave_min(tmax);
                if (tmax == INFINITY) break;   // unbounded direction: left to k_x2
                bool pass[SLOTS], mine = false;
#pragma unroll
                for (int sl = 0; sl < SLOTS; ++sl) { pass[sl] = elig[sl] && !(ratio[sl] > tmax); mine = mine || (lane + 64 * sl == row && pass[sl]); }
                int l = -1;
                double rpiv;
                if (__any(mine)) { l = row; rpiv = at(a, row); }
                else {
                    double am = 0.0;
#pragma unroll
                    for (int sl = 0; sl < SLOTS; ++sl) if (pass[sl]) am = fmax(am, a[sl]);
                    rpiv = dpp_wave_max(am);
#pragma unroll
                    for (int sl = SLOTS - 1; sl >= 0; --sl) {
                        const unsigned long long bl = __ballot(pass[sl] && a[sl] == rpiv);
                        if (bl) l = __ffsll((long long)bl) - 1 + 64 * sl;
                    }
                    if (l < 0) break;
                    l = uni(l);
                }
                const double inv = fast_rcp(rpiv);
                growth = fmax(growth, (double)(colmax * (float)inv));
                if (growth > GROWTH_SAFE) { feas = -2; break; }   // a doubtful pivot: k_x2 would repeat the run and flag it (-2: the caller may flag it at once)
                piv_local++;
                if (l == row) { feas = 1; break; }          // the new row's slack leaves the basis at zero
                if (it + 1 == XQ_ITERS) break;
                // pivot (l, q): remember it, update the values and the new row
                rp[it] = l; qp[it] = q; invp[it] = inv;
#pragma unroll
                for (int sl = 0; sl < SLOTS; ++sl) E[it][sl] = a[sl];
                const double xb = at(qb, l) * inv;
#pragma unroll
                for (int sl = 0; sl < SLOTS; ++sl) qb[sl] = (lane + 64 * sl == l) ? xb : fma(-a[sl], xb, qb[sl]);
                double rowl = (lane < ncol) ? pd[(size_t)lane * mr + l] : 0.0;   // pivot row l at the current time, entry j in lane j
#pragma unroll
                for (int p_ = 0; p_ < it; ++p_) {
                    if (l == rp[p_]) rowl = R[p_];
                    else {
                        const double xr = at(E[p_], l);
                        rowl = fma(-xr, R[p_], lane == qp[p_] ? 0.0 : rowl);
                    }
                }
                R[it] = (lane == q) ? inv : rowl * inv;
                const double xr = at(a, row);
                xrow = fma(-xr, R[it], lane == q ? 0.0 : xrow);
            }
        }
    }
    return feas;
}

#ifdef MPC_XQ_HIST
__device__ unsigned long long g_xq_hist[64];
__device__ unsigned long long g_xq_hist2[64];
#endif
template <int SLOTS>
MPC_GLOBAL void MPC_LB(64, XQ_WAVES) k_xq(const DevProblem *__restrict__ Pg, const int32_t *__restrict__ cands, int k,
                                               const int32_t *__restrict__ list, int n_list, uint8_t *__restrict__ status,
                                               LevelCounters *__restrict__ ctr, DictCache dc, int NXC) {
    const DevProblem &P = *Pg;
    const int lane = lane_id(), nv = P.n_x + P.n_t, mr = P.n_d0r;
    if (dc.n_list_dev) {
        n_list = *dc.n_list_dev;
        // Round 6: a SHORT list is left to k_x2 as it is.  What the one-thread pass leaves over are the candidates that need several
        // iterations, and here every iteration is a chain of dependent reads of the parent's record (sixteen of them: 0.37 ms for config 4's
        // 2.7 k candidates, the launch lasts as long as its longest item) -- k_x2 loads the record once and pivots in registers.  The product
        // form pays when the list is long enough for its smaller traffic to matter.
        if (n_list < dc.skip_below) return;
        const long long active = dc.max_blocks > 0 ? min((long long)gridDim.x, (long long)dc.max_blocks) : (long long)gridDim.x;
        if (dc.chunk <= 0) dc.chunk = (int)max(1ll, min(16ll, (long long)n_list / (active * 4)));
        if ((long long)blockIdx.x >= active || (long long)blockIdx.x * dc.chunk >= n_list) return;
    }
    unsigned long long pivots = 0, n_quick = 0, n_doubt = 0;
    for (;;) {
        unsigned int w0 = 0;
        if (lane == 0) w0 = atomicAdd(&ctr->work_q, (unsigned)dc.chunk);
        w0 = (unsigned)__builtin_amdgcn_readfirstlane((int)w0);
        if (w0 >= (unsigned)n_list) break;
        const int cnt = min(dc.chunk, n_list - (int)w0);
        int my_c = 0, my_ps = -1, my_st = 0, my_v = 0;
        if (lane < cnt) {
            my_c = list[w0 + lane];
            my_ps = dc.parent_slot[my_c];
            my_st = status[my_c];
            my_v = nv + cands[(size_t)my_c * k + (k - 1)];
        }
        for (int u = 0; u < cnt; ++u) {
            const int c = __builtin_amdgcn_readlane(my_c, u);
            const int ps = __builtin_amdgcn_readlane(my_ps, u);
            if (ps < 0) continue;
            const bool singular = __builtin_amdgcn_readlane(my_st, u) == ST_NEEDX_SING;
            const int v = __builtin_amdgcn_readlane(my_v, u);
            const int32_t *pi = dc.prev_i + (size_t)ps * dc.stride_i;
            const double *pd = dc.prev_d + (size_t)ps * dc.stride_d;
            int piv_local = 0;
            const int feas = xq_decide<SLOTS>(pd, pi, mr, NXC, P.n_d0c + 1, v, lane, piv_local);
#ifdef MPC_XQ_HIST
            if (lane == 0) atomicAdd(&g_xq_hist[(feas + 1) * 20 + min(piv_local, 19)], 1ull);   // debug build: outcome x ratio tests passed
            {
                // experiment: is there ANY column whose Harris test lets the new row leave at zero?  (brute force, debug build only)
                int row = -1;
                const int vi = lane < mr ? pi[lane] : -1, ki = lane < mr ? (pi[mr + lane] & 0xff) : RK_DEAD;
                const unsigned long long br = __ballot(vi == v && ki == RK_INEQ);
                if (br) row = __ffsll((long long)br) - 1;
                int any = 0;
                if (row >= 0) {
                    const double b = lane < mr ? pd[lane] : 0.0;
                    for (int q = 1; q < P.n_d0c + 1; ++q) {
                        const double a = lane < mr ? pd[(size_t)q * mr + lane] : 0.0;
                        const bool elig = ki == RK_INEQ && a > TOL_PIV;
                        double tm = INFINITY, ratio = 0.0;
                        if (elig) { const double b0 = fmax(b, 0.0), ia = fast_rcp(a); ratio = b0 * ia; tm = (b0 + HARRIS_DELTA) * ia; }
                        tm = dpp_wave_min(tm);
                        if (__any(lane == row && elig && !(ratio > tm))) any = 1;
                    }
                }
                if (lane == 0) atomicAdd(&g_xq_hist2[any * 20 + min(piv_local, 19)], 1ull);
            }
#endif
            if (feas >= 0) {
                n_quick++;
                pivots += piv_local;
                if (lane == 0) status[c] = (uint8_t)(feas ? (singular ? ST_SINGULAR : ST_FEASIBLE) : ST_INFEASIBLE);
            } else if (feas == -2 && dc.flag_retry) {
                n_doubt++;
                if (lane == 0) status[c] = (uint8_t)ST_RETRY;
            }
        }
    }
    if (lane == 0) {
        atomicAdd(&ctr->pivots, pivots); atomicAdd(&ctr->xq_pivots, pivots); atomicAdd(&ctr->x_cached, n_quick); atomicAdd(&ctr->xtheta_lps, n_quick);
        if (n_doubt) atomicAdd(&ctr->xtheta_fallbacks, n_doubt);
    }
}

// k_xq_thread (round 5): the part of xq_decide that ends within its FIRST ratio test, with ONE THREAD per candidate.
//
// Two thirds of config 4's last level end there (tools/xq_hist.py: 14.7 % before any ratio test -- the new row's slack is already
// nonbasic / already zero / has no improving column --, 52.3 % because the new row itself passes the Harris test of its hinted
// column and leaves the basis at zero).  None of that needs a wavefront: the test is a scan down ONE column of the parent's record
// with no dependence between candidates.  Lane = candidate; siblings are neighbours in the list, so the parent's values, kinds and
// variable ids are same-address loads and only the hinted column is the lane's own walk.  Same operations on the same numbers as
// xq_decide (min / max are exact, so the order of the scan does not matter): a candidate decided here has the verdict k_xq would
// have given it; everything else -- a pivot is needed, an unbounded direction, a doubtful pivot, no parent record -- keeps its
// NEEDX status and goes on to k_xq unchanged.  ~20 wave-instructions per row of the record for 64 candidates, against ~150-200
// per candidate in k_xq.
// the test against ONE parent record: 1 feasible, 0 infeasible, -1 open; *by_test: decided by the ratio test (one pivot in k_xq's count)
// *step (when the answer is "feasible"): how the parent's dictionary becomes the candidate's -- XS_DROP << 16 | column: the slack is a
// live nonbasic column, which is deleted;  XS_ZERO << 16 | row << 8: the slack is basic at zero, the row pivots on its largest entry;
// XS_PIVOT << 16 | row << 8 | column: ONE pivot, the row leaves through the Harris test of its hinted column.
constexpr int XS_DROP = 1, XS_ZERO = 2, XS_PIVOT = 3;
__device__ __forceinline__ int xq_first_test(const double *__restrict__ pd, const int32_t *__restrict__ pi, int mr, int ncol, int NXC, int nv, int v, bool *by_test,
                                             int *step = nullptr) {
    *by_test = false;
    const int32_t *pm = pi + dict_ints_head(mr, NXC);
    const int p = reinterpret_cast<const uint8_t *>(pm + 4)[v - nv];   // where the record says this slack sits (checked below)
    if (p >= 128) {
        // a live nonbasic column of the parent's dictionary: the slack is zero at the parent's vertex
        const int j = p - 128;
        if (j >= 1 && j < ncol && pi[2 * mr + j] == v && (((unsigned)pi[2 * mr + NXC] >> j) & 1u)) { if (step) *step = (XS_DROP << 16) | j; return 1; }
        return -1;
    }
    if (p >= mr || pi[p] != v) return -1;
    const int kraw = pi[mr + p];
    if ((kraw & 0xff) != RK_INEQ) return -1;
    const int row = p, q0 = kraw >> 8;
    const double brow = pd[row];
    if (brow <= TOL_FEAS) { if (step) *step = (XS_ZERO << 16) | (row << 8); return 1; }
    if (q0 <= 0) return 0;
    // Four rows per trip, each vector fetched with 16-byte loads (records are 8-byte aligned, the hardware takes unaligned
    // global accesses): a lane's column is ITS OWN walk -- 64 distinct lines per wave-level load -- and the kernel is bound
    // by the number of such requests, not by their bytes (tools/ubench/xq_thread_bench.hip: one 8-byte load per row 0.57 ms
    // for config 4's level, seven rows issued together 0.28, four rows in two 16-byte loads 0.19)
    struct __attribute__((packed, aligned(8))) D4 { double v[4]; };
    struct __attribute__((packed, aligned(4))) I4 { int v[4]; };
    const I4 mk = *reinterpret_cast<const I4 *>(pm);
    const double *col = pd + (size_t)q0 * mr;
    float cmf = 0.0f;
    double tmax = INFINITY, ratio_row = 0.0, a_row = 0.0;
    bool elig_row = false;
    for (int i0 = 0; i0 < mr; i0 += 4) {
        double a4[4], b4[4];
        if (i0 + 4 <= mr) {
            const D4 xa = *reinterpret_cast<const D4 *>(col + i0);
            const D4 xb = *reinterpret_cast<const D4 *>(pd + i0);
#pragma unroll
            for (int u = 0; u < 4; ++u) { a4[u] = xa.v[u]; b4[u] = xb.v[u]; }
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int i = min(i0 + u, mr - 1); a4[u] = col[i]; b4[u] = pd[i]; }
        }
        const unsigned mw = (unsigned)(i0 < 32 ? mk.v[0] : (i0 < 64 ? mk.v[1] : (i0 < 96 ? mk.v[2] : mk.v[3]))) >> (i0 & 31);   // (i0 is a multiple of four)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u;
            if (i < mr) {
                const double a = a4[u];
                // (rows of these records are RK_INEQ or RK_DEAD: k_x2 leaves no other kind behind, so "used" == "ratio-test row")
                const bool used = (mw >> u) & 1u;
                if (used) cmf = fmaxf(cmf, fabsf((float)a));
                const bool elig = used && a > TOL_PIV;
                if (elig) {
                    const double b0 = fmax(b4[u], 0.0), ia = fast_rcp(a);
                    const double ratio = b0 * ia;
                    tmax = fmin(tmax, (b0 + HARRIS_DELTA) * ia);
                    if (i == row) { elig_row = true; ratio_row = ratio; a_row = a; }
                }
            }
        }
    }
    // the new row is inside the Harris bound: it leaves the basis at zero (xq_decide: `mine`, l == row)
    if (tmax != INFINITY && elig_row && !(ratio_row > tmax)) {
        const double growth0 = __hiloint2double(pi[2 * mr + NXC + 1], pi[2 * mr + NXC + 2]);
        const double inv = fast_rcp(a_row);
        const double growth = fmax(growth0, (double)(cmf * (float)inv));
        if (!(growth > GROWTH_SAFE)) { *by_test = true; if (step) *step = (XS_PIVOT << 16) | (row << 8) | q0; return 1; }
    }
    return -1;
}

// Other parents (alt_tries > 0): a candidate {a < b < c < d < e} was generated from {a,b,c,d}, but {a,b,c,e}, {a,b,d,e} ... were
// candidates of the previous level too, and those that were feasible left a dictionary.  From ANY of them the same first test is a
// valid proof: a vertex of the face of that parent from which the missing row's slack reaches zero along one edge (feasible), or on
// whose whole face it cannot decrease (infeasible).  The previous frontier is in lexicographic order (children are written parent
// by parent, new row ascending), so a parent is found by binary search; prev_stored says whether it left a dictionary.
struct XqAlt {
    const int32_t *prev_frontier;   // [n_prev][k - 1]
    const uint8_t *prev_stored;     // [n_prev]
    int n_prev, tries;
};
// Plan mode (x1_list != nullptr): a level that KEEPS dictionaries.  Every candidate that needs one -- the open ones (`list`) and, before
// them, the ones the theta stage already decided (pre1, pre2: status untouched) -- is asked the same question, and the answer is kept
// as a plan: from which parent slot, by which single step (xq_first_test's *step).  Planned candidates are appended to x1_list (k_x1
// streams the parent's record through that one pivot), the others to rest[segment] (k_x2, the register simplex, as before).
struct XqPlan {
    int32_t *plan_slot, *plan_step;   // [n] by candidate
    int32_t *x1_list, *x1_n;
    int32_t *rest[3], *rest_n[3];
    const int32_t *pre1, *pre2;
    int n_pre1, n_pre2;
    const int32_t *n_pre1_dev, *n_pre2_dev;   // != nullptr: the lengths are read from device memory (the stage is queued before the host knows them)
};
// One wavefront = 64 candidates at a time, lanes always full: a candidate its generating parent leaves open goes into the wavefront's
// queue (LDS) with "try 1"; whenever 64 are queued (or the input has run out) the wavefront takes them up again, each lane with its
// own try number.  No global atomics, no compaction between the tries.
MPC_GLOBAL void MPC_LB(64) k_xq_thread(const DevProblem *__restrict__ Pg, const int32_t *__restrict__ cands, int k,
                                       const int32_t *__restrict__ list, int n_list, uint8_t *__restrict__ status,
                                       LevelCounters *__restrict__ ctr, DictCache dc, int NXC, XqAlt alt, XqPlan pl) {
    __shared__ int q_c[128], q_t[128];
    const DevProblem &P = *Pg;
    const int nv = P.n_x + P.n_t, mr = P.n_d0r, ncol = P.n_d0c + 1, lane = threadIdx.x, km = k - 1;
    if (dc.n_list_dev) n_list = *dc.n_list_dev;
    const bool plan = pl.x1_list != nullptr;
    if (pl.n_pre1_dev) pl.n_pre1 = *pl.n_pre1_dev;
    if (pl.n_pre2_dev) pl.n_pre2 = *pl.n_pre2_dev;
    const int n_pre = plan ? pl.n_pre1 + pl.n_pre2 : 0;
    const long long n_items = (long long)n_pre + n_list;
    const int max_try = (alt.tries > 0 && alt.n_prev > 0 && k >= 2) ? min(alt.tries, km - P.n_eq) : 0;   // tries 1..max_try leave out position km - t
    unsigned int n_dec = 0, n_piv = 0, n_alt = 0;   // per lane: candidates decided / decided by a ratio test / decided from another parent
    int qn = 0;                                      // queued items (wave-uniform)
    const unsigned long long below = (1ull << lane) - 1ull;
    // appends the candidates of the lanes with `want` to a list in global memory (one atomic per wavefront and call)
    auto append = [&](bool want, int c, int32_t *dst, int32_t *cnt) {
        const unsigned long long m = __ballot(want);
        if (m == 0ull) return;
        int base = 0;
        if (lane == 0) base = atomicAdd(cnt, __popcll(m));
        base = __builtin_amdgcn_readfirstlane(base);
        if (want) dst[base + __popcll(m & below)] = c;
    };
    // an item ends here: feas 1 / 0 / -1 (nothing found), seg 0 / 1: decided by the theta stage (dictionary only), 2: open
    auto finish = [&](bool live, int c, int seg, int feas, bool by_test, bool from_alt, int slot, int step) {
        bool planned = false, left = false;
        if (live) {
            if (seg < 2 && feas == 0) feas = -1;          // (cannot be: the theta stage found it feasible) -- left to k_x2
            if (feas >= 0 && seg == 2) {
                const bool singular = status[c] == ST_NEEDX_SING;
                status[c] = (uint8_t)(feas ? (singular ? ST_SINGULAR : ST_FEASIBLE) : ST_INFEASIBLE);
            }
            if (feas >= 0) { n_dec++; if (by_test) n_piv++; if (from_alt) n_alt++; }
            planned = plan && feas == 1;
            left = plan && feas < 0;
            if (planned) { pl.plan_slot[c] = slot; pl.plan_step[c] = step; if (dc.stored) dc.stored[c] = 1; }   // (k_x1 WILL store it: the level's end need not wait for that, round 6)
        }
        if (plan) {
            append(planned, c, pl.x1_list, pl.x1_n);
            append(left && seg == 0, c, pl.rest[0], pl.rest_n[0]);
            append(left && seg == 1, c, pl.rest[1], pl.rest_n[1]);
            append(left && seg == 2, c, pl.rest[2], pl.rest_n[2]);
        }
    };
    auto push = [&](bool want, int c, int t) {   // wave-uniform call
        const unsigned long long m = __ballot(want);
        if (want) { const int pos = qn + __popcll(m & below); q_c[pos] = c; q_t[pos] = t; }
        qn += __popcll(m);
        wave_sync();
    };
    long long w0 = (long long)blockIdx.x * 64;
    for (;;) {
        const bool have_input = w0 < n_items;
        if (have_input) {
            const long long w = w0 + lane;
            w0 += (long long)gridDim.x * 64;
            int c = -1, feas = -1, seg = 2, slot = -1, step = 0;
            bool by_test = false, open = false, live = false;
            if (w < n_items) {
                live = true;
                if (w < n_pre) { seg = w < pl.n_pre1 ? 0 : 1; c = seg == 0 ? pl.pre1[w] : pl.pre2[w - pl.n_pre1]; }
                else c = list[w - n_pre];
                slot = dc.parent_slot[c];
                if (slot >= 0) {
                    feas = xq_first_test(dc.prev_d + (size_t)slot * dc.stride_d, dc.prev_i + (size_t)slot * dc.stride_i, mr, ncol, NXC, nv, nv + cands[(size_t)c * k + km], &by_test, &step);
                    open = feas < 0 && max_try > 0;
                }
            }
            finish(live && !open && (plan || feas >= 0), c, seg, feas, by_test, false, slot, step);
            if (max_try > 0) push(open, c, 1 | (seg << 8));
        }
        // the queue is taken up when it holds a full wavefront, or when nothing new will come
        while (qn >= 64 || (!have_input && qn > 0)) {
            const int take = min(qn, 64);
            qn -= take;
            int c = -1, t = 0, seg = 2;
            if (lane < take) { c = q_c[qn + lane]; t = q_t[qn + lane] & 0xff; seg = q_t[qn + lane] >> 8; }
            wave_sync();
            bool again = false, by_test = false;
            int feas = -1, found = -1, step = 0;
            if (c >= 0) {
                const int32_t *as = cands + (size_t)c * k;
                const int drop = km - t;   // position of the member this parent does not have (t = 1: the second largest)
                // as[] without as[drop] in the previous frontier (lexicographic order)
                // the wanted set, and the comparison with a row of the previous frontier: the row's members are fetched together (one
                // round trip per step of the search instead of one per member -- the look-up is a chain of dependent reads)
                int want[8];
#pragma unroll
                for (int a = 0; a < 8; ++a) want[a] = a < km ? as[a < drop ? a : a + 1] : 0;
                auto cmp_row = [&](int i) -> int {   // sign of (row i) - (the wanted set)
                    const int32_t *row = alt.prev_frontier + (size_t)i * km;
                    int cmp = 0;
                    if (km <= 8) {
                        int rv[8];
#pragma unroll
                        for (int a = 0; a < 8; ++a) rv[a] = a < km ? row[a] : 0;
#pragma unroll
                        for (int a = 0; a < 8; ++a) if (cmp == 0 && a < km) cmp = (rv[a] > want[a]) - (rv[a] < want[a]);
                        return cmp;
                    }
                    for (int a = 0; a < km && cmp == 0; ++a) {
                        const int qa = as[a < drop ? a : a + 1];
                        cmp = (row[a] > qa) - (row[a] < qa);
                    }
                    return cmp;
                };
                int lo = 0, hi = alt.n_prev - 1;
                if (t == 1) {
                    // {.., e} without the second largest member is a SIBLING of the generating parent {.., d}: same prefix, larger last
                    // member, i.e. a few rows further on in the same block of the previous frontier -- walked, not searched
                    const int ps = dc.parent_slot[c];
                    lo = ps + 1;
                    for (int stp = 0; stp < 24 && lo <= hi; ++stp, ++lo) {
                        const int cmp = cmp_row(lo);
                        if (cmp >= 0) { if (cmp == 0) found = lo; hi = lo - 1; break; }
                    }
                }
                while (lo <= hi) {
                    const int mid = (lo + hi) >> 1;
                    const int cmp = cmp_row(mid);
                    if (cmp == 0) { found = mid; break; }
                    if (cmp < 0) lo = mid + 1; else hi = mid - 1;
                }
                if (found >= 0 && alt.prev_stored[found])
                    feas = xq_first_test(dc.prev_d + (size_t)found * dc.stride_d, dc.prev_i + (size_t)found * dc.stride_i, mr, ncol, NXC, nv, nv + as[drop], &by_test, &step);
                if (seg < 2 && feas == 0) feas = -1;
                again = feas < 0 && t < max_try;
            }
            finish(c >= 0 && !again && (plan || feas >= 0), c, seg, feas, by_test, true, found, step);
            push(again, c, (t + 1) | (seg << 8));
        }
        if (!have_input) break;
    }
    // per wavefront one set of atomics (as the wavefront kernel)
    for (int off = 32; off > 0; off >>= 1) { n_dec += __shfl_xor(n_dec, off); n_piv += __shfl_xor(n_piv, off); n_alt += __shfl_xor(n_alt, off); }
    if (lane == 0 && n_dec) {
        if (!plan) { atomicAdd(&ctr->pivots, (unsigned long long)n_piv); atomicAdd(&ctr->xq_pivots, (unsigned long long)n_piv); }   // (plan mode: k_x1 counts the pivots it executes)
        atomicAdd(&ctr->x_cached, (unsigned long long)n_dec); atomicAdd(&ctr->xtheta_lps, (unsigned long long)n_dec);
        atomicAdd(&ctr->xq_thread, n_dec);
        if (n_alt) atomicAdd(&ctr->pad_xq, n_alt);
    }
}

// k_xq_grouped: the same decisions with the parent's dictionary read ONCE per parent.  The candidates of a level are
// ordered by parent (children of one parent are consecutive in the frontier, hence in the list), so the list falls into
// groups with a common parent record.  One workgroup of four wavefronts takes a group: all 256 threads copy the record
// (16 KB at config 4) from HBM into LDS with coalesced loads, then the wavefronts share the group's candidates and run
// xq_decide against LDS -- the dependent round trips of the product-form iterations cost an LDS access instead of an HBM
// access, and HBM sees one streaming read per parent instead of scattered 64-byte sectors per candidate and iteration.
// gstart[g] = position in `list` of the first candidate of group g (k_group_flags + scan + scatter), *n_groups_p groups.
MPC_GLOBAL void k_group_flags(const int32_t *__restrict__ list, int n_list, const int32_t *__restrict__ parent_slot,
                              int32_t *__restrict__ flag) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_list) flag[i] = (i == 0 || parent_slot[list[i]] != parent_slot[list[i - 1]]) ? 1 : 0;
}
template <int SLOTS>
MPC_GLOBAL void MPC_LB(256, XQG_WAVES) k_xq_grouped(const DevProblem *__restrict__ Pg, const int32_t *__restrict__ cands, int k,
                                                          const int32_t *__restrict__ list, int n_list, uint8_t *__restrict__ status,
                                                          LevelCounters *__restrict__ ctr, DictCache dc, int NXC,
                                                          const int32_t *__restrict__ gstart, const int32_t *__restrict__ n_groups_p) {
    extern __shared__ __attribute__((aligned(16))) double xq_smem[];
    __shared__ unsigned int item_s;
    const DevProblem &P = *Pg;
    double *sd = xq_smem;
    int32_t *si = reinterpret_cast<int32_t *>(xq_smem + dc.stride_d);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nv = P.n_x + P.n_t, mr = P.n_d0r;
    const unsigned n_groups = (unsigned)*n_groups_p;
    const int nd = min(((P.n_d0c + 1) * mr + 1) & ~1, (int)dc.stride_d), ni = (int)dc.stride_i;   // the used columns only (even count: 16-byte copies)
    unsigned long long pivots = 0, n_quick = 0, n_doubt = 0;
    for (;;) {
        if (tid == 0) item_s = atomicAdd(&ctr->work_q, 1u);
        __syncthreads();
        const unsigned g = item_s;
        if (g >= n_groups) break;
        const int first = gstart[g], last = g + 1 < n_groups ? gstart[g + 1] : n_list;
        const int ps = dc.parent_slot[list[first]];
        if (ps >= 0) {
            const double *pd = dc.prev_d + (size_t)ps * dc.stride_d;
            const int32_t *pi = dc.prev_i + (size_t)ps * dc.stride_i;
            if ((nd & 1) == 0) {
                const double2 *p2 = reinterpret_cast<const double2 *>(pd);
                double2 *s2 = reinterpret_cast<double2 *>(sd);
                for (int i = tid; i < nd / 2; i += 256) s2[i] = p2[i];
            } else {
                for (int i = tid; i < nd; i += 256) sd[i] = pd[i];
            }
            for (int i = tid; i < ni; i += 256) si[i] = pi[i];
        }
        __syncthreads();
        if (ps >= 0) {
            for (int u = first + wave; u < last; u += 4) {
                const int c = list[u];
                const bool singular = status[c] == ST_NEEDX_SING;
                const int v = nv + cands[(size_t)c * k + (k - 1)];
                int piv_local = 0;
                const int feas = xq_decide<SLOTS>(sd, si, mr, NXC, P.n_d0c + 1, v, lane, piv_local);
                if (feas >= 0) {
                    n_quick++;
                    pivots += piv_local;
                    if (lane == 0) status[c] = (uint8_t)(feas ? (singular ? ST_SINGULAR : ST_FEASIBLE) : ST_INFEASIBLE);
                } else if (feas == -2 && dc.flag_retry) {
                    n_doubt++;
                    if (lane == 0) status[c] = (uint8_t)ST_RETRY;
                }
            }
        }
        __syncthreads();
    }
    if (lane == 0) {
        atomicAdd(&ctr->pivots, pivots); atomicAdd(&ctr->xq_pivots, pivots); atomicAdd(&ctr->x_cached, n_quick); atomicAdd(&ctr->xtheta_lps, n_quick);
        if (n_doubt) atomicAdd(&ctr->xtheta_fallbacks, n_doubt);
    }
}

#ifndef R2_CERT
#define R2_CERT 1
#endif
#ifndef R2W
#define R2W 2
#endif
#ifndef X2_WAVES
#define X2_WAVES 2   // 32-column tableau: 252 registers, no scratch (three wavefronts per SIMD: 168 registers and 132 bytes; round 4: level 4 of config 4 1.06 -> 0.98 ms)
#endif
#ifndef X2_WAVES_16
#define X2_WAVES_16 3   // the 16-column instantiations (168 registers, no scratch at 3 waves per SIMD; 4 waves: 64 sub-programs per launch 175.0 -> 173.4 ms on the device, config 2 unchanged -- not worth the spills)
#endif
template <int NXC, int SLOTS>
MPC_GLOBAL void MPC_LB(64, (NXC * SLOTS >= 64 ? 2 : (NXC * SLOTS >= 32 ? X2_WAVES : X2_WAVES_16))) k_x2(const DevProblem *__restrict__ Pg, const int32_t *__restrict__ cands, int k,
                                              const int32_t *__restrict__ list, int n_list, uint8_t *__restrict__ status,
                                              LevelCounters *__restrict__ ctr, DictCache dc) {
    const DevProblem &P = *Pg;
    const int lane = lane_id(), nv = P.n_x + P.n_t, e = P.n_eq;
    const int mr = P.n_d0r, nc0 = P.n_d0c;
    const double *d0T = P.d0T;
    const int32_t *d0_rows = P.d0_rows, *d0_cols = P.d0_cols;
    const bool all_dict_only = dc.dict_only != 0;
    const bool last_only = !all_dict_only && dc.cur_d == nullptr;   // nothing is stored: only the verdict is wanted
    if (dc.n_list_dev) n_list = *dc.n_list_dev;
    if (dc.n_pre1_dev) dc.n_pre1 = *dc.n_pre1_dev;
    if (dc.n_pre2_dev) dc.n_pre2 = *dc.n_pre2_dev;
    const int n_pre = dc.n_pre1 + dc.n_pre2, n_items = n_pre + n_list;
    if (dc.n_list_dev) {
        const long long active = dc.max_blocks > 0 ? min((long long)gridDim.x, (long long)dc.max_blocks) : (long long)gridDim.x;
        if (dc.chunk <= 0) dc.chunk = (int)max(1ll, min(16ll, (long long)n_items / (active * 8)));
        if ((long long)blockIdx.x >= active || (long long)blockIdx.x * dc.chunk >= n_items) return;
    }
    unsigned long long pivots = 0, n_retry = 0, n_cached = 0;
    long long cyc_x = 0;
    int sink = 0;
    for (;;) {
        unsigned int w0 = 0;
        if (lane == 0) w0 = atomicAdd(&ctr->work_x, (unsigned)dc.chunk);
        w0 = (unsigned)__builtin_amdgcn_readfirstlane((int)w0);
        if (w0 >= (unsigned)n_items) break;
        const int cnt = min(dc.chunk, n_items - (int)w0);
        // the chunk's bookkeeping in one batch: lane u holds candidate, parent slot and status of item u
        int my_c = 0, my_ps = -1, my_st = 0;
        if (lane < cnt) {
            const int idx = (int)w0 + lane;
            my_c = idx < dc.n_pre1 ? dc.pre1[idx] : (idx < n_pre ? dc.pre2[idx - dc.n_pre1] : list[idx - n_pre]);
            my_ps = dc.parent_slot ? dc.parent_slot[my_c] : -1;
            my_st = status[my_c];
        }
      for (int u = 0; u < cnt; ++u) {
        const bool dict_only = all_dict_only || (int)w0 + u < n_pre;
        const int c = __builtin_amdgcn_readlane(my_c, u);
        int ps = __builtin_amdgcn_readlane(my_ps, u);
        const int32_t *as = cands + (size_t)c * k;
        const bool singular = __builtin_amdgcn_readlane(my_st, u) == ST_NEEDX_SING;
        bool retry = false;
        int st = -1;
        const long long t2 = clock64();
        // Round 6: a run from a cached record that turns out doubtful (the growth it inherited or met) is repeated HERE from the program's own
        // dictionary -- k drives from D0, nothing inherited -- while the level's budget of such repeats lasts (DictCache::second_max, counted in
        // LevelCounters::x_second); only what is doubtful again goes to the LDS engine.  A handful of doubtful candidates used to cost a level
        // a launch of the LDS engine that lasts as long as its longest LP (config 3's last level: 217 candidates, 0.62 ms) plus the repeated
        // end of the level; a level with 10^5 of them keeps the LDS engine (its throughput on them is the better one).
        for (int attempt = 0;; ++attempt) {
            retry = false; st = -1;
        {
            // the next item's dictionary is pulled towards the L2 while this one is solved (one 128-byte line per lane)
            if (attempt == 0 && u + 1 < cnt) {
                const int psn = __builtin_amdgcn_readlane(my_ps, u + 1);
                if (psn >= 0) {
                    const int32_t *nx = reinterpret_cast<const int32_t *>(dc.prev_d + (size_t)psn * dc.stride_d);
                    const long long words = (long long)(nc0 + 1) * mr * 2;
                    if ((long long)lane * 32 < words) sink ^= nx[lane * 32];
                    if ((long long)(lane + 64) * 32 < words) sink ^= nx[(lane + 64) * 32];
                }
            }
            // ---- (x,theta) feasibility from the pre-crashed vertex dictionary ------------------------------------------
            RegLp<NXC, SLOTS> lx;
            lx.m = mr; lx.iters = 0; lx.max_iter = 50 * (mr + nc0) + 100; lx.growth = 0.0;
            // one load path for both sources (two paths make the register allocator keep two tableaux): the parent's
            // dictionary from the cache (only the child's own, last, row is new) or the program's D0 (all rows new)
            const bool cached = ps >= 0;
            const double *src_d = cached ? dc.prev_d + (size_t)ps * dc.stride_d : d0T;
            const int32_t *src_i = cached ? dc.prev_i + (size_t)ps * dc.stride_i : d0_rows;
            const int jmax = nc0, voff = cached ? 0 : nv;   // column slots beyond nc0 are never used: neither stored nor read
            const int first = cached ? k - 1 : e;   // first active row that still has to be switched on
            {
                // cv and, behind it in the cached record, alive / growth: one load, fields taken out of their lanes
                int cvv = -1;
                if (cached) { if (lane < NXC + 3) cvv = src_i[2 * mr + lane]; }
                else if (lane >= 1 && lane <= nc0) cvv = nv + d0_cols[lane - 1];
                lx.alive = nc0 >= 31 ? 0xfffffffeu : ((1u << (nc0 + 1)) - 2u);
                if (cached) {
                    lx.alive = (unsigned)__builtin_amdgcn_readlane(cvv, NXC);
                    lx.growth = __hiloint2double(__builtin_amdgcn_readlane(cvv, NXC + 1), __builtin_amdgcn_readlane(cvv, NXC + 2));
                    n_cached++;
                }
                lx.cv = lane < NXC ? cvv : -1;
            }
#pragma unroll
            for (int sl = 0; sl < SLOTS; ++sl) {
                const int i = lane + 64 * sl;
                lx.var[sl] = i < mr ? voff + src_i[i] : -1;
                lx.kind[sl] = i < mr ? (cached ? (src_i[mr + i] & 0xff) : RK_INEQ) : RK_DEAD;   // (bits 8.. of a stored kind: the row's entering-column hint, below)
#pragma unroll
                for (int j = 0; j < NXC; ++j) lx.t[sl][j] = (i < mr && j <= jmax) ? src_d[(size_t)j * mr + i] : 0.0;
            }
            // every active row is switched on at the feasible vertex: a nonbasic slack is simply fixed at zero (column
            // deleted); a basic one is driven to zero by a primal simplex run that keeps all other rows feasible, so no
            // phase 1 is needed afterwards.  "Its minimum is positive" <=> the candidate is infeasible.
            int r = LP_OPTIMAL;
            for (int a = first; a < k && r == LP_OPTIMAL; ++a) {
                const int v = nv + as[a];
                const unsigned long long bc = __ballot(lx.cv == v && lane >= 1 && lane < NXC && ((lx.alive >> lane) & 1u));
                if (bc) lx.drop_col(__ffsll((long long)bc) - 1);
                int row = -1;
#pragma unroll
                for (int sl = SLOTS - 1; sl >= 0; --sl) {
                    const unsigned long long br = __ballot(lx.var[sl] == v && lx.kind[sl] == RK_INEQ);
                    if (br) row = __ffsll((long long)br) - 1 + 64 * sl;
                }
                if (row >= 0) r = lx.drive_to_zero(row, last_only && a + 1 == k);   // last level, last row: the decision is enough
                else if (!bc) retry = true;
            }
            if (!retry && r == LP_OPTIMAL && !(lx.growth > GROWTH_SAFE) && dc.cur_d) {
                // leave the dictionary for the children
                double *od = dc.cur_d + (size_t)c * dc.stride_d;
                int32_t *oi = dc.cur_i + (size_t)c * dc.stride_i;
#pragma unroll
                for (int sl = 0; sl < SLOTS; ++sl) {
                    const int i = lane + 64 * sl;
                    if (i < mr) {
                        // Entering-column hint for the children's quick test (k_xq, round 4): the first iteration of a child whose new row
                        // is THIS row prices this row -- largest coefficient above the cost tolerance, lowest column among equals.  Every
                        // lane knows that of its own row now, for nothing (the row is in registers); stored in bits 8.. of the kind word,
                        // it saves the child one dependent read of the record (the row itself) before its first column can be asked for.
                        int qhint = 0;
                        double gbest = TOL_COST;
#pragma unroll
                        for (int j = 1; j < NXC; ++j) { const double v = lx.t[sl][j]; if (j <= nc0 && v > gbest) { gbest = v; qhint = j; } }
                        oi[i] = lx.var[sl];
                        oi[mr + i] = lx.kind[sl] | (qhint << 8);
#pragma unroll
                        for (int j = 0; j < NXC; ++j) if (j <= nc0) od[(size_t)j * mr + i] = lx.t[sl][j];
                    }
                }
                if (lane < NXC) oi[2 * mr + lane] = lx.cv;
                {
                    // for the children's one-thread tests (k_xq_thread): which rows are ratio-test rows, and where each program constraint's slack sits
                    int32_t *om = oi + dict_ints_head(mr, NXC);
                    uint8_t *opos = reinterpret_cast<uint8_t *>(om + 4);
                    const int ncp = P.n_c;
#pragma unroll
                    for (int sl = 0; sl < 2; ++sl) {
                        const unsigned long long bm = sl < SLOTS ? __ballot(lane + 64 * sl < mr && lx.kind[sl < SLOTS ? sl : 0] == RK_INEQ) : 0ull;
                        if (lane == 0) { om[2 * sl] = (int)(unsigned)bm; om[2 * sl + 1] = (int)(unsigned)(bm >> 32); }
                    }
#pragma unroll
                    for (int sl = 0; sl < SLOTS; ++sl) {
                        const int i = lane + 64 * sl, cidx = lx.var[sl] - nv;
                        if (i < mr && cidx >= 0 && cidx < ncp) opos[cidx] = (uint8_t)i;
                    }
                    { const int cidx = lx.cv - nv; if (lane >= 1 && lane <= nc0 && ((lx.alive >> lane) & 1u) && cidx >= 0 && cidx < ncp) opos[cidx] = (uint8_t)(128 + lane); }
                }
                if (lane == 0) {
                    oi[2 * mr + NXC] = (int)lx.alive;
                    oi[2 * mr + NXC + 1] = __double2hiint(lx.growth);
                    oi[2 * mr + NXC + 2] = __double2loint(lx.growth);
                    dc.stored[c] = 1;
                }
            }
            pivots += lx.iters;
            if (!retry && r != LP_ITERLIMIT && lx.growth > ((!cached && dc.fresh_limit > 0.0) ? dc.fresh_limit : GROWTH_SAFE)) retry = true;
            if (!retry) {
                if (r == LP_OPTIMAL) st = singular ? ST_SINGULAR : ST_FEASIBLE;
                else if (r == LP_ITERLIMIT) st = ST_LP_LIMIT;
                else st = ST_INFEASIBLE;
            }
        }
            if (!(retry && ps >= 0 && attempt == 0 && dc.second_max > 0)) break;
            unsigned int taken = 0;
            if (lane == 0) taken = atomicAdd(&ctr->x_second, 1u);
            if ((unsigned)__builtin_amdgcn_readfirstlane((int)taken) >= (unsigned)dc.second_max) break;
            ps = -1;
        }
        cyc_x += clock64() - t2;
        if (retry) { st = ST_RETRY; n_retry++; }
        if (lane == 0 && !dict_only) status[c] = (uint8_t)st;
      }
    }
    if (sink == 0x5a5a5a5a) n_cached++;   // keeps the prefetch loads alive; practically never true
    if (lane == 0) {
        atomicAdd(&ctr->cycles[2], (unsigned long long)cyc_x);
        atomicAdd(&ctr->pivots, pivots); atomicAdd(&ctr->xtheta_fallbacks, n_retry); atomicAdd(&ctr->x_cached, n_cached);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// k_x1 (round 5): the dictionary of a candidate whose plan (k_xq_thread, plan mode) is ONE step from a parent's record -- the step
// k_x2 would take from that record, executed as a STREAM: the pivot column and the pivot row's scale are the only state, every other
// column is read, updated with one fma per entry and written back.  No tableau in registers (a dozen VGPRs instead of 250: eight
// wavefronts per SIMD where k_x2<32,.> has two), no pricing, no ratio test (the plan is the outcome of both), one or two tableau rows
// per lane alike.  Same operations on the same numbers as RegLp::pivot_core / drop_col and k_x2's store: the record is bit for bit
// the one k_x2 leaves when it starts from that parent (tests/test_gpu_deep.py, MPC_X1=1 against MPC_X1=0).
//   XS_DROP   the new row's slack is a live nonbasic column: the column is deleted (zeroed), nothing else changes
//   XS_ZERO   it is basic at zero: the row pivots on its largest entry (RegLp::best_col) and that column is deleted; no entry: the row is dead
//   XS_PIVOT  it leaves the basis through the Harris test of column q: pivot (row, q), growth monitor, column q deleted
#ifndef X1_WAVES
#define X1_WAVES 7   // wavefronts per SIMD of k_x1 (72 registers: the next item's header rides along with the current item's columns, round 6)
#endif
template <int SLOTS>
MPC_GLOBAL void MPC_LB(64, X1_WAVES) k_x1(const DevProblem *__restrict__ Pg, const int32_t *__restrict__ x1_list, const int32_t *__restrict__ x1_n,
                                    LevelCounters *__restrict__ ctr, DictCache dc, int NXC, const int32_t *__restrict__ plan_slot,
                                    const int32_t *__restrict__ plan_step) {
    const DevProblem &P = *Pg;
    const int lane = lane_id(), nv = P.n_x + P.n_t, mr = P.n_d0r, nc0 = P.n_d0c;
    const int n = *x1_n;
    unsigned long long pivots = 0;
    auto at = [&](const double (&vec)[SLOTS], int r) -> double {
        if (SLOTS == 1) return readlane_f64(vec[0], r & 63);
        return readlane_f64(r < 64 ? opaque(vec[0]) : opaque(vec[SLOTS - 1]), r & 63);
    };
    auto at_i = [&](const int (&vec)[SLOTS], int r) -> int {
        if (SLOTS == 1) return __builtin_amdgcn_readlane(vec[0], r & 63);
        return __builtin_amdgcn_readlane(r < 64 ? opaque(vec[0]) : opaque(vec[SLOTS - 1]), r & 63);
    };
    // Round 6: an item's HEADER -- list entry, plan (parent slot, step), the record's integer part -- is requested while the item before
    // it is still streaming its columns: an item used to start with three dependent round trips (list -> plan -> integers / pivot
    // column) before its first column group could be asked for, a third of its time at config 4's level 4.  Deleted columns of the
    // parent's record (known from the integer part: `alive`) are not read at all -- they are zero by construction -- only written.
    struct Head { int c, ps, step; int var[SLOTS], kindw[SLOTS], cvv; };
    auto head_plan = [&](int w, Head &hd) {      // first hop: which candidate, which parent, which step (wave-uniform)
        hd.c = x1_list[w];
        hd.ps = plan_slot[hd.c]; hd.step = plan_step[hd.c];
    };
    auto head_ints = [&](Head &hd) {             // second hop: the integer part of the parent's record
        const int32_t *pi = dc.prev_i + (size_t)hd.ps * dc.stride_i;
#pragma unroll
        for (int sl = 0; sl < SLOTS; ++sl) {
            const int i = lane + 64 * sl;
            hd.var[sl] = i < mr ? pi[i] : -1;
            hd.kindw[sl] = i < mr ? pi[mr + i] : RK_DEAD;
        }
        hd.cvv = lane < NXC + 3 ? pi[2 * mr + lane] : -1;
    };
    const int w0 = blockIdx.x, G = gridDim.x;
    Head cur{}, nxt{};
    if (w0 < n) { head_plan(w0, cur); head_ints(cur); }
    for (int w = w0; w < n; w += G) {
        const bool more = w + G < n;
        if (more) head_plan(w + G, nxt);
        const int c = cur.c, ps = cur.ps, step = cur.step;
        const int type = step >> 16, r = (step >> 8) & 0xff;
        int q = step & 0xff;
        const double *pd = dc.prev_d + (size_t)ps * dc.stride_d;
        double *od = dc.cur_d + (size_t)c * dc.stride_d;
        int32_t *oi = dc.cur_i + (size_t)c * dc.stride_i;
        int var[SLOTS], kind[SLOTS];
#pragma unroll
        for (int sl = 0; sl < SLOTS; ++sl) { var[sl] = cur.var[sl]; kind[sl] = (lane + 64 * sl) < mr ? (cur.kindw[sl] & 0xff) : RK_DEAD; }
        int cvv = cur.cvv;
        unsigned alive = (unsigned)__builtin_amdgcn_readlane(cvv, NXC);
        const unsigned alive_in = alive | 1u;     // columns of the parent's record that hold data (column 0: the values)
        double growth = __hiloint2double(__builtin_amdgcn_readlane(cvv, NXC + 1), __builtin_amdgcn_readlane(cvv, NXC + 2));
        bool pivot = type == XS_PIVOT;
        if (type == XS_ZERO) {
            // RegLp::best_col: the largest |entry| of the row above the pivot tolerance, lowest column among equals (deleted columns are zero)
            const double e = (lane >= 1 && lane <= nc0) ? fabs(pd[(size_t)lane * mr + r]) : 0.0;
            const double em = dpp_wave_max(e);
            if (em > TOL_PIV) { q = uni(__ffsll((long long)__ballot(e == em && lane >= 1 && lane <= nc0)) - 1); pivot = true; }
            else {
#pragma unroll
                for (int sl = 0; sl < SLOTS; ++sl) if (lane + 64 * sl == r) kind[sl] = RK_DEAD;
            }
        }
        double f[SLOTS], fz[SLOTS];
        double inv = 0.0;
#pragma unroll
        for (int sl = 0; sl < SLOTS; ++sl) { f[sl] = 0.0; fz[sl] = 0.0; }
        if (pivot) {
            float cmf = 0.0f;
#pragma unroll
            for (int sl = 0; sl < SLOTS; ++sl) {
                const int i = lane + 64 * sl;
                f[sl] = i < mr ? pd[(size_t)q * mr + i] : 0.0;
                if (i < mr && kind[sl] != RK_DEAD) cmf = fmaxf(cmf, fabsf((float)f[sl]));
                fz[sl] = i == r ? 0.0 : f[sl];
            }
            inv = fast_rcp(at(f, r));
            if (type == XS_PIVOT) growth = fmax(growth, (double)(dpp_wave_max_f32(cmf) * (float)inv));   // (the ratio-test pivot is monitored, the pivot on a zero row is not: RegLp::primal / pivot)
            pivots++;
        }
        const int qdel = (pivot || type == XS_DROP) ? q : -1;   // the column that is deleted
        int qhint[SLOTS];
        double gbest[SLOTS];
#pragma unroll
        for (int sl = 0; sl < SLOTS; ++sl) { qhint[sl] = 0; gbest[sl] = TOL_COST; }
        // columns in groups of CB: all loads of a group are issued before the first is used (the stream is bound by how many bytes are in
        // flight, one column at a time left the kernel at a third of the memory rate)
        // ... and the loads of the NEXT group are issued before the stores of this one: loads and stores share one in-order counter
        // on this hardware, so a load that follows a store in program order waits for that store to reach memory -- with the groups
        // one after the other an item took four round trips of load AND store latency (config 4, level 4: 0.63 ms for 138 k records)
        constexpr int CB = SLOTS == 1 ? 8 : 4;
        double tn[CB][SLOTS];
        auto load_group = [&](int j0) {
#pragma unroll
            for (int u = 0; u < CB; ++u) {
                const int j = min(j0 + u, nc0);
                const bool data = (alive_in >> j) & 1u;      // (uniform) a deleted column is zero in every row: not read
#pragma unroll
                for (int sl = 0; sl < SLOTS; ++sl) { const int i = lane + 64 * sl; tn[u][sl] = (data && i < mr) ? pd[(size_t)j * mr + i] : 0.0; }
            }
        };
        load_group(0);
        for (int j0 = 0; j0 <= nc0; j0 += CB) {
            double t[CB][SLOTS];
#pragma unroll
            for (int u = 0; u < CB; ++u) {
#pragma unroll
                for (int sl = 0; sl < SLOTS; ++sl) t[u][sl] = tn[u][sl];
            }
            if (j0 + CB <= nc0) load_group(j0 + CB);
            else if (more) head_ints(nxt);      // behind the item's last column group: the next item's integers (its plan has long arrived)
#pragma unroll
            for (int u = 0; u < CB; ++u) {
                const int j = j0 + u;
                if (j <= nc0) {
                    if (j == qdel) {
#pragma unroll
                        for (int sl = 0; sl < SLOTS; ++sl) t[u][sl] = 0.0;
                    } else if (pivot) {
                        const double trj = at(t[u], r) * inv;    // the pivot row's entry, scaled in its own lane by pivot_core
#pragma unroll
                        for (int sl = 0; sl < SLOTS; ++sl) t[u][sl] = fma(-fz[sl], trj, (lane + 64 * sl == r) ? trj : t[u][sl]);   // (the pivot lane too: fz = 0, as pivot_core)
                    }
#pragma unroll
                    for (int sl = 0; sl < SLOTS; ++sl) {
                        const int i = lane + 64 * sl;
                        if (i < mr) {
                            od[(size_t)j * mr + i] = t[u][sl];
                            if (j >= 1 && t[u][sl] > gbest[sl]) { gbest[sl] = t[u][sl]; qhint[sl] = j; }
                        }
                    }
                }
            }
        }
        if (pivot) {
            // the entering variable becomes basic in row r, the row's old variable sits in the (deleted) column
            const int vq = __builtin_amdgcn_readlane(cvv, q), vr = at_i(var, r);
#pragma unroll
            for (int sl = 0; sl < SLOTS; ++sl) if (lane + 64 * sl == r) { var[sl] = vq; kind[sl] = RK_INEQ; }
            if (lane == q) cvv = vr;
        }
        if (qdel >= 0) alive &= ~(1u << qdel);
#pragma unroll
        for (int sl = 0; sl < SLOTS; ++sl) {
            const int i = lane + 64 * sl;
            if (i < mr) { oi[i] = var[sl]; oi[mr + i] = kind[sl] | (qhint[sl] << 8); }
        }
        if (lane < NXC) oi[2 * mr + lane] = cvv;
        {
            int32_t *om = oi + dict_ints_head(mr, NXC);
            uint8_t *opos = reinterpret_cast<uint8_t *>(om + 4);
            const int ncp = P.n_c;
#pragma unroll
            for (int sl = 0; sl < 2; ++sl) {
                const unsigned long long bm = sl < SLOTS ? __ballot(lane + 64 * sl < mr && kind[sl < SLOTS ? sl : 0] == RK_INEQ) : 0ull;
                if (lane == 0) { om[2 * sl] = (int)(unsigned)bm; om[2 * sl + 1] = (int)(unsigned)(bm >> 32); }
            }
#pragma unroll
            for (int sl = 0; sl < SLOTS; ++sl) {
                const int i = lane + 64 * sl, cidx = var[sl] - nv;
                if (i < mr && cidx >= 0 && cidx < ncp) opos[cidx] = (uint8_t)i;
            }
            { const int cidx = cvv - nv; if (lane >= 1 && lane <= nc0 && ((alive >> lane) & 1u) && cidx >= 0 && cidx < ncp) opos[cidx] = (uint8_t)(128 + lane); }
        }
        if (lane == 0) {
            oi[2 * mr + NXC] = (int)alive;
            oi[2 * mr + NXC + 1] = __double2hiint(growth);
            oi[2 * mr + NXC + 2] = __double2loint(growth);
            dc.stored[c] = 1;
        }
        cur = nxt;
    }
    if (ctr && lane == 0 && pivots) atomicAdd(&ctr->pivots, pivots);
}

// ------------------------------------------------------------------------------------------------------------------
// k_region2: gen_cr_from_active_set (utils/mpqp_utils.py:89-195) on the register engine, compact output.
//
//   rows      built exactly as the reference builds them (multiplier rows, inactive rows through the x-law, A_t rows),
//             zero rows dropped, unit L2 norm; master copy E|f in LDS, dictionary rows at the theta vertex in VGPRs
//   full dim  Chebyshev LP  max r : E theta + ||E_i|| r <= f  (phase 1 + phase 2 on a cost row)
//   facets    the reference solves one LP per row ("is {E theta <= f, E_i theta = f_i} non-empty?", lines 143-178).
//             Here ONE feasible dictionary is kept and walked: r is driven back to 0, then for every row its slack is
//             either already zero at the current vertex (kept), or minimised by a short primal run from the current
//             vertex (kept iff the minimum is <= 1e-7); every slack that is zero at a visited vertex is marked kept on
//             the way, so most rows need no run of their own.  Same decision, 5-10x fewer pivots.
//   output    compact: fixed "head" per optimal candidate (x-law, multipliers, index sets) + E|f rows appended to a row
//             pool (atomic bump allocation), exact duplicate rows removed (mpqp_utils.py:191).
// head_d (stride fd): A_x[n_x*n_t] b_x[n_x] A_l[k*n_t] b_l[k]
// head_i (stride fi): status cand nE n_om n_la n_re e_off 0 | active[k] | omega[n_tc] | lambda[k] | reg_idx[n_c-k] | reg_con[n_c-k]
// pool row: f, E[0..n_t)
typedef const __attribute__((address_space(1))) double *gdp;   // a pointer the compiler may treat as global memory (members of DevProblem are generic: flat loads that also wait for LDS)
constexpr double BOX_REDUNDANT_MARGIN = 1e-6;   // unit-norm row units; ten times the LP feasibility tolerance

// Streaming of the region records to the host WHILE the kernel runs (single-GPU solve loop).  head_d / head_i / epool then
// point into page-locked host memory (zero-copy stores over PCIe, no device-to-host copy afterwards) and the slots are
// grouped in chunks of 2^shift: the wavefront that completes the last slot of a chunk raises flags[chunk] in host memory,
// after a system-scope release, so the host can build that chunk's region objects while later chunks are still computed.
// count == nullptr: no streaming (records stay in device memory).
struct RegionStream {
    unsigned int *count;   // [n_chunks] slots completed per chunk (device memory, zeroed before the launch)
    int32_t *flags;        // [n_chunks] host-mapped: 1 = every slot of the chunk is complete and visible to the host
    int shift;             // log2(chunk size)
    int n_slots;           // == n_opt
    // != nullptr: n_opt is read from device memory, and with W == 0 the number of wavefronts per candidate is chosen in the
    // kernel by the host's rule (the largest power of two <= w_max with n_opt * W <= w_cap)
    const int32_t *n_opt_dev;
    int w_cap, w_max;
    int max_blocks;        // > 0 (with n_opt_dev): only that many blocks of the launch work (a member's share of a shared launch)
    // Round 6, the queue form (large last level): opt_list is the queue k_theta2 fills (LevelCounters::q_tail; an entry is -1 until written),
    // ctr->work_r2 its head; one wavefront per candidate, slot = queue position.  early == 1: the launch runs BESIDE the theta kernel --
    // a wavefront claims a position and sleeps until its entry arrives or q_closed is up; early == 2: the drain launch behind the theta
    // kernel (the queue is closed when it starts: it never waits).
    int early, spin_max, q_cap;   // q_cap: entries of the queue array
};

template <int NT, int SLOTS>
MPC_GLOBAL void MPC_LB(64, (SLOTS >= 2 ? 2 : R2W)) k_region2(
    const DevProblem *__restrict__ Pg, const int32_t *__restrict__ cands, int k, const int32_t *__restrict__ opt_list, int n_opt,
    uint8_t *__restrict__ status, double *__restrict__ head_d, int32_t *__restrict__ head_i, int fd, int fi,
    double *__restrict__ epool, LevelCounters *__restrict__ ctr, const uint8_t *__restrict__ kkcode, const double *__restrict__ Lin,
    int W, uint8_t *__restrict__ kept_g, int ldk, unsigned int *__restrict__ done_g, const double *__restrict__ box, RegionStream rs) {
    // W > 1 (few optimal candidates, idle CUs): W wavefronts share one candidate.  Each builds the same dictionary and runs
    // the same Chebyshev LP (deterministic, identical), then tests only the rows it owns (row % W == part).  The flags go
    // to kept_g; the wavefront that finishes last (done_g counter) merges them and writes the record.
    const DevProblem &P = *Pg;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    Smem s = carve(P, smem);
    const int lane = lane_id(), nt = P.n_t, nr = nt + 1, nx = P.n_x, nc = P.n_c, ntc = P.n_tc, e = P.n_eq;
    constexpr int NC = NT + 3;      // value | n_t sigma columns | r | one spare slot for x0
    constexpr int ID_SIGMA = 4096, ID_R = 8192;
    unsigned long long pivots = 0;
    long long cyc = 0, rc_rows = 0, rc_cheb = 0, rc_facet = 0, rc_tot = 0, rc_refac = 0, rc_fpiv = 0, rc_box = 0;
    // the kernel's own duration on the constant-rate wall clock (first wavefront in, last wavefront out): the kernel may run on
    // a side stream under other kernels, where event markers are timestamped when the busy command processor reaches them
    if (rs.n_opt_dev) {
        n_opt = *rs.n_opt_dev;
        if (W == 0) { W = rs.w_max > 0 ? rs.w_max : 1; while (W > 1 && (long long)n_opt * W > rs.w_cap) W >>= 1; }
        if ((long long)blockIdx.x >= (long long)n_opt * W || (rs.max_blocks > 0 && (int)blockIdx.x >= rs.max_blocks)) return;   // surplus block of a launch sized by a bound
    }
    if (lane == 0) atomicMax(&ctr->r2_not_t0, ~(unsigned long long)wall_clock64());
    // parameter vertex (inverse of its tight rows, the vertex itself) and the bounding box of the parameter set: staged once per wavefront
    double *tvm = s.T, *tvt = s.T + NT * NT, *blo = tvt + NT, *bhi = blo + NT;
    { gdp src = (gdp)P.tvp; for (int idx = lane; idx < NT * NT + 3 * NT; idx += 64) s.T[idx] = src[idx]; }
    wave_sync();
    if (rs.early) W = 1;
    for (;;) {
        unsigned int item = 0;
        int c_q = -1;
        if (rs.early) {
            // Claim the next POSITION of the queue, then wait for its entry: every waiting wavefront polls its own address (a first form
            // in which all of them polled head / tail / closed -- three words of one L2 channel -- slowed the theta kernel beside it
            // 2.5x and the thread pass 5x).  The early launch sleeps ~5 us between looks and asks every eighth look whether the queue
            // has been closed (then an empty entry means "beyond the tail": leave); the drain launch starts behind the theta kernel,
            // where every entry below the tail is visible: an empty one ends the wavefront at once.
            if (lane == 0) {
                item = atomicAdd(&ctr->work_r2, 1u);
                if (item >= (unsigned)rs.q_cap) item = 0xffffffffu;
                else {
                    const int32_t *qe = opt_list + item;
                    for (int looks = 0;; ++looks) {
                        c_q = __hip_atomic_load(qe, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (-1 = not written)
                        if (c_q >= 0 || rs.early == 2) break;
                        if ((looks & 7) == 7 && __hip_atomic_load(&ctr->q_closed, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) {
                            c_q = __hip_atomic_load(qe, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                            break;
                        }
                        if (looks >= rs.spin_max) { atomicAdd(&ctr->q_fault, 1u); break; }   // (a second of waiting: the level fails instead of hanging)
                        __builtin_amdgcn_s_sleep(127); __builtin_amdgcn_s_sleep(127);
                    }
                    if (c_q < 0) item = 0xffffffffu;
                    else {
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                        if (rs.early == 1) atomicAdd(&ctr->q_early, 1u);
                    }
                }
            }
            item = (unsigned)__builtin_amdgcn_readfirstlane((int)item);
            c_q = __builtin_amdgcn_readfirstlane(c_q);
            if (item == 0xffffffffu) break;
        } else {
            if (lane == 0) item = atomicAdd(&ctr->work_r2, 1u);
            item = (unsigned)__builtin_amdgcn_readfirstlane((int)item);
            if (item >= (unsigned)n_opt * (unsigned)W) break;
        }
        const unsigned int w = item / (unsigned)W;
        const int part = (int)(item - w * (unsigned)W);
        auto own = [&](int row) -> bool { return W == 1 || row % W == part; };
        bool is_last = true;
        const long long t0 = clock64();
        const int c = rs.early ? c_q : opt_list[w];
        // Round 6: the chain of dependent memory round trips in front of the first pivot is as short as the data allows -- (1) the
        // candidate's index, (2) its members + KKT code + multipliers together, (3) the rows of A' / b / F / A_t behind the complement
        // list, each kind of row's loads issued before anything waits; the parameter vertex and the bounding box come from LDS.
        // Same operations in the same order on every entry as before (x_law_schur, the A X products, the sigma substitution).
        bool ill = false;
        int kk = -1;
        if (kkcode) {
            int code = kkcode[c];
            if (code != KK_UNDECIDED) {
                if (code == KK_ILL) { ill = true; code = 0; }
                if (code == 0) {
                    const int cnt = k * nr;
                    gdp src = (gdp)Lin + (size_t)c * cnt;
                    for (int idx = lane; idx < cnt; idx += 64) s.L[idx] = src[idx];
                }
                kk = code;
            }
        }
        const int nin = load_active_set(P, cands + (size_t)c * k, k, s);
        double *hd = head_d + (size_t)w * fd;
        int32_t *hi = head_i + (size_t)w * fi;
        for (int i = lane; i < fi; i += 64) hi[i] = i < 8 ? 0 : -1;
        int st = ST_REGION;
        if (kk < 0) kk = kkt_solve(P, k, s, &ill);
        if (kk != 0) st = kk == 1 ? ST_INFEASIBLE : ST_SINGULAR;
        int nE = 0, n_om = 0, n_la = 0, n_re = 0, e_off = 0;
        bool retry = false;
        int reason = 0;
        if (st == ST_REGION) {
            if (P.kkt_mode == 0) {
                // x-law [b_x | A_x] = X0H - sum_a G'[as[a]] (x) L[a]  (x_law_schur's sums, four entries per lane and four members per step in flight)
                gdp X0H = (gdp)P.X0H, Gt = (gdp)P.Gt;
                const int nxr = nx * nr;
                for (int idx0 = lane; idx0 < nxr; idx0 += 256) {
                    double acc[4];
                    int ii[4], tt[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int idx = idx0 + 64 * u, ic = idx < nxr ? idx : 0;
                        ii[u] = ic / nr; tt[u] = ic - ii[u] * nr;
                        acc[u] = X0H[ic];
                    }
                    int a = 0;
                    for (; a + 4 <= k; a += 4) {
                        double gv[4][4];
#pragma unroll
                        for (int v = 0; v < 4; ++v) {
                            const int ra = s.as[a + v] * nx;
#pragma unroll
                            for (int u = 0; u < 4; ++u) gv[v][u] = Gt[ra + ii[u]];
                        }
#pragma unroll
                        for (int v = 0; v < 4; ++v)
#pragma unroll
                            for (int u = 0; u < 4; ++u) acc[u] = fma(-gv[v][u], s.L[(a + v) * nr + tt[u]], acc[u]);
                    }
                    for (; a < k; ++a) {
                        const int ra = s.as[a] * nx;
                        double gv[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) gv[u] = Gt[ra + ii[u]];
#pragma unroll
                        for (int u = 0; u < 4; ++u) acc[u] = fma(-gv[u], s.L[a * nr + tt[u]], acc[u]);
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) if (idx0 + 64 * u < nxr) s.X[idx0 + 64 * u] = acc[u];
                }
                wave_sync();
            }
            const int nlam = k - e, m = nlam + nin + ntc, ldE = nr;
            RegLp<NC, SLOTS> lp;
            lp.m = m + 1; lp.iters = 0; lp.max_iter = 50 * (m + nt) + 200; lp.growth = 0.0;
            lp.alive = (1u << (nt + 2)) - 2u;          // columns 1..nt (sigma) and nt+1 (r)
            lp.cv = lane <= nt ? ID_SIGMA + lane : ID_R;
            wave_sync();
            // FULLC: n_theta equals the instantiation's NT -- no guards on the parameter index (uniform branches otherwise)
            auto build_rows = [&](auto FULLC) {
                constexpr bool FULL = decltype(FULLC)::value;
                gdp gb = (gdp)P.b, gF = (gdp)P.F, gbt = (gdp)P.b_t, gAt = (gdp)P.A_t, gAT = (gdp)P.AT;
#pragma unroll
                for (int sl = 0; sl < SLOTS; ++sl) {
                    const int i = lane + 64 * sl;
                    double h = 0.0, g[NT];
#pragma unroll
                    for (int t = 0; t < NT; ++t) g[t] = 0.0;
                    const bool is_lam = i < nlam, is_in = !is_lam && i < nlam + nin, is_tc = !is_lam && !is_in && i < m;
                    // right-hand side and F / A_t row of this lane's constraint: one batch of loads for both kinds of rows
                    const int ci = is_in ? s.inact[i - nlam] : 0, it = is_tc ? i - nlam - nin : 0;
                    double hv = 0.0, fv[NT];
#pragma unroll
                    for (int t = 0; t < NT; ++t) fv[t] = 0.0;
                    if (is_in || is_tc) {
                        gdp hp = is_in ? gb + ci : gbt + it;
                        gdp fp = is_in ? gF + (size_t)ci * nt : gAt + (size_t)it * nt;
                        hv = *hp;
#pragma unroll
                        for (int t = 0; t < NT; ++t) if (FULL || t < nt) fv[t] = fp[t];
                    }
                    if (is_lam) {
                        h = s.L[(e + i) * nr];
#pragma unroll
                        for (int t = 0; t < NT; ++t) if (FULL || t < nt) g[t] = -s.L[(e + i) * nr + 1 + t];
                    } else if (is_in) {
                        gdp at = gAT + ci;
                        double acc[NT + 1];
#pragma unroll
                        for (int t = 0; t <= NT; ++t) acc[t] = 0.0;
                        int l = 0;
                        for (; l + 4 <= nx; l += 4) {
                            double a4[4];
#pragma unroll
                            for (int u = 0; u < 4; ++u) a4[u] = at[(size_t)(l + u) * nc];
#pragma unroll
                            for (int u = 0; u < 4; ++u)
#pragma unroll
                                for (int t = 0; t <= NT; ++t) if (FULL || t <= nt) acc[t] = fma(a4[u], s.X[(l + u) * nr + t], acc[t]);
                        }
                        for (; l < nx; ++l) {
                            const double a = at[(size_t)l * nc];
#pragma unroll
                            for (int t = 0; t <= NT; ++t) if (FULL || t <= nt) acc[t] = fma(a, s.X[l * nr + t], acc[t]);
                        }
                        h = hv - acc[0];
#pragma unroll
                        for (int t = 0; t < NT; ++t) if (FULL || t < nt) g[t] = acc[1 + t] - fv[t];
                    } else if (is_tc) {
                        h = hv;
#pragma unroll
                        for (int t = 0; t < NT; ++t) if (FULL || t < nt) g[t] = fv[t];
                    }
                    bool keep = false;
                    double ss = 0.0;
#pragma unroll
                    for (int t = 0; t < NT; ++t) { if (!(fabs(g[t]) <= ZERO_ROW_ATOL)) keep = true; ss = fma(g[t], g[t], ss); }
                    keep = keep && i < m;
                    double nrm = 0.0;
                    if (keep) {
                        const double inv = 1.0 / sqrt(ss);
                        h *= inv;
                        double s2 = 0.0;
#pragma unroll
                        for (int t = 0; t < NT; ++t) { g[t] *= inv; s2 = fma(g[t], g[t], s2); }
                        nrm = sqrt(s2);
                        s.E[i * ldE] = h;
#pragma unroll
                        for (int t = 0; t < NT; ++t) if (FULL || t < nt) s.E[i * ldE + 1 + t] = g[t];
                    }
                    // Box screen: the region lies inside the parameter polytope, hence inside its (outward padded) bounding box
                    // [lo, hi].  A row whose left-hand side cannot reach f_i - 1e-6 anywhere in the box is never tight on the
                    // region: the reference's LP "row i as an equality" is infeasible for it (strongly redundant, dropped), and
                    // it cannot bring the Chebyshev radius below 1e-8 either (a ball of radius min(r', 1e-6) around the centre
                    // found without it satisfies it).  Such rows leave the LP before the first pivot.
                    bool boxred = false;
                    if (keep && box && i < nlam + nin) {
                        double mx = 0.0;
#pragma unroll
                        for (int t = 0; t < NT; ++t) if (g[t] != 0.0) mx = fma(g[t], g[t] > 0.0 ? bhi[t] : blo[t], mx);
                        boxred = mx < h - BOX_REDUNDANT_MARGIN;
                    }
                    if (i < m) s.kept[i] = keep ? (boxred ? 2 : 0) : 3;   // 0 undecided, 1 kept, 2 redundant, 3 dropped (numerically zero row)
                    rc_box += __popcll(__ballot(boxred));
                    if (boxred) keep = false;
                    lp.var[sl] = i;
                    lp.kind[sl] = keep ? RK_INEQ : RK_DEAD;
                    // dictionary row at the parameter vertex (theta = tv_theta - tv_minv sigma); the staged blocks are zero beyond n_theta,
                    // which leaves every sum as it is (a sum that starts at +0 never turns -0)
                    double b0 = h;
#pragma unroll
                    for (int t = 0; t < NT; ++t) b0 = fma(-g[t], tvt[t], b0);
                    lp.t[sl][0] = keep ? b0 : 0.0;
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        double acc = 0.0;
#pragma unroll
                        for (int t = 0; t < NT; ++t) acc = fma(g[t], tvm[t * NT + j], acc);
                        lp.t[sl][1 + j] = keep ? -acc : 0.0;
                    }
                    // the r column sits at index nt+1 (runtime): select chain over the compile-time slots
#pragma unroll
                    for (int j = 1; j < NC; ++j) if (j == nt + 1) lp.t[sl][j] = nrm; else if (j > nt + 1) lp.t[sl][j] = 0.0;
                    if (i == m) {   // cost row: minimise -r
                        lp.kind[sl] = RK_COST;
#pragma unroll
                        for (int j = 0; j < NC; ++j) lp.t[sl][j] = (j == nt + 1) ? -1.0 : 0.0;
                    }
                }
            };
            if (nt == NT) build_rows(std::true_type{}); else build_rows(std::false_type{});
            wave_sync();
            const long long tr1 = clock64();
            rc_rows += tr1 - t0;
            // ---- full dimensionality: Chebyshev ball ---------------------------------------------------------------
            int r1 = lp.phase1();
            int r2 = 0;
            if (r1 == LP_OPTIMAL) r2 = lp.primal(-1, m);
            double radius = 0.0;
            int rrow = -1;
            {
#pragma unroll
                for (int sl = SLOTS - 1; sl >= 0; --sl) {
                    const unsigned long long br = __ballot(lp.var[sl] == ID_R && lp.kind[sl] == RK_INEQ);
                    if (br) rrow = __ffsll((long long)br) - 1 + 64 * sl;
                }
                if (rrow >= 0) radius = lp.beta(rrow);
            }
            if (r1 == LP_ITERLIMIT || r2 == 3) st = ST_LP_LIMIT;
            else if (lp.growth > GROWTH_SAFE) {
                // The walk used small pivots, so the tableau's radius is not trusted.  Full dimensionality only needs a
                // LOWER bound above 1e-8: evaluate the inscribed radius of the LP's centre directly on the master rows.
                double th[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) th[t] = t < nt ? P.tv_theta[t] : 0.0;
                for (int j = 0; j < nt; ++j) {
                    double sj = 0.0;
#pragma unroll
                    for (int sl = SLOTS - 1; sl >= 0; --sl) {
                        const unsigned long long br = __ballot(lp.var[sl] == ID_SIGMA + 1 + j && lp.kind[sl] == RK_INEQ);
                        if (br) { double c0[SLOTS];
#pragma unroll
                            for (int q = 0; q < SLOTS; ++q) c0[q] = lp.t[q][0];
                            sj = lp.row_entry(__ffsll((long long)br) - 1 + 64 * sl, c0); }
                    }
#pragma unroll
                    for (int t = 0; t < NT; ++t) if (t < nt) th[t] = fma(-P.tv_minv[t * nt + j], sj, th[t]);
                }
                double rv = INFINITY;
#pragma unroll
                for (int sl = 0; sl < SLOTS; ++sl) {
                    const int i = lane + 64 * sl;
                    if (i < m && s.kept[i] != 3) {
                        double sl_i = s.E[i * ldE];
#pragma unroll
                        for (int t = 0; t < NT; ++t) if (t < nt) sl_i = fma(-s.E[i * ldE + 1 + t], th[t], sl_i);
                        rv = fmin(rv, sl_i);   // rows have unit norm
                    }
                }
                rv = dpp_wave_min(rv);
                if (!(r1 == LP_OPTIMAL && r2 != 2 && rv > 2 * FULL_DIM_RADIUS)) { retry = true; reason = 1; }
            }
            else if (r1 != LP_OPTIMAL || r2 == 2 || !(radius > FULL_DIM_RADIUS)) st = ST_OPT_NO_REGION;
            const long long tr2 = clock64();
            rc_cheb += tr2 - tr1;
            const int it_cheb = lp.iters;
            // ---- facets: walk the feasible dictionary ------------------------------------------------------------------
            if (st == ST_REGION && !retry) {
                lp.set_kind(m, RK_DEAD);                              // the cost row is no longer needed
                int rz = LP_OPTIMAL;
                if (rrow >= 0) rz = lp.drive_to_zero(rrow);           // r back to 0 (slacks only grow: stays feasible)
                else {
                    const unsigned long long bc = __ballot(lp.cv == ID_R && lane >= 1 && lane < NC && ((lp.alive >> lane) & 1u));
                    if (bc) lp.drop_col(__ffsll((long long)bc) - 1);
                }
                if (rz != LP_OPTIMAL) { retry = true; reason = 2; }
                constexpr double KEEP_TOL = TOL_FEAS;
                auto mark_tight = [&]() {
                    // every slack that is zero at the current vertex belongs to a kept row
                    if (lane >= 1 && lane < NC && ((lp.alive >> lane) & 1u) && lp.cv < m && own(lp.cv)) s.kept[lp.cv] = 1;
#pragma unroll
                    for (int sl = 0; sl < SLOTS; ++sl)
                        if (lp.kind[sl] == RK_INEQ && lp.var[sl] < m && lp.t[sl][0] <= KEEP_TOL && own(lp.var[sl])) s.kept[lp.var[sl]] = 1;
                    // ... and every basic slack whose row has no improving column is at ITS minimum here: the run that would
                    // test this row starts at its optimum (zero pivots), so a positive value means the row is redundant
#pragma unroll
                    for (int sl = 0; sl < SLOTS; ++sl) {
                        if (lp.kind[sl] == RK_INEQ && lp.var[sl] < m && lp.t[sl][0] > KEEP_TOL && own(lp.var[sl])) {
                            double mx = 0.0;
#pragma unroll
                            for (int j = 1; j < NC; ++j) mx = bare_max(mx, (double)lp.t[sl][j]);   // (finite tableau entries: the bare instruction, no canonicalising copy)
                            if (R2_CERT && !(mx > TOL_COST) && s.kept[lp.var[sl]] == 0) {
                                s.kept[lp.var[sl]] = 2;
#ifdef R2_DEBUG
                                printf("row %d: certificate at a final vertex, beta=%.3e mx=%.3e\n", lp.var[sl], (double)lp.t[sl][0], mx);
#endif
                            }
                        }
                    }
                    wave_sync();
                };
                // the same two certificates at the intermediate vertices of a run (no barrier: the flags are read after the
                // run's closing mark_tight)
                auto mark_vertex = [&]() {
                    if (lp.growth > GROWTH_SAFE) return;   // a doubtful pivot has happened in this run: no certificates from it
                    if (lane >= 1 && lane < NC && ((lp.alive >> lane) & 1u) && lp.cv < m && own(lp.cv)) s.kept[lp.cv] = 1;
#pragma unroll
                    for (int sl = 0; sl < SLOTS; ++sl) {
                        if (lp.kind[sl] == RK_INEQ && lp.var[sl] < m && own(lp.var[sl])) {
                            if (lp.t[sl][0] <= KEEP_TOL) s.kept[lp.var[sl]] = 1;
                            else {
                                double mx = 0.0;
#pragma unroll
                                for (int j = 1; j < NC; ++j) mx = bare_max(mx, (double)lp.t[sl][j]);   // (finite tableau entries: the bare instruction, no canonicalising copy)
                                if (R2_CERT && !(mx > TOL_COST) && s.kept[lp.var[sl]] == 0) {
                                    s.kept[lp.var[sl]] = 2;
#ifdef R2_DEBUG
                                    printf("row %d: certificate at an intermediate vertex, beta=%.3e mx=%.3e\n", lp.var[sl], (double)lp.t[sl][0], mx);
#endif
                                }
                            }
                        }
                    }
                };
                // constraint (value h, coefficients g) behind a variable id: a region row or the slack of a vertex row of A_t
                auto row_of = [&](int id, double &h, double (&g)[NT]) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) g[t] = 0.0;
                    h = 0.0;
                    if (id < m) {
                        h = s.E[id * ldE];
#pragma unroll
                        for (int t = 0; t < NT; ++t) if (t < nt) g[t] = s.E[id * ldE + 1 + t];
                    } else if (id > ID_SIGMA && id <= ID_SIGMA + nt) {
                        const int tr = P.tv_tight[id - ID_SIGMA - 1];
                        h = P.b_t[tr];
#pragma unroll
                        for (int t = 0; t < NT; ++t) if (t < nt) g[t] = P.A_t[tr * nt + t];
                    }
                };
                // Rebuilds the dictionary at the current basis from the master rows (a fresh n_t x n_t LU in LDS): removes
                // the round-off a long pivot sequence has accumulated.  Columns are renumbered 1..n_t.
                auto refactor = [&]() -> bool {
                    const unsigned al = (unsigned)uni((int)lp.alive);
                    if (__popc(al) != nt) return false;
                    rc_refac++;
                    double *Bm = s.K, *R = s.K + nt * nt;
                    wave_sync();
                    if (lane >= 1 && lane < NC && ((al >> lane) & 1u)) {
                        const int c = __popc(al & ((1u << lane) - 1u));   // basis position of this column
                        double h, g[NT];
                        row_of(lp.cv, h, g);
                        s.colvar[c] = lp.cv;
#pragma unroll
                        for (int t = 0; t < NT; ++t) if (t < nt) Bm[c * nt + t] = g[t];
                        R[c * (nt + 1)] = h;
                        for (int cc = 0; cc < nt; ++cc) R[c * (nt + 1) + 1 + cc] = cc == c ? 1.0 : 0.0;
                    }
                    wave_sync();
                    if (!lu_solve(Bm, nt, R, nt + 1, 1e-12)) return false;
                    bool ok = true;
#pragma unroll
                    for (int sl = 0; sl < SLOTS; ++sl) {
                        double h, g[NT];
                        row_of(lp.var[sl], h, g);
                        const bool live = lp.kind[sl] == RK_INEQ;
                        double b0 = h;
#pragma unroll
                        for (int t = 0; t < NT; ++t) if (t < nt) b0 = fma(-g[t], R[t * (nt + 1)], b0);
                        lp.t[sl][0] = live ? b0 : 0.0;
                        if (live && b0 < -10 * TOL_FEAS) ok = false;
#pragma unroll
                        for (int j = 1; j < NC; ++j) {
                            double acc = 0.0;
                            if (j <= nt) {
#pragma unroll
                                for (int t = 0; t < NT; ++t) if (t < nt) acc = fma(g[t], R[t * (nt + 1) + j], acc);
                            }
                            lp.t[sl][j] = live ? -acc : 0.0;
                        }
                    }
                    lp.cv = (lane >= 1 && lane <= nt) ? s.colvar[lane - 1] : -1;
                    lp.alive = (1u << (nt + 1)) - 2u;
                    lp.growth = 0.0;
                    wave_sync();
                    return !__any(!ok);
                };
                if (!retry && lp.growth > GROWTH_SAFE && !refactor()) { retry = true; reason = 3; }   // the Chebyshev walk ended on small pivots
                if (!retry) mark_tight();
                int refactors = 0;
                bool retested = false;   // the current row is being tested a second time, from a freshly factorised dictionary
                for (int cidx = 0; cidx < m && !retry && st == ST_REGION; ++cidx) {
                    if (s.kept[cidx] != 0 || !own(cidx)) continue;
                    int row = -1;
#pragma unroll
                    for (int sl = SLOTS - 1; sl >= 0; --sl) {
                        const unsigned long long br = __ballot(lp.var[sl] == cidx && lp.kind[sl] == RK_INEQ);
                        if (br) row = __ffsll((long long)br) - 1 + 64 * sl;
                    }
                    if (row < 0) { retry = true; reason = 4; break; }
                    lp.set_kind(row, RK_X0);
                    const int pr = lp.primal_hook(row, -1, false, false, mark_vertex);
                    if (pr == 3) { st = ST_LP_LIMIT; break; }
                    bool kept_c = pr == 4;
                    if (pr != 4) { kept_c = lp.beta(row) <= KEEP_TOL; lp.set_kind(row, RK_INEQ); }
                    bool rebuild = false;
                    if (lp.growth > GROWTH_SAFE) {
                        // second alarm on the same row: a "redundant" verdict stands only with a margin far above what the small
                        // pivot can have cost (error ~ eps * growth * |entries| <= 1e-8), a "kept" verdict stands as it is
                        const bool clear_cut = kept_c || lp.beta(row) > 1e3 * TOL_FEAS;
                        if (!retested || lp.growth > 1e6 || !clear_cut) {
                            // doubtful pivots: this decision is discarded, the dictionary is rebuilt and the row is tested again
                            if (retested || ++refactors > 64 || !refactor()) { retry = true; reason = retested ? 5 : 6; break; }
                            mark_tight();
                            retested = true;
                            --cidx;
                            continue;
                        }
                        // The alarm came back although this short run started from a freshly factorised dictionary: the small
                        // pivot belongs to this row, nothing has accumulated.  A clear-cut decision stands; the run's pivots stay in
                        // the dictionary, the growth monitor starts again, and the next alarm rebuilds it.
                        lp.growth = 0.0;
                    }
                    retested = false;
                    wave_sync();
#ifdef R2_DEBUG
                    if (lane == 0) printf("row %d: LP pr=%d beta=%.3e growth=%.3e kept=%d iters=%d\n", cidx, pr, lp.beta(row), lp.growth, (int)kept_c, lp.iters);
#endif
                    if (lane == 0) s.kept[cidx] = kept_c ? 1 : 2;
                    wave_sync();
                    if (rebuild && (++refactors > 64 || !refactor())) { retry = true; reason = 6; break; }
                    mark_tight();
                }
            }
            pivots += lp.iters;
            const long long tr3 = clock64();
            rc_facet += tr3 - tr2;
            rc_fpiv += lp.iters - it_cheb;
            if (W > 1) {
                // publish the flags of the owned rows, then find out whether this wavefront is the last of its candidate
                for (int i = lane; i < m; i += 64) if (own(i)) kept_g[(size_t)w * ldk + i] = (uint8_t)s.kept[i];
                const unsigned fb = (retry ? 1u : 0u) | (st == ST_LP_LIMIT ? 2u : 0u);
                if (fb && lane == 0) atomicOr(&done_g[2 * w + 1], fb);
                __threadfence();
                unsigned prev = 0;
                if (lane == 0) prev = atomicAdd(&done_g[2 * w], 1u);
                prev = (unsigned)__builtin_amdgcn_readfirstlane((int)prev);
                is_last = prev == (unsigned)(W - 1);
                if (is_last) {
                    __threadfence();
                    for (int i = lane; i < m; i += 64) if (!own(i) && s.kept[i] != 3) s.kept[i] = kept_g[(size_t)w * ldk + i];
                    const unsigned fall = __hip_atomic_load(&done_g[2 * w + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (fall & 1u) retry = true;
                    if ((fall & 2u) && st == ST_REGION) st = ST_LP_LIMIT;
                    wave_sync();
                }
            }
            // ---- record ----------------------------------------------------------------------------------------------
            if (is_last && st == ST_REGION && !retry) {
                int32_t *act = hi + 8, *om = act + k, *la = om + ntc, *ridx = la + k, *rcon = ridx + (nc - k);
                // duplicate rows: row i is dropped from E if an earlier kept row has identical (f, E)
                bool dup[SLOTS], kp[SLOTS];
#pragma unroll
                for (int sl = 0; sl < SLOTS; ++sl) { const int i = lane + 64 * sl; kp[sl] = i < m && s.kept[i] == 1; dup[sl] = false; }
                // (round 4: every lane fingerprints its own row once -- zeros of either sign alike, as == sees them -- and the kept rows are
                //  walked through their ballot mask: a fingerprint comes from its lane by v_readlane and only equal fingerprints are compared
                //  entry by entry.  The loop over all m rows with its LDS reads took 35 k cycles of the ~470 k a region costs, on the wavefront
                //  that arrives last.  Same rows dropped: the comparison that decides is the old one.)
                unsigned long long fp[SLOTS];
#pragma unroll
                for (int sl = 0; sl < SLOTS; ++sl) {
                    const int i = lane + 64 * sl;
                    fp[sl] = 0ull;
                    if (kp[sl])
                        for (int t = 0; t <= nt; ++t) {
                            const double x = s.E[i * ldE + t];
                            const unsigned long long bits = x == 0.0 ? 0ull : (unsigned long long)__double_as_longlong(x);
                            fp[sl] = (fp[sl] ^ bits) * 0x9E3779B97F4A7C15ull;
                            fp[sl] ^= fp[sl] >> 29;
                        }
                }
#pragma unroll
                for (int sj = 0; sj < SLOTS; ++sj) {
                    unsigned long long km = __ballot(kp[sj]);
                    while (km) {
                        const int jl = __ffsll((long long)km) - 1;
                        km &= km - 1ull;
                        const int j = jl + 64 * sj;
                        const unsigned long long fj = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(fp[sj] >> 32), jl) << 32) |
                                                      (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)fp[sj], jl);
#pragma unroll
                        for (int sl = 0; sl < SLOTS; ++sl) {
                            const int i = lane + 64 * sl;
                            if (kp[sl] && i > j && fp[sl] == fj) {
                                bool same = true;
                                for (int t = 0; t <= nt; ++t) same = same && (s.E[i * ldE + t] == s.E[j * ldE + t]);
                                dup[sl] = dup[sl] || same;
                            }
                        }
                    }
                }
                int base_la = 0, base_re = 0, base_om = 0, base_e = 0;
                unsigned long long bE[SLOTS];
#pragma unroll
                for (int sl = 0; sl < SLOTS; ++sl) bE[sl] = __ballot(kp[sl] && !dup[sl]);
#pragma unroll
                for (int sl = 0; sl < SLOTS; ++sl) nE += __popcll(bE[sl]);
                if (lane == 0) e_off = (int)atomicAdd(&ctr->e_rows, (unsigned)nE);
                e_off = __builtin_amdgcn_readfirstlane(e_off);
                const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
                for (int sl = 0; sl < SLOTS; ++sl) {
                    const int i = lane + 64 * sl;
                    const int cls = i < nlam ? 0 : (i < nlam + nin ? 1 : 2);
                    const unsigned long long b0 = __ballot(kp[sl] && cls == 0), b1 = __ballot(kp[sl] && cls == 1), b2 = __ballot(kp[sl] && cls == 2);
                    if (kp[sl] && cls == 0) la[base_la + __popcll(b0 & below)] = s.as[e + i];
                    if (kp[sl] && cls == 1) { const int p = base_re + __popcll(b1 & below); ridx[p] = i - nlam; rcon[p] = s.inact[i - nlam]; }
                    if (kp[sl] && cls == 2) om[base_om + __popcll(b2 & below)] = i - nlam - nin;
                    base_la += __popcll(b0); base_re += __popcll(b1); base_om += __popcll(b2);
                    // the region's rows are one contiguous run of the pool: list the kept rows in order, then write the run with
                    // consecutive lanes on consecutive doubles (coalesced; the pool may be host memory behind PCIe)
                    if (kp[sl] && !dup[sl]) s.stored[base_e + __popcll(bE[sl] & below)] = i;
                    base_e += __popcll(bE[sl]);
                }
                wave_sync();
                {
                    double *dst = epool + (size_t)e_off * nr;
                    for (int idx = lane; idx < nE * nr; idx += 64) { const int rr = idx / nr; dst[idx] = s.E[s.stored[rr] * ldE + (idx - rr * nr)]; }
                }
                n_la = base_la; n_re = base_re; n_om = base_om;
                for (int idx = lane; idx < nx * nt; idx += 64) hd[idx] = s.X[(idx / nt) * nr + 1 + idx % nt];
                for (int i = lane; i < nx; i += 64) hd[nx * nt + i] = s.X[i * nr];
                double *Al = hd + nx * nt + nx, *bl = Al + k * nt;
                for (int idx = lane; idx < k * nt; idx += 64) Al[idx] = s.L[(idx / nt) * nr + 1 + idx % nt];
                for (int i = lane; i < k; i += 64) bl[i] = s.L[i * nr];
                for (int i = lane; i < k; i += 64) act[i] = s.as[i];
            }
        }
        rc_tot += clock64() - t0;
        // "optimal but lower-dimensional" from an ill-conditioned Schur system is not trusted: the candidate is expanded like a
        // feasible, non-optimal one instead of being pruned together with its supersets
        if (ill && st == ST_OPT_NO_REGION && !retry) st = ST_FEASIBLE;
        if (retry) { st = ST_RETRY; if (lane == 0 && is_last) atomicAdd(&ctr->n_rretry, 1u); }
        if (lane == 0 && is_last) {
            hi[0] = st; hi[1] = c; hi[2] = nE; hi[3] = n_om; hi[4] = n_la; hi[5] = n_re; hi[6] = e_off; hi[7] = reason;
            status[c] = (uint8_t)(retry ? ST_RRETRY : st);   // distinct from the verdict stages' ST_RETRY, which may be pending on other candidates
        }
        if (rs.count && is_last) {
            // every store of this slot's record is released to system scope before the slot is counted; the wavefront that
            // completes the chunk tells the host (its flag store follows all counted slots' releases)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
            if (lane == 0) {
                const int ch = (int)(w >> rs.shift);
                const unsigned int full = (unsigned int)min(1 << rs.shift, rs.n_slots - (ch << rs.shift));
                if (atomicAdd(&rs.count[ch], 1u) + 1u == full) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "");
                    __hip_atomic_store(&rs.flags[ch], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
        }
        cyc += clock64() - t0;
    }
    if (lane == 0) {
        atomicAdd(&ctr->pivots, pivots); atomicAdd(&ctr->cycles[3], (unsigned long long)cyc);
        atomicAdd(&ctr->rcycles[0], (unsigned long long)rc_rows); atomicAdd(&ctr->rcycles[1], (unsigned long long)rc_cheb);
        atomicAdd(&ctr->rcycles[2], (unsigned long long)rc_facet); atomicAdd(&ctr->rcycles[3], (unsigned long long)rc_tot);
        atomicAdd(&ctr->rcycles[4], (unsigned long long)rc_refac); atomicAdd(&ctr->rcycles[5], (unsigned long long)rc_fpiv);
        atomicAdd(&ctr->r_box, (unsigned long long)rc_box);
        atomicMax(&ctr->r2_t1, (unsigned long long)wall_clock64());
    }
}

}  // namespace mpc
