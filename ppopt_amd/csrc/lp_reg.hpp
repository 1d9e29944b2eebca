// lp_reg.hpp -- register-resident simplex for the hot kernels (gfx950): one wavefront per LP, NO LDS, NO barriers.
//
// Lane i owns tableau row i (and row i+64 when SLOTS == 2) in VGPRs: t[s][0] = value of the basic variable,
// t[s][1..NC-1] = coefficients of the nonbasic columns (NC-1 slots for at most NC-2 variables: phase 1 puts its
// artificial x0 into whichever slot is free).  All column loops
// are unrolled at compile time (NC is a template parameter) so every tableau entry is a named register.
//   * a row is an LLVM vector (8 or 16 doubles; two halves for 32 columns): a column chosen at run time is read or
//     written with ONE indexed VGPR access (s_set_gpr_idx) because the index is wave-uniform -- no select chains
//   * the pivot row is scaled inside its own lane, then broadcast column by column with v_readlane -> SGPRs; every other
//     lane does one fma per column (the pivot lane's multiplier is forced to 0, column q is first replaced by e_r)
//   * pricing is a scan over the cost row inside the lane that owns it, one v_readlane returns the column
//   * the ratio test is the only cross-lane step: two f64 DPP (row_shr) reductions + ballots, one f32 reduction for the
//     growth monitor; reciprocals are v_rcp_f64 + two Newton steps instead of IEEE divisions
//   * a deleted column is a cleared bit in a uniform mask AND zeroed in every row, so pricing needs no liveness test;
//     the var id of column j lives in lane j of `cv`
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
// Wave reductions of the ratio test (round 6).  A lane value moved by DPP / v_readlane reaches fmin / fmax as two integer halves, so the
// compiler puts a canonicalising v_max x, x in front of every min / max, and a row_shr step has lanes without a source, which costs two
// copies of the running value per step: seven instructions per step of an f64 reduction, three reductions per pivot.  Here a step is a
// ROTATION within the 16-lane row (row_ror: every lane has a source, no copy) and the min / max itself is the bare instruction -- the
// reduced values are ratios, pivots and absolute values: never NaN, never -0, so the result is the same number whatever the order.
// After four steps every lane of a row holds the row's result; four v_readlane + four min / max with a scalar operand finish it.
template <int CTRL> __device__ __forceinline__ int dpp_all_i32(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }
template <int CTRL> __device__ __forceinline__ double dpp_all_f64(double v) {
    return __hiloint2double(dpp_all_i32<CTRL>(__double2hiint(v)), dpp_all_i32<CTRL>(__double2loint(v)));
}
__device__ __forceinline__ double bare_min(double a, double b) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ double bare_max(double a, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ double bare_min_s(double a_uniform, double b) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "s"(a_uniform), "v"(b)); return r; }
__device__ __forceinline__ double bare_max_s(double a_uniform, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "s"(a_uniform), "v"(b)); return r; }
__device__ __forceinline__ float bare_maxf(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float bare_maxf_s(float a_uniform, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "s"(a_uniform), "v"(b)); return r; }
// (results: the same value in every lane)
__device__ __forceinline__ double rot_wave_min(double v) {
    v = bare_min(v, dpp_all_f64<0x121>(v)); v = bare_min(v, dpp_all_f64<0x122>(v)); v = bare_min(v, dpp_all_f64<0x124>(v)); v = bare_min(v, dpp_all_f64<0x128>(v));
    const double a = readlane_f64(v, 0), b = readlane_f64(v, 16), c = readlane_f64(v, 32), d = readlane_f64(v, 48);
    return bare_min_s(d, bare_min_s(c, bare_min_s(b, bare_min_s(a, v))));
}
__device__ __forceinline__ double rot_wave_max(double v) {
    v = bare_max(v, dpp_all_f64<0x121>(v)); v = bare_max(v, dpp_all_f64<0x122>(v)); v = bare_max(v, dpp_all_f64<0x124>(v)); v = bare_max(v, dpp_all_f64<0x128>(v));
    const double a = readlane_f64(v, 0), b = readlane_f64(v, 16), c = readlane_f64(v, 32), d = readlane_f64(v, 48);
    return bare_max_s(d, bare_max_s(c, bare_max_s(b, bare_max_s(a, v))));
}
// The row_shr forms (rounds 1-5), kept for the kernels that run under a tight register cap (k_x1, k_xq, k_xq_grouped: the rotation forms
// above cost them 90-150 bytes of scratch each) -- row_shr:1,2,4,8 leave the reduction of each 16-lane row in its last lane; four
// v_readlane finish it
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

__device__ __forceinline__ double fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    double e = fma(-x, r, 1.0);
    r = fma(r, e, r);
    e = fma(-x, r, 1.0);
    return fma(r, e, r);
}
__device__ __forceinline__ float dpp_wave_max_f32(float v) {
    v = fmaxf(v, __int_as_float(dpp_i32<0x111>(__float_as_int(v)))); v = fmaxf(v, __int_as_float(dpp_i32<0x112>(__float_as_int(v))));
    v = fmaxf(v, __int_as_float(dpp_i32<0x114>(__float_as_int(v)))); v = fmaxf(v, __int_as_float(dpp_i32<0x118>(__float_as_int(v))));
    const int b = __float_as_int(v);
    return fmaxf(fmaxf(__int_as_float(__builtin_amdgcn_readlane(b, 15)), __int_as_float(__builtin_amdgcn_readlane(b, 31))),
                 fmaxf(__int_as_float(__builtin_amdgcn_readlane(b, 47)), __int_as_float(__builtin_amdgcn_readlane(b, 63))));
}
__device__ __forceinline__ float rot_wave_max_f32(float v) {
    v = bare_maxf(v, __int_as_float(dpp_all_i32<0x121>(__float_as_int(v)))); v = bare_maxf(v, __int_as_float(dpp_all_i32<0x122>(__float_as_int(v))));
    v = bare_maxf(v, __int_as_float(dpp_all_i32<0x124>(__float_as_int(v)))); v = bare_maxf(v, __int_as_float(dpp_all_i32<0x128>(__float_as_int(v))));
    const int b = __float_as_int(v);
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(b, 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(b, 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(b, 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(b, 48));
    return bare_maxf_s(r3, bare_maxf_s(r2, bare_maxf_s(r1, bare_maxf_s(r0, v))));
}
__device__ __forceinline__ int dpp_wave_min_i32(int v) {
    v = min(v, dpp_i32<0x111>(v)); v = min(v, dpp_i32<0x112>(v)); v = min(v, dpp_i32<0x114>(v)); v = min(v, dpp_i32<0x118>(v));
    return min(min(__builtin_amdgcn_readlane(v, 15), __builtin_amdgcn_readlane(v, 31)), min(__builtin_amdgcn_readlane(v, 47), __builtin_amdgcn_readlane(v, 63)));
}

// A value forced through a register (empty asm).  With two rows per lane the choice "slot 0 or slot 1" by a wave-uniform run-time row is
// written as a select of two values; without this the optimiser turns the select (or two branches with the same code) into ONE access at a
// run-time index of the array of slots, and an array indexed at run time lives in scratch -- the whole tableau did (round 4).
__device__ __forceinline__ double opaque(double e) { asm volatile("" : "+v"(e)); return e; }
__device__ __forceinline__ int opaque(int e) { asm volatile("" : "+v"(e)); return e; }

template <int W> struct RowVec;
template <> struct RowVec<8> { typedef double type __attribute__((ext_vector_type(8))); };
template <> struct RowVec<16> { typedef double type __attribute__((ext_vector_type(16))); };

// One tableau row in VGPRs.  operator[] with a compile-time index (unrolled loops) is a plain register; getq / setq take
// a WAVE-UNIFORM run-time index.
template <int NC, bool SELECT = false>
struct RegRow {
    static constexpr int W = NC <= 8 ? 8 : 16;
    static constexpr int H = (NC + W - 1) / W;
    static_assert(H <= 2, "at most 32 columns");
    typename RowVec<W>::type v[H];
    __device__ __forceinline__ double get(int j) const { return v[j / W][j % W]; }
    __device__ __forceinline__ void set(int j, double x) { v[j / W][j % W] = x; }
    __device__ __forceinline__ double getq(int q) const {
        if (H == 1) return v[0][q];
        return q < W ? v[0][q] : v[H - 1][q - W];
    }
    __device__ __forceinline__ void setq(int q, double x) {
        if (H == 1) v[0][q] = x;
        else if (q < W) v[0][q] = x;
        else v[H - 1][q - W] = x;
    }
    struct Ref {
        RegRow &row; int j;
        __device__ __forceinline__ operator double() const { return row.get(j); }
        __device__ __forceinline__ Ref &operator=(double x) { row.set(j, x); return *this; }
    };
    __device__ __forceinline__ Ref operator[](int j) { return Ref{*this, j}; }
    __device__ __forceinline__ double operator[](int j) const { return get(j); }
};

// Two rows per lane (SLOTS == 2): with two run-time-indexed 16-double vectors per lane the compiler keeps the tableau rows in a stack object
// (round 4: 320-896 bytes of scratch in every <.,2> instantiation, 47 k cycles per pivot in k_region2<8,2> against 3.6 k in <8,1>).  Here a
// row is NC named registers; the two accesses with a wave-uniform run-time column per pivot (read the entering column, write e_r) are
// select chains over the compile-time columns -- 2 NC v_cndmask per pivot instead of scratch traffic on every entry.
template <int NC>
struct RegRow<NC, true> {
    double v[NC];
    __device__ __forceinline__ double get(int j) const { return v[j]; }
    __device__ __forceinline__ void set(int j, double x) { v[j] = x; }
    // (the element goes through an empty asm: otherwise the optimiser folds the chain back into ONE load / store at a run-time index,
    //  which puts the row -- and a mirror store for every update of every entry -- into scratch)
    static __device__ __forceinline__ double in_register(double e) { return opaque(e); }
    __device__ __forceinline__ double getq(int q) const {
        double r = in_register(v[0]);
#pragma unroll
        for (int j = 1; j < NC; ++j) r = (j == q) ? in_register(v[j]) : r;
        return r;
    }
    __device__ __forceinline__ void setq(int q, double x) {
#pragma unroll
        for (int j = 0; j < NC; ++j) v[j] = (j == q) ? x : in_register(v[j]);
    }
    struct Ref {
        RegRow &row; int j;
        __device__ __forceinline__ operator double() const { return row.get(j); }
        __device__ __forceinline__ Ref &operator=(double x) { row.set(j, x); return *this; }
    };
    __device__ __forceinline__ Ref operator[](int j) { return Ref{*this, j}; }
    __device__ __forceinline__ double operator[](int j) const { return get(j); }
};

// WEIGHTED: row i carries a power-of-two scale w (its entries are w times the constraint as posed).  Phase 1 then measures
// infeasibility in the constraint's own units: x0 enters row i with coefficient -w, so "x0 <= 1e-7" means every constraint
// holds within 1e-7 as posed, however much the row was scaled for the pivoting.
template <int NC, int SLOTS, bool WEIGHTED = false>
struct RegLp {
    RegRow<NC, (SLOTS >= 2)> t[SLOTS];
    double w[SLOTS], winv[SLOTS];   // only read when WEIGHTED
    int kind[SLOTS];
    int var[SLOTS];
    int cv;           // lane j holds the variable id of column j
    unsigned alive;   // uniform bit mask of live columns (bits 1..NC-1); dead columns are zero in every row
    int m;            // rows in use (row index = lane + 64*slot)
    int iters, max_iter;
    double growth;
    // the wave reductions of this instantiation: row rotations with bare min / max (round 6), except for the two-row 32-column tableau
    // (k_x2<32,2>: 256 registers and 260 bytes of scratch already -- the rotation forms cost it 36 bytes more and 6-8 % of its time)
    static constexpr bool ROT = !(SLOTS >= 2 && NC > 16);
    static __device__ __forceinline__ double wred_min(double v) { return ROT ? rot_wave_min(v) : dpp_wave_min(v); }
    static __device__ __forceinline__ double wred_max(double v) { return ROT ? rot_wave_max(v) : dpp_wave_max(v); }
    static __device__ __forceinline__ float wred_max_f32(float v) { return ROT ? rot_wave_max_f32(v) : dpp_wave_max_f32(v); }

    __device__ __forceinline__ double row_entry(int r, const double (&c)[SLOTS]) const {
        const int rl = r & 63;
        if (SLOTS == 1) return readlane_f64(c[0], rl);
        return readlane_f64((r >> 6) ? opaque(c[SLOTS - 1]) : opaque(c[0]), rl);
    }
    __device__ __forceinline__ int row_entry_i(int r, const int (&c)[SLOTS]) const {
        const int rl = r & 63;
        if (SLOTS == 1) return __builtin_amdgcn_readlane(c[0], rl);
        return __builtin_amdgcn_readlane((r >> 6) ? opaque(c[SLOTS - 1]) : opaque(c[0]), rl);
    }

    // Pivot on (r, q): f = column q before the pivot, inv = 1 / f[r].
    __device__ __forceinline__ void pivot_core(int r, int q, const double (&f)[SLOTS], double inv) {
        const int lane = lane_id();
        double fz[SLOTS];
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            const bool is_r = (lane + 64 * s) == r;
            t[s].setq(q, is_r ? 1.0 : 0.0);          // column q := e_r; the generic update then yields inv / -f*inv
            fz[s] = is_r ? 0.0 : f[s];               // the pivot lane's own row is only scaled
            if (is_r) {
#pragma unroll
                for (int j = 0; j < NC; ++j) t[s].set(j, t[s].get(j) * inv);
            }
        }
        if (SLOTS == 1) {
#pragma unroll
            for (int j = 0; j < NC; ++j) {
                const double trj = readlane_f64(t[0].get(j), r & 63);
#pragma unroll
                for (int s = 0; s < SLOTS; ++s) t[s].set(j, fma(-fz[s], trj, t[s].get(j)));
            }
        } else {
            const bool upper = (r >> 6) != 0;
#pragma unroll
            for (int j = 0; j < NC; ++j) {
                const double trj = readlane_f64(upper ? opaque(t[SLOTS - 1].get(j)) : opaque(t[0].get(j)), r & 63);
#pragma unroll
                for (int s = 0; s < SLOTS; ++s) t[s].set(j, fma(-fz[s], trj, t[s].get(j)));
            }
        }
        const int vq = __builtin_amdgcn_readlane(cv, q);
        const int vr = row_entry_i(r, var);
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) if ((lane + 64 * s) == r) var[s] = vq;
        if (lane == q) cv = vr;
        iters++;
    }
    __device__ __forceinline__ void pivot(int r_in, int q_in) {
        const int r = uni(r_in), q = uni(q_in);
        double f[SLOTS];
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) f[s] = t[s].getq(q);
        pivot_core(r, q, f, fast_rcp(row_entry(r, f)));
    }

    __device__ __forceinline__ void drop_col(int q_in) {
        const int q = uni(q_in);
        alive = (unsigned)uni((int)(alive & ~(1u << q)));
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) t[s].setq(q, 0.0);
    }

    __device__ __forceinline__ void set_kind(int r, int k) {
        const int lane = lane_id();
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) if ((lane + 64 * s) == r) kind[s] = k;
    }
    __device__ __forceinline__ int get_kind(int r) const { return row_entry_i(r, kind); }
    __device__ __forceinline__ double beta(int r) const {
        double c0[SLOTS];
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) c0[s] = t[s].get(0);
        return row_entry(r, c0);
    }

    // largest |entry| > TOL_PIV of row r (ties: lowest column); -1 if none.  Scan inside the owning lane.
    __device__ __forceinline__ int best_col(int r_in) const {
        const int r = uni(r_in);
        int q[SLOTS];
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            q[s] = -1;
            double best = TOL_PIV;
#pragma unroll
            for (int j = 1; j < NC; ++j) {
                const double a = fabs(t[s].get(j));
                if (a > best) { best = a; q[s] = j; }
            }
        }
        return row_entry_i(r, q);
    }

    // Dantzig pricing of row `crow` (phase 1: the row of x0, minimise; else the cost row): scan inside the owning lane.
    // P1: entries > TOL_COST are improving; else entries < -TOL_COST.  Returns the column or -1.
    template <bool P1>
    __device__ __forceinline__ int price(int crow) const {
        // round 6: the largest improving entry by a chain of bare max (min for the cost row: the entries are compared as stored, no
        // negation), then the lowest column that holds it -- a compare and one select per column where "if (g > best) { best = g; q = j; }"
        // was a compare and three selects.  Same column: the first of the largest entries, none unless it exceeds TOL_COST.
        int q[SLOTS];
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            double best = P1 ? TOL_COST : -TOL_COST;
#pragma unroll
            for (int j = 1; j < NC; ++j) best = P1 ? bare_max(best, t[s].get(j)) : bare_min(best, t[s].get(j));
            q[s] = -1;
#pragma unroll
            for (int j = NC - 1; j >= 1; --j) q[s] = (t[s].get(j) == best) ? j : q[s];
            if (!(P1 ? best > TOL_COST : best < -TOL_COST)) q[s] = -1;
        }
        return row_entry_i(crow, q);
    }
    // the improving columns of row `crow` as a bit mask (Bland's rule only: off the path of an ordinary pivot since round 6)
    template <bool P1>
    __device__ __forceinline__ unsigned improving_mask(int crow) const {
        int mk[SLOTS];
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            mk[s] = 0;
#pragma unroll
            for (int j = 1; j < NC; ++j) {
                const double g = P1 ? t[s].get(j) : -t[s].get(j);
                if (g > TOL_COST) mk[s] |= (1 << j);
            }
        }
        return (unsigned)row_entry_i(crow, mk);
    }

    // primal simplex; phase1_row >= 0: minimise x0 (basic in that row), else minimise the RK_COST row `cost_row`.
    // 0 optimal, 2 unbounded, 3 iteration limit, 4 x0 left the basis
    // drop_on_leave: when the minimised variable leaves the basis its column is deleted (it stays fixed at zero);
    // false = it stays a regular nonbasic variable (facet tests in k_region2).
    // decide_only: the caller only wants to know whether x0 can leave (last level: the dictionary is not kept) -- the
    // final pivot is skipped, the tableau is left one pivot behind.
    struct NoHook { __device__ __forceinline__ void operator()() const {} };
    __device__ __forceinline__ int primal(int phase1_row, int cost_row, bool drop_on_leave = true, bool decide_only = false) {
        return primal_hook(phase1_row, cost_row, drop_on_leave, decide_only, NoHook());
    }
    // the same with a callable that is invoked after every pivot (the dictionary then sits at a new feasible vertex)
    template <class Hook>
    __device__ __forceinline__ int primal_hook(int phase1_row, int cost_row, bool drop_on_leave, bool decide_only, Hook after_pivot) {
        const int lane = lane_id();
        int deg = 0;
        for (;;) {
            if (iters > max_iter) return 3;
            const bool bland = deg > DEG_SWITCH;
            const int crow = uni(phase1_row >= 0 ? phase1_row : cost_row);
            if (phase1_row >= 0 && beta(phase1_row) <= TOL_FEAS) return 0;
            int q = phase1_row >= 0 ? price<true>(crow) : price<false>(crow);
            if (q < 0) return 0;
            if (bland) {   // smallest variable id among the improving columns
                const unsigned improving = phase1_row >= 0 ? improving_mask<true>(crow) : improving_mask<false>(crow);
                const bool mine = lane >= 1 && lane < NC && ((improving >> lane) & 1u);
                const int vmin = dpp_wave_min_i32(mine ? cv : 0x7fffffff);
                q = __ffsll((long long)__ballot(mine && cv == vmin)) - 1;
            }
            q = uni(q);
            // ratio test (Harris two-pass; textbook + Bland while stalled)
            double a[SLOTS], ratio[SLOTS];
            bool elig[SLOTS];
            float cmf = 0.0f;
            double tmax = INFINITY;
#pragma unroll
            for (int s = 0; s < SLOTS; ++s) {
                const int i = lane + 64 * s;
                a[s] = t[s].getq(q);
                const bool used = (i < m) & (kind[s] != RK_DEAD);      // (bitwise: no short-circuit branches)
                if (used) cmf = fmaxf(cmf, fabsf((float)a[s]));
                elig[s] = used & ((kind[s] == RK_INEQ) | (kind[s] == RK_X0)) & (a[s] > TOL_PIV);
                // (round 6: straight-line -- the reciprocal is taken in every lane, of 1 where the row is not eligible, and the results are
                //  selected: the nested exec-mask blocks of the branchy form cost more than the five instructions they skipped)
                const double b0 = fmax(t[s].get(0), 0.0), ia = fast_rcp(elig[s] ? a[s] : 1.0);
                ratio[s] = elig[s] ? b0 * ia : 0.0;
                tmax = fmin(tmax, elig[s] ? (b0 + HARRIS_DELTA) * ia : INFINITY);
            }
            const float colmax = wred_max_f32(cmf);
            tmax = wred_min(tmax);
            if (tmax == INFINITY) return 2;
            int r = -1;
            bool leaving_x0 = false;
            double rpiv = 0.0;
            if (bland) {
                double rm = INFINITY;
#pragma unroll
                for (int s = 0; s < SLOTS; ++s) if (elig[s]) rm = fmin(rm, ratio[s]);
                rm = wred_min(rm);
                unsigned long long key = 0; int idx = -1;
#pragma unroll
                for (int s = 0; s < SLOTS; ++s)
                    if (elig[s] && ratio[s] == rm) {
                        const unsigned long long kk = ((unsigned long long)(kind[s] == RK_X0) << 40) | (unsigned long long)(0x7fffffff - var[s]);
                        if (key_better(kk, lane + 64 * s, key, idx)) { key = kk; idx = lane + 64 * s; }
                    }
                r = dpp_wave_argmax(key, idx);
                if (r < 0) return 2;
                rpiv = row_entry(r, a);
                leaving_x0 = get_kind(r) == RK_X0;
            } else {
                // among the rows inside the Harris bound: the x0 row if it is one of them, else the largest pivot
                bool pass[SLOTS];
#pragma unroll
                for (int s = 0; s < SLOTS; ++s) pass[s] = elig[s] && !(ratio[s] > tmax);
#pragma unroll
                for (int s = SLOTS - 1; s >= 0; --s) {
                    const unsigned long long bx = __ballot(pass[s] && kind[s] == RK_X0);
                    if (bx) { r = __ffsll((long long)bx) - 1 + 64 * s; leaving_x0 = true; }
                }
                if (leaving_x0) rpiv = row_entry(r, a);
                else {
                    double am = 0.0;
#pragma unroll
                    for (int s = 0; s < SLOTS; ++s) if (pass[s]) am = fmax(am, a[s]);
                    rpiv = wred_max(am);
#pragma unroll
                    for (int s = SLOTS - 1; s >= 0; --s) {
                        const unsigned long long br = __ballot(pass[s] && a[s] == rpiv);
                        if (br) r = __ffsll((long long)br) - 1 + 64 * s;
                    }
                    if (r < 0) return 2;
                }
            }
            r = uni(r);
            const double rmin = row_entry(r, ratio), inv = fast_rcp(rpiv);
            growth = fmax(growth, (double)(colmax * (float)inv));
            deg = (rmin <= 0.0) ? deg + 1 : 0;
            const bool skip = leaving_x0 && decide_only;
            if (!skip) { pivot_core(r, q, a, inv); after_pivot(); }
            else iters++;
            if (leaving_x0) { set_kind(r, RK_INEQ); if (drop_on_leave && !skip) drop_col(q); return 4; }
        }
    }

    // Drives the basic variable of row r to zero and out of the basis while every other row stays feasible: a primal
    // simplex run whose objective is that variable (the x0 machinery with the row's own slack in the role of x0).
    // Used to activate a constraint at a feasible vertex.  LP_OPTIMAL (done, column deleted) / LP_INFEASIBLE (its
    // minimum over the polytope is positive) / LP_ITERLIMIT.
    __device__ __forceinline__ int drive_to_zero(int r, bool decide_only = false) {
        set_kind(r, RK_X0);
        const int st = primal(r, -1, true, decide_only);
        if (st == 3) return LP_ITERLIMIT;
        if (st != 4) {
            if (beta(r) > TOL_FEAS) return LP_INFEASIBLE;
            if (decide_only) return LP_OPTIMAL;
            const int qq = best_col(r);
            if (qq < 0) set_kind(r, RK_DEAD);
            else { pivot(r, qq); set_kind(r, RK_INEQ); drop_col(qq); }
        }
        return LP_OPTIMAL;
    }

    // Phase 1 on the current dictionary (re-entrant).  LP_OPTIMAL (feasible) / LP_INFEASIBLE / LP_ITERLIMIT.
    __device__ __forceinline__ int phase1() {
        const int lane = lane_id();
        // most negative basic value (ties: lowest row)
        bool neg[SLOTS];
        double vmin = INFINITY;
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            const double v0 = WEIGHTED ? t[s].get(0) * winv[s] : t[s].get(0);   // value in the constraint's own units
            neg[s] = lane + 64 * s < m && kind[s] == RK_INEQ && v0 < -TOL_FEAS;
            if (neg[s]) vmin = fmin(vmin, v0);
        }
        vmin = wred_min(vmin);
        if (vmin == INFINITY) return LP_OPTIMAL;
        int r = -1;
#pragma unroll
        for (int s = SLOTS - 1; s >= 0; --s) {
            const double v0 = WEIGHTED ? t[s].get(0) * winv[s] : t[s].get(0);
            const unsigned long long br = __ballot(neg[s] && v0 == vmin);
            if (br) r = __ffsll((long long)br) - 1 + 64 * s;
        }
        r = uni(r);
        // x0 takes a free column slot.  (Not a fixed one: when x0 left the basis in an earlier phase 1 it was deleted from
        // the column it had moved to, and the slot it first entered through now holds a live slack.)
        const unsigned al = (unsigned)uni((int)alive);
        const int xc = uni(__ffs((int)(~al & ~1u & ((NC >= 32 ? 0u : (1u << NC)) - 1u))) - 1);
        alive = al | (1u << xc);
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) t[s].setq(xc, (lane + 64 * s < m && kind[s] == RK_INEQ) ? (WEIGHTED ? -w[s] : -1.0) : 0.0);
        if (lane == xc) cv = X0_VAR;
        pivot(r, xc);
        return drive_to_zero(r);
    }
};

}  // namespace mpc
