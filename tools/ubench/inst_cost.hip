// inst_cost.hip -- what the register simplex's instruction mix costs on a gfx950 SIMD, at 1 / 2 / 4 wavefronts per SIMD.
//
// Round 4 found that one work item of k_theta2 takes 29 / 48 / 130 us with 1 / 2 / 4 wavefronts per SIMD: a SIMD's throughput on
// this code peaks at two wavefronts although "VALU busy" reads 33 %.  This program prices the ingredients:
//   part A  single instructions (inline asm, 16 copies per loop trip, 2,000 trips): cycles per instruction of ONE wavefront
//           and of W wavefronts sharing a SIMD (throughput), dependent and independent forms
//   part B  the real code: RegLp<NC,1>::primal on random bounded LPs (cycles per pivot), and its parts in isolation
//           (pivot_core / pricing / the ratio test with its three wave reductions)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I ppopt_amd/csrc tools/ubench/inst_cost.hip -o tools/ubench/inst_cost
// One workgroup of 256 * W threads puts W wavefronts on each SIMD of a CU (a workgroup's waves go round the four SIMDs);
// 100 KB of dynamic LDS keep a second workgroup off the CU.  `grid` workgroups run at once (1 = one CU, 256 = the whole chip).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>
#include "lp_reg.hpp"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

#define R4(x) x x x x
#define R16(x) R4(x) R4(x) R4(x) R4(x)

struct Out { long long cycles; long long units; };

template <int ID>
__global__ void __launch_bounds__(1024) k_inst(int trips, Out *out, double seed) {
    extern __shared__ double lds_pad[];
    double x = seed + threadIdx.x * 1e-3, y = 0.5 + seed, z = 1e-9 * seed;
    double x1 = x + 1, x2 = x + 2, x3 = x + 3;
    float fx = (float)x, fy = 0.5f, fz = 1e-9f;
    int ix = threadIdx.x, iy = 3;
    int sacc = 0;
    const long long t0 = clock64();
    for (int t = 0; t < trips; ++t) {
        if constexpr (ID == 0) asm volatile(R16("v_fma_f64 %0, %0, %1, %2\n") : "+v"(x) : "v"(y), "v"(z));
        if constexpr (ID == 1) asm volatile(R4("v_fma_f64 %0, %0, %4, %5\nv_fma_f64 %1, %1, %4, %5\nv_fma_f64 %2, %2, %4, %5\nv_fma_f64 %3, %3, %4, %5\n")
                                            : "+v"(x), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(y), "v"(z));
        if constexpr (ID == 2) asm volatile(R16("v_fma_f32 %0, %0, %1, %2\n") : "+v"(fx) : "v"(fy), "v"(fz));
        if constexpr (ID == 3) asm volatile(R16("v_mul_f64 %0, %0, %1\n") : "+v"(x) : "v"(y));
        if constexpr (ID == 4) asm volatile(R16("v_max_f64 %0, %0, %1\n") : "+v"(x) : "v"(y));
        if constexpr (ID == 5) asm volatile(R16("v_rcp_f64 %0, %0\n") : "+v"(x));
        if constexpr (ID == 6) asm volatile(R16("v_readlane_b32 %0, %1, 7\n") : "=s"(sacc) : "v"(ix));
        if constexpr (ID == 7)   // the pivot's column step: two v_readlane and one fma that reads the SGPR pair
            asm volatile(R16("v_readlane_b32 s20, %1, 7\nv_readlane_b32 s21, %2, 7\nv_fma_f64 %0, %3, s[20:21], %0\n")
                         : "+v"(x) : "v"(ix), "v"(iy), "v"(z) : "s20", "s21");
        if constexpr (ID == 8) asm volatile(R16("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n") : "+v"(ix) : "v"(iy));
        if constexpr (ID == 9) asm volatile(R16("v_cndmask_b32 %0, %0, %1, vcc\n") : "+v"(ix) : "v"(iy) : "vcc");
        if constexpr (ID == 10) asm volatile(R16("v_cmp_gt_f64 vcc, %0, %1\n") : : "v"(x), "v"(y) : "vcc");
        if constexpr (ID == 11)   // VGPR index mode round trip with one indexed move inside (index 0: plain semantics)
            asm volatile(R16("s_set_gpr_idx_on %2, 0x1\nv_mov_b32 %0, %1\ns_set_gpr_idx_off\n") : "+v"(ix) : "v"(iy), "s"(0));
        if constexpr (ID == 12) asm volatile(R16("v_readfirstlane_b32 %0, %1\n") : "=s"(sacc) : "v"(ix));
        if constexpr (ID == 13) asm volatile(R16("s_memtime s[20:21]\ns_waitcnt lgkmcnt(0)\n") : : : "s20", "s21");
        if constexpr (ID == 14) asm volatile(R16("v_cmp_gt_f64 s[20:21], %0, %1\ns_and_b64 s[20:21], s[20:21], exec\n") : : "v"(x), "v"(y) : "s20", "s21", "scc");
        if constexpr (ID == 15) asm volatile(R16("v_cvt_f32_f64 %0, %1\n") : "=v"(fx) : "v"(x));
        if constexpr (ID == 16) asm volatile(R16("v_writelane_b32 %0, %1, 5\n") : "+v"(ix) : "s"(3));
        if constexpr (ID == 17) asm volatile(R16("v_min_f64 %0, %0, %1\n") : "+v"(x) : "v"(y));
        if constexpr (ID == 18) asm volatile(R16("v_add_f64 %0, %0, %1\n") : "+v"(x) : "v"(z));
        if constexpr (ID == 19) asm volatile(R16("s_mov_b32 s20, s21\n") : : : "s20");
        if constexpr (ID == 20) asm volatile(R16("v_mov_b32 %0, %1\n") : "=v"(ix) : "v"(iy));
        if constexpr (ID == 21)   // DPP step of a f64 reduction as the compiler writes it: two DPP moves and a min, dependent
            asm volatile(R16("s_nop 1\nv_mov_b32_dpp %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %2, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n")
                         : "+v"(ix), "+v"(iy), "+v"(sacc) : "v"(ix));
        if constexpr (ID == 22) asm volatile(R16("v_cmp_eq_u32 vcc, %0, %1\n") : : "v"(ix), "v"(iy) : "vcc");
        if constexpr (ID == 23) asm volatile(R16("v_mul_f64 %0, %0, %1\nv_mul_f64 %2, %2, %1\n") : "+v"(x), "+v"(x1) : "v"(y));
        if constexpr (ID == 24) asm volatile("s_mov_b64 s[20:21], 0x55\n" R16("v_cndmask_b32_e64 %0, %0, %1, s[20:21]\n") : "+v"(ix) : "v"(iy) : "s20", "s21");
        if constexpr (ID == 25) asm volatile(R16("v_cndmask_b32 %0, %1, %2, vcc\n") : "=v"(sacc) : "v"(ix), "v"(iy) : "vcc");
        if constexpr (ID == 26) asm volatile(R16("v_bfi_b32 %0, %1, %2, %0\n") : "+v"(ix) : "v"(iy), "v"(sacc));
        if constexpr (ID == 27) asm volatile(R16("v_and_b32 %0, %0, %1\n") : "+v"(ix) : "v"(iy));
        if constexpr (ID == 28) asm volatile(R4("v_cndmask_b32 %0, %0, %4, vcc\nv_cndmask_b32 %1, %1, %4, vcc\nv_cndmask_b32 %2, %2, %4, vcc\nv_cndmask_b32 %3, %3, %4, vcc\n")
                                             : "+v"(ix), "+v"(iy), "+v"(sacc), "+v"(fx) : "v"(fy) : "vcc");
        if constexpr (ID == 29) asm volatile(R16("v_cmp_gt_f64 vcc, %1, %2\nv_cndmask_b32 %0, %0, %3, vcc\n") : "+v"(ix) : "v"(x), "v"(y), "v"(iy) : "vcc");
        if constexpr (ID == 30) asm volatile("s_mov_b64 vcc, 0x55\n" R16("v_cndmask_b32 %0, %0, %1, vcc\n") : "+v"(ix) : "v"(iy) : "vcc");
        if constexpr (ID == 31) asm volatile(R16("v_cmp_gt_f64 s[20:21], %1, %2\nv_cndmask_b32_e64 %0, %0, %3, s[20:21]\n") : "+v"(ix) : "v"(x), "v"(y), "v"(iy) : "s20", "s21");
        if constexpr (ID == 32) asm volatile(R16("v_max_f64 %0, %0, %1\nv_cmp_eq_f64 vcc, %0, %1\n") : "+v"(x) : "v"(y) : "vcc");
        if constexpr (ID == 33) asm volatile(R16("v_add_u32 %0, %0, %1\n") : "+v"(ix) : "v"(iy));
        if constexpr (ID == 34) asm volatile(R16("v_lshlrev_b32 %0, 1, %0\n") : "+v"(ix));
        if constexpr (ID == 35) asm volatile(R16("v_or_b32 %0, %0, %1\n") : "+v"(ix) : "v"(iy));
        if constexpr (ID == 36) asm volatile(R16("v_mov_b32 %0, %0\n") : "+v"(ix));
        if constexpr (ID == 37) asm volatile(R16("v_fma_f64 %0, %0, %2, %3\nv_mov_b32 %1, %1\n") : "+v"(x), "+v"(ix) : "v"(y), "v"(z));
    }
    const long long t1 = clock64();
    if ((threadIdx.x & 63) == 0) {
        Out o; o.cycles = t1 - t0; o.units = (long long)trips * 16;
        out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = o;
    }
    if (x + x1 + x2 + x3 + fx + ix + sacc == 1.2345e-300) lds_pad[0] = x;
}

// ---- part B: the register simplex itself -----------------------------------------------------------------------------------
__device__ __forceinline__ double rnd(unsigned &s) { s = s * 1664525u + 1013904223u; return (double)(s >> 8) * (1.0 / 16777216.0); }

// a bounded random LP at a feasible vertex: rows 0..NC-2 are the box x_j <= 1, the others random with positive right-hand side,
// row 63 is the cost row.  MODE 0: primal to optimality (real pivots);  1: pivot_core only (forced r, q);  2: pricing only;
// 3: the ratio test with its reductions only (no pivot);  4: getq + setq only (the indexed VGPR accesses)
template <int NC, int MODE>
__global__ void __launch_bounds__(1024) k_lp(int reps, Out *out, unsigned seed0) {
    extern __shared__ double lds_pad[];
    using namespace mpc;
    const int lane = threadIdx.x & 63;
    unsigned s = seed0 * 2654435761u + (blockIdx.x * blockDim.x + threadIdx.x) * 40503u + 17u;
    long long cyc = 0, units = 0;
    double sink = 0.0;
    for (int rep = 0; rep < reps; ++rep) {
        RegLp<NC, 1> lp;
        lp.m = 64; lp.iters = 0; lp.max_iter = 400; lp.growth = 0.0;
        lp.alive = (NC >= 32 ? 0xfffffffeu : ((1u << NC) - 2u)) & ~(1u << (NC - 1));   // the last slot stays free (x0's)
        lp.cv = lane;
        lp.var[0] = 100 + lane;
        lp.kind[0] = lane == 63 ? RK_COST : RK_INEQ;
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            double v = 2.0 * rnd(s) - 1.0;
            if (j == 0) v = lane < NC - 2 ? 1.0 : 1.0 + 4.0 * rnd(s);
            else if (lane < NC - 2) v = (lane + 1 == j) ? 1.0 : 0.0;
            if (j == NC - 1) v = 0.0;
            if (lane == 63) v = j == 0 ? 0.0 : (j == NC - 1 ? 0.0 : -rnd(s));   // cost row: every column improves at the start
            lp.t[0][j] = v;
        }
        const long long t0 = clock64();
        if constexpr (MODE == 0) {
            const int st = lp.primal(-1, 63);
            sink += st;
            units += lp.iters;
        } else if constexpr (MODE == 1) {
            for (int it = 0; it < 32; ++it) {
                const int q = 1 + (it % (NC - 2)), r = (it * 7 + 3) & 63;
                double f[1] = {lp.t[0].getq(q)};
                lp.pivot_core(r, q, f, 1.0 + 1e-3 * it);
            }
            units += 32;
        } else if constexpr (MODE == 2) {
            for (int it = 0; it < 32; ++it) {
                unsigned mk = 0;
                const int q = lp.template price<false>(63, &mk);
                sink += q + (int)mk;
                lp.t[0][1 + (it % (NC - 2))] = lp.t[0][1 + (it % (NC - 2))] * 0.999;
            }
            units += 32;
        } else if constexpr (MODE == 3) {
            for (int it = 0; it < 32; ++it) {
                const int q = 1 + (it % (NC - 2));
                const double a = lp.t[0].getq(q);
                const bool used = lp.kind[0] != RK_DEAD;
                float cmf = used ? fabsf((float)a) : 0.0f;
                const bool elig = used && lp.kind[0] == RK_INEQ && a > TOL_PIV;
                double ratio = 0.0, tmax = INFINITY;
                if (elig) { const double b0 = fmax(lp.t[0].get(0), 0.0), ia = fast_rcp(a); ratio = b0 * ia; tmax = (b0 + HARRIS_DELTA) * ia; }
                const float colmax = dpp_wave_max_f32(cmf);
                tmax = dpp_wave_min(tmax);
                const bool pass = elig && !(ratio > tmax);
                const double rpiv = dpp_wave_max(pass ? a : 0.0);
                const unsigned long long br = __ballot(pass && a == rpiv);
                const int r = uni(__ffsll((long long)br) - 1);
                const double rmin = readlane_f64(ratio, r & 63), inv = fast_rcp(rpiv);
                sink += rmin + inv + colmax;
                lp.t[0][0] = lp.t[0][0] + 1e-9;
            }
            units += 32;
        } else {
            for (int it = 0; it < 32; ++it) {
                const int q = uni(1 + (it % (NC - 2)));
                const double a = lp.t[0].getq(q);
                lp.t[0].setq(q, a * 1.0001);
            }
            units += 32;
        }
        cyc += clock64() - t0;
#pragma unroll
        for (int j = 0; j < NC; ++j) sink += lp.t[0][j];
    }
    if (lane == 0) { Out o; o.cycles = cyc; o.units = units; out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = o; }
    if (sink == 1.2345e-300) lds_pad[0] = sink;
}

template <int ID> static void inst(const char *name) {
    std::printf("%-46s", name);
    for (int gi = 0; gi < 2; ++gi) {
        const int grid = gi == 0 ? 1 : 256;
        for (int W : {1, 2, 4}) {
            const int waves = grid * W * 4;
            Out *out = nullptr;
            CHECK(hipMalloc(&out, sizeof(Out) * waves));
            for (int r = 0; r < 2; ++r) {
                hipLaunchKernelGGL(k_inst<ID>, dim3(grid), dim3(256 * W), 100 * 1024, 0, 2000, out, 1.0 + r);
                CHECK(hipDeviceSynchronize());
            }
            std::vector<Out> h(waves);
            CHECK(hipMemcpy(h.data(), out, sizeof(Out) * waves, hipMemcpyDeviceToHost));
            double c = 0, u = 0;
            for (auto &o : h) { c += (double)o.cycles; u += (double)o.units; }
            std::printf("  %7.2f /%6.2f", c / u, c / u / W);
            CHECK(hipFree(out));
        }
        std::printf(gi == 0 ? "  |" : "\n");
    }
}

template <int NC, int MODE> static void lpb(const char *name) {
    std::printf("%-46s", name);
    for (int gi = 0; gi < 2; ++gi) {
        const int grid = gi == 0 ? 1 : 256;
        for (int W : {1, 2, 4}) {
            const int waves = grid * W * 4;
            Out *out = nullptr;
            CHECK(hipMalloc(&out, sizeof(Out) * waves));
            for (int r = 0; r < 2; ++r) {
                hipLaunchKernelGGL((k_lp<NC, MODE>), dim3(grid), dim3(256 * W), 100 * 1024, 0, 64, out, 7u + r);
                CHECK(hipDeviceSynchronize());
            }
            std::vector<Out> h(waves);
            CHECK(hipMemcpy(h.data(), out, sizeof(Out) * waves, hipMemcpyDeviceToHost));
            double c = 0, u = 0;
            for (auto &o : h) { c += (double)o.cycles; u += (double)o.units; }
            std::printf("  %7.0f /%6.0f", c / u, c / u / W);
            if (MODE == 0 && gi == 0 && W == 1) std::fprintf(stderr, "[%s: %.1f pivots per LP]\n", name, u / (waves * 64.0));
            CHECK(hipFree(out));
        }
        std::printf(gi == 0 ? "  |" : "\n");
    }
}

int main() {
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_inst<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    std::printf("cycles per unit: one wavefront's own / SIMD time (own / W); W = 1, 2, 4 wavefronts per SIMD; left: one CU busy, right: 256 CUs busy\n");
    std::printf("%-46s  %15s %15s %15s  | %15s %15s %15s\n", "unit", "W=1", "W=2", "W=4", "W=1", "W=2", "W=4");
#define I(ID, NAME) CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_inst<ID>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024)); inst<ID>(NAME)
    I(0, "v_fma_f64 dependent");
    I(1, "v_fma_f64 4 chains");
    I(2, "v_fma_f32 dependent");
    I(3, "v_mul_f64 dependent");
    I(23, "v_mul_f64 2 chains (per pair)");
    I(18, "v_add_f64 dependent");
    I(4, "v_max_f64 dependent");
    I(17, "v_min_f64 dependent");
    I(5, "v_rcp_f64 dependent");
    I(15, "v_cvt_f32_f64");
    I(10, "v_cmp_gt_f64 vcc");
    I(14, "v_cmp_gt_f64 sgpr + s_and exec (ballot)");
    I(22, "v_cmp_eq_u32 vcc");
    I(9, "v_cndmask_b32");
    I(20, "v_mov_b32");
    I(36, "v_mov_b32 dependent");
    I(33, "v_add_u32 dependent");
    I(34, "v_lshlrev_b32 dependent");
    I(27, "v_and_b32 dependent");
    I(35, "v_or_b32 dependent");
    I(26, "v_bfi_b32 dependent");
    I(30, "v_cndmask_b32 vcc (vcc set), dependent");
    I(24, "v_cndmask_b32_e64 sgpr mask, dependent");
    I(25, "v_cndmask_b32 vcc, independent dst");
    I(28, "v_cndmask_b32 vcc, 4 chains");
    I(29, "v_cmp_gt_f64 vcc + v_cndmask (pair)");
    I(31, "v_cmp_gt_f64 sgpr + v_cndmask_e64 (pair)");
    I(32, "v_max_f64 + v_cmp_eq_f64 (pair)");
    I(37, "v_fma_f64 dep + v_mov_b32 (pair)");
    I(6, "v_readlane_b32");
    I(12, "v_readfirstlane_b32");
    I(16, "v_writelane_b32");
    I(7, "2 x v_readlane + v_fma_f64 (sgpr pair)");
    I(8, "v_mov_b32_dpp row_shr");
    I(21, "s_nop 1 + 2 x v_mov_b32_dpp");
    I(11, "s_set_gpr_idx_on + v_mov + _off");
    I(19, "s_mov_b32");
    I(13, "s_memtime + wait");
#undef I
#define L(NC, MODE, NAME) CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_lp<NC, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024)); lpb<NC, MODE>(NAME)
    L(10, 0, "RegLp<10>: primal, per pivot");
    L(10, 1, "RegLp<10>: pivot_core");
    L(10, 2, "RegLp<10>: pricing");
    L(10, 3, "RegLp<10>: ratio test + reductions");
    L(10, 4, "RegLp<10>: getq + setq");
    L(11, 0, "RegLp<11>: primal, per pivot");
    L(16, 0, "RegLp<16>: primal, per pivot");
    L(32, 0, "RegLp<32>: primal, per pivot");
    L(32, 1, "RegLp<32>: pivot_core");
    L(32, 2, "RegLp<32>: pricing");
    L(32, 3, "RegLp<32>: ratio test + reductions");
    L(32, 4, "RegLp<32>: getq + setq");
#undef L
    return 0;
}
