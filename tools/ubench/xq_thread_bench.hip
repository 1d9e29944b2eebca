// xq_thread_bench.hip -- the one-thread-per-candidate quick test (k_xq_thread, kernels2.hpp) on synthetic parent records with config 4's
// shape (35 rows, 29 columns, 138 k parents, 7 children each): which part of it costs what, and variants of its memory access.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I ppopt_amd/csrc tools/ubench/xq_thread_bench.hip -o tools/ubench/xq_thread_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "kernels2.hpp"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)
using namespace mpc;

__global__ void fill_records(double *pd, int32_t *pi, long long n_par, int mr, int ncol, int NXC, long long sd, long long si, int nv) {
    const long long p = blockIdx.x;
    if (p >= n_par) return;
    unsigned s = (unsigned)p * 2654435761u + 12345u;
    double *d = pd + p * sd;
    int32_t *ii = pi + p * si;
    for (int idx = threadIdx.x; idx < NXC * mr; idx += blockDim.x) {
        unsigned t = s + idx * 40503u; t ^= t >> 13; t *= 0x5bd1e995u; t ^= t >> 15;
        const double u = (double)(t >> 8) * (1.0 / 16777216.0);
        d[idx] = idx < mr ? 0.1 + u : 2.0 * u - 1.0;
    }
    for (int i = threadIdx.x; i < mr; i += blockDim.x) {
        unsigned t = s + i * 7919u; t ^= t >> 11; t *= 0x9E3779B1u; t ^= t >> 15;
        ii[i] = nv + i;
        ii[mr + i] = RK_INEQ | ((1 + (int)(t % (unsigned)(ncol - 1))) << 8);
    }
    for (int j = threadIdx.x; j < NXC; j += blockDim.x) ii[2 * mr + j] = nv + mr + j;
    if (threadIdx.x == 0) { ii[2 * mr + NXC] = (int)0xfffffffeu; ii[2 * mr + NXC + 1] = __double2hiint(1.0); ii[2 * mr + NXC + 2] = __double2loint(1.0); }
}
__global__ void fill_cands(int32_t *cands, int32_t *parent_slot, int32_t *list, uint8_t *status, long long n, int k, int per, int mr) {
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    unsigned t = (unsigned)c * 2246822519u + 99u; t ^= t >> 13; t *= 0x5bd1e995u; t ^= t >> 15;
    for (int a = 0; a < k - 1; ++a) cands[c * k + a] = a;
    cands[c * k + k - 1] = (int)(t % (unsigned)mr);
    parent_slot[c] = (int32_t)(c / per);
    list[c] = (int32_t)c;
    status[c] = (uint8_t)ST_NEEDX;
}
__global__ void reset_status(uint8_t *status, long long n) {
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c < n) status[c] = (uint8_t)ST_NEEDX;
}

// MODE 1: only the integer part (column check, row search);  2: + the ratio scan, loads batched seven rows at a time;
// 3: the column staged through LDS by the whole wavefront (coalesced loads), scanned by its lane from LDS
template <int MODE>
__global__ void __launch_bounds__(256) k_variant(const DevProblem *__restrict__ Pg, const int32_t *__restrict__ cands, int k, const int32_t *__restrict__ list, int n_list,
                                                  uint8_t *__restrict__ status, DictCache dc, int NXC) {
    extern __shared__ double colbuf[];   // MODE 3: [4 waves][64 candidates][mr | 1]
    const DevProblem &P = *Pg;
    const int nv = P.n_x + P.n_t, mr = P.n_d0r, ncol = P.n_d0c + 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (long long w0 = (long long)blockIdx.x * 256; w0 < n_list; w0 += (long long)gridDim.x * 256) {
        const long long w = w0 + threadIdx.x;
        const bool valid = w < n_list;
        const int c = valid ? list[w] : 0;
        const int ps = valid ? dc.parent_slot[c] : -1;
        const int32_t *pi = dc.prev_i + (size_t)(ps < 0 ? 0 : ps) * dc.stride_i;
        const double *pd = dc.prev_d + (size_t)(ps < 0 ? 0 : ps) * dc.stride_d;
        const int v = nv + cands[(size_t)c * k + (k - 1)];
        const unsigned al = (unsigned)pi[2 * mr + NXC];
        int feas = -1;
#pragma unroll 8
        for (int j = 1; j < ncol; ++j) if (pi[2 * mr + j] == v && ((al >> j) & 1u)) feas = 1;
        int row = -1, q0 = 0;
#pragma unroll 8
        for (int i = mr - 1; i >= 0; --i) {
            const int kraw = pi[mr + i];
            if (pi[i] == v && (kraw & 0xff) == RK_INEQ) { row = i; q0 = kraw >> 8; }
        }
        if (ps < 0) { feas = -1; row = -1; }
        if (feas >= 0) row = -1;
        bool need = false;
        if (row >= 0) {
            const double brow = pd[row];
            if (brow <= TOL_FEAS) feas = 1;
            else if (q0 <= 0) feas = 0;
            else need = true;
        }
        if (MODE == 1) { if (valid && (feas >= 0 || need)) status[c] = (uint8_t)(feas >= 0 ? ST_FEASIBLE : ST_INFEASIBLE); continue; }
        float cmf = 0.0f;
        double tmax = INFINITY, ratio_row = 0.0, a_row = 0.0;
        bool elig_row = false;
        if (MODE == 2) {
            if (need) {
                const double *col = pd + (size_t)q0 * mr;
                for (int i0 = 0; i0 < mr; i0 += 7) {
                    double a7[7], b7[7]; int k7[7];
#pragma unroll
                    for (int u = 0; u < 7; ++u) { const int i = min(i0 + u, mr - 1); a7[u] = col[i]; b7[u] = pd[i]; k7[u] = pi[mr + i]; }
#pragma unroll
                    for (int u = 0; u < 7; ++u) {
                        const int i = i0 + u;
                        if (i < mr) {
                            const double a = a7[u];
                            const int kd = k7[u] & 0xff;
                            const bool used = kd != RK_DEAD;
                            if (used) cmf = fmaxf(cmf, fabsf((float)a));
                            const bool elig = used && kd == RK_INEQ && a > TOL_PIV;
                            if (elig) {
                                const double b0 = fmax(b7[u], 0.0), ia = fast_rcp(a);
                                const double ratio = b0 * ia;
                                tmax = fmin(tmax, (b0 + HARRIS_DELTA) * ia);
                                if (i == row) { elig_row = true; ratio_row = ratio; a_row = a; }
                            }
                        }
                    }
                }
            }
        } else if (MODE == 4 || MODE == 5) {
            if (need) {
                struct __attribute__((packed, aligned(8))) D4 { double v[4]; };
                struct __attribute__((packed, aligned(4))) I4 { int v[4]; };
                const double *col = pd + (size_t)q0 * mr;
                for (int i0 = 0; i0 < mr; i0 += 4) {
                    double a4[4], b4[4]; int k4[4];
                    if (i0 + 4 <= mr) {
                        const D4 xa = *reinterpret_cast<const D4 *>(col + i0);
                        const D4 xb = *reinterpret_cast<const D4 *>(pd + i0);
                        const I4 xk = *reinterpret_cast<const I4 *>(pi + mr + i0);
#pragma unroll
                        for (int u = 0; u < 4; ++u) { a4[u] = xa.v[u]; b4[u] = xb.v[u]; k4[u] = xk.v[u]; }
                    } else {
#pragma unroll
                        for (int u = 0; u < 4; ++u) { const int i = min(i0 + u, mr - 1); a4[u] = col[i]; b4[u] = pd[i]; k4[u] = pi[mr + i]; }
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int i = i0 + u;
                        if (i < mr) {
                            const double a = a4[u];
                            const int kd = k4[u] & 0xff;
                            const bool used = kd != RK_DEAD;
                            if (used) cmf = fmaxf(cmf, fabsf((float)a));
                            const bool elig = used && kd == RK_INEQ && a > TOL_PIV;
                            if (elig) {
                                const double b0 = fmax(b4[u], 0.0), ia = fast_rcp(a);
                                const double ratio = b0 * ia;
                                tmax = fmin(tmax, (b0 + HARRIS_DELTA) * ia);
                                if (i == row) { elig_row = true; ratio_row = ratio; a_row = a; }
                            }
                        }
                    }
                }
            }
        } else {
            // stage: for every candidate of this wavefront that needs the scan, lanes 0..mr-1 read its column coalesced
            const int ld = mr | 1;
            double *cb = colbuf + (size_t)wave * 64 * ld;
            const unsigned long long nm = __ballot(need);
            const unsigned long long pd_bits = (unsigned long long)(pd + (size_t)q0 * mr);
            for (unsigned long long m = nm; m; m &= m - 1ull) {
                const int u = __ffsll((long long)m) - 1;
                const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)pd_bits, u), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(pd_bits >> 32), u);
                const double *colu = reinterpret_cast<const double *>(((unsigned long long)hi << 32) | lo);
                if (lane < mr) cb[u * ld + lane] = colu[lane];
            }
            __builtin_amdgcn_wave_barrier();
            if (need) {
                for (int i = 0; i < mr; ++i) {
                    const double a = cb[lane * ld + i];
                    const int kd = pi[mr + i] & 0xff;
                    const bool used = kd != RK_DEAD;
                    if (used) cmf = fmaxf(cmf, fabsf((float)a));
                    const bool elig = used && kd == RK_INEQ && a > TOL_PIV;
                    if (elig) {
                        const double b0 = fmax(pd[i], 0.0), ia = fast_rcp(a);
                        const double ratio = b0 * ia;
                        tmax = fmin(tmax, (b0 + HARRIS_DELTA) * ia);
                        if (i == row) { elig_row = true; ratio_row = ratio; a_row = a; }
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (need && tmax != INFINITY && elig_row && !(ratio_row > tmax)) {
            const double growth0 = __hiloint2double(pi[2 * mr + NXC + 1], pi[2 * mr + NXC + 2]);
            const double inv = fast_rcp(a_row);
            const double growth = fmax(growth0, (double)(cmf * (float)inv));
            if (!(growth > GROWTH_SAFE)) feas = 1;
        }
        if (valid && feas >= 0) status[c] = (uint8_t)(feas ? ST_FEASIBLE : ST_INFEASIBLE);
    }
}

int main(int argc, char **argv) {
    const int mr = 35, ncol = 29, NXC = 32, k = 5, per = 7, nx = 20, nt = 8;
    const long long n_par = argc > 1 ? std::atoll(argv[1]) : 138315;
    const long long n = n_par * per;
    const long long sd = (long long)NXC * mr, si = 2LL * mr + NXC + 4;
    double *pd; int32_t *pi, *cands, *pslot, *list; uint8_t *status; DevProblem *Pd; LevelCounters *ctr;
    CHECK(hipMalloc(&pd, n_par * sd * 8)); CHECK(hipMalloc(&pi, n_par * si * 4));
    CHECK(hipMalloc(&cands, n * k * 4)); CHECK(hipMalloc(&pslot, n * 4)); CHECK(hipMalloc(&list, n * 4)); CHECK(hipMalloc(&status, n));
    CHECK(hipMalloc(&Pd, sizeof(DevProblem))); CHECK(hipMalloc(&ctr, sizeof(LevelCounters)));
    CHECK(hipMemset(ctr, 0, sizeof(LevelCounters)));
    DevProblem P{}; P.n_x = nx; P.n_t = nt; P.n_d0r = mr; P.n_d0c = ncol - 1;
    CHECK(hipMemcpy(Pd, &P, sizeof(P), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(fill_records, dim3((unsigned)n_par), dim3(256), 0, 0, pd, pi, n_par, mr, ncol, NXC, sd, si, nx + nt);
    hipLaunchKernelGGL(fill_cands, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, cands, pslot, list, status, n, k, per, mr);
    CHECK(hipDeviceSynchronize());
    DictCache dc{}; dc.parent_slot = pslot; dc.prev_d = pd; dc.prev_i = pi; dc.stride_d = sd; dc.stride_i = si;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_variant<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_variant<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_variant<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    auto count = [&]() { std::vector<uint8_t> h(n); CHECK(hipMemcpy(h.data(), status, n, hipMemcpyDeviceToHost)); long long f = 0, i = 0; for (auto s : h) { f += s == ST_FEASIBLE; i += s == ST_INFEASIBLE; } std::printf("  feasible %lld infeasible %lld open %lld\n", f, i, n - f - i); };
    auto timed = [&](const char *name, auto launch) {
        float best = 1e9f;
        for (int r = 0; r < 4; ++r) {
            hipLaunchKernelGGL(reset_status, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, status, n);
            CHECK(hipEventRecord(e0, 0)); launch(); CHECK(hipEventRecord(e1, 0)); CHECK(hipDeviceSynchronize());
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
        }
        std::printf("%-60s %.3f ms", name, best); count();
    };
    for (int gmul : {8}) {
        const unsigned g = (unsigned)std::min<long long>((n + 255) / 256, 256LL * gmul);
        std::printf("grid %u blocks of 256\n", g);
        timed("k_xq_thread (library)", [&] { hipLaunchKernelGGL(k_xq_thread, dim3(g * 4), dim3(64), 0, 0, Pd, cands, k, list, (int)n, status, ctr, dc, NXC, XqAlt{}, XqPlan{}); });
        timed("variant 1: column check + row search only", [&] { hipLaunchKernelGGL(k_variant<1>, dim3(g), dim3(256), 0, 0, Pd, cands, k, list, (int)n, status, dc, NXC); });
        timed("variant 2: ratio scan, loads batched by seven", [&] { hipLaunchKernelGGL(k_variant<2>, dim3(g), dim3(256), 0, 0, Pd, cands, k, list, (int)n, status, dc, NXC); });
        timed("variant 3: column staged through LDS", [&] { hipLaunchKernelGGL(k_variant<3>, dim3(g), dim3(256), 4 * 64 * (mr | 1) * 8, 0, Pd, cands, k, list, (int)n, status, dc, NXC); });
        timed("variant 4: 32-byte loads, four rows per batch", [&] { hipLaunchKernelGGL(k_variant<4>, dim3(g), dim3(256), 0, 0, Pd, cands, k, list, (int)n, status, dc, NXC); });
        for (int lds_kb : {20, 40, 80}) {
            char nm[96]; std::snprintf(nm, sizeof nm, "variant 4 with %d KB LDS per block (%d blocks per CU)", lds_kb, 160 / lds_kb);
            timed(nm, [&] { hipLaunchKernelGGL(k_variant<4>, dim3(g), dim3(256), lds_kb * 1024, 0, Pd, cands, k, list, (int)n, status, dc, NXC); });
            std::snprintf(nm, sizeof nm, "variant 2 with %d KB LDS per block (%d blocks per CU)", lds_kb, 160 / lds_kb);
            timed(nm, [&] { hipLaunchKernelGGL(k_variant<2>, dim3(g), dim3(256), lds_kb * 1024, 0, Pd, cands, k, list, (int)n, status, dc, NXC); });
        }
    }
    {
        const unsigned g = (unsigned)((n + 255) / 256);
        std::printf("grid %u blocks (one pass)\n", g);
        timed("k_xq_thread (library)", [&] { hipLaunchKernelGGL(k_xq_thread, dim3(g * 4), dim3(64), 0, 0, Pd, cands, k, list, (int)n, status, ctr, dc, NXC, XqAlt{}, XqPlan{}); });
        timed("variant 2", [&] { hipLaunchKernelGGL(k_variant<2>, dim3(g), dim3(256), 0, 0, Pd, cands, k, list, (int)n, status, dc, NXC); });
        timed("variant 3", [&] { hipLaunchKernelGGL(k_variant<3>, dim3(g), dim3(256), 4 * 64 * (mr | 1) * 8, 0, Pd, cands, k, list, (int)n, status, dc, NXC); });
    }
    return 0;
}
