// batch_level.hip -- the launches of one level for many member programs (batch_level.hpp).
//
// MPC_GLOBAL turns every kernel of kernels.hpp / kernels2.hpp into an inlined device function in this translation unit; the
// kernels below fetch their member's argument block (blockIdx.y) and call those bodies with exactly the arguments
// level_run_small (mpcombi_hip.hip) passes for a single program.  blockIdx.x / gridDim.x keep their meaning inside the bodies:
// the x-extent of a launch is the largest any member of the group needs, a member's surplus blocks leave at once (every body
// tests its index or its work queue against the member's own counts).
#define MPC_GLOBAL __device__ __forceinline__
#define MPC_LB(...)
#include "batch_level.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <tuple>
#include <vector>

namespace mpc {
namespace {

#define MEMBER const BatchMember &m = tab[blockIdx.y]
__device__ __forceinline__ int32_t *part_list_of(const BatchMember &m, int c) { return m.part_lists + (size_t)c * (size_t)m.n; }

// ---- clears -------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) m_zero(const BatchMember *__restrict__ tab) {
    MEMBER;
    for (int z = 0; z < m.n_zero; ++z) {
        const unsigned long long bytes = m.zero[z].bytes;
        if ((reinterpret_cast<unsigned long long>(m.zero[z].p) & 7ull) == 0 && (bytes & 7ull) == 0) {
            unsigned long long *p = reinterpret_cast<unsigned long long *>(m.zero[z].p);
            for (unsigned long long i = blockIdx.x * 256ull + threadIdx.x; i < bytes / 8; i += (unsigned long long)gridDim.x * 256ull) p[i] = 0ull;
        } else {
            unsigned char *p = reinterpret_cast<unsigned char *>(m.zero[z].p);
            for (unsigned long long i = blockIdx.x * 256ull + threadIdx.x; i < bytes; i += (unsigned long long)gridDim.x * 256ull) p[i] = 0;
        }
    }
}
__global__ void m_zero_counter(const BatchMember *__restrict__ tab, int which) {   // 0: work_retry, 1: work_region
    MEMBER;
    if (threadIdx.x == 0) { if (which == 0) m.ctr->work_retry = 0u; else m.ctr->work_region = 0u; }
}

// ---- theta stage --------------------------------------------------------------------------------------------------------------
template <int K, int NT, int SP>
__global__ void __launch_bounds__(256) m_kkt_thread(const BatchMember *__restrict__ tab) {
    MEMBER;
    k_kkt_thread<K, NT, SP>(m.pf, m.fr, m.n, m.kkt_code, m.kkt_L, m.status, m.targs, m.ctr);
}
__global__ void __launch_bounds__(1024) m_compact_small(const BatchMember *__restrict__ tab, int lo, int hi, int into_retry, int slot) {
    MEMBER;
    k_compact_small(m.status, (int)m.n, lo, hi, into_retry ? m.retry_list : m.theta_list, m.dcnt + slot);
}
template <int NT, int SLOTS>
__global__ void __launch_bounds__(64, (SLOTS >= 2 ? (NT <= 4 ? TH_WAVES_S2 : 2) : (NT >= 8 ? 3 : 4))) m_theta2(const BatchMember *__restrict__ tab) {
    MEMBER;
    ThetaArgs ta = m.targs;
    ta.chunk = 1;
    if (m.th_blocks > 0) { ta.wave_max = m.th_blocks; ta.wave_div = 0; }
    const uint8_t *kkc = nullptr;
    const double *kkl = nullptr;
    const int32_t *list = nullptr;
    if (m.use_kkt) { kkc = m.kkt_code; kkl = m.kkt_L; list = m.theta_list; ta.n_dev = m.dcnt + 0; }
    k_theta2<NT, SLOTS>(m.pf, m.fr, m.n, m.k, m.status, m.ctr, kkc, kkl, ta, list);
}
__global__ void __launch_bounds__(1024) m_partition_small(const BatchMember *__restrict__ tab, unsigned long long spec, int slot) {
    MEMBER;
    k_partition_small(m.status, (int)m.n, spec, m.part_lists, m.n, m.dcnt + slot);
}
__global__ void __launch_bounds__(64) m_verdict(const BatchMember *__restrict__ tab, int slot) {
    MEMBER;
    k_verdict(m.Pv, m.fr, m.n, m.k, m.status, m.ctr, part_list_of(m, 0), m.dcnt + slot);
}

// members whose parameter set is open: the candidates called optimal are asked whether the reference's max-t LP is bounded (kernels.hpp)
__global__ void __launch_bounds__(64) m_recession(const BatchMember *__restrict__ tab) {
    MEMBER;
    if (!m.theta_open) return;
    k_recession(m.Pv, m.fr, m.n, m.k, m.status);
}

// ---- region stage -------------------------------------------------------------------------------------------------------------
template <int NT, int SLOTS>
__global__ void __launch_bounds__(64, (SLOTS >= 2 ? 2 : R2W)) m_region2(const BatchMember *__restrict__ tab) {
    MEMBER;
    const uint8_t *kkc = m.use_kkt ? m.kkt_code : nullptr;
    const double *kkl = m.use_kkt ? m.kkt_L : nullptr;
    RegionStream rs = m.rs;
    // the member's share also bounds the wavefronts that may SHARE one candidate (W: each of them repeats the Chebyshev LP, which pays
    // on an idle device only).  A different W walks the facets in a different order: the records are bit for bit the single program's
    // where W agrees (MPC_NO_RSPLIT=1 on both sides: always), else except for facets on the LP tolerance (tests/test_gpu_batch.py)
    rs.max_blocks = m.r2_blocks;
    if (m.r2_wcap > 0) rs.w_cap = m.r2_wcap;
    k_region2<NT, SLOTS>(m.pr, m.fr, m.k, part_list_of(m, 2), (int)m.n, m.status, m.headd, m.headi, m.fd, m.fi, m.epool, m.ctr, kkc, kkl, m.W,
                         m.kept_g, m.ldk, m.done_g, m.no_rbox ? (const double *)nullptr : m.targs.tvp + (size_t)NT * NT + NT, rs);
}

// the candidates k_region2 gave up on (status RRETRY, compacted into retry_list, length at dcnt[28]): the LDS-engine kernel in its
// latency form, as launch_region_v1 runs it for a short list -- one wavefront per (candidate, region row), then the assembly pass
template <int MODE>
__global__ void __launch_bounds__(64) m_region_v1(const BatchMember *__restrict__ tab) {
    MEMBER;
    const int n_list = m.dcnt[28];
    if (n_list <= 0 || n_list > m.rcap) return;      // none (the usual case) / more than the reserved slots: the member repeats the level alone
    k_region<MODE>(m.Pr, m.fr, m.k, m.retry_list, n_list, m.status, m.recd, m.reci, m.rec_d, m.rec_i, m.ctr, m.facet_flags);
}

// ---- (x,theta) stage ------------------------------------------------------------------------------------------------------------
template <int SLOTS>
__global__ void __launch_bounds__(64, XQ_WAVES) m_xq(const BatchMember *__restrict__ tab) {
    MEMBER;
    DictCache dq = m.dc;
    dq.n_list_dev = m.dcnt + 7;
    dq.max_blocks = m.xq_blocks;
    k_xq<SLOTS>(m.pf, m.fr, m.k, part_list_of(m, 3), (int)m.n, m.status, m.ctr, dq, m.nxc);
}
// one-step plans (round 5; mpcombi_hip.hip launch_plans): the members with `plan` ask, one thread per candidate, whether a parent's record is
// one known step away; k_x1 streams those records through the step; m_x2 below gets what is left (lists in the member's x1_buf)
__global__ void __launch_bounds__(64) m_xq_plan(const BatchMember *__restrict__ tab) {
    MEMBER;
    if (!m.plan) return;
    DictCache dq = m.dc;
    dq.n_list_dev = m.dcnt + 7;
    int32_t *xb = m.x1_buf;
    const size_t nn = (size_t)m.n;
    XqPlan pl{};
    pl.plan_slot = xb; pl.plan_step = xb + nn; pl.x1_list = xb + 2 * nn; pl.x1_n = m.dcnt + 9;
    for (int sg = 0; sg < 3; ++sg) { pl.rest[sg] = xb + (3 + sg) * nn; pl.rest_n[sg] = m.dcnt + 29 + sg; }
    pl.pre1 = part_list_of(m, 1); pl.n_pre1_dev = m.dcnt + 5;
    pl.pre2 = part_list_of(m, 2); pl.n_pre2_dev = m.dcnt + 6;
    k_xq_thread(m.pf, m.fr, m.k, part_list_of(m, 3), (int)m.n, m.status, m.ctr, dq, m.nxc, m.alt, pl);
}
template <int SLOTS>
__global__ void __launch_bounds__(64, 8) m_x1(const BatchMember *__restrict__ tab) {
    MEMBER;
    if (!m.plan) return;
    int32_t *xb = m.x1_buf;
    const size_t nn = (size_t)m.n;
    k_x1<SLOTS>(m.pf, xb + 2 * nn, m.dcnt + 9, m.ctr, m.dc, m.nxc, xb, xb + nn);
}
template <int NXC, int SLOTS>
__global__ void __launch_bounds__(64, (NXC * SLOTS >= 64 ? 2 : (NXC * SLOTS >= 32 ? X2_WAVES : X2_WAVES_16))) m_x2(const BatchMember *__restrict__ tab) {
    MEMBER;
    if (m.plan) {     // what the plan pass left: three lists in x1_buf, lengths in dcnt[29..31]
        int32_t *xb = m.x1_buf;
        const size_t nn = (size_t)m.n;
        DictCache dr = m.dc;
        dr.pre1 = xb + 3 * nn; dr.n_pre1 = 0; dr.n_pre1_dev = m.dcnt + 29;
        dr.pre2 = xb + 4 * nn; dr.n_pre2 = 0; dr.n_pre2_dev = m.dcnt + 30;
        dr.n_list_dev = m.dcnt + 31;
        dr.max_blocks = m.x2_blocks;
        k_x2<NXC, SLOTS>(m.pf, m.fr, m.k, xb + 5 * nn, (int)m.n, m.status, m.ctr, dr);
        return;
    }
    DictCache d = m.dc;
    d.n_list_dev = m.dcnt + (m.quick_test ? 8 : 7);
    d.max_blocks = m.x2_blocks;
    k_x2<NXC, SLOTS>(m.pf, m.fr, m.k, m.quick_test ? m.retry_list : part_list_of(m, 3), (int)m.n, m.status, m.ctr, d);
}

// ---- pruned masks, children, counters -----------------------------------------------------------------------------------------
template <int MW>
__global__ void m_pruned_append(const BatchMember *__restrict__ tab) {
    MEMBER;
    k_pruned_append<MW>(m.fr, m.n, m.k, m.status, m.pruned + (size_t)m.n_pruned * MW, m.ctr, m.keep_lowdim);
}
template <int MW>
__global__ void __launch_bounds__(64) m_children_count(const BatchMember *__restrict__ tab) {
    MEMBER;
    k_children_count<MW>(m.Pv, m.fr, m.n, m.k, m.status, m.pruned, m.n_pruned, m.childmask, m.count, m.keep_lowdim);
}
__global__ void __launch_bounds__(1024) m_scan_small(const BatchMember *__restrict__ tab) {
    MEMBER;
    k_scan_small(m.count, m.offset, (int)m.n, m.dcnt + 20);
}
__global__ void __launch_bounds__(64) m_children_write(const BatchMember *__restrict__ tab) {
    MEMBER;
    k_children_write(m.fr, m.n, m.k, m.mw, m.childmask, m.offset, m.children, m.storing ? m.dict_stored_cur : (const uint8_t *)nullptr, m.parent_slot_next);
}
__global__ void m_histogram(const BatchMember *__restrict__ tab) {
    MEMBER;
    k_histogram(m.status, m.n, m.ctr);
}
__global__ void m_publish(const BatchMember *__restrict__ tab) {
    MEMBER;
    k_publish_words(reinterpret_cast<const unsigned int *>(m.ctr), m.pub_ctr, (int)(sizeof(LevelCounters) / 4));
    k_publish_words(reinterpret_cast<const unsigned int *>(m.dcnt), m.pub_cnt, 32);
}

}  // namespace
// ---- one program: region kernel + (x,theta) kernel as one grid (batch_level.hpp, SmallRX) -------------------------------------------
// (outside the unnamed namespace: the profiler then shows mpc::s_region2_x2<...>)
template <int NT, int SL, int NXC, int SLX>
__global__ void __launch_bounds__(64, 2) s_region2_x2(SmallRX a) {
    if (blockIdx.y == 0)
        k_region2<NT, SL>(a.pr, a.fr, a.k, a.opt_list, a.n, a.status, a.headd, a.headi, a.fd, a.fi, a.epool, a.ctr, a.kkc, a.kkl, a.W, a.kept_g, a.ldk, a.done_g, a.tvp_box, a.rs);
    else
        k_x2<NXC, SLX>(a.pf, a.fr, a.k, a.list, a.n, a.status, a.ctr, a.dc);
}
namespace {

unsigned long long spec_of(std::initializer_list<std::pair<int, int>> classes) {
    unsigned long long spec = ~0ull;
    for (const auto &sc : classes) spec = (spec & ~(15ull << (4 * sc.first))) | ((unsigned long long)sc.second << (4 * sc.first));
    return spec;
}

auto group_key(const BatchMember &m) { return std::make_tuple(m.k, m.kd, m.fast_t, m.fast_x, m.fast_r, m.mw, m.use_kkt, m.kkt_listed, m.quick_test, m.gen_children); }

int env_int(const char *name, int dflt) { const char *v = std::getenv(name); return v && *v ? std::atoi(v) : dflt; }

template <class F>
hipError_t raise_lds(F *fn, int bytes) {
    if (bytes <= 48 * 1024) return hipSuccess;
    return hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

}  // namespace

#define TRY(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return e_; } while (0)

bool small_region2_x2_launch(int fast_r, int fast_x, unsigned grid_r, unsigned grid_x, int lds_r2, hipStream_t st, const SmallRX &a, hipError_t *err) {
    // the pairs the named configurations and the enumeration's sub-programs use (each instantiation holds both kernels' code)
    const dim3 g(std::max(std::max(grid_r, grid_x), 1u), 2), b(64);
    *err = hipSuccess;
#define MPC_RX(NT_, SL_, NXC_, SLX_) do { *err = raise_lds(s_region2_x2<NT_, SL_, NXC_, SLX_>, lds_r2); if (*err == hipSuccess) { hipLaunchKernelGGL((s_region2_x2<NT_, SL_, NXC_, SLX_>), g, b, lds_r2, st, a); *err = hipGetLastError(); } return true; } while (0)
    if (fast_r == 0 && fast_x == 0) MPC_RX(4, 1, 16, 1);
    if (fast_r == 0 && fast_x == 2) MPC_RX(4, 1, 32, 1);
    if (fast_r == 2 && fast_x == 0) MPC_RX(8, 1, 16, 1);
    if (fast_r == 2 && fast_x == 2) MPC_RX(8, 1, 32, 1);
#undef MPC_RX
    return false;
}

hipError_t batch_level_launch(BatchMember *members, int B, hipStream_t st, BatchMember *g_tab_host, BatchMember *g_tab_dev) {
    if (B <= 0) return hipSuccess;
    if (!g_tab_host || !g_tab_dev) return hipErrorInvalidValue;
    std::stable_sort(members, members + B, [](const BatchMember &a, const BatchMember &b) { return group_key(a) < group_key(b); });
    // Wavefront shares.  The persistent kernels of a stage run fastest with few wavefronts per SIMD (mpcombi_hip.hip: a single program's
    // k_theta2 two, k_x2 three, k_xq five, k_region2 two) -- for the whole launch, not per member: each member gets its share of that budget in proportion
    // to its candidates, so that all members of a group run side by side and finish together.  (Round 3 gave every member the width
    // of a single program's launch: 64 members x 4,096 blocks, the members one after the other, each with its own tail.)
    // (budgets in wavefronts per CU, swept on the bench enumeration -- four-parameter sub-programs, the NT = 4 instantiations: the theta
    //  kernel 8 / 16 / 32 -> 67.5 / 60.3 / 58.3 ms of shared levels, the region kernel 8 / 16 -> 67.5 / 66.4, k_x2 8 / 12 / 16 -> 63.0 / 60.3 / 62.4)
    static const int wpc_th = env_int("MPC_BATCH_WPC_TH", 32), wpc_xq = env_int("MPC_BATCH_WPC_XQ", 20), wpc_x2 = env_int("MPC_BATCH_WPC_X2", 12),
                     wpc_r2 = env_int("MPC_BATCH_WPC_R2", 16), wpc_plan = env_int("MPC_BATCH_WPC_PLAN", 16), wpc_x1 = env_int("MPC_BATCH_WPC_X1", 16), use_shares = env_int("MPC_BATCH_SHARES", 1), w_share = env_int("MPC_BATCH_WSHARE", 1);
    for (int g0 = 0; g0 < B && use_shares;) {
        int g1 = g0 + 1;
        while (g1 < B && group_key(members[g1]) == group_key(members[g0])) ++g1;
        double n_sum = 0.0;
        for (int i = g0; i < g1; ++i) n_sum += (double)members[i].n;
        for (int i = g0; i < g1; ++i) {
            BatchMember &m = members[i];
            const double f = n_sum > 0.0 ? (double)m.n / n_sum : 1.0;
            auto share = [&](int wpc) { return (int)std::max(4.0, std::ceil(f * wpc * m.n_cu)); };
            m.r2_wcap = w_share ? share(wpc_r2) : 0;
            m.th_blocks = share(wpc_th); m.xq_blocks = share(wpc_xq); m.x2_blocks = share(wpc_x2); m.r2_blocks = share(wpc_r2);
            m.plan_blocks = share(wpc_plan); m.x1_blocks = share(wpc_x1);
        }
        g0 = g1;
    }
    std::memcpy(g_tab_host, members, (size_t)B * sizeof(BatchMember));
    TRY(hipMemcpyAsync(g_tab_dev, g_tab_host, (size_t)B * sizeof(BatchMember), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(m_zero, dim3(64, (unsigned)B), dim3(256), 0, st, g_tab_dev);
    TRY(hipGetLastError());
    // ---- one launch sequence per group ----------------------------------------------------------------------------------------
    for (int g0 = 0; g0 < B;) {
        int g1 = g0 + 1;
        while (g1 < B && group_key(members[g1]) == group_key(members[g0])) ++g1;
        const BatchMember &r = members[g0];           // the group's representative
        const BatchMember *tab = g_tab_dev + g0;
        const unsigned G = (unsigned)(g1 - g0);
        long long n_max = 0;
        int grid_f = 1, grid_r2 = 1, lds_f = 0, lds_v = 0, lds_r2 = 0, rsplit = 1;
        int th_max = 0, xq_max = 0, x2_max = 0, r2_max = 0;     // largest share of the group (0: no shares, the single program's widths)
        for (int i = g0; i < g1; ++i) {
            th_max = std::max(th_max, members[i].th_blocks); xq_max = std::max(xq_max, members[i].xq_blocks);
            x2_max = std::max(x2_max, members[i].x2_blocks); r2_max = std::max(r2_max, members[i].r2_blocks);
            n_max = std::max(n_max, members[i].n);
            grid_f = std::max(grid_f, members[i].grid_f); grid_r2 = std::max(grid_r2, members[i].grid_r2);
            lds_f = std::max(lds_f, members[i].lds_f); lds_v = std::max(lds_v, members[i].lds_v); lds_r2 = std::max(lds_r2, members[i].lds_r2);
            rsplit = std::max(rsplit, members[i].rsplit_max);
        }
        const unsigned blocks256 = (unsigned)((n_max + 255) / 256);
        // KKT solves + box screen, theta stage
        if (r.use_kkt) {
            // BATCH_KKT_SPREAD lanes per candidate (k_kkt_thread's comment) while the group's launch stays short of the chip's thread slots
            const bool spread = r.use_kkt == 2 && (long long)n_max * G * BATCH_KKT_SPREAD <= BATCH_KKT_SPREAD_THREADS;
            const dim3 g(spread ? (unsigned)((n_max * BATCH_KKT_SPREAD + 255) / 256) : blocks256, G), b(256);
#define MPC_LAUNCH_KKT_(K_, SP_) if (r.fast_t >= 4) hipLaunchKernelGGL((m_kkt_thread<K_, 10, SP_>), g, b, 0, st, tab); \
                                    else if (r.fast_t >= 2) hipLaunchKernelGGL((m_kkt_thread<K_, 8, SP_>), g, b, 0, st, tab); \
                                    else hipLaunchKernelGGL((m_kkt_thread<K_, 4, SP_>), g, b, 0, st, tab)
#define MPC_LAUNCH_KKT(K_) case K_: MPC_LAUNCH_KKT_(K_, 1); break
#define MPC_LAUNCH_KKT_S(K_) case K_: if (spread) { MPC_LAUNCH_KKT_(K_, BATCH_KKT_SPREAD); } else { MPC_LAUNCH_KKT_(K_, 1); } break
            switch (r.kd) { MPC_LAUNCH_KKT_S(1); MPC_LAUNCH_KKT_S(2); MPC_LAUNCH_KKT_S(3); MPC_LAUNCH_KKT_S(4); MPC_LAUNCH_KKT_S(5); MPC_LAUNCH_KKT_S(6); MPC_LAUNCH_KKT(7); MPC_LAUNCH_KKT(8); }
#undef MPC_LAUNCH_KKT_S
#undef MPC_LAUNCH_KKT
#undef MPC_LAUNCH_KKT_
            if (!r.kkt_listed) hipLaunchKernelGGL(m_compact_small, dim3(1, G), dim3(1024), 0, st, tab, ST_TODO, ST_TODO, 0, 0);
        }
        {
            const dim3 g((unsigned)std::min<long long>(n_max, th_max > 0 ? th_max : grid_f), G), b(64);
#define MPC_LAUNCH_TH(NT_, SL_) do { TRY(raise_lds(m_theta2<NT_, SL_>, lds_f)); hipLaunchKernelGGL((m_theta2<NT_, SL_>), g, b, lds_f, st, tab); } while (0)
            switch (r.fast_t) {
                case 0: MPC_LAUNCH_TH(4, 1); break;
                case 1: MPC_LAUNCH_TH(4, 2); break;
                case 2: MPC_LAUNCH_TH(8, 1); break;
                case 3: MPC_LAUNCH_TH(8, 2); break;
                case 4: MPC_LAUNCH_TH(10, 1); break;
                default: MPC_LAUNCH_TH(10, 2); break;
            }
#undef MPC_LAUNCH_TH
        }
        TRY(raise_lds(m_verdict, lds_v));
        hipLaunchKernelGGL(m_partition_small, dim3(1, G), dim3(1024), 0, st, tab, spec_of({{ST_RETRY, 0}}), 12);
        hipLaunchKernelGGL(m_verdict, dim3((unsigned)std::min<long long>(n_max, 128), G), dim3(64), lds_v, st, tab, 12);
        hipLaunchKernelGGL(m_partition_small, dim3(1, G), dim3(1024), 0, st, tab,
                           spec_of({{ST_FEASIBLE, 1}, {ST_OPT_PENDING, 2}, {ST_NEEDX, 3}, {ST_NEEDX_SING, 3}}), 4);
        TRY(hipGetLastError());
        hipLaunchKernelGGL(m_zero_counter, dim3(1, G), dim3(64), 0, st, tab, 0);
        // (x,theta) stage
        if (r.quick_test) {
            const dim3 gg((unsigned)std::min<long long>(n_max, xq_max > 0 ? xq_max : (long long)r.n_cu * 32), G), bb(64);
            if (r.fast_x & 1) hipLaunchKernelGGL((m_xq<2>), gg, bb, 0, st, tab);
            else hipLaunchKernelGGL((m_xq<1>), gg, bb, 0, st, tab);
            hipLaunchKernelGGL(m_compact_small, dim3(1, G), dim3(1024), 0, st, tab, ST_NEEDX, ST_NEEDX_SING, 1, 8);
        }
        {
            bool any_plan = false;
            int pl_max = 0, x1_max = 0;
            for (int i = g0; i < g1; ++i) { any_plan = any_plan || members[i].plan; pl_max = std::max(pl_max, members[i].plan_blocks); x1_max = std::max(x1_max, members[i].x1_blocks); }
            if (any_plan) {
                const dim3 gp((unsigned)std::max<long long>(1, std::min<long long>((n_max + 63) / 64, pl_max > 0 ? pl_max : (long long)r.n_cu * 16)), G), bb(64);
                hipLaunchKernelGGL(m_xq_plan, gp, bb, 0, st, tab);
                const dim3 g1((unsigned)std::max<long long>(1, std::min<long long>(n_max, x1_max > 0 ? x1_max : (long long)r.n_cu * 16)), G);
                if (r.fast_x & 1) hipLaunchKernelGGL((m_x1<2>), g1, bb, 0, st, tab);
                else hipLaunchKernelGGL((m_x1<1>), g1, bb, 0, st, tab);
                TRY(hipGetLastError());
            }
        }
        {
            const dim3 gg((unsigned)std::min<long long>(n_max, x2_max > 0 ? x2_max : (long long)r.n_cu * 16), G), bb(64);
            switch (r.fast_x) {
                case 0: hipLaunchKernelGGL((m_x2<16, 1>), gg, bb, 0, st, tab); break;
                case 1: hipLaunchKernelGGL((m_x2<16, 2>), gg, bb, 0, st, tab); break;
                case 2: hipLaunchKernelGGL((m_x2<32, 1>), gg, bb, 0, st, tab); break;
                default: hipLaunchKernelGGL((m_x2<32, 2>), gg, bb, 0, st, tab); break;
            }
        }
        TRY(hipGetLastError());
        hipLaunchKernelGGL(m_partition_small, dim3(1, G), dim3(1024), 0, st, tab, spec_of({{ST_RETRY, 0}}), 24);
        hipLaunchKernelGGL(m_verdict, dim3((unsigned)std::min<long long>(n_max, 128), G), dim3(64), lds_v, st, tab, 24);
        // region stage, behind the (x,theta) stage: every optimal candidate of the level is known -- those of the theta stage and those a
        // re-solve of a doubtful (x,theta) run has just found -- so one launch covers them all (class 2 of the partition at [16..19];
        // a single program overlaps its region stage with the (x,theta) stage instead and handles late candidates separately: with
        // other members filling the device there is nothing to gain from that here)
        {
            bool any_open = false;
            for (int i = g0; i < g1; ++i) any_open = any_open || members[i].theta_open;
            if (any_open) {
                TRY(raise_lds(m_recession, lds_v));
                hipLaunchKernelGGL(m_recession, dim3((unsigned)std::min<long long>(n_max, 256), G), dim3(64), lds_v, st, tab);
            }
        }
        hipLaunchKernelGGL(m_partition_small, dim3(1, G), dim3(1024), 0, st, tab, spec_of({{ST_OPT_PENDING, 2}}), 16);
        TRY(hipGetLastError());
        {
            const dim3 g((unsigned)std::min<long long>(n_max * std::max(rsplit, 1), r2_max > 0 ? r2_max : grid_r2), G), b(64);
#define MPC_LAUNCH_R2(NT_, SL_) do { TRY(raise_lds(m_region2<NT_, SL_>, lds_r2)); hipLaunchKernelGGL((m_region2<NT_, SL_>), g, b, lds_r2, st, tab); } while (0)
            switch (r.fast_r) {
                case 0: MPC_LAUNCH_R2(4, 1); break;
                case 1: MPC_LAUNCH_R2(4, 2); break;
                case 2: MPC_LAUNCH_R2(8, 1); break;
                case 3: MPC_LAUNCH_R2(8, 2); break;
                case 4: MPC_LAUNCH_R2(10, 1); break;
                default: MPC_LAUNCH_R2(10, 2); break;
            }
#undef MPC_LAUNCH_R2
            TRY(hipGetLastError());
            // what k_region2 gave up on (rare): LDS-engine kernel, fixed-layout records in the member's retry slots
            int lds_r = 0, rcap = 1;
            for (int i = g0; i < g1; ++i) { lds_r = std::max(lds_r, members[i].lds_r); rcap = std::max(rcap, members[i].rcap); }
            TRY(raise_lds(m_region_v1<RG_FACET>, lds_r));
            TRY(raise_lds(m_region_v1<RG_ASSEMBLE>, lds_r));
            hipLaunchKernelGGL(m_compact_small, dim3(1, G), dim3(1024), 0, st, tab, ST_RRETRY, ST_RRETRY, 1, 28);
            hipLaunchKernelGGL(m_zero_counter, dim3(1, G), dim3(64), 0, st, tab, 1);
            hipLaunchKernelGGL(m_region_v1<RG_FACET>, dim3((unsigned)std::min(rcap * 8, 512), G), dim3(64), lds_r, st, tab);
            hipLaunchKernelGGL(m_zero_counter, dim3(1, G), dim3(64), 0, st, tab, 1);
            hipLaunchKernelGGL(m_region_v1<RG_ASSEMBLE>, dim3((unsigned)std::min(rcap, 128), G), dim3(64), lds_r, st, tab);
            TRY(hipGetLastError());
        }
        // pruned masks of this level + children
        if (r.mw == 2) hipLaunchKernelGGL(m_pruned_append<2>, dim3(blocks256, G), dim3(256), 0, st, tab);
        else hipLaunchKernelGGL(m_pruned_append<4>, dim3(blocks256, G), dim3(256), 0, st, tab);
        if (r.gen_children) {
            if (r.mw == 2) hipLaunchKernelGGL(m_children_count<2>, dim3((unsigned)n_max, G), dim3(64), 0, st, tab);
            else hipLaunchKernelGGL(m_children_count<4>, dim3((unsigned)n_max, G), dim3(64), 0, st, tab);
            hipLaunchKernelGGL(m_scan_small, dim3(1, G), dim3(SCAN_BLOCK), 0, st, tab);
            hipLaunchKernelGGL(m_children_write, dim3((unsigned)n_max, G), dim3(64), 0, st, tab);
        }
        hipLaunchKernelGGL(m_histogram, dim3(std::min(blocks256, 1024u), G), dim3(256), 0, st, tab);
        hipLaunchKernelGGL(m_publish, dim3(1, G), dim3(128), 0, st, tab);
        TRY(hipGetLastError());
        g0 = g1;
    }
    return hipSuccess;
}

}  // namespace mpc
