// graph.hpp -- device-side bookkeeping of the connected-graph traversals (gfx950).
//
// Reference: mp_solvers/mpqp_combi_graph.py:68-145 (sets S "to visit" and E "ever queued", explore_subset / explore_superset)
// and mp_solvers/mpqp_graph.py:38-108 (attempted, generate_reduce / generate_extra, solver_utils.py:68-106).  The reference
// keeps Python sets of tuples and visits one active set at a time; here an active set is a bit mask of MW 64-bit words, the
// traversal advances a whole WAVE of sets at a time, and S / E never leave HBM:
//
//   wave      masks sorted by (cardinality, mask): one contiguous group per cardinality = one frontier of the level kernels
//   pending   neighbours emitted by the groups of the wave (count -> scan -> emit: positions are deterministic)
//   close     radix sort of pending (rocPRIM, 64*MW-bit keys), first occurrences that a binary search does not find in
//             `visited` are the next wave; visited := sort(visited ++ new)
//
// Every array is sorted, so the traversal -- and the order of the regions it returns -- is deterministic.
#pragma once
#include <stdint.h>

namespace mpc {

template <int MW> struct GMask { unsigned long long w[MW]; };

template <int MW>
__device__ __forceinline__ bool gmask_less(const GMask<MW> &a, const GMask<MW> &b) {
#pragma unroll
    for (int j = MW - 1; j >= 0; --j) { if (a.w[j] != b.w[j]) return a.w[j] < b.w[j]; }
    return false;
}
template <int MW>
__device__ __forceinline__ bool gmask_eq(const GMask<MW> &a, const GMask<MW> &b) {
    bool e = true;
#pragma unroll
    for (int j = 0; j < MW; ++j) e = e && a.w[j] == b.w[j];
    return e;
}
template <int MW>
__device__ __forceinline__ int gmask_popc(const GMask<MW> &a) {
    int c = 0;
#pragma unroll
    for (int j = 0; j < MW; ++j) c += __popcll(a.w[j]);
    return c;
}

// cardinality of every mask (as the 32-bit sort key of the grouping pass) and the identity permutation
template <int MW>
__global__ void k_g_card(const GMask<MW> *__restrict__ m, long long n, unsigned int *__restrict__ card, unsigned int *__restrict__ idx) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    card[i] = (unsigned int)gmask_popc<MW>(m[i]);
    idx[i] = (unsigned int)i;
}
template <int MW>
__global__ void k_g_gather(const GMask<MW> *__restrict__ src, const unsigned int *__restrict__ perm, long long n, GMask<MW> *__restrict__ dst) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[perm[i]];
}
// group boundaries of the SORTED cardinalities: first[k] = index of the first mask with k rows (untouched = -1: no such mask).
// (A histogram with one atomic per mask serialises on the few distinct cardinalities of a wave: 0.46 ms per call.)
__global__ void k_g_first(const unsigned int *__restrict__ card, long long n, int32_t *__restrict__ first) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned int c = card[i] > 256u ? 256u : card[i];
    if (i == 0 || (card[i - 1] > 256u ? 256u : card[i - 1]) != c) first[c] = (int32_t)i;
}

// masks with exactly k rows -> sorted index lists (the frontier layout of the level kernels)
template <int MW>
__global__ void k_g_frontier(const GMask<MW> *__restrict__ m, long long n, int k, int32_t *__restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const GMask<MW> a = m[i];
    int32_t *o = out + (size_t)i * k;
    int p = 0;
#pragma unroll
    for (int j = 0; j < MW; ++j) {
        unsigned long long v = a.w[j];
        while (v && p < k) { const int b = __ffsll((long long)v) - 1; v &= v - 1; o[p++] = 64 * j + b; }
    }
}

// variant 1 (mpqp_graph.py): the facet constraints of every region of the group -- CriticalRegion.regular_set[1], the inactive
// constraints whose rows are facets -- as one mask per candidate.  Slot form (k_region2): head_i row = status cand nE n_om n_la
// n_re e_off 0 | active[k] | omega[n_tc] | lambda[k] | reg_idx[n_c-k] | reg_con[n_c-k].
template <int MW>
__global__ void k_g_facets_slots(const int32_t *__restrict__ head_i, int fi, long long n_slots, int k, int n_c, int n_tc,
                                 GMask<MW> *__restrict__ facet) {
    const long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_slots) return;
    const int32_t *hi = head_i + (size_t)s * fi;
    if (hi[0] != 3) return;   // MPC_REGION
    const int32_t *rcon = hi + 8 + k + n_tc + k + (n_c - k);
    GMask<MW> f;
#pragma unroll
    for (int j = 0; j < MW; ++j) f.w[j] = 0;
    for (int r = 0; r < hi[5]; ++r) {
        const int c = rcon[r];
#pragma unroll
        for (int j = 0; j < MW; ++j) if ((c >> 6) == j) f.w[j] |= 1ull << (c & 63);
    }
    facet[hi[1]] = f;
}
// the same for regions that the LDS-engine kernel produced (fixed-stride records: k nE n_om n_la n_re | active[n_c] |
// omega[n_tc] | lambda[n_c] | reg_idx[n_c] | reg_con[n_c]); list[r] = candidate of record r
template <int MW>
__global__ void k_g_facets_fixed(const int32_t *__restrict__ rec_i, long long stride, const int32_t *__restrict__ list, long long n_list,
                                 const uint8_t *__restrict__ status, int n_c, int n_tc, GMask<MW> *__restrict__ facet) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_list) return;
    const int cand = list[r];
    if (status[cand] != 3) return;
    const int32_t *ri = rec_i + (size_t)r * stride;
    const int32_t *rcon = ri + 5 + n_c + n_tc + n_c + n_c;
    GMask<MW> f;
#pragma unroll
    for (int j = 0; j < MW; ++j) f.w[j] = 0;
    for (int q = 0; q < ri[4]; ++q) {
        const int c = rcon[q];
#pragma unroll
        for (int j = 0; j < MW; ++j) if ((c >> 6) == j) f.w[j] |= 1ull << (c & 63);
    }
    facet[cand] = f;
}

// The neighbours one visited set hands on.  status: the level's verdict (values above MPC_REGION count as MPC_FEASIBLE).
//   variant 0 (mpqp_combi_graph.py:114-143, MPC_LEVEL_GRAPH verdicts): rank deficient -> subsets; region non-empty (2, 3) ->
//             subsets and all supersets
//   variant 1 (mpqp_graph.py:69-108, full verdicts): everything but "optimal, lower dimensional" -> subsets; region -> also
//             the supersets through its facet constraints
// shrink never removes a program equality (eq); grow adds rows < n_c that are not in the set.
template <int MW>
__device__ __forceinline__ void g_rules(int st, int variant, const GMask<MW> &a, const GMask<MW> &eq, const GMask<MW> &all,
                                        const GMask<MW> *facet, long long i, GMask<MW> &sub, GMask<MW> &sup) {
    if (st > 3) st = 1;
    const bool shrink = variant == 0 ? (st == 0 || st >= 2) : (st != 2);
    const bool grow = variant == 0 ? (st >= 2) : (st == 3);
#pragma unroll
    for (int j = 0; j < MW; ++j) {
        sub.w[j] = shrink ? (a.w[j] & ~eq.w[j]) : 0ull;
        sup.w[j] = grow ? ((variant == 0 ? all.w[j] : facet[i].w[j]) & ~a.w[j]) : 0ull;
    }
}
template <int MW>
__global__ void k_g_count(const GMask<MW> *__restrict__ m, long long n, const uint8_t *__restrict__ status, int const_status, int variant,
                          GMask<MW> eq, GMask<MW> all, const GMask<MW> *__restrict__ facet, int32_t *__restrict__ count) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    GMask<MW> sub, sup;
    g_rules<MW>(status ? status[i] : const_status, variant, m[i], eq, all, facet, i, sub, sup);
    count[i] = gmask_popc<MW>(sub) + gmask_popc<MW>(sup);
}
template <int MW>
__global__ void k_g_emit(const GMask<MW> *__restrict__ m, long long n, const uint8_t *__restrict__ status, int const_status, int variant,
                         GMask<MW> eq, GMask<MW> all, const GMask<MW> *__restrict__ facet, const int32_t *__restrict__ offset,
                         GMask<MW> *__restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const GMask<MW> a = m[i];
    GMask<MW> sub, sup;
    g_rules<MW>(status ? status[i] : const_status, variant, a, eq, all, facet, i, sub, sup);
    GMask<MW> *o = out + offset[i];
#pragma unroll
    for (int j = 0; j < MW; ++j) {
        unsigned long long v = sub.w[j];
        while (v) { const unsigned long long b = v & (~v + 1ull); v ^= b; GMask<MW> c = a; c.w[j] &= ~b; *o++ = c; }
    }
#pragma unroll
    for (int j = 0; j < MW; ++j) {
        unsigned long long v = sup.w[j];
        while (v) { const unsigned long long b = v & (~v + 1ull); v ^= b; GMask<MW> c = a; c.w[j] |= b; *o++ = c; }
    }
}

// sorted pending -> flag of the masks that enter the next wave: first of its run of equal keys, and not in `visited` (sorted)
template <int MW>
__global__ void k_g_newflags(const GMask<MW> *__restrict__ p, long long n, const GMask<MW> *__restrict__ visited, long long nv,
                             int32_t *__restrict__ flag) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const GMask<MW> a = p[i];
    bool fresh = i == 0 || !gmask_eq<MW>(a, p[i - 1]);
    if (fresh) {
        long long lo = 0, hi = nv;   // lower bound
        while (lo < hi) { const long long mid = (lo + hi) >> 1; if (gmask_less<MW>(visited[mid], a)) lo = mid + 1; else hi = mid; }
        fresh = !(lo < nv && gmask_eq<MW>(visited[lo], a));
    }
    flag[i] = fresh ? 1 : 0;
}
template <int MW>
__global__ void k_g_compact(const GMask<MW> *__restrict__ p, long long n, const int32_t *__restrict__ flag, const int32_t *__restrict__ pos,
                            GMask<MW> *__restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && flag[i]) out[pos[i]] = p[i];
}
__global__ void k_g_fill_u8(uint8_t *p, long long n, int v) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = (uint8_t)v;
}

}  // namespace mpc
