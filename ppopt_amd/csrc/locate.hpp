// locate.hpp -- point location over the critical regions of a Solution, and evaluation of x*(theta), batched (gfx950).
//
// Reference: Solution.get_region / evaluate (solution.py:45-112): a loop over the regions calling
// CriticalRegion.is_inside, all(E theta - f < tol) (critical_region.py:83-86); without overlap the first region of the list
// that contains theta wins, with overlap the containing region with the lowest objective (ties: the later one).
//
// Mapping: one LANE per query point, the wavefront walks the region list.  Region and row indices are wave-uniform, so a
// row [f | E] is fetched with scalar loads and applied to 64 points at once; no gathers, no divergence except the exit
// masks.  A region is left as soon as no lane of the wave is still inside it, the list as soon as every lane has its
// region (first-match mode).  The stacked rows (13.7 MB for the 9,432 regions of config 4) stay in L2 / Infinity Cache.
#pragma once
#include <stdint.h>

namespace mpc {

// Rows are staged through LDS in tiles of LOC_TILE rows, loaded cooperatively (coalesced) by the 256 points of a block;
// with each row come the region it belongs to and the index of the first row of the next region, so that a wavefront
// which has no lane left inside a region jumps over the rest of its rows.
constexpr int LOC_TILE = 256;
template <int NT>
__global__ void __launch_bounds__(256) k_locate(long long m, int nt, int nx, long long n_regions, long long n_rows,
                                                const int32_t *__restrict__ row_region, const int32_t *__restrict__ row_end,
                                                const double *__restrict__ ef, const double *__restrict__ xlaw,
                                                const double *__restrict__ Q, const double *__restrict__ cvec, const double *__restrict__ H,
                                                const double *__restrict__ theta, double tol, int overlapping, int inclusive,
                                                long long *__restrict__ region_out) {
    __shared__ double tile[LOC_TILE][NT + 1];
    __shared__ int trid[LOC_TILE], tend[LOC_TILE];
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int nr = nt + 1;
    double th[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) th[t] = (p < m && t < nt) ? theta[p * nt + t] : 0.0;
    long long found = -1;
    double best = INFINITY;
    bool alive = p < m, inside = false;
    int cur = -1;
    long long i = 0;   // next row of this wavefront (wave-uniform)
    // the point is inside every row of region `cur`: first match, or candidate for the lowest objective
    auto commit = [&]() {
        if (cur < 0 || !inside) return;
        if (!overlapping) { found = cur; alive = false; return; }
        // objective 1/2 x'Qx + theta'H'x + c'x at x = A theta + b (terms without x are the same for every region)
        const double *xl = xlaw + (size_t)cur * nx * nr;
        double obj = 0.0;
        for (int a = 0; a < nx; ++a) {
            double xa = xl[a * nr];
#pragma unroll
            for (int t = 0; t < NT; ++t) if (t < nt) xa = fma(xl[a * nr + 1 + t], th[t], xa);
            double g = cvec ? cvec[a] : 0.0;
            if (H) {
#pragma unroll
                for (int t = 0; t < NT; ++t) if (t < nt) g = fma(H[a * nt + t], th[t], g);
            }
            if (Q) {
                double qx = 0.0;
                for (int j = 0; j < nx; ++j) {
                    double xj = xl[j * nr];
#pragma unroll
                    for (int t = 0; t < NT; ++t) if (t < nt) xj = fma(xl[j * nr + 1 + t], th[t], xj);
                    qx = fma(Q[a * nx + j], xj, qx);
                }
                g = fma(0.5, qx, g);
            }
            obj = fma(g, xa, obj);
        }
        if (obj <= best) { best = obj; found = cur; }
    };
    for (long long tile0 = 0; tile0 < n_rows; tile0 += LOC_TILE) {
        __syncthreads();
        {
            const long long row = tile0 + threadIdx.x;
            if (row < n_rows) {
#pragma unroll
                for (int t = 0; t <= NT; ++t) tile[threadIdx.x][t] = t <= nt ? ef[row * nr + t] : 0.0;
                trid[threadIdx.x] = row_region[row];
                tend[threadIdx.x] = row_end[row];
            }
        }
        __syncthreads();
        const long long tile1 = tile0 + LOC_TILE < n_rows ? tile0 + LOC_TILE : n_rows;
        while (i < tile1) {
            const int li = (int)(i - tile0);
            const int rid = trid[li];
            if (rid != cur) {
                commit();
                cur = rid;
                inside = overlapping ? (p < m) : alive;
                if (!overlapping && !__any(alive)) { i = n_rows; break; }   // every lane has its region
            }
            if (inclusive) {
                // E theta <= f + tol with the product formed first and then compared, like `A @ theta <= b` of the reference's
                // PointLocation (upop/point_location.py:46,59): a point exactly on a facet belongs to the region
                double v = 0.0;
#pragma unroll
                for (int t = 0; t < NT; ++t) v = fma(tile[li][1 + t], th[t], v);
                inside = inside && (v <= tile[li][0] + tol);
            } else {
                double v = -tile[li][0];
#pragma unroll
                for (int t = 0; t < NT; ++t) v = fma(tile[li][1 + t], th[t], v);
                inside = inside && (v < tol);
            }
            if (!__any(inside)) { i = tend[li]; continue; }   // nobody is left in this region: on to the next one
            ++i;
        }
    }
    commit();
    if (p < m) region_out[p] = found;
}

// ------------------------------------------------------------------------------------------------------------------
// k_locate_walk: point location by walking through ADJACENT regions (complete, non-overlapping solutions with many regions).
//
// The list scan above costs every point a pass over all rows of all regions it is not in -- 4.1e6 rows for the 227,349
// regions of the complete config 4.  The regions of an mpQP solution are glued along their facets, and a facet knows what
// lies behind it: the row of a multiplier lambda_a >= 0 is shared with the region whose active set lacks a, the row of an
// inactive constraint c with the region whose active set has c in addition (Bemporad et al. 2002; the reference's graph
// algorithm moves the same way, mpqp_graph.py:97-106), a row of A_t theta <= b_t with nothing.  One lane per point: test the
// rows of the current region; if all hold the point is located, otherwise cross the most violated row to the neighbour, whose
// region index comes from a binary search of its active-set mask in the sorted mask table.  A point that meets a row of the
// parameter set is outside the solution (-1); a walk that finds no neighbour behind any violated row (not even with a second
// row exchanged, the degenerate case), meets a region without facet information or reaches the step limit leaves the point
// UNRESOLVED (-2), and the host hands it to k_locate_few / the list scan.
// To return the region the scan would return (the FIRST containing region of the list), a located point is also offered to
// the neighbours across the rows it satisfies by less than 2*tol, if their index is smaller.
// row_info[row] = kind << 16 | id:  kind 0 multiplier row of active constraint id, 1 inactive constraint id, 2 A_t row, 3 unknown.
template <int NT, int MW>
__global__ void __launch_bounds__(256) k_locate_walk(long long m, int nt, long long n_regions, const long long *__restrict__ row_off,
                                                     const double *__restrict__ ef, const int32_t *__restrict__ row_info,
                                                     const unsigned long long *__restrict__ masks,          // [n_regions][MW], region order
                                                     const unsigned long long *__restrict__ sorted_masks,   // [n_regions][MW], ascending
                                                     const int32_t *__restrict__ sorted_region,             // region of sorted_masks[i]
                                                     const double *__restrict__ theta, double tol, int start_region, int max_steps,
                                                     int n_c, long long *__restrict__ region_out) {
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= m) return;
    const int nr = nt + 1;
    double th[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) th[t] = t < nt ? theta[p * nt + t] : 0.0;
    auto lookup = [&](const unsigned long long (&key)[MW]) -> long long {
        long long lo = 0, hi = n_regions;
        while (lo < hi) {
            const long long mid = (lo + hi) >> 1;
            bool less = false, decided = false;
#pragma unroll
            for (int j = MW - 1; j >= 0; --j) {
                const unsigned long long v = sorted_masks[mid * MW + j];
                if (!decided && v != key[j]) { less = v < key[j]; decided = true; }
            }
            if (less) lo = mid + 1; else hi = mid;
        }
        if (lo >= n_regions) return -1;
#pragma unroll
        for (int j = 0; j < MW; ++j) if (sorted_masks[lo * MW + j] != key[j]) return -1;
        return sorted_region[lo];
    };
    // worst row of region r at theta (value and row); returns false if the region has a row of unknown kind
    auto worst_row = [&](long long r, double &worst, long long &wrow) {
        worst = -INFINITY; wrow = -1;
        for (long long i = row_off[r]; i < row_off[r + 1]; ++i) {
            double v = -ef[i * nr];
#pragma unroll
            for (int t = 0; t < NT; ++t) if (t < nt) v = fma(ef[i * nr + 1 + t], th[t], v);
            if (v > worst) { worst = v; wrow = i; }
        }
    };
    auto neighbour = [&](long long r, long long row) -> long long {   // -1 outside, -2 unknown
        const int info = row_info[row], kind = info >> 16, id = info & 0xffff;
        if (kind == 2) return -1;
        if (kind == 3) return -2;
        unsigned long long key[MW];
#pragma unroll
        for (int j = 0; j < MW; ++j) key[j] = masks[r * MW + j];
#pragma unroll
        for (int j = 0; j < MW; ++j) if ((id >> 6) == j) key[j] ^= 1ull << (id & 63);
        const long long q = lookup(key);
        return q < 0 ? -2 : q;
    };
    long long r = start_region, prev = -1, found = -2;
    for (int step = 0; step < max_steps; ++step) {
        // the violated rows of region r in decreasing order of violation, until one has a region behind it.  (Redundancy removal
        // keeps rows that only touch the region in a lower-dimensional face -- mpqp_utils.py:143-178 keeps weakly redundant
        // rows -- and nothing lies behind those; a true facet of a complete solution has a neighbour unless it is part of the
        // boundary of the feasible parameter set.)
        double bound = INFINITY;
        long long next = -2;
        bool any_violated = false, met_omega = false, met_unknown = false;
        for (;;) {
            double worst = -INFINITY; long long wrow = -1;
            for (long long i = row_off[r]; i < row_off[r + 1]; ++i) {
                double v = -ef[i * nr];
#pragma unroll
                for (int t = 0; t < NT; ++t) if (t < nt) v = fma(ef[i * nr + 1 + t], th[t], v);
                if (v > worst && v < bound) { worst = v; wrow = i; }
            }
            if (wrow < 0 || worst < tol) break;          // no (further) violated row
            any_violated = true;
            const long long q = neighbour(r, wrow);
            if (q >= 0 && q != prev) { next = q; break; }   // never straight back: two regions that both see the point behind their common facet
            if (q == -1) { met_omega = true; if (worst > 10.0 * tol) break; }   // clearly outside the parameter set: no region can contain the point
            else if ((row_info[wrow] >> 16) >= 3) met_unknown = true;
            bound = worst;
        }
        if (!any_violated) { found = row_off[r + 1] > row_off[r] ? r : -2; break; }
        if (next >= 0) { prev = r; r = next; continue; }
        // Every violated row leads nowhere with one row added or removed.  Where the constraint gradients are dependent the
        // region behind a facet differs by TWO rows (one enters, one leaves): try, for the most violated rows, the neighbour's
        // active set with one further row exchanged, and take the first that exists and is violated less than this region.
        if (!met_omega) {
            double vbound = INFINITY;
            for (int attempt = 0; attempt < 3 && next < 0; ++attempt) {
                double worst = -INFINITY; long long wrow = -1;
                for (long long i = row_off[r]; i < row_off[r + 1]; ++i) {
                    double v = -ef[i * nr];
#pragma unroll
                    for (int t = 0; t < NT; ++t) if (t < nt) v = fma(ef[i * nr + 1 + t], th[t], v);
                    if (v > worst && v < vbound) { worst = v; wrow = i; }
                }
                if (wrow < 0 || worst < tol) break;
                vbound = worst;
                const int info = row_info[wrow], kind = info >> 16, id = info & 0xffff;
                if (kind >= 2) continue;
                unsigned long long base[MW];
#pragma unroll
                for (int j = 0; j < MW; ++j) base[j] = masks[r * MW + j];
#pragma unroll
                for (int j = 0; j < MW; ++j) if ((id >> 6) == j) base[j] ^= 1ull << (id & 63);
                for (int c2 = 0; c2 < n_c && next < 0; ++c2) {
                    if (c2 == id) continue;
                    const bool in_set = (base[c2 >> 6] >> (c2 & 63)) & 1ull;
                    if ((kind == 1) != in_set) continue;          // a row entered: one of the others leaves; a row left: another enters
                    unsigned long long key[MW];
#pragma unroll
                    for (int j = 0; j < MW; ++j) key[j] = base[j];
                    key[c2 >> 6] ^= 1ull << (c2 & 63);
                    const long long q = lookup(key);
                    if (q >= 0 && q != r) {
                        double w; long long wr;
                        worst_row(q, w, wr);
                        if (wr >= 0 && w < worst) next = q;       // strictly less violated: the walk cannot return here through this move
                    }
                }
            }
        }
        if (next >= 0) { prev = r; r = next; continue; }
        found = met_omega ? -1 : -2;
        break;
    }
    if (found >= 0) {
        // first-match rule of the list scan: an earlier region that also contains the point within the tolerance wins
        long long best = found;
        for (long long i = row_off[found]; i < row_off[found + 1]; ++i) {
            double v = -ef[i * nr];
#pragma unroll
            for (int t = 0; t < NT; ++t) if (t < nt) v = fma(ef[i * nr + 1 + t], th[t], v);
            if (v > -2.0 * tol) {
                const long long q = neighbour(found, i);
                if (q >= 0 && q < best) {
                    double w; long long wr;
                    worst_row(q, w, wr);
                    if (wr >= 0 && w < tol) best = q;
                }
            }
        }
        found = best;
    }
    region_out[p] = found;
}

// The few points the walk left unresolved: one thread per (point, region), the first containing region by atomicMin.
// Work = points x rows, spread over the whole device -- the list scan would serialise 4e6 rows behind one wavefront.
template <int NT>
__global__ void __launch_bounds__(256) k_locate_few(long long n_pts, int nt, long long n_regions, const long long *__restrict__ row_off,
                                                    const double *__restrict__ ef, const double *__restrict__ theta, double tol,
                                                    long long *__restrict__ region_out) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long p = blockIdx.y;
    if (r >= n_regions || p >= n_pts) return;
    const int nr = nt + 1;
    double th[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) th[t] = t < nt ? theta[p * nt + t] : 0.0;
    const long long i0 = row_off[r], i1 = row_off[r + 1];
    if (i1 <= i0) return;
    bool inside = true;
    for (long long i = i0; i < i1 && inside; ++i) {
        double v = -ef[i * nr];
#pragma unroll
        for (int t = 0; t < NT; ++t) if (t < nt) v = fma(ef[i * nr + 1 + t], th[t], v);
        inside = v < tol;
    }
    if (inside) atomicMin(reinterpret_cast<unsigned long long *>(region_out + p), (unsigned long long)r);
}

// x*(theta) = A theta + b of the region each point was located in (NaN where there is none)
__global__ void __launch_bounds__(256) k_evaluate(long long m, int nt, int nx, const double *__restrict__ xlaw, const double *__restrict__ theta,
                                                  const long long *__restrict__ region, double *__restrict__ x) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= m * nx) return;
    const long long p = idx / nx;
    const int i = (int)(idx - p * nx);
    const long long r = region[p];
    if (r < 0) { x[idx] = __longlong_as_double(0x7ff8000000000000ll); return; }
    const int nr = nt + 1;
    const double *row = xlaw + ((size_t)r * nx + i) * nr;
    double v = row[0];
    for (int t = 0; t < nt; ++t) v = fma(row[1 + t], theta[p * nt + t], v);
    x[idx] = v;
}

}  // namespace mpc
