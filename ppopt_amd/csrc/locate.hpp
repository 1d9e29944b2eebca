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
