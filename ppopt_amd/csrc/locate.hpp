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

template <int NT>
__global__ void __launch_bounds__(256) k_locate(long long m, int nt, int nx, long long n_regions, const long long *__restrict__ row_off,
                                                const double *__restrict__ ef, const double *__restrict__ xlaw,
                                                const double *__restrict__ Q, const double *__restrict__ cvec, const double *__restrict__ H,
                                                const double *__restrict__ theta, double tol, int overlapping,
                                                long long *__restrict__ region_out) {
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int nr = nt + 1;
    double th[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) th[t] = (p < m && t < nt) ? theta[p * nt + t] : 0.0;
    long long found = -1;
    double best = INFINITY;
    bool alive = p < m;
    for (long long r = 0; r < n_regions; ++r) {
        if (!overlapping && !__any(alive)) break;   // first-match mode: this wavefront is done
        bool inside = alive;
        const long long r0 = row_off[r], r1 = row_off[r + 1];
        for (long long row = r0; row < r1; ++row) {
            const double *e = ef + row * nr;
            double v = -e[0];
#pragma unroll
            for (int t = 0; t < NT; ++t) if (t < nt) v = fma(e[1 + t], th[t], v);
            inside = inside && (v < tol);
            if (!__any(inside)) break;
        }
        if (!inside) continue;
        if (!overlapping) { found = r; alive = false; continue; }
        // overlapping regions: objective 1/2 x'Qx + theta'H'x + c'x at x = A theta + b (terms without x are the same for every region)
        const double *xl = xlaw + (size_t)r * nx * nr;
        double obj = 0.0;
        for (int i = 0; i < nx; ++i) {
            double xi = xl[i * nr];
#pragma unroll
            for (int t = 0; t < NT; ++t) if (t < nt) xi = fma(xl[i * nr + 1 + t], th[t], xi);
            double g = cvec ? cvec[i] : 0.0;
            if (H) {
#pragma unroll
                for (int t = 0; t < NT; ++t) if (t < nt) g = fma(H[i * nt + t], th[t], g);
            }
            if (Q) {
                double qx = 0.0;
                for (int j = 0; j < nx; ++j) {
                    double xj = xl[j * nr];
#pragma unroll
                    for (int t = 0; t < NT; ++t) if (t < nt) xj = fma(xl[j * nr + 1 + t], th[t], xj);
                    qx = fma(Q[i * nx + j], xj, qx);
                }
                g = fma(0.5, qx, g);
            }
            obj = fma(g, xi, obj);
        }
        if (obj <= best) { best = obj; found = r; }
    }
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
