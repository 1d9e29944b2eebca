// setup_mfma.hip -- the dense Hessian factor and the one-off Schur blocks of a program on the matrix cores (setup_mfma.hpp).
//
// One workgroup of four wavefronts per program; every product is a sequence of 16 x 16 x 4 fp64 MFMA tiles
// (v_mfma_f64_16x16x4_f64), the tiles of a phase are dealt round-robin to the four wavefronts, phases are separated by
// workgroup barriers.  All operands live in a per-program scratch block in HBM (tens of KB: L2-resident); the only LDS use is
// the 16 x 16 diagonal block of the Cholesky panel and its inverse.  The work per program is tiny (config 4: Q 20 x 20,
// A 47 x 20 -> about 150 MFMA instructions); the point of doing it here is that a batch of programs -- the sub-programs of
// the mixed-integer enumeration -- is ONE launch with no host arithmetic, and that the blocks are born in HBM.
//
// Separate translation unit: compiled on its own (seconds) and linked into libmpcombi_hip.so.
#include "setup_mfma.hpp"

namespace mpc {
namespace {

typedef double d4 __attribute__((ext_vector_type(4)));

// acc[m][n] += sum_{k < K} a(m, k) b(k, n) for one 16 x 16 tile, K a multiple of 4.
// Operand maps of v_mfma_f64_16x16x4_f64 (cdna_hip_programming.md): lane l supplies A[m = l & 15][k = l >> 4] and
// B[k = l >> 4][n = l & 15]; result register i of lane l is D[row = (l >> 4) + 4 i][col = l & 15].
template <class FA, class FB>
__device__ __forceinline__ d4 tile_mma(int K, FA a, FB b, int lane) {
    d4 acc = {0.0, 0.0, 0.0, 0.0};
    const int r = lane & 15, q = lane >> 4;
    for (int k0 = 0; k0 < K; k0 += 4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a(r, k0 + q), b(k0 + q, r), acc, 0, 0, 0);
    return acc;
}
template <class FS>
__device__ __forceinline__ void tile_store(const d4 &acc, FS store, int lane) {
    const int col = lane & 15, r0 = lane >> 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) store(r0 + 4 * i, col, acc[i]);
}

__device__ __forceinline__ void phase_sync() {
    __threadfence_block();
    __syncthreads();
}

__global__ void __launch_bounds__(256) k_setup_mfma(const SetupJob *__restrict__ jobs) {
    const SetupJob J = jobs[blockIdx.x];
    __shared__ double sD[16][17], sI[16][17];
    __shared__ double s_dmax;
    __shared__ int s_fail;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nx = J.nx, nt = J.nt, nc = J.nc, nr = nt + 1, NP = J.NP, MP = J.MP, RP = J.RP, NB = MP + RP, nb = NP / 16;
    double *Lq = J.work, *Dinv = Lq + (size_t)NP * NP, *Ap = Dinv + (size_t)nb * 256, *Bm = Ap + (size_t)MP * NP, *Ym = Bm + (size_t)NP * NB;

    // ---- padded operands ---------------------------------------------------------------------------------------------------
    for (int i = tid; i < MP * NP; i += 256) { const int r = i / NP, c = i % NP; Ap[i] = (r < nc && c < nx) ? J.A[(size_t)r * nx + c] : 0.0; }
    if (J.Q) {
        // symmetrised Q, identity on the padding; right-hand sides [A' | c | H]
        for (int i = tid; i < NP * NP; i += 256) {
            const int r = i / NP, c = i % NP;
            Lq[i] = (r < nx && c < nx) ? 0.5 * (J.Q[(size_t)r * nx + c] + J.Q[(size_t)c * nx + r]) : (r == c ? 1.0 : 0.0);
        }
        for (int i = tid; i < NP * NB; i += 256) {
            const int r = i / NB, c = i % NB;
            double v = 0.0;
            if (r < nx) {
                if (c < MP) { if (c < nc) v = J.A[(size_t)c * nx + r]; }
                else { const int t = c - MP; if (t < nr) v = t == 0 ? J.c[r] : J.H[(size_t)r * nt + t - 1]; }
            }
            Bm[i] = v;
        }
        if (tid == 0) {
            double dmax = 0.0;
            for (int i = 0; i < nx; ++i) dmax = fmax(dmax, fabs(J.Q[(size_t)i * nx + i]));
            s_dmax = dmax;
            s_fail = 0;
        }
    }
    phase_sync();

    // ---- Gram matrix A A' (rank screen of the candidates) ------------------------------------------------------------------
    {
        const int mt = MP / 16;
        int t = 0;
        for (int i = 0; i < mt; ++i)
            for (int j = 0; j <= i; ++j, ++t) {
                if ((t & 3) != wave) continue;
                const d4 acc = tile_mma(NP, [&](int m, int k) { return Ap[(size_t)(i * 16 + m) * NP + k]; },
                                        [&](int k, int n) { return Ap[(size_t)(j * 16 + n) * NP + k]; }, lane);
                tile_store(acc, [&](int m, int n, double v) {
                    const int r = i * 16 + m, c = j * 16 + n;
                    if (r < nc && c < nc) { J.AAT[(size_t)r * nc + c] = v; if (i != j) J.AAT[(size_t)c * nc + r] = v; }
                }, lane);
            }
    }
    if (!J.Q) { if (tid == 0) *J.flag = 1; return; }

    // ---- Q = L L': blocked right-looking Cholesky ---------------------------------------------------------------------------
    for (int kb = 0; kb < nb; ++kb) {
        { const int i = tid >> 4, l = tid & 15; sD[i][l] = Lq[(size_t)(kb * 16 + i) * NP + kb * 16 + l]; }
        __syncthreads();
        for (int j = 0; j < 16; ++j) {
            if (tid == 0) {
                double d = sD[j][j];
                if (kb * 16 + j < nx && !(d > 1e-10 * s_dmax)) { s_fail = 1; d = 1.0; }   // not positive definite (within the tolerance of the host test)
                sD[j][j] = sqrt(d);
            }
            __syncthreads();
            if (tid > j && tid < 16) sD[tid][j] /= sD[j][j];
            __syncthreads();
            { const int i = tid >> 4, l = tid & 15; if (i > j && l > j && l <= i) sD[i][l] -= sD[i][j] * sD[l][j]; }
            __syncthreads();
        }
        // inverse of the lower-triangular diagonal block, column c by forward substitution (one thread per column)
        if (tid < 16) {
            const int c = tid;
            for (int i = 0; i < 16; ++i) {
                double v = 0.0;
                if (i == c) v = 1.0 / sD[c][c];
                else if (i > c) { double s = 0.0; for (int m = c; m < i; ++m) s += sD[i][m] * sI[m][c]; v = -s / sD[i][i]; }
                sI[i][c] = v;
            }
        }
        __syncthreads();
        { const int i = tid >> 4, l = tid & 15; Lq[(size_t)(kb * 16 + i) * NP + kb * 16 + l] = l <= i ? sD[i][l] : 0.0; Dinv[(size_t)kb * 256 + i * 16 + l] = sI[i][l]; }
        // panel below the diagonal block:  L[ib, kb] = Q[ib, kb] inv(L[kb, kb])'
        for (int ib = kb + 1 + wave; ib < nb; ib += 4) {
            const d4 acc = tile_mma(16, [&](int m, int k) { return Lq[(size_t)(ib * 16 + m) * NP + kb * 16 + k]; },
                                    [&](int k, int n) { return sI[n][k]; }, lane);
            tile_store(acc, [&](int m, int n, double v) { Lq[(size_t)(ib * 16 + m) * NP + kb * 16 + n] = v; }, lane);
        }
        phase_sync();
        // trailing update of the lower triangle:  Q[ib, jb] -= L[ib, kb] L[jb, kb]'
        {
            int t = 0;
            for (int ib = kb + 1; ib < nb; ++ib)
                for (int jb = kb + 1; jb <= ib; ++jb, ++t) {
                    if ((t & 3) != wave) continue;
                    const d4 acc = tile_mma(16, [&](int m, int k) { return Lq[(size_t)(ib * 16 + m) * NP + kb * 16 + k]; },
                                            [&](int k, int n) { return Lq[(size_t)(jb * 16 + n) * NP + kb * 16 + k]; }, lane);
                    tile_store(acc, [&](int m, int n, double v) { Lq[(size_t)(ib * 16 + m) * NP + jb * 16 + n] -= v; }, lane);
                }
        }
        phase_sync();
    }
    if (s_fail) { if (tid == 0) *J.flag = 1; return; }

    // ---- Y = L^-1 [A' | c | H]: blocked forward substitution ---------------------------------------------------------------------
    const int ct = NB / 16;
    for (int kb = 0; kb < nb; ++kb) {
        for (int cb = wave; cb < ct; cb += 4) {
            const d4 acc = tile_mma(16, [&](int m, int k) { return Dinv[(size_t)kb * 256 + m * 16 + k]; },
                                    [&](int k, int n) { return Bm[(size_t)(kb * 16 + k) * NB + cb * 16 + n]; }, lane);
            tile_store(acc, [&](int m, int n, double v) { Ym[(size_t)(kb * 16 + m) * NB + cb * 16 + n] = v; }, lane);
        }
        phase_sync();
        int t = 0;
        for (int ib = kb + 1; ib < nb; ++ib)
            for (int cb = 0; cb < ct; ++cb, ++t) {
                if ((t & 3) != wave) continue;
                const d4 acc = tile_mma(16, [&](int m, int k) { return Lq[(size_t)(ib * 16 + m) * NP + kb * 16 + k]; },
                                        [&](int k, int n) { return Ym[(size_t)(kb * 16 + k) * NB + cb * 16 + n]; }, lane);
                tile_store(acc, [&](int m, int n, double v) { Bm[(size_t)(ib * 16 + m) * NB + cb * 16 + n] -= v; }, lane);
            }
        phase_sync();
    }

    // ---- W = Y_A' Y_A = A Q^-1 A' (symmetric by construction) ------------------------------------------------------------------
    {
        const int mt = MP / 16;
        int t = 0;
        for (int i = 0; i < mt; ++i)
            for (int j = 0; j <= i; ++j, ++t) {
                if ((t & 3) != wave) continue;
                const d4 acc = tile_mma(NP, [&](int m, int k) { return Ym[(size_t)k * NB + i * 16 + m]; },
                                        [&](int k, int n) { return Ym[(size_t)k * NB + j * 16 + n]; }, lane);
                tile_store(acc, [&](int m, int n, double v) {
                    const int r = i * 16 + m, c = j * 16 + n;
                    if (r < nc && c < nc) { J.W[(size_t)r * nc + c] = v; if (i != j) J.W[(size_t)c * nc + r] = v; }
                }, lane);
            }
    }
    phase_sync();

    // ---- Z = L^-T Y = Q^-1 [A' | c | H]: blocked back substitution (Z overwrites the right-hand-side block) ------------------
    for (int kb = nb - 1; kb >= 0; --kb) {
        for (int cb = wave; cb < ct; cb += 4) {
            const d4 acc = tile_mma(16, [&](int m, int k) { return Dinv[(size_t)kb * 256 + k * 16 + m]; },
                                    [&](int k, int n) { return Ym[(size_t)(kb * 16 + k) * NB + cb * 16 + n]; }, lane);
            tile_store(acc, [&](int m, int n, double v) { Bm[(size_t)(kb * 16 + m) * NB + cb * 16 + n] = v; }, lane);
        }
        phase_sync();
        int t = 0;
        for (int ib = 0; ib < kb; ++ib)
            for (int cb = 0; cb < ct; ++cb, ++t) {
                if ((t & 3) != wave) continue;
                const d4 acc = tile_mma(16, [&](int m, int k) { return Lq[(size_t)(kb * 16 + k) * NP + ib * 16 + m]; },
                                        [&](int k, int n) { return Bm[(size_t)(kb * 16 + k) * NB + cb * 16 + n]; }, lane);
                tile_store(acc, [&](int m, int n, double v) { Ym[(size_t)(ib * 16 + m) * NB + cb * 16 + n] -= v; }, lane);
            }
        phase_sync();
    }

    // ---- outputs ---------------------------------------------------------------------------------------------------------------
    for (int i = tid; i < nc * nx; i += 256) { const int r = i / nx, l = i % nx; J.Gt[i] = Bm[(size_t)l * NB + r]; }
    for (int i = tid; i < nx * nr; i += 256) { const int l = i / nr, t = i % nr; J.X0H[i] = -Bm[(size_t)l * NB + MP + t]; }
    {
        const int mt = MP / 16, rt = RP / 16;
        int t = 0;
        for (int i = 0; i < mt; ++i)
            for (int j = 0; j < rt; ++j, ++t) {
                if ((t & 3) != wave) continue;
                const d4 acc = tile_mma(NP, [&](int m, int k) { return Ap[(size_t)(i * 16 + m) * NP + k]; },
                                        [&](int k, int n) { return Bm[(size_t)k * NB + MP + j * 16 + n]; }, lane);
                tile_store(acc, [&](int m, int n, double v) {
                    const int r = i * 16 + m, c = j * 16 + n;
                    if (r < nc && c < nr) J.UV[(size_t)r * nr + c] = v + (c == 0 ? J.b[r] : J.F[(size_t)r * nt + c - 1]);
                }, lane);
            }
    }
    if (tid == 0) *J.flag = 0;
}

}  // namespace

hipError_t setup_launch(const SetupJob *jobs_dev, int n_jobs, hipStream_t stream) {
    if (n_jobs <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_setup_mfma, dim3((unsigned)n_jobs), dim3(256), 0, stream, jobs_dev);
    return hipGetLastError();
}

}  // namespace mpc
