// setup_mfma.hip -- the dense Hessian factor and the one-off Schur blocks of a program on the matrix cores (setup_mfma.hpp).
//
// One workgroup of four wavefronts per program; every product is a sequence of 16 x 16 x 4 fp64 MFMA tiles
// (v_mfma_f64_16x16x4_f64), the tiles of a phase are dealt round-robin to the four wavefronts, phases are separated by
// workgroup barriers.  All operands live in a per-program scratch block in HBM (tens of KB: L2-resident); the only LDS use is
// the 16 x 16 diagonal block of a Cholesky panel and its inverse.  The work per program is tiny (config 4: Q 20 x 20,
// A 47 x 20 -> about 150 MFMA instructions); the point of doing it here is that a batch of programs -- the sub-programs of
// the mixed-integer enumeration -- is ONE launch with no host arithmetic, and that the blocks are born in HBM.
//
// Three uses of the same three routines (blocked Cholesky, blocked forward / back substitution against tile right-hand sides):
//   Q = L L'            ->  W = A Q^-1 A', UV, Gt, X0H                                    (the Hessian factor of the KKT systems)
//   W_EE = L2 L2'       ->  Wr, UVr, Me, Ne     (the program's equality rows eliminated from every Schur system, setup_mfma.hpp)
//   A_E A_E' = L3 L3'   ->  AATr, gE            (the same for the Gram matrix of the rank screen)
//
// Separate translation unit: compiled on its own (seconds) and linked into libmpcombi_hip.so.
#include "setup_mfma.hpp"

namespace mpc {
namespace {

typedef double d4 __attribute__((ext_vector_type(4)));

// acc[m][n] = sum_{k < K} a(m, k) b(k, n) for one 16 x 16 tile, K a multiple of 4.
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

struct Shared {
    double (*sD)[17];
    double (*sI)[17];
    int *fail;
};

// In place: the n x n matrix L (row-major, n a multiple of 16; lower triangle and diagonal blocks valid) becomes its lower Cholesky
// factor, blocked right-looking: 16 x 16 diagonal block and its inverse in LDS, panel and trailing updates as MFMA tiles.  Rows
// below n_real are padding (identity) and are not tested; a pivot <= tol * dmax of a real row sets *sh.fail.  Dinv receives the
// inverses of the diagonal blocks (n / 16 blocks of 256 doubles).
__device__ void chol_blocked(double *L, int n, int n_real, double tol_dmax, double *Dinv, const Shared &sh, int tid, int lane, int wave) {
    const int nb = n / 16;
    for (int kb = 0; kb < nb; ++kb) {
        { const int i = tid >> 4, l = tid & 15; sh.sD[i][l] = L[(size_t)(kb * 16 + i) * n + kb * 16 + l]; }
        __syncthreads();
        for (int j = 0; j < 16; ++j) {
            if (tid == 0) {
                double d = sh.sD[j][j];
                if (kb * 16 + j < n_real && !(d > tol_dmax)) { *sh.fail = 1; d = 1.0; }
                sh.sD[j][j] = sqrt(d);
            }
            __syncthreads();
            if (tid > j && tid < 16) sh.sD[tid][j] /= sh.sD[j][j];
            __syncthreads();
            { const int i = tid >> 4, l = tid & 15; if (i > j && l > j && l <= i) sh.sD[i][l] -= sh.sD[i][j] * sh.sD[l][j]; }
            __syncthreads();
        }
        // inverse of the lower-triangular diagonal block, column c by forward substitution (one thread per column)
        if (tid < 16) {
            const int c = tid;
            for (int i = 0; i < 16; ++i) {
                double v = 0.0;
                if (i == c) v = 1.0 / sh.sD[c][c];
                else if (i > c) { double s = 0.0; for (int m = c; m < i; ++m) s += sh.sD[i][m] * sh.sI[m][c]; v = -s / sh.sD[i][i]; }
                sh.sI[i][c] = v;
            }
        }
        __syncthreads();
        { const int i = tid >> 4, l = tid & 15; L[(size_t)(kb * 16 + i) * n + kb * 16 + l] = l <= i ? sh.sD[i][l] : 0.0; Dinv[(size_t)kb * 256 + i * 16 + l] = sh.sI[i][l]; }
        // panel below the diagonal block:  L[ib, kb] = M[ib, kb] inv(L[kb, kb])'
        for (int ib = kb + 1 + wave; ib < nb; ib += 4) {
            const d4 acc = tile_mma(16, [&](int m, int k) { return L[(size_t)(ib * 16 + m) * n + kb * 16 + k]; },
                                    [&](int k, int nn) { return sh.sI[nn][k]; }, lane);
            tile_store(acc, [&](int m, int nn, double v) { L[(size_t)(ib * 16 + m) * n + kb * 16 + nn] = v; }, lane);
        }
        phase_sync();
        // trailing update of the lower triangle:  M[ib, jb] -= L[ib, kb] L[jb, kb]'
        {
            int t = 0;
            for (int ib = kb + 1; ib < nb; ++ib)
                for (int jb = kb + 1; jb <= ib; ++jb, ++t) {
                    if ((t & 3) != wave) continue;
                    const d4 acc = tile_mma(16, [&](int m, int k) { return L[(size_t)(ib * 16 + m) * n + kb * 16 + k]; },
                                            [&](int k, int nn) { return L[(size_t)(jb * 16 + nn) * n + kb * 16 + k]; }, lane);
                    tile_store(acc, [&](int m, int nn, double v) { L[(size_t)(ib * 16 + m) * n + jb * 16 + nn] -= v; }, lane);
                }
        }
        phase_sync();
    }
}

// Y = L^-1 B for the n x (16 ct) right-hand side block B (row stride NB; destroyed), blocked forward substitution.
__device__ void trsm_forward(const double *L, int n, const double *Dinv, double *B, double *Y, int NB, int ct, int lane, int wave) {
    const int nb = n / 16;
    for (int kb = 0; kb < nb; ++kb) {
        for (int cb = wave; cb < ct; cb += 4) {
            const d4 acc = tile_mma(16, [&](int m, int k) { return Dinv[(size_t)kb * 256 + m * 16 + k]; },
                                    [&](int k, int nn) { return B[(size_t)(kb * 16 + k) * NB + cb * 16 + nn]; }, lane);
            tile_store(acc, [&](int m, int nn, double v) { Y[(size_t)(kb * 16 + m) * NB + cb * 16 + nn] = v; }, lane);
        }
        phase_sync();
        int t = 0;
        for (int ib = kb + 1; ib < nb; ++ib)
            for (int cb = 0; cb < ct; ++cb, ++t) {
                if ((t & 3) != wave) continue;
                const d4 acc = tile_mma(16, [&](int m, int k) { return L[(size_t)(ib * 16 + m) * n + kb * 16 + k]; },
                                        [&](int k, int nn) { return Y[(size_t)(kb * 16 + k) * NB + cb * 16 + nn]; }, lane);
                tile_store(acc, [&](int m, int nn, double v) { B[(size_t)(ib * 16 + m) * NB + cb * 16 + nn] -= v; }, lane);
            }
        phase_sync();
    }
}

// Z = L^-T Y (Y destroyed, Z may be the buffer the forward substitution consumed), blocked back substitution.
__device__ void trsm_backward(const double *L, int n, const double *Dinv, double *Y, double *Z, int NB, int ct, int lane, int wave) {
    const int nb = n / 16;
    for (int kb = nb - 1; kb >= 0; --kb) {
        for (int cb = wave; cb < ct; cb += 4) {
            const d4 acc = tile_mma(16, [&](int m, int k) { return Dinv[(size_t)kb * 256 + k * 16 + m]; },
                                    [&](int k, int nn) { return Y[(size_t)(kb * 16 + k) * NB + cb * 16 + nn]; }, lane);
            tile_store(acc, [&](int m, int nn, double v) { Z[(size_t)(kb * 16 + m) * NB + cb * 16 + nn] = v; }, lane);
        }
        phase_sync();
        int t = 0;
        for (int ib = 0; ib < kb; ++ib)
            for (int cb = 0; cb < ct; ++cb, ++t) {
                if ((t & 3) != wave) continue;
                const d4 acc = tile_mma(16, [&](int m, int k) { return L[(size_t)(kb * 16 + k) * n + ib * 16 + m]; },
                                        [&](int k, int nn) { return Z[(size_t)(kb * 16 + k) * NB + cb * 16 + nn]; }, lane);
                tile_store(acc, [&](int m, int nn, double v) { Y[(size_t)(ib * 16 + m) * NB + cb * 16 + nn] -= v; }, lane);
            }
        phase_sync();
    }
}

__global__ void __launch_bounds__(256) k_setup_mfma(const SetupJob *__restrict__ jobs) {
    const SetupJob J = jobs[blockIdx.x];
    __shared__ double sD[16][17], sI[16][17];
    __shared__ double s_dmax;
    __shared__ int s_fail;
    const Shared sh{sD, sI, &s_fail};
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nx = J.nx, nt = J.nt, nc = J.nc, nr = nt + 1, NP = J.NP, MP = J.MP, RP = J.RP, NB = MP + RP;
    double *Lq = J.work, *Dinv = Lq + (size_t)NP * NP, *Ap = Dinv + (size_t)(NP / 16) * 256, *Bm = Ap + (size_t)MP * NP, *Ym = Bm + (size_t)NP * NB;
    double *L2 = Ym + (size_t)NP * NB, *Dinv2 = L2 + (size_t)MP * MP, *B2 = Dinv2 + (size_t)(MP / 16) * 256, *Y2 = B2 + (size_t)MP * NB;

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
    if (!J.Q) { if (tid == 0) { *J.flag = 1; if (J.flag_e) *J.flag_e = 1; } return; }

    // ---- Q = L L' ------------------------------------------------------------------------------------------------------------
    chol_blocked(Lq, NP, nx, 1e-10 * s_dmax, Dinv, sh, tid, lane, wave);
    if (s_fail) { if (tid == 0) { *J.flag = 1; if (J.flag_e) *J.flag_e = 1; } return; }   // not positive definite (within the tolerance of the host test)

    // ---- Y = L^-1 [A' | c | H],  W = Y_A' Y_A = A Q^-1 A' (symmetric by construction) ----------------------------------------------
    const int ct = NB / 16;
    trsm_forward(Lq, NP, Dinv, Bm, Ym, NB, ct, lane, wave);
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

    // ---- Z = L^-T Y = Q^-1 [A' | c | H] (Z overwrites the right-hand-side block); Gt, X0H, UV ------------------------------------
    trsm_backward(Lq, NP, Dinv, Ym, Bm, NB, ct, lane, wave);
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
    const int ne = J.ne;
    if (ne <= 0 || !J.flag_e) return;
    phase_sync();   // W, UV, AAT complete

    // ---- the equality rows E = {0..ne-1} eliminated from the Schur systems:  W_EE = L2 L2' ------------------------------------------
    const int EP = (ne + 15) & ~15;
    const int mt = MP / 16, rt = RP / 16, et = EP / 16;
    for (int i = tid; i < EP * EP; i += 256) { const int r = i / EP, c = i % EP; L2[i] = (r < ne && c < ne) ? J.W[(size_t)r * nc + c] : (r == c ? 1.0 : 0.0); }
    for (int i = tid; i < EP * NB; i += 256) {
        const int r = i / NB, c = i % NB;
        double v = 0.0;
        if (r < ne) { if (c < MP) { if (c < nc) v = J.W[(size_t)r * nc + c]; } else if (c - MP < nr) v = J.UV[(size_t)r * nr + c - MP]; }
        B2[i] = v;
    }
    if (tid == 0) { double dmax = 0.0; for (int i = 0; i < ne; ++i) dmax = fmax(dmax, J.W[(size_t)i * nc + i]); s_dmax = dmax; }
    phase_sync();
    chol_blocked(L2, EP, ne, 1e-10 * s_dmax, Dinv2, sh, tid, lane, wave);
    if (s_fail) { if (tid == 0) *J.flag_e = 1; return; }
    trsm_forward(L2, EP, Dinv2, B2, Y2, NB, ct, lane, wave);      // Y2 = L2^-1 [W[E,:] | UV[E,:]]
    {
        int t = 0;                                                // Wr = W - Y2_W' Y2_W,  UVr = UV - Y2_W' Y2_UV
        for (int i = 0; i < mt; ++i)
            for (int j = 0; j < mt + rt; ++j, ++t) {
                if ((t & 3) != wave) continue;
                const d4 acc = tile_mma(EP, [&](int m, int k) { return Y2[(size_t)k * NB + i * 16 + m]; },
                                        [&](int k, int n) { return Y2[(size_t)k * NB + j * 16 + n]; }, lane);
                tile_store(acc, [&](int m, int n, double v) {
                    const int r = i * 16 + m, c = j * 16 + n;
                    if (r >= nc) return;
                    if (c < MP) { if (c < nc) J.Wr[(size_t)r * nc + c] = J.W[(size_t)r * nc + c] - v; }
                    else if (c - MP < nr) J.UVr[(size_t)r * nr + c - MP] = J.UV[(size_t)r * nr + c - MP] - v;
                }, lane);
            }
    }
    phase_sync();
    trsm_backward(L2, EP, Dinv2, Y2, B2, NB, ct, lane, wave);     // Z2 = W_EE^-1 [W[E,:] | UV[E,:]] = [Ne | Me]
    for (int i = tid; i < ne * nc; i += 256) { const int r = i / nc, c = i % nc; J.Ne[i] = B2[(size_t)r * NB + c]; }
    for (int i = tid; i < ne * nr; i += 256) { const int r = i / nr, t = i % nr; J.Me[i] = B2[(size_t)r * NB + MP + t]; }
    phase_sync();

    // ---- the same for the Gram matrix of the rank screen:  A_E A_E' = L3 L3',  AATr = AAT - AAT[:,E] (A_E A_E')^-1 AAT[E,:] -----------
    for (int i = tid; i < EP * EP; i += 256) { const int r = i / EP, c = i % EP; L2[i] = (r < ne && c < ne) ? J.AAT[(size_t)r * nc + c] : (r == c ? 1.0 : 0.0); }
    for (int i = tid; i < EP * NB; i += 256) { const int r = i / NB, c = i % NB; B2[i] = (r < ne && c < nc) ? J.AAT[(size_t)r * nc + c] : 0.0; }
    if (tid == 0) { double dmax = 0.0; for (int i = 0; i < ne; ++i) dmax = fmax(dmax, J.AAT[(size_t)i * nc + i]); s_dmax = dmax; }
    phase_sync();
    chol_blocked(L2, EP, ne, 1e-10 * s_dmax, Dinv2, sh, tid, lane, wave);
    if (s_fail) { if (tid == 0) *J.flag_e = 1; return; }
    for (int i = tid; i < ne; i += 256) { const double l = L2[(size_t)i * EP + i]; J.gE[i] = l * l; J.gE[ne + i] = J.AAT[(size_t)i * nc + i]; }
    trsm_forward(L2, EP, Dinv2, B2, Y2, NB, mt, lane, wave);
    {
        int t = 0;
        for (int i = 0; i < mt; ++i)
            for (int j = 0; j <= i; ++j, ++t) {
                if ((t & 3) != wave) continue;
                const d4 acc = tile_mma(EP, [&](int m, int k) { return Y2[(size_t)k * NB + i * 16 + m]; },
                                        [&](int k, int n) { return Y2[(size_t)k * NB + j * 16 + n]; }, lane);
                tile_store(acc, [&](int m, int n, double v) {
                    const int r = i * 16 + m, c = j * 16 + n;
                    if (r < nc && c < nc) { const double x = J.AAT[(size_t)r * nc + c] - v; J.AATr[(size_t)r * nc + c] = x; if (i != j) J.AATr[(size_t)c * nc + r] = x; }
                }, lane);
            }
    }
    (void)et;
    if (tid == 0) *J.flag_e = 0;
}

}  // namespace

hipError_t setup_launch(const SetupJob *jobs_dev, int n_jobs, hipStream_t stream) {
    if (n_jobs <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_setup_mfma, dim3((unsigned)n_jobs), dim3(256), 0, stream, jobs_dev);
    return hipGetLastError();
}

}  // namespace mpc
