// setup_mfma.hpp -- one-off dense blocks of a program, formed on the device with v_mfma_f64_16x16x4_f64 (setup_mfma.hip).
//
// Reference counterpart: the dense KKT solve of MPQP_Program.optimal_control_law (mpqp_program.py:182-198) factors
// [[A_as, 0], [Q, A_as']] once PER ACTIVE SET.  With Q > 0 everything that depends on Q alone is formed once per PROGRAM:
//     Q = L L'                              (blocked Cholesky, 16 x 16 tiles: panel and trailing updates on the matrix cores)
//     Y = L^-1 [A' | c | H]                 (blocked forward substitution)
//     W = Y_A' Y_A = A Q^-1 A'              (n_c x n_c; the per-candidate Schur complement is the gathered block W[as, as])
//     Z = L^-T Y = Q^-1 [A' | c | H]        (blocked back substitution)
//     Gt = Z_A' = A Q^-1,  X0H = -Z_[c|H],  UV = A Z_[c|H] + [b | F],   A A' (Gram matrix of the rank screen)
// north_star: "MFMA used only for the dense Hessian factor in the QP KKT".
#pragma once
#include <hip/hip_runtime.h>

namespace mpc {

// One program's set-up job; an array of them (device memory) is one launch: blockIdx.x = job.
struct SetupJob {
    int nx, nt, nc;          // n_x, n_theta, n_c
    int NP, MP, RP;          // the same rounded up to multiples of 16 (RP: n_theta + 1)
    const double *A, *b, *F, *c, *H, *Q;   // device, row-major, unpadded; Q == nullptr: only A A' is formed
    double *W, *UV, *Gt, *X0H, *AAT;       // device outputs, row-major, unpadded (n_c x n_c, n_c x (n_t+1), n_c x n_x, n_x x (n_t+1), n_c x n_c)
    double *work;            // setup_work_doubles(...) doubles of scratch, any content
    int *flag;               // out: 0 = Q is positive definite, blocks valid; 1 = a Cholesky pivot <= 1e-10 max|Q_ii| (blocks invalid)
    // ---- elimination of the program's equality rows E = {0..ne-1} (they are in every active set): with
    //        Wr  = W  - W[:,E] W_EE^-1 W[E,:]      UVr = UV - W[:,E] W_EE^-1 UV[E,:]      AATr = AAT - AAT[:,E] AAT_EE^-1 AAT[E,:]
    //      the Schur system of an active set E + a is the one of `a` alone on the reduced blocks, lambda_a = -Wr[a,a]^-1 UVr[a], and
    //        lambda_E = -(Me + Ne[:,a] lambda_a),   Me = W_EE^-1 UV[E,:],   Ne = W_EE^-1 W[E,:]
    //      so the one-thread-per-candidate KKT kernel (k_kkt_thread, at most 8 rows) covers non-condensed MPC programs whose active
    //      sets carry ten equality rows (config 2).  ne == 0: nothing is formed.  All outputs unpadded, row-major.
    int ne;
    double *Wr, *UVr, *AATr;   // n_c x n_c, n_c x (n_t+1), n_c x n_c
    double *Me, *Ne;           // ne x (n_t+1), ne x n_c
    double *gE;                // ne pivots of the Gram elimination of the equality rows, then ne diagonal entries of A_E A_E'
    int *flag_e;               // out: 0 = reduced blocks valid; 1 = W_EE or A_E A_E' not positive definite (no elimination)
};

inline int setup_pad16(int v) { return (v + 15) & ~15; }
inline size_t setup_work_doubles(int nx, int nt, int nc) {
    const size_t NP = setup_pad16(nx), MP = setup_pad16(nc), RP = setup_pad16(nt + 1), NB = MP + RP;
    return NP * NP /* L */ + (NP / 16) * 256 /* inverses of the diagonal blocks */ + MP * NP /* padded A */ + 2 * NP * NB /* right-hand sides, Y / Z */
           + MP * MP + (MP / 16) * 256 + 2 * MP * NB /* the same three for the equality-row elimination (at most n_c rows) */;
}

// Launches the set-up kernel for n_jobs programs on `stream` (jobs_dev: device array of SetupJob).
hipError_t setup_launch(const SetupJob *jobs_dev, int n_jobs, hipStream_t stream);

}  // namespace mpc
