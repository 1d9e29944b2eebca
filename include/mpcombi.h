/*
 * mpcombi.h -- C ABI of the MI355X-native combinatorial mpLP/mpQP engine (libmpcombi_hip.so).
 *
 * The reference (PPOPT, pure Python) has no FFI; its plug points for this path are Python callables.
 * Each entry point below names the reference interface it replaces (paths relative to
 * /root/reference/src/ppopt):
 *
 *   mpc_create / mpc_destroy     the read-only `program` object every worker receives
 *                                (mpqp_program.py:29-42, mplp_program.py:60-134; matrices AFTER presolve)
 *   mpc_check_level              the batched operator  pool.map(lambda a: full_process(program, a, murder_list,
 *                                gen_children), to_check)   mp_solvers/mpqp_parrallel_combinatorial.py:110-116, :17-64
 *   mpc_frontier_* / mpc_level_* the same operator with the frontier (to_check), the pruned list (murder_list,
 *                                mp_solvers/solver_utils.py:15-55) and the children (future_list, driver :127-135) kept
 *                                resident in HBM between levels
 *   mpc_lp_solve_batch           the deterministic-solver plug  Solver.solve_lp(c, A, b, equality_constraints)
 *                                solver.py:211-246 -> solver_interface/cvxopt_interface.py:153-208, batched
 *
 * Conventions: plain pointers and sizes, row-major float64, int32 indices, caller-owned buffers that are
 * copied at creation; integer return codes (0 = MPC_OK), never C++ exceptions; per-candidate numerical
 * trouble is a status value, not an error.  A handle is bound to one device and one stream and is not
 * re-entrant; use one handle per GPU.  Calls that read or write caller-owned memory (host or device pointers) have
 * completed on the handle's stream when they return; device pointers handed in must be ready (the caller synchronises
 * its own stream first).
 */
#ifndef MPCOMBI_H
#define MPCOMBI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MPC_OK 0
#define MPC_ERR_INVALID 1      /* bad argument / unsupported dimensions                      */
#define MPC_ERR_HIP 2          /* a HIP runtime call failed (mpc_last_error has the text)     */
#define MPC_ERR_CAPACITY 3     /* caller buffer too small; required size returned in-place    */
#define MPC_ERR_STATE 4        /* call sequence error (e.g. level results requested before run) */

/* per-candidate status == verdict of full_process (driver lines 17-64) */
#define MPC_INFEASIBLE 0        /* rank deficient (numpy's SVD rule, constraint_utilities.py:222-236) or (x,theta) LP infeasible -> pruned */
#define MPC_FEASIBLE 1          /* feasible, not optimal -> children.  "Not optimal" is the reference's: its optimality LP maximises t
                                   (mpqp_program.py:203-322) and anything but an optimal status is None -- also an UNBOUNDED t on a
                                   parameter set that is open in some direction (k_recession poses that question for such programs) */
#define MPC_OPTIMAL_NO_REGION 2 /* optimal, region lower dimensional (gen_cr returned None) -> pruned      */
#define MPC_REGION 3            /* critical region -> children                                             */
#define MPC_SINGULAR_KKT 4      /* feasible, KKT matrix numerically singular: no region, children expanded */
#define MPC_LP_LIMIT 5          /* an LP hit its iteration limit: treated as MPC_FEASIBLE by the driver    */

/* LP status of mpc_lp_solve_batch (anything but OPTIMAL is `None` in the reference) */
#define MPC_LP_OPTIMAL 0
#define MPC_LP_INFEASIBLE 1
#define MPC_LP_UNBOUNDED 2
#define MPC_LP_ITERLIMIT 3

#define MPC_MASK_WORDS 2        /* active sets as bit masks of 64-bit words: 2 words for n_c <= 128, 4 for n_c <= 256 (mpc_mask_words) */
#define MPC_MAX_NC 256
#define MPC_MAX_ROWS 192        /* n_c + n_tc + 1 */

typedef struct mpc_handle mpc_handle;

/* A presolved multiparametric program:  min 1/2 x'Qx + theta'H'x + c'x  s.t.  A x <= b + F theta
 * (first n_eq rows are equalities),  A_t theta <= b_t.   Q == NULL => mpLP. */
typedef struct {
    int32_t n_x, n_t, n_c, n_eq, n_tc;
    const double *A;   /* n_c  x n_x */
    const double *b;   /* n_c        */
    const double *F;   /* n_c  x n_t */
    const double *c;   /* n_x        */
    const double *H;   /* n_x  x n_t */
    const double *Q;   /* n_x  x n_x, or NULL */
    const double *A_t; /* n_tc x n_t */
    const double *b_t; /* n_tc       */
} mpc_problem;

typedef struct {
    int64_t n;             /* candidates in the level                                 */
    int32_t k;             /* their cardinality (incl. equalities)                    */
    int32_t kkt_mode;      /* 0 Schur/Cholesky (Q > 0), 1 dense KKT LU (mpLP, PSD Q)    */
    int64_t n_status[6];   /* histogram of the status byte                            */
    int64_t n_regions;     /* == n_status[MPC_REGION]                                 */
    int64_t n_children;    /* size of the next frontier (0 when gen_children == 0)    */
    int64_t n_pruned_new;  /* masks appended to the pruned list by this level         */
    int64_t lp_pivots;     /* simplex pivots executed by this level (all LPs)         */
    float ms_verdict, ms_region, ms_children, ms_total; /* HIP-event times on the handle's stream */
    int64_t n_xtheta_lp;   /* candidates whose feasibility needed the large (x,theta) LP */
    int64_t n_xtheta_fallback; /* ... of which the vertex warm start was abandoned for a from-scratch solve */
    int64_t wave_cycles[4];    /* wavefront cycles in: KKT solve, theta LP, (x,theta) LP, region build */
    int64_t n_region_retry;    /* optimal candidates re-solved by the LDS-engine region kernel */
    int64_t n_x_cached;        /* (x,theta) solves that started from the parent's dictionary cached in HBM */
    /* HIP-event time of single launches of this level (0 when the kernel did not run): k_theta2, the main k_x2 launch
     * (over n_x_items candidates), k_region2 (over n_opt candidates); region_side_stream = 1: that k_region2 launch ran on the
     * handle's side stream UNDER the (x,theta) stage (its time is not on the level's critical path), 0: in line */
    float ms_theta, ms_x, ms_region2, region_side_stream;
    int64_t n_x_items, n_opt;
    int64_t dict_read_bytes;   /* bytes of one cached dictionary record as k_x2 reads it (0: no cache on this level)   */
    int64_t dict_write_bytes;  /* bytes of one record as k_x2 stores it for the next level (0: nothing stored)         */
    int64_t n_theta_items;     /* candidates k_theta2 processed (those the thread kernel's screen left open)             */
    int64_t n_region_rows;     /* rows [f | E] the region kernel appended to the row pool (streamed levels: rows of erows in use) */
    /* HIP-event times of k_kkt_thread (over n candidates) and of the last level's quick test k_xq / k_xq_grouped (over n_xq_items
     * candidates, xq_pivots product-form iterations: each reads one column and one row of the parent's record) */
    float ms_kkt, ms_xq;
    int64_t n_xq_items, xq_pivots;
    int64_t xq_record_ints, xq_record_rows, xq_record_cols;   /* 2 mr + NXC + 3 ints, mr rows, n_d0c + 1 columns of a record */
    /* round 5: the quick test's first pass with one thread per candidate (k_xq_thread): candidates it decided (of n_xq_items; the
     * wavefront kernel gets the others) and its HIP-event time (inside ms_xq) */
    int64_t n_xq_thread;
    float ms_xq_thread, xq_thread_beside_theta;   /* 1: the pass ran on the second stream beside the theta stage (outside ms_xq), 0: inside ms_xq */
    /* round 5: one-step plans on a level that keeps dictionaries: records written by k_x1 (the streamed single pivot), its HIP-event
     * time and that of the plan pass (k_xq_thread in plan mode) -- both inside ms_x, whose remainder is the register simplex k_x2 */
    int64_t n_x1;
    float ms_x1, ms_x_plan;
} mpc_level_stats;

/* ---- library / device ------------------------------------------------------------------------------ */
int mpc_device_count(void);
const char *mpc_version(void);
/* text of the last error of a call that had no handle (mpc_create, mpc_lp_solve_batch) */
const char *mpc_last_global_error(void);

/* ---- program handle ---------------------------------------------------------------------------------- */
/* stream: a hipStream_t (as void*) the handle should launch on, or NULL to create its own. */
int mpc_create(const mpc_problem *problem, int32_t device, void *stream, mpc_handle **out);
int mpc_destroy(mpc_handle *h);
const char *mpc_last_error(const mpc_handle *h);
/* fixed strides of one region record for this program (see mpc_level_regions) */
int64_t mpc_region_doubles(const mpc_handle *h);
int64_t mpc_region_ints(const mpc_handle *h);
/* dynamic LDS bytes per wavefront of the verdict / region kernels (for reports) */
int32_t mpc_lds_bytes(const mpc_handle *h, int32_t which);
/* The one-off dense blocks of the program as the device holds them (diagnostics, tests).  For a positive definite Q they are
 * formed at mpc_create by the MFMA set-up kernel (csrc/setup_mfma.hip: blocked Cholesky of Q and the products below as
 * v_mfma_f64_16x16x4_f64 tiles) and replace the per-active-set dense KKT factorisation of MPQP_Program.optimal_control_law
 * (mpqp_program.py:182-198):  which = 0: W = A Q^-1 A' (n_c x n_c), 1: UV = [A Q^-1 c + b | A Q^-1 H + F] (n_c x (n_t+1)),
 * 2: Gt = A Q^-1 (n_c x n_x), 3: X0H = -Q^-1 [c | H] (n_x x (n_t+1)), 4: A A' (n_c x n_c; any program).  *n = number of doubles
 * (0 for which < 4 when Q is absent or not positive definite); MPC_ERR_CAPACITY when cap < *n.
 * which = 5..10: the blocks with the program's n_eq equality rows E (members of every active set) eliminated, so that the Schur
 * system of an active set E + a is the one of `a` alone:  5: Wr = W - W[:,E] W_EE^-1 W[E,:], 6: UVr = UV - W[:,E] W_EE^-1 UV[E,:],
 * 7: (A A')r likewise, 8: Me = W_EE^-1 UV[E,:] (n_eq x (n_t+1)), 9: Ne = W_EE^-1 W[E,:] (n_eq x n_c), 10: the n_eq pivots of the
 * Gram elimination of E, then the n_eq diagonal entries of A_E A_E'.  *n = 0 when n_eq = 0 or the elimination is not in use. */
int mpc_program_block(mpc_handle *h, int32_t which, double *out_host, int64_t cap, int64_t *n);
/* the HIP stream the handle launches on (hipStream_t as void*) */
void *mpc_stream(const mpc_handle *h);

/* ---- frontier (to_check) and pruned list (murder_list), resident on the device ------------------------ */
/* generate_children_sets(equality_indices, n_c)  (driver line 98) */
int mpc_frontier_root(mpc_handle *h);
int mpc_frontier_set(mpc_handle *h, const int32_t *cand_host, int64_t n, int32_t k);
int mpc_frontier_set_device(mpc_handle *h, const int32_t *cand_dev, int64_t n, int32_t k);
/* keep candidates rank, rank+world, ... of the resident frontier (multi-GPU: the ranks hold identical frontiers up to the
 * level at which they split; the kept candidates still find their parents' dictionaries in this GPU's cache) */
int mpc_frontier_shard(mpc_handle *h, int32_t rank, int32_t world);
int mpc_frontier_info(const mpc_handle *h, int64_t *n, int32_t *k);
int mpc_frontier_get(mpc_handle *h, int32_t *cand_host, int64_t cap);
int mpc_pruned_clear(mpc_handle *h);
int32_t mpc_mask_words(const mpc_handle *h);                                    /* words of one active-set mask for this program */
int mpc_pruned_add(mpc_handle *h, const uint64_t *masks_host, int64_t m);        /* m x mpc_mask_words(h) */
int mpc_pruned_add_device(mpc_handle *h, const uint64_t *masks_dev, int64_t m);
int64_t mpc_pruned_count(const mpc_handle *h);
int mpc_pruned_get(mpc_handle *h, uint64_t *masks_host, int64_t cap);

/* ---- one BFS level over the resident frontier --------------------------------------------------------- */
/* Runs verdict -> region -> (gen_children ? child generation against the CURRENT pruned list : nothing).
 * Sets pruned by this level become visible to child generation only after mpc_frontier_advance, which is the
 * reference's worker semantics (workers hold the murder_list of the previous levels, driver :110-131). */
int mpc_level_run(mpc_handle *h, int32_t gen_children, mpc_level_stats *stats);
int mpc_level_run_ex(mpc_handle *h, int32_t gen_children, int32_t flags, mpc_level_stats *stats);   /* flags: MPC_LEVEL_GRAPH, MPC_LEVEL_KEEP_LOWDIM */
int mpc_level_status(mpc_handle *h, uint8_t *status_host);                       /* n bytes, frontier order */
/* One level of SEVERAL programs per launch (reference: mp_solvers/mpmiqp_enumeration.py:41-50 maps solve_mpqp over the
 * binary fixations one sub-program at a time).  handles[0..n_handles) are distinct programs on one device, each with its own
 * resident frontier; gen_children[i] as in mpc_level_run; flags: MPC_LEVEL_KEEP_LOWDIM.  Every stage of the level is ONE launch for
 * all members (blockIdx.y = member, arguments from a table in device memory), the host synchronises once, and each member is left
 * in the state mpc_level_run would have left it -- same kernels' bodies, same lists: identical statuses, children and pruned masks,
 * and bit-identical region records except for the rare candidates that turn out optimal only after the theta stage (built here by
 * the register-resident region kernel with all others, by the LDS-engine kernel in the single-program form: same regions, coefficients
 * equal to ~1e-10, see csrc/batch_level.hpp); afterwards every handle is used on its own as usual (mpc_level_regions_slots, mpc_frontier_advance,
 * ...).  Members the batch form does not cover (no register-resident LP instantiation, MPC_NO_SMALLPATH=1, a level that needs the
 * LDS-engine region kernel or has a late optimal candidate) are run by mpc_level_run's own paths inside this call; *n_batched
 * (optional) = members that went through the shared launches.  stats (optional): n_handles entries. */
int mpc_level_run_batch(mpc_handle **handles, int32_t n_handles, const int32_t *gen_children, int32_t flags, mpc_level_stats *stats,
                        int32_t *n_batched);
/* For drivers that hold many handles: the device memory (GB) the next level of h's frontier will hold in a batch (what is charged
 * against MPC_BATCH_BUDGET_GB), and mpc_trim: an idle handle gives its level buffers (frontier, lists, region records, dictionary
 * cache, pruned list) back to the library's pool -- the program stays, a later solve allocates again. */
int mpc_frontier_advance_batch(mpc_handle **handles, int32_t n_handles);   /* mpc_frontier_advance for every handle of a list */
double mpc_level_memory_gb(const mpc_handle *h, int32_t gen_children);
int mpc_trim(mpc_handle *h);
/* The two halves of mpc_level_run_batch: _start queues the shared launches and returns (*token owns the state; the handles must not
 * be used until _wait), _wait synchronises, completes every member and runs the members outside the shared launches.  A token is
 * consumed by exactly one _wait. */
int mpc_level_batch_start(mpc_handle **handles, int32_t n_handles, const int32_t *gen_children, int32_t flags, void **token);
int mpc_level_batch_wait(void *token, mpc_level_stats *stats, int32_t *n_batched);
/* The same level driven by the handle's worker thread: mpc_level_start returns at once, mpc_level_wait joins it and returns
 * what mpc_level_run would have returned.  Between the two only mpc_level_stream_info / mpc_level_chunk_wait may be called
 * on the handle.  (Reference: the parent process is free while pool.map runs, driver :116, and merges results as they come.)
 * flags & MPC_LEVEL_STREAM: the region records of the level are STREAMED to the host while the region kernel runs -- the
 * kernel writes head_d / head_i / erows (layout of mpc_level_regions_slots, one slot per optimal candidate) straight into
 * page-locked host memory and raises one flag per chunk of slots, so the caller builds its region objects chunk by chunk
 * under the kernel instead of fetching everything afterwards.
 *   mpc_level_stream_info  blocks until the region stage of the running level has been launched; hands over the three
 *                          arrays (the CALLER owns them from then on: mpc_host_free), the number of slots, the row
 *                          capacity of erows, the chunk size (slots) and the number of chunks.  n_slots == 0: this level
 *                          does not stream (no optimal candidate, a shape outside the register-engine region kernel, or
 *                          more than 1 GiB of records) -- fetch with mpc_level_regions_slots after mpc_level_wait.
 *   mpc_level_chunk_wait   blocks until every slot of chunk j (slots j*chunk .. ) is complete in host memory.  head_d and
 *                          erows of a chunk are complete with its flag as well.
 *   mpc_level_stream_fixup after mpc_level_wait, when stats.n_region_retry > 0: fills the slots of the candidates that were
 *                          re-solved by the LDS-engine kernel (their head_i[0] was MPC_STREAM_RETRY while streaming); rows
 *                          are appended behind stats.n_region_rows, *n_rows = rows in use afterwards. */
#define MPC_LEVEL_STREAM 1
/* MPC_LEVEL_THEN_BASE (mpc_level_start, with gen_children == 0): behind this -- the last -- level the worker also checks the
 * base active set, the equality rows alone (driver :142-146), while the caller still consumes the level's streamed records;
 * mpc_base_result returns its status and fixed-stride region record (layout of mpc_level_regions) once, MPC_ERR_STATE when
 * the check did not run (the caller then uses mpc_check_level).  Frontier, level results and the pruned list of the handle
 * then belong to the base set. */
#define MPC_LEVEL_THEN_BASE 8
/* MPC_LEVEL_GRAPH (gen_children must be 0): the question of the connected-graph traversals (mp_solvers/mpqp_combi_graph.py:
 * 48-66, feasability_check) instead of full_process -- rank test, KKT solve, "is the critical region non-empty" (the theta
 * LP over multiplier, slack and A_t rows), region.  The (x,theta) feasibility LP is not posed.  Statuses: MPC_INFEASIBLE =
 * rank deficient, MPC_FEASIBLE = full rank, empty region, MPC_OPTIMAL_NO_REGION = non-empty but lower dimensional,
 * MPC_REGION.  Nothing is appended to the pruned list. */
#define MPC_LEVEL_GRAPH 4
/* MPC_LEVEL_KEEP_LOWDIM: the rule of the SERIAL driver (mp_solvers/mpqp_combinatorial.py:44-61) and of the _exp parallel driver
 * (mpqp_parallel_combinatorial_exp.py:38-52): a set that is optimal with a lower-dimensional region is expanded like any
 * feasible set and is not put on the pruned list.  Without the flag the parallel driver's rule applies (:57-59): such a set is
 * pruned together with its supersets, which can lose regions on degenerate programs. */
#define MPC_LEVEL_KEEP_LOWDIM 16
/* MPC_LEVEL_ONLY_BASE (mpc_level_start): the worker checks the base active set ONLY -- no level is run, the frontier is replaced by
   the base set; result through mpc_base_result after mpc_level_wait.  For a second handle of the same program (`twin`): the
   driver starts the check there when the solve reaches its first large level, and the check's chain of small kernels (0.45 ms
   at config 4) runs under the large levels of the main handle instead of behind the last one. */
#define MPC_LEVEL_ONLY_BASE 32
#define MPC_STREAM_RETRY 7
/* Whether the region stage of a level may be launched under its (x,theta) stage (default 1; environment MPC_NO_ROVERLAP=1 = 0).
 * An overlapped launch reserves spare record slots for candidates that turn out optimal only after it (re-solved doubtful ones);
 * should a level ever find more of them than it reserved, mpc_level_run / mpc_level_wait return MPC_ERR_CAPACITY -- no
 * candidate is demoted -- and the caller repeats the solve with on = 0, where every optimal candidate is known at launch.
 * (Reference: full_process never loses a region, mp_solvers/mpqp_parrallel_combinatorial.py:52-61.) */
int mpc_set_region_overlap(mpc_handle *h, int32_t on);
/* mpc_set_timing(h, 0): the levels of this handle run WITHOUT the HIP-event records around their stages and heavy kernels -- about
 * fourteen records per large level, four per small one, each a marker the queue stops at (measured, one MI355X: 0.16 ms of a 4.9 ms
 * config-4 solve, 0.11 of a 1.1 ms config-2 solve).  The ms_* fields of mpc_level_stats are then 0, every count is as before.  The host
 * layer switches the records on for the solves that ask for a profile.  Default: on (MPC_NO_KEV=1 in the environment: off). */
int mpc_set_timing(mpc_handle *h, int32_t on);
/* mpc_engine_kind: which kernels this handle's levels run on (diagnostics: the shape sweep records it per cell).  out[0] = 1 the
 * register-resident kernels (k_kkt_thread / k_theta2 / k_x2 / k_region2 ...), 0 the LDS-engine kernels (k_verdict / k_region);
 * out[1] = rows per lane of the theta LP (1 / 2; 0: none), out[2] = of the (x,theta) dictionary, out[3] = of the region LP (0: the
 * LDS-engine region kernel); out[4] = compile-time n_theta of the instantiation (4 / 8 / 10); out[5] = 1: the parameter set is open in
 * some direction (k_recession behind the verdict stages); out[6] = KKT mode (0 Schur / Cholesky, 1 dense); out[7] = largest number of
 * inequality rows k_kkt_thread solves for.  (No counterpart in the reference: it has one code path for every program,
 * mplp_program.py:411-444.) */
int mpc_engine_kind(mpc_handle *h, int32_t out[8]);
int mpc_level_start(mpc_handle *h, int32_t gen_children, int32_t flags);
int mpc_level_stream_info(mpc_handle *h, double **head_d, int32_t **head_i, double **erows, int64_t *n_slots, int64_t *cap_rows,
                          int32_t *chunk, int32_t *n_chunks);
int mpc_level_chunk_wait(mpc_handle *h, int32_t j);
int mpc_level_wait(mpc_handle *h, mpc_level_stats *stats);
int mpc_level_stream_fixup(mpc_handle *h, double *head_d, int32_t *head_i, double *erows, int64_t *n_rows);
int mpc_base_result(mpc_handle *h, uint8_t *status, int64_t *n_regions, double *rec_d, int32_t *rec_i);
/* The WHOLE level loop of the reference's driver behind one call (mp_solvers/mpqp_parrallel_combinatorial.py:98-139: root frontier =
 * children of the equality set, then per level pool.map(full_process) -> merge pruned / children / regions -> next level), run by
 * the handle's worker thread: no level is handed back to the caller, so the device never waits for the host between levels
 * (round 3: 0.38 ms of a 6.7 ms solve of config 4 sat in those hand-overs, tools/gpu_idle.sh).  The caller only CONSUMES: per level
 * it receives the region records while later levels already run.
 *   mpc_solve_start(h, max_levels, flags)  clears the pruned list, roots the frontier and starts the loop for at most max_levels
 *          levels (the last one without children, driver :108).  flags: MPC_LEVEL_STREAM (large levels stream their records),
 *          MPC_LEVEL_KEEP_LOWDIM, MPC_SOLVE_FETCH (records of levels that do not stream are copied to page-locked host arrays by the
 *          loop), MPC_LEVEL_THEN_BASE (the base active set is checked behind the last level: mpc_base_result).
 *   mpc_solve_level(h, level, &info)       blocks until level `level` (0-based) has records to hand over or has finished.
 *          info.mode: 1 = streamed (arrays are being written; chunk j readable after mpc_solve_chunk_wait(h, level, j); layout of
 *          mpc_level_stream_info), 2 = complete arrays (n_slots slots, n_rows rows in use), 0 = the level has no records,
 *          -1 = the loop ended before this level (return code = the loop's).  Arrays are the caller's from then on (mpc_host_free).
 *   mpc_solve_level_wait(h, level, &stats, &ms_wall)  blocks until the level has finished: its statistics; when
 *          stats.n_region_retry > 0 the slots of the re-solved candidates of a streamed level have been filled in place (list the
 *          level's slots again: head_i[slot][0] == MPC_REGION).
 *   mpc_solve_wait(h, &n_levels)           joins the loop: its return code (MPC_ERR_CAPACITY as for mpc_level_wait) and the number
 *          of levels completed.  The handle is then in the state after the last level (mpc_frontier_info, mpc_level_* fetches). */
#define MPC_SOLVE_FETCH 64
typedef struct {
    int32_t level, k;          /* 0-based level, cardinality of its candidates                                   */
    int32_t mode;              /* see above                                                                      */
    int32_t chunk, n_chunks;   /* mode 1: slots per chunk, number of chunks                                      */
    int32_t pad_;
    int64_t n;                 /* candidates of the level                                                        */
    int64_t n_slots;           /* rows of head_d / head_i                                                        */
    int64_t n_rows;            /* mode 1: row capacity of erows; mode 2: rows in use                             */
    double *head_d;            /* [n_slots][fd]     fd = n_x n_t + n_x + k n_t + k                               */
    int32_t *head_i;           /* [n_slots][fi]     fi = 8 + k + n_tc + k + 2 (n_c - k)                          */
    double *erows;             /* [n_rows][n_t + 1] */
} mpc_solve_level_info;
int mpc_solve_start(mpc_handle *h, int32_t max_levels, int32_t flags);
int mpc_solve_level(mpc_handle *h, int32_t level, mpc_solve_level_info *info);
int mpc_solve_chunk_wait(mpc_handle *h, int32_t level, int32_t j);
int mpc_solve_level_wait(mpc_handle *h, int32_t level, mpc_level_stats *stats, double *ms_wall);
int mpc_solve_wait(mpc_handle *h, int32_t *n_levels);
/* Region records of this level in frontier order.  cand_index[i] = position of region i's candidate.
 *   rec_d (mpc_region_doubles each): A_x[n_x*n_t] b_x[n_x] A_l[n_c*n_t] b_l[n_c] E[(n_c+n_tc)*n_t] f[n_c+n_tc]
 *   rec_i (mpc_region_ints each):    k n_E n_omega n_lambda n_regular | active[n_c] | omega[n_tc] | lambda[n_c]
 *                                     | regular_idx[n_c] | regular_con[n_c]                      (unused = -1)
 * i.e. every field of CriticalRegion (critical_region.py:34-48); E/f are the non-redundant unit-norm rows
 * before exact-duplicate removal. */
int mpc_level_regions(mpc_handle *h, double *rec_d_host, int32_t *rec_i_host, int64_t *cand_index_host, int64_t cap);
/* The same regions in compact form (what the device writes; no padding to n_c):
 *   head_d (fd doubles each): A_x[n_x*n_t] b_x[n_x] A_l[k*n_t] b_l[k]
 *   head_i (fi ints each):    3 cand nE n_omega n_lambda n_regular e_off 0 | active[k] | omega[n_tc] | lambda[k]
 *                             | regular_idx[n_c-k] | regular_con[n_c-k]            (unused = -1)
 *   erows  (n_t+1 doubles each): f, E[0..n_t)  -- region i owns rows e_off .. e_off+nE-1
 * mpc_compact_strides gives fd, fi for the current frontier (they depend on its cardinality k) and an upper bound
 * of the number of rows this level's regions own. */
int mpc_compact_strides(const mpc_handle *h, int64_t *fd, int64_t *fi, int64_t *max_rows);
int mpc_level_regions_compact(mpc_handle *h, double *head_d, int32_t *head_i, int64_t cap_regions, double *erows,
                              int64_t cap_rows, int64_t *n_regions, int64_t *n_rows);
/* The same data without any host-side repacking: ALL slots the region kernel wrote (one per optimal candidate, frontier
 * order), copied straight into the caller's arrays.  head_i[slot][0] is the candidate's final status: only slots with
 * MPC_REGION are regions (the others were optimal but lower dimensional); rows are referenced through e_off and are
 * not in slot order.  n_slots <= cap_slots = mpc_level_slots(h); cap_rows from mpc_compact_strides.  With arrays from
 * mpc_host_alloc the copies are direct DMA transfers. */
int64_t mpc_level_slots(const mpc_handle *h);
int mpc_level_regions_slots(mpc_handle *h, double *head_d, int32_t *head_i, int64_t cap_slots, double *erows,
                            int64_t cap_rows, int64_t *n_slots, int64_t *n_rows);
/* mpc_level_regions_slots that returns as soon as head_i has arrived: head_d and erows (the large arrays) are still
 * being written by DMA on the handle's stream when it returns and are complete after the next call that synchronises
 * the handle -- mpc_sync, or any mpc_level_run.  The arrays must be page-locked (mpc_host_alloc) and must stay alive until
 * then.  When some record needs host-side repacking the call behaves like mpc_level_regions_slots. */
int mpc_level_regions_slots_async(mpc_handle *h, double *head_d, int32_t *head_i, int64_t cap_slots, double *erows,
                                  int64_t cap_rows, int64_t *n_slots, int64_t *n_rows);
/* The same, not even waiting for head_i: all three arrays are complete after mpc_sync or the next level of the handle
 * (mpc_level_run, mpc_level_batch_start) -- the driver of a batch queues the fetches of every member, starts the next level for all of
 * them and builds the region objects of the finished level while the device works. */
int mpc_level_regions_slots_nowait(mpc_handle *h, double *head_d, int32_t *head_i, int64_t cap_slots, double *erows,
                                   int64_t cap_rows, int64_t *n_slots, int64_t *n_rows);
/* mpc_level_regions_slots_nowait for every handle of a list in one call (the driver of a batch: one call per level instead of one
 * per member).  Arrays of n_handles entries; a member without regions gets n_slots = n_rows = 0 and may pass null buffers. */
int mpc_level_batch_fetch(mpc_handle **handles, int32_t n_handles, double *const *head_d, int32_t *const *head_i, const int64_t *cap_slots,
                          double *const *erows, const int64_t *cap_rows, int64_t *n_slots, int64_t *n_rows);
int mpc_sync(mpc_handle *h);   /* waits for everything queued on the handle's stream (and for a pending shared record copy of mpc_level_batch_fetch) */
/* mpc_level_batch_fetch copies the records of all members that hold them in slot form with ONE launch (round 5); mpc_fetch_wait waits
 * for that launch on the given device (mpc_sync of any member and the next mpc_level_batch_start do so too) */
int mpc_fetch_wait(int32_t device);
/* ---- the level loop of MANY programs inside the library (round 5) -------------------------------------------------------------------
 * The caller of the combinatorial path for many small programs -- the mixed-integer enumeration maps solve_mpqp over its sub-programs
 * (reference mp_solvers/mpmiqp_enumeration.py:41-50), each of which runs the level loop of mpqp_parrallel_combinatorial.py:102-139 -- as
 * ONE loop on a thread of the library: shared launches of a level for all members (mpc_level_batch_start / _wait), the record copy of
 * the level for all members into three page-locked blocks (mpc_level_batch_fetch), the members' frontier hand-overs, the next level.
 * The device never waits for the caller between levels; the caller only consumes, level by level:
 *   mpc_solve_many_start(handles, n, max_levels[n], flags, &job)   every handle: pruned list cleared, frontier rooted, then the loop
 *          (flags: MPC_LEVEL_KEEP_LOWDIM).  The handles must not be touched until mpc_solve_many_wait has returned.
 *   mpc_solve_many_level(job, level, &info)   blocks until level `level` (0-based) has been run and its records are COMPLETE in the
 *          three blocks (member j's arrays start at off_d[j] / off_i[j] / off_e[j] elements, n_slots[j] slots of the member's fd / fi,
 *          n_rows[j] rows of n_t + 1).  The blocks are the caller's from then on (mpc_host_free; null when no member found a region);
 *          the small arrays of `info` live until mpc_solve_many_wait.  info.n_members == 0: the loop ended before this level --
 *          info.done = 1: every member has run its last level; 2: the next level of the remaining members does not fit the memory budget
 *          of the shared launches together (MPC_BATCH_BUDGET_GB): their frontiers are advanced, their next level has NOT been started,
 *          the caller takes over (admission / parking: the host layer's loop).
 *   mpc_solve_many_wait(job)   joins the loop and frees the job: the loop's return code (the failing member carries the message).
 * flags & MPC_SOLVE_MANY_BASE: when every member has run its last level (not after a hand-over) the loop closes with one more shared level
 * over the BASE active set of every program (the equality rows alone, one candidate each, pruned lists cleared: the reference tests it
 * last, mpqp_parrallel_combinatorial.py:142-146); that level's info carries base = 1. */
#define MPC_SOLVE_MANY_BASE 128
typedef struct {
    int32_t level, n_members, n_shared, done;
    int32_t base, pad_;              /* base = 1: this is the closing level of the base active sets (MPC_SOLVE_MANY_BASE)   */
    const int32_t *member;           /* [n_members] positions in the handle array of mpc_solve_many_start          */
    const mpc_level_stats *stats;    /* [n_members]                                                                 */
    const int64_t *n_slots, *n_rows; /* [n_members] record slots / region rows copied (0: the member found no region) */
    const int64_t *off_d, *off_i, *off_e;   /* [n_members] element offsets inside head_d / head_i / erows            */
    double *head_d;
    int32_t *head_i;
    double *erows;
    int64_t len_d, len_i, len_e;     /* elements of the three blocks                                                */
    double ms_wall;                  /* wall time of the level on the loop's thread                                 */
} mpc_many_level_info;
int mpc_solve_many_start(mpc_handle **handles, int32_t n_handles, const int32_t *max_levels, int32_t flags, void **job);
int mpc_solve_many_level(void *job, int32_t level, mpc_many_level_info *info);
int mpc_solve_many_wait(void *job);
/* Page-locked host memory from a recycling pool (blocks return to the pool on mpc_host_free and are handed out again
 * without re-pinning).  For result arrays that are filled by mpc_level_regions_slots. */
int mpc_host_alloc(uint64_t bytes, void **out);
int mpc_host_free(void *p);
int mpc_level_children(mpc_handle *h, int32_t *children_host, int64_t cap);     /* n_children x (k+1) */
int mpc_level_children_device(mpc_handle *h, int32_t *children_dev, int64_t cap);
int mpc_level_pruned_new(mpc_handle *h, uint64_t *masks_host, int64_t cap);
/* The slot arrays of mpc_level_regions_slots copied device -> device into caller-owned DEVICE buffers (the multi-GPU
 * driver all-gathers them over RCCL without a host round trip; reference: the parent's merge of the workers' regions,
 * mpqp_parrallel_combinatorial.py:127-131).  cap_slots >= mpc_level_slots(h), cap_rows from mpc_compact_strides.
 * MPC_ERR_STATE when some record of the level is not in slot form on the device (regions re-solved by the LDS-engine
 * kernel): the caller then uses mpc_level_regions_slots. */
int mpc_level_regions_device(mpc_handle *h, double *head_d_dev, int32_t *head_i_dev, double *erows_dev, int64_t cap_slots,
                             int64_t cap_rows, int64_t *n_slots, int64_t *n_rows);
int mpc_level_pruned_new_device(mpc_handle *h, uint64_t *masks_dev, int64_t cap);
/* frontier := children of this level; pruned list += sets pruned by this level  (driver :127-135) */
int mpc_frontier_advance(mpc_handle *h);

/* ---- connected-graph traversals with the bookkeeping on the device ----------------------------------------------------- */
/* Reference: mp_solvers/mpqp_combi_graph.py:68-145 (variant 0: sets S / E, explore_subset / explore_superset around every
 * active set whose critical region is non-empty) and mp_solvers/mpqp_graph.py:38-108 (variant 1: attempted, generate_reduce,
 * generate_extra through the facet constraints of every full-dimensional region).  The traversal advances a whole WAVE of
 * active sets at a time; the wave, the set of everything ever queued and the neighbours emitted by the wave stay in HBM as
 * sorted arrays of masks (mpc_mask_words words each), so the order of the regions is deterministic.
 *   mpc_graph_begin       seeds (host masks, duplicates allowed) -> first wave
 *   mpc_graph_wave        the groups of the current wave, one (or, beyond 4M masks, several) per cardinality; n_groups == 0:
 *                         the traversal is complete
 *   mpc_graph_group_run   one group = one frontier of the level kernels (MPC_LEVEL_GRAPH verdicts: neither traversal needs the
 *                         (x,theta) feasibility LP -- without a pruning list "infeasible" and "not optimal" hand on the same
 *                         neighbours); afterwards mpc_level_status / mpc_level_regions_slots etc. describe that group; the
 *                         group's neighbours are appended to the pending list
 *   mpc_graph_wave_close  pending -> sorted, deduplicated, minus everything queued before = the next wave */
int mpc_graph_begin(mpc_handle *h, const uint64_t *seed_masks_host, int64_t n_seeds, int32_t variant);
int mpc_graph_wave(mpc_handle *h, int32_t *k_list, int64_t *count_list, int32_t cap, int32_t *n_groups, int64_t *n_wave, int64_t *n_visited);
int mpc_graph_group_run(mpc_handle *h, int32_t group, mpc_level_stats *stats);
int mpc_graph_wave_close(mpc_handle *h, int64_t *n_next, int64_t *n_visited);

/* ---- the batched operator with host buffers (drop-in for pool.map(full_process)) ---------------------- */
/* Uploads cand (n x k) and the pruned masks (m x mpc_mask_words(h); replaces the handle's list), runs the level
 * and downloads status, regions and children.  On MPC_ERR_CAPACITY *n_regions / *n_children hold the
 * required capacities. */
int mpc_check_level(mpc_handle *h, const int32_t *cand, int64_t n, int32_t k, const uint64_t *pruned_masks, int64_t m,
                    int32_t gen_children, uint8_t *status, int64_t *n_regions, double *rec_d, int32_t *rec_i,
                    int64_t *region_cand, int64_t region_cap, int64_t *n_children, int32_t *children,
                    int64_t children_cap);

/* ---- deterministic-solver plug: a batch of small dense LPs, one wavefront each ------------------------- */
/* min c'x s.t. A x <= b (rows flagged in eq as equalities), x free.  A: n_lp x m x n unless shared_A != 0
 * (then m x n, likewise b with shared_b, c with shared_c; c may be NULL = feasibility); eq: n_lp x m bytes.
 * Outputs (host): status[n_lp]; x[n_lp x n] and obj[n_lp] may be NULL. */
int mpc_lp_solve_batch(int32_t device, int64_t n_lp, int32_t m, int32_t n, const double *A, int32_t shared_A,
                       const double *b, int32_t shared_b, const double *c, int32_t shared_c, const uint8_t *eq,
                       int32_t *status, double *x, double *obj, int32_t *iters);

/* ---- the QP of the program at fixed parameter points, batched: one wavefront per point --------------------------------- */
/* Replaces MPQP_Program.solve_theta (mpqp_program.py:109-143 -> Solver.solve_qp, quad_prog_interface.py:16-89) for programs
 * with positive definite Q:  min 1/2 x'Qx + (c + H theta)'x  s.t.  A x <= b + F theta, first n_eq rows as equalities.
 * theta m x n_t (host).  status[p]: 0 optimal, 1 infeasible at that point, 3 iteration limit.  x (m x n_x, NaN where not
 * optimal), lambda (m x n_c multipliers, >= 0 on inequality rows), active (m x n_c bytes: 1 = the constraint is tight) and
 * iters (complementary pivots) may be NULL.  The constants c_c + c_t'theta + 1/2 theta'Q_t theta are the caller's. */
int mpc_qp_solve_batch(mpc_handle *h, int64_t m, const double *theta_host, int32_t *status, double *x, double *lambda,
                       uint8_t *active, int32_t *iters);

/* ---- Chebyshev centre and radius of every facet of a batch of polytopes, one wavefront per facet ------------------------- */
/* Replaces get_facet_centers (mp_solvers/solver_utils.py:204-250; one chebyshev_ball LP per facet, utils/chebyshev_ball.py:10-63)
 * of the geometric algorithm.  ef_rows: stacked rows [f | E] (n_t + 1 doubles each) of all polytopes, row_off[n_regions + 1];
 * facet q is row q.  Outputs per row: centre (n_t), radius, status (MPC_LP_*; centre and radius are 0 unless optimal). */
int mpc_facet_centres(int32_t device, int32_t n_t, int64_t n_regions, const int64_t *row_off, const double *ef_rows, double *centre,
                      double *radius, int32_t *status);

/* ---- consumer of the path: point location over a solution's critical regions, batched ---------------------------- */
/* Replaces the loop of Solution.get_region / Solution.evaluate (solution.py:45-112, CriticalRegion.is_inside
 * critical_region.py:83-86) for many parameter points at once.
 *   row_off   n_regions + 1 offsets into ef_rows;  ef_rows  total_rows x (n_t+1): [f | E] of every region, stacked
 *   xlaw      n_regions x n_x x (n_t+1): [b | A] of every region
 *   Q, c, H   objective of the program (n_x x n_x, n_x, n_x x n_t; any may be NULL) -- only used with overlapping != 0
 * mpc_locator_query: theta m x n_t (host).  region[p] = index of the first region that contains the point (flags without
 * MPC_LOCATE_OVERLAPPING) or of the containing region with the lowest objective, ties to the later one (with it); -1 if
 * none.  "Contains" is all(E theta - f < tol), the strict test of CriticalRegion.is_inside (critical_region.py:83-86), or
 * with MPC_LOCATE_INCLUSIVE all(E theta <= f + tol): with tol = 0 the test `A @ theta <= b` of the reference's
 * PointLocation (upop/point_location.py:46,59), under which points on a facet belong to the region.
 * x (m x n_x, may be NULL) = A theta + b of that region, NaN where there is none. */
#define MPC_LOCATE_OVERLAPPING 1
#define MPC_LOCATE_INCLUSIVE 2
/* MPC_LOCATE_WALK: locate by walking through adjacent regions instead of scanning the list (needs
 * mpc_locator_set_adjacency; ignored with the two flags above).  Same result as the scan -- the first region of the list that
 * contains the point within tol -- at a cost that follows the length of the walk, not the number of regions; points the walk
 * cannot resolve (no neighbour behind any violated row, step limit) are tested against every region in parallel. */
#define MPC_LOCATE_WALK 4
typedef struct mpc_locator mpc_locator;
int mpc_locator_create(int32_t device, int32_t n_x, int32_t n_t, int64_t n_regions, const int64_t *row_off, const double *ef_rows,
                       const double *xlaw, const double *Q, const double *c, const double *H, mpc_locator **out);
/* Facet adjacency of a complete, non-overlapping solution: masks n_regions x mask_words (active set of every region, all
 * different), row_info one int per stacked row = kind << 16 | id with kind 0: multiplier row of active constraint id (behind
 * it: the region without id), 1: row of inactive constraint id (behind it: the region with id added), 2: row of the
 * parameter set (behind it: nothing), 3: unknown. */
int mpc_locator_set_adjacency(mpc_locator *loc, int32_t mask_words, int32_t n_c, const uint64_t *masks, const int32_t *row_info);
int mpc_locator_query(mpc_locator *loc, int64_t m, const double *theta, double tol, int32_t flags, int64_t *region,
                      double *x, float *ms_locate);
int mpc_locator_destroy(mpc_locator *loc);

#ifdef __cplusplus
}
#endif
#endif /* MPCOMBI_H */
