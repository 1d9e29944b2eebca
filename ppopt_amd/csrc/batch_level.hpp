// batch_level.hpp -- one level of MANY programs per launch ("batch several programs per launch").
//
// Reference counterpart: mp_solvers/mpmiqp_enumeration.py:41-50 maps solve_mpqp over the feasible binary fixations, one
// sub-program after the other (or one per pool worker).  Here the sub-programs advance level by level TOGETHER: every stage of the
// level -- KKT solves, theta LPs, partitions, (x,theta) LPs, region construction, pruned masks, children -- is ONE launch whose
// blockIdx.y selects the member program and whose arguments come from a table in device memory (one BatchMember per program),
// instead of one launch per stage and program.  The kernels' bodies are the single-program kernels of kernels.hpp / kernels2.hpp,
// statement for statement (MPC_GLOBAL, kernels.hpp).  A member's level computes what mpc_level_run computes for it: bit for bit the same
// statuses, children, pruned masks and region records, with ONE qualification.  A single program launches its region stage on the
// candidates the theta stage found optimal and hands the few that turn out optimal later (a re-solved doubtful (x,theta) run) to the
// LDS-engine region kernel; here the region stage comes last and builds all of them with k_region2, and the number of wavefronts
// sharing a candidate follows the larger count.  Those regions are the same sets with coefficients equal to ~1e-10; a sliver facet
// may be kept by one form and dropped by the other (tools/fuzz_batch.py: 600 random programs, 1,217,481 regions -- 2 regions with a
// different facet list, 3 with coefficients differing by at most 6.5e-11, all others bit-identical).  Against the single-program
// form with the region stage behind the (x,theta) stage (MPC_NO_ROVERLAP=1, which builds every optimal candidate with k_region2 as
// well) the same fuzz finds NO difference: all 1,217,481 regions bit for bit.
//
// Round 4: the number of wavefronts sharing a candidate in k_region2 follows the member's SHARE of the launch (below), one when the
// device is full, where a single program's follows its own width (up to eight on an idle device).  Each of them repeats the Chebyshev
// LP, which pays on idle CUs only: 64 sub-programs of the bench enumeration 83 -> 67 ms of device time.  A different split walks the
// facets in another order; with the same split (MPC_NO_RSPLIT=1 on both sides) every record is bit for bit the single program's, with
// each side's own the x-law, multipliers, statuses, children and pruned masks still are, and a facet list may differ where it sits on
// the LP tolerance (tests/test_gpu_batch.py `_own_split`; tools/fuzz_batch.py: 999,231 regions of 600 random programs, 2 other facet
// lists, 18 regions with coefficients differing by <= 3.4e-11; tools/batch_w_debug2.py: 14 of 12,871 regions, in each of them the CPU
// oracle's own list changes when its 1e-7 tolerance moves two decades).
//
// Wavefront shares.  The persistent kernels run fastest with FEW wavefronts per SIMD (k_theta2 two, k_x2 three, k_xq five, k_region2
// two; mpcombi_hip.hip) -- for the launch as a whole.  batch_level_launch gives every member a share of that budget in proportion to
// its candidates (BatchMember::*_blocks; a member's surplus blocks leave at once), so the members of a group run side by side and end
// together.  Round 3 gave each member the width of a single program's launch -- 64 members x 4,096 blocks, one member's tail after the
// other: the bench enumeration's shared levels took 143 ms of device time, with shares 83 ms (MPC_BATCH_SHARES=0 A/B), 67 with the split above.
//
// The level is the no-round-trip form of mpcombi_hip.hip (level_run_small): list lengths live in device memory, launches are sized
// by the members' candidate counts, the host synchronises once per level for ALL members.  Members whose kernels are different
// template instantiations (n_theta class, rows per lane, mask words, cardinality) form separate groups of the same launch sequence.
#pragma once
#include <string>

#include "kernels2.hpp"

namespace mpc {

// use_kkt == 2: k_kkt_thread with this many lanes per candidate (levels of a batch are small by construction), up to KMAX inequality rows
constexpr int BATCH_KKT_SPREAD = 8, BATCH_KKT_SPREAD_KMAX = 6, BATCH_KKT_SPREAD_THREADS = 1 << 19;

struct BatchZero { void *p; unsigned long long bytes; };

struct BatchMember {
    int id;                  // position in the caller's array (batch_level_launch reorders the members into groups)
    // ---- what selects the kernels (members of one group agree on all of these) ------------------------------------------------
    int k, kd, fast_t, fast_x, fast_r, mw, use_kkt /* 0, 1, 2: spread */, kkt_listed, quick_test, gen_children;
    // ---- sizes ------------------------------------------------------------------------------------------------------------------
    long long n;             // candidates of the level
    int grid_f, grid_r2, n_cu, lds_f, lds_v, lds_r2, rsplit_max;
    int W, ldk, fd, fi, no_rbox, nxc, storing, keep_lowdim;
    // the member's share of the group's launches (batch_level_launch: wavefronts in proportion to the candidates; 0 = no bound)
    int th_blocks, xq_blocks, x2_blocks, r2_blocks, r2_wcap;
    int theta_open;          // the member's parameter set is open in some direction: k_recession behind the verdict stages (mpcombi_hip.hip)
    int lds_r, rcap;         // LDS-engine region kernel (candidates k_region2 gives up on): dynamic LDS, record slots reserved
    long long rec_d, rec_i;  // strides of its fixed-layout records
    // ---- the arguments of level_run_small's launches ---------------------------------------------------------------------------
    const DevProblem *pf, *pr;
    DevProblem Pv, Pr;
    const int32_t *fr;
    uint8_t *status;
    LevelCounters *ctr;
    int32_t *dcnt;
    uint8_t *kkt_code;
    double *kkt_L;
    int32_t *theta_list, *retry_list, *part_lists;
    double *headd;
    int32_t *headi;
    double *epool;
    uint8_t *kept_g;
    unsigned int *done_g;
    double *recd;
    int32_t *reci;
    uint8_t *facet_flags;
    ThetaArgs targs;
    RegionStream rs;
    DictCache dc;
    unsigned long long *pruned;
    long long n_pruned;
    unsigned long long *childmask;
    int32_t *count, *offset, *children, *parent_slot_next;
    const uint8_t *dict_stored_cur;
    unsigned int *pub_ctr, *pub_cnt;     // device aliases of the member's pinned counters block
    BatchZero zero[6];                   // cleared before the first launch
    int n_zero;
    // round 5: one-step plans on a storing level whose candidates have parents' records (the single program's k_xq_thread plan mode + k_x1):
    // plan = 1: the member takes part; x1_buf: 6 n ints (plan slot, plan step, k_x1's list, the three lists left to k_x2); lengths in
    // dcnt[9] (k_x1's list) and dcnt[29..31]; alt: the previous frontier for the look-up of other parents (empty: generating parent only)
    int plan, plan_blocks, x1_blocks;
    int32_t *x1_buf;
    XqAlt alt;
};

// ---- one program, small level: the region kernel and the (x,theta) kernel in ONE launch (round 5) ------------------------------------------
// On a level of a few thousand candidates both kernels are 20-80 us of latency with a handful of wavefronts each; they work on disjoint
// candidates (optimal / open + dictionary-only), so the no-round-trip path launches them as ONE grid: blockIdx.y = 0 runs k_region2's body,
// blockIdx.y = 1 k_x2's (the x-extent is the larger of the two needs; surplus blocks of either leave at once) -- the overlap the classic
// path gets from a second stream, without the two event hops that cost as much as it saves at this size.
struct SmallRX {
    // k_region2's arguments
    const DevProblem *pr; const int32_t *fr; int k; const int32_t *opt_list; int n; uint8_t *status; double *headd; int32_t *headi; int fd, fi;
    double *epool; LevelCounters *ctr; const uint8_t *kkc; const double *kkl; int W; uint8_t *kept_g; int ldk; unsigned int *done_g; const double *tvp_box; RegionStream rs;
    // k_x2's
    const DevProblem *pf; const int32_t *list; DictCache dc;
};
// false: no merged instantiation for this pair of kernel selectors (the caller launches the two kernels one after the other)
bool small_region2_x2_launch(int fast_r, int fast_x, unsigned grid_r, unsigned grid_x, int lds_r2, hipStream_t st, const SmallRX &a, hipError_t *err);

// Queues the launches of one level for the B members on `st` (no synchronisation).  The members are reordered into groups.
// tab_host (page-locked) / tab_dev: room for B BatchMember each, owned by the caller until the launches have completed (the kernels
// read their arguments from tab_dev).  Returns hipSuccess or the first error.
hipError_t batch_level_launch(BatchMember *members, int B, hipStream_t st, BatchMember *tab_host, BatchMember *tab_dev);

}  // namespace mpc
