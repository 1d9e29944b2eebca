"""Mixed-integer enumeration: the caller of the combinatorial path that SURVEY.md §8(f) ranks second
(reference: mp_solvers/mpmiqp_enumeration.py:12-64).

  1. the feasible binary fixations -- one device batch of LPs (MPMILP_Program.leaf_feasibility);
  2. per fixation the substituted continuous mpLP/mpQP, presolved (its redundancy LPs are a device batch) and solved
     by the MI355X combinatorial path (`solve_mpqp`);
  3. regions tagged with their fixation and concatenated, in fixation order.

With a combinatorial ``cont_algorithm`` (the default) the sub-programs are solved TOGETHER, level by level, every stage of a
level being one launch for all of them (``mpqp_hip_combinatorial.solve_many`` -> mpc_level_run_batch, "several programs per
launch"); the regions are those of the one-by-one solves (bit for bit up to the qualification in csrc/batch_level.hpp).  ``MPC_NO_BATCH=1`` keeps the one-by-one form below.

``num_cores`` keeps its meaning of "sub-problems in flight": the reference maps them over a process pool
(mpmiqp_enumeration.py:46-50); here each in-flight sub-problem is a host thread driving its own device handle and
HIP streams (the C ABI releases the GIL, handles are independent, pools are locked), so the small kernels of
different sub-problems overlap on the GPU and one sub-problem's host bookkeeping hides behind another's kernels.
``num_cores=1`` runs them one after the other; -1 picks min(8, host cores).
"""
import os
import warnings
from concurrent.futures import ThreadPoolExecutor

from ..solution import Solution
from ..utils.general_utils import num_cpu_cores
from . import mpqp_hip_combinatorial
from .solve_mpqp import _COMBINATORIAL, mpqp_algorithm, solve_mpqp

BATCH_CHUNKS = int(os.environ.get('MPC_BATCH_CHUNKS', '2'))   # chunks of fixations solved together (see below)
MAX_BATCH = int(os.environ.get('MPC_MAX_BATCH', '256'))       # most sub-programs alive at once (one host thread each while they are constructed)


def solve_mpmiqp_enumeration(program, num_cores: int = -1,
                             cont_algorithm: mpqp_algorithm = mpqp_algorithm.combinatorial) -> Solution:
    feasible_combinations = program.feasible_combinations()
    if num_cores == -1:
        num_cores = min(8, num_cpu_cores())

    def solve_fixation(fix):
        with warnings.catch_warnings():        # the substituted program repeats the parent's construction warnings
            warnings.simplefilter('ignore')
            sub = program.generate_substituted_problem(fix)
            try:
                return solve_mpqp(sub, cont_algorithm)
            finally:
                sub.release_engine()       # the handle's device blocks go straight to the next sub-problem

    batched = cont_algorithm in _COMBINATORIAL and len(feasible_combinations) > 1 and os.environ.get('MPC_NO_BATCH', '0') != '1'
    if batched:
        device = int(getattr(getattr(program, 'solver', None), 'device', 0) or 0)
        prune_lowdim = cont_algorithm is mpqp_algorithm.combinatorial_parallel

        def substitute(fixes):
            """The sub-programs of ``fixes``, presolved and set up, WITHOUT a thread per construction (round 4): every fixation is
            substituted and brought through the LP-free part of the constructor in this thread (the rows that carry continuous content
            are shared by all fixations: only the right-hand side differs), the redundancy LPs of ALL sub-programs are posed as one device
            batch per shape (Solver.lp_feasible_many; statuses only), their results applied, and the set-ups (mpc_create: C, no interpreter
            lock) run on a few threads.  The two diagnostic LPs of the constructor, whose only outcome is a warning that this function
            discards, are not posed.  (Round 3: one thread per fixation with the LP calls of all threads coalesced -- 65 ms for 64
            sub-programs, of which half was interpreter-lock hand-over; MPC_MI_THREADS=1 keeps that form.)"""
            import copy
            from ..solver import LPCoalescer
            if os.environ.get('MPC_MI_THREADS', '0') != '1' and hasattr(program.solver, 'lp_feasible_many'):
                with warnings.catch_warnings():
                    warnings.simplefilter('ignore')
                    subs = [program.generate_substituted_problem(fix, deferred=True) for fix in fixes]
                    requests = [sub._redundancy_request() for sub in subs]
                    answers = program.solver.lp_feasible_many([(PA, Pb, [[*eq, i] for i in todo]) for PA, Pb, eq, todo in requests])
                    for sub, req, ok in zip(subs, requests, answers):
                        sub._redundancy_apply(req, ok.tolist())
                    if len(subs) > 2 and num_cores != 1:
                        with ThreadPoolExecutor(max_workers=min(8, len(subs))) as setup_pool:
                            list(setup_pool.map(lambda sub: sub.engine(device, closed=True), subs))
                    else:
                        for sub in subs:
                            sub.engine(device, closed=True)
                return subs
            if len(fixes) <= 1 or num_cores == 1 or os.environ.get('MPC_NO_LP_COALESCE', '0') == '1':
                with warnings.catch_warnings():    # the substituted programs repeat the parent's construction warnings
                    warnings.simplefilter('ignore')
                    subs = [program.generate_substituted_problem(fix) for fix in fixes]
                    for sub in subs:
                        sub.engine(device, closed=True)     # set-up of the sub-program (MFMA set-up kernel)
                    return subs
            co = LPCoalescer(program.solver, len(fixes))
            parked = copy.copy(program)
            parked.solver = co.solver()

            def one(fix):
                try:
                    sub = parked.generate_substituted_problem(fix)
                finally:
                    co.worker_done()
                sub.solver = program.solver
                sub.engine(device, closed=True)
                return sub
            # exactly one thread per fixation, all alive until every construction is through: the coalescer counts on each of them
            # reaching its next LP call (an executor may run two constructions on one thread, one after the other)
            import threading
            results, errors = [None] * len(fixes), []

            def run(j):
                try:
                    results[j] = one(fixes[j])
                except BaseException as ex:        # re-raised below, in the caller's thread
                    errors.append(ex)
            with warnings.catch_warnings():        # one filter around all threads (catch_warnings is not thread-safe)
                warnings.simplefilter('ignore')
                threads = [threading.Thread(target=run, args=(j,)) for j in range(len(fixes))]
                for th in threads:
                    th.start()
                for th in threads:
                    th.join()
            if errors:
                for sub in results:
                    if sub is not None:
                        sub.release_engine()
                raise errors[0]
            return results

        # MPC_BATCH_CHUNKS > 1: the fixations are solved in chunks -- while the device works on the levels of one chunk (the host waits
        # inside the C ABI, GIL released) a second thread substitutes, presolves and sets up the sub-programs of the next one.  Measured
        # on the bench workload (64 fixations): 1 chunk 345 ms, 2 chunks 368, 4 chunks 350-435, 8 chunks 425-450 -- the host work of
        # the two threads shares one interpreter lock and smaller batches fill the device less.  Round 5: the level loop of a chunk runs
        # inside the library (mpc_solve_many_*), this thread sleeps in the C ABI for most of a chunk, and the second thread's constructions
        # no longer fight for the lock: 1 chunk 112 ms, 2 chunks 103-106, 3 chunks 105-110, 4 chunks 118 (tools/mi_phases.py); default 2.
        n_fix = len(feasible_combinations)
        chunk = min(MAX_BATCH, n_fix if num_cores <= 1 else max(8, -(-n_fix // BATCH_CHUNKS)))
        chunks = [feasible_combinations[i:i + chunk] for i in range(0, n_fix, chunk)]
        sols = []
        with ThreadPoolExecutor(max_workers=1) as pool:
            pending = pool.submit(substitute, chunks[0])
            for j in range(len(chunks)):
                subs = pending.result()
                if j + 1 < len(chunks):
                    pending = pool.submit(substitute, chunks[j + 1])
                try:
                    sols.extend(mpqp_hip_combinatorial.solve_many(subs, device=device, prune_lowdim=prune_lowdim))
                finally:
                    for sub in subs:
                        sub.release_engine()
        for sol in sols:                       # what solve_mpqp sets for one program (solve_mpqp.py: every MPLP_Program instance)
            sol.is_overlapping = True
    elif num_cores <= 1 or len(feasible_combinations) <= 1:
        sols = [solve_fixation(fix) for fix in feasible_combinations]
    else:
        with ThreadPoolExecutor(max_workers=num_cores) as pool:
            sols = list(pool.map(solve_fixation, feasible_combinations))       # results in fixation order

    collected = []
    for fix, sol in zip(feasible_combinations, sols):
        batches = getattr(sol, 'region_batches', None)
        if batches is not None:                # lazy regions: the three fields are read from their batch (region_batch.py)
            for batch in batches:
                batch.y_fixation, batch.y_indices, batch.x_indices = fix, program.binary_indices, program.cont_indices
        for region in (sol.critical_regions if batches is None else sol.loose_regions):
            region.y_fixation = fix
            region.y_indices = program.binary_indices
            region.x_indices = program.cont_indices
        collected.extend(sol.critical_regions)
    return Solution(program, collected, is_overlapping=True)
