"""The parallel combinatorial algorithm (Gupta et al. 2011) with the per-candidate work on the MI355X.

Same level-synchronous structure and pruning semantics as the reference driver
(mp_solvers/mpqp_parrallel_combinatorial.py:67-150):

    to_check = children(equality_indices)                      driver :98
    for depth in range(max_depth):                             driver :102
        outputs = pool.map(full_process, to_check)             driver :110-116  -> Engine.level_run (device kernels)
        murder_list += pruned ; to_check = children ; regions  driver :127-135  -> Engine.frontier_advance
    finally the base active set itself                         driver :142-146

The frontier, the pruned list and the children never leave HBM between levels; per level the host reads back a
small statistics block and the region records.  ``solve`` drives ONE GPU (``device``).  The multi-GPU form of the same
loop is ``ppopt_amd.distributed.solve_distributed`` (one process per GPU under ``torch.distributed``): small levels are
processed by every rank, the first large level is split ``to_check[rank::world]``, afterwards children stay on the GPU
of their parent and the ranks exchange only the newly pruned sets per level and the regions at the end -- children are
never moved between ranks.

``prune_lowdim`` selects between the two pruning rules of the reference's drivers (see ``solve``); ``solve_mpqp`` maps
``combinatorial_parallel`` to the parallel driver's rule and ``combinatorial`` / ``combinatorial_parallel_exp`` to the serial one.

Differences from the reference that are visible to a user:
  * ``shuffle(to_check)`` (driver :114) is dropped -- candidate order is deterministic, and so is the region order;
  * a numerically singular KKT matrix is a per-candidate status (no region, children expanded) instead of the
    ``numpy.linalg.LinAlgError`` that aborts the reference solve (mpqp_program.py:187).
"""
import os
import time
from typing import Dict, List, Optional

import numpy

from ..critical_region import CriticalRegion
from ..region_batch import RegionBatch, gc_paused
from ..solution import Solution


def unpack_regions(rec_d: numpy.ndarray, rec_i: numpy.ndarray, n_x: int, n_t: int, n_c: int, n_tc: int) -> List[CriticalRegion]:
    """Device region records (layout: include/mpcombi.h, mpc_level_regions) -> CriticalRegion objects.  The
    matrices are sliced for all records at once; E/f arrive already free of exact duplicates (k_region)."""
    n = len(rec_d)
    if n == 0:
        return []
    o = 0
    A = rec_d[:, o:o + n_x * n_t].reshape(n, n_x, n_t); o += n_x * n_t
    b = rec_d[:, o:o + n_x].reshape(n, n_x, 1); o += n_x
    C = rec_d[:, o:o + n_c * n_t].reshape(n, n_c, n_t); o += n_c * n_t
    d = rec_d[:, o:o + n_c].reshape(n, n_c, 1); o += n_c
    E = rec_d[:, o:o + (n_c + n_tc) * n_t].reshape(n, n_c + n_tc, n_t); o += (n_c + n_tc) * n_t
    f = rec_d[:, o:o + n_c + n_tc].reshape(n, n_c + n_tc, 1)
    hdr = rec_i[:, :5].tolist()
    q = 5
    act = rec_i[:, q:q + n_c].tolist(); q += n_c
    om = rec_i[:, q:q + n_tc].tolist(); q += n_tc
    la = rec_i[:, q:q + n_c].tolist(); q += n_c
    ridx = rec_i[:, q:q + n_c].tolist(); q += n_c
    rcon = rec_i[:, q:q + n_c].tolist()
    out = []
    for j in range(n):
        k, nE, n_om, n_la, n_re = hdr[j]
        out.append(CriticalRegion(A[j], b[j], C[j, :k], d[j, :k], E[j, :nE], f[j, :nE], act[j][:k], om[j][:n_om],
                                  la[j][:n_la], [ridx[j][:n_re], rcon[j][:n_re]]))
    return out


def unpack_region(rec_d: numpy.ndarray, rec_i: numpy.ndarray, n_x: int, n_t: int, n_c: int, n_tc: int) -> CriticalRegion:
    return unpack_regions(rec_d.reshape(1, -1), rec_i.reshape(1, -1), n_x, n_t, n_c, n_tc)[0]


REGION_STATUS = 3   # MPC_REGION
# levels below this size are run synchronously and fetched afterwards: the hand-overs to the worker thread cost more than the
# overlap gains there (sub-programs of the mixed-integer enumeration, the first levels of every solve)
STREAM_MIN_CANDIDATES = int(os.environ.get('MPC_STREAM_MIN', '512'))
BASE_ON_TWIN = os.environ.get('MPC_NO_TWIN', '0') != '1'   # the base-set check on a second handle, started with the first streamed level
MANY_LOOP = os.environ.get('MPC_NO_MANY_LOOP', '0') != '1'     # solve_many: the level loop of all members inside the library (mpc_solve_many_start); '1' = from here (A/B, tests)
SOLVE_LOOP = os.environ.get('MPC_NO_SOLVE_LOOP', '0') != '1'   # the level loop inside the library (mpc_solve_start); '1' = level by level from here (A/B)


def _closing_rows_unused(eng, solution) -> bool:
    """True unless some region lists one of the closing rows MPLP_Program._engine_parameter_rows appended to a parameter set without
    a vertex (they are strictly redundant: this is a guard, not an expected outcome)."""
    n_own = getattr(eng, 'n_tc_program', eng.n_tc)
    if n_own >= eng.n_tc:
        return True
    return all(max(cr.omega_set, default=-1) < n_own for cr in solution.critical_regions)


def solve(program, num_cores: int = -1, device: int = 0, profile: Optional[List[Dict]] = None,
          collect_regions: bool = True, max_levels: Optional[int] = None, stream: Optional[bool] = None,
          prune_lowdim: bool = True) -> Solution:
    """``_solve`` below; should a level report more late optimal candidates than its overlapped region stage had reserved record
    slots for (MPC_ERR_CAPACITY, include/mpcombi.h: mpc_set_region_overlap -- never observed, forced in the tests), the solve is
    repeated with the region stage behind the (x,theta) stage, where nothing can be late.  No candidate is ever demoted."""
    with gc_paused():      # (round 6) ~10^4 region objects per solve, none in a cycle: region_batch.gc_paused / _promote_young
        return _solve_guarded(program, num_cores, device, profile, collect_regions, max_levels, stream, prune_lowdim)


def _solve_guarded(program, num_cores, device, profile, collect_regions, max_levels, stream, prune_lowdim) -> Solution:
    from .._lib import MpcCapacityError, MpcError
    closed = True
    try:
        sol = _solve(program, num_cores, device, profile, collect_regions, max_levels, stream, prune_lowdim)
        if not _closing_rows_unused(program.engine(device, closed=True), sol):
            # never observed: a closing row of the parameter set in a region -- the program is solved again on its own rows only
            # (an argument, not an environment variable: other threads of a mixed-integer enumeration keep their closed handles)
            closed = False
            if profile is not None:
                del profile[:]
            sol = _solve(program, num_cores, device, profile, collect_regions, max_levels, stream, prune_lowdim, closed=False)
        return sol
    except MpcCapacityError:
        eng = program.engine(device, closed=closed)
        twin = getattr(eng, '_twin', None)
        if twin is not None:
            try:
                twin.level_wait()      # the base-set check that was started beside the failed solve
            except MpcError:
                pass
        eng.set_region_overlap(False)
        if profile is not None:
            del profile[:]
        try:
            return _solve(program, num_cores, device, profile, collect_regions, max_levels, stream, prune_lowdim, closed=closed)
        finally:
            eng.set_region_overlap(True)      # the slower form was for this solve only


def _solve(program, num_cores: int = -1, device: int = 0, profile: Optional[List[Dict]] = None,
           collect_regions: bool = True, max_levels: Optional[int] = None, stream: Optional[bool] = None,
           prune_lowdim: bool = True, closed: bool = True) -> Solution:
    """Solves the mpLP/mpQP on one GPU.  ``num_cores`` is accepted for signature compatibility with the reference
    drivers and ignored.  ``profile``: optional list that receives one dict of statistics per level.  ``stream``: region
    records are streamed to the host while the region kernel runs (default; ``MPC_NO_STREAM=1`` or False = fetch after
    each level).  ``prune_lowdim``: True = the parallel driver's rule, an active set that is optimal with a lower-dimensional
    region is pruned together with its supersets (mpqp_parrallel_combinatorial.py:57-59); False = the serial driver's rule
    (mpqp_combinatorial.py:44-61, also the _exp parallel driver), every feasible set is expanded -- more candidates, and on
    degenerate programs more regions."""
    if stream is None:
        stream = os.environ.get('MPC_NO_STREAM', '0') != '1'
    eng = program.engine(device, closed=closed)
    eng.set_timing(profile is not None)      # HIP-event records inside the levels only when somebody reads their times
    n_x, n_t, n_c, n_tc = eng.n_x, eng.n_t, eng.n_c, eng.n_tc
    solution = Solution(program, [])
    max_depth = max(n_x, n_t) - eng.n_eq
    if max_levels is not None:
        max_depth = min(max_depth, max_levels)
    if collect_regions and stream and SOLVE_LOOP and max_depth > 0:
        return _solve_in_library(program, eng, solution, max_depth, profile, prune_lowdim, max_levels)
    eng.pruned_clear()
    eng.frontier_root()
    base_behind_level = False
    twin = None      # second handle of the program: the base-set check runs there, under this handle's large levels
    for depth in range(max_depth):
        gen_children = depth + 1 != max_depth
        t0 = time.perf_counter()
        if collect_regions and stream and eng.frontier_info()[0] >= STREAM_MIN_CANDIDATES:
            # The level runs on the handle's worker thread; its region kernel writes the records straight into page-locked
            # host arrays and raises a flag per chunk of slots, so the region objects of a chunk are built while the kernel
            # is still working on the later ones -- no fetch afterwards, nothing waits for Python.
            if twin is None and BASE_ON_TWIN:
                try:
                    twin = eng.twin()
                    twin.level_start(False, only_base=True)
                except Exception:
                    twin = None
            eng.level_start(gen_children, stream=True, then_base=not gen_children and twin is None, keep_lowdim=not prune_lowdim)
            base_behind_level = not gen_children and twin is None
            info = eng.level_stream_info()
            new_regions: List[CriticalRegion] = []
            batch = None
            if info is not None:
                hd, hi, er, chunk, n_chunks = info
                batch = RegionBatch(hd, hi, er, n_x, n_t, n_c, n_tc, eng.frontier_info()[1], ())
                status_col = hi[:, 0]
                for j in range(n_chunks):
                    eng.level_chunk_wait(j)
                    lo = j * chunk
                    new_regions.extend(batch.regions_of((lo + numpy.flatnonzero(status_col[lo:lo + chunk] == REGION_STATUS)).tolist()))
            st = eng.level_wait()
            if st.n_regions and info is None:
                # a level outside the streaming path (run without host round trips: its records are in device buffers; LDS-engine
                # region kernel; very large record sets): the integer heads are waited for, the large arrays arrive under the next level
                hd, hi, er, kk, slots = eng.level_regions_slots(early_return=True)
                new_regions = RegionBatch(hd, hi, er, n_x, n_t, n_c, n_tc, kk, slots).regions()
            elif st.n_region_retry and batch is not None:
                # some candidates were re-solved by the LDS-engine kernel after the stream: fill their slots, list the level again
                eng.level_stream_fixup(hd, hi, er)
                new_regions = batch.regions_of(numpy.flatnonzero(hi[:, 0] == REGION_STATUS).tolist())
            solution.critical_regions.extend(new_regions)
        else:
            # small level, in this thread: the region kernel still writes its records straight into page-locked host arrays
            # (complete when the call returns), so nothing is fetched afterwards
            st = eng.level_run(gen_children, keep_lowdim=not prune_lowdim, stream=bool(collect_regions and stream))
            if collect_regions and st.n_regions:
                info = eng.level_stream_info() if stream else None
                if info is not None:
                    hd, hi, er, _, _ = info
                    if st.n_region_retry:
                        eng.level_stream_fixup(hd, hi, er)
                    batch = RegionBatch(hd, hi, er, n_x, n_t, n_c, n_tc, eng.frontier_info()[1], ())
                    solution.critical_regions.extend(batch.regions_of(numpy.flatnonzero(hi[:, 0] == REGION_STATUS).tolist()))
                else:
                    # the integer heads are waited for (the region objects are built from them); the two large arrays keep
                    # arriving by DMA while Python builds the objects -- eng.sync() below completes them
                    hd, hi, er, kk, slots = eng.level_regions_slots(early_return=True)
                    solution.critical_regions.extend(RegionBatch(hd, hi, er, n_x, n_t, n_c, n_tc, kk, slots).regions())
        if profile is not None:
            profile.append(_level_profile(depth + 1, st, (time.perf_counter() - t0) * 1e3))
        if not gen_children or st.n_children == 0:
            break
        eng.frontier_advance()
    eng.sync()
    # the base active set (= the equality rows) is tested last, like the reference (driver :142-146)
    res = eng.base_result() if base_behind_level else None      # the worker may have checked it behind the last level
    if twin is not None:
        twin.level_wait()
        res = twin.base_result()
    if res is None:
        base = numpy.arange(eng.n_eq, dtype=numpy.int32).reshape(1, -1)
        status, rd, ri, _, _ = eng.check_level(base, numpy.zeros((0, 2), dtype=numpy.uint64), False)
    else:
        status, rd, ri = res
    if profile is not None:
        profile.append({'depth': 0, 'k': eng.n_eq, 'candidates': 1, 'status': numpy.bincount(status, minlength=6).tolist(),
                        'regions': len(rd)})
    if collect_regions and len(rd):
        solution.add_region(unpack_region(rd[0], ri[0], n_x, n_t, n_c, n_tc))
    solution.is_complete = collect_regions and max_levels is None      # every cardinality up to max(n_x, n_theta) was enumerated
    return solution


def _level_profile(depth, st, ms_wall):
    return {'depth': depth, 'k': int(st.k), 'candidates': int(st.n),
            'status': [int(v) for v in st.n_status], 'regions': int(st.n_regions),
            'children': int(st.n_children), 'pruned_new': int(st.n_pruned_new),
            'lp_pivots': int(st.lp_pivots), 'xtheta_lps': int(st.n_xtheta_lp), 'xtheta_fallbacks': int(st.n_xtheta_fallback), 'ms_verdict': float(st.ms_verdict),
            'ms_region': float(st.ms_region), 'ms_children': float(st.ms_children),
            'ms_theta': float(st.ms_theta), 'ms_x': float(st.ms_x), 'ms_region2': float(st.ms_region2), 'region_side_stream': bool(st.region_side_stream),
            'n_x_items': int(st.n_x_items), 'n_opt': int(st.n_opt), 'n_theta_items': int(st.n_theta_items),
            'dict_read_bytes': int(st.dict_read_bytes), 'dict_write_bytes': int(st.dict_write_bytes),
            'ms_kkt': float(st.ms_kkt), 'ms_xq': float(st.ms_xq), 'n_xq_items': int(st.n_xq_items), 'xq_pivots': int(st.xq_pivots),
            'n_xq_thread': int(st.n_xq_thread), 'ms_xq_thread': float(st.ms_xq_thread), 'xq_thread_beside_theta': bool(st.xq_thread_beside_theta),
            'n_x1': int(st.n_x1), 'ms_x1': float(st.ms_x1), 'ms_x_plan': float(st.ms_x_plan),
            'xq_record': [int(st.xq_record_ints), int(st.xq_record_rows), int(st.xq_record_cols)],
            'ms_wall': ms_wall}


def _solve_in_library(program, eng, solution, max_depth, profile, prune_lowdim, max_levels) -> Solution:
    """The level loop of the reference's driver (mpqp_parrallel_combinatorial.py:102-139) runs INSIDE the library, on the handle's
    worker thread (``Engine.solve_start`` -> mpc_solve_start): root frontier, level, frontier hand-over, next level -- the device never
    waits for this interpreter between levels.  This thread only consumes: per level it is handed the region records (large levels:
    page-locked arrays the region kernel is still writing, chunk by chunk; small ones: complete arrays) and builds the region objects
    while later levels already run.  The base active set (driver :142-146) is checked on the program's second handle beside the levels.
    ``MPC_NO_SOLVE_LOOP=1`` keeps the level-by-level loop in ``_solve``."""
    n_x, n_t, n_c, n_tc = eng.n_x, eng.n_t, eng.n_c, eng.n_tc
    twin = None
    # The second handle costs a set-up (~0.5 ms) and saves the base-set check at the end of a solve (~0.1 ms): a program's FIRST solve
    # checks the base set on its own handle behind the last level, the second handle is made when the program is solved again
    # (tools/first_solve.py: config 2's first solve 1.85 ms with the second handle against 1.1 ms in the steady state).
    if BASE_ON_TWIN and (getattr(eng, '_solved_before', False) or getattr(eng, '_twin', None) is not None):
        try:
            twin = eng.twin()
            twin.level_start(False, only_base=True)
        except Exception:
            twin = None
    eng._solved_before = True
    eng.solve_start(max_depth, stream=True, fetch=True, then_base=twin is None, keep_lowdim=not prune_lowdim)
    regions = solution.critical_regions
    level = 0
    consumed = False
    try:
        while True:
            info = eng.solve_level(level)
            if info is None:
                break
            mode, k, _, hd, hi, er, chunk, n_chunks = info
            first = len(regions)
            if mode == 1:
                batch = RegionBatch(hd, hi, er, n_x, n_t, n_c, n_tc, k, ())
                status_col = hi[:, 0]
                for j in range(n_chunks):
                    eng.solve_chunk_wait(level, j)
                    lo = j * chunk
                    regions.extend(batch.regions_of((lo + numpy.flatnonzero(status_col[lo:lo + chunk] == REGION_STATUS)).tolist()))
            elif mode == 2:
                regions.extend(RegionBatch(hd, hi, er, n_x, n_t, n_c, n_tc, k, numpy.flatnonzero(hi[:, 0] == REGION_STATUS)).regions())
            st, ms_wall = eng.solve_level_wait(level)
            if mode == 1 and st.n_region_retry:
                # some candidates were re-solved by the LDS-engine kernel after the stream (their slots are filled now): list the level again
                del regions[first:]
                regions.extend(batch.regions_of(numpy.flatnonzero(hi[:, 0] == REGION_STATUS).tolist()))
            if profile is not None:
                profile.append(_level_profile(level + 1, st, ms_wall))
            level += 1
        consumed = True
    finally:
        if not consumed:
            # whatever stopped this thread (an error of the loop seen through a wait, an interrupt): the loop is joined before the handle
            # is touched again, and what the loop itself failed with (MpcCapacityError: the solve is repeated without the overlap)
            # takes precedence
            eng.solve_wait()
    eng.solve_wait()
    # the base active set (= the equality rows) is tested last, like the reference (driver :142-146)
    if twin is not None:
        twin.level_wait()
        res = twin.base_result()
    else:
        res = eng.base_result()
    if res is None:
        base = numpy.arange(eng.n_eq, dtype=numpy.int32).reshape(1, -1)
        status, rd, ri, _, _ = eng.check_level(base, numpy.zeros((0, 2), dtype=numpy.uint64), False)
    else:
        status, rd, ri = res
    if profile is not None:
        profile.append({'depth': 0, 'k': eng.n_eq, 'candidates': 1, 'status': numpy.bincount(status, minlength=6).tolist(),
                        'regions': len(rd)})
    if len(rd):
        solution.add_region(unpack_region(rd[0], ri[0], n_x, n_t, n_c, n_tc))
    solution.is_complete = max_levels is None
    return solution


def solve_many(programs, device: int = 0, max_levels: Optional[int] = None, prune_lowdim: bool = True,
               profile: Optional[List[Dict]] = None) -> List[Solution]:
    """``_solve_many`` with the cycle collector held (region_batch.gc_paused): 64 sub-programs leave 10^5 region objects, and the full
    collections they trigger on the way walk all of them again and again (bench enumeration 140 -> 132 ms).  A single ``solve`` is NOT
    run that way: its 10^4 objects cost thirteen young-generation collections of 700 objects, and one collection of all of them when the
    collector is switched on again costs more (5.6 -> 5.8 ms per solve of the bench program, measured both ways)."""
    with gc_paused():
        return _solve_many(programs, device, max_levels, prune_lowdim, profile)


def _solve_many(programs, device: int = 0, max_levels: Optional[int] = None, prune_lowdim: bool = True,
                profile: Optional[List[Dict]] = None) -> List[Solution]:
    """Solves SEVERAL programs together, level by level: every stage of a level is one launch for all programs that still have a
    frontier (``Engine.level_run_batch`` -> mpc_level_run_batch, include/mpcombi.h; SURVEY.md 8(f)2, reference caller
    mp_solvers/mpmiqp_enumeration.py:41-50, which maps solve_mpqp over the sub-programs).  Each program's result is the one
    ``solve`` gives for it alone -- the same kernels' bodies on the same lists: the same regions, bit for bit except for the rare candidates
    that turn out optimal only after the theta stage (csrc/batch_level.hpp: built by another region kernel there, coefficients equal to ~1e-10).
    ``profile``: receives one dict per level (members, members that shared the launches, candidates, regions, wall time)."""
    from .._lib import Engine, MpcCapacityError
    programs = list(programs)
    if not programs:
        return []
    engs = [p.engine(device, closed=True) for p in programs]
    if len(set(id(e) for e in engs)) != len(engs):
        raise ValueError('solve_many: the programs must be distinct objects (one device handle each)')
    sols = [Solution(p, []) for p in programs]
    for sol in sols:
        sol.region_batches, sol.loose_regions = [], []      # the RegionBatch objects behind the lazy regions / regions that are plain objects
    depth_max = []
    for e in engs:
        d = max(e.n_x, e.n_t) - e.n_eq
        depth_max.append(d if max_levels is None else min(d, max_levels))
        e.set_timing(profile is not None)
        if not MANY_LOOP:      # (the library's loop does both itself)
            e.pruned_clear()
            e.frontier_root()
    # Device memory: a member's level holds buffers sized by its number of candidates (region records, children, two generations of
    # the dictionary cache: Engine.level_memory_gb).  Members whose next level does not fit the budget next to the others are PARKED
    # at their current level and resumed when the running ones have finished (and given their level buffers back: Engine.trim).
    in_flight = [None]      # the token of a level that has been started and not yet waited for
    budget_gb = 0.6 * float(os.environ.get('MPC_BATCH_BUDGET_GB', '160'))      # headroom: size classes round up, the pool keeps free blocks
    depth_of = [0] * len(engs)
    parked: List[int] = []

    def gen_of(i):
        return depth_of[i] + 1 != depth_max[i]

    def admit(candidates):
        """The members of ``candidates`` that run their next level together (in index order); the others go to ``parked``."""
        need = {i: engs[i].level_memory_gb(gen_of(i)) for i in candidates}
        take, used = [], 0.0
        for i in sorted(candidates, key=lambda j: (need[j], j)):
            if take and used + need[i] > budget_gb:
                parked.append(i)
            else:
                take.append(i)
                used += need[i]
        return sorted(take)
    def take_records(fetched):
        for i, (hd, hi, er, kk) in fetched:
            eng = engs[i]
            slots = numpy.flatnonzero(hi[:, 0] == REGION_STATUS)
            batch = RegionBatch(hd, hi, er, eng.n_x, eng.n_t, eng.n_c, eng.n_tc, kk, slots)
            sols[i].region_batches.append(batch)
            sols[i].critical_regions.extend(batch.regions())

    level_no = 0
    base_done = False
    first = [i for i in range(len(engs)) if depth_max[i] > 0]
    job = None
    try:
        if MANY_LOOP and first:
            # The loop over the levels of ALL members runs on a thread of the library (mpc_solve_many_*): shared launches, the level's record
            # copy, the frontier hand-overs, the next level -- the device does not wait for this interpreter between levels; this thread
            # builds the region objects of a level while later levels run.  The loop stops when the members' next level would not fit the
            # memory budget together (done == 2): the admission / parking loop below takes over from the frontiers it left.
            job = Engine.solve_many_start(engs, depth_max, keep_lowdim=not prune_lowdim, base=True)
            first = []
            n_lib = 0
            while True:
                done, lv = Engine.solve_many_level(job, n_lib)
                if lv is None:
                    break
                n_lib += 1
                members, stats, n_shared, ms_wall, records, is_base = lv
                take_records([(i, (hd, hi, er, kk)) for i, hd, hi, er, kk in records])
                if is_base:      # the closing level of the base active sets (reference driver :142-146), run by the library's loop as well
                    base_done = True
                    if profile is not None:
                        profile.append({'depth': 0, 'members': len(members), 'shared_launches': n_shared, 'candidates': len(members),
                                        'regions': int(sum(int(st.n_regions) for st in stats)), 'ms_wall': ms_wall, 'in_library': True})
                    continue
                first = []
                for i, st in zip(members, stats):
                    if gen_of(i) and st.n_children:
                        depth_of[i] += 1
                        first.append(i)
                level_no += 1
                if profile is not None:
                    profile.append({'depth': level_no, 'members': len(members), 'shared_launches': n_shared, 'parked': 0,
                                    'candidates': int(sum(int(st.n) for st in stats)), 'regions': int(sum(int(st.n_regions) for st in stats)),
                                    'ms_launches': float(stats[0].ms_total) if stats else 0.0, 'ms_wait': 0.0, 'ms_wall': ms_wall, 'in_library': True})
            jb, job = job, None
            Engine.solve_many_wait(jb)
            if done != 2:
                first = []      # every member ran its last level inside the library
        active = admit(first) if first else []
        gens = [gen_of(i) for i in active]
        token = Engine.level_batch_start([engs[i] for i in active], gens, keep_lowdim=not prune_lowdim) if active else None
        in_flight[0] = token
        while active:
            t0 = time.perf_counter()
            in_flight[0] = None             # the wait consumes the token, whatever it returns
            stats, n_shared = Engine.level_batch_wait(token)
            t_wait = time.perf_counter() - t0
            # the region records of this level: copies queued on each member's stream, nobody waits ...
            with_regions = [i for i, st in zip(active, stats) if st.n_regions]
            fetched = list(zip(with_regions, Engine.level_batch_fetch([engs[i] for i in with_regions])))
            nxt = [i for i, st, gen in zip(active, stats, gens) if gen and st.n_children]
            done = [i for i in active if i not in set(nxt)]
            Engine.frontier_advance_batch([engs[i] for i in nxt])
            for i in nxt:
                depth_of[i] += 1
            # admission by the size of everybody's NEXT level; skipped while the frontiers are far from the budget (128 KB per candidate
            # is more than a level of the register-resident kernels can hold per candidate: two dictionary records of 32 x 128 doubles)
            if parked or 128e-6 * sum(int(st.n_children) for st in stats) > 0.25 * budget_gb:
                candidates, parked[:] = nxt + parked, []
                nxt = admit(candidates)
            pressure = bool(parked)
            if pressure:
                for i in done:
                    engs[i].trim()      # (synchronises the member: its copies are complete) its level buffers serve the waiting members
            # ... the next level is started for all members that have a frontier and fit (its preparation completes the copies) ...
            gens_next = [gen_of(i) for i in nxt]
            if nxt:
                token = Engine.level_batch_start([engs[i] for i in nxt], gens_next, keep_lowdim=not prune_lowdim)
                in_flight[0] = token
            if not pressure:
                for i in done:
                    if any(i == j for j, _ in fetched):
                        engs[i].sync()  # a member that has just finished: nothing else completes its copies
            # a member that produced regions, has children and was PARKED by the admission above is in neither list: its copies were only
            # queued (mpc_level_regions_slots_nowait) and nothing of the next level completes them -- wait for them here, before the
            # integer heads are read (regions were silently lost when a parked member's records were large)
            started, finished = set(nxt), set(done)
            for i, _ in fetched:
                if i not in started and i not in finished:
                    engs[i].sync()
            # ... and the region objects of the finished level are built while the device works on it (the records of all members came by
            # one copy launch: complete once its event has passed -- it has, whenever a next level was started)
            if fetched:
                Engine.fetch_wait(engs[fetched[0][0]])
            take_records(fetched)
            level_no += 1
            if profile is not None:
                profile.append({'depth': level_no, 'members': len(active), 'shared_launches': n_shared, 'parked': len(parked),
                                'candidates': int(sum(int(st.n) for st in stats)), 'regions': int(sum(int(st.n_regions) for st in stats)),
                                'ms_launches': float(stats[0].ms_total) if stats else 0.0, 'ms_wait': t_wait * 1e3, 'ms_wall': (time.perf_counter() - t0) * 1e3})
            active, gens = nxt, gens_next
        for e in engs:
            e.sync()
        # the base active set of every program (driver :142-146), as one more shared level of one candidate each
        t0 = time.perf_counter()
        if not base_done:
            for e in engs:
                e.frontier_set(numpy.arange(e.n_eq, dtype=numpy.int32).reshape(1, -1))
                e.pruned_clear()
            stats, n_shared = Engine.level_run_batch(engs, [False] * len(engs), keep_lowdim=not prune_lowdim)
            for i, (e, st) in enumerate(zip(engs, stats)):
                if st.n_regions:
                    rd, ri, _ = e.level_regions()
                    sols[i].add_region(unpack_region(rd[0], ri[0], e.n_x, e.n_t, e.n_c, e.n_tc))
                    sols[i].loose_regions.append(sols[i].critical_regions[-1])
            if profile is not None:
                profile.append({'depth': 0, 'members': len(engs), 'shared_launches': n_shared, 'candidates': len(engs),
                                'regions': int(sum(int(st.n_regions) for st in stats)), 'ms_wall': (time.perf_counter() - t0) * 1e3})
    except MpcCapacityError:
        # a member that fell back to the overlapped single-program path ran out of reserved record slots (never observed): one by one
        if job is not None:
            try:
                Engine.solve_many_wait(job)      # (the loop has ended with that error: joined before the handles are used again)
            except Exception:
                pass
        return [solve(p, device=device, max_levels=max_levels, prune_lowdim=prune_lowdim) for p in programs]
    except BaseException:
        # a level may have been started and not waited for: let it finish (the handles must be idle before anybody touches them again)
        if job is not None:
            try:
                Engine.solve_many_wait(job)
            except Exception:
                pass
        if in_flight[0] is not None:
            try:
                Engine.level_batch_wait(in_flight[0])
            except Exception:
                pass
        raise
    for i, sol in enumerate(sols):
        sol.is_complete = max_levels is None
        if not _closing_rows_unused(engs[i], sol):      # (see solve: never observed) this program again, alone, on its own rows
            sols[i] = solve(programs[i], device=device, max_levels=max_levels, prune_lowdim=prune_lowdim)
    return sols
