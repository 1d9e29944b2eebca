"""Multi-GPU driver: one process per GPU under ``torch.distributed`` (backend "nccl" = RCCL over xGMI).

The reference's only parallel construct is ``pool.map(full_process, to_check)`` over the independent candidates of one
BFS level, followed by a merge in the parent (mp_solvers/mpqp_parrallel_combinatorial.py:110-135).  The candidates of a
level are independent, and a child set has exactly ONE generating parent (the set without its largest index), so the
tree can be split by subtrees without ever producing a candidate twice:

  * replicated phase   while a level is small (< shard_min x world candidates, shard_min = 128) every rank processes all of it.  The ranks
                       run the same deterministic kernels on the same data, so they stay identical without talking.
  * split              at the first level that is large enough every rank keeps the candidates rank, rank+world, ... of
                       the (still identical) frontier -- mpc_frontier_shard.  Their parents' dictionaries are in every
                       GPU's cache, so the split costs nothing.
  * sharded phase      each rank expands only its own candidates; children stay on the GPU that holds their parent's
                       dictionary.  The one thing the ranks must share is the pruned list (the reference's murder_list,
                       solver_utils.py:15-55): per level one all-gather of the newly pruned masks (KBs) and one of a small
                       statistics row.  No frontier, no dictionary and no region ever crosses xGMI inside the loop.
  * regions            every sharded level's regions are all-gathered asynchronously (padded: RCCL has no all-gather-v; RCCL's
                       stream) as soon as the level is done; they are copied to the host and turned into objects while the NEXT
                       level runs on the device (the engine's worker thread drives it), so every rank returns the complete
                       Solution and only the last level's regions are handled after the kernels.

The exchange is written against a small engine interface on torch tensors so that the same code runs on CUDA tensors
with RCCL (``HipLevelEngine``) and -- in the CPU tests -- on CPU tensors with gloo and an oracle-backed engine.
"""
from typing import Dict, List, Optional

import os

import numpy
import torch
import torch.distributed as dist

from .region_batch import RegionBatch, gc_paused
from .solution import Solution


class HipLevelEngine:
    """`_lib.Engine` behind the interface solve_distributed drives.  Device pointers are passed to the C ABI as plain
    integers.  The engine runs on its own HIP stream; ordering with torch / RCCL work is explicit: torch's stream is
    synchronised before a tensor is handed in, and every C-ABI call that touches caller memory has completed when it
    returns (include/mpcombi.h)."""

    def __init__(self, program, device_index: int):
        from . import _lib
        torch.cuda.set_device(device_index)
        self.device = torch.device('cuda', device_index)
        Q = getattr(program, 'Q', None)
        self.eng = _lib.Engine(program.A, program.b, program.F, program.c, program.H, Q, program.A_t, program.b_t,
                               len(program.equality_indices), device=device_index)
        e = self.eng
        self.n_x, self.n_t, self.n_c, self.n_tc, self.n_eq = e.n_x, e.n_t, e.n_c, e.n_tc, e.n_eq
        self._stats = None
        self._base_twin = None

    @property
    def capacity_error(self):
        from . import _lib
        return _lib.MpcCapacityError

    def set_region_overlap(self, on: bool):
        if self._base_twin is not None:      # a base-set check started beside the abandoned solve
            twin, self._base_twin = self._base_twin, None
            twin.level_wait()
        self.eng.set_region_overlap(on)

    def set_timing(self, on: bool):
        self.eng.set_timing(on)

    def clear_pruned(self):
        self.eng.pruned_clear()

    def root(self):
        self.eng.frontier_root()

    def frontier_size(self):
        return self.eng.frontier_info()

    def shard(self, rank: int, world: int):
        self.eng.frontier_shard(rank, world)

    def run(self, gen_children: bool) -> Dict:
        st = self.eng.level_run(gen_children)
        self._stats = st
        return self._as_dict(st)

    @staticmethod
    def _as_dict(st) -> Dict:
        return {'n': int(st.n), 'k': int(st.k), 'status': [int(v) for v in st.n_status], 'n_regions': int(st.n_regions),
                'n_children': int(st.n_children), 'n_pruned_new': int(st.n_pruned_new), 'lp_pivots': int(st.lp_pivots),
                'ms_verdict': float(st.ms_verdict), 'ms_region': float(st.ms_region), 'ms_children': float(st.ms_children),
                'ms_theta': float(st.ms_theta), 'ms_x': float(st.ms_x), 'ms_region2': float(st.ms_region2),
                'n_x_items': int(st.n_x_items), 'n_opt': int(st.n_opt), 'n_theta_items': int(st.n_theta_items), 'dict_read_bytes': int(st.dict_read_bytes),
                'dict_write_bytes': int(st.dict_write_bytes), 'ms_kkt': float(st.ms_kkt), 'ms_xq': float(st.ms_xq),
                'n_xq_items': int(st.n_xq_items), 'xq_pivots': int(st.xq_pivots),
                'n_xq_thread': int(st.n_xq_thread), 'ms_xq_thread': float(st.ms_xq_thread), 'xq_thread_beside_theta': bool(st.xq_thread_beside_theta),
                'n_x1': int(st.n_x1), 'ms_x1': float(st.ms_x1), 'ms_x_plan': float(st.ms_x_plan),
                'xq_record': [int(st.xq_record_ints), int(st.xq_record_rows), int(st.xq_record_cols)]}

    def _fresh(self, shape, dtype) -> torch.Tensor:
        """A tensor the ENGINE's stream may write.  torch's caching allocator only orders the reuse of a block against
        torch's own streams: a recycled block can still be the source of work queued earlier on torch's current stream
        (the padding copy that feeds the previous level's all-gather), which the engine's private stream knows nothing
        about.  Waiting for the current stream closes that window (blocks last used by RCCL's stream are held back by
        the allocator itself until that use has completed)."""
        t = torch.empty(shape, dtype=dtype, device=self.device)
        torch.cuda.current_stream(self.device).synchronize()
        return t

    def run_start(self, gen_children: bool, stream: bool = False):
        """``run`` on the engine's worker thread (mpc_level_start): returns at once, ``run_wait`` joins.  solve_distributed uses
        the time in between to bring the previous level's gathered regions to the host and build their objects.  ``stream``: this
        rank's own records are written by the region kernel straight into page-locked host arrays (MPC_LEVEL_STREAM), as in the
        single-GPU loop -- ``own_regions`` consumes them."""
        self.eng.level_start(gen_children, stream=stream)

    streams_own_records = True

    def own_regions(self):
        """Between ``run_start(..., stream=True)`` and ``run_wait``: this rank's region objects of the running level, built chunk by
        chunk while the region kernel writes them, and the host arrays behind them -- (regions, (head_d, head_i, erows, k)) or
        (None, None) when the level does not stream (then ``regions()`` after ``run_wait``)."""
        info = self.eng.level_stream_info()
        if info is None:
            return None, None
        hd, hi, er, chunk, n_chunks = info
        k = self.eng.frontier_info()[1]
        batch = RegionBatch(hd, hi, er, self.n_x, self.n_t, self.n_c, self.n_tc, k, ())
        status_col = hi[:, 0]
        out = []
        for j in range(n_chunks):
            self.eng.level_chunk_wait(j)
            lo = j * chunk
            out.extend(batch.regions_of((lo + numpy.flatnonzero(status_col[lo:lo + chunk] == REGION_STATUS)).tolist()))
        self._own_batch = batch
        return out, (hd, hi, er, k)

    def own_regions_after(self, regions, arrays):
        """After ``run_wait`` of a streamed level: (regions, rows of erows in use) -- candidates the LDS-engine kernel re-solved after
        the stream get their slots filled and the level is listed again."""
        st = self._stats
        rows = int(st.n_region_rows)
        if st.n_region_retry:
            hd, hi, er, _ = arrays
            rows = self.eng.level_stream_fixup(hd, hi, er)
            regions = self._own_batch.regions_of(numpy.flatnonzero(hi[:, 0] == REGION_STATUS).tolist())
        return regions, rows

    def run_wait(self) -> Dict:
        st = self.eng.level_wait()
        self._stats = st
        return self._as_dict(st)

    def pruned_new(self) -> torch.Tensor:
        m = int(self._stats.n_pruned_new)
        out = self._fresh((m, self.eng.mask_words), torch.int64)
        if m:
            self.eng.level_pruned_new_device(out.data_ptr(), m)
        return out

    def add_pruned(self, masks: torch.Tensor):
        if masks.numel():
            m = masks.contiguous()
            torch.cuda.current_stream(self.device).synchronize()
            self.eng.pruned_add_device(m.data_ptr(), m.shape[0])

    def frontier_tensor(self) -> torch.Tensor:
        """The current frontier [n, k] as an int32 tensor on the engine's device (re-shard step)."""
        return torch.from_numpy(self.eng.frontier_get()).to(self.device)

    def set_frontier_tensor(self, cands: torch.Tensor):
        """Replaces the frontier (the pruned list stays); the candidates start from the program's own vertex dictionary."""
        cands = cands.to(torch.int32).contiguous()
        if cands.is_cuda:
            torch.cuda.current_stream(cands.device).synchronize()
            self.eng.frontier_set_device(cands.data_ptr(), int(cands.shape[0]), int(cands.shape[1]))
        else:
            self.eng.frontier_set(cands.numpy())

    def advance(self):
        self.eng.frontier_advance()

    def regions(self):
        """(head_d, head_i, erows, k, slots) of the level just run (include/mpcombi.h, mpc_level_regions_slots)."""
        return self.eng.level_regions_slots()

    def regions_tensors(self):
        """(head_d, head_i, erows) of the level just run as tensors on this GPU, every slot (the gather filters the region
        slots afterwards): device-to-device copies of the kernel's output, no host round trip.  Levels with records
        that only exist in host form (regions re-solved by the LDS-engine kernel) go through the host."""
        ns, fd, fi, rows = self.eng.level_region_shapes()
        hd = self._fresh((ns, fd), torch.float64)
        hi = self._fresh((ns, fi), torch.int32)
        er = self._fresh((max(rows, 1), self.n_t + 1), torch.float64)
        got = self.eng.level_regions_device(hd.data_ptr(), hi.data_ptr(), er.data_ptr(), ns, rows) if ns else (0, 0)
        if got is None:
            h_hd, h_hi, h_er, _, _ = self.eng.level_regions_slots()
            return tuple(torch.from_numpy(numpy.ascontiguousarray(a)).to(self.device) for a in (h_hd, h_hi, h_er))
        return hd[:got[0]], hi[:got[0]], er[:got[1]]

    def base_start(self):
        """Starts the check of the base active set on a second handle of the program (Engine.twin, MPC_LEVEL_ONLY_BASE): its
        chain of small kernels runs under the large levels of the main handle; check_base() collects the result."""
        if self._base_twin is None:
            try:
                twin = self.eng.twin()
                twin.level_start(False, only_base=True)
                self._base_twin = twin
            except Exception:
                self._base_twin = None

    def check_base(self):
        """The base active set (the equality rows alone): (status histogram, region pieces or None)."""
        from .mp_solvers.mpqp_hip_combinatorial import unpack_regions
        res = None
        if self._base_twin is not None:
            twin, self._base_twin = self._base_twin, None
            twin.level_wait()
            res = twin.base_result()
        if res is not None:
            status, rd, ri = res
        else:
            base = numpy.arange(self.n_eq, dtype=numpy.int32).reshape(1, -1)
            status, rd, ri, _, _ = self.eng.check_level(base, numpy.zeros((0, 2), dtype=numpy.uint64), False)
        regs = unpack_regions(rd, ri, self.n_x, self.n_t, self.n_c, self.n_tc) if len(rd) else []
        return numpy.bincount(status, minlength=6).tolist(), regs

    def close(self):
        self.eng.close()


def allgather_rows(t: torch.Tensor, counts: List[int], group=None) -> List[torch.Tensor]:
    """All-gather of a 2-D tensor whose row count differs per rank: pad to the largest count, gather, trim.  Returns the
    per-rank pieces in rank order."""
    world = len(counts)
    if world == 1:
        return [t]
    mx = max(max(counts), 1)
    pad = torch.zeros((mx, t.shape[1]), dtype=t.dtype, device=t.device)
    if t.shape[0]:
        pad[:t.shape[0]] = t
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return [p[:c] for p, c in zip(parts, counts)]


def allgather_rows_start(t: torch.Tensor, counts: List[int], group=None):
    """allgather_rows without waiting: returns (per-rank pieces, work handle or None).  The pieces are valid after
    ``work.wait()``; RCCL runs the transfer on its own stream, so the next level's kernels overlap it."""
    world = len(counts)
    if world == 1:
        return [t], None
    mx = max(max(counts), 1)
    pad = t
    if t.shape[0] != mx:
        pad = torch.zeros((mx, t.shape[1]), dtype=t.dtype, device=t.device)
        if t.shape[0]:
            pad[:t.shape[0]] = t
    parts = [torch.empty_like(pad) for _ in range(world)]
    work = dist.all_gather(parts, pad.contiguous(), group=group, async_op=True)
    return [p[:c] for p, c in zip(parts, counts)], work


def _region_tensors(engine):
    """(head_d, head_i, erows) tensors of the level just run on the engine's device (all slots)."""
    if hasattr(engine, 'regions_tensors'):
        return engine.regions_tensors()
    hd, hi, er, _, _ = engine.regions()
    return tuple(torch.from_numpy(numpy.ascontiguousarray(a)).to(engine.device) for a in (hd, hi, er))


REGION_STATUS = 3


def to_host(t: torch.Tensor, non_blocking: bool = False) -> numpy.ndarray:
    """Device tensor -> numpy array in pooled page-locked memory (DMA copy); CPU tensors are returned as they are.  With
    ``non_blocking`` the copy is only queued on torch's current stream: the caller waits for an event recorded behind it."""
    if not t.is_cuda:
        return t.numpy()
    from . import _lib
    arr = _lib.pinned_empty(tuple(t.shape), _NP_DTYPE[t.dtype])
    torch.from_numpy(arr).copy_(t, non_blocking=non_blocking)
    return arr


_NP_DTYPE = {torch.float64: numpy.float64, torch.int32: numpy.int32, torch.int64: numpy.int64}


def allgather_table(row: List[int], device, group=None) -> List[List[int]]:
    """One small integer row per rank -> the table of all rows (rank order)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return [list(row)]      # one rank: nothing to exchange (no device tensor, no read-back)
    mine = torch.tensor(row, dtype=torch.int64, device=device)
    world = dist.get_world_size(group)
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine, group=group)
    return torch.stack(parts).cpu().tolist()


RESHARD = float(os.environ.get('MPC_RESHARD', '0') or 0)            # > 0: re-split the frontier evenly when the largest shard exceeds this multiple of the mean (e.g. 1.15)
RESHARD_MIN = int(os.environ.get('MPC_RESHARD_MIN', '0') or 0)        # ... and the next frontier has at least this many candidates


class DistributedLevelError(RuntimeError):
    """Raised on EVERY rank of the group, in the same level, when some rank's sharded level failed (device error, out of memory, any
    exception of its engine): the failing rank reports the fault in that level's statistics exchange instead of leaving the others
    waiting in it, so no rank hangs in a collective -- all of them leave the solve together (round 5).  ``failed_ranks`` lists the
    ranks that reported a fault; on those ranks ``__cause__`` is the engine's own exception."""

    def __init__(self, message, failed_ranks):
        super().__init__(message)
        self.failed_ranks = list(failed_ranks)


GATHER_TIMEOUT_S = 600.0     # an asynchronous region gather that has not completed after this long is a dead peer, not a slow one


def _wait(work):
    """Wait for an asynchronous collective.  gloo: bounded on the host (GATHER_TIMEOUT_S).  RCCL (backend 'nccl'): ``wait()`` WITHOUT a
    timeout only orders the current stream behind the transfer and returns at once -- which is what the overlap of the region gathers
    with the next level relies on; a user-supplied timeout would make torch's WorkNCCL::wait poll on the host until the collective has
    completed.  A dead peer then surfaces through the process group's own timeout / watchdog."""
    if work is None:
        return
    if _is_host_backend(work):
        try:
            import datetime
            work.wait(datetime.timedelta(seconds=GATHER_TIMEOUT_S))
            return
        except TypeError:
            pass
    work.wait()


def _is_host_backend(work) -> bool:
    """True for work objects of a CPU backend (gloo).  Decided from the default group's backend: the driver's collectives all run on
    the group it was given, and a mixed 'cpu:gloo,cuda:nccl' group carries device tensors over RCCL."""
    try:
        import torch.distributed as dist
        return 'nccl' not in str(dist.get_backend()).lower()
    except Exception:
        return True


class _RepeatWithoutOverlap(Exception):
    """Raised on EVERY rank of the group in the same level: some rank's level returned MPC_ERR_CAPACITY (more late optimal
    candidates than the overlapped region stage had reserved slots for; include/mpcombi.h, mpc_set_region_overlap)."""


def solve_distributed(engine, program=None, group=None, profile: Optional[List[Dict]] = None,
                      collect_regions: bool = True, max_levels: Optional[int] = None, shard_min: int = 128, force_shard: bool = False,
                      full_solution: str = 'all') -> Solution:
    """``full_solution``: 'all' -- every rank returns the complete Solution (the reference's parent process holds it,
    mpqp_parrallel_combinatorial.py:127-131; default); 'rank0' -- only rank 0 turns the other ranks' gathered records into host
    arrays and region objects, every other rank returns the regions of the replicated levels and of its own shards (its
    ``Solution`` is then partial: the eight-fold repetition of the host work is what the option removes).

    ``_solve_distributed``; when a level on some rank runs out of spare region slots, all ranks learn it from that level's
    statistics exchange (replicated levels: every rank hits it by itself, the kernels are deterministic) and repeat the solve
    together with the region stage behind the (x,theta) stage.  No candidate is ever demoted."""
    if full_solution not in ('all', 'rank0'):
        raise ValueError("full_solution must be 'all' or 'rank0'")
    with gc_paused():      # (round 6) as the single-GPU solve(): the region objects are created with the cycle collector held
        return _solve_distributed_guarded(engine, program, group, profile, collect_regions, max_levels, shard_min, force_shard, full_solution)


def _solve_distributed_guarded(engine, program, group, profile, collect_regions, max_levels, shard_min, force_shard, full_solution) -> Solution:
    try:
        return _solve_distributed(engine, program, group, profile, collect_regions, max_levels, shard_min, force_shard, full_solution)
    except _RepeatWithoutOverlap:
        # the abandoned solve left work in flight: host copies into pooled page-locked arrays, all-gathers on RCCL's stream.  Everything
        # is waited for before its buffers can be handed out again, the repeat runs with the region stage behind the (x,theta) stage, and
        # the faster form is restored afterwards
        if torch.cuda.is_available() and getattr(engine, 'device', None) is not None:
            torch.cuda.synchronize(engine.device)
        if hasattr(engine, 'set_region_overlap'):
            engine.set_region_overlap(False)
        if profile is not None:
            del profile[:]
        try:
            return _solve_distributed(engine, program, group, profile, collect_regions, max_levels, shard_min, force_shard, full_solution)
        finally:
            if hasattr(engine, 'set_region_overlap'):
                engine.set_region_overlap(True)


def _solve_distributed(engine, program=None, group=None, profile: Optional[List[Dict]] = None,
                       collect_regions: bool = True, max_levels: Optional[int] = None, shard_min: int = 128, force_shard: bool = False,
                       full_solution: str = 'all') -> Solution:
    """The level loop of the parallel combinatorial algorithm over the ranks of ``group`` (see the module docstring).
    Every rank returns the complete Solution.  Works without an initialised process group (world size 1)."""
    active = dist.is_available() and dist.is_initialized()
    rank = dist.get_rank(group) if active else 0
    world = dist.get_world_size(group) if active else 1
    e = engine.n_eq
    max_depth = max(engine.n_x, engine.n_t) - e
    if max_levels is not None:
        max_depth = min(max_depth, max_levels)
    solution = Solution(program, [])
    getattr(engine, 'set_timing', lambda on: None)(profile is not None)      # HIP-event records inside the levels only for a profile
    engine.clear_pruned()
    engine.root()
    sharded = False
    pending = None    # (k, three all-gathers in flight) of the most recent sharded level with regions
    tail_events = []  # completion of the large arrays' host copies (waited for once, before the solve returns)

    def finish(entry):
        """Waits for a level's gathers, copies every rank's piece to pinned host memory and lists its regions (rank order).  The
        rank's OWN shard is not copied back: its objects were built from the streamed records while the level ran."""
        kk, gathers, own = entry
        if gathers is None:      # one rank: nothing was gathered
            solution.critical_regions.extend(own or [])
            return
        if gathers[1][1] is not None:
            _wait(gathers[1][1])      # the integer heads' gather (the current stream waits for it, not the host)
        # Host copies are queued in the order the host needs them: the integer heads of every piece first (a region object is a lazy
        # view: building it needs the heads only), then the two large arrays.  The objects of a level are built while its large
        # arrays are still on their way; the solve waits for them once, before it returns.
        queued = []
        for src, (a, b, c) in enumerate(zip(gathers[0][0], gathers[1][0], gathers[2][0])):
            if src == rank and own is not None:
                queued.append(own)      # (kept in rank order: every rank returns the regions in the same order)
                continue
            if full_solution == 'rank0' and rank != 0 and src != rank:
                continue      # another rank's shard: only rank 0 builds its objects
            if a.shape[0]:
                hi = to_host(b, True)
                ev = None
                if b.is_cuda:
                    ev = torch.cuda.Event()
                    ev.record()
                queued.append([a, c, hi, ev])
        for j in (0, 2):
            if gathers[j][1] is not None:
                _wait(gathers[j][1])
        copied = [q for q in queued if q is not own]
        for q in copied:
            q[0], q[1] = to_host(q[0], True), to_host(q[1], True)
        if copied and copied[0][3] is not None:
            done = torch.cuda.Event()
            done.record()
            tail_events.append(done)
        for q in queued:
            if q is own:
                solution.critical_regions.extend(own)
                continue
            hd, er, hi, ev = q
            if ev is not None:
                ev.synchronize()
            slots = numpy.flatnonzero(hi[:, 0] == REGION_STATUS)
            solution.critical_regions.extend(RegionBatch(hd, hi, er, engine.n_x, engine.n_t, engine.n_c, engine.n_tc, kk, slots).regions())
    for depth in range(max_depth):
        gen_children = depth + 1 != max_depth
        n, k = engine.frontier_size()
        if not sharded and (world > 1 or force_shard) and n >= shard_min * world:   # force_shard: self-test with one rank
            engine.shard(rank, world)
            sharded = True
            if hasattr(engine, 'base_start'):
                engine.base_start()      # the base-set check runs beside the sharded levels (second handle)
        capacity = getattr(engine, 'capacity_error', ())      # exception type(s) of "out of spare region slots", if the engine has one
        own_regs = own_arrays = None
        own_rows = 0
        fault = None
        try:
            if sharded and collect_regions and getattr(engine, 'streams_own_records', False):
                # this rank's records are streamed to the host while the region kernel runs and become objects chunk by chunk (the
                # single-GPU loop's form); the previous level's pieces of the OTHER ranks are copied and listed first
                engine.run_start(gen_children, stream=True)
                if pending is not None:
                    finish(pending)
                    pending = None
                own_regs, own_arrays = engine.own_regions()
                st = engine.run_wait()
                if own_regs is not None:
                    own_regs, own_rows = engine.own_regions_after(own_regs, own_arrays)
            elif pending is not None and hasattr(engine, 'run_start'):
                # the previous sharded level's regions (gathered on RCCL's stream meanwhile) come to the host and become objects
                # while this level runs on the device
                engine.run_start(gen_children)
                finish(pending)
                pending = None
                st = engine.run_wait()
            else:
                if pending is not None:
                    finish(pending)
                    pending = None
                st = engine.run(gen_children)
        except capacity:
            if not sharded:
                raise _RepeatWithoutOverlap()      # replicated level: every rank is here
            st = {'n': 0, 'n_children': 0, 'n_pruned_new': 0, 'n_regions': 0, 'lp_pivots': -1, 'status': [0] * 6}
        except Exception as exc:      # noqa: BLE001 -- whatever the engine raised
            if not sharded or world == 1:
                raise      # replicated level / one rank: nobody is waiting in a collective of this level
            # a sharded level: the other ranks are on their way into this level's statistics exchange -- this rank joins it with a
            # fault marker, and every rank leaves the solve there (DistributedLevelError) instead of hanging
            fault = exc
            own_regs = own_arrays = None
            st = {'n': 0, 'n_children': 0, 'n_pruned_new': 0, 'n_regions': 0, 'lp_pivots': -2, 'status': [0] * 6}
        total = st
        tensors = None
        if sharded:
            if collect_regions and st['n_regions']:
                if own_regs is not None:
                    # streamed: the records are in host arrays; the other ranks get them from a device copy (queued on torch's stream,
                    # under the next level).  With one rank nothing is gathered and nothing is copied.
                    if world > 1:
                        hd_o, hi_o, er_o, _ = own_arrays
                        tensors = tuple(torch.from_numpy(a).to(engine.device, non_blocking=True) for a in (hd_o, hi_o, er_o[:max(own_rows, 0)]))
                else:
                    tensors = _region_tensors(engine)
                    if getattr(engine, 'streams_own_records', False):      # the level did not stream: this rank's objects from the fetched arrays
                        hd_o, hi_o, er_o, kk_o, slots_o = engine.regions()
                        own_regs = RegionBatch(hd_o, hi_o, er_o, engine.n_x, engine.n_t, engine.n_c, engine.n_tc, kk_o, slots_o).regions()
            n_slots, n_rows = (int(tensors[0].shape[0]), int(tensors[2].shape[0])) if tensors is not None else (0, 0)
            table = allgather_table([st['n'], st['n_children'], st['n_pruned_new'], st['n_regions'], st['lp_pivots'],
                                     n_slots, n_rows, *st['status']], engine.device, group)
            bad = [r_ for r_, row_ in enumerate(table) if row_[4] == -2]
            if bad:
                err = DistributedLevelError(f'level {depth + 1} (cardinality {int(k)}) failed on rank(s) {bad}; every rank leaves the solve', bad)
                if fault is not None:
                    raise err from fault
                raise err
            if any(r[4] < 0 for r in table):
                raise _RepeatWithoutOverlap()      # some rank's level failed for want of spare slots: all ranks repeat
            total = {'n': sum(r[0] for r in table), 'n_children': sum(r[1] for r in table),
                     'n_pruned_new': sum(r[2] for r in table), 'n_regions': sum(r[3] for r in table),
                     'lp_pivots': sum(r[4] for r in table),
                     'status': [sum(r[7 + j] for r in table) for j in range(len(st['status']))]}
            if gen_children and total['n_pruned_new'] and world > 1:
                # the other ranks' newly pruned sets join this rank's list (its own are added by advance())
                parts = allgather_rows(engine.pruned_new(), [r[2] for r in table], group)
                others = [p for r, p in enumerate(parts) if r != rank and p.shape[0]]
                if others:
                    engine.add_pruned(torch.cat(others, dim=0))
            if collect_regions and total['n_regions']:
                # this level's regions start their way to every rank now (device buffers, asynchronous all-gathers) and are
                # collected after the last level
                kk = int(k)
                fd = engine.n_x * engine.n_t + engine.n_x + kk * engine.n_t + kk
                fi = 8 + kk + engine.n_tc + kk + 2 * (engine.n_c - kk)
                if world == 1 and own_regs is not None:
                    pending = (kk, None, own_regs)
                else:
                    if tensors is None:
                        dev = engine.device
                        tensors = (torch.zeros((0, fd), dtype=torch.float64, device=dev), torch.zeros((0, fi), dtype=torch.int32, device=dev),
                                   torch.zeros((0, engine.n_t + 1), dtype=torch.float64, device=dev))
                    pending = (kk, [allgather_rows_start(tensors[0], [r[5] for r in table], group),
                                    allgather_rows_start(tensors[1], [r[5] for r in table], group),
                                    allgather_rows_start(tensors[2], [r[6] for r in table], group)], own_regs)
        elif collect_regions and st['n_regions']:
            hd, hi, er, kk, slots = engine.regions()
            solution.critical_regions.extend(RegionBatch(hd, hi, er, engine.n_x, engine.n_t, engine.n_c, engine.n_tc, kk, slots).regions())
        if profile is not None:
            profile.append({'depth': depth + 1, 'k': int(k), 'candidates': total['n'], 'status': total['status'],
                            'regions': total['n_regions'], 'children': total['n_children'],
                            'pruned_new': total['n_pruned_new'], 'lp_pivots': total['lp_pivots'],
                            'ms_verdict': st.get('ms_verdict', 0.0), 'ms_region': st.get('ms_region', 0.0),
                            'ms_children': st.get('ms_children', 0.0), 'local_candidates': st['n'], 'sharded': sharded,
                            **{key: st.get(key, 0) for key in ('ms_theta', 'ms_x', 'ms_region2', 'n_x_items', 'n_opt', 'n_theta_items',
                                                               'dict_read_bytes', 'dict_write_bytes', 'ms_kkt', 'ms_xq', 'n_xq_items',
                                                               'xq_pivots', 'xq_record', 'n_xq_thread', 'ms_xq_thread', 'xq_thread_beside_theta', 'n_x1', 'ms_x1', 'ms_x_plan')}})
        if not gen_children or total['n_children'] == 0:
            break
        engine.advance()
        if sharded and world > 1 and RESHARD > 0.0 and hasattr(engine, 'frontier_tensor'):
            # Re-shard (a switch, off by default): the split is made once and the subtrees grow unevenly -- measured
            # largest / mean shard 1.00-1.08 on configs 3 and 4, 1.19 on the late levels of a wide double integrator
            # (profiles/r05_ranks_*.json).  When the next frontier's largest share exceeds RESHARD x the mean, the ranks' children are
            # all-gathered (rank order: a deterministic list, no duplicates -- the subtrees are disjoint) and every rank takes an equal
            # slice.  Every rank holds every pruned set (they are exchanged each level), so any rank can take any child; the children start
            # from the program's own vertex dictionary on the next level (their parents' records stayed where they were made).
            counts = [int(r_[1]) for r_ in table]
            mean = sum(counts) / float(world)
            if mean > 0 and max(counts) > RESHARD * mean and sum(counts) >= RESHARD_MIN:
                parts = allgather_rows(engine.frontier_tensor(), counts, group)
                allc = torch.cat([p_ for p_ in parts if p_.shape[0]], dim=0) if any(p_.shape[0] for p_ in parts) else parts[0]
                n_all = int(allc.shape[0])
                lo, hi = rank * n_all // world, (rank + 1) * n_all // world
                engine.set_frontier_tensor(allc[lo:hi].contiguous())
                if profile is not None and profile:
                    profile[-1]['resharded'] = {'before': counts, 'after': [(r_ + 1) * n_all // world - r_ * n_all // world for r_ in range(world)]}
    if pending is not None:      # the last sharded level
        finish(pending)
    # the base active set, on every rank (one candidate; identical result everywhere)
    hist, regs = engine.check_base()
    if profile is not None:
        profile.append({'depth': 0, 'k': e, 'candidates': 1, 'status': hist, 'regions': len(regs)})
    if collect_regions:
        solution.critical_regions.extend(regs)
    for ev in tail_events:
        ev.synchronize()
    return solution
