"""Deterministic-solver plug of the host layer.

Mirrors ``Solver`` / ``SolverOutput`` of the reference (solver.py:56-246, solver_interface_utils.py:7-40) for the
problem classes the combinatorial path and its mixed-integer caller need: linear programs, and mixed-integer
linear programs with binary variables.  The only product backend is ``'hip'``: LPs are solved on the MI355X by the
one-wavefront LDS simplex behind ``mpc_lp_solve_batch`` (include/mpcombi.h); a MILP is the batch of LPs over all
fixations of its binaries, one launch (``solve_milp``).  There is no CPU fallback -- without the HIP library or a
GPU, ``solve_lp`` raises ``MpcError``.

``solve_lp`` returns ``None`` unless the LP has an optimal solution, exactly like the reference
(cvxopt_interface.py:20-23,186-198).
"""
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy

from . import _lib


@dataclass
class SolverOutput:
    """obj / sol / slack / active_set / dual of an LP solution (solver_interface_utils.py:7-40)."""
    obj: float
    sol: numpy.ndarray
    slack: Optional[numpy.ndarray] = None
    active_set: Optional[numpy.ndarray] = None
    dual: Optional[numpy.ndarray] = None


def _default_solvers() -> Dict[str, str]:
    return {'lp': 'hip', 'milp': 'hip'}


@dataclass
class Solver:
    """Chooses the backend per problem class (LP and binary MILP; both run as LP batches on the device)."""
    solvers: Dict[str, str] = field(default_factory=_default_solvers)
    device: int = 0

    supported_problems = ('lp', 'milp')
    supported_solvers = ('hip',)

    def __post_init__(self):
        for problem, backend in self.solvers.items():
            if problem not in self.supported_problems:
                raise RuntimeError(f'Problem {problem} is not supported! ppopt_amd supports {self.supported_problems}')
            if backend not in self.supported_solvers:
                raise RuntimeError(f'Solver {backend} is not supported! ppopt_amd supports {self.supported_solvers}')

    def solve_lp(self, c: Optional[numpy.ndarray], A: Optional[numpy.ndarray], b: Optional[numpy.ndarray],
                 equality_constraints: Optional[Sequence[int]] = None, verbose: bool = False,
                 get_duals: bool = True) -> Optional[SolverOutput]:
        """min c'x s.t. Ax <= b, rows ``equality_constraints`` as equalities, x free."""
        if A is None or A.shape[0] == 0 or A.shape[1] == 0:
            return None
        out = self.solve_lp_batch(c, A, b, [list(equality_constraints or [])])
        return out[0]

    def solve_lp_batch(self, c, A, b, equality_sets: List[Sequence[int]]) -> List[Optional[SolverOutput]]:
        """The same LP data with a different equality set per instance (what presolve needs,
        constraint_utilities.py:186-200): one device launch, one wavefront per instance."""
        A = numpy.ascontiguousarray(A, dtype=numpy.float64)
        m, n = A.shape
        bb = numpy.ascontiguousarray(b, dtype=numpy.float64).reshape(-1)
        flags = numpy.zeros((len(equality_sets), m), dtype=numpy.uint8)
        for i, eq in enumerate(equality_sets):
            flags[i, list(eq)] = 1
        cc = None if c is None else numpy.ascontiguousarray(c, dtype=numpy.float64).reshape(-1)
        status, x, obj, _ = _lib.lp_solve_batch(A, bb, cc, flags, device=self.device)
        res: List[Optional[SolverOutput]] = []
        for i in range(len(equality_sets)):
            if status[i] != _lib.LP_OPTIMAL:
                res.append(None)
                continue
            slack = bb - A @ x[i]
            active = numpy.nonzero(numpy.abs(slack) <= 1e-10)[0]
            res.append(SolverOutput(float(obj[i]), x[i].copy(), slack, active, None))
        return res

    def lp_feasible_many(self, requests) -> List[numpy.ndarray]:
        """Feasibility of MANY families of LPs in as few device batches as there are shapes.  ``requests``: a list of
        (A [m, n], b [m], equality sets) -- per family one matrix, one right-hand side and several equality sets (what the presolve's
        redundancy test poses per program, constraint_utilities.py:186-200).  Returns one bool array per request (entry j: the LP with
        equality set j has a solution).  Only the status comes back from the device (no solution vectors)."""
        out: List[Optional[numpy.ndarray]] = [None] * len(requests)
        groups: Dict[tuple, List[int]] = {}
        for r, (A, b, eqs) in enumerate(requests):
            if len(eqs) == 0:
                out[r] = numpy.zeros(0, dtype=bool)
            else:
                groups.setdefault(tuple(numpy.shape(A)), []).append(r)
        for (m, n), members in groups.items():
            # every LP of a device batch carries its own copy of A (the families differ in A): the batches are cut by the same host-byte
            # budget as the fixation batches below, so many binaries / large sub-programs never make one allocation of gigabytes
            per_lp = 8 * (m * n + m) + m
            budget = max(1, int(self.MILP_BATCH_BYTES // per_lp))
            # (family, first equality set, number of sets) pieces, packed into batches of at most `budget` LPs
            batches: List[List[tuple]] = [[]]
            room = budget
            for r in members:
                cnt, first = len(requests[r][2]), 0
                out[r] = numpy.empty(cnt, dtype=bool)
                while first < cnt:
                    take = min(cnt - first, room)
                    batches[-1].append((r, first, take))
                    first += take
                    room -= take
                    if room == 0:
                        batches.append([])
                        room = budget
            for pieces in batches:
                total = sum(cnt for _, _, cnt in pieces)
                if total == 0:
                    continue
                A3 = numpy.empty((total, m, n))
                b2 = numpy.empty((total, m))
                flags = numpy.zeros((total, m), dtype=numpy.uint8)
                pos = 0
                for r, first, cnt in pieces:
                    A, b, eqs = requests[r]
                    A3[pos:pos + cnt] = A
                    b2[pos:pos + cnt] = numpy.asarray(b, dtype=numpy.float64).reshape(-1)
                    for j in range(cnt):
                        flags[pos + j, list(eqs[first + j])] = 1
                    pos += cnt
                status, _, _, _ = _lib.lp_solve_batch(A3, b2, None, flags, device=self.device, want_x=False)
                pos = 0
                for r, first, cnt in pieces:
                    out[r][first:first + cnt] = status[pos:pos + cnt] == _lib.LP_OPTIMAL
                    pos += cnt
        return out

    # ---- binary MILPs as LP batches (solver.py:248-282; the reference hands these to Gurobi) -------------------------
    MAX_BINARIES = 20
    # host bytes one device batch of fixation LPs may take (equality flags + the solution block when it is asked for)
    MILP_BATCH_BYTES = 256 << 20
    # up to this many binaries every leaf is posed directly; beyond, prefixes are pruned through their LP relaxation
    MILP_DIRECT_BINARIES = 10

    @staticmethod
    def binary_fixations(n_bin: int, first: int = 0, count: Optional[int] = None) -> numpy.ndarray:
        """Fixations first .. first+count-1 of the 2^n_bin (all by default), [count, n_bin], first binary most
        significant: row order == the order in which MITree.get_full_leafs lists leaves (0-branch before 1-branch at
        every depth, mitree.py:84-102)."""
        count = (1 << n_bin) - first if count is None else count
        idx = numpy.arange(first, first + count, dtype=numpy.int64)[:, None]
        shifts = numpy.arange(n_bin - 1, -1, -1, dtype=numpy.int64)[None, :]
        return ((idx >> shifts) & 1).astype(numpy.int8)

    def _milp_blocks(self, A, b, bin_vars):
        """The shared LP data of every fixation: rows  y_j <= 1  and  -y_j <= 0  appended (flagging the first as an
        equality fixes y_j = 1, the second y_j = 0, neither leaves y_j relaxed to [0, 1])."""
        A = numpy.ascontiguousarray(A, dtype=numpy.float64)
        m, n = A.shape
        bins = list(bin_vars or [])
        nb = len(bins)
        if nb > self.MAX_BINARIES:
            raise ValueError(f'{nb} binary variables: the enumeration backend is limited to {self.MAX_BINARIES}')
        bb = numpy.ascontiguousarray(b, dtype=numpy.float64).reshape(-1)
        up = numpy.zeros((nb, n))
        up[numpy.arange(nb), bins] = 1.0
        return numpy.vstack([A, up, -up]), numpy.concatenate([bb, numpy.ones(nb), numpy.zeros(nb)]), m, n, bins

    def _fixation_lps(self, A_aug, b_aug, m, nb, set_flags, set_idx, fix, c=None, want_x=False):
        """LPs ``(equality set set_idx[i], fixation fix[i])`` for i in range(len(fix)); fix entries 0 / 1 / -1 (relaxed).
        Posed in device batches of at most MILP_BATCH_BYTES host bytes.  Returns (status, obj, x or None)."""
        n_items, n = len(fix), A_aug.shape[1]
        status = numpy.empty(n_items, dtype=numpy.int32)
        obj = numpy.empty(n_items)
        x = numpy.empty((n_items, n)) if want_x else None
        per_item = (m + 2 * nb) + 16 + (8 * n if want_x else 0)
        step = max(1, int(self.MILP_BATCH_BYTES // per_item))
        for lo in range(0, n_items, step):
            hi = min(n_items, lo + step)
            flags = numpy.zeros((hi - lo, m + 2 * nb), dtype=numpy.uint8)
            flags[:, :m] = set_flags[set_idx[lo:hi]]
            flags[:, m:m + nb] = fix[lo:hi] == 1
            flags[:, m + nb:] = fix[lo:hi] == 0
            st, xx, ob, _ = _lib.lp_solve_batch(A_aug, b_aug, c, flags, device=self.device, want_x=want_x)
            status[lo:hi], obj[lo:hi] = st, ob
            if want_x:
                x[lo:hi] = xx
        return status, obj, x

    @staticmethod
    def _set_flags(equality_sets, m):
        flags = numpy.zeros((len(equality_sets), m), dtype=numpy.uint8)
        for i, eq in enumerate(equality_sets):
            flags[i, list(eq)] = 1
        return flags

    def milp_leaf_feasibility(self, A, b, equality_constraints: Sequence[int], bin_vars: Sequence[int]) -> numpy.ndarray:
        """bool[2^n_bin]: does the LP of that full fixation (row order of ``binary_fixations``) have a solution.

        Few binaries: every leaf is one LP of a single batch.  Many: the tree is walked level by level from depth
        MILP_DIRECT_BINARIES -- a prefix whose LP relaxation (remaining binaries in [0, 1]) is infeasible has no feasible
        leaf and is dropped with its whole subtree, so the work follows the number of feasible nodes (like the
        reference's tree walk, mitree.py:22-64), not 2^n_bin, and every batch respects MILP_BATCH_BYTES."""
        A_aug, b_aug, m, n, bins = self._milp_blocks(A, b, bin_vars)
        nb = len(bins)
        sf = self._set_flags([list(equality_constraints or [])], m)
        table = numpy.zeros(1 << nb, dtype=bool)
        depth = min(nb, self.MILP_DIRECT_BINARIES)
        prefixes = numpy.arange(1 << depth, dtype=numpy.int64)          # values of the first `depth` binaries
        while True:
            fix = numpy.full((len(prefixes), nb), -1, dtype=numpy.int8)
            if depth:
                fix[:, :depth] = (prefixes[:, None] >> numpy.arange(depth - 1, -1, -1, dtype=numpy.int64)[None, :]) & 1
            st, _, _ = self._fixation_lps(A_aug, b_aug, m, nb, sf, numpy.zeros(len(prefixes), dtype=numpy.int64), fix)
            prefixes = prefixes[st == _lib.LP_OPTIMAL]
            if depth == nb or len(prefixes) == 0:
                break
            prefixes = numpy.stack([prefixes << 1, (prefixes << 1) | 1], axis=1).reshape(-1)
            depth += 1
        table[prefixes] = True
        return table

    def milp_any_feasible(self, A, b, equality_sets: List[Sequence[int]], bin_vars: Sequence[int],
                          leaves: Optional[numpy.ndarray] = None) -> numpy.ndarray:
        """bool[n_sets]: is the mixed-integer system feasible with that set of rows as equalities -- i.e. does some
        fixation among ``leaves`` (indices into ``binary_fixations``; all by default) give a feasible LP.  The (set, leaf)
        pairs go to the device in batches bounded by MILP_BATCH_BYTES, leaf-major, and a set leaves the work list as
        soon as one of its LPs is feasible."""
        A_aug, b_aug, m, n, bins = self._milp_blocks(A, b, bin_vars)
        nb, n_sets = len(bins), len(equality_sets)
        sf = self._set_flags(equality_sets, m)
        leaves = numpy.arange(1 << nb, dtype=numpy.int64) if leaves is None else numpy.asarray(leaves, dtype=numpy.int64)
        found = numpy.zeros(n_sets, dtype=bool)
        per_item = (m + 2 * nb) + 16
        budget_items = max(1, int(self.MILP_BATCH_BYTES // per_item))
        shifts = numpy.arange(nb - 1, -1, -1, dtype=numpy.int64)[None, :]
        pos = 0
        while pos < len(leaves) and not found.all():
            open_sets = numpy.flatnonzero(~found)
            n_leaves = max(1, min(len(leaves) - pos, budget_items // len(open_sets)))
            lv = leaves[pos:pos + n_leaves]
            pos += n_leaves
            fix_l = ((lv[:, None] >> shifts) & 1).astype(numpy.int8) if nb else numpy.zeros((len(lv), 0), dtype=numpy.int8)
            set_idx = numpy.repeat(open_sets, len(lv))
            fix = numpy.tile(fix_l, (len(open_sets), 1))
            st, _, _ = self._fixation_lps(A_aug, b_aug, m, nb, sf, set_idx, fix)
            ok = (st == _lib.LP_OPTIMAL).reshape(len(open_sets), len(lv)).any(axis=1)
            found[open_sets[ok]] = True
        return found

    def solve_milp_batch(self, c, A, b, equality_sets: List[Sequence[int]], bin_vars: Sequence[int], want_x: bool = True):
        """For every equality set, the LP of every binary fixation: returns (status, x, obj) shaped
        [n_sets, 2^n_bin(, n)] -- status == 0 where that fixation's LP has an optimal solution; ``x`` is None with
        ``want_x=False``.  Dense in (set, fixation): meant for few binaries; ``milp_leaf_feasibility``,
        ``milp_any_feasible`` and ``solve_milp`` answer the questions of the mixed-integer programs within a bounded
        footprint.  The device batches themselves respect MILP_BATCH_BYTES."""
        A_aug, b_aug, m, n, bins = self._milp_blocks(A, b, bin_vars)
        nb, n_sets = len(bins), len(equality_sets)
        n_fix = 1 << nb
        dense = n_sets * n_fix * (8 * n if want_x else 12)
        if dense > 16 * self.MILP_BATCH_BYTES:
            raise MemoryError(f'solve_milp_batch: {n_sets} sets x {n_fix} fixations is {dense >> 20} MiB of results; '
                              'use milp_leaf_feasibility / milp_any_feasible / solve_milp')
        fix = self.binary_fixations(nb)
        sf = self._set_flags(equality_sets, m)
        cc = None if c is None else numpy.ascontiguousarray(c, dtype=numpy.float64).reshape(-1)
        status, obj, x = self._fixation_lps(A_aug, b_aug, m, nb, sf, numpy.repeat(numpy.arange(n_sets), n_fix),
                                            numpy.tile(fix, (n_sets, 1)), cc, want_x)
        if want_x:
            x = x.reshape(n_sets, n_fix, n)
            x[:, :, bins] = fix[None]
        return status.reshape(n_sets, n_fix), x, obj.reshape(n_sets, n_fix)

    def solve_milp(self, c: Optional[numpy.ndarray], A: Optional[numpy.ndarray], b: Optional[numpy.ndarray],
                   equality_constraints: Optional[Sequence[int]] = None, bin_vars: Optional[Sequence[int]] = None,
                   verbose: bool = False, get_duals: bool = True) -> Optional[SolverOutput]:
        """min c'[x,y] s.t. A[x,y] <= b, rows ``equality_constraints`` as equalities, y binary, x free.  ``None``
        unless some fixation has an optimal LP; the best objective wins, the first fixation on ties.  Only the
        feasible leaves (``milp_leaf_feasibility``) are posed with the objective, in bounded batches; one more LP
        returns the minimiser of the winner."""
        if A is None or A.shape[0] == 0 or A.shape[1] == 0:
            return None
        eq = list(equality_constraints or [])
        leaves = numpy.flatnonzero(self.milp_leaf_feasibility(A, b, eq, bin_vars))
        if len(leaves) == 0:
            return None
        A_aug, b_aug, m, n, bins = self._milp_blocks(A, b, bin_vars)
        nb = len(bins)
        sf = self._set_flags([eq], m)
        cc = None if c is None else numpy.ascontiguousarray(c, dtype=numpy.float64).reshape(-1)
        shifts = numpy.arange(nb - 1, -1, -1, dtype=numpy.int64)[None, :]
        fix = ((leaves[:, None] >> shifts) & 1).astype(numpy.int8)
        status, obj, _ = self._fixation_lps(A_aug, b_aug, m, nb, sf, numpy.zeros(len(leaves), dtype=numpy.int64), fix, cc)
        ok = status == _lib.LP_OPTIMAL
        if not ok.any():
            return None      # every feasible leaf is unbounded below
        best = int(numpy.argmin(numpy.where(ok, obj, numpy.inf)))
        _, _, x = self._fixation_lps(A_aug, b_aug, m, nb, sf, numpy.zeros(1, dtype=numpy.int64), fix[best:best + 1], cc, True)
        sol = x[0].copy()
        sol[bins] = fix[best]
        bb = numpy.asarray(b, dtype=numpy.float64).reshape(-1)
        slack = bb - numpy.asarray(A, dtype=numpy.float64) @ sol
        return SolverOutput(float(obj[best]), sol, slack, numpy.nonzero(numpy.abs(slack) <= 1e-10)[0], None)


class LPCoalescer:
    """Turns the LP calls of several host threads into shared device batches.

    The mixed-integer enumeration builds one continuous sub-program per binary fixation; each construction presolves (a Chebyshev
    LP, a feasibility LP, one redundancy LP per row -- mplp_program.py) through ``Solver.solve_lp_batch``, three small device
    batches per sub-program, each paying the fixed cost of a launch and a synchronisation.  With the constructions running in
    ``n_workers`` threads against ``solver()``, a call parks its LPs here; when every live worker is parked the last one to arrive
    poses ALL parked LPs as one batch per shape (per-instance matrices, mpc_lp_solve_batch) and hands the results back.  The kernel
    and its arithmetic are those of the separate calls, instance for instance."""

    def __init__(self, base: Solver, n_workers: int):
        import threading
        self.base, self.live = base, n_workers
        self.cv = threading.Condition()
        self.queue: List[dict] = []
        self.n_flushes = self.n_calls = 0

    def solver(self) -> 'Solver':
        outer = self

        class _Parked(Solver):
            def solve_lp_batch(self, c, A, b, equality_sets):      # solve_lp goes through here too
                return outer.submit(c, A, b, equality_sets)
        return _Parked(solvers=dict(self.base.solvers), device=self.base.device)

    def worker_done(self):
        with self.cv:
            self.live -= 1
            if self.queue and len(self.queue) >= self.live:
                self._flush()

    def submit(self, c, A, b, equality_sets):
        A = numpy.ascontiguousarray(A, dtype=numpy.float64)
        item = {'c': None if c is None else numpy.ascontiguousarray(c, dtype=numpy.float64).reshape(-1), 'A': A,
                'b': numpy.ascontiguousarray(b, dtype=numpy.float64).reshape(-1), 'eq': [list(e) for e in equality_sets], 'res': None, 'err': None}
        with self.cv:
            self.n_calls += 1
            self.queue.append(item)
            if len(self.queue) >= self.live:
                self._flush()
            while item['res'] is None and item['err'] is None:
                self.cv.wait()
        if item['err'] is not None:
            raise item['err']
        return item['res']

    def _flush(self):       # the lock is held; every other live worker is waiting
        items, self.queue = self.queue, []
        self.n_flushes += 1
        try:
            groups: Dict[tuple, List[dict]] = {}
            for it in items:
                groups.setdefault((it['A'].shape, it['c'] is None), []).append(it)
            for (shape, no_c), grp in groups.items():
                m, n = shape
                counts = [len(it['eq']) for it in grp]
                total = sum(counts)
                if total == 0:
                    for it in grp:
                        it['res'] = []
                    continue
                A = numpy.empty((total, m, n))
                bb = numpy.empty((total, m))
                cc = None if no_c else numpy.empty((total, n))
                flags = numpy.zeros((total, m), dtype=numpy.uint8)
                pos = 0
                for it, cnt in zip(grp, counts):
                    A[pos:pos + cnt], bb[pos:pos + cnt] = it['A'], it['b']
                    if cc is not None:
                        cc[pos:pos + cnt] = it['c']
                    for j, eq in enumerate(it['eq']):
                        flags[pos + j, eq] = 1
                    pos += cnt
                status, x, obj, _ = _lib.lp_solve_batch(A, bb, cc, flags, device=self.base.device)
                pos = 0
                for it, cnt in zip(grp, counts):
                    res: List[Optional[SolverOutput]] = []
                    for i in range(pos, pos + cnt):
                        if status[i] != _lib.LP_OPTIMAL:
                            res.append(None)
                            continue
                        slack = it['b'] - it['A'] @ x[i]
                        res.append(SolverOutput(float(obj[i]), x[i].copy(), slack, numpy.nonzero(numpy.abs(slack) <= 1e-10)[0], None))
                    it['res'] = res
                    pos += cnt
        except Exception as ex:      # every parked caller sees the failure
            for it in items:
                if it['res'] is None:
                    it['err'] = ex
        self.cv.notify_all()
