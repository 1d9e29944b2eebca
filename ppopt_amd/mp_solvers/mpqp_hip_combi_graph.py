"""The joint combinatorial / connected-graph algorithm (Arnström et al., arXiv:2404.05511) on the MI355X.

Reference: mp_solvers/mpqp_combi_graph.py:68-145.  The algorithm keeps a set ``S`` of active sets to visit and a set ``E``
of everything ever put into ``S``.  A visited active set is

    rank deficient            -> its subsets (one row removed) are explored           (:119-121)
    full rank, region empty   -> nothing                                              (:123)
    region non-empty          -> region built (kept when full dimensional), subsets AND supersets (one row added)
                                 are explored                                         (:123-143)

so only active sets within one row of a non-empty critical region are ever examined: a few tens of candidates per region
instead of the combinatorial tree, which makes complete solutions possible where ``max(n_x, n_theta)`` levels of the
combinatorial algorithm are out of reach (config 3: 20 levels over 80 rows).

Here ``S`` is processed a whole wave at a time: the wave is split by cardinality, every group is one frontier of the
device engine (rank test, KKT solve, "is the region non-empty" LP, region kernel: MPC_LEVEL_GRAPH -- the same kernels as
the combinatorial path without its (x,theta) stage).  ``S`` and ``E`` live on the device as sorted arrays of 128/256-bit
masks: the neighbours of a wave are emitted by a kernel, radix-sorted, deduplicated and checked against ``E`` by binary
search (csrc/graph.hpp, mpc_graph_*); ``MPC_GRAPH_HOST=1`` keeps the same bookkeeping in numpy arrays on the host
instead (cross-check).  The reference pops ``S`` in arbitrary (hash) order; the set of examined active sets and the
resulting regions do not depend on the order.

Seeds: the reference starts from ``program.sample_theta_space(1)`` -- one QP solve at a random parameter point.  Here
the traversal is seeded with the active sets of the regions the first levels of the combinatorial algorithm find (level
by level until one has a region), which is deterministic, or with ``seeds`` (e.g. from ``MPQP_Program.sample_theta_space``,
which runs on the device through ``mpc_qp_solve_batch``).
"""
import os
from typing import Dict, Iterable, List, Optional

import numpy

from ..region_batch import RegionBatch
from ..solution import Solution

INFEASIBLE, FEASIBLE, OPTIMAL_NO_REGION, REGION = 0, 1, 2, 3


def _bit_table(n_c: int, words: int) -> numpy.ndarray:
    """[n_c, words] uint64: the mask of the single row i."""
    t = numpy.zeros((n_c, words), dtype=numpy.uint64)
    i = numpy.arange(n_c)
    t[i, i >> 6] = numpy.uint64(1) << (i & 63).astype(numpy.uint64)
    return t


def _sets_to_masks(sets: Iterable[Iterable[int]], words: int) -> numpy.ndarray:
    sets = [list(s) for s in sets]
    m = numpy.zeros((len(sets), words), dtype=numpy.uint64)
    for r, s in enumerate(sets):
        for v in s:
            m[r, int(v) >> 6] |= numpy.uint64(1) << numpy.uint64(int(v) & 63)
    return m


def _masks_to_index_rows(masks: numpy.ndarray, k: int, n_c: int) -> numpy.ndarray:
    """Masks that all have k bits set -> [n, k] int32 sorted index lists."""
    n = len(masks)
    if k == 0:
        return numpy.zeros((n, 0), dtype=numpy.int32)
    bits = numpy.unpackbits(numpy.ascontiguousarray(masks).view(numpy.uint8), axis=1, bitorder='little')[:, :n_c]
    return numpy.nonzero(bits)[1].reshape(n, k).astype(numpy.int32)


def _popcount(masks: numpy.ndarray) -> numpy.ndarray:
    return numpy.unpackbits(numpy.ascontiguousarray(masks).view(numpy.uint8), axis=1).sum(axis=1)


class _SetBook:
    """The set ``E`` of every active set that has ever been queued, as masks.  Membership goes through a 64-bit hash of the
    mask words (sorted arrays, binary search); every hash match is confirmed on the full mask, and a true collision --
    two different sets with one hash -- switches the book to exact structured keys for good."""

    _C1, _C2 = numpy.uint64(0x9E3779B97F4A7C15), numpy.uint64(0xC2B2AE3D27D4EB4F)

    def __init__(self, words: int):
        self.words = words
        self.exact = False
        self.h = numpy.zeros(0, dtype=numpy.uint64)
        self.m = numpy.zeros((0, words), dtype=numpy.uint64)

    def _hash(self, masks: numpy.ndarray) -> numpy.ndarray:
        with numpy.errstate(over='ignore'):
            h = masks[:, 0] * self._C1
            for w in range(1, self.words):
                h = (h ^ (h >> numpy.uint64(29)) ^ masks[:, w]) * self._C2
            return h ^ (h >> numpy.uint64(32))

    def _structured(self, masks):
        return numpy.ascontiguousarray(masks).view([(f'w{j}', numpy.uint64) for j in range(self.words)]).reshape(-1)

    def add(self, masks: numpy.ndarray) -> numpy.ndarray:
        """Adds the sets; returns those that were not in the book yet, each once."""
        if len(masks) == 0:
            return masks
        if not self.exact:
            h = self._hash(masks)
            order = numpy.argsort(h, kind='stable')
            hs, ms = h[order], masks[order]
            dup = numpy.concatenate([[False], hs[1:] == hs[:-1]])
            ok = not dup.any() or bool((ms[dup] == ms[numpy.flatnonzero(dup) - 1]).all())
            hs, ms = hs[~dup], ms[~dup]
            pos = numpy.searchsorted(self.h, hs)
            hit = pos < len(self.h)
            hit[hit] = self.h[pos[hit]] == hs[hit]
            ok = ok and bool((self.m[pos[hit]] == ms[hit]).all())
            if ok:
                new_h, new_m = hs[~hit], ms[~hit]
                allh = numpy.concatenate([self.h, new_h])
                o2 = numpy.argsort(allh, kind='stable')
                self.h, self.m = allh[o2], numpy.concatenate([self.m, new_m])[o2]
                return new_m
            self.exact = True                      # a genuine hash collision: exact keys from now on
            self.k = numpy.unique(self._structured(self.m))
        keys = numpy.unique(self._structured(masks))
        keys = keys[~numpy.isin(keys, self.k, assume_unique=True)]
        self.k = numpy.union1d(self.k, keys)
        return keys.view(numpy.uint64).reshape(-1, self.words)


def _neighbours(masks: numpy.ndarray, shrink: numpy.ndarray, grow: numpy.ndarray, bit: numpy.ndarray, eq_mask: numpy.ndarray,
                grow_rows: Optional[numpy.ndarray] = None) -> numpy.ndarray:
    """Subsets (one non-equality row removed) of the sets flagged ``shrink`` and supersets (one row added) of those flagged
    ``grow`` -- every missing row, or only the rows ``grow_rows[j]`` ([n, n_c] bool) allows."""
    words = masks.shape[1]
    removable = ~(eq_mask[None, :] & bit).any(axis=1)                     # [n_c]: rows that are not program equalities
    parts = []
    step = max(1, (64 << 20) // (8 * words * len(bit)))                    # rows per block: about 64 MiB of intermediates
    for lo in range(0, len(masks), step):
        m = masks[lo:lo + step]
        member = (m[:, None, :] & bit[None, :, :]).any(axis=2)            # [n, n_c]: row i belongs to the set
        r, i = numpy.nonzero(member & removable[None, :] & shrink[lo:lo + step, None])
        if len(r):
            parts.append(m[r] & ~bit[i])
        allowed = ~member & grow[lo:lo + step, None]
        if grow_rows is not None:
            allowed &= grow_rows[lo:lo + step]
        r, i = numpy.nonzero(allowed)
        if len(r):
            parts.append(m[r] | bit[i])
    if not parts:
        return numpy.zeros((0, words), dtype=numpy.uint64)
    return numpy.concatenate(parts, axis=0)


def _traverse(program, device, seeds, profile, max_candidates, graph_question: bool) -> Solution:
    from ..mpqp_program import MPQP_Program
    if not isinstance(program, MPQP_Program):
        # mplp_program.py's optimal_control_law (pseudo-inverse for k != n_x) gives these traversals different semantics
        raise NotImplementedError('the connected-graph traversals are implemented for mpQPs')
    eng = program.engine(device)
    n_x, n_t, n_c, n_tc, n_eq, words = eng.n_x, eng.n_t, eng.n_c, eng.n_tc, eng.n_eq, eng.mask_words
    bit = _bit_table(n_c, words)
    eq_mask = _sets_to_masks([range(n_eq)], words)[0]
    solution = Solution(program, [])
    if seeds is None:
        seeds = _seed_active_sets(program, eng)
    if not seeds:
        return solution
    if os.environ.get('MPC_GRAPH_HOST', '0') != '1':
        return _traverse_device(eng, solution, _sets_to_masks(seeds, words), profile, max_candidates, graph_question)
    # ---- the same traversal with the bookkeeping in host arrays (A/B and cross-check of the device bookkeeping) ----------
    book = _SetBook(words)
    todo = book.add(_sets_to_masks(seeds, words))
    examined = 0
    while len(todo):
        masks = todo
        card = _popcount(masks)
        status = numpy.empty(len(masks), dtype=numpy.uint8)
        facet_rows = None if graph_question else numpy.zeros((len(masks), n_c), dtype=bool)
        n_regions = 0
        for k in numpy.unique(card).tolist():
            sel = numpy.flatnonzero(card == k)
            if k > min(n_x, n_c):           # more rows than variables: rank deficient by counting (is_full_rank)
                status[sel] = INFEASIBLE
                continue
            eng.frontier_set(_masks_to_index_rows(masks[sel], k, n_c))
            st = eng.level_run(False, graph=True)    # neither traversal needs the (x,theta) feasibility LP (see solve_graph)
            status[sel] = eng.level_status()
            if st.n_regions:
                hd, hi, er, kk, slots = eng.level_regions_slots()
                batch = RegionBatch(hd, hi, er, n_x, n_t, n_c, n_tc, kk, slots)
                solution.critical_regions.extend(batch.regions())
                n_regions += int(st.n_regions)
                if facet_rows is not None:
                    # the inactive constraints that are facets of each region (CriticalRegion.regular_set[1]): head_i holds
                    # them per slot, -1 padded; hi[:, 1] is the slot's candidate position in this frontier
                    reg = hi[batch.slots]
                    cons = reg[:, batch.irc:batch.irc + (n_c - kk)]
                    rr, cc = numpy.nonzero(cons >= 0)
                    facet_rows[sel[reg[rr, 1]], cons[rr, cc]] = True
        status[status > REGION] = FEASIBLE   # numerically singular KKT / LP limit: no region, nothing explored from it
        examined += len(masks)
        if graph_question:
            # mpqp_combi_graph.py:114-143 -- rank deficient: subsets; region non-empty: subsets and supersets
            nonempty = status >= OPTIMAL_NO_REGION
            new = _neighbours(masks, (status == INFEASIBLE) | nonempty, nonempty, bit, eq_mask)
        else:
            # mpqp_graph.py:69-108 -- rank deficient / infeasible / not optimal: subsets; full-dimensional region: subsets and
            # the supersets through its facets; optimal but lower dimensional: nothing
            is_region = status == REGION
            new = _neighbours(masks, status != OPTIMAL_NO_REGION, is_region, bit, eq_mask, facet_rows)
        new = book.add(new)
        if profile is not None:
            profile.append({'candidates': int(len(masks)), 'regions': n_regions,
                            'status': numpy.bincount(status, minlength=4).tolist(), 'queued': int(len(new))})
        if max_candidates is not None and examined >= max_candidates:
            break
        todo = new
    else:
        solution.is_complete = True
    return solution


def _traverse_device(eng, solution: Solution, seed_masks: numpy.ndarray, profile, max_candidates, graph_question: bool) -> Solution:
    """The wave loop with the wave, the visited set and the neighbours resident on the device (csrc/graph.hpp): per wave one
    device frontier per cardinality, then one sort / deduplicate / subtract pass.  The host only collects the regions."""
    n_x, n_t, n_c, n_tc = eng.n_x, eng.n_t, eng.n_c, eng.n_tc
    eng.graph_begin(seed_masks, 0 if graph_question else 1)
    examined = 0
    while True:
        groups, _ = eng.graph_wave()
        if not groups:
            solution.is_complete = True      # the traversal ran dry: every reachable region is in the solution
            break
        n_regions, hist = 0, numpy.zeros(6, dtype=numpy.int64)
        for gi in range(len(groups)):
            st = eng.graph_group_run(gi)
            hist += numpy.array([int(v) for v in st.n_status], dtype=numpy.int64)
            if st.n_regions:
                hd, hi, er, kk, slots = eng.level_regions_slots()
                solution.critical_regions.extend(RegionBatch(hd, hi, er, n_x, n_t, n_c, n_tc, kk, slots).regions())
                n_regions += int(st.n_regions)
        n_wave = sum(c for _, c in groups)
        examined += n_wave
        n_next, _ = eng.graph_wave_close()
        if profile is not None:
            profile.append({'candidates': int(n_wave), 'regions': n_regions, 'status': hist.tolist(), 'queued': int(n_next)})
        if max_candidates is not None and examined >= max_candidates:
            break
    return solution


def solve(program, num_cores: int = -1, device: Optional[int] = None, seeds: Optional[List[List[int]]] = None,
          profile: Optional[List[Dict]] = None, max_candidates: Optional[int] = None) -> Solution:
    """mpqp_algorithm.combinatorial_graph (mpqp_combi_graph.py:68-145): all critical regions reachable from the seeds
    through chains of non-empty regions that differ by one row.  ``profile`` receives one dict per wave;
    ``max_candidates``: stop after that many examined active sets."""
    from ..region_batch import gc_paused
    with gc_paused():      # (round 6) the region objects are created with the cycle collector held, as in mpqp_hip_combinatorial.solve
        return _traverse(program, device, seeds, profile, max_candidates, True)


def solve_graph(program, num_cores: int = -1, device: Optional[int] = None, seeds: Optional[List[List[int]]] = None,
                profile: Optional[List[Dict]] = None, max_candidates: Optional[int] = None) -> Solution:
    """mpqp_algorithm.graph and its variants (mpqp_graph.py:38-108, Oberdieck et al. 2016): from every full-dimensional
    region the traversal moves to the active sets with one row less and to those with one of the region's facet constraints
    added; failed active sets only hand on their subsets.  The reference's pruning list only spares evaluations (a
    superset of an infeasible set is infeasible) and is not kept here -- the semantics of ``use_pruning=False``.  Without
    it, an infeasible set and a feasible but not optimal one are treated alike (mpqp_graph.py:69-91 hand on the same
    subsets), so the (x,theta) feasibility LP is not posed: every visited set gets the rank test, the KKT solve and the
    "region non-empty" LP of MPC_LEVEL_GRAPH.  Like the reference, the method can miss regions whose neighbours differ
    by more than one row (mpqp_graph.py:50)."""
    from ..region_batch import gc_paused
    with gc_paused():
        return _traverse(program, device, seeds, profile, max_candidates, False)


def _seed_active_sets(program, eng) -> List[List[int]]:
    """Active sets of the regions the first levels of the combinatorial algorithm find (the base set first)."""
    from .mpqp_hip_combinatorial import unpack_region
    base = numpy.arange(eng.n_eq, dtype=numpy.int32).reshape(1, -1)
    status, rd, ri, _, _ = eng.check_level(base, numpy.zeros((0, eng.mask_words), dtype=numpy.uint64), False)
    if len(rd):
        return [unpack_region(rd[0], ri[0], eng.n_x, eng.n_t, eng.n_c, eng.n_tc).active_set]
    eng.pruned_clear()
    eng.frontier_root()
    max_depth = max(eng.n_x, eng.n_t) - eng.n_eq
    for depth in range(max_depth):
        gen = depth + 1 != max_depth
        st = eng.level_run(gen)
        if st.n_regions:
            hd, hi, er, kk, slots = eng.level_regions_slots()
            return [r.active_set for r in RegionBatch(hd, hi, er, eng.n_x, eng.n_t, eng.n_c, eng.n_tc, kk, slots).regions()]
        if not gen or st.n_children == 0:
            break
        eng.frontier_advance()
    return []
