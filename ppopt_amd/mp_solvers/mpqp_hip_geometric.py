"""The geometric algorithm (Spjøtvold et al.) on the MI355X, wave-parallel like the reference's parallel variant.

Reference: mp_solvers/mpqp_geometric.py:8-62 (serial), mp_solvers/mpqp_parallel_geometric.py:38-108 (one ``pool.map`` over
the facets of the regions found in the previous round), solver_utils.py:204-325 (``get_facet_centers``, ``fathem_facet``).
Every region hands on its facets; for each facet the algorithm steps from the facet's Chebyshev centre across the facet
(distance radius * 1e-6 * 2^j, j = 1, 2, ... while it stays below the radius), solves the QP at that parameter point and
takes the optimiser's active set as the neighbour's; it needs no assumption on how neighbouring active sets differ, which is
what makes it the robust choice for degenerate programs (the graph traversals only move by one row).  For an mpLP the program at a
parameter point is an LP (MPLP_Program.solve_theta, mplp_program.py:324-352) and the probes are a batch of the device LP plug.

Here the three kinds of work of a round are three device batches:

    facet centres   one Chebyshev LP per facet  {E theta + ||E_j|| r <= f, row i as an equality, r >= 0}    mpc_facet_centres
    probes          one QP per still-open facet and step j (linear complementarity form, csrc/qp.hpp)     mpc_qp_solve_batch
    regions         rank test, KKT solve, "region non-empty" LP, region kernel for the new active sets    MPC_LEVEL_GRAPH

and the bookkeeping (which active sets are known) is the mask book of the graph traversals.  Differences from the
reference: the shortcut "the probe point already lies in a found region" (fathem_facet :268-272, one point location per
probe) is replaced by the test that follows it anyway -- the QP's active set is already indexed; the first region comes from
the first levels of the combinatorial algorithm (deterministic) unless ``active_set`` is given, instead of random sampling
(gen_optimal_active_set).
"""
from typing import Dict, List, Optional

import numpy

from ..region_batch import RegionBatch
from ..solution import Solution
from . import mpqp_hip_combi_graph as _g

REGION = 3


def _facet_centres(E_rows: numpy.ndarray, row_off: numpy.ndarray, device: int):
    """Chebyshev centre, radius and validity of every facet (= every row) of every region.  E_rows [R, n_t + 1] = [f | E]
    stacked, row_off [n_regions + 1].  Returns (centre [R, n_t], radius [R], ok [R])."""
    from .. import _lib
    n_t = E_rows.shape[1] - 1
    R = len(E_rows)
    centre, radius, ok = numpy.zeros((R, n_t)), numpy.zeros(R), numpy.zeros(R, dtype=bool)
    if R == 0:
        return centre, radius, ok
    if n_t == 1:        # solver_utils.py:232-235: the facet of an interval is a point
        centre[:, 0] = E_rows[:, 0] / E_rows[:, 1]
        radius[:] = 1.0
        ok[:] = True
        return centre, radius, ok
    # one wavefront per facet builds its LP in LDS from the region's rows and solves it (csrc/kernels.hpp, k_facet_centres)
    centre, radius, status = _lib.facet_centres(E_rows, row_off, device)
    ok = (status == _lib.LP_OPTIMAL) & (numpy.abs(radius) > 1e-12)   # solver_utils.py:245-247: facets of numerically zero radius are skipped
    return centre, radius, ok


def _sub_active_set(program, active: List[int]) -> List[int]:
    """solver_utils.py:169-202: a full-rank subset of an overdetermined active set (equalities first, then greedily)."""
    eq = list(program.equality_indices)
    kept: List[int] = []
    rank = numpy.linalg.matrix_rank(program.A[eq]) if eq else 0
    for i in [a for a in active if a not in eq]:
        r = numpy.linalg.matrix_rank(program.A[[*eq, *kept, i]])
        if r > rank:
            kept.append(i)
            rank = r
        if rank == program.num_x():
            break
    return [*eq, *kept]


def solve(program, active_set: Optional[List[int]] = None, num_cores: int = -1, device: Optional[int] = None,
          profile: Optional[List[Dict]] = None, max_regions: Optional[int] = None) -> Solution:
    """mpqp_algorithm.geometric / geometric_parallel / geometric_parallel_exp.  ``profile`` receives one dict per round."""
    from ..mplp_program import MPLP_Program
    from ..mpqp_program import MPQP_Program
    from .. import _lib
    is_qp = isinstance(program, MPQP_Program)
    if not is_qp and not isinstance(program, MPLP_Program):
        raise NotImplementedError('the geometric algorithm needs the LP / QP of the program at a parameter point')
    if is_qp:
        ev = numpy.linalg.eigvalsh(0.5 * (program.Q + program.Q.T))
        if not ev.min() > 1e-10 * max(ev.max(), 1.0):
            raise NotImplementedError('the geometric algorithm needs a positive definite Q (the probes are QPs solved on the device)')
    eng = program.engine(device)
    dev = eng.device
    n_x, n_t, n_c, n_tc, n_eq, words = eng.n_x, eng.n_t, eng.n_c, eng.n_tc, eng.n_eq, eng.mask_words
    solution = Solution(program, [])
    eq_flags_row = numpy.zeros(n_c, dtype=numpy.uint8)
    eq_flags_row[list(program.equality_indices)] = 1

    def probe(pts: numpy.ndarray):
        """The program at the parameter points: (solved [m] bool, active [m, n_c] bool).  mpQP: the QP as a complementarity
        problem (mpc_qp_solve_batch); mpLP: the LP min (c + H theta)'x s.t. A x <= b + F theta as one batch of the LP plug
        (MPLP_Program.solve_theta, mplp_program.py:324-352), active = rows tight at the optimiser."""
        if is_qp:
            status, _, _, act = eng.qp_solve_batch(pts)
            return status == 0, act
        b = program.b.reshape(1, -1) + pts @ program.F.T
        c = program.c.reshape(1, -1) + pts @ program.H.T
        status, x, _, _ = _lib.lp_solve_batch(program.A, b, c, numpy.tile(eq_flags_row, (len(pts), 1)), device=dev)
        slack = b - x @ program.A.T
        return status == _lib.LP_OPTIMAL, numpy.abs(slack) <= 1e-9 * (1.0 + numpy.abs(b))

    if active_set is not None:
        seeds = [list(active_set)]
    elif is_qp:
        seeds = _g._seed_active_sets(program, eng)
    else:
        # an mpLP has regions only at n_x active rows: the first levels of the combinatorial algorithm would be the whole
        # enumeration.  The optimiser's active set at parameter points around the Chebyshev centre of the parameter set instead
        # (mplp_program.py:588-618, gen_optimal_active_set).
        seeds = []
        ball = program.solver.solve_lp(numpy.vstack([numpy.zeros((n_t, 1)), [[-1.0]]]),
                                       numpy.hstack([program.A_t, numpy.linalg.norm(program.A_t, axis=1, keepdims=True)]), program.b_t, []) if n_tc else None
        th0 = ball.sol[:n_t].reshape(1, -1) if ball is not None else numpy.zeros((1, n_t))
        r0 = min(float(ball.sol[-1]), 1.0) if ball is not None else 1.0
        rng0 = numpy.random.default_rng(0)
        pts0 = numpy.vstack([th0, th0 + rng0.uniform(-0.5 * r0, 0.5 * r0, (63, n_t))])
        solved, act = probe(pts0)
        for j in numpy.flatnonzero(solved):
            a = numpy.flatnonzero(act[j]).tolist()
            if len(a) > n_x:
                a = _sub_active_set(program, a)
            if len(a) == n_x and a not in seeds:
                seeds.append(a)
    if not seeds:
        return solution
    book = _g._SetBook(words)

    def build_regions(masks: numpy.ndarray):
        """Region records of the active sets that turn out to be full-dimensional regions: (accepted row indices into
        ``masks``, list of (RegionBatch, slots)) -- one device frontier per cardinality."""
        accepted, batches = [], []
        card = _g._popcount(masks)
        for k in numpy.unique(card).tolist():
            sel = numpy.flatnonzero(card == k)
            if k > min(n_x, n_c):
                continue
            eng.frontier_set(_g._masks_to_index_rows(masks[sel], k, n_c))
            st = eng.level_run(False, graph=True)
            if st.n_regions:
                hd, hi, er, kk, slots = eng.level_regions_slots()
                accepted.append(sel[hi[slots, 1]])
                batches.append((RegionBatch(hd, hi, er, n_x, n_t, n_c, n_tc, kk, slots), slots))
        return (numpy.concatenate(accepted) if accepted else numpy.zeros(0, dtype=numpy.int64)), batches

    def rows_of(batches):
        """Stacked [f | E] rows, row offsets and active-set masks of the regions of ``batches`` (in batch / slot order)."""
        ef, cnt, mk = [], [], []
        for B, slots in batches:
            nE, off = B.hi[slots, 2].astype(numpy.int64), B.hi[slots, 6].astype(numpy.int64)
            rows = numpy.repeat(off - numpy.concatenate([[0], numpy.cumsum(nE)[:-1]]), nE) + numpy.arange(int(nE.sum()))
            ef.append(B.er[rows])
            cnt.append(nE)
            act = B.hi[slots, B.iact:B.iact + B.k].astype(numpy.int64)
            m = numpy.zeros((len(slots), words), dtype=numpy.uint64)
            for col in range(B.k):
                numpy.bitwise_or.at(m, (numpy.arange(len(slots)), act[:, col] >> 6), numpy.uint64(1) << (act[:, col] & 63).astype(numpy.uint64))
            mk.append(m)
        counts = numpy.concatenate(cnt)
        return numpy.vstack(ef), numpy.concatenate([[0], numpy.cumsum(counts)]).astype(numpy.int64), numpy.concatenate(mk, axis=0)

    seed_masks = book.add(_g._sets_to_masks(seeds, words))
    _, batches = build_regions(seed_masks)
    for B, slots in batches:
        solution.critical_regions.extend(B.regions())
    while batches:
        E_rows, row_off, region_masks = rows_of(batches)
        centre, radius, ok = _facet_centres(E_rows, row_off, dev)
        facet = numpy.flatnonzero(ok)                             # open facets (row indices into E_rows)
        owner = numpy.repeat(numpy.arange(len(row_off) - 1), numpy.diff(row_off))
        normal = E_rows[:, 1:]
        new_masks_all = []
        n_qp = 0
        step = 1
        A_tT, b_t_row = numpy.ascontiguousarray(program.A_t.T), program.b_t.reshape(1, -1)
        while len(facet) and step <= 21:
            dist = radius[facet] * 1e-6 * (2.0 ** step)          # fathem_facet: dist starts at radius * 1e-6 and doubles before use
            alive = dist / 2.0 < radius[facet]                    # "while dist < radius" is tested before the doubling
            facet, dist = facet[alive], dist[alive]
            if not len(facet):
                break
            pts = centre[facet] + normal[facet] * dist[:, None]
            solved, act = probe(pts)
            n_qp += len(pts)
            # infeasible program or a point outside A_t theta <= b_t (solve_theta returns None): looking outside the feasible
            # space, the facet is done
            feasible = solved & ((pts @ A_tT - b_t_row).max(axis=1) <= 0.0) if n_tc else solved.copy()
            # active sets as masks right away (the [probes, n_c] flag array is never indexed or copied again)
            bits = numpy.packbits(act, axis=1, bitorder='little')
            pad = numpy.zeros((len(bits), words * 8), dtype=numpy.uint8)
            pad[:, :bits.shape[1]] = bits
            pm = pad.view(numpy.uint64)[feasible]
            facet = facet[feasible]
            if not len(facet):
                break
            too_many = _g._popcount(pm) > n_x
            for j in numpy.flatnonzero(too_many):                 # overdetermined active set: a full-rank subset (rare)
                full = numpy.flatnonzero(numpy.unpackbits(pm[j].view(numpy.uint8), bitorder='little')[:n_c]).tolist()
                pm[j] = _g._sets_to_masks([_sub_active_set(program, full)], words)[0]
            own = (pm == region_masks[owner[facet]]).all(axis=1)  # accidental self inclusion: step further
            # The probes of a round hit few distinct active sets (config 4: ~2e5 probes, ~2e4 sets): the book is asked, and the
            # level kernels are run, once per distinct set.  Sets are told apart by a 64-bit key (the mask itself when one word
            # holds it, the book's hash otherwise -- every key class is confirmed on the full masks, a collision falls back to
            # exact structured keys).
            one_word = words == 1 or not pm[:, 1:].any()
            key = pm[:, 0] if one_word else book._hash(pm)
            uk, first, inv = numpy.unique(key, return_index=True, return_inverse=True)
            um_all = pm[first]
            if not one_word and not (um_all[inv] == pm).all():
                uq, inv = numpy.unique(_structured(pm, words), return_inverse=True)
                um_all = uq.view(numpy.uint64).reshape(-1, words)
            known_u = _g_known(book, um_all)                      # a region (or set) already indexed: the facet is done
            keep_open = own.copy()
            new_u = numpy.flatnonzero(~known_u)
            if len(new_u):
                um = um_all[new_u]
                acc, nb = build_regions(um)                       # rank test + region kernel for the distinct new active sets
                region_u = numpy.zeros(len(um_all), dtype=bool)
                region_u[new_u[acc]] = True
                if len(acc):
                    book.add(um[acc])
                    new_masks_all.extend(nb)
                # a facet whose probe gave a set that is no full-dimensional region keeps stepping (fathem_facet :300-309)
                keep_open |= ~known_u[inv] & ~region_u[inv]
            facet = facet[keep_open]
            step += 1
        for B, slots in new_masks_all:
            solution.critical_regions.extend(B.regions())
        if profile is not None:
            profile.append({'regions_in': int(len(row_off) - 1), 'facets': int(len(E_rows)), 'facets_with_centre': int(ok.sum()),
                            'qps': int(n_qp), 'regions_out': int(sum(len(s) for _, s in new_masks_all))})
        batches = new_masks_all
        if max_regions is not None and len(solution.critical_regions) >= max_regions:
            break
    else:
        solution.is_complete = True
    return solution


def _structured(masks: numpy.ndarray, words: int) -> numpy.ndarray:
    return numpy.ascontiguousarray(masks).view([(f'w{j}', numpy.uint64) for j in range(words)]).reshape(-1)


def _g_known(book, masks: numpy.ndarray) -> numpy.ndarray:
    """bool per mask: already in the book (without adding it)."""
    if book.exact:
        return numpy.isin(_structured(masks, book.words), book.k)
    h = book._hash(masks)
    pos = numpy.searchsorted(book.h, h)
    hit = pos < len(book.h)
    hit[hit] = book.h[pos[hit]] == h[hit]
    hit[hit] = (book.m[pos[hit]] == masks[hit]).all(axis=1)
    return hit
