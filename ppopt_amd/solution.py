"""Solution: the program plus the list of critical regions (reference: solution.py:15-199)."""
from typing import List, Optional

import numpy

from .critical_region import CriticalRegion


class Solution:
    # Batched point location walks through adjacent regions (csrc/locate.hpp, k_locate_walk) instead of scanning the region
    # list when the solution is complete, has at least this many regions, is not overlapping and came from the device (facet
    # information per row); ``use_walk = False`` forces the scan.
    WALK_MIN_REGIONS = 2048
    use_walk = True
    # the regions cover the whole feasible parameter set (set by the solvers that run to completion): the walk is only used
    # then -- in a partial solution most walks end at a missing neighbour
    is_complete = False

    def __init__(self, program, critical_regions: List[CriticalRegion], is_overlapping: bool = False,
                 point_location_tolerance: float = 1e-5):
        self.program = program
        self.critical_regions = critical_regions
        self.is_overlapping = is_overlapping
        self.point_location_tolerance = point_location_tolerance

    def add_region(self, region: CriticalRegion) -> None:
        self.critical_regions.append(region)

    def get_region(self, theta_point: numpy.ndarray) -> Optional[CriticalRegion]:
        """First containing region, or the containing region with the lowest objective when regions may overlap
        (solution.py:60-112)."""
        tol = self.point_location_tolerance
        if not self.is_overlapping:
            for cr in self.critical_regions:
                if cr.is_inside(theta_point, tol):
                    return cr
            return None
        best, best_obj = None, float('inf')
        for cr in self.critical_regions:
            if cr.is_inside(theta_point, tol):
                obj = self.program.evaluate_objective(cr.evaluate(theta_point), theta_point)
                if obj <= best_obj:
                    best, best_obj = cr, obj
        return best

    def evaluate(self, theta_point: numpy.ndarray) -> Optional[numpy.ndarray]:
        cr = self.get_region(theta_point)
        return None if cr is None else cr.evaluate(theta_point)

    def evaluate_objective(self, theta_point) -> Optional[float]:
        x = self.evaluate(theta_point)
        return None if x is None else self.program.evaluate_objective(x, theta_point)

    # ---- batched point location on the device (the loop of get_region / evaluate over many parameter points) --------------
    def _stacked(self):
        """([f | E] rows of all regions, row offsets, [b | A] of all regions), in list order."""
        from .region_batch import BatchCriticalRegion
        regs = self.critical_regions
        n_t = self.program.num_t() if self.program is not None else regs[0].E.shape[1]
        ef_parts, cnt, xl_parts = [], [], []
        adj_masks, adj_info, adj_ok = [], [], True       # facet adjacency for the walk locator (device-solved regions only)
        n_c = self.program.num_constraints() if hasattr(self.program, 'num_constraints') else 0
        adj_ok = n_c > 0 and n_t > 1       # one parameter: regions are intervals built by the 1-D variant, whose two rows carry no facet kinds
        words = 2 if n_c <= 128 else 4
        i = 0
        while i < len(regs):
            r = regs[i]
            if isinstance(r, BatchCriticalRegion) and '_batch' in r.__dict__ and r.y_fixation is None:
                # a run of regions backed by the same level arrays: cut everything out with index arithmetic
                B = r._batch
                j = i
                while j < len(regs) and isinstance(regs[j], BatchCriticalRegion) and regs[j].__dict__.get('_batch') is B \
                        and regs[j].y_fixation is None:
                    j += 1
                slots = numpy.fromiter((q._j for q in regs[i:j]), dtype=numpy.int64, count=j - i)
                nE, off = B.hi[slots, 2].astype(numpy.int64), B.hi[slots, 6].astype(numpy.int64)
                rows = numpy.repeat(off - numpy.concatenate([[0], numpy.cumsum(nE)[:-1]]), nE) + numpy.arange(int(nE.sum()))
                ef_parts.append(B.er[rows])
                cnt.append(nE)
                A = B.hd[slots, B.oA:B.ob].reshape(len(slots), B.n_x, B.n_t)
                b = B.hd[slots, B.ob:B.oC].reshape(len(slots), B.n_x, 1)
                xl_parts.append(numpy.concatenate([b, A], axis=2))
                if adj_ok:
                    # rows of E in order: kept multiplier rows (lambda_set), kept inactive rows (regular_set[1]), kept A_t rows
                    # (omega_set); a region that lost exact duplicate rows (fewer rows than entries) gets kind "unknown"
                    hi = B.hi[slots]
                    la = hi[:, B.ila:B.ila + B.k]
                    rc_ = hi[:, B.irc:B.irc + (B.n_c - B.k)]
                    om = hi[:, B.iom:B.iom + B.n_tc]
                    pad = numpy.concatenate([numpy.where(la >= 0, la, -1), numpy.where(rc_ >= 0, rc_ + (1 << 16), -1),
                                             numpy.where(om >= 0, om + (2 << 16), -1)], axis=1)
                    n_lists = (pad >= 0).sum(axis=1)
                    good = n_lists == nE
                    info = pad[pad >= 0]
                    if not good.all():
                        # rebuild per region: unknown rows where the counts disagree
                        parts, pos = [], 0
                        for q in range(len(slots)):
                            seg = info[pos:pos + n_lists[q]]
                            pos += n_lists[q]
                            parts.append(seg if good[q] else numpy.full(int(nE[q]), 3 << 16, dtype=numpy.int64))
                        info = numpy.concatenate(parts) if parts else info
                    adj_info.append(info.astype(numpy.int32))
                    act = hi[:, B.iact:B.iact + B.k].astype(numpy.int64)
                    mk = numpy.zeros((len(slots), words), dtype=numpy.uint64)
                    for col in range(B.k):
                        numpy.bitwise_or.at(mk, (numpy.arange(len(slots)), act[:, col] >> 6),
                                            numpy.uint64(1) << (act[:, col] & 63).astype(numpy.uint64))
                    adj_masks.append(mk)
                i = j
            else:
                ef_parts.append(numpy.hstack([numpy.asarray(r.f, dtype=float).reshape(-1, 1), numpy.asarray(r.E, dtype=float).reshape(-1, n_t)]))
                cnt.append(numpy.array([ef_parts[-1].shape[0]], dtype=numpy.int64))
                bx, Ax = numpy.asarray(r.b, dtype=float).reshape(-1, 1), numpy.asarray(r.A, dtype=float).reshape(-1, n_t)
                if r.y_fixation is not None:
                    # region of a mixed-integer solution (whose E, f may have been replaced by the overlap reduction): the
                    # law of the full variable vector, binaries as constant rows (critical_region.py:64-77)
                    n_full = len(r.x_indices) + len(r.y_indices)
                    b_full, A_full = numpy.zeros((n_full, 1)), numpy.zeros((n_full, n_t))
                    b_full[r.x_indices], A_full[r.x_indices] = bx, Ax
                    b_full[r.y_indices, 0] = r.y_fixation
                    bx, Ax = b_full, A_full
                xl_parts.append(numpy.concatenate([bx[None], Ax[None]], axis=2))
                # a hand-built region carries its index sets too, but not necessarily in row order: no walk through it
                adj_info.append(numpy.full(ef_parts[-1].shape[0], 3 << 16, dtype=numpy.int32))
                mk = numpy.zeros((1, words), dtype=numpy.uint64)
                for v in (r.active_set if r.y_fixation is None else []):
                    mk[0, int(v) >> 6] |= numpy.uint64(1) << numpy.uint64(int(v) & 63)
                adj_masks.append(mk)
                if r.y_fixation is not None:
                    adj_ok = False        # regions of different fixations share active sets
                i += 1
        counts = numpy.concatenate(cnt) if cnt else numpy.zeros(0, dtype=numpy.int64)
        row_off = numpy.concatenate([[0], numpy.cumsum(counts)]).astype(numpy.int64)
        self._adjacency = (numpy.concatenate(adj_masks, axis=0), numpy.concatenate(adj_info)) if adj_ok and adj_masks else None
        return numpy.vstack(ef_parts), row_off, numpy.concatenate(xl_parts, axis=0)

    def locator(self, device: int = 0):
        """The device-resident copy of the regions; rebuilt when the region list has changed."""
        from . import _lib
        key = (len(self.critical_regions), id(self.critical_regions[0]) if self.critical_regions else 0,
               id(self.critical_regions[-1]) if self.critical_regions else 0, device)
        if getattr(self, '_locator_key', None) != key:
            if getattr(self, '_locator', None) is not None:
                self._locator.close()
            ef, row_off, xlaw = self._stacked()
            P = self.program
            self._locator = _lib.Locator(row_off, ef, xlaw, getattr(P, 'Q', None), getattr(P, 'c', None), getattr(P, 'H', None), device)
            adj = getattr(self, '_adjacency', None)
            if adj is not None and len(adj[1]) == int(row_off[-1]) and len(self.critical_regions) >= self.WALK_MIN_REGIONS:
                self._locator.set_adjacency(adj[0], adj[1], P.num_constraints())
            self._locator_key = key
        return self._locator

    def get_region_batch(self, theta_points: numpy.ndarray, device: int = 0, inclusive: bool = False) -> numpy.ndarray:
        """Index into critical_regions of get_region(theta) for every row of theta_points (-1: no region)
        (solution.py:60-112, all points at once on the GPU).  ``inclusive``: membership is ``E theta <= f + tol`` instead
        of get_region's strict ``E theta - f < tol`` (used by upop.PointLocation with tol = 0)."""
        if not self.critical_regions:
            return numpy.full(len(numpy.atleast_2d(theta_points)), -1, dtype=numpy.int64)
        return self.locator(device).query(theta_points, self.point_location_tolerance, self.is_overlapping, want_x=False,
                                          inclusive=inclusive, walk=self.use_walk and self.is_complete)[0]

    def evaluate_batch(self, theta_points: numpy.ndarray, device: int = 0, inclusive: bool = False):
        """(x* [m, n_x] (NaN rows where no region contains the point), region index [m])  -- evaluate() for many points."""
        th = numpy.atleast_2d(numpy.asarray(theta_points, dtype=float))
        if not self.critical_regions:
            return numpy.full((len(th), 0), numpy.nan), numpy.full(len(th), -1, dtype=numpy.int64)
        region, x = self.locator(device).query(th, self.point_location_tolerance, self.is_overlapping, want_x=True,
                                               inclusive=inclusive, walk=self.use_walk and self.is_complete)
        return x, region

    # ---- verification without a QP solver: the KKT conditions of the program at theta ----------------------------------------
    def kkt_residuals(self, region: CriticalRegion, theta_point: numpy.ndarray) -> dict:
        """Largest violation of each optimality condition of the program at ``theta_point`` by the region's laws
        x* = A theta + b, lambda* = C theta + d: primal feasibility, multiplier sign, stationarity, complementarity
        (all scaled by 1 + the magnitude of the quantities involved)."""
        P = self.program
        th = numpy.asarray(theta_point, dtype=float).reshape(-1, 1)
        x = numpy.asarray(region.A) @ th + numpy.asarray(region.b).reshape(-1, 1)
        lam = numpy.asarray(region.C) @ th + numpy.asarray(region.d).reshape(-1, 1)
        aset = list(region.active_set)
        n_eq = len(P.equality_indices)
        rhs = P.b + P.F @ th
        slack = rhs - P.A @ x
        scale = 1.0 + numpy.abs(rhs)
        primal = float(numpy.max(numpy.concatenate([(-slack / scale)[n_eq:].ravel(), (numpy.abs(slack) / scale)[:n_eq].ravel(), [0.0]])))
        grad = P.H @ th + P.c + P.A[aset].T @ lam
        if hasattr(P, 'Q'):
            grad = grad + P.Q @ x
        stationarity = float(numpy.max(numpy.abs(grad)) / (1.0 + numpy.max(numpy.abs(P.c)) + numpy.max(numpy.abs(lam), initial=0.0)))
        ineq = [j for j, i in enumerate(aset) if i >= n_eq]
        sign = float(max(0.0, -numpy.min(lam[ineq], initial=0.0)) / (1.0 + numpy.max(numpy.abs(lam), initial=0.0)))
        compl = float(numpy.max(numpy.abs(slack[aset]) / scale[aset], initial=0.0))
        return {'primal': primal, 'multiplier_sign': sign, 'stationarity': stationarity, 'complementarity': compl}

    def verify_theta(self, theta_point: numpy.ndarray, tol: float = 1e-6) -> bool:
        """Is the explicit solution optimal for the program at ``theta_point``?  The reference compares with a
        deterministic solve (solution.py:149-174); here the region's own x*(theta), lambda*(theta) are put through the
        KKT conditions, which are sufficient for the convex programs of this package and need no QP solver.  A point
        in no region verifies when the program is infeasible there (one LP on the device)."""
        region = self.get_region(theta_point)
        if region is None:
            P = self.program
            th = numpy.asarray(theta_point, dtype=float).reshape(-1, 1)
            if not P.valid_parameter_realization(th):
                return True
            feasible = P.solver.solve_lp(None, P.A, P.b + P.F @ th, P.equality_indices) is not None
            return not feasible
        if region.y_fixation is not None:
            raise NotImplementedError('verify_theta covers continuous programs')
        return max(self.kkt_residuals(region, theta_point).values()) <= tol

    def chebyshev_centres(self, device: int = 0):
        """(centres [R, n_theta], radii [R]) of all regions: the Chebyshev LPs of every region (chebyshev_ball.py:10-63)
        as ONE batch on the device, rows padded to the largest region."""
        from . import _lib
        regs = self.critical_regions
        n_t = self.program.num_t()
        rows = max(numpy.asarray(r.E).shape[0] for r in regs) + 1
        A = numpy.zeros((len(regs), rows, n_t + 1))
        b = numpy.ones((len(regs), rows))
        for i, r in enumerate(regs):
            E = numpy.asarray(r.E, dtype=float).reshape(-1, n_t)
            m = E.shape[0]
            A[i, :m, :n_t] = E
            A[i, :m, n_t] = numpy.linalg.norm(E, axis=1)
            b[i, :m] = numpy.asarray(r.f, dtype=float).reshape(-1)
            A[i, m, n_t] = -1.0            # -r <= 0
            b[i, m] = 0.0
        c = numpy.zeros(n_t + 1)
        c[n_t] = -1.0
        status, x, _, _ = _lib.lp_solve_batch(A, b, c, numpy.zeros((len(regs), rows), dtype=numpy.uint8), device=device)
        radii = numpy.where(status == _lib.LP_OPTIMAL, x[:, n_t], numpy.nan)
        return x[:, :n_t], radii

    def verify_solution(self, tol: float = 1e-6, device: int = 0) -> bool:
        """Every region is optimal at its own Chebyshev centre and, without overlaps, is the region found there
        (solution.py:114-147 with the KKT conditions in place of the deterministic solve)."""
        if not self.critical_regions:
            return True
        centres, radii = self.chebyshev_centres(device)
        if numpy.any(~numpy.isfinite(radii)):
            return False
        located = self.get_region_batch(centres, device)
        # the reference's own test where the program offers it: the deterministic solve at the centre (one device batch of QPs
        # / LPs, MPQP_Program.solve_theta_batch) must give the region's x* -- solution.py:128-145
        det = None
        if hasattr(self.program, 'solve_theta_batch') and all(r.y_fixation is None for r in self.critical_regions):
            try:
                det = self.program.solve_theta_batch(centres)
            except Exception:          # e.g. a positive semidefinite Q: the QP batch does not apply, the KKT test below stands alone
                det = None
        for i, region in enumerate(self.critical_regions):
            if max(self.kkt_residuals(region, centres[i]).values()) > tol:
                return False
            if det is not None and det[i] is not None and not self.is_overlapping:
                xr = region.evaluate(centres[i].reshape(-1, 1)).ravel()
                if numpy.max(numpy.abs(det[i].sol - xr)) > 1e-5 * (1.0 + numpy.max(numpy.abs(xr))):
                    return False
            if not self.is_overlapping and located[i] != i:
                # a thin region (radius below the point-location tolerance) may be preceded in the list by a neighbour that
                # contains the centre within that tolerance: consistent as long as both give the same objective there
                if located[i] < 0:
                    return False
                th = centres[i].reshape(-1, 1)
                here = self.program.evaluate_objective(region.evaluate(th), th)
                there = self.program.evaluate_objective(self.critical_regions[int(located[i])].evaluate(th), th)
                if abs(here - there) > tol * (1.0 + abs(here)):
                    return False
        return True

    def materialize(self) -> 'Solution':
        """Cuts every field of every region out of the per-level arrays the device returned (the regions a solve hands back are lazy
        views, ppopt_amd/region_batch.py) -- batch-wise, a few array operations per level.  Returns self."""
        from .region_batch import materialize_regions
        materialize_regions(self.critical_regions)
        return self

    def is_mixed_integer_sol(self) -> bool:
        from .mpmilp_program import MPMILP_Program
        return isinstance(self.program, MPMILP_Program)

    def theta_dim(self) -> int:
        return self.program.num_t()

    def __len__(self):
        return len(self.critical_regions)
