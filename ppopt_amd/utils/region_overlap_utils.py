"""Overlap removal for 1-parameter mixed-integer solutions (reference: utils/region_overlap_utils.py:15-262).

After enumeration the regions of different binary fixations overlap.  With one parameter every region is an
interval and every objective is affine in theta, so for each ordered pair of regions the cheaper one keeps the
shared stretch: containment splits the outer interval, partial overlap moves one end point, and crossing objectives
cut at the crossing point.  Equal objectives over the whole shared stretch are kept as they are (dual degeneracy).
The pair order, the in-place end-point updates and the tolerances follow the reference so that the resulting list
is the same, region for region.  Pure host bookkeeping on a handful of intervals -- nothing here is device work.
"""
import copy
from collections import deque
from itertools import permutations
from typing import List, Optional, Tuple

import numpy

from ..critical_region import CriticalRegion


def get_bounds_1d(E: numpy.ndarray, f: numpy.ndarray) -> Tuple[float, float]:
    """(lower, upper) of {theta: E theta <= f} for one parameter (mpqp_utils.py:304-315)."""
    lo, hi = float('-inf'), float('inf')
    for coef, rhs in zip(numpy.asarray(E).reshape(-1), numpy.asarray(f).reshape(-1)):
        if coef > 0:
            hi = min(hi, rhs / coef)
        else:
            lo = max(lo, rhs / coef)
    return lo, hi


def _bounds(cr: CriticalRegion) -> Tuple[float, float]:
    return get_bounds_1d(cr.E, cr.f)


def cr_new_bounds(cr: CriticalRegion, lb_new: Optional[float], ub_new: Optional[float]) -> CriticalRegion:
    """Replaces the interval of ``cr`` in place; ``None`` keeps that end."""
    lb, ub = _bounds(cr)
    lb = lb if lb_new is None else lb_new
    ub = ub if ub_new is None else ub_new
    cr.E = numpy.array([[1.0], [-1.0]])
    cr.f = numpy.array([[ub], [-lb]], dtype=numpy.float64)
    return cr


def full_overlap(cr_1: CriticalRegion, cr_2: CriticalRegion) -> bool:
    """cr_2 lies inside cr_1."""
    (lb1, ub1), (lb2, ub2) = _bounds(cr_1), _bounds(cr_2)
    return lb1 <= lb2 and ub1 >= ub2


def partial_overlap(cr_1: CriticalRegion, cr_2: CriticalRegion) -> bool:
    """cr_1 starts left of cr_2 and ends inside it."""
    (lb1, ub1), (lb2, ub2) = _bounds(cr_1), _bounds(cr_2)
    return lb1 < lb2 < ub1 < ub2


def find_overlap_bounds(cr_1: CriticalRegion, cr_2: CriticalRegion) -> Tuple[float, float]:
    ends = sorted([*_bounds(cr_1), *_bounds(cr_2)])
    return ends[1], ends[2]


def evaluate_objective_at_overlap_bounds(program, cr_1: CriticalRegion, cr_2: CriticalRegion):
    """(f_1(lower), f_1(upper), f_2(lower), f_2(upper)) at the ends of the shared stretch."""
    vals = []
    ends = [numpy.array([[v]]) for v in find_overlap_bounds(cr_1, cr_2)]
    for cr in (cr_1, cr_2):
        for theta in ends:
            vals.append(program.evaluate_objective(cr.evaluate(theta), theta))
    return tuple(vals)


def equal_linear_objective(program, cr_1, cr_2) -> bool:
    f1l, f1u, f2l, f2u = evaluate_objective_at_overlap_bounds(program, cr_1, cr_2)
    return f1l == f2l and f1u == f2u


def region_dominates(program, cr_1, cr_2) -> bool:
    """cr_1 is at least as cheap as cr_2 at both ends (hence everywhere) of the shared stretch."""
    f1l, f1u, f2l, f2u = evaluate_objective_at_overlap_bounds(program, cr_1, cr_2)
    return f1l <= f2l and f1u <= f2u


def compute_objective_intersection_point(program, cr_1, cr_2) -> Tuple[float, bool]:
    """Where the two affine objectives cross inside the shared stretch, and whether cr_1 is the cheaper one to the
    left of that point."""
    f1l, f1u, f2l, f2u = evaluate_objective_at_overlap_bounds(program, cr_1, cr_2)
    lower, upper = find_overlap_bounds(cr_1, cr_2)
    point = (f2l - f1l) / ((f1u - f1l) - (f2u - f2l)) * (upper - lower) + lower
    return point, f1l < f2l


def split_outer_region(new_regions: List[CriticalRegion], outer_region: CriticalRegion,
                       inner_region: CriticalRegion):
    """The inner region is cheaper throughout: the outer one keeps the part left of it, a copy gets the part right."""
    inner_lb, inner_ub = _bounds(inner_region)
    right_part = copy.deepcopy(outer_region)
    new_regions.append(right_part)
    cr_new_bounds(outer_region, None, inner_lb)
    cr_new_bounds(right_part, inner_ub, None)
    return new_regions, outer_region


def adjust_fully_overlapping_regions(program, new_regions: List[CriticalRegion], inner_region: CriticalRegion,
                                     outer_region: CriticalRegion):
    """Objectives cross inside the inner region: the inner one shrinks to its cheaper side, the outer one is cut
    into the piece left of it and a copy right of it."""
    point, outer_cheaper_left = compute_objective_intersection_point(program, outer_region, inner_region)
    right_part = copy.deepcopy(outer_region)
    new_regions.append(right_part)
    inner_lb, inner_ub = _bounds(inner_region)
    if outer_cheaper_left:
        cr_new_bounds(outer_region, None, point)
        cr_new_bounds(inner_region, point, None)
        cr_new_bounds(right_part, inner_ub, None)
    else:
        cr_new_bounds(outer_region, None, inner_lb)
        cr_new_bounds(inner_region, None, point)
        cr_new_bounds(right_part, point, None)
    return new_regions, inner_region, outer_region


def _listed(cr, group) -> bool:
    return any(cr is other for other in group)


def identify_overlaps_1d(program, regions: List[CriticalRegion]) -> Tuple[bool, List[CriticalRegion]]:
    new_regions: List[CriticalRegion] = []
    removed: List[CriticalRegion] = []
    degenerate = False
    queue = deque(permutations(regions, 2))
    while queue:
        cr_1, cr_2 = queue.popleft()
        if _listed(cr_1, removed) or _listed(cr_2, removed):
            continue
        added = False
        if full_overlap(cr_1, cr_2):
            if equal_linear_objective(program, cr_1, cr_2):
                degenerate = True
            elif region_dominates(program, cr_1, cr_2):
                removed.append(cr_2)
            elif region_dominates(program, cr_2, cr_1):
                split_outer_region(new_regions, outer_region=cr_1, inner_region=cr_2)
                added = True
            else:
                adjust_fully_overlapping_regions(program, new_regions, inner_region=cr_2, outer_region=cr_1)
                added = True
        elif partial_overlap(cr_1, cr_2):
            if region_dominates(program, cr_1, cr_2):
                cr_new_bounds(cr_2, _bounds(cr_1)[1], None)
            elif region_dominates(program, cr_2, cr_1):
                cr_new_bounds(cr_1, None, _bounds(cr_2)[0])
            else:
                point = compute_objective_intersection_point(program, cr_1, cr_2)[0]
                cr_new_bounds(cr_1, None, point)
                cr_new_bounds(cr_2, point, None)
        if added:
            # the new piece is a part of cr_1, so it only has to meet the other regions
            others = [cr for cr in regions if cr is not cr_1 and cr is not cr_2 and not _listed(cr, removed)]
            queue.extend([r, new_regions[-1]] for r in others)
            queue.extend([new_regions[-1], r] for r in others)

    result = [cr for cr in [*regions, *new_regions] if not _listed(cr, removed)]
    spans = [_bounds(cr) for cr in result]
    result = [cr for cr, (lo, hi) in zip(result, spans) if abs(lo - hi) > 1e-8]
    return degenerate, result


def reduce_overlapping_critical_regions_1d(program, regions: List[CriticalRegion]):
    """Returns (regions without resolvable overlaps, True if overlaps may remain)."""
    if program.num_t() != 1:
        raise ValueError('reduce_overlapping_critical_regions_1d requires a 1d-parameter problem')
    still_overlapping, regions = identify_overlaps_1d(program, regions)
    return regions, still_overlapping
