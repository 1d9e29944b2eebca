"""Solution: the program plus the list of critical regions (reference: solution.py:15-199)."""
from typing import List, Optional

import numpy

from .critical_region import CriticalRegion


class Solution:
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

    def theta_dim(self) -> int:
        return self.program.num_t()

    def __len__(self):
        return len(self.critical_regions)
