"""PointLocation: the compiled point-location object of the reference (upop/point_location.py:10-120), backed by the
MI355X locator (csrc/locate.hpp, ``mpc_locator_*``) instead of numba on the host.

Same semantics: a point is inside a region when ``E theta <= f`` holds (inclusive, no tolerance, :46,:59 -- a point on a
facet, a vertex of the parameter box or theta = 0 on a ``-theta <= 0`` row belongs to the region); without overlaps
the first region in list order wins, with overlaps the containing region with the lowest objective (ties to the later
one, :72-78).  ``locate`` / ``evaluate`` / ``is_inside`` take one column vector like the reference; ``locate_batch`` /
``evaluate_batch`` take [m, n_theta] arrays and are what the device is for.
"""
from typing import Optional

import numpy

from ..solution import Solution


class PointLocation:
    def __init__(self, solution: Solution, device: int = 0):
        self.solution = solution
        self.num_regions = len(solution.critical_regions)
        # a shallow twin with zero tolerance, queried in the locator's inclusive mode: E theta <= f as posed
        self._exact = Solution(solution.program, solution.critical_regions, solution.is_overlapping, 0.0)
        self._device = device
        self._loc = self._exact.locator(device) if self.num_regions else None

    def locate_batch(self, thetas: numpy.ndarray) -> numpy.ndarray:
        """Region index per row of ``thetas`` (-1: in no region)."""
        thetas = numpy.ascontiguousarray(thetas, dtype=numpy.float64).reshape(-1, self.solution.program.num_t())
        if self._loc is None:
            return numpy.full(len(thetas), -1, dtype=numpy.int64)
        return self._exact.get_region_batch(thetas, self._device, inclusive=True)

    def evaluate_batch(self, thetas: numpy.ndarray):
        """(x* [m, n_x] with NaN rows where no region contains the point, region index [m])."""
        thetas = numpy.ascontiguousarray(thetas, dtype=numpy.float64).reshape(-1, self.solution.program.num_t())
        return self._exact.evaluate_batch(thetas, self._device, inclusive=True)

    def locate(self, theta: numpy.ndarray) -> int:
        return int(self.locate_batch(numpy.asarray(theta, dtype=numpy.float64).reshape(1, -1))[0])

    def is_inside(self, theta: numpy.ndarray) -> bool:
        return self.locate(theta) != -1

    def evaluate(self, theta: numpy.ndarray) -> Optional[numpy.ndarray]:
        idx = self.locate(theta)
        if idx < 0:
            return None
        return self.solution.critical_regions[idx].evaluate(numpy.asarray(theta, dtype=numpy.float64).reshape(-1, 1))
