"""ppopt_amd -- MI355X-native combinatorial mpLP/mpQP solver with PPOPT's program / solve_mpqp API.

Host layer (Python): program classes with the reference's presolve, Solution / CriticalRegion, algorithm dispatch.
Device layer (HIP, gfx950): csrc/ behind the C ABI of include/mpcombi.h, loaded through ppopt_amd._lib.
"""
from .critical_region import CriticalRegion
from .mplp_program import MPLP_Program
from .mpqp_program import MPQP_Program
from .mpmilp_program import MPMILP_Program
from .mpmiqp_program import MPMIQP_Program
from .solution import Solution
from .solver import Solver, SolverOutput

__all__ = ['CriticalRegion', 'MPLP_Program', 'MPQP_Program', 'MPMILP_Program', 'MPMIQP_Program', 'Solution', 'Solver', 'SolverOutput']
