"""Host mirrors of the frontier bookkeeping of the combinatorial algorithms.

``CombinationTester`` and ``generate_children_sets`` follow mp_solvers/solver_utils.py:15-55 and :154-166 of the
reference.  The product path keeps the same information on the device as 128-bit masks (csrc/kernels.hpp,
k_children_count); these classes serve API parity, the tests and the multi-GPU host bookkeeping.
"""
from typing import Iterable, List, Optional, Set, Tuple


class CombinationTester:
    """Remembers pruned active sets; ``check`` rejects every superset of a remembered set."""

    def __init__(self):
        self.combos: Set[Tuple[int, ...]] = set()
        self.new_combos: Set[Tuple[int, ...]] = set()

    def check(self, active_set: Iterable[int]) -> bool:
        s = active_set if isinstance(active_set, set) else set(active_set)
        if not s:
            return True
        return not any(s.issuperset(c) for c in self.combos)

    def add_combo(self, active_set) -> None:
        self.combos.add(tuple(active_set))

    def add_combos(self, set_list: Iterable[Tuple[int, ...]]) -> None:
        self.combos.update(set_list)


def generate_children_sets(active_set, num_constraints: int, murder_list: Optional[CombinationTester] = None) -> \
        List[List[int]]:
    """All supersets with one more index, larger than the last one, that survive the pruning list."""
    ok = (lambda s: True) if murder_list is None else murder_list.check
    start = active_set[-1] + 1 if len(active_set) else 0
    return [[*active_set, i] for i in range(start, num_constraints) if ok([*active_set, i])]
