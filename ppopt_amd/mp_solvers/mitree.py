"""MITree: the tree of feasible binary fixations (reference: mp_solvers/mitree.py:11-120).

The reference grows the tree depth-first and asks a MILP solver at every node whether the partial fixation can
still be completed.  Here the verdict of every leaf (full fixation) comes from ONE device batch of LPs
(MPMILP_Program.leaf_feasibility), and an inner node is feasible exactly when one of its leaves is; the tree
object is kept because callers use its node structure (``left`` = 1-branch, ``right`` = 0-branch, ``fixed_bins``,
``get_full_leafs``, ``count_nodes``).
"""
import copy
from typing import List, Optional


class MITree:
    def __init__(self, problem, fixed_bins: Optional[List[int]] = None, depth: int = 0):
        self.problem = problem
        self.depth = depth
        self.bin_indices = problem.binary_indices
        self.fixed_bins = [] if fixed_bins is None else fixed_bins
        self.left: Optional['MITree'] = None
        self.right: Optional['MITree'] = None
        if depth < len(self.bin_indices):
            self.is_leaf = False
            zero_fix = [*self.fixed_bins, 0]
            one_fix = [*self.fixed_bins, 1]
            if problem.check_bin_feasibility(zero_fix):
                self.right = MITree(problem, zero_fix, depth + 1)
            if problem.check_bin_feasibility(one_fix):
                self.left = MITree(problem, one_fix, depth + 1)
        else:
            self.is_leaf = True

    def count_nodes(self) -> int:
        return 1 + sum(child.count_nodes() for child in (self.left, self.right) if child is not None)

    def get_full_leafs(self) -> List['MITree']:
        """Fully fixed descendants, 0-branch before 1-branch."""
        if self.is_leaf and self.depth == len(self.bin_indices):
            return [copy.copy(self)]
        leaves = []
        for child in (self.right, self.left):
            if child is not None:
                leaves.extend(child.get_full_leafs())
        return leaves

    def num_children(self) -> int:
        return int(self.left is not None) + int(self.right is not None)

    def leaf_path(self) -> bool:
        if self.is_leaf:
            return True
        return any(child.leaf_path() for child in (self.left, self.right) if child is not None)
