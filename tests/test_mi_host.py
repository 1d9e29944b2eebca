"""Host logic of the mixed-integer caller against goldens captured from the reference
(oracle/ref_harness/gen_mi_goldens.py); no device needed."""
import glob
import os

import numpy
import pytest

from ppopt_amd.critical_region import CriticalRegion
from ppopt_amd.solver import Solver
from ppopt_amd.utils.region_overlap_utils import get_bounds_1d, reduce_overlapping_critical_regions_1d

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
MI_FILES = sorted(glob.glob(os.path.join(GOLDEN, 'mi_*.npz')))


def unpack_regions(g, prefix, with_y=True):
    regs = []
    for i in range(int(g[prefix + 'n'])):
        k, ne = int(g[prefix + 'k'][i]), int(g[prefix + 'nE'][i])
        r = CriticalRegion(g[prefix + 'A'][i].copy(), g[prefix + 'b'][i].reshape(-1, 1).copy(),
                           g[prefix + 'C'][i, :k].copy(), g[prefix + 'd'][i, :k].reshape(-1, 1).copy(),
                           g[prefix + 'E'][i, :ne].copy(), g[prefix + 'f'][i, :ne].reshape(-1, 1).copy(),
                           g[prefix + 'as'][i, :k].tolist())
        if with_y:
            r.y_fixation = g[prefix + 'y'][i].tolist()
            r.y_indices = g['binary_indices'].tolist()
            r.x_indices = g['cont_indices'].tolist()
        regs.append(r)
    return regs


class _ObjectiveOnly:
    """The part of an MPMILP_Program the overlap reduction touches, built from the golden's presolved matrices."""

    def __init__(self, g):
        self.c, self.H, self.c_c, self.c_t, self.Q_t = g['proc_c'], g['proc_H'], g['proc_c_c'], g['proc_c_t'], g['proc_Q_t']
        self._nt = g['proc_F'].shape[1]

    def num_t(self):
        return self._nt

    def evaluate_objective(self, x, th):
        v = th.T @ self.H.T @ x + self.c.T @ x + self.c_c + self.c_t.T @ th + 0.5 * th.T @ self.Q_t @ th
        return float(v[0, 0])


def test_goldens_present():
    assert len(MI_FILES) == 13


def test_binary_fixations_order_is_tree_leaf_order():
    fix = Solver.binary_fixations(3)
    assert fix.tolist() == [[0, 0, 0], [0, 0, 1], [0, 1, 0], [0, 1, 1], [1, 0, 0], [1, 0, 1], [1, 1, 0], [1, 1, 1]]
    assert Solver.binary_fixations(0).shape == (1, 0)
    for path in MI_FILES:
        g = numpy.load(path)
        nb = len(g['binary_indices'])
        order = {tuple(r): i for i, r in enumerate(Solver.binary_fixations(nb).tolist())}
        ranks = [order[tuple(c)] for c in g['combos'].tolist()]
        assert ranks == sorted(ranks), path


@pytest.mark.parametrize('path', MI_FILES, ids=[os.path.basename(p)[3:-4] for p in MI_FILES])
def test_overlap_reduction_matches_reference(path):
    """P_ (enumeration output of the reference) -> our reduction -> F_ (the reference's final list), in list order."""
    g = numpy.load(path)
    reducible = g['proc_F'].shape[1] == 1 and 'proc_Q' not in g.files
    if not reducible:
        assert int(g['P_n']) == int(g['F_n'])
        return
    regions, still = reduce_overlapping_critical_regions_1d(_ObjectiveOnly(g), unpack_regions(g, 'P_'))
    want = unpack_regions(g, 'F_')
    assert still == bool(g['F_overlapping'])
    assert len(regions) == len(want)
    for got, ref in zip(regions, want):
        assert got.y_fixation == ref.y_fixation and got.active_set == ref.active_set
        numpy.testing.assert_allclose(get_bounds_1d(got.E, got.f), get_bounds_1d(ref.E, ref.f), rtol=0, atol=1e-9)
        numpy.testing.assert_allclose(got.A, ref.A, atol=1e-12)
        numpy.testing.assert_allclose(got.b, ref.b, atol=1e-12)


def test_reduction_rejects_more_than_one_parameter():
    g = numpy.load(os.path.join(GOLDEN, 'mi_acevedo_mpmilp.npz'))
    with pytest.raises(ValueError):
        reduce_overlapping_critical_regions_1d(_ObjectiveOnly(g), [])


def test_mitree_from_leaf_table():
    """Tree structure from a table of leaf verdicts: node count and leaf order of the reference's goldens."""
    from ppopt_amd.mp_solvers.mitree import MITree

    class Fake:
        def __init__(self, nb, feasible):
            self.binary_indices = list(range(nb))
            self._ok = set(map(tuple, feasible))

        def check_bin_feasibility(self, fix):
            return any(c[:len(fix)] == tuple(fix) for c in self._ok)

    for path in MI_FILES:
        g = numpy.load(path)
        nb = len(g['binary_indices'])
        tree = MITree(Fake(nb, g['combos'].tolist()))
        assert tree.count_nodes() == int(g['n_nodes']), path
        assert [leaf.fixed_bins for leaf in tree.get_full_leafs()] == g['combos'].tolist(), path
