"""Rank-test goldens from the REAL reference (build container only): is_full_rank (utils/constraint_utilities.py:222-236, numpy's SVD
rule) on the reference's own six unit cases (tests/other_tests/test_constraint_utilities.py:81-111) and on active sets with duplicate /
near-parallel rows around the SVD threshold, each embedded in a small mpQP so that the device can be asked through the C ABI
(status MPC_INFEASIBLE <=> rank deficient: the right-hand sides make every row system consistent, so the feasibility LP never says no).

    python oracle/ref_harness/gen_rank_goldens.py        -> tests/golden/rank_cases.npz
"""
import os
import sys
import warnings

import numpy

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_goldens as gg  # noqa: E402  (loads the reference through ref_shims)

from ppopt.mpqp_program import MPQP_Program  # noqa: E402
from ppopt.utils.constraint_utilities import is_full_rank  # noqa: E402


def program_for(A, rng, idx):
    """min 1/2 x'x s.t. A x <= b (F = 0), |theta| <= 1: b = A x0 on the rows of the active set (a consistent equality system), A x0 + 1
    on the others."""
    m, n = A.shape
    x0 = rng.standard_normal((n, 1))
    b = A @ x0 + 1.0
    b[idx] -= 1.0
    return dict(A=A, b=b, c=numpy.zeros((n, 1)), H=numpy.zeros((n, 2)), Q=numpy.eye(n), A_t=numpy.vstack([numpy.eye(2), -numpy.eye(2)]),
                b_t=numpy.ones((4, 1)), F=numpy.zeros((m, 2)))


def cases():
    rng = numpy.random.default_rng(11)
    out = []
    # the reference's unit cases
    out.append(('unit1_eye5_all', numpy.eye(5), list(range(5))))
    out.append(('unit2_2x3', numpy.array([[1.0, 2, 3], [1, 0, 3]]), [0, 1]))
    e10 = numpy.eye(10); e10[-1, -1] = 0
    out.append(('unit3_eye10_zero_row', e10, list(range(10))))
    out.append(('unit4_eye4_rows123', numpy.eye(4), [1, 2, 3]))
    B = numpy.array([[1.0, 0], [1, 0], [0, 1]])
    out.append(('unit5_dup_all', B, [0, 1, 2]))
    out.append(('unit5_dup_01', B, [0, 1]))
    out.append(('unit5_dup_12', B, [1, 2]))
    out.append(('unit6_eye2_empty', numpy.eye(2), []))
    # near-parallel pairs: row 1 = row 0 + eps * (unit vector orthogonal to row 0), plus two generic rows, in R^5
    for n in (5, 12):
        base = rng.standard_normal((4, n))
        r0 = base[0] / numpy.linalg.norm(base[0])
        v = base[1] - (base[1] @ r0) * r0
        v /= numpy.linalg.norm(v)
        for e in (0, 17, 16, 15.5, 15, 14.5, 14, 13.5, 13, 12, 11, 10, 9, 8, 6, 4, 2):
            eps = 0.0 if e == 0 else 10.0 ** (-e)
            A = numpy.vstack([r0, r0 + eps * v, base[2], base[3]])
            out.append((f'near_n{n}_eps1e-{e}', A, [0, 1]))
            out.append((f'near3_n{n}_eps1e-{e}', A, [0, 1, 2]))
        # a row that is a combination of two others up to eps
        for e in (0, 16, 15, 14, 13, 12, 10, 8):
            eps = 0.0 if e == 0 else 10.0 ** (-e)
            A = numpy.vstack([base[0], base[2], 0.3 * base[0] - 1.7 * base[2] + eps * base[3], base[3]])
            out.append((f'comb_n{n}_eps1e-{e}', A, [0, 1, 2]))
        # scaled duplicates (exact, different magnitudes)
        A = numpy.vstack([base[0], 1e6 * base[0], base[2], base[3]])
        out.append((f'scaled_dup_n{n}', A, [0, 1]))
    return out


if __name__ == '__main__':
    rng = numpy.random.default_rng(5)
    rec = {}
    names = []
    for name, A, idx in cases():
        d = program_for(A, rng, idx)
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            prog = MPQP_Program(d['A'], d['b'], d['c'], d['H'], d['Q'], d['A_t'], d['b_t'], d['F'], post_process=False)
        same_rows = prog.A.shape == d['A'].shape      # (the constructor drops an all-zero row: that case is pinned on the raw matrix)
        src = {key: getattr(prog, key) for key in ('A', 'b', 'c', 'H', 'Q', 'A_t', 'b_t', 'F')} if same_rows else d
        full = bool(is_full_rank(src['A'], idx))
        if len(idx) == A.shape[0] and len(idx) > 0:
            assert full == bool(is_full_rank(src['A']))          # the reference's unit tests call it without indices
        feas = bool(prog.check_feasibility(idx)) if same_rows else full
        if feas != full:
            print(f'   (check_feasibility {feas} with full rank {full}: the LP of a nearly dependent equality system)')
        sv = numpy.linalg.svd(src['A'][idx], compute_uv=False) if len(idx) else numpy.array([1.0])
        names.append(name)
        for key in ('A', 'b', 'c', 'H', 'Q', 'A_t', 'b_t', 'F'):
            rec[f'{name}__{key}'] = src[key]
        rec[f'{name}__idx'] = numpy.array(idx, dtype=numpy.int32)
        rec[f'{name}__full_rank'] = numpy.array(full)
        rec[f'{name}__feasible'] = numpy.array(feas)
        rec[f'{name}__sv_ratio'] = numpy.array(float(sv.min() / sv.max()) if sv.max() > 0 else 0.0)
        print(f'{name:28s} full_rank {full!s:5s} check_feasibility {feas!s:5s} sigma_min/sigma_max {rec[name + "__sv_ratio"]:.3e}')
    rec['names'] = numpy.array(names)
    numpy.savez_compressed(os.path.join(gg.GOLDEN, 'rank_cases.npz'), **rec)
