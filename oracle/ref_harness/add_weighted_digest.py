"""Adds an ORDER-SENSITIVE digest to the deep goldens (c4_deep / c3_deep / c5_deep): S_wdigest[i, j] = sum_m (m + 1) * a_m over the
flattened (C order) array a = A, b, C, d, E, f of region i.  Sum and sum of squares (S_digest) do not change when rows of A / C / E are
swapped; this one does.  The regions are rebuilt by the REAL reference's gen_cr_from_active_set from the stored active sets (build
container only); the rebuilt arrays must reproduce the stored S_digest before anything is written.

    python oracle/ref_harness/add_weighted_digest.py c4|c3|c5 [workers]
"""
import os
import sys
import warnings

import numpy

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_goldens as gg  # noqa: E402
from ppopt.utils.mpqp_utils import gen_cr_from_active_set  # noqa: E402

BUILD = {'c4': lambda: gg.pg.generate_mpqp_data(20, 8, 20, 0), 'c3': lambda: gg.pg.quad_tank_data(10), 'c5': lambda: gg.pg.control_allocation_data()}
_P = None


def _init(program):
    global _P
    _P = program
    warnings.simplefilter('ignore')


def _work(active):
    r = gen_cr_from_active_set(_P, list(active))
    out = []
    for arr in (r.A, r.b, r.C, r.d, r.E, r.f):
        a = numpy.asarray(arr, dtype=numpy.float64).ravel()
        out.append((a.sum(), (a * a).sum(), float(numpy.dot(numpy.arange(1, a.size + 1, dtype=numpy.float64), a))))
    return out


if __name__ == '__main__':
    import multiprocess
    name = sys.argv[1]
    workers = int(sys.argv[2]) if len(sys.argv) > 2 else 7
    path = os.path.join(gg.GOLDEN, name + '_deep.npz')
    z = dict(numpy.load(path))
    program = gg.build_reference_program(BUILD[name]())
    acts = [tuple(int(v) for v in row[:k]) for row, k in zip(z['R_active'], z['R_k'])]
    with multiprocess.Pool(workers, initializer=_init, initargs=(program,)) as pool:
        res = pool.map(_work, acts, chunksize=16)
    dig = numpy.array(res)                      # [nr, 6, 3]
    assert numpy.allclose(dig[:, :, :2], z['S_digest'], rtol=1e-12, atol=1e-12), 'the rebuilt regions do not reproduce the stored digests'
    z['S_wdigest'] = dig[:, :, 2].copy()
    numpy.savez_compressed(path, **z)
    print(f'{name}: {len(acts)} regions, weighted digests added -> {path} ({os.path.getsize(path) / 1e6:.1f} MB)')
