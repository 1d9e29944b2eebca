"""Deep goldens: the benchmarked solves themselves, candidate by candidate, through the REAL reference.

TEST INFRASTRUCTURE, build container only (needs /root/reference).  Usage:
    python oracle/ref_harness/gen_deep_goldens.py c4 [workers]      # -> tests/golden/c4_deep.npz
    python oracle/ref_harness/gen_deep_goldens.py c3 [workers]      # -> tests/golden/c3_deep.npz

bench.py's workloads are config 4 (`generate_mpqp(20,8,20,seed=0)`, BFS levels 1-5, 1,151,350 candidates) and
config 3 (quad tank N=10, levels 1-4).  gen_goldens.py traces only their first two levels; this script runs EVERY
candidate of EVERY benchmarked level through the reference's own primitives in full_process order
(mpqp_parrallel_combinatorial.py:17-64: check_feasibility -> check_optimality -> gen_cr_from_active_set), with HiGHS
standing in for GLPK exactly as gen_goldens.py does (ref_shims.py).

The one thing not executed by reference code is the parent's child bookkeeping at full size: CombinationTester.check
(solver_utils.py:28-46) scans every pruned tuple per child (1e5 tuples x 1e6 children).  `fast_children` below produces
the same list with a hash set of pruned tuples; the script asserts it equal to the reference's
generate_children_sets(parent, n_c, murder_list) on an evenly strided sample of parents of every level.

Stored (small enough to commit):
  L{i}_cands   uint8 [n, i+1]  candidates of level i+1 in sorted order     L{i}_verdict  uint8 [n]   (0/1/2/3/4 as gen_goldens)
  L{i}_cond    float32 [n]     cond(KKT) where the reference reached the KKT solve, else nan
  R_*          every region: active set, nE, omega / lambda / regular index sets (the index sets are what "bit-exact" is about)
  S_*          per region digests of the coefficient arrays: sum and sum of squares of A,b,C,d,E,f (compared to 1e-8 relative)
  F_*          full coefficient arrays A,b,C,d,E,f of an evenly strided sample of ~1,000 regions (F_index = positions in R_*)
  base_verdict, seconds, workers
"""
import itertools
import os
import sys
import time
import warnings

import numpy

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import gen_goldens as gg  # noqa: E402  (loads the reference through ref_shims)
from ppopt.mp_solvers.solver_utils import CombinationTester, generate_children_sets  # noqa: E402

WORKLOADS = {
    # name: (raw data builder, levels benchmarked by bench.py)
    'c4': (lambda: gg.pg.generate_mpqp_data(20, 8, 20, 0), 5),
    'c3': (lambda: gg.pg.quad_tank_data(10), 4),
}

_PROGRAM = None


def _init(program):
    global _PROGRAM
    _PROGRAM = program
    warnings.simplefilter('ignore')


def _work(cand):
    v, region, cond = gg.classify(_PROGRAM, list(cand))
    return v, region, cond


def fast_children(parents, n_c, pruned_by_size):
    """generate_children_sets (solver_utils.py:154-166) for many parents: child = parent + [j], j > parent[-1], kept
    unless some pruned tuple is a subset of it.  A pruned subset of the child that does not contain j is a subset of the
    parent, so only subsets containing j are looked up -- and the subsets of the parent too, because the pruned sets of the
    parent's own generation were not known when the parent was generated (driver :127-131 merges after the level)."""
    out = []
    sizes = sorted(pruned_by_size)
    for p in parents:
        subs = []          # subsets of the parent of every pruned size (incl. empty for size-1 lookups with j)
        for s in sizes:
            subs.append((s, list(itertools.combinations(p, s - 1)), list(itertools.combinations(p, s))))
        # the parent itself contains a pruned set?  (possible only for sets pruned at the parent's level - 1)
        dead = any(c in pruned_by_size[s] for s, _, full in subs for c in full)
        if dead:
            continue
        for j in range(p[-1] + 1, n_c):
            ok = True
            for s, less, _ in subs:
                ps = pruned_by_size[s]
                for c in less:
                    if c + (j,) in ps:
                        ok = False
                        break
                if not ok:
                    break
            if ok:
                out.append(p + (j,))
    return out


def main(name, workers):
    import multiprocess

    builder, n_levels = WORKLOADS[name]
    d = builder()
    program = gg.build_reference_program(d)
    n_c, n_x, n_t = program.num_constraints(), program.num_x(), program.num_t()
    e = len(program.equality_indices)
    assert e == 0
    print(f'== {name}: n_x {n_x} n_t {n_t} n_c {n_c}, levels 1-{n_levels}, {workers} workers', flush=True)
    t_all = time.time()
    out = {}
    murder = CombinationTester()
    pruned_by_size = {}
    to_check = [tuple(c) for c in generate_children_sets(program.equality_indices, n_c)]
    regions = []
    with multiprocess.Pool(workers, initializer=_init, initargs=(program,)) as pool:
        for lev in range(n_levels):
            to_check = sorted(to_check)
            t0 = time.time()
            res = pool.map(_work, to_check, chunksize=max(1, min(512, len(to_check) // (workers * 8) + 1)))
            verdicts = numpy.array([r[0] for r in res], dtype=numpy.uint8)
            conds = numpy.array([r[2] for r in res], dtype=numpy.float32)
            hist = numpy.bincount(verdicts, minlength=5)
            print(f'  level {lev + 1}: {len(to_check)} candidates, hist {hist}, {time.time() - t0:.0f}s', flush=True)
            out[f'L{lev}_cands'] = numpy.array(to_check, dtype=numpy.uint8).reshape(len(to_check), lev + 1)
            out[f'L{lev}_verdict'] = verdicts
            out[f'L{lev}_cond'] = conds
            regions.extend(r[1] for r in res if r[0] == 3)
            numpy.savez_compressed(f'/tmp/{name}_deep_partial.npz', **out)
            if lev + 1 == n_levels:
                break
            parents = [c for c, v in zip(to_check, verdicts) if v in (1, 3)]
            t0 = time.time()
            kids = fast_children(parents, n_c, pruned_by_size)
            # the reference's own bookkeeping on a strided sample of the parents
            sample = parents[::max(1, len(parents) // 200)]
            ref_kids = [tuple(k) for p in sample for k in generate_children_sets(list(p), n_c, murder)]
            sample_set = set(sample)
            mine = [k for k in kids if k[:-1] in sample_set]
            assert sorted(ref_kids) == sorted(mine), 'fast_children differs from the reference bookkeeping'
            new_pruned = {c for c, v in zip(to_check, verdicts) if v in (0, 2, 4)}
            murder.add_combos(new_pruned)
            pruned_by_size[lev + 1] = new_pruned
            print(f'    {len(parents)} parents -> {len(kids)} children ({time.time() - t0:.0f}s; reference bookkeeping '
                  f'agrees on {len(sample)} sampled parents, {len(ref_kids)} children)', flush=True)
            to_check = kids
    base = gg.classify(program, list(program.equality_indices))
    if base[0] == 3:
        regions.append(base[1])
    out['base_verdict'] = numpy.array(base[0])
    regions = sorted(regions, key=lambda r: (len(r.active_set), list(r.active_set)))
    nr = len(regions)
    out['R_k'] = numpy.array([len(r.active_set) for r in regions], dtype=numpy.int16)
    out['R_active'] = gg.pad_int([r.active_set for r in regions]).astype(numpy.int16)
    out['R_nE'] = numpy.array([r.E.shape[0] for r in regions], dtype=numpy.int16)
    out['R_omega'] = gg.pad_int([r.omega_set for r in regions]).astype(numpy.int16)
    out['R_lambda'] = gg.pad_int([r.lambda_set for r in regions]).astype(numpy.int16)
    out['R_regular_idx'] = gg.pad_int([r.regular_set[0] for r in regions]).astype(numpy.int16)
    out['R_regular_con'] = gg.pad_int([r.regular_set[1] for r in regions]).astype(numpy.int16)
    dig = numpy.zeros((nr, 6, 2))
    for i, r in enumerate(regions):
        for j, arr in enumerate((r.A, r.b, r.C, r.d, r.E, r.f)):
            dig[i, j] = (arr.sum(), (arr * arr).sum())
    out['S_digest'] = dig
    idx = numpy.arange(0, nr, max(1, nr // 1000))
    out['F_index'] = idx.astype(numpy.int32)
    packed = gg.pack_regions([regions[i] for i in idx], n_x, n_t)
    for key in ('R_A', 'R_b', 'R_C', 'R_d', 'R_E', 'R_f'):
        out['F_' + key[2:]] = packed[key]
    out['seconds'] = numpy.array(time.time() - t_all)
    out['workers'] = numpy.array(workers)
    path = os.path.join(gg.GOLDEN, f'{name}_deep.npz')
    numpy.savez_compressed(path, **out)
    n_cand = sum(len(out[f'L{i}_verdict']) for i in range(n_levels))
    print(f'== {name}: {n_cand} candidates, {nr} regions, {time.time() - t_all:.0f}s -> {path} '
          f'({os.path.getsize(path) / 1e6:.1f} MB)', flush=True)


def main_c5(workers):
    """Config 5 (doc/control_allocation, Q of rank 4 of 8).  The reference's own driver cannot walk this program: its first
    level is decided by KKT matrices of condition 4e16 (garbage regions or LinAlgError, mpqp_program.py:187), so
    c5_control_allocation.npz holds 29 candidates.  Here the tree is walked with the rule the build uses for such sets -- a
    candidate whose KKT matrix is singular or ill-conditioned (cond >= 1e10) is EXPANDED, never trusted and never pruned
    (MPC_SINGULAR_KKT) -- and EVERY candidate met is put through the reference's primitives (classify;
    LinAlgError -> verdict 4) with cond(KKT) recorded where the reference got as far as the KKT solve.  Tests pin the
    verdicts with cond < 1e10 (or taken before any KKT solve: infeasible / not optimal) on the candidates both sides visit."""
    import multiprocess
    d = gg.pg.control_allocation_data()
    program = gg.build_reference_program(d)
    n_c = program.num_constraints()
    max_depth = max(program.num_x(), program.num_t())
    out = {}
    t_all = time.time()
    n_pinned = n_all = 0
    regions = []
    pruned_by_size = {}
    to_check = [tuple(c) for c in generate_children_sets(program.equality_indices, n_c)]
    with multiprocess.Pool(workers, initializer=_init, initargs=(program,)) as pool:
        for lev in range(max_depth):
            to_check = sorted(to_check)
            res = pool.map(_work, to_check, chunksize=32)
            verdicts = numpy.array([r[0] for r in res], dtype=numpy.uint8)
            conds = numpy.array([r[2] for r in res], dtype=numpy.float64)
            pinned = numpy.isnan(conds) | (conds < 1e10)
            n_pinned += int(pinned.sum())
            n_all += len(to_check)
            print(f'  level {lev + 1}: {len(to_check)} candidates, reference hist {numpy.bincount(verdicts, minlength=5)}, pinned {int(pinned.sum())}', flush=True)
            out[f'L{lev}_cands'] = numpy.array(to_check, dtype=numpy.uint8).reshape(len(to_check), lev + 1)
            out[f'L{lev}_verdict'] = verdicts
            out[f'L{lev}_cond'] = conds
            regions.extend(r[1] for r, ok in zip(res, pinned) if r[0] == 3 and ok)
            if lev + 1 == max_depth:
                break
            expand = [(v in (1, 3)) or not ok for v, ok in zip(verdicts.tolist(), pinned.tolist())]
            parents = [c for c, e in zip(to_check, expand) if e]
            kids = fast_children(parents, n_c, pruned_by_size)
            pruned_by_size[lev + 1] = {c for c, e in zip(to_check, expand) if not e}
            to_check = kids
            if not to_check:
                break
    out['n_levels'] = numpy.array(lev + 1)
    regions = sorted(regions, key=lambda r: (len(r.active_set), list(r.active_set)))
    out['R_k'] = numpy.array([len(r.active_set) for r in regions], dtype=numpy.int16)
    out['R_active'] = gg.pad_int([r.active_set for r in regions]).astype(numpy.int16)
    out['R_nE'] = numpy.array([r.E.shape[0] for r in regions], dtype=numpy.int16)
    out['R_omega'] = gg.pad_int([r.omega_set for r in regions]).astype(numpy.int16)
    out['R_lambda'] = gg.pad_int([r.lambda_set for r in regions]).astype(numpy.int16)
    out['R_regular_idx'] = gg.pad_int([r.regular_set[0] for r in regions]).astype(numpy.int16)
    out['R_regular_con'] = gg.pad_int([r.regular_set[1] for r in regions]).astype(numpy.int16)
    dig = numpy.zeros((len(regions), 6, 2))
    for i, r in enumerate(regions):
        for j, arr in enumerate((r.A, r.b, r.C, r.d, r.E, r.f)):
            dig[i, j] = (arr.sum(), (arr * arr).sum())
    out['S_digest'] = dig
    out['seconds'] = numpy.array(time.time() - t_all)
    path = os.path.join(gg.GOLDEN, 'c5_deep.npz')
    numpy.savez_compressed(path, **out)
    print(f'== c5: {n_all} candidates, {n_pinned} pinned, {len(regions)} pinned regions, {time.time() - t_all:.0f}s -> {path}', flush=True)


if __name__ == '__main__':
    if sys.argv[1] == 'c5':
        main_c5(int(sys.argv[2]) if len(sys.argv) > 2 else 2)
    else:
        main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 6)
