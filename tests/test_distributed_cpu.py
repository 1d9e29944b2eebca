"""World-size-2 test of the level exchange on CPU (gloo).  The per-candidate work is done by the CPU oracle behind the
engine interface of ppopt_amd/distributed.py; what is under test is the sharding, the padded all-gathers and the merge:
both ranks must end with the reference's complete region set and must have visited the reference's candidates."""
import os
import sys

import numpy
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, golden_regions, load_golden


class OracleLevelEngine:
    """CPU stand-in for HipLevelEngine (test infrastructure): same methods, CPU tensors, oracle arithmetic."""

    def __init__(self, P):
        self.P = P
        self.device = torch.device('cpu')
        self.n_x, self.n_t, self.n_c, self.n_tc, self.n_eq = P.n_x, P.n_t, P.n_c, P.n_tc, P.n_eq
        self.rec_d, self.rec_i = P.rec_d, P.rec_i
        self.pruned = []

    def clear_pruned(self):
        self.pruned = []

    def add_pruned(self, masks):
        from ppopt_amd._lib import masks_to_sets
        self.pruned.extend(masks_to_sets(masks.numpy().view(numpy.uint64)))

    def set_frontier(self, cands):
        self.cands = cands.numpy().astype(numpy.int32)

    def run(self, gen_children):
        import ctypes
        from oracle import oracle as orc
        n, k = self.cands.shape
        self.status = numpy.zeros(n, dtype=numpy.uint8)
        self.d = numpy.zeros((n, self.rec_d))
        self.i = numpy.zeros((n, self.rec_i), dtype=numpy.int32)
        if n:
            c = numpy.ascontiguousarray(self.cands)
            orc.lib().orc_check_level(ctypes.byref(self.P.cs), orc._ip(c), n, k, 1,
                                      self.status.ctypes.data_as(orc._c_uint8_p), orc._dp(self.d), orc._ip(self.i))
        self.kids = self.P.generate_children(self.cands, self.status, self.pruned, mplp_filter=not self.P.is_qp) \
            if gen_children and n else numpy.zeros((0, k + 1), dtype=numpy.int32)
        self.new = [tuple(int(v) for v in self.cands[j]) for j in numpy.nonzero((self.status == 0) | (self.status == 2))[0]]
        return {'n': n, 'status': numpy.bincount(self.status, minlength=6).tolist(), 'n_regions': int((self.status == 3).sum()),
                'n_children': len(self.kids), 'n_pruned_new': len(self.new), 'lp_pivots': 0}

    def children(self):
        return torch.from_numpy(self.kids)

    def pruned_new(self):
        from ppopt_amd._lib import sets_to_masks
        return torch.from_numpy(sets_to_masks(self.new).view(numpy.int64).reshape(-1, 2))

    def regions(self):
        sel = self.status == 3
        return torch.from_numpy(self.d[sel]), torch.from_numpy(self.i[sel])


def _worker(rank, world, port, name, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from oracle import oracle as orc
        from ppopt_amd.distributed import solve_distributed
        g = load_golden(name)
        eng = OracleLevelEngine(orc.problem_from_golden(g))
        profile = []
        sol = solve_distributed(eng, profile=profile)
        out[rank] = ([tuple(r.active_set) for r in sol.critical_regions],
                     [(p['candidates'], p['status']) for p in profile if p['depth'] > 0],
                     [p.get('local_candidates') for p in profile if p['depth'] > 0])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('name', ['dblint_n3', 'rand_4_2_10_s0', 'c1_transport_mplp'])
def test_two_rank_exchange_reproduces_reference(name):
    world = 2
    port = 29500 + (os.getpid() % 2000)
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(world, port, name, out), nprocs=world, join=True)
        res = dict(out)
    g = load_golden(name)
    ref = sorted(golden_regions(g))
    for rank in range(world):
        sets, levels, local = res[rank]
        assert sorted(sets) == ref, f'rank {rank}'
        for i, (n, hist) in enumerate(levels):
            assert n == len(g[f'L{i}_verdict'])
            assert hist[:5] == numpy.bincount(g[f'L{i}_verdict'], minlength=5).tolist()
    # the shards really were disjoint halves
    for a, b, (n, _) in zip(res[0][2], res[1][2], res[0][1]):
        assert a + b == n and abs(a - b) <= 1


def test_single_process_path_without_process_group(oracle):
    from ppopt_amd.distributed import solve_distributed
    g = load_golden('transport_mpqp')
    sol = solve_distributed(OracleLevelEngine(oracle.problem_from_golden(g)))
    assert sorted(tuple(r.active_set) for r in sol.critical_regions) == sorted(golden_regions(g))
