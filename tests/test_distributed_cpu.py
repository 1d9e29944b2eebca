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


def pack_compact(regs, n_x, n_t, n_c, n_tc, k):
    """CriticalRegion objects -> the compact slot arrays of include/mpcombi.h (test infrastructure)."""
    fd = n_x * n_t + n_x + k * n_t + k
    fi = 8 + k + n_tc + k + 2 * (n_c - k)
    hd = numpy.zeros((len(regs), fd))
    hi = numpy.full((len(regs), fi), -1, dtype=numpy.int32)
    rows = []
    for j, r in enumerate(regs):
        hd[j, :n_x * n_t] = r.A.ravel()
        hd[j, n_x * n_t:n_x * n_t + n_x] = r.b.ravel()
        o = n_x * n_t + n_x
        hd[j, o:o + k * n_t] = r.C.ravel()
        hd[j, o + k * n_t:o + k * n_t + k] = r.d.ravel()
        nE = r.E.shape[0]
        hi[j, :8] = [3, 0, nE, len(r.omega_set), len(r.lambda_set), len(r.regular_set[0]), len(rows), 0]
        q = 8
        hi[j, q:q + k] = r.active_set; q += k
        hi[j, q:q + len(r.omega_set)] = r.omega_set; q += n_tc
        hi[j, q:q + len(r.lambda_set)] = r.lambda_set; q += k
        hi[j, q:q + len(r.regular_set[0])] = r.regular_set[0]; q += n_c - k
        hi[j, q:q + len(r.regular_set[1])] = r.regular_set[1]
        rows.extend(numpy.hstack([r.f.reshape(-1, 1), r.E]).tolist())
    er = numpy.array(rows).reshape(-1, n_t + 1)
    return hd, hi, er, k, numpy.arange(len(regs))


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

    def root(self):
        e = self.n_eq
        self.cands = numpy.array([[*range(e), i] for i in range(e, self.n_c)], dtype=numpy.int32).reshape(-1, e + 1)

    def frontier_size(self):
        return self.cands.shape

    def shard(self, rank, world):
        self.cands = numpy.ascontiguousarray(self.cands[rank::world])

    def add_pruned(self, masks):
        from ppopt_amd._lib import masks_to_sets
        self.pruned.extend(masks_to_sets(masks.numpy().view(numpy.uint64)))

    def _check(self, cands):
        import ctypes
        from oracle import oracle as orc
        n, k = cands.shape
        status = numpy.zeros(n, dtype=numpy.uint8)
        d = numpy.zeros((n, self.rec_d))
        i = numpy.zeros((n, self.rec_i), dtype=numpy.int32)
        if n:
            c = numpy.ascontiguousarray(cands)
            orc.lib().orc_check_level(ctypes.byref(self.P.cs), orc._ip(c), n, k, 1,
                                      status.ctypes.data_as(orc._c_uint8_p), orc._dp(d), orc._ip(i))
        return status, d, i

    def run(self, gen_children):
        n, k = self.cands.shape
        self.status, self.d, self.i = self._check(self.cands)
        self.kids = self.P.generate_children(self.cands, self.status, self.pruned, mplp_filter=not self.P.is_qp) \
            if gen_children and n else numpy.zeros((0, k + 1), dtype=numpy.int32)
        self.new = [tuple(int(v) for v in self.cands[j]) for j in numpy.nonzero((self.status == 0) | (self.status == 2))[0]]
        return {'n': n, 'k': k, 'status': numpy.bincount(self.status, minlength=6).tolist(), 'n_regions': int((self.status == 3).sum()),
                'n_children': len(self.kids), 'n_pruned_new': len(self.new), 'lp_pivots': 0}

    def pruned_new(self):
        from ppopt_amd._lib import sets_to_masks
        return torch.from_numpy(sets_to_masks(self.new).view(numpy.int64).reshape(-1, 2))

    def advance(self):
        self.pruned.extend(self.new)
        self.cands = self.kids

    def frontier_tensor(self):
        return torch.from_numpy(numpy.ascontiguousarray(self.cands, dtype=numpy.int32))

    def set_frontier_tensor(self, cands):
        self.cands = numpy.ascontiguousarray(cands.numpy(), dtype=numpy.int32)

    def _unpack(self, status, d, i):
        from ppopt_amd.mp_solvers.mpqp_hip_combinatorial import unpack_regions
        sel = status == 3
        return unpack_regions(d[sel], i[sel], self.n_x, self.n_t, self.n_c, self.n_tc)

    def regions(self):
        return pack_compact(self._unpack(self.status, self.d, self.i), self.n_x, self.n_t, self.n_c, self.n_tc, self.cands.shape[1])

    def check_base(self):
        base = numpy.arange(self.n_eq, dtype=numpy.int32).reshape(1, -1)
        status, d, i = self._check(base)
        return numpy.bincount(status, minlength=6).tolist(), self._unpack(status, d, i)


class _OutOfSlots(Exception):
    pass


class FlakyEngine(OracleLevelEngine):
    """Fails ONE level on ONE rank the way the HIP engine reports 'out of spare region slots' (MpcCapacityError), until the driver
    switches the overlapped region stage off."""
    capacity_error = _OutOfSlots

    def __init__(self, P, rank, fail_rank, fail_level):
        super().__init__(P)
        self.rank, self.fail_rank, self.fail_level, self.level, self.overlap, self.failed = rank, fail_rank, fail_level, 0, True, 0

    def root(self):
        self.level = 0
        super().root()

    def run(self, gen_children):
        self.level += 1
        if self.overlap and self.rank == self.fail_rank and self.level == self.fail_level:
            self.failed += 1
            raise _OutOfSlots('late optimal candidates exceed the spare region slots')
        return super().run(gen_children)

    def set_region_overlap(self, on):
        self.overlap = on


def _worker_opts(rank, world, port, name, shard_min, mode, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from oracle import oracle as orc
        from ppopt_amd.distributed import solve_distributed
        P = orc.problem_from_golden(load_golden(name))
        if mode == 'rank0':
            sol = solve_distributed(OracleLevelEngine(P), shard_min=shard_min, full_solution='rank0')
            out[rank] = ([tuple(r.active_set) for r in sol.critical_regions], 0)
        else:
            eng = FlakyEngine(P, rank, fail_rank=1, fail_level=int(mode))
            sol = solve_distributed(eng, shard_min=shard_min)
            out[rank] = ([tuple(r.active_set) for r in sol.critical_regions], eng.failed)
    finally:
        dist.destroy_process_group()


def _worker(rank, world, port, name, shard_min, out):
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
        sol = solve_distributed(eng, profile=profile, shard_min=shard_min)
        out[rank] = ([tuple(r.active_set) for r in sol.critical_regions],
                     [(p['candidates'], p['status']) for p in profile if p['depth'] > 0],
                     [p.get('local_candidates') for p in profile if p['depth'] > 0],
                     [p.get('sharded') for p in profile if p['depth'] > 0])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('name,shard_min', [('dblint_n3', 1), ('rand_4_2_10_s0', 1), ('c1_transport_mplp', 1),
                                            ('rand_4_2_10_s0', 20), ('dblint_n3', 10 ** 9)])
def test_two_rank_split_reproduces_reference(name, shard_min):
    """shard_min=1: split at the root; 20: replicated first levels, then split; 1e9: never split (pure replication)."""
    world = 2
    port = 29500 + (os.getpid() % 2000)
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(world, port, name, shard_min, out), nprocs=world, join=True)
        res = dict(out)
    g = load_golden(name)
    ref = sorted(golden_regions(g))
    for rank in range(world):
        sets, levels, local, sharded = res[rank]
        assert sorted(sets) == ref, f'rank {rank}'
        for i, (n, hist) in enumerate(levels):
            assert n == len(g[f'L{i}_verdict'])
            assert hist[:5] == numpy.bincount(g[f'L{i}_verdict'], minlength=5).tolist()
    # sharded levels are disjoint covers of the level; replicated levels are processed in full by both ranks
    for a, b, (n, _), sh in zip(res[0][2], res[1][2], res[0][1], res[0][3]):
        if sh:
            assert a + b == n
        else:
            assert a == n and b == n
    if shard_min == 1:
        assert all(res[0][3])
    if shard_min == 10 ** 9:
        assert not any(res[0][3])


def _worker_reshard(rank, world, port, name, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from oracle import oracle as orc
        from ppopt_amd import distributed
        distributed.RESHARD = 1.0001      # any imbalance of the next frontier triggers the step
        g = load_golden(name)
        profile = []
        sol = distributed.solve_distributed(OracleLevelEngine(orc.problem_from_golden(g)), profile=profile, shard_min=1)
        out[rank] = ([tuple(r.active_set) for r in sol.critical_regions],
                     [(p['candidates'], p['status'], p.get('local_candidates'), p.get('resharded')) for p in profile if p['depth'] > 0])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('name', ['dblint_n3', 'rand_4_2_10_s0'])
def test_reshard_step_keeps_the_levels_and_balances_them(name):
    """MPC_RESHARD: after a sharded level whose ranks hold unequal numbers of children, the children are
    all-gathered in rank order and every rank takes an equal slice.  With the threshold at 1.0001 the step runs whenever the shares
    differ: every level must still be the reference's (candidates, status histogram), both ranks must return the reference's regions,
    the local shares after a step differ by at most one candidate, and the step must have run at least once."""
    world = 2
    port = 33500 + (os.getpid() % 2000)
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker_reshard, args=(world, port, name, out), nprocs=world, join=True)
        res = dict(out)
    g = load_golden(name)
    ref = sorted(golden_regions(g))
    steps = 0
    for rank in range(world):
        sets, levels = res[rank]
        assert sorted(sets) == ref, f'rank {rank}'
        for i, (n, hist, local, resh) in enumerate(levels):
            assert n == len(g[f'L{i}_verdict'])
            assert hist[:5] == numpy.bincount(g[f'L{i}_verdict'], minlength=5).tolist()
    for i, ((n, _, a, resh), (_, _, b, _)) in enumerate(zip(res[0][1], res[1][1])):
        assert a + b == n
        if i > 0 and res[0][1][i - 1][3]:      # the level before ended with a re-shard: this level's shares are equal
            assert abs(a - b) <= 1
            assert sum(res[0][1][i - 1][3]['after']) == n
        steps += 1 if resh else 0
    assert steps > 0, 'the re-shard step never ran'


def test_single_process_path_without_process_group(oracle):
    from ppopt_amd.distributed import solve_distributed
    g = load_golden('transport_mpqp')
    sol = solve_distributed(OracleLevelEngine(oracle.problem_from_golden(g)))
    assert sorted(tuple(r.active_set) for r in sol.critical_regions) == sorted(golden_regions(g))


@pytest.mark.parametrize('mode', ['rank0', '1', '3'])
def test_two_rank_options(mode):
    """'rank0': only rank 0 returns the complete Solution, rank 1 the regions of the replicated levels and of its own shards.
    '1' / '3': rank 1's level 1 (replicated) / level 3 (sharded) fails with the engine's 'out of spare region slots' error --
    BOTH ranks must learn it in that level's exchange, repeat the solve together with the region overlap switched off, and
    return the complete region set (a rank that carried on alone would hang in the next collective)."""
    world, name, shard_min = 2, 'rand_4_2_10_s0', 20
    port = 31500 + (os.getpid() % 2000)
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker_opts, args=(world, port, name, shard_min, mode, out), nprocs=world, join=True)
        res = dict(out)
    ref = sorted(golden_regions(load_golden(name)))
    assert sorted(res[0][0]) == ref
    if mode == 'rank0':
        assert set(res[1][0]) < set(ref) and len(res[1][0]) > 0
    else:
        assert sorted(res[1][0]) == ref
        assert res[1][1] == 1 and res[0][1] == 0


class BrokenEngine(OracleLevelEngine):
    """Raises an ordinary exception (a device fault, out of memory ...) in ONE level on ONE rank."""

    def __init__(self, P, rank, fail_rank, fail_level):
        super().__init__(P)
        self.rank, self.fail_rank, self.fail_level, self.level = rank, fail_rank, fail_level, 0

    def root(self):
        self.level = 0
        super().root()

    def run(self, gen_children):
        self.level += 1
        if self.rank == self.fail_rank and self.level == self.fail_level:
            raise RuntimeError('device fault (test)')
        return super().run(gen_children)


def _worker_fault(rank, world, port, name, shard_min, fail_level, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import datetime
    dist.init_process_group('gloo', rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    try:
        from oracle import oracle as orc
        from ppopt_amd.distributed import DistributedLevelError, solve_distributed
        P = orc.problem_from_golden(load_golden(name))
        try:
            solve_distributed(BrokenEngine(P, rank, fail_rank=1, fail_level=fail_level), shard_min=shard_min)
            out[rank] = ('returned', None, None)
        except DistributedLevelError as exc:
            out[rank] = ('level error', exc.failed_ranks, type(exc.__cause__).__name__ if exc.__cause__ is not None else None)
        except Exception as exc:      # noqa: BLE001
            out[rank] = ('other', type(exc).__name__, str(exc)[:200])
    finally:
        dist.destroy_process_group()


def test_a_rank_that_fails_in_a_sharded_level_takes_every_rank_out_together():
    """rank 1's engine raises inside a SHARDED level (level 3 of rand_4_2_10_s0 with shard_min 20).  The failing
    rank must still join that level's statistics exchange (with a fault marker), and BOTH ranks must leave the solve with
    DistributedLevelError naming rank 1 -- the healthy rank within seconds, not after a collective's timeout (the process group of
    this test gives up after 60 s; the whole test is far below that)."""
    import time
    world, name, shard_min = 2, 'rand_4_2_10_s0', 20
    port = 33500 + (os.getpid() % 2000)
    t0 = time.time()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker_fault, args=(world, port, name, shard_min, 3, out), nprocs=world, join=True)
        res = dict(out)
    assert time.time() - t0 < 50.0
    assert res[0][0] == 'level error' and res[0][1] == [1] and res[0][2] is None
    assert res[1][0] == 'level error' and res[1][1] == [1] and res[1][2] == 'RuntimeError'
