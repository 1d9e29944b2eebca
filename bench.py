#!/usr/bin/env python3
"""Benchmark of the combinatorial mpQP hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload c4|c3|c2]

A "step" is one complete pass of the parallel combinatorial algorithm over the workload program: every BFS level
(verdict -> region -> child generation on the device, region records copied back and turned into CriticalRegion
objects) plus the final base-set check.  The program's presolved matrices are resident in HBM before the timed region
starts.  Metric: candidate active sets checked per second, whole job (BASELINE.json); regions/s is reported next to it.

N > 1 is launched by the driver as  python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...  : one
process per GPU, the frontier of every level sharded over the ranks, one RCCL exchange per level
(ppopt_amd/distributed.py).  The problem is one program, so total work is fixed: scaling is "strong".

rank 0 prints ONE JSON line.  Besides the contract keys it carries
  roofline      algorithmic HBM bytes of the dominant kernel over its HIP-event time (SURVEY.md §8(d) formula)
  cpu_baseline  the CPU oracle (C port of the reference algorithm, OpenMP) timed on a bounded sample of the same
                candidates on this box's host cores -- a reported baseline, not the target.
"""
import argparse
import json
import os
import sys
import time
import warnings

import numpy

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (description, builder, max_levels)
    'c4': ('random dense mpQP generate_mpqp(n_x=20, n_theta=8, m=20, seed=0): 60 rows, 47 after presolve; '
           'BFS levels 1-5 (1,151,349 candidate active sets)', lambda pg: pg.generate_mpqp_data(20, 8, 20, 0), 5),
    'c3': ('quad-tank MPC mpQP, 4 states / 2 inputs, N=10 condensed (n_x=20, n_theta=4, 80 rows after presolve); '
           'BFS levels 1-4 (1,076,342 candidates)', lambda pg: pg.quad_tank_data(10), 4),
    'c2': ('double-integrator explicit-MPC mpQP, N=5 (n_x=15, n_theta=2, 32 rows, 10 equalities); full tree '
           '(4,795 candidates, 9 regions)', lambda pg: pg.double_integrator_data(5), None),
    # BASELINE.json quotes config 2 with "~50 regions": the doc formulation's state box |x| <= 4 gives 9 regions, |x| <= 20 gives 51
    'c2x20': ('double-integrator explicit-MPC mpQP, N=5, state box |x| <= 20 (n_x=15, n_theta=2, 10 equalities); full tree '
              '(4,381 candidates, 51 regions)', lambda pg: pg.double_integrator_data(5, x_bound=20.0), None),
    # BASELINE config 5: the reference's control-allocation example (doc/control_allocation_example.rst), degenerate / LICQ-violating active sets
    'c5': ('control-allocation mpQP of the reference documentation (degenerate and rank-deficient active sets); full tree',
           lambda pg: pg.control_allocation_data(), None),
    # BASELINE config 1: the 2-variable / 2-parameter transport mpLP of the reference tutorial (plumbing case)
    'c1': ('transport mpLP of the reference tutorial (2 variables, 2 parameters); full tree', lambda pg: pg.transport_mplp_data(), None),
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def program_from_data(d, device=0):
    from ppopt_amd import MPLP_Program, MPQP_Program, Solver
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        if d['Q'] is None:
            return MPLP_Program(d['A'], d['b'], d['c'], d['H'], d['A_t'], d['b_t'], d['F'],
                                equality_indices=d['equality_indices'], solver=Solver(device=device))
        return MPQP_Program(d['A'], d['b'], d['c'], d['H'], d['Q'], d['A_t'], d['b_t'], d['F'],
                            equality_indices=d['equality_indices'], solver=Solver(device=device))


def build_program(name, device=0):
    from ppopt_amd import problem_generator as pg
    return program_from_data(WORKLOADS[name][1](pg), device)


def theta_box(prog):
    """(lo, hi) of the parameter set when it is a box: read off the rows of A_t theta <= b_t that have a single entry."""
    nt = prog.num_t()
    lo, hi = numpy.full(nt, -numpy.inf), numpy.full(nt, numpy.inf)
    for row, rhs in zip(prog.A_t, prog.b_t.ravel()):
        nz = numpy.flatnonzero(numpy.abs(row) > 1e-12)
        if len(nz) == 1:
            j = int(nz[0])
            if row[j] > 0:
                hi[j] = min(hi[j], rhs / row[j])
            else:
                lo[j] = max(lo[j], rhs / row[j])
    return lo, hi


def algorithmic_bytes(prog, k, rho, n_e):
    """B_alg per candidate: shared problem block streamed once per wavefront + candidate indices +
    status/child count + (regions per candidate) x region record."""
    nx, nt, nc, ntc = prog.num_x(), prog.num_t(), prog.num_constraints(), prog.A_t.shape[0]
    P = 8 * (nc * (nx + nt + 1) + nx * nx + nx * nt + nx + ntc * (nt + 1))
    R = 8 * (nt + 1) * (nx + k + n_e) + 4 * (2 * k + n_e)
    return P + 4 * k + 8 + rho * R


def cpu_baseline(prog, frontiers, gpu_status, target_candidates):
    """Times the CPU oracle on an evenly strided sample of every level's candidates (same mix as the workload) and
    compares its verdicts with the GPU's for the same candidates."""
    from oracle import oracle as orc
    orc.build()
    P = orc.OracleProblem(prog.A, prog.b, prog.F, prog.c, prog.H, getattr(prog, 'Q', None), prog.A_t, prog.b_t,
                          len(prog.equality_indices))
    cores = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    total = sum(len(f) for f in frontiers)
    frac = min(1.0, target_candidates / max(total, 1))
    samples, picks = [], []
    for f in frontiers:
        take = max(1, int(round(frac * len(f))))
        idx = numpy.linspace(0, len(f) - 1, take).astype(numpy.int64)
        samples.append(numpy.ascontiguousarray(f[idx]))
        picks.append(idx)
    # Thread scaling of the port on ONE sub-sample -- every 32nd candidate of the sample, level by level, the same candidates for every
    # thread count (1, 8, 64, all the box offers): a coherent row, bounded at about ten seconds in all.  The headline figure is then
    # taken with the thread count that scaled best (on the GPU boxes the OpenMP loop stops scaling well below the 256 hardware threads
    # the container reports: `cores` is the count actually used, `cores_available` what the box offers).
    subs = [numpy.ascontiguousarray(s[::32]) for s in samples]
    n_sub = sum(len(x) for x in subs)
    scaling = []
    for th in sorted({1, 8, 64, cores} & set(range(1, cores + 1))):
        P.check_level(subs[0][:min(8, len(subs[0]))], th, False)
        t1 = time.perf_counter()
        for sub in subs:
            if len(sub):
                P.check_level(sub, th, False)
        dts = time.perf_counter() - t1
        scaling.append({'threads': th, 'candidates_per_s': n_sub / max(dts, 1e-9), 'seconds': dts})
    one = scaling[0]['candidates_per_s']
    for row in scaling:
        row['speedup_over_one_thread'] = row['candidates_per_s'] / one
    cores_available = cores
    cores = max(scaling, key=lambda r: r['candidates_per_s'])['threads']
    P.check_level(samples[0][:min(8, len(samples[0]))], cores, False)  # warm the thread pool
    t0 = time.perf_counter()
    n = 0
    regions = 0
    differ = 0
    for s, idx, gst in zip(samples, picks, gpu_status):
        status, _ = P.check_level(s, cores, False)
        n += len(s)
        regions += int((status == orc.REGION).sum())
        differ += int((status != gst[idx]).sum())
    dt = time.perf_counter() - t0
    return {'value': n / dt, 'unit': 'candidate active sets checked/s', 'cores': cores, 'kind': 'port',
            'label': 'CPU port of the reference algorithm (oracle/mpcombi_oracle.c: BLAS-free C, every LP posed the way the reference poses it and solved '
                     'by a dense two-phase simplex from scratch), OpenMP over the candidates with per-thread scratch.  Per thread it is about ten times '
                     'the Python reference with HiGHS (about 175 candidates/s per worker process, BASELINE.md 4).  A baseline for orientation: the '
                     'GPU/CPU ratio is not a quality claim',
            'value_one_thread': one, 'thread_scaling': scaling, 'cores_available': cores_available,
            'thread_scaling_sample': f'{n_sub} candidates: every 32nd of the sample below, the same for every thread count',
            'sample': f'{n} candidates ({100 * frac:.1f}% of every BFS level, evenly strided), {dt:.1f} s on {cores} threads, '
                      f'{regions} regions; oracle/mpcombi_oracle.c (C port of the reference algorithm, OpenMP)',
            'regions_per_s': regions / dt,
            'verdicts_differing_from_gpu': differ,
            'note': 'a differing verdict is a candidate whose decision sits on a tolerance (tests/conftest.py::is_knife_edge); '
                    'the parity tests bound their number'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--workload', default='c4', choices=sorted(WORKLOADS))
    ap.add_argument('--cpu-sample', type=int, default=300000, help='candidates in the CPU baseline sample (0 = skip)')
    ap.add_argument('--locate', type=int, default=100000, help='points of the point-location extra (0 = skip)')
    ap.add_argument('--mi', type=int, default=1, help='mixed-integer enumeration extra on a synthetic mpMIQP (0 = skip)')
    ap.add_argument('--complete', type=int, default=1, help='complete solution of the workload by the connected-graph traversal, as an extra (0 = skip)')
    ap.add_argument('--events-in-value', action='store_true', help='one timed region only, with the HIP-event records of the per-kernel accounting inside it (the protocol of rounds 1-4)')
    ap.add_argument('--dist-single', action='store_true', help='run the multi-GPU driver with a process group of one rank (self-test)')
    ap.add_argument('--sweep', action='store_true', help='shape sweep over the target class (n_x <= 20, n_theta <= 10): one JSON line with a row per shape (tools/shape_sweep.py); not the headline')
    ap.add_argument('--deep', type=int, default=0, help='also time the six-level variant of c4 on one GPU (the scaling workload the multi-GPU runs report under extra)')
    args = ap.parse_args()
    if args.sweep:
        sys.path.insert(0, os.path.join(ROOT, 'tools'))
        import shape_sweep
        print(json.dumps(shape_sweep.run()), flush=True)
        return

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    distributed = world > 1 or args.dist_single
    if distributed and args.gpus != world:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
    if args.gpus > 1 and not distributed:
        raise SystemExit('launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 '
                         'bench.py --gpus N ...')

    import torch
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: torch.cuda.is_available() is False (there is no CPU fallback)')
    if os.environ.get('PPOPT_BENCH_BACKEND', 'nccl') != 'nccl':
        local_rank = 0   # self-test: every rank on device 0
    torch.cuda.set_device(local_rank)
    import torch.distributed as dist
    real_stdout = None
    if distributed:
        # stdout carries the ONE JSON line and nothing else: RCCL writes a version banner to fd 1 through C stdio when the first
        # communicator is created -- everything written to fd 1 during the run goes to stderr, the line at the end to the real stdout
        sys.stdout.flush()
        real_stdout = os.dup(1)
        os.dup2(2, 1)
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        backend = os.environ.get('PPOPT_BENCH_BACKEND', 'nccl')   # 'gloo': several ranks on ONE GPU (self-test of this script)
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import __graft_entry__ as entry
    if rank == 0:
        entry.build()
    if distributed:
        dist.barrier()
    from ppopt_amd.mp_solvers import mpqp_hip_combinatorial
    from ppopt_amd.distributed import HipLevelEngine, solve_distributed

    descr, _, max_levels = WORKLOADS[args.workload]
    prog = build_program(args.workload, local_rank)
    engine = HipLevelEngine(prog, local_rank) if distributed else None

    def step(profile):
        if distributed:
            # rank 0 holds the complete Solution (it is the one that reports); the other ranks build only their own shards' objects
            return solve_distributed(engine, prog, profile=profile, max_levels=max_levels, force_shard=args.dist_single, full_solution='rank0')
        return mpqp_hip_combinatorial.solve(prog, device=local_rank, profile=profile, max_levels=max_levels)

    # warm-up solves are held together and released together, so that the allocators (the engine's device buffers, the
    # pooled page-locked result arrays) reach their steady state before the timed region
    warm = [step([] if (args.events_in_value or i % 2) else None) for i in range(args.warmup)]
    del warm
    # Round 6: no gc.freeze() here any more (rounds 4-5 froze torch, the program and the engine so that the collector would not walk them
    # during the timed steps): the process is a user's process.  The product's solve() holds the collector while it creates its region
    # objects and hands them to the oldest generation without a walk (region_batch.gc_paused / _promote_young; MPC_KEEP_GC=1 and
    # MPC_GC_PROMOTE=0 switch that off).
    import gc
    gc.collect()

    def fence():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    # Two timed regions of K steps each over the same program.  The FIRST is the one `value` / `ms_per_step` come from: the solve as a user
    # runs it (no profile asked for, so the library records no HIP events inside its levels: mpc_set_timing).  The SECOND repeats the K
    # steps with a profile per step: the library then brackets every stage and heavy kernel of a level with HIP events on the stream the
    # kernel is launched on (about fourteen records per large level), and the per-kernel durations of `roofline` / `kernel_ms_per_step`
    # are those, live, of this run; that region's own rate is reported beside the first (`ms_per_step_with_kernel_events`): the event
    # records cost 0.1-0.2 ms per solve (markers the queue stops at), which is why they are not in the first.  --events-in-value keeps
    # them in the first region as rounds 1-4 did.
    # Round 6: `value` / `ms_per_step` are timed with Python's cyclic collector ON, exactly as a user's
    # `solve()` runs (the product's solve holds it while it creates its objects: that IS the product's behaviour, tested).  A third
    # region repeats the K steps with the collector off (round 5's headline protocol, `ms_per_step_collector_off`): a solve returns ~10^4
    # region objects, none in a reference cycle, and the young-generation passes over them cost ~0.1 ms per solve of config 4.
    def timed_region(with_profile, collector=True):
        profs, ms_list, last = [], [], None
        fence()
        if not collector:
            gc.disable()
        t0_ = time.perf_counter()
        for _ in range(args.steps):
            pr = [] if with_profile else None
            ts = time.perf_counter()
            last = step(pr)
            ms_list.append(1e3 * (time.perf_counter() - ts))
            profs.append(pr)
        fence()
        el = time.perf_counter() - t0_
        gc.enable()
        if distributed:
            t = torch.tensor([el], dtype=torch.float64, device='cuda')
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, profs, ms_list, last

    elapsed_nogc = None
    if args.events_in_value:
        elapsed, profiles, step_ms, sol = timed_region(True)
        elapsed_events = elapsed
    else:
        elapsed, _, step_ms, sol = timed_region(False)
        del sol
        step(None)      # (hand-over: the region objects of the last untimed-events step are released outside the next region)
        elapsed_events, profiles, _, sol = timed_region(True)
        del sol
        step(None)
        elapsed_nogc, _, _, sol_c = timed_region(False, collector=False)
        del sol_c
        sol = step(None)

    # ---- accounting (identical on every rank; rank 0 reports) ----------------------------------------------
    prof = profiles[-1]
    levels = [p for p in prof if p['depth'] > 0]
    candidates = sum(p['candidates'] for p in prof)
    regions = len(sol.critical_regions)
    steps = max(args.steps, 1)
    all_levels = [p for pr in profiles for p in pr if p['depth'] > 0]
    # per-stage HIP-event times of this rank, summed over all timed steps
    ms = {kname: sum(p.get(kname, 0.0) for p in all_levels) for kname in ('ms_verdict', 'ms_region', 'ms_children')}
    local_cands = sum(p.get('local_candidates', p['candidates']) for p in all_levels)
    n_e = float(numpy.mean([r.E.shape[0] for r in sol.critical_regions])) if regions else 0.0
    rho = regions / max(candidates, 1)
    nx, nt, nc = prog.num_x(), prog.num_t(), prog.num_constraints()
    # SURVEY.md 8(d) figure for the whole path (every kernel of a level together)
    bytes_path = sum(algorithmic_bytes(prog, p['k'], rho, n_e) * p.get('local_candidates', p['candidates']) for p in all_levels)
    ms_path = ms['ms_verdict'] + ms['ms_region'] + ms['ms_children']
    # the three heavy kernels, each timed by its own HIP events inside mpc_level_run (one launch per level):
    #   k_theta2  per candidate: active set (4k) + KKT code (1) + multipliers from k_kkt_thread (8 k (n_t+1)) + status (1);
    #             the shared blocks W, UV (8 n_c (n_c + n_t + 1)) once per launch
    #   k_x2      per candidate: the parent's dictionary record read (+ the record written for the children) + 4 (list) + 4k + 1
    #   k_region2 per optimal candidate: 4k + multipliers + the region record it writes
    kern = {}
    def add(name, key, units_key, bytes_fn):
        tot_ms = sum(p.get(key, 0.0) for p in all_levels)
        launches = sum(1 for p in all_levels if p.get(key, 0.0) > 0.0)
        units = sum(p.get(units_key, 0) for p in all_levels if p.get(key, 0.0) > 0.0)
        nbytes = sum(bytes_fn(p) for p in all_levels if p.get(key, 0.0) > 0.0)
        kern[name] = {'total_ms': tot_ms, 'launches': launches, 'avg_launch_ms': tot_ms / max(launches, 1), 'units': units,
                      'algorithmic_bytes': nbytes, 'achieved_GBs': nbytes / max(tot_ms, 1e-9) / 1e6}
    add('k_theta2', 'ms_theta', 'n_theta_items',
        lambda p: p['n_theta_items'] * (4 * p['k'] + 2 + 8 * p['k'] * (nt + 1)) + 8 * nc * (nc + nt + 1))
    #   round 5: on a level that keeps dictionaries ms_x spans the plan pass (k_xq_thread in plan mode, ms_x_plan), the streamed one-step
    #             dictionaries (k_x1: n_x1 records read and written, ms_x1) and the register simplex k_x2 for what has no plan (the rest)
    for p in all_levels:
        p['ms_x2'] = max(0.0, p.get('ms_x', 0.0) - p.get('ms_x1', 0.0) - p.get('ms_x_plan', 0.0))
        p['n_x2_items'] = max(0, p.get('n_x_items', 0) - p.get('n_x1', 0))
    add('k_x2', 'ms_x2', 'n_x2_items',
        lambda p: p['n_x2_items'] * (p['dict_read_bytes'] + p['dict_write_bytes'] + 4 + 4 * p['k'] + 1))
    add('k_x1', 'ms_x1', 'n_x1', lambda p: p.get('n_x1', 0) * (p['dict_read_bytes'] + p['dict_write_bytes'] + 12))
    R = lambda k: 8 * (nt + 1) * (nx + k + n_e) + 4 * (8 + 2 * k + prog.A_t.shape[0] + 2 * (nc - k))
    add('k_region2', 'ms_region2', 'n_opt', lambda p: p['n_opt'] * (4 * p['k'] + 8 * p['k'] * (nt + 1) + R(p['k'])))
    #   k_kkt_thread  per candidate: active set (4k) in, KKT code + status out, the multipliers (8 k (n_t+1)) of the candidates its
    #             box screen leaves open (= the items of k_theta2); W, UV, A A' once per launch (gathers are served from L2)
    add('k_kkt_thread', 'ms_kkt', 'candidates',
        lambda p: p.get('local_candidates', p['candidates']) * (4 * p['k'] + 2) + p['n_theta_items'] * 8 * p['k'] * (nt + 1)
        + 8 * nc * (2 * nc + nt + 1))
    #   k_xq      (last level) per candidate: list entry, parent slot, last index, status (13) + the integer part of the parent's
    #             record (4 x ints) + its value column (8 x rows) + the new row (8 x cols); per product-form iteration one column
    #             (8 x rows) and one row (8 x cols) of the record: "vectors touched x length"
    #   round 5: most candidates of the quick test are decided by its one-thread first pass k_xq_thread (one byte of the record's position
    #             table, three integers, the row's value, then the hinted column and the value column: 16 x rows, the row-kind mask 16),
    #             the wavefront kernel gets what that pass leaves open.  When the pass ran inside the wavefront kernel's event window
    #             (xq_thread_beside_theta == 0) its time is taken out of ms_xq.
    def xq_wave_ms(p):
        return max(0.0, p.get('ms_xq', 0.0) - (0.0 if p.get('xq_thread_beside_theta') else p.get('ms_xq_thread', 0.0)))
    for p in all_levels:
        p['ms_xq_wave'] = xq_wave_ms(p)
        p['n_xq_wave_items'] = max(0, p.get('n_xq_items', 0) - p.get('n_xq_thread', 0))
    def xq_bytes(p):
        ints, rows, cols = p.get('xq_record', [0, 0, 0])
        return p['n_xq_wave_items'] * (13 + 4 * ints + 8 * rows + 8 * cols) + max(0, p['xq_pivots'] - p.get('n_xq_thread', 0)) * 8 * (rows + cols)
    add('k_xq', 'ms_xq_wave', 'n_xq_wave_items', xq_bytes)
    def xqt_bytes(p):
        ints, rows, cols = p.get('xq_record', [0, 0, 0])
        return (p.get('n_xq_items', 0) if not p.get('ms_x_plan') else p.get('n_x_items', 0)) * (13 + 1 + 12 + 8 + 16 + 16 * rows)   # one test per candidate (tests against other parents come on top: a lower bound)
    add('k_xq_thread', 'ms_xq_thread', 'n_xq_items', xqt_bytes)
    # The roofline object follows SURVEY.md 8(d): achieved = B_alg x candidates/s for the path (all kernels of a level), against
    # the HBM peak.  `dominant_kernel` describes the kernel with the most time ON THE STEP'S CRITICAL PATH (its own algorithmic
    # bytes over its own HIP-event time), every heavy kernel stands under `kernels`.  Critical path: k_region2 runs on the handle's
    # side stream UNDER the (x,theta) stage on the large levels (mpc_level_stats.region_side_stream); there only what it takes beyond
    # that stage's kernels counts.  (Rounds 1-3 named the kernel with the largest total time, k_region2, most of which is hidden.)
    hidden = sum(min(p.get('ms_region2', 0.0), p.get('ms_x', 0.0) + p.get('ms_xq', 0.0)) for p in all_levels if p.get('region_side_stream'))
    hidden_xqt = sum(min(p.get('ms_xq_thread', 0.0), p.get('ms_theta', 0.0)) for p in all_levels if p.get('xq_thread_beside_theta'))   # the pass under the theta stage
    for name, kk in kern.items():
        kk['on_path_ms'] = kk['total_ms'] - (hidden if name == 'k_region2' else (hidden_xqt if name == 'k_xq_thread' else 0.0))
    dominant = max(kern, key=lambda k: kern[k]['on_path_ms'])
    dom = kern[dominant]
    HBM_KERNELS = ('k_x1', 'k_x2', 'k_xq', 'k_xq_thread')      # the kernels whose time is reads / writes of the cached dictionary records
    dominant_hbm = max(HBM_KERNELS, key=lambda k: kern[k]['on_path_ms'])
    # Counter record of the same command (tools/profile_round3.sh -> tools/pmc_round.py -> profiles/r03_pmc.json): HBM-side bytes
    # with the FETCH_SIZE / WRITE_SIZE factors calibrated on known byte counts in each kernel's own access pattern, and the SQ
    # instruction counters per launch.  From the latter the VALU-issue roofline of the simplex kernels: wave-level VALU instructions
    # x 4 cycles (one 64-lane fp64 instruction occupies its 16-lane SIMD for 4 cycles; 78.6 TFLOP/s = 1024 SIMDs x 16 lanes x 2 x
    # 2.4 GHz) / (SIMD-cycles of the launch = live HIP-event time x 2.4 GHz x 1024 SIMDs).  With the F64 instruction counters the
    # fp64 share is priced at 4 cycles and the rest at 2 (v_fma_f32 wave64: 2 cycles, MI355X_MICROARCH.md).
    traffic = dom_traffic = None
    n_simd, clk_hz = 1024, 2.4e9
    tpath = next((q for q in (os.path.join(ROOT, 'profiles', t + '_pmc.json') for t in ('r06', 'r05', 'r04', 'r03')) if os.path.exists(q)), '')
    if tpath and os.path.exists(tpath):
        try:
            rec = json.load(open(tpath))
            tw = rec.get(args.workload)
            if tw:
                traffic = {'bytes_per_step': tw['bytes_per_step'], 'fetch_bytes_per_step': tw['fetch_bytes_per_step'],
                           'write_bytes_per_step': tw['write_bytes_per_step'], 'factors': tw.get('factors'),
                           'calibration': 'tools/calib/pmc_calib.hip (known byte counts; ' + os.path.relpath(tpath, ROOT) + ' -> calibration)',
                           'source': os.path.relpath(tpath, ROOT) + ' (a recorded counter pass of the same command, not this run)'}
                for name, kk in kern.items():
                    pk = tw['kernels'].get(name) or (tw['kernels'].get('k_xq_grouped') if name == 'k_xq' else None)
                    if not pk:
                        continue
                    # counters are sums over every dispatch of the kernel in a solve (small levels included); times are the live
                    # HIP-event times of this run's timed launches (the large levels: they carry the kernel's time)
                    kk['traffic_bytes_per_step'] = pk.get('bytes_per_step')
                    sq = pk.get('sq_per_step')
                    ms_step = kk['total_ms'] / steps
                    if sq and ms_step > 0:
                        simd_cycles = ms_step * 1e-3 * clk_hz * n_simd
                        valu = sq.get('SQ_INSTS_VALU', 0.0)
                        f64 = sum(sq.get(c, 0.0) for c in ('SQ_INSTS_VALU_ADD_F64', 'SQ_INSTS_VALU_MUL_F64', 'SQ_INSTS_VALU_FMA_F64',
                                                           'SQ_INSTS_VALU_TRANS_F64'))
                        kk['valu_insts_per_step'] = valu
                        kk['fp64_insts_per_step'] = f64 if f64 > 0 else None
                        kk['frac_valu_issue_4cyc'] = 4.0 * valu / simd_cycles          # every VALU instruction priced as fp64 (upper bound)
                        kk['frac_fp64_issue'] = (4.0 * f64 / simd_cycles) if f64 > 0 else None
                        kk['frac_valu_issue'] = ((4.0 * f64 + 2.0 * (valu - f64)) / simd_cycles) if f64 > 0 else None   # fp64 at 4 cycles, the rest at 2
                        wc = sq.get('SQ_WAVE_CYCLES', 0.0)
                        if wc > 0:
                            kk['wave_cycles_share'] = {'active_inst': sq.get('SQ_ACTIVE_INST_ANY', 0.0) / wc, 'wait_inst': sq.get('SQ_WAIT_INST_ANY', 0.0) / wc,
                                                       'wait_any': sq.get('SQ_WAIT_ANY', 0.0) / wc}
                dom_traffic = tw['kernels'].get(dominant) or (tw['kernels'].get('k_xq_grouped') if dominant == 'k_xq' else None)
        except (OSError, ValueError, KeyError, TypeError):
            traffic = dom_traffic = None
    b_alg = bytes_path / max(local_cands, 1)
    out = {
        'metric': 'candidate active-sets checked/sec (combinatorial mpQP)',
        'value': candidates * steps / elapsed,
        'unit': 'candidates/s',
        'n_gpus': world,
        'steps': args.steps,
        'warmup': args.warmup,
        'ms_per_step': 1e3 * elapsed / steps,
        'ms_per_step_with_kernel_events': 1e3 * elapsed_events / steps,
        'ms_per_step_collector_off': (1e3 * elapsed_nogc / steps) if elapsed_nogc is not None else None,
        'collector': ('value / ms_per_step: the cyclic collector ON in the process, as a user\'s solve() runs -- solve() itself holds it while it creates its region objects (region_batch.gc_paused; MPC_KEEP_GC=1 leaves it running) -- (round 5 timed them with gc.disable()); '
                      'ms_per_step_collector_off: the same K steps with gc.disable(), round 5\'s protocol'),
        'kernel_timing': ('HIP events inside the one timed region (--events-in-value)' if args.events_in_value else
                          'value / ms_per_step: K timed steps of the solve as a user runs it (no profile: the library records no HIP events inside '
                          'its levels); per-kernel durations (roofline.dominant_kernel, roofline.kernels, kernel_ms_per_step): a second timed region '
                          'of the same K steps with a profile per step = HIP events around every stage and heavy kernel on the stream it is '
                          'launched on; that region runs at ms_per_step_with_kernel_events'),
        'higher_is_better': True,
        'scaling': 'strong',
        'vs_baseline': None,
        'dtype': 'f64',
        'data': 'synthetic',
        'config': {'workload': f'{args.workload}: {descr}', 'n_x': nx, 'n_theta': nt,
                   'n_c': nc, 'n_eq': len(prog.equality_indices), 'n_tc': int(prog.A_t.shape[0]),
                   'candidates_per_step': candidates, 'regions_per_step': regions,
                   # (protocol figures where the driver's parser keeps them: VERDICT r5 item 8a)
                   'ms_per_step_with_kernel_events': 1e3 * elapsed_events / steps,
                   'ms_per_step_collector_off': (1e3 * elapsed_nogc / steps) if elapsed_nogc is not None else None,
                   'value_protocol': 'collector on, no HIP-event records inside the levels',
                   'parallelism': (f'{world} GPU(s): small levels replicated, one split, then subtree-local levels that exchange '
                                   f'only pruned masks') if distributed else 'single GPU'},
        'regions_per_s': regions * steps / elapsed,
        'levels': [{'k': p['k'], 'candidates': p['candidates'], 'status': p['status'], 'regions': p['regions']} for p in levels],
        'kernel_ms_per_step': {k: v / steps for k, v in ms.items()},
        'step_ms': [round(v, 2) for v in step_ms],
        'roofline': {'bound': 'hbm', 'kernel': 'path: every kernel of a level (SURVEY.md 8(d): achieved = B_alg x candidates/s)',
                     'achieved': b_alg * (candidates * steps / elapsed) / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                     'frac': b_alg * (candidates * steps / elapsed) / 1e9 / HBM_PEAK_GBS, 'traffic': traffic,
                     'algorithmic_bytes_per_candidate': b_alg,
                     'achieved_over_kernel_time': bytes_path / max(ms_path, 1e-9) / 1e6,
                     'dominant_kernel': {'kernel': dominant, 'chosen_by': 'time on the critical path of the step (k_region2 under the (x,theta) stage is not)',
                                         'on_path_ms_per_step': {name: kk['on_path_ms'] / steps for name, kk in kern.items()},
                                         'bound': ('hbm (a record streamed in and out per candidate)' if dominant == 'k_x1' else 'hbm (dependent reads of the cached dictionaries)') if dominant in ('k_x1', 'k_x2', 'k_xq', 'k_xq_thread') else 'fp64 VALU issue / dependent latency',
                                         # `frac` is the fraction of the roof that BINDS this kernel: HBM bytes for the streaming kernels; for the register-simplex
                                         # kernels the share of the SIMDs' issue slots their VALU instructions take (fp64 priced at 4 cycles, the rest at 2;
                                         # counters of profiles/r05_pmc.json over this run's HIP-event time) -- their HBM figure stands beside it
                                         'achieved': dom['achieved_GBs'], 'hbm_frac': dom['achieved_GBs'] / HBM_PEAK_GBS,
                                         'frac': (dom['achieved_GBs'] / HBM_PEAK_GBS) if dominant in HBM_KERNELS else (dom.get('frac_valu_issue') or dom.get('frac_valu_issue_4cyc') or dom['achieved_GBs'] / HBM_PEAK_GBS),
                                         'frac_of': 'HBM peak (8 TB/s)' if dominant in HBM_KERNELS else ('VALU issue slots (1024 SIMDs x 2.4 GHz)' if (dom.get('frac_valu_issue') or dom.get('frac_valu_issue_4cyc')) else 'HBM peak (8 TB/s; no instruction counters recorded for this workload)'),
                                         'launches': dom['launches'], 'avg_launch_ms': dom['avg_launch_ms'],
                                         'algorithmic_bytes_per_launch': dom['algorithmic_bytes'] / max(dom['launches'], 1),
                                         'frac_fp64_issue': dom.get('frac_fp64_issue'), 'frac_valu_issue': dom.get('frac_valu_issue'),
                                         'frac_valu_issue_4cyc': dom.get('frac_valu_issue_4cyc'),
                                         'traffic': dom_traffic,
                                         **({'note': 'this program has no register-resident kernels (its parameter set has no vertex after the presolve, DESIGN.md 6c; '
                                                     'MPC_DEBUG_CREATE=1 prints the decision): every stage runs on the LDS-engine kernels k_verdict / k_region, '
                                                     'whose time is kernel_ms_per_step'} if dom['total_ms'] <= 0 else {})},
                     'dominant_hbm_kernel': None if kern[dominant_hbm]['on_path_ms'] <= 0 else {'kernel': dominant_hbm, 'on_path_ms_per_step': kern[dominant_hbm]['on_path_ms'] / steps,
                                             'achieved': kern[dominant_hbm]['achieved_GBs'], 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                                             'frac': kern[dominant_hbm]['achieved_GBs'] / HBM_PEAK_GBS, 'launches': kern[dominant_hbm]['launches'],
                                             'avg_launch_ms': kern[dominant_hbm]['avg_launch_ms'],
                                             'algorithmic_bytes_per_launch': kern[dominant_hbm]['algorithmic_bytes'] / max(kern[dominant_hbm]['launches'], 1),
                                             'traffic_bytes_per_step': kern[dominant_hbm].get('traffic_bytes_per_step'),
                                             'note': 'the HBM-bound kernel with the most time on the critical path (the register-simplex kernels k_region2 / k_theta2 / k_kkt_thread move almost nothing)'},
                     'kernels': kern,
                     'note': 'B_alg = P + 4k + 8 + rho*R per candidate (SURVEY.md 8(d)): the shared problem block P is counted once per '
                             'candidate although it is served from L2, so `achieved` is the figure the survey defines, not measured DRAM '
                             'traffic (that is `traffic`, from the rocprofv3 --pmc passes in profiles/, FETCH_SIZE / WRITE_SIZE with factors '
                             'calibrated on known byte counts per access pattern; null if not collected).  avg_launch_ms are HIP-event times taken inside the library on the stream the kernel '
                             'runs on.  k_region2 and k_theta2 are fp64 simplex pivots in registers (VALU issue / dependent latency, '
                             'almost no HBM traffic); k_x2 streams one cached dictionary per candidate and k_xq reads the vectors of the '
                             'parent\'s dictionary its product-form iterations touch: the two kernels the HBM roof applies to.'},
    }
    # CriticalRegion objects returned by the solve are views into per-level arrays that are cut out on first access;
    # the time to touch every field of every region is reported separately (not part of `value`)
    t_mat = time.perf_counter()
    sol.materialize()                     # batch-wise (Solution.materialize); then every region is touched once more, field by field
    for cr in sol.critical_regions:
        if hasattr(cr, 'materialize'):
            cr.materialize()
    out['materialize_all_regions_ms'] = 1e3 * (time.perf_counter() - t_mat)
    # regions/s both ways: `regions_per_s` counts the lazy CriticalRegion views the solve returns; the materialised figure adds the
    # time to cut every field of every region out of the level arrays to each step
    out['regions_per_s_lazy'] = out['regions_per_s']
    out['regions_per_s_materialised'] = regions / (1e-3 * (out['ms_per_step'] + out['materialize_all_regions_ms']))
    if rank == 0 and not distributed and args.locate > 0 and regions:
        # consumer of the path: point location + evaluation of x*(theta) over the solution, batched on the GPU;
        # beside it the reference's loop (Solution.get_region, numpy per region) on a few points.  Not part of `value`.
        rng = numpy.random.default_rng(0)
        nt = prog.num_t()
        # sample around the Chebyshev centres of random regions so that a good share of the points lies inside the solution
        picks = rng.integers(0, regions, size=args.locate)
        ef, row_off, xlaw = sol._stacked()
        centres = numpy.zeros((64, nt))
        from ppopt_amd.utils.chebyshev_ball import chebyshev_ball
        for j in range(64):
            cr = sol.critical_regions[int(picks[j])]
            cb = chebyshev_ball(cr.E, cr.f, solver=prog.solver)
            centres[j] = cb.sol[:nt] if cb is not None else 0.0
        th = centres[rng.integers(0, 64, size=args.locate)] + 0.05 * rng.standard_normal((args.locate, nt))
        loc = sol.locator(local_rank)
        loc.query(th[:1024])
        tq = time.perf_counter()
        x_b, idx_b = sol.evaluate_batch(th)
        wall = time.perf_counter() - tq
        n_host = 20
        th0 = time.perf_counter()
        host_idx = []
        for p in range(n_host):
            cr = sol.get_region(th[p].reshape(-1, 1))
            host_idx.append(-1 if cr is None else next(i for i, r in enumerate(sol.critical_regions) if r is cr))
        host_dt = time.perf_counter() - th0
        out['point_location'] = {'points': int(args.locate), 'regions': regions, 'rows': int(row_off[-1]),
                                 'located': int((idx_b >= 0).sum()), 'kernel_ms': loc.last_ms,
                                 'points_per_s_kernel': args.locate / max(loc.last_ms, 1e-9) * 1e3,
                                 'points_per_s_wall': args.locate / wall,
                                 'host_loop_points_per_s': n_host / host_dt,
                                 'host_loop_agrees': bool(numpy.array_equal(numpy.array(host_idx), idx_b[:n_host]))}
    if rank == 0 and not distributed and args.mi > 0:
        # caller of the path: mixed-integer enumeration, one continuous sub-program per binary fixation.
        # Synthetic mpMIQP generate_mpmiqp_data(8, 4, 16, n_bin=6, seed=1): 64 feasible fixations, sub-programs 8/4/29.
        # The reference (HiGHS stand-ins, one core, build container) takes 109 s on the 32-fixation sibling
        # (6, 3, 12, n_bin=5, seed=0; tests/golden/mi_rand_6_3_12_b5_s0.npz).  Not part of `value`.
        import warnings
        from ppopt_amd import MPMIQP_Program
        from ppopt_amd.mp_solvers.solve_mpmiqp import solve_mpmiqp
        from ppopt_amd.problem_generator import generate_mpmiqp_data
        dmi = generate_mpmiqp_data(8, 4, 16, 6, 1)
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            pmi = MPMIQP_Program(dmi['A'], dmi['b'], dmi['c'], dmi['H'], dmi['Q'], dmi['A_t'], dmi['b_t'], dmi['F'],
                                 dmi['binary_indices'])
            from ppopt_amd.mp_solvers import mpqp_hip_combinatorial

            def timed(n_rep):
                solve_mpmiqp(pmi)
                best_, n_ = float('inf'), 0
                for _ in range(n_rep):
                    tq = time.perf_counter()
                    smi = solve_mpmiqp(pmi)
                    best_ = min(best_, time.perf_counter() - tq)
                    n_ = len(smi)
                return best_, n_
            best, n_reg = timed(3)          # default: the sub-programs share every launch of a level (mpc_level_run_batch)
            os.environ['MPC_NO_BATCH'] = '1'
            try:
                best_1, n_reg_1 = timed(3)  # one handle per fixation, eight host threads (round 2's form)
            finally:
                del os.environ['MPC_NO_BATCH']
            # the device's share of the batched form: the launches of each shared level, first to last (events)
            subs = [pmi.generate_substituted_problem(f) for f in pmi.feasible_combinations()]
            prof_mi = []
            mpqp_hip_combinatorial.solve_many(subs, device=local_rank, profile=prof_mi)
            for sub in subs:
                sub.release_engine()
        n_fix = len(pmi.feasible_combinations())
        out['mi_enumeration'] = {'workload': 'generate_mpmiqp_data(8,4,16,n_bin=6,seed=1)', 'fixations': n_fix,
                                 'regions': n_reg, 'ms': 1e3 * best, 'sub_programs_per_s': n_fix / best,
                                 'regions_per_s': n_reg / best,
                                 'form': 'several programs per launch (mpc_level_run_batch): every stage of a level is one launch for all sub-programs',
                                 'candidates': int(sum(p['candidates'] for p in prof_mi)),
                                 'device_ms_shared_levels': float(sum(p.get('ms_launches', 0.0) for p in prof_mi)),
                                 'shared_levels': [{'depth': p['depth'], 'members': p['members'], 'members_in_shared_launches': p['shared_launches'],
                                                    'candidates': p['candidates'], 'regions': p['regions'], 'ms_launches': p.get('ms_launches', 0.0)}
                                                   for p in prof_mi],
                                 'one_by_one': {'form': 'MPC_NO_BATCH=1: one handle per fixation, eight host threads', 'ms': 1e3 * best_1,
                                                'regions': n_reg_1, 'sub_programs_per_s': n_fix / best_1}}
    if rank == 0 and not distributed and args.mi > 0:
        # Many small programs at once (tools/many_programs.py): 128 random mpQPs generate_mpqp_data(6, 3, 12, seed) -- each a latency-bound
        # chain of small levels on its own -- one after the other (solve) and together (solve_many: one launch per stage and level for
        # all of them, mpc_level_run_batch).  Same regions either way (tests/test_gpu_batch.py).  Not part of `value`.
        from ppopt_amd import MPQP_Program, problem_generator as pgm
        from ppopt_amd.mp_solvers import mpqp_hip_combinatorial as mhc
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            small = []
            for seed in range(128):
                dd = pgm.generate_mpqp_data(6, 3, 12, 5000 + seed)
                small.append(MPQP_Program(dd['A'], dd['b'], dd['c'], dd['H'], dd['Q'], dd['A_t'], dd['b_t'], dd['F']))
            for ps in small:
                ps.engine(local_rank)
            t_one = t_many = float('inf')
            # each form in its own steady state (three runs each, best of three): alternating them makes every run start from the other
            # form's buffers and pinned blocks
            for _ in range(3):
                tq = time.perf_counter()
                sols_one = [mhc.solve(ps, device=local_rank) for ps in small]
                t_one = min(t_one, time.perf_counter() - tq)
            n_one = [len(sv) for sv in sols_one]
            del sols_one
            sols_many = None
            for _ in range(3):      # (the same repeat count for both forms: ADVICE r5)
                sols_many = None
                tq = time.perf_counter()
                sols_many = mhc.solve_many(small, device=local_rank)
                t_many = min(t_many, time.perf_counter() - tq)
            prof_sm = []
            sols_many = None
            sols_many = mhc.solve_many(small, device=local_rank, profile=prof_sm)      # (with the HIP-event records: device time of the shared launches)
            for ps in small:
                ps.release_engine()
        out['many_programs'] = {'workload': '128 x generate_mpqp_data(6,3,12,seed=5000..5127), complete solves', 'programs': len(small),
                                'regions': int(sum(len(sv) for sv in sols_many)), 'candidates': int(sum(pp['candidates'] for pp in prof_sm)),
                                'same_region_counts': bool(all(a == len(b) for a, b in zip(n_one, sols_many))),
                                'one_by_one_ms': 1e3 * t_one, 'together_ms': 1e3 * t_many, 'speedup': t_one / t_many,
                                'device_ms_shared_levels': float(sum(pp.get('ms_launches', 0.0) for pp in prof_sm)),
                                'programs_per_s_together': len(small) / t_many}
    if rank == 0 and not distributed and args.complete > 0 and args.workload in ('c4', 'c3', 'c2', 'c2x20', 'c5'):
        # The COMPLETE explicit solution of the same program by the connected-graph traversal (mpqp_algorithm.graph, reference
        # mp_solvers/mpqp_graph.py) on the same kernels, wave / visited set / neighbours resident on the device.  Not part of `value`.
        from ppopt_amd.mp_solvers import mpqp_hip_combi_graph
        mpqp_hip_combi_graph.solve_graph(prog, device=local_rank)      # warm-up (buffers, pinned result arrays)
        best, gprof, n_reg = float('inf'), [], 0
        last = None
        for _ in range(4):
            last = None                       # the previous solution's page-locked arrays go back to the pool before the next run
            gp = []
            tq = time.perf_counter()
            gsol = mpqp_hip_combi_graph.solve_graph(prog, device=local_rank, profile=gp)
            dtq = time.perf_counter() - tq
            if dtq < best:
                best, gprof, n_reg = dtq, gp, len(gsol.critical_regions)
            last = gsol
            del gsol
        n_sets = sum(p['candidates'] for p in gprof)
        out['complete_solution'] = {'algorithm': 'graph (connected-graph traversal, device bookkeeping)', 'regions': n_reg,
                                    'active_sets_examined': n_sets, 'waves': len(gprof), 'ms': 1e3 * best,
                                    'active_sets_per_s': n_sets / best, 'regions_per_s': n_reg / best}
        if args.locate > 0:
            # point location over the COMPLETE solution: uniform points of the parameter box, located by walking through
            # adjacent regions (k_locate_walk); the list scan is timed on a small subsample for comparison
            lo_b, hi_b = theta_box(prog)
            rngp = numpy.random.default_rng(1)
            pts = lo_b + rngp.random((max(args.locate, 1000), prog.num_t())) * (hi_b - lo_b)
            locw = last.locator(local_rank)
            last.evaluate_batch(pts[:2048])
            tq = time.perf_counter(); _, idxw = last.evaluate_batch(pts); wallw = time.perf_counter() - tq
            kms = locw.last_ms
            last.use_walk = False
            sub = pts[:2000]
            tq = time.perf_counter(); _, idxs = last.evaluate_batch(sub); walls = time.perf_counter() - tq
            out['complete_solution']['point_location'] = {
                'points': int(len(pts)), 'located': int((idxw >= 0).sum()), 'walk_kernel_ms': kms, 'walk_points_per_s_kernel': len(pts) / max(kms, 1e-9) * 1e3,
                'walk_points_per_s_wall': len(pts) / wallw, 'scan_points_per_s_wall': len(sub) / walls, 'scan_sample': int(len(sub)),
                'walk_equals_scan_on_sample': bool(numpy.array_equal(idxw[:len(sub)], idxs))}
        del last
    if rank == 0 and not distributed and args.cpu_sample > 0:
        # frontiers of every level for the CPU sample: one extra untimed pass
        eng = prog.engine(local_rank)
        eng.pruned_clear()
        eng.frontier_root()
        frontiers, gpu_status = [], []
        for i, p in enumerate(levels):
            frontiers.append(eng.frontier_get())
            gen = i + 1 != len(levels)
            eng.level_run(gen)
            gpu_status.append(eng.level_status())
            if gen:
                eng.frontier_advance()
        out['cpu_baseline'] = cpu_baseline(prog, frontiers, gpu_status, args.cpu_sample)
    else:
        out['cpu_baseline'] = None
    if args.workload == 'c4' and (distributed or args.deep):
        # The scaling workload beside the BASELINE headline: the SAME program one level deeper (levels 1-6, 6.7 M candidates) -- five
        # levels are 6 ms of work, of which the replicated first levels and the fixed cost per level do not shrink with the number of
        # GPUs; the sixth level alone is 16 ms.  Same timing protocol (barrier + synchronise, max over ranks), fewer
        # steps.  Not part of `value`.
        def deep_step(profile):
            if distributed:
                return solve_distributed(engine, prog, profile=profile, max_levels=6, force_shard=args.dist_single, full_solution='rank0')
            return mpqp_hip_combinatorial.solve(prog, device=local_rank, profile=profile, max_levels=6)
        for _ in range(2):
            deep_step([])
        fence()
        td = time.perf_counter()
        n_deep = 3
        for _ in range(n_deep):
            dprof = []
            dsol = deep_step(dprof)
        fence()
        deep_elapsed = time.perf_counter() - td
        if distributed:
            t = torch.tensor([deep_elapsed], dtype=torch.float64, device='cuda')
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            deep_elapsed = float(t.item())
        deep_cands = sum(p['candidates'] for p in dprof)
        out['extra'] = {'scaling_workload': {
            'workload': 'c4, BFS levels 1-6 (the benchmark program one level deeper)', 'candidates_per_step': deep_cands,
            'regions_per_step': len(dsol.critical_regions), 'steps': n_deep, 'ms_per_step': 1e3 * deep_elapsed / n_deep,
            'candidates_per_s': deep_cands * n_deep / deep_elapsed, 'n_gpus': world,
            'levels': [{'k': p['k'], 'candidates': p['candidates'], 'regions': p['regions']} for p in dprof if p['depth'] > 0]}}
        del dsol
    if distributed:
        dist.destroy_process_group()
    sys.stdout.flush()
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)     # C stdio buffers (the RCCL banner) leave before fd 1 is restored
    except OSError:
        pass
    if real_stdout is not None:
        os.dup2(real_stdout, 1)
        os.close(real_stdout)
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
