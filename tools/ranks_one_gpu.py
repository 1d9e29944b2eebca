"""N ranks of the multi-GPU driver on ONE GPU (gloo carries the device tensors): the sharded path end to end with world sizes
2, 4 and 8 where only a single device is available.  For every world size: each rank's region set against the single-rank
solve, and per level the ranks' shard sizes (candidates, regions, children) -- the imbalance of the static subtree ownership
that follows the one split.  Times are NOT multi-GPU times (the ranks share one device).

    python tools/ranks_one_gpu.py [workload=c4] [world sizes, default 2 4 8]   -> gpurun_out/r6/ranks_<workload>.json   (workloads: bench.py's, c4x6, qt6, di8x20)
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def workload(wl):
    """(program, max_levels): bench.py's workloads, the six-level variant of config 4 (c4x6) and two further MPC programs."""
    import bench
    from ppopt_amd import problem_generator as pg
    extra = {'c4x6': (lambda: pg.generate_mpqp_data(20, 8, 20, 0), 6),
             'qt6': (lambda: pg.quad_tank_data(6), None),                                 # quad tank, horizon 6: n_x = 12, n_theta = 4
             'di8x20': (lambda: pg.double_integrator_data(8, x_bound=20.0), None)}          # double integrator, horizon 8, wide state box
    if wl in extra:
        return bench.program_from_data(extra[wl][0](), 0), extra[wl][1]
    return bench.build_program(wl, 0), bench.WORKLOADS[wl][2]


def worker(rank, world, port, wl, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import bench
    from ppopt_amd.distributed import HipLevelEngine, solve_distributed
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        prog, ml = workload(wl)
        eng = HipLevelEngine(prog, 0)
        prof = []
        sol = solve_distributed(eng, prog, profile=prof, max_levels=ml)
        out[rank] = (sorted(tuple(r.active_set) for r in sol.critical_regions),
                     [{'k': p['k'], 'candidates': p['candidates'], 'local': p.get('local_candidates'), 'sharded': bool(p.get('sharded')),
                       'regions': p['regions'], 'children': p['children'], 'ms_kernels': p.get('ms_verdict', 0) + p.get('ms_region', 0) + p.get('ms_children', 0),
                       # round 6: what the one-thread pass of the (x,theta) question decided on THIS rank -- after a shard the
                       # look-up of a candidate's other parents sees only the rank's own part of the previous frontier
                       'xq_items': int(p.get('n_xq_items', 0) or 0), 'xq_thread': int(p.get('n_xq_thread', 0) or 0),
                       'x_items': int(p.get('n_x_items', 0) or 0), 'x1': int(p.get('n_x1', 0) or 0)}
                      for p in prof if p['depth'] > 0])
        eng.close()
    finally:
        dist.destroy_process_group()


if __name__ == '__main__':
    wl = sys.argv[1] if len(sys.argv) > 1 else 'c4'
    worlds = [int(v) for v in sys.argv[2:]] or [2, 4, 8]
    import bench
    from ppopt_amd.mp_solvers import mpqp_hip_combinatorial
    prog0, ml0 = workload(wl)
    solo_prof = []
    solo = mpqp_hip_combinatorial.solve(prog0, max_levels=ml0, profile=solo_prof)
    solo_rates = {p['k']: {'thread_pass_hit_rate': round(p['n_xq_thread'] / p['n_xq_items'], 4) if p.get('n_xq_items') else None,
                           'one_step_plan_rate': round(p['n_x1'] / p['n_x_items'], 4) if p.get('n_x_items') else None} for p in solo_prof if p['depth'] > 0}
    ref = sorted(tuple(r.active_set) for r in solo.critical_regions)
    report = {'workload': wl, 'regions_single_rank': len(ref), 'worlds': {}}
    ctx = mp.get_context('spawn')          # fresh children: a process that has touched the GPU is never re-executed
    for world in worlds:
        with ctx.Manager() as mgr:
            out = mgr.dict()
            ps = [ctx.Process(target=worker, args=(r, world, 29611 + world, wl, out)) for r in range(world)]
            [p.start() for p in ps]
            [p.join(900) for p in ps]
            res = dict(out)
        ok = len(res) == world and all(res[r][0] == ref for r in res)
        levels = []
        n_lev = len(res[0][1]) if res else 0
        for i in range(n_lev):
            loc = [res[r][1][i]['local'] or res[r][1][i]['candidates'] for r in sorted(res)]
            ms = [res[r][1][i]['ms_kernels'] for r in sorted(res)]
            levels.append({'k': res[0][1][i]['k'], 'candidates': res[0][1][i]['candidates'], 'sharded': res[0][1][i]['sharded'], 'shards': loc,
                           'imbalance_max_over_mean': max(loc) / (sum(loc) / len(loc)) if res[0][1][i]['sharded'] else 1.0,
                           'kernel_ms_per_rank_sharing_one_gpu': [round(v, 3) for v in ms],
                           # share of the quick test's candidates the thread pass decided (last level) / of the dictionaries built by a one-step plan
                           # (storing levels), per rank; `single_rank` = the same figure of the one-rank solve
                           'thread_pass_hit_rate_per_rank': [round(res[r][1][i]['xq_thread'] / res[r][1][i]['xq_items'], 4) if res[r][1][i]['xq_items'] else None for r in sorted(res)],
                           'one_step_plan_rate_per_rank': [round(res[r][1][i]['x1'] / res[r][1][i]['x_items'], 4) if res[r][1][i]['x_items'] else None for r in sorted(res)],
                           'single_rank': solo_rates.get(res[0][1][i]['k'])})
        report['worlds'][str(world)] = {'all_ranks_equal_single_rank_solve': ok, 'exit_codes': [p.exitcode for p in ps], 'levels': levels}
        print(f'world {world}: every rank returns the single-rank region set: {ok}; shards per level: '
              + '; '.join(f"k={lv['k']}: {lv['shards']} (max/mean {lv['imbalance_max_over_mean']:.3f})" for lv in levels if lv['sharded']), flush=True)
    os.makedirs('gpurun_out/r6', exist_ok=True)
    report['reshard_threshold'] = float(os.environ.get('MPC_RESHARD', '0') or 0)
    json.dump(report, open(f'gpurun_out/r6/ranks_{wl}' + ('_reshard' if report['reshard_threshold'] > 0 else '') + '.json', 'w'), indent=1)
