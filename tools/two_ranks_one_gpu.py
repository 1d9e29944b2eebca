"""Two ranks of the multi-GPU driver on ONE GPU (gloo carries the device tensors): an end-to-end check of the sharded path
with world_size 2 where only a single device is available.  Prints the region count per rank and compares with one rank."""
import os, sys
sys.path.insert(0, '.')
import torch, torch.distributed as dist, torch.multiprocessing as mp


def worker(rank, world, port, wl, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    import bench
    from ppopt_amd.distributed import HipLevelEngine, solve_distributed
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        prog = bench.build_program(wl, 0)
        eng = HipLevelEngine(prog, 0)
        prof = []
        sol = solve_distributed(eng, prog, profile=prof, max_levels=bench.WORKLOADS[wl][2])
        out[rank] = (sorted(tuple(r.active_set) for r in sol.critical_regions),
                     [(p['candidates'], p.get('local_candidates'), p.get('sharded')) for p in prof if p['depth'] > 0])
    finally:
        dist.destroy_process_group()


if __name__ == '__main__':
    wl = sys.argv[1] if len(sys.argv) > 1 else 'c2'
    ctx = mp.get_context('spawn')
    with ctx.Manager() as mgr:
        out = mgr.dict()
        ps = [ctx.Process(target=worker, args=(r, 2, 29611, wl, out)) for r in range(2)]
        [p.start() for p in ps]; [p.join(600) for p in ps]
        res = dict(out)
    import bench
    from ppopt_amd.mp_solvers import mpqp_hip_combinatorial
    solo = mpqp_hip_combinatorial.solve(bench.build_program(wl, 0), max_levels=bench.WORKLOADS[wl][2])
    ref = sorted(tuple(r.active_set) for r in solo.critical_regions)
    for r in sorted(res):
        print('rank', r, 'regions', len(res[r][0]), 'equal to single-rank solve:', res[r][0] == ref, 'levels', res[r][1][:6])
    print('exit codes', [p.exitcode for p in ps])
