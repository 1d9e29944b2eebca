"""Where the time of the multi-GPU driver goes, measured with ONE rank (process group of one, forced sharding): wall time per stage of
solve_distributed summed over a solve of the workload, against the single-GPU loop.   python tools/dist_split.py [c4]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29544')
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
import bench
from ppopt_amd import distributed as D
from ppopt_amd.mp_solvers import mpqp_hip_combinatorial
wl = sys.argv[1] if len(sys.argv) > 1 else 'c4'
prog = bench.build_program(wl, 0); ml = bench.WORKLOADS[wl][2]
eng = D.HipLevelEngine(prog, 0)
acc = {}
def timed(name, fn):
    def w(*a, **k):
        t = time.perf_counter(); r = fn(*a, **k); acc[name] = acc.get(name, 0.0) + time.perf_counter() - t; return r
    return w
for name in ('run', 'run_start', 'run_wait', 'regions_tensors', 'pruned_new', 'add_pruned', 'advance', 'check_base', 'shard', 'base_start'):
    setattr(eng, name, timed('eng.' + name, getattr(eng, name)))
D.allgather_table = timed('allgather_table', D.allgather_table)
D.allgather_rows = timed('allgather_rows', D.allgather_rows)
D.allgather_rows_start = timed('allgather_rows_start', D.allgather_rows_start)
D.to_host = timed('to_host', D.to_host)
for _ in range(3): D.solve_distributed(eng, prog, max_levels=ml, force_shard=True, full_solution='rank0')
acc.clear(); n = 10
t0 = time.perf_counter()
for _ in range(n): sol = D.solve_distributed(eng, prog, max_levels=ml, force_shard=True, full_solution='rank0')
tot = (time.perf_counter() - t0) / n
print(f'solve_distributed (1 rank, forced sharding): {1e3 * tot:.2f} ms per solve, {len(sol.critical_regions)} regions; per stage (ms): '
      + ', '.join(f'{k} {1e3 * v / n:.2f}' for k, v in sorted(acc.items(), key=lambda kv: -kv[1])) + f'; unaccounted {1e3 * (tot - sum(acc.values()) / n):.2f}')
import cProfile, pstats, gc
gc.collect(); gc.freeze()
pr = cProfile.Profile(); pr.enable()
for _ in range(n): D.solve_distributed(eng, prog, max_levels=ml, force_shard=True, full_solution='rank0')
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(14)
for _ in range(3): mpqp_hip_combinatorial.solve(prog, max_levels=ml)
t0 = time.perf_counter()
for _ in range(n): mpqp_hip_combinatorial.solve(prog, max_levels=ml)
print(f'single-GPU loop: {1e3 * (time.perf_counter() - t0) / n:.2f} ms per solve')
sys.stdout.flush()
os._exit(0)      # (torch's process-group teardown aborts in this one-rank set-up; nothing is left to do)
