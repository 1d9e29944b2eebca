"""Two independent solves of the bench workload at the same time on one GPU (two engines, two threads) against one after the
other: how much of the machine does one solve leave idle?  (run on the GPU box)"""
import sys, time, threading, gc
sys.path.insert(0, '.')
import bench
from ppopt_amd.mp_solvers import mpqp_hip_combinatorial
wl = sys.argv[1] if len(sys.argv) > 1 else 'c4'
ml = bench.WORKLOADS[wl][2]
progs = [bench.build_program(wl) for _ in range(2)]
for p in progs:
    for _ in range(4):
        mpqp_hip_combinatorial.solve(p, max_levels=ml)
gc.collect(); gc.freeze()


def run(p, out, i):
    out[i] = len(mpqp_hip_combinatorial.solve(p, max_levels=ml).critical_regions)


for rep in range(5):
    t = time.perf_counter()
    for p in progs:
        mpqp_hip_combinatorial.solve(p, max_levels=ml)
    seq = time.perf_counter() - t
    out = [0, 0]
    th = [threading.Thread(target=run, args=(progs[i], out, i)) for i in range(2)]
    t = time.perf_counter()
    [x.start() for x in th]; [x.join() for x in th]
    par = time.perf_counter() - t
    print(f'{wl}: two solves one after the other {seq*1e3:.2f} ms, at the same time {par*1e3:.2f} ms ({out})')
