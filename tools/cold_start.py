"""Where the first solves of a fresh process spend their time (run on the GPU box): python tools/cold_start.py [c4]"""
import sys, time
t00 = time.perf_counter()
sys.path.insert(0, '.')
import bench
from ppopt_amd.mp_solvers import mpqp_hip_combinatorial
wl = sys.argv[1] if len(sys.argv) > 1 else 'c4'
t0 = time.perf_counter()
prog = bench.build_program(wl)
t1 = time.perf_counter()
eng = prog.engine(0)
t2 = time.perf_counter()
print(f'imports {1e3*(t0-t00):.0f} ms, program (presolve on the device) {1e3*(t1-t0):.0f} ms, engine {1e3*(t2-t1):.1f} ms')
ml = bench.WORKLOADS[wl][2]
for i in range(5):
    prof = []
    t = time.perf_counter(); s = mpqp_hip_combinatorial.solve(prog, max_levels=ml, profile=prof); dt = time.perf_counter() - t
    print(f'solve {i}: {dt*1e3:.1f} ms | levels', [round(p['ms_wall'], 2) for p in prof if p.get('depth', 0) > 0])
