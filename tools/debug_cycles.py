"""One solve of a workload under MPC_DEBUG_CYCLES=1 (the library prints a cycle breakdown per level on stderr): python tools/debug_cycles.py [workload]"""
import os, sys
os.environ['MPC_DEBUG_CYCLES'] = '1'
os.environ['MPC_NO_SOLVE_LOOP'] = '1'
sys.path.insert(0, '.')
import bench
from ppopt_amd.mp_solvers import mpqp_hip_combinatorial as m
wl = sys.argv[1] if len(sys.argv) > 1 else 'c4'
prog = bench.build_program(wl)
m.solve(prog, max_levels=bench.WORKLOADS[wl][2])
