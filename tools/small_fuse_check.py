"""A/B of the small path's fused form (round 5) against its round-4 form and the classic path on one random program:
    python tools/small_fuse_check.py n_x n_theta m seed
prints per level: candidates, status histogram, regions for MPC_NO_SMALL_FUSE=0/1 and MPC_NO_SMALLPATH=1 (each in its own process)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import sys, json, warnings
sys.path.insert(0, %r)
from ppopt_amd import problem_generator as pg
from ppopt_amd.mpqp_program import MPQP_Program
from ppopt_amd.mp_solvers import mpqp_hip_combinatorial as m
nx, nt, mm, seed = %d, %d, %d, %d
d = pg.generate_mpqp_data(nx, nt, mm, seed)
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    p = MPQP_Program(d['A'], d['b'], d['c'], d['H'], d['Q'], d['A_t'], d['b_t'], d['F'])
pr = []
s = m.solve(p, profile=pr)
print(json.dumps({'regions': len(s.critical_regions), 'levels': [(q['k'], q['candidates'], q['status'], q['regions'], q.get('children')) for q in pr],
                  'sets': sorted(tuple(int(v) for v in r.active_set) for r in s.critical_regions)}))
'''


def run(env, args):
    e = dict(os.environ)
    e.update(env)
    out = subprocess.run([sys.executable, '-c', CHILD % ((ROOT,) + tuple(args))], env=e, capture_output=True, text=True)
    if out.returncode:
        print(out.stderr[-2000:])
        raise SystemExit(1)
    err = [l for l in out.stderr.splitlines() if 'small levels' in l]
    return json.loads(out.stdout.strip().splitlines()[-1]), err


if __name__ == '__main__':
    args = [int(v) for v in sys.argv[1:5]]
    res = {}
    for name, env in (('fused', {'MPC_DEBUG_SMALL': '1'}), ('round4', {'MPC_NO_SMALL_FUSE': '1', 'MPC_DEBUG_SMALL': '1'}), ('classic', {'MPC_NO_SMALLPATH': '1'})):
        res[name], err = run(env, args)
        print(name, res[name]['regions'], err)
        for lv in res[name]['levels']:
            print('   ', lv)
    a, b = set(map(tuple, res['fused']['sets'])), set(map(tuple, res['round4']['sets']))
    print('only in fused', sorted(a - b)[:10], 'only in round4', sorted(b - a)[:10])
