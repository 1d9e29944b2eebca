"""Source-code export of an explicit solution: the generated C++ (compiled with g++ here) and
JavaScript (run with node when present) must locate and evaluate exactly like Solution.get_region / evaluate on the host.
Solutions are rebuilt from golden region sets of the reference, so no device is needed."""
import os
import shutil
import subprocess

import numpy
import pytest

from conftest import golden_regions, load_golden
from ppopt_amd.critical_region import CriticalRegion
from ppopt_amd.solution import Solution
from ppopt_amd.upop.linear_code_gen import export_tables, generate_code_cpp, generate_code_js, generate_code_matlab
from test_mi_host import unpack_regions

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


class _Program:
    def __init__(self, c, H, Q, c_c, c_t, Q_t, n_t):
        self.c, self.H, self.c_c, self.c_t, self.Q_t, self._nt = c, H, c_c, c_t, Q_t, n_t
        if Q is not None:
            self.Q = Q

    def num_t(self):
        return self._nt

    def evaluate_objective(self, x, th):
        v = th.T @ self.H.T @ x + self.c.T @ x + self.c_c + self.c_t.T @ th + 0.5 * th.T @ self.Q_t @ th
        if hasattr(self, 'Q'):
            v = v + 0.5 * x.T @ self.Q @ x
        return float(v[0, 0])


def continuous_solution(name):
    g = load_golden(name)
    n_t = g['proc_F'].shape[1]
    regs = [CriticalRegion(q['A'], q['b'].reshape(-1, 1), q['C'], q['d'].reshape(-1, 1), q['E'], q['f'].reshape(-1, 1), list(k))
            for k, q in sorted(golden_regions(g).items())]
    prog = _Program(g['raw_c'], g['raw_H'], g['raw_Q'] if 'raw_Q' in g.files else None, numpy.zeros((1, 1)),
                    numpy.zeros((n_t, 1)), numpy.zeros((n_t, n_t)), n_t)
    return Solution(prog, regs, is_overlapping=False), g['proc_A_t'], g['proc_b_t']


def mixed_integer_solution(name):
    g = numpy.load(os.path.join(GOLDEN, f'mi_{name}.npz'))
    n_t = g['proc_F'].shape[1]
    prog = _Program(g['proc_c'], g['proc_H'], g['proc_Q'] if 'proc_Q' in g.files else None, g['proc_c_c'], g['proc_c_t'],
                    g['proc_Q_t'], n_t)
    return Solution(prog, unpack_regions(g, 'F_'), is_overlapping=bool(g['F_overlapping'])), g['proc_A_t'], g['proc_b_t']


CASES = [('continuous', 'rand_6_3_12_s1'), ('continuous', 'transport_mpqp'), ('continuous', 'c1_transport_mplp'),
         ('mi', 'mpMILP_market_problem'), ('mi', 'mpMIQP_market_problem'), ('mi', 'bard_mpMILP_adapted_2'),
         ('mi', 'acevedo_mpmilp'), ('mi', 'rand_4_2_8_b3_s1')]


def sample_points(sol, A_t, b_t, n=160, seed=5):
    """Points inside regions (perturbed Chebyshev-free interior guesses: vertices averaged) and points outside."""
    rng = numpy.random.default_rng(seed)
    n_t = sol.program.num_t()
    scale = max(1.0, float(numpy.max(numpy.abs(b_t))))
    pts = [rng.uniform(-1, 1, n_t) * scale * 1.2 for _ in range(n // 2)]
    # walk from random points towards the inside of random regions with a few projection steps
    for _ in range(n - len(pts)):
        r = sol.critical_regions[int(rng.integers(0, len(sol.critical_regions)))]
        E, f = numpy.asarray(r.E, float), numpy.asarray(r.f, float).reshape(-1)
        th = rng.uniform(-1, 1, n_t) * scale
        for _ in range(200):
            viol = E @ th - f
            i = int(numpy.argmax(viol))
            if viol[i] <= -1e-3:
                break
            th = th - (viol[i] + 2e-3) * E[i] / max(float(E[i] @ E[i]), 1e-30)
        pts.append(th)
    return numpy.array(pts)


def expected(sol, pts):
    idx, xs = [], []
    for th in pts:
        cr = sol.get_region(th.reshape(-1, 1))
        idx.append(-1 if cr is None else next(i for i, r in enumerate(sol.critical_regions) if r is cr))
        xs.append(None if cr is None else cr.evaluate(th.reshape(-1, 1)).flatten())
    return idx, xs


MAIN = r'''
#include <cstdio>
#include "solution.hpp"
int main() {
    using namespace ppopt_solution;
    double theta[64], x[256];
    for (;;) {
        for (int t = 0; t < n_theta; ++t) if (std::scanf("%lf", &theta[t]) != 1) return 0;
        const int r = locate(theta);
        std::printf("%d", r);
        if (evaluate(theta, x)) for (int i = 0; i < n_x; ++i) std::printf(" %.17g", x[i]);
        std::printf("\n");
    }
}
'''


@pytest.mark.parametrize('kind,name', CASES, ids=[c[1] for c in CASES])
def test_generated_cpp_locates_and_evaluates_like_the_solution(kind, name, tmp_path):
    sol, A_t, b_t = continuous_solution(name) if kind == 'continuous' else mixed_integer_solution(name)
    pts = sample_points(sol, A_t, b_t)
    idx, xs = expected(sol, pts)
    assert sum(i >= 0 for i in idx) >= 20, 'the sample must hit regions'
    (tmp_path / 'solution.hpp').write_text(generate_code_cpp(sol, float_type='double'))
    (tmp_path / 'main.cpp').write_text(MAIN)
    exe = str(tmp_path / 'a.out')
    subprocess.check_call(['g++', '-O1', '-std=c++11', '-Wall', '-Werror', str(tmp_path / 'main.cpp'), '-o', exe])
    feed = '\n'.join(' '.join(repr(float(v)) for v in th) for th in pts) + '\n'
    out = subprocess.run([exe], input=feed, capture_output=True, text=True, check=True).stdout.strip().splitlines()
    assert len(out) == len(pts)
    for line, want_i, want_x in zip(out, idx, xs):
        tok = line.split()
        assert int(tok[0]) == want_i
        if want_i >= 0:
            numpy.testing.assert_allclose([float(v) for v in tok[1:]], want_x, rtol=1e-9, atol=1e-9)


@pytest.mark.skipif(shutil.which('node') is None, reason='node is not installed')
def test_generated_js_matches(tmp_path):
    sol, A_t, b_t = mixed_integer_solution('mpMIQP_market_problem')
    pts = sample_points(sol, A_t, b_t, n=60)
    idx, xs = expected(sol, pts)
    (tmp_path / 'solution.js').write_text(generate_code_js(sol))
    (tmp_path / 'run.js').write_text(
        "const s = require('./solution.js');\n"
        "const pts = JSON.parse(require('fs').readFileSync(0, 'utf8'));\n"
        "console.log(JSON.stringify(pts.map(p => [s.locate(p), s.evaluate(p)])));\n")
    import json
    res = subprocess.run(['node', str(tmp_path / 'run.js')], input=json.dumps(pts.tolist()), capture_output=True, text=True,
                         check=True, cwd=str(tmp_path))
    got = json.loads(res.stdout)
    for (gi, gx), want_i, want_x in zip(got, idx, xs):
        assert gi == want_i
        if want_i >= 0:
            numpy.testing.assert_allclose(gx, want_x, rtol=1e-9, atol=1e-9)


def test_export_tables_share_hyperplanes_and_matlab_file(tmp_path):
    sol, _, _ = continuous_solution('rand_6_3_12_s1')
    t = export_tables(sol)
    total_rows = sum(numpy.asarray(r.E).shape[0] for r in sol.critical_regions)
    assert len(t['region_plane']) == total_rows and t['region_start'][-1] == total_rows
    assert len(t['plane_offset']) < total_rows            # neighbours share facets
    numpy.testing.assert_allclose(numpy.linalg.norm(t['plane_normal'], axis=1), 1.0, atol=1e-12)
    # every stored (plane, side) reproduces the region's own row
    r0 = sol.critical_regions[0]
    for e, (row, rhs) in enumerate(zip(numpy.asarray(r0.E), numpy.asarray(r0.f).reshape(-1))):
        p, s = t['region_plane'][e], t['region_side'][e]
        nrm = numpy.linalg.norm(row)
        numpy.testing.assert_allclose(s * t['plane_normal'][p], row / nrm, atol=1e-8)
        numpy.testing.assert_allclose(s * t['plane_offset'][p], rhs / nrm, atol=1e-8)
    generate_code_matlab(sol, str(tmp_path))
    text = (tmp_path / 'ppopt_solution.m').read_text()
    assert text.startswith('function [x, region] = ppopt_solution(theta)') and 'plane_normal' in text
    assert generate_code_js(sol).count('function ') == 2


# ---- the reference's own format (uPOP payload), against fixtures produced by the reference itself -------------------------
@pytest.mark.parametrize('name', ['transport_mpqp', 'c1_transport_mplp', 'rand_5_3_8_s3'])
def test_upop_payload_equals_the_reference(name):
    """tests/golden/export_*.npz (oracle/ref_harness/gen_export_goldens.py): for a solution the reference computed, the text its
    generate_code_cpp / generate_code_js paste into the uPOP templates, the find_unique_hyperplanes tables and the .mat
    structure of generate_code_matlab.  ppopt_amd.upop.upop_payload must reproduce all of it exactly -- same names, same order,
    same digits."""
    import scipy.io as sio
    from ppopt_amd.upop import upop_payload as up
    g = numpy.load(os.path.join(GOLDEN, f'export_{name}.npz'))
    n_x, n_t = int(g['n_x']), int(g['n_t'])

    class Prog(_Program):
        def num_x(self):
            return n_x
    packed = golden_regions(g)
    keys = sorted(packed, key=lambda k: (len(k), list(k)))
    regs = []
    for j in g['solution_order'].tolist():      # the reference's solution order
        q = packed[keys[j]]
        regs.append(CriticalRegion(q['A'], q['b'].reshape(-1, 1), q['C'], q['d'].reshape(-1, 1), q['E'], q['f'].reshape(-1, 1), list(keys[j])))
    prog = Prog(g['prog_c'], g['prog_H'], g['prog_Q'] if 'prog_Q' in g.files else None, g['prog_c_c'], g['prog_c_t'], g['prog_Q_t'], n_t)
    sol = Solution(prog, regs, is_overlapping=bool(g['is_overlapping']))
    t = up.upop_tables(sol)
    for key in ('fundamental_c', 'original_c', 'parity_c', 'fundamental_f', 'original_f', 'parity_f'):
        assert t[key] == g['T_' + key].tolist(), key
    assert up.payload_cpp(sol, 'double') == str(g['payload_cpp'])
    assert up.payload_js(sol) == str(g['payload_js'])
    m = up.matlab_struct(sol)
    for key in ('constraint_block', 'constraint_vector', 'function_block', 'function_vec', 'Q', 'H', 'c', 'c_c', 'c_t', 'Q_t'):
        assert numpy.array_equal(numpy.atleast_2d(numpy.asarray(m[key], dtype=float)), numpy.atleast_2d(g['M_' + key])), key
    assert numpy.array_equal(numpy.asarray(m['region_list']).ravel(), g['M_region_list'].ravel())
    assert int(m['num_regions']) == int(g['M_num_regions'].ravel()[0])
    # and the file written by save_matlab loads back to the same structure
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, 's.mat')
        up.save_matlab(sol, path)
        back = sio.loadmat(path)['upop_solution'][0, 0]
        assert set(back.dtype.names) == {k[2:] for k in g.files if k.startswith('M_')}
        assert numpy.array_equal(back['constraint_block'], g['M_constraint_block'])
