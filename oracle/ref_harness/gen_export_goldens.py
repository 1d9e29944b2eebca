"""Export fixtures: the payload the REAL reference's uPOP exports produce (upop/linear_code_gen.py:21-278) for solutions the
reference itself computed.  Build container only.  Usage:  python oracle/ref_harness/gen_export_goldens.py

Writes tests/golden/export_<name>.npz with: the solution's regions in solution order (inputs), the program's objective terms,
and the expected outputs -- the text the reference pastes into its C++ / JavaScript templates at <==PayloadHere==> (cut out of
the generated file; the templates themselves are NOT stored), the tables of upop_utils.find_unique_region_hyperplanes /
_functions, and the arrays of the .mat structure written by generate_code_matlab."""
import os
import sys
import tempfile
import warnings

import numpy

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_goldens as gg  # noqa: E402  (loads the reference through ref_shims)

from ppopt.mp_solvers import mpqp_combinatorial  # noqa: E402
from ppopt.upop import linear_code_gen as ref_gen  # noqa: E402
from ppopt.upop.lib_upop.upop_cpp_template import cpp_upop  # noqa: E402
from ppopt.upop.upop_utils import find_unique_region_functions, find_unique_region_hyperplanes  # noqa: E402

CASES = {'transport_mpqp': lambda: gg.pg.transport_mpqp_data(), 'c1_transport_mplp': lambda: gg.pg.transport_mplp_data(),
         'rand_5_3_8_s3': lambda: gg.pg.generate_mpqp_data(5, 3, 8, 3)}


def main():
    import scipy.io as sio
    for name, builder in CASES.items():
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            prog = gg.build_reference_program(builder())
            sol = mpqp_combinatorial.solve(prog)
        out = gg.pack_regions(sol.critical_regions, prog.num_x(), prog.num_t())
        # pack_regions sorts by active set: keep the solution's own order as a permutation
        order = sorted(range(len(sol.critical_regions)), key=lambda i: (len(sol.critical_regions[i].active_set), list(sol.critical_regions[i].active_set)))
        out['solution_order'] = numpy.argsort(numpy.array(order)).astype(numpy.int32)   # position in the packed arrays of solution region j
        for key in ('c', 'H', 'c_c', 'c_t', 'Q_t'):
            out['prog_' + key] = numpy.asarray(getattr(prog, key), dtype=numpy.float64)
        if hasattr(prog, 'Q'):
            out['prog_Q'] = prog.Q
        out['n_x'], out['n_t'] = numpy.array(prog.num_x()), numpy.array(prog.num_t())
        out['is_overlapping'] = numpy.array(bool(sol.is_overlapping))
        pre, post = cpp_upop.split('<==PayloadHere==>')
        cpp = ref_gen.generate_code_cpp(sol, 'double')
        assert cpp.startswith(pre) and cpp.endswith(post)
        out['payload_cpp'] = numpy.array(cpp[len(pre):len(cpp) - len(post)])
        js = ref_gen.generate_code_js(sol)
        a = js.index('const region_indices = [')
        b = js.index('var Q_t =[')
        b = js.index('];', b) + 2
        out['payload_js'] = numpy.array(js[a:b])
        fc, oc, pc = find_unique_region_hyperplanes(sol)
        ff, of, pf = find_unique_region_functions(sol)
        for key, val in (('fundamental_c', fc), ('original_c', oc), ('parity_c', pc), ('fundamental_f', ff), ('original_f', of), ('parity_f', pf)):
            out['T_' + key] = numpy.array(val, dtype=numpy.int64)
        with tempfile.TemporaryDirectory() as tmp:
            path = os.path.join(tmp, 'sol.mat')
            ref_gen.generate_code_matlab(sol, path)
            m = sio.loadmat(path)['upop_solution'][0, 0]
            for key in m.dtype.names:
                out['M_' + key] = numpy.asarray(m[key])
        numpy.savez_compressed(os.path.join(gg.GOLDEN, f'export_{name}.npz'), **out)
        print(f'== export_{name}: {len(sol.critical_regions)} regions, cpp payload {len(str(out["payload_cpp"]))} chars, '
              f'{len(fc)} fundamental hyperplanes of {len(oc)}', flush=True)


if __name__ == '__main__':
    main()
