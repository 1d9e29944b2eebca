"""The data payload of the reference's uPOP exports, in the reference's own format.

The reference's ``generate_code_cpp`` / ``generate_code_js`` (upop/linear_code_gen.py:21-229) paste a block of data
declarations into a C++ / JavaScript runtime template (``upop/lib_upop/upop_cpp_template.py``, ``upop_js_template.py``) at the
marker ``<==PayloadHere==>``; ``generate_code_matlab`` (:232-278) saves a ``.mat`` structure.  The templates are the reference's
runtime and are not part of this code base; what a solution contributes -- and what a user who already deploys uPOP needs from
this package -- is the payload.  This module produces it line for line as the reference does:

  * ``payload_cpp(solution, float_type)`` / ``payload_js(solution)``: the text that replaces the marker (same names, same order,
    same number formatting: ``str`` of Python ints and floats), so ``template.replace('<==PayloadHere==>', payload)`` gives the
    reference's file;
  * ``matlab_struct(solution)`` / ``save_matlab(solution, path)``: the ``upop_solution`` structure with the reference's keys.

The tables behind them restate upop/upop_utils.py:11-70 (``find_unique_hyperplanes``: rows [E | f] and [A | b] reduced to the
fundamental hyperplanes / functions, an index back from every row, and the side of the plane the row is on) and :210-219
(``get_descriptions``); mixed-integer regions first get the law of the full variable vector (``convert_mi_critical_region``,
:86-107).  tests/test_export.py compares payloads and tables with fixtures produced by the reference itself
(oracle/ref_harness/gen_export_goldens.py).  Pure host formatting; the package's own self-contained exports are in
linear_code_gen.py.
"""
from typing import Dict, List, Tuple

import numpy

from ..solution import Solution
from .linear_code_gen import _full_law


def find_unique_hyperplanes(overall: numpy.ndarray) -> Tuple[List[int], List[int], List[int]]:
    """(indices of the rows that introduce a new hyperplane, per row the position of its hyperplane in that list, per row +1 / -1
    for the side).  Rows are identified through their entries scaled by 1e9 and truncated to integers; a row is the same
    hyperplane as an earlier one if it equals it or its negative in that representation (upop_utils.py:40-67)."""
    overall = numpy.asarray(overall, dtype=numpy.float64)
    pos = (overall * 1000000000).astype(numpy.int64)
    neg = (overall * -1000000000).astype(numpy.int64)
    key_p = [row.tobytes() for row in pos]
    key_n = [row.tobytes() for row in neg]
    where: Dict[bytes, int] = {}
    fundamental: List[int] = []
    for i, (kp, kn) in enumerate(zip(key_p, key_n)):
        if kp not in where and kn not in where:
            where[kp] = len(fundamental)
            fundamental.append(i)
    index, parity = [], []
    for kp, kn in zip(key_p, key_n):
        if kp in where:
            index.append(where[kp])
            parity.append(1)
        else:
            index.append(where[kn])
            parity.append(-1)
    return fundamental, index, parity


def upop_tables(solution: Solution) -> Dict:
    """Everything the three exports are made of, as arrays and lists."""
    regions = solution.critical_regions
    n_t = solution.program.num_t()
    E = numpy.vstack([numpy.asarray(r.E, dtype=numpy.float64).reshape(-1, n_t) for r in regions])
    f = numpy.vstack([numpy.asarray(r.f, dtype=numpy.float64).reshape(-1, 1) for r in regions])
    laws = [_full_law(r, n_t) for r in regions]
    A = numpy.vstack([law[0] for law in laws])
    b = numpy.vstack([law[1] for law in laws])
    fund_c, orig_c, par_c = find_unique_hyperplanes(numpy.hstack([E, f]))
    fund_f, orig_f, par_f = find_unique_hyperplanes(numpy.hstack([A, b]))
    bounds = [0]
    for r in regions:
        bounds.append(bounds[-1] + numpy.asarray(r.E).reshape(-1, n_t).shape[0])
    prog = solution.program
    x_dim = prog.num_x() if hasattr(prog, 'num_x') else A.shape[0] // max(len(regions), 1)
    return {'region_boundary_index': bounds, 'fundamental_c': fund_c, 'original_c': orig_c, 'parity_c': par_c,
            'fundamental_f': fund_f, 'original_f': orig_f, 'parity_f': par_f, 'E': E, 'f': f, 'A': A, 'b': b,
            'theta_dim': n_t, 'x_dim': x_dim, 'num_constraints': E.shape[0], 'num_functions': A.shape[0],
            'num_regions': len(regions), 'is_overlapping': bool(solution.is_overlapping), 'has_Q': hasattr(prog, 'Q')}


def _csv(values) -> str:
    return ','.join(str(v) for v in values)


def _flat(a) -> list:
    return numpy.asarray(a).flatten().tolist()


def payload_cpp(solution: Solution, float_type: str = 'float') -> str:
    """The declarations linear_code_gen.py:47-132 joins with newlines and pastes into the C++ template."""
    t = upop_tables(solution)
    prog = solution.program
    arr = lambda data, name, vartype: f'const {vartype} {name} [{len(data)}] = ' + '{' + _csv(data) + '};'
    var = lambda data, name, vartype: f'const {vartype} {name} = {data!s};'
    tf = {True: 'true', False: 'false'}
    bits = lambda parity: ''.join('1' if p == 1 else '0' for p in parity[::-1])
    lines = [f'typedef {float_type} float_;',
             arr(t['region_boundary_index'], 'region_indicies', 'uint16_t'), '',
             arr(t['original_c'], 'constraint_indices', 'uint16_t'),
             f'const std::bitset<{len(t["parity_c"])}> constraint_parity("{bits(t["parity_c"])}");', '',
             arr(t['original_f'], 'function_indices', 'uint16_t'),
             f'const std::bitset<{len(t["parity_f"])}> function_parity("{bits(t["parity_f"])}");',
             f'const bool solution_overlap = {tf[t["is_overlapping"]]};',
             f'const bool is_qp = {tf[t["has_Q"]]};',
             var(t['theta_dim'], 'theta_dim', 'int'), var(t['x_dim'], 'x_dim', 'int'),
             var(t['num_constraints'], 'num_hyperplanes', 'int'), var(t['num_functions'], 'num_functions', 'int'),
             var(t['num_regions'], 'num_regions', 'int'),
             var(len(t['fundamental_c']), 'num_fundamental_hyper_planes', 'int'),
             arr(_flat(t['E'][t['fundamental_c']]), 'constraint_matrix_data', float_type),
             arr(_flat(t['f'][t['fundamental_c']]), 'constraint_vector_data', float_type),
             arr(_flat(t['A'][t['fundamental_f']]), 'function_matrix_data', float_type),
             arr(_flat(t['b'][t['fundamental_f']]), 'function_vector_data', float_type)]
    if t['has_Q']:
        lines.append('const std::array<float_, x_dim*x_dim> Q ={' + _csv(_flat(prog.Q)) + '};')
    else:
        lines.append('const std::array<float_, 1> Q = {1};')
    lines.append('const std::array<float_, x_dim> c ={' + _csv(_flat(prog.c)) + '};')
    lines.append('const std::array<float_, x_dim*theta_dim> H ={' + _csv(_flat(prog.H)) + '};')
    lines.append(f'const float_ c_c = {_flat(prog.c_c)[0]};')
    lines.append('const std::array<float_, theta_dim> c_t ={' + _csv(_flat(prog.c_t)) + '};')
    lines.append('const std::array<float_, theta_dim*theta_dim> Q_t ={' + _csv(_flat(prog.Q_t)) + '};')
    return '\n'.join(lines)


def payload_js(solution: Solution) -> str:
    """The declarations linear_code_gen.py:155-226 pastes into the JavaScript template."""
    t = upop_tables(solution)
    prog = solution.program
    arr = lambda data, name: f'const {name} = [' + _csv(data) + '];'
    var = lambda data, name: f'const {name} = {data!s};'
    tf = {True: 'true', False: 'false'}
    lines = [arr(t['region_boundary_index'], 'region_indices'), 'var NOT_IN_FEASIBLE_SPACE = -1;',
             arr(t['original_c'], 'constraint_indices'), arr([tf[p == 1] for p in t['parity_c']], 'constraint_parity'),
             arr(t['original_f'], 'function_indices'), arr([tf[p == 1] for p in t['parity_f']], 'function_parity'),
             f'var solution_overlap = {tf[t["is_overlapping"]]};',
             var(t['theta_dim'], 'theta_dim'), var(t['x_dim'], 'x_dim'), var(t['num_constraints'], 'num_hyperplanes'),
             var(t['num_functions'], 'num_functions'), var(t['num_regions'], 'num_regions'),
             var(len(t['fundamental_c']), 'num_fundamental_hyper_planes'),
             arr(_flat(t['E'][t['fundamental_c']]), 'constraint_matrix_data'),
             arr(_flat(t['f'][t['fundamental_c']]), 'constraint_vector_data'),
             arr(_flat(t['A'][t['fundamental_f']]), 'function_matrix_data'),
             arr(_flat(t['b'][t['fundamental_f']]), 'function_vector_data'),
             ('var Q =[' + _csv(_flat(prog.Q)) + '];') if t['has_Q'] else 'var Q = [1];',
             'var c =[' + _csv(_flat(prog.c)) + '];', 'var H =[' + _csv(_flat(prog.H)) + '];',
             f'var c_c = {_flat(prog.c_c)[0]};', 'var c_t =[' + _csv(_flat(prog.c_t)) + '];',
             'var Q_t =[' + _csv(_flat(prog.Q_t)) + '];']
    return '\n'.join(lines)


def matlab_struct(solution: Solution) -> Dict:
    """The ``upop_solution`` structure of linear_code_gen.py:245-276 (1-based region_list, Q = 0 for an mpLP)."""
    t = upop_tables(solution)
    p = solution.program
    x_dim = t['x_dim']
    Q = numpy.asarray(p.Q) if t['has_Q'] else 0.0 * numpy.eye(x_dim)
    return {'constraint_block': t['E'], 'constraint_vector': t['f'], 'function_block': t['A'], 'function_vec': t['b'],
            'region_list': numpy.array(t['region_boundary_index']) + 1, 'num_regions': t['num_regions'], 'Q': Q, 'H': p.H,
            'c': p.c, 'c_c': p.c_c, 'c_t': p.c_t, 'Q_t': p.Q_t}


def save_matlab(solution: Solution, path: str) -> None:
    import scipy.io as sio
    sio.savemat(path, {'upop_solution': matlab_struct(solution)})
