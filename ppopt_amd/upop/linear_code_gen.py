"""Source-code export of an explicit solution (reference: upop/linear_code_gen.py:21-278).

``generate_code_cpp`` / ``generate_code_js`` return one self-contained source text, ``generate_code_matlab`` writes a
``.m`` file: the regions, their control laws and a sequential point-location routine, for deployment where neither
Python nor a GPU exists.  Names and signatures follow the reference; the layout of the generated text is this
package's own (the reference fills templates that are not part of this code base):

  * every hyperplane that bounds some region is stored once -- neighbouring regions share facets with opposite
    orientation -- as a unit normal and an offset; a region is a list of (plane index, side) pairs;
  * the control laws are stored as dense [A | b] blocks (for mixed-integer solutions the law of the full variable
    vector, binaries as constant rows);
  * ``locate(theta)`` returns the first region that contains theta (E theta <= f + tol, ``tol`` =
    ``Solution.point_location_tolerance``) or, for overlapping solutions, the containing region with the lowest
    objective -- the objective of every region is exported as an explicit quadratic in theta; ``evaluate(theta, x)``
    writes x* = A theta + b of that region.

Pure host formatting: nothing here touches the device.
"""
import os
from typing import Dict, List

import numpy

from ..solution import Solution

__all__ = ['generate_code_cpp', 'generate_code_js', 'generate_code_matlab', 'export_tables']


def _full_law(region, n_t: int):
    A = numpy.asarray(region.A, dtype=numpy.float64).reshape(-1, n_t)
    b = numpy.asarray(region.b, dtype=numpy.float64).reshape(-1, 1)
    if region.y_fixation is None:
        return A, b
    n_full = len(region.x_indices) + len(region.y_indices)
    A_full, b_full = numpy.zeros((n_full, n_t)), numpy.zeros((n_full, 1))
    A_full[region.x_indices], b_full[region.x_indices] = A, b
    b_full[region.y_indices, 0] = region.y_fixation
    return A_full, b_full


def _objective_in_theta(program, A: numpy.ndarray, b: numpy.ndarray):
    """(P, q, r) with objective(x*(theta), theta) = 1/2 theta' P theta + q' theta + r for x* = A theta + b
    (mplp_program.py:158-160, mpqp_program.py:44-57)."""
    n_t = A.shape[1]
    Q = getattr(program, 'Q', None)
    H, c = numpy.asarray(program.H, dtype=numpy.float64), numpy.asarray(program.c, dtype=numpy.float64).reshape(-1, 1)
    c_c = float(numpy.asarray(program.c_c).reshape(-1)[0])
    c_t = numpy.asarray(program.c_t, dtype=numpy.float64).reshape(-1, 1)
    Q_t = numpy.asarray(program.Q_t, dtype=numpy.float64)
    P = Q_t + H.T @ A + A.T @ H                   # theta' H' A theta, symmetrised
    q = c_t + H.T @ b + A.T @ c
    r = c_c + float((c.T @ b)[0, 0])
    if Q is not None:
        Q = numpy.asarray(Q, dtype=numpy.float64)
        P = P + A.T @ Q @ A
        q = q + A.T @ Q @ b
        r = r + 0.5 * float((b.T @ Q @ b)[0, 0])
    return 0.5 * (P + P.T), q.reshape(n_t), r


def export_tables(solution: Solution) -> Dict:
    """The numeric content of an export: unique hyperplanes, the (plane, side) lists of the regions, laws and objectives."""
    regions = solution.critical_regions
    n_t = solution.program.num_t()
    planes: List[numpy.ndarray] = []
    index: Dict[tuple, int] = {}
    start, plane_of, side_of = [0], [], []
    laws, objectives = [], []
    n_x = 0
    for region in regions:
        E = numpy.asarray(region.E, dtype=numpy.float64).reshape(-1, n_t)
        f = numpy.asarray(region.f, dtype=numpy.float64).reshape(-1)
        for row, rhs in zip(E, f):
            norm = float(numpy.linalg.norm(row))
            if norm == 0.0:
                continue
            unit = numpy.concatenate([row, [rhs]]) / norm
            lead = unit[numpy.flatnonzero(numpy.abs(unit[:n_t]) > 1e-12)[0]]
            side = 1 if lead > 0 else -1
            key = tuple(numpy.round(side * unit, 9) + 0.0)
            if key not in index:
                index[key] = len(planes)
                planes.append(side * unit)
            plane_of.append(index[key])
            side_of.append(side)
        start.append(len(plane_of))
        A, b = _full_law(region, n_t)
        n_x = A.shape[0]
        laws.append(numpy.hstack([A, b]))
        objectives.append(_objective_in_theta(solution.program, A, b))
    planes_arr = numpy.array(planes).reshape(len(planes), n_t + 1)
    return {'n_theta': n_t, 'n_x': n_x, 'n_regions': len(regions), 'tol': float(solution.point_location_tolerance),
            'overlapping': bool(solution.is_overlapping),
            'plane_normal': planes_arr[:, :n_t], 'plane_offset': planes_arr[:, n_t],
            'region_start': numpy.array(start, dtype=numpy.int64), 'region_plane': numpy.array(plane_of, dtype=numpy.int64),
            'region_side': numpy.array(side_of, dtype=numpy.int64),
            'law': numpy.array(laws).reshape(len(regions), n_x, n_t + 1),
            'obj_P': numpy.array([o[0] for o in objectives]).reshape(len(regions), n_t, n_t),
            'obj_q': numpy.array([o[1] for o in objectives]).reshape(len(regions), n_t),
            'obj_r': numpy.array([o[2] for o in objectives]).reshape(len(regions))}


def _numbers(values, integer: bool = False) -> str:
    flat = numpy.asarray(values).reshape(-1)
    if integer:
        return ', '.join(str(int(v)) for v in flat)
    return ', '.join(repr(float(v)) for v in flat)


_LOCATE_BODY = '''
    int best = -1;
    {real} best_value = 0;
    for (int r = 0; r < n_regions; ++r) {{
        bool inside = true;
        for (int e = region_start[r]; e < region_start[r + 1] && inside; ++e) {{
            const int p = region_plane[e];
            {real} lhs = 0;
            for (int t = 0; t < n_theta; ++t) lhs += plane_normal[p * n_theta + t] * theta[t];
            inside = region_side[e] * (lhs - plane_offset[p]) <= tol;
        }}
        if (!inside) continue;
        if (!overlapping) return r;
        {real} value = obj_r[r];
        for (int i = 0; i < n_theta; ++i) {{
            {real} row = 0;
            for (int j = 0; j < n_theta; ++j) row += obj_P[(r * n_theta + i) * n_theta + j] * theta[j];
            value += theta[i] * (obj_q[r * n_theta + i] + row / 2);
        }}
        if (best < 0 || value <= best_value) {{ best = r; best_value = value; }}
    }}
    return best;'''


def generate_code_cpp(solution: Solution, float_type: str = 'float') -> str:
    """A header-only C++ translation unit (namespace ``ppopt_solution``) with ``locate`` and ``evaluate``."""
    t = export_tables(solution)
    real = float_type
    arr = lambda ctype, name, vals, integer=False: (f'static const {ctype} {name}[] = {{{_numbers(vals, integer) or "0"}}};\n')
    out = ['// explicit solution exported by ppopt_amd.upop.linear_code_gen.generate_code_cpp\n',
           '#pragma once\n', 'namespace ppopt_solution {\n',
           f'typedef {real} real;\n',
           f'static const int n_theta = {t["n_theta"]}, n_x = {t["n_x"]}, n_regions = {t["n_regions"]}, '
           f'n_planes = {len(t["plane_offset"])};\n',
           f'static const bool overlapping = {"true" if t["overlapping"] else "false"};\n',
           f'static const real tol = {t["tol"]!r};\n',
           arr('real', 'plane_normal', t['plane_normal']), arr('real', 'plane_offset', t['plane_offset']),
           arr('int', 'region_start', t['region_start'], True), arr('int', 'region_plane', t['region_plane'], True),
           arr('int', 'region_side', t['region_side'], True), arr('real', 'law', t['law']),
           arr('real', 'obj_P', t['obj_P']), arr('real', 'obj_q', t['obj_q']), arr('real', 'obj_r', t['obj_r']),
           '// index of the region that contains theta (-1: none)\n',
           'inline int locate(const real *theta) {', _LOCATE_BODY.format(real='real'), '\n}\n',
           '// x[0..n_x) = optimal decision at theta; false if theta lies in no region\n',
           'inline bool evaluate(const real *theta, real *x) {\n'
           '    const int r = locate(theta);\n'
           '    if (r < 0) return false;\n'
           '    for (int i = 0; i < n_x; ++i) {\n'
           '        const real *row = law + (r * n_x + i) * (n_theta + 1);\n'
           '        real v = row[n_theta];\n'
           '        for (int t = 0; t < n_theta; ++t) v += row[t] * theta[t];\n'
           '        x[i] = v;\n'
           '    }\n'
           '    return true;\n'
           '}\n',
           '}  // namespace ppopt_solution\n']
    return ''.join(out)


def generate_code_js(solution: Solution) -> str:
    """A JavaScript module text: ``locate(theta)`` -> region index, ``evaluate(theta)`` -> array or null."""
    t = export_tables(solution)
    arr = lambda name, vals, integer=False: f'const {name} = [{_numbers(vals, integer)}];\n'
    body = _LOCATE_BODY.format(real='let').replace('int best', 'let best').replace('for (int ', 'for (let ') \
        .replace('const int p', 'const p').replace('bool inside', 'let inside')
    out = ['// explicit solution exported by ppopt_amd.upop.linear_code_gen.generate_code_js\n',
           f'const n_theta = {t["n_theta"]}, n_x = {t["n_x"]}, n_regions = {t["n_regions"]};\n',
           f'const overlapping = {"true" if t["overlapping"] else "false"};\n', f'const tol = {t["tol"]!r};\n',
           arr('plane_normal', t['plane_normal']), arr('plane_offset', t['plane_offset']),
           arr('region_start', t['region_start'], True), arr('region_plane', t['region_plane'], True),
           arr('region_side', t['region_side'], True), arr('law', t['law']),
           arr('obj_P', t['obj_P']), arr('obj_q', t['obj_q']), arr('obj_r', t['obj_r']),
           'function locate(theta) {', body, '\n}\n',
           'function evaluate(theta) {\n'
           '    const r = locate(theta);\n'
           '    if (r < 0) return null;\n'
           '    const x = new Array(n_x);\n'
           '    for (let i = 0; i < n_x; ++i) {\n'
           '        const o = (r * n_x + i) * (n_theta + 1);\n'
           '        let v = law[o + n_theta];\n'
           '        for (let t = 0; t < n_theta; ++t) v += law[o + t] * theta[t];\n'
           '        x[i] = v;\n'
           '    }\n'
           '    return x;\n'
           '}\n',
           "if (typeof module !== 'undefined') module.exports = {locate, evaluate, n_theta, n_x, n_regions};\n"]
    return ''.join(out)


def generate_code_matlab(solution: Solution, path: str = '') -> None:
    """Writes ``ppopt_solution.m`` (a function file: ``[x, region] = ppopt_solution(theta)``) into ``path``."""
    t = export_tables(solution)
    mat = lambda name, vals, cols: (f'{name} = reshape([{_numbers(vals).replace(",", "")}], {cols}, [])\';\n')
    lines = ['function [x, region] = ppopt_solution(theta)\n',
             '% explicit solution exported by ppopt_amd.upop.linear_code_gen.generate_code_matlab\n',
             '% theta: column vector; x: optimal decision ([] if theta lies in no region); region: 1-based index (0: none)\n',
             f'n_theta = {t["n_theta"]}; n_x = {t["n_x"]}; n_regions = {t["n_regions"]}; tol = {t["tol"]!r};\n',
             f'overlapping = {"true" if t["overlapping"] else "false"};\n',
             mat('plane_normal', t['plane_normal'], t['n_theta']),
             f'plane_offset = [{_numbers(t["plane_offset"]).replace(",", "")}]\';\n',
             f'region_start = [{_numbers(t["region_start"], True).replace(",", "")}];\n',
             f'region_plane = [{_numbers(t["region_plane"] + 1, True).replace(",", "")}];\n',
             f'region_side = [{_numbers(t["region_side"], True).replace(",", "")}];\n',
             mat('law', t['law'], t['n_theta'] + 1),
             mat('obj_P', t['obj_P'], t['n_theta']), mat('obj_q', t['obj_q'], t['n_theta']),
             f'obj_r = [{_numbers(t["obj_r"]).replace(",", "")}];\n',
             'region = 0; best = inf; x = [];\n',
             'for r = 1:n_regions\n',
             '    e = region_start(r) + 1 : region_start(r + 1);\n',
             '    p = region_plane(e);\n',
             '    if all(region_side(e)\' .* (plane_normal(p, :) * theta - plane_offset(p)) <= tol)\n',
             '        if ~overlapping, region = r; break; end\n',
             '        P = obj_P((r - 1) * n_theta + 1 : r * n_theta, :);\n',
             '        value = 0.5 * theta\' * P * theta + obj_q(r, :) * theta + obj_r(r);\n',
             '        if value <= best, best = value; region = r; end\n',
             '    end\n',
             'end\n',
             'if region > 0\n',
             '    L = law((region - 1) * n_x + 1 : region * n_x, :);\n',
             '    x = L(:, 1:n_theta) * theta + L(:, n_theta + 1);\n',
             'end\n',
             'end\n']
    with open(os.path.join(path, 'ppopt_solution.m'), 'w') as fh:
        fh.write(''.join(lines))
