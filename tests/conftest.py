"""Shared test helpers.  GPU tests are marked ``@pytest.mark.gpu``; everything else runs on CPU.

The CPU oracle (oracle/) is the checker here and nowhere else.  Golden vectors under tests/golden/ were produced
by importing the real reference (oracle/ref_harness/gen_goldens.py); tests never read /root/reference.
"""
import os
import sys

import numpy
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    # A test that hangs (a kernel that never ends, a lost wake-up) must FAIL with the stacks of all threads instead of stalling the
    # run: pytest-timeout, when it is installed and the command line does not set its own limit.  The longest tests take ~20 s
    # on the GPU box and ~2 min in the build container (the C-ABI layout test compiles).
    # (applied per item below, through the plugin's own marker)


def pytest_collection_modifyitems(config, items):
    if not config.pluginmanager.hasplugin('timeout') or getattr(config.option, 'timeout', None):
        return
    for item in items:
        if item.get_closest_marker('timeout') is None:
            item.add_marker(pytest.mark.timeout(900, method='thread'))


# ---- knife-edge exceptions: explicit, per golden, and counted ----------------------------------------------------------
# north_star: region count and active-set indices bit-exact.  A parity test may accept a difference from the reference ONLY
# for an active set that is listed here by name AND is knife-edge by the rule of is_knife_edge (the oracle's own verdict flips
# when its LP tolerance moves two decades, or cond(KKT) is beyond the limit).  Every exception a run consumes is recorded and
# written to gpurun_out/parity_exceptions.json (and printed in the terminal summary), so the record shows how many were used.
KNIFE_EDGE_VERDICTS = {
    # golden name: {active set: reason}
}
KNIFE_EDGE_REGIONS = {
    # golden name: {active set: reason}   (region present on one side only)
}
KNIFE_EDGE_FACETS = {
    # golden name: {active set: reason}   (same region, omega / lambda / regular sets differ in rows redundant within the LP tolerance)
    'rand_6_3_12_s1': {(0, 1, 3, 4, 5, 6): 'sliver region, cond(KKT) 1e9: its facet LPs sit on the 1e-7 tolerance in the reference run (min slack '
                                           '3e-8 .. 1e-7); the same region is excepted in tests/test_oracle_goldens.py (the CPU oracle differs there too)'},
}
KNIFE_EDGE_RANK = {
    # case of tests/golden/rank_cases.npz: reason   (device verdict "rank deficient / infeasible" differs from the reference's check_feasibility)
    'near_n5_eps1e-10': 'full rank by the SVD rule on both sides; the reference\'s feasibility LP (HiGHS) then calls the nearly dependent equality '
                        'system infeasible (rows 5e-11 apart, right-hand sides equal to 1e-16): the device\'s LP finds the point',
}
_CONSUMED = []


def consume_exception(kind, golden, active_set, detail=''):
    table = {'verdict': KNIFE_EDGE_VERDICTS, 'region': KNIFE_EDGE_REGIONS, 'facets': KNIFE_EDGE_FACETS}[kind]
    key = tuple(int(v) for v in active_set)
    listed = key in table.get(golden, {})
    if listed:
        _CONSUMED.append({'kind': kind, 'golden': golden, 'active_set': list(key), 'detail': detail})
    return listed


def _exception_report():
    listed = {kind: {g: len(v) for g, v in table.items()} for kind, table in
              (('verdict', KNIFE_EDGE_VERDICTS), ('region', KNIFE_EDGE_REGIONS), ('facets', KNIFE_EDGE_FACETS))}
    return {'listed': listed, 'consumed': _CONSUMED, 'n_consumed': len(_CONSUMED)}


def pytest_sessionfinish(session, exitstatus):
    import json
    try:
        out = os.path.join(ROOT, 'gpurun_out')
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, 'parity_exceptions.json'), 'w') as fh:
            json.dump(_exception_report(), fh, indent=1)
    except OSError:
        pass


def pytest_terminal_summary(terminalreporter):
    rep = _exception_report()
    terminalreporter.write_line(f"parity exceptions consumed: {rep['n_consumed']} "
                                f"(listed: {sum(sum(v.values()) for v in rep['listed'].values())}) -> gpurun_out/parity_exceptions.json")


def load_golden(name):
    return numpy.load(os.path.join(GOLDEN, name + '.npz'))


def golden_regions(g):
    """{active-set tuple: dict of fields} from the packed R_* arrays of a golden file."""
    out = {}
    for i in range(len(g['R_k'])):
        k, ne = int(g['R_k'][i]), int(g['R_nE'][i])
        unpad = lambda a: [int(v) for v in a if v >= 0]
        out[tuple(int(v) for v in g['R_active'][i][:k])] = {
            'A': g['R_A'][i], 'b': g['R_b'][i], 'C': g['R_C'][i][:k], 'd': g['R_d'][i][:k],
            'E': g['R_E'][i][:ne], 'f': g['R_f'][i][:ne], 'omega_set': unpad(g['R_omega'][i]),
            'lambda_set': unpad(g['R_lambda'][i]), 'regular_set': [unpad(g['R_regular_idx'][i]),
                                                                   unpad(g['R_regular_con'][i])]}
    return out


def rel_err(a, b):
    a, b = numpy.asarray(a, float).ravel(), numpy.asarray(b, float).ravel()
    if a.size == 0 and b.size == 0:
        return 0.0
    return float(numpy.max(numpy.abs(a - b) / (1.0 + numpy.abs(b))))


def rows_match(E1, f1, E2, f2, tol=1e-8):
    """Set equality of the rows [E | f] under a tolerance (order and exact duplicates ignored)."""
    R1 = numpy.hstack([numpy.asarray(E1), numpy.asarray(f1).reshape(-1, 1)])
    R2 = numpy.hstack([numpy.asarray(E2), numpy.asarray(f2).reshape(-1, 1)])
    def covered(X, Y):
        return all(numpy.min(numpy.max(numpy.abs(Y - x) / (1.0 + numpy.abs(x)), axis=1)) <= tol for x in X)
    if len(R1) == 0 or len(R2) == 0:
        return len(R1) == len(R2)
    return covered(R1, R2) and covered(R2, R1)


def kkt_condition(P, active_set):
    """cond of the KKT matrix of mpqp_program.py:182 for a presolved problem object with .A/.Q."""
    a = list(active_set)
    A_hat = P.A[a]
    k = len(a)
    if not getattr(P, 'is_qp', True):
        return float(numpy.linalg.cond(A_hat)) if k == P.n_x else float('inf')
    M = numpy.block([[A_hat, numpy.zeros((k, k))], [P.Q, A_hat.T]])
    return float(numpy.linalg.cond(M))


def is_knife_edge(P, active_set, cond_limit=1e8):
    """A verdict is 'knife-edge' when it is not a property of the problem but of a tolerance: the oracle's own verdict
    changes when its 1e-7 feasibility tolerance moves by two decades either way, or the KKT matrix is ill-conditioned."""
    from oracle import oracle as orc
    verdicts = set()
    try:
        for tol in (1e-9, 1e-7, 1e-5):
            orc.set_feas_tol(tol)
            verdicts.add(P.full_process(active_set))
    finally:
        orc.set_feas_tol(1e-7)
    if len(verdicts) > 1:
        return True
    return kkt_condition(P, active_set) > cond_limit


@pytest.fixture(scope='session')
def oracle():
    from oracle import oracle as orc
    orc.build()
    return orc
