"""Synthetic problem builders: the random dense mpQP generator and the
configuration problems named in BASELINE.json.

Every builder returns the RAW matrices as a dict
``{A, b, c, H, Q (or None), A_t, b_t, F, equality_indices}`` -- exactly the
positional arguments of the program constructors -- so the same data can be
fed to ``ppopt_amd.MPQP_Program`` and (in the golden-vector generator only) to
the reference's ``MPQP_Program``.

Reference behaviour restated here:
  * ``generate_mpqp``  -- src/ppopt/problem_generator.py:25-78 (same sequence of
    ``numpy.random.Generator`` draws, so a given seed yields the same matrices
    under the same numpy; tests/golden/ holds the matrices themselves for the
    seeds that are benchmarked so nothing depends on the bit stream).
  * double integrator  -- doc/mpc.rst:69-119 generalised to horizon N.
  * transport mpLP/mpQP -- doc/mplp_tut.rst:28-36, doc/tutorial.rst.
  * control allocation -- doc/control_allocation_example.rst:21-236.
  * quad-tank MPC      -- not in the reference; defined here.
"""
from typing import Dict, Optional

import numpy


def _pack(A, b, c, H, Q, A_t, b_t, F, eq=None) -> Dict:
    f = lambda m: None if m is None else numpy.array(m, dtype=numpy.float64)
    return {'A': f(A), 'b': f(b).reshape(-1, 1), 'c': f(c).reshape(-1, 1), 'H': f(H), 'Q': f(Q), 'A_t': f(A_t),
            'b_t': f(b_t).reshape(-1, 1), 'F': f(F), 'equality_indices': list(eq) if eq is not None else []}


def generate_mpqp_data(x: int = 2, t: int = 2, m: int = 10, seed: Optional[int] = None) -> Dict:
    """Random dense mpQP: Q = R'R + I, m sparse integer rows in A|F, a +-1e7 box on x
    and a +-Range box on theta (problem_generator.py:25-78)."""
    prng = numpy.random.default_rng(seed)

    Q = prng.random((x, x))
    Q = Q.T @ Q + numpy.eye(x)

    draw = lambda: prng.random(1)

    range_value = numpy.round(20 * draw() + 5)
    x_border = numpy.round(8 * draw() + 1) / 10
    x_shift = numpy.round(8 * draw() + 1) / 10
    t_border = numpy.round(8 * draw() + 1) / 10
    t_shift = numpy.round(8 * draw() + 1) / 10

    c = (prng.random((x, 1)) - .5) / draw()

    eig = numpy.linalg.eigvals(Q)
    span = range_value * (max(eig) - min(eig))

    A = numpy.zeros((m, x))
    F = numpy.zeros((m, t))
    for i in range(m):
        while True:
            pick = prng.random(x) >= x_border
            A[i][pick] = numpy.floor((prng.random(sum(pick)) - x_shift) * span)
            if any(A[i] != 0):
                break
        pick = prng.random(t) >= t_border
        F[i][pick] = numpy.floor((prng.random(sum(pick)) - t_shift) * span)

    A = numpy.vstack([A, numpy.eye(x), -numpy.eye(x)])
    F = numpy.vstack([F, numpy.zeros((2 * x, t))])
    b = numpy.vstack([prng.random((m, 1)) / prng.random(1), 10 ** 7 * numpy.ones((2 * x, 1))])
    A_t = numpy.vstack([numpy.eye(t), -numpy.eye(t)])
    b_t = span * numpy.ones((2 * t, 1))
    H = numpy.zeros((x, t))
    return _pack(A, b, c, H, Q, A_t, b_t, F)


def transport_mplp_data() -> Dict:
    """Config 1: the 2-plant/2-market transport mpLP of doc/mplp_tut.rst:28-36."""
    A = [[1, 1, 0, 0], [0, 0, 1, 1], [-1, 0, -1, 0], [0, -1, 0, -1], [-1, 0, 0, 0], [0, -1, 0, 0], [0, 0, -1, 0],
         [0, 0, 0, -1]]
    b = [350, 600, 0, 0, 0, 0, 0, 0]
    c = [178, 187, 187, 151]
    F = [[0, 0], [0, 0], [-1, 0], [0, -1], [0, 0], [0, 0], [0, 0], [0, 0]]
    A_t = numpy.vstack((numpy.eye(2), -numpy.eye(2)))
    b_t = [1000, 1000, 0, 0]
    H = numpy.zeros((4, 2))
    return _pack(A, b, c, H, None, A_t, b_t, F)


def transport_mpqp_data() -> Dict:
    """The quadratic-cost version (doc/tutorial.rst; tests/test_fixtures.py:17-31 `qp_problem`)."""
    d = transport_mplp_data()
    d['c'] = 25.0 * numpy.ones((4, 1))
    d['Q'] = 2.0 * numpy.diag([153.0, 162.0, 162.0, 126.0])
    return d


def double_integrator_data(horizon: int = 5, x_bound: float = 4.0, u_bound: float = 1.0) -> Dict:
    """Config 2: explicit MPC of the double integrator, non-condensed (states and inputs are
    decision variables, dynamics are 2*N equality rows), doc/mpc.rst:69-119 with N = horizon."""
    N = horizon
    A_ss = numpy.array([[1.0, 1.0], [0.0, 1.0]])
    B_ss = numpy.array([[0.5], [1.0]])
    nz = 3 * N  # x_1..x_N (2 each) then u_0..u_{N-1}
    A_eq = numpy.zeros((2 * N, nz))
    F_eq = numpy.zeros((2 * N, 2))
    for k in range(N):
        A_eq[2 * k:2 * k + 2, 2 * k:2 * k + 2] = numpy.eye(2)
        if k > 0:
            A_eq[2 * k:2 * k + 2, 2 * (k - 1):2 * k] = -A_ss
        A_eq[2 * k:2 * k + 2, 2 * N + k:2 * N + k + 1] = -B_ss
    F_eq[0:2] = A_ss
    ub = numpy.vstack([x_bound * numpy.ones((2 * N, 1)), u_bound * numpy.ones((N, 1))])
    A = numpy.vstack([A_eq, numpy.eye(nz), -numpy.eye(nz)])
    b = numpy.vstack([numpy.zeros((2 * N, 1)), ub, ub])
    F = numpy.vstack([F_eq, numpy.zeros((2 * nz, 2))])
    A_t = numpy.vstack([numpy.eye(2), -numpy.eye(2)])
    b_t = x_bound * numpy.ones((4, 1))
    return _pack(A, b, numpy.zeros((nz, 1)), numpy.zeros((nz, 2)), numpy.eye(nz), A_t, b_t, F, range(2 * N))


# discrete quadruple-tank model (Johansson 2000 minimum-phase parameters, zero-order hold, Ts = 5 s);
# numbers are frozen here to six decimals -- they ARE the definition of config 3.
_QT_A = numpy.array([[0.922521, 0.0, 0.187440, 0.0],
                     [0.0, 0.945959, 0.0, 0.149217],
                     [0.0, 0.0, 0.804615, 0.0],
                     [0.0, 0.0, 0.0, 0.846482]])
_QT_B = numpy.array([[0.399908, 0.023573],
                     [0.012086, 0.305498],
                     [0.0, 0.215063],
                     [0.143779, 0.0]])


def quad_tank_data(horizon: int = 10, x_bound: float = 5.0, u_bound: float = 1.0, theta_bound: float = 5.0,
                   r_weight: float = 0.1) -> Dict:
    """Config 3: condensed MPC of the linearised quadruple tank (4 states, 2 inputs).

    Decision variables are the N*2 inputs; parameters are the 4 initial (deviation) states.
    Rows: input box (2*2N) then state box over the horizon (2*4N)."""
    N = horizon
    nx, nu = 4, 2
    Phi = numpy.zeros((nx * N, nx))
    Gam = numpy.zeros((nx * N, nu * N))
    Ak = numpy.eye(nx)
    powers = [numpy.eye(nx)]
    for k in range(N):
        powers.append(powers[-1] @ _QT_A)
    for k in range(N):
        Phi[nx * k:nx * k + nx] = powers[k + 1]
        for j in range(k + 1):
            Gam[nx * k:nx * k + nx, nu * j:nu * j + nu] = powers[k - j] @ _QT_B
    Qb = numpy.eye(nx * N)
    Rb = r_weight * numpy.eye(nu * N)
    Q = Gam.T @ Qb @ Gam + Rb
    Q = 0.5 * (Q + Q.T)
    H = Gam.T @ Qb @ Phi
    c = numpy.zeros((nu * N, 1))
    A = numpy.vstack([numpy.eye(nu * N), -numpy.eye(nu * N), Gam, -Gam])
    b = numpy.vstack([u_bound * numpy.ones((2 * nu * N, 1)), x_bound * numpy.ones((2 * nx * N, 1))])
    F = numpy.vstack([numpy.zeros((2 * nu * N, nx)), -Phi, Phi])
    A_t = numpy.vstack([numpy.eye(nx), -numpy.eye(nx)])
    b_t = theta_bound * numpy.ones((2 * nx, 1))
    return _pack(A, b, c, H, Q, A_t, b_t, F)


def control_allocation_data() -> Dict:
    """Config 5: octocopter control allocation (doc/control_allocation_example.rst:21-236).
    Q has rank 4 of 8, so many active sets give a singular KKT matrix. Note the doc rebinds
    ``m = 5.0`` (mass) to ``m = 4`` (axes) before the trim point is computed."""
    g, r, n, m = 9.8, 0.35, 8, 4
    rot = numpy.array([+1.0, -1.0, +1.0, -1.0, +1.0, -1.0, +1.0, -1.0])
    phi = numpy.linspace(0.0, 2.0 * numpy.pi, n + 1)[0:-1]
    xr, yr = numpy.sin(phi), numpy.cos(phi)
    Ct, FoM = 0.014, 0.7
    Cq = Ct ** 1.5 / FoM / numpy.sqrt(2.0)
    ratio = 1.4
    J = numpy.zeros((m, n))
    dT = Ct / (r * Cq)
    J[0, :] = -dT * numpy.ones(n)
    J[1, :] = -dT * yr
    J[2, :] = dT * xr
    J[3, :] = rot
    trim_cmd = numpy.array([-m * g, 0.0, 0.0, 0.0])
    x_trim = (numpy.linalg.pinv(J) @ trim_cmd.reshape((4, 1))).reshape(n)
    W = numpy.diag([20.0, 100.0, 100.0, 5.0])
    x_min = numpy.zeros(n)
    x_max = ratio * numpy.mean(x_trim) * numpy.ones(n)
    rp, yaw, thrust = numpy.array([-15.0, 15.0]), numpy.array([-3.0, 3.0]), numpy.array([-1.2 * m * g, -0.8 * m * g])
    cmd_min = numpy.array([thrust[0], rp[0], rp[0], yaw[0]]) * 1.1
    cmd_max = numpy.array([thrust[1], rp[1], rp[1], yaw[1]]) * 1.1
    Q = J.T @ W @ J + J.T @ J
    c = -J.T @ J @ x_trim.reshape((n, 1))
    H = -J.T @ W
    A = numpy.vstack([-numpy.eye(n), numpy.eye(n)])
    b = numpy.vstack([-x_min.reshape((n, 1)), x_max.reshape((n, 1))])
    F = numpy.zeros((2 * n, m))
    A_t = numpy.vstack([-numpy.eye(m), numpy.eye(m)])
    b_t = numpy.vstack([-cmd_min.reshape((m, 1)), cmd_max.reshape((m, 1))])
    return _pack(A, b, c, H, Q, A_t, b_t, F)


def generate_mpmiqp_data(x: int = 6, t: int = 3, m: int = 12, n_bin: int = 4, seed: Optional[int] = None) -> Dict:
    """Random dense mixed-integer mpQP for measurements (not in the reference): ``generate_mpqp_data`` on
    x + n_bin variables whose last n_bin are declared binary.  Adds the key ``binary_indices``."""
    d = generate_mpqp_data(x + n_bin, t, m, seed)
    d['binary_indices'] = list(range(x, x + n_bin))
    return d


def generate_mpqp(x: int = 2, t: int = 2, m: int = 10, seed: Optional[int] = None):
    """The random mpQP as a program object, like the reference's ``generate_mpqp`` (problem_generator.py:25-78); the
    constructor's presolve runs on the device."""
    from .mpqp_program import MPQP_Program
    d = generate_mpqp_data(x, t, m, seed)
    return MPQP_Program(d['A'], d['b'], d['c'], d['H'], d['Q'], d['A_t'], d['b_t'], d['F'])


def generate_mplp(x: int = 2, t: int = 2, m: int = 10, seed: Optional[int] = None):
    """The random mpLP of the reference (problem_generator.py:9-22): the presolved rows of ``generate_mpqp`` without the
    quadratic term."""
    from .mplp_program import MPLP_Program
    q = generate_mpqp(x, t, m, seed)
    return MPLP_Program(q.A, q.b, q.c, q.H, q.A_t, q.b_t, q.F)
