"""The random program draw of tools/fuzz_scan.py, on its own so that oracle/ref_harness/gen_fuzz_pins.py can replay a fuzz run's
stream (same kind, same rng seed) and rebuild the program a logged CASE line names -- no GPU involved."""
import numpy


def draw(kind, rng, pg):
    """One program of the class ``kind`` from the generator stream ``rng``; returns (data dict, tag).  ``pg`` = ppopt_amd.problem_generator."""
    if kind == 'mpc':
        choice = int(rng.integers(0, 2))
        N = int(rng.integers(2, 5))
        d = pg.double_integrator_data(N, x_bound=float(rng.uniform(2, 6)), u_bound=float(rng.uniform(0.5, 2))) if choice == 0 else pg.quad_tank_data(int(rng.integers(2, 4)))
        tag = ('dblint', N) if choice == 0 else ('quadtank',)
    elif kind == 'big':   # shapes near the limits of the kernel instantiations (two tableau rows per lane, n_theta 9-10, 32 columns)
        nx, nt = int(rng.integers(10, 27)), int(rng.integers(3, 11))
        m = int(rng.integers(nx + 5, 2 * nx + 40))
        seed = int(rng.integers(0, 10 ** 6))
        d = pg.generate_mpqp_data(nx, nt, m, seed)
        tag = (nx, nt, m, seed)
    else:
        nx, nt = int(rng.integers(3, 9)), int(rng.integers(1, 7))
        m = int(rng.integers(nx + 3, 3 * nx + 4))
        seed = int(rng.integers(0, 10 ** 6))
        d = pg.generate_mpqp_data(nx, nt, m, seed)
        tag = (nx, nt, m, seed)
        if kind == 'open':      # parameter sets open in some direction (round 4, k_recession): the big-M box of x dropped (half of the
            # programs), of the rows of A_t only the lower bounds / one lower and one upper bound / a random subset kept
            if rng.random() < 0.5:
                keep = numpy.abs(d['b']).ravel() < 1e6
                d['A'], d['b'], d['F'] = d['A'][keep], d['b'][keep], d['F'][keep]
            lo = [i for i in range(d['A_t'].shape[0]) if d['A_t'][i].min() < 0]
            hi = [i for i in range(d['A_t'].shape[0]) if d['A_t'][i].max() > 0]
            mode = int(rng.integers(0, 3))
            rows = lo if mode == 0 else (lo[:1] + hi[:1] if mode == 1 else sorted(rng.choice(d['A_t'].shape[0], size=max(1, d['A_t'].shape[0] // 2), replace=False).tolist()))
            d['A_t'], d['b_t'] = d['A_t'][rows], d['b_t'][rows]
            tag = tag + ('open', mode)
        if kind == 'mpqp_eq':   # the first one or two rows become equalities
            d['equality_indices'] = list(range(int(rng.integers(1, 3))))
            tag = tag + (len(d['equality_indices']),)
    return d, tag
