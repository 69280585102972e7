"""Generate the ``hard_*`` fixtures: configurations of the random generator
(tests/_cases.py:random_case) on which a Gauss-Newton Levenberg-Marquardt loop stalls or
stops early, kept as regression cases for the solver shared by the oracle and the engine.

Run in the build container only (needs /root/reference; see oracle/refshim.py):

    python tests/golden/make_golden_hard.py

Same file format as make_golden.py: inputs + what the REFERENCE returned with its defaults
(refA) and converged (refB: tol=1e-14, maxiter=1000).
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

import make_golden as mg  # noqa: E402  (loads the reference through oracle/refshim.py)
import _cases  # noqa: E402

# name -> (seed, why it is here)
HARD = {
    # no background column: the background starts ON its lower bound 0; freeing it gave a
    # projected step with a negative predicted decrease, which was taken for convergence
    'hard_bg_at_bound': 9035,
    'hard_bg_at_bound_modes': 9030,
    # three nearly coincident features: Gauss-Newton crawls along a valley (>100 iterations)
    'hard_valley_triple': 9021,
    # pair in a valley: 1900 Gauss-Newton iterations, 17 with the exact Hessian
    'hard_valley_pair': 9164,
    # constrained fits on which the round-1 solver (l1 merit function, Powell penalty weight)
    # stalled or ended in a state that depended on the summation order -- engine and oracle then
    # disagreed in the soak test (tests/tools/soak_random.py seeds 201619, 202686, 203865, 52301)
    # or failed where the reference succeeds (9099, 9131).  Since round 2 the constrained
    # iteration is a feasible-point method (oracle/ctr_oracle.c:solve, retract).
    'hard_cons_trimer_sizecluster': 201619,
    'hard_cons_dimer_bounds': 202686,
    'hard_cons_trimer_big': 203865,
    'hard_cons_trimer': 52301,
    'hard_cons_dimer_sizevar': 9099,
    'hard_cons_trimer_2': 9131,
}


def tetramer2d():
    """2D tetramers (constraints.py:102-123: the 4 smallest of the 6 pair distances equal the
    bond length -- a rhombus): six of them, clean and noisy."""
    import numpy as np
    from clustertracking_amd import artificial
    rng = np.random.RandomState(77)
    size = 3.5
    im = np.zeros((170, 250), np.uint8)
    truth = []
    for gy in range(2):
        for gx in range(3):
            c = np.array([45. + gy * 80, 45. + gx * 80]) + rng.uniform(-.5, .5, 2)
            a = rng.uniform(0, 2 * np.pi)
            # rhombus angle; well away from 60 degrees, where the short diagonal equals the bond
            # length and the reference's sort-based constraint function has a kink
            shear = rng.uniform(np.pi * 5 / 12, np.pi / 2)
            e1 = np.array([np.sin(a), np.cos(a)]) * 2 * size
            e2 = np.array([np.sin(a + shear), np.cos(a + shear)]) * 2 * size
            truth += [c, c + e1, c + e1 + e2, c + e2]
    truth = np.array(truth)
    for p in truth:
        artificial.draw_gaussian(im, p, size, 110)
    p0 = truth + rng.uniform(-0.6, 0.6, truth.shape)
    call = dict(diameter=15, separation=30,
                constraints=dict(kind='tetramer', dist=[2 * size] * 2, ndim=2))
    mg.save_case('tetramer2d_constrained', mg.table(p0, size, 100., 0., 2, True), im[None], call,
                 do_intermediates=False)
    imn = artificial.add_poisson_noise(im, 10, rng)
    f0n = mg.table(p0, size, 100., 5., 2, True)
    mg.save_case('tetramer2d_constrained_noisy', f0n, imn[None], call, do_intermediates=False)


def big_clusters():
    """Clusters beyond the block kernel's 64 features / 127 variables (the engine's large-cluster
    path; BASELINE cfg 3 at its stated density percolates into such clusters): one 2D cluster of
    90 features (271 variables) and one 3D anisotropic cluster of 75 features (301 variables),
    Poisson noise.  The reference's SLSQP needs seconds to minutes on these."""
    import numpy as np
    import pandas as pd
    from clustertracking_amd import artificial
    rng = np.random.RandomState(5)
    size, ny, nx, sp = 3., 9, 10, 11.
    im = np.zeros((int(sp * (ny + 1)), int(sp * (nx + 1))), np.uint8)
    truth = np.array([[sp * (1 + gy), sp * (1 + gx)] for gy in range(ny) for gx in range(nx)]) \
        + rng.uniform(-1.5, 1.5, (ny * nx, 2))
    for p in truth:
        artificial.draw_gaussian(im, p, size, 100)
    im = artificial.add_poisson_noise(im, 10, rng)
    p0 = truth + rng.uniform(-0.5, 0.5, truth.shape)
    mg.save_case('big_cluster_2d', mg.table(p0, size, 90., 5., 2, True), im[None],
                 dict(diameter=13), do_intermediates=False)

    rng = np.random.RandomState(6)
    size3, sp3, grid = (2., 4., 4.), (7., 13., 13.), (3, 5, 5)
    shape = tuple(int(s * (g + 1)) for s, g in zip(sp3, grid))
    im = np.zeros(shape, np.uint8)
    truth = np.array([[sp3[0] * (1 + gz), sp3[1] * (1 + gy), sp3[2] * (1 + gx)]
                      for gz in range(grid[0]) for gy in range(grid[1]) for gx in range(grid[2])]) \
        + rng.uniform(-1., 1., (grid[0] * grid[1] * grid[2], 3))
    for p in truth:
        artificial.draw_gaussian(im, p, size3, 100)
    im = artificial.add_poisson_noise(im, 10, rng)
    p0 = truth + rng.uniform(-0.5, 0.5, truth.shape)
    mg.save_case('big_cluster_3d', mg.table(p0, size3, 90., 5., 3, False), im[None],
                 dict(diameter=[9, 17, 17]), do_intermediates=False)


def big_cluster_close_pairs():
    """A cluster of the large-cluster path (80 features, 241 variables) in which some start
    positions are 0.4-0.7 px apart, as in BASELINE cfg 3 at its stated density: features that
    close leave nearly dependent columns (the case the preconditioner's aggregates are for,
    DESIGN.md 4.5).  What does the REFERENCE do on such a cluster?  It fits it (both runs)."""
    import numpy as np
    from clustertracking_amd import artificial
    rng = np.random.RandomState(8)
    size, ny, nx, sp = 3., 8, 9, 11.
    truth = np.array([[sp * (1 + gy), sp * (1 + gx)] for gy in range(ny) for gx in range(nx)]) \
        + rng.uniform(-1.5, 1.5, (ny * nx, 2))
    # eight more features right next to an existing one (0.4-0.7 px apart, 0.13-0.23 sizes)
    partners = rng.choice(len(truth), 8, replace=False)
    ang = rng.uniform(0, 2 * np.pi, 8)
    rad = rng.uniform(0.4, 0.7, 8)
    truth = np.concatenate([truth, truth[partners] + np.stack([rad * np.sin(ang), rad * np.cos(ang)], 1)])
    im = np.zeros((int(sp * (ny + 1)), int(sp * (nx + 1))), np.uint8)
    for p in truth:
        artificial.draw_gaussian(im, p, size, 100)
    im = artificial.add_poisson_noise(im, 10, rng)
    p0 = truth + rng.uniform(-0.15, 0.15, truth.shape)
    mg.save_case('big_cluster_close_pairs', mg.table(p0, size, 90., 5., 2, True), im[None],
                 dict(diameter=13), do_intermediates=False)


def main(only=None):
    if only is None or 'big_cluster_close_pairs' in only:
        big_cluster_close_pairs()
    if only is None or 'tetramer2d' in only:
        tetramer2d()
    if only is None or 'big_clusters' in only:
        big_clusters()
    for name, seed in HARD.items():
        if only is not None and name not in only:
            continue
        f0, im, diameter, kw = _cases.random_case(seed)
        call = dict(diameter=diameter)
        for key, val in kw.items():
            if val is None:
                continue
            if key == 'constraints':
                c = val[0]
                val = dict(kind=c['kind'], dist=[float(x) for x in c['args'][0]], ndim=int(im.ndim))
            if key == 'bounds':
                val = {k: (list(v) if isinstance(v, tuple) else v) for k, v in val.items()}
            call[key] = val
        mg.save_case(name, f0, im[None], call, do_intermediates=False)


if __name__ == '__main__':
    main(sys.argv[1:] or None)
