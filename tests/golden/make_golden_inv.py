"""Generate the ``inv*`` fixtures: refine_leastsq with ``fit_function='inv_series_<N>'`` (reference
fitfunc.py:148-154,334-343: ``signal_mult / polyval([1, param_a, ...], r2)``, N + 1 profile
parameters, all 1 by default), features drawn with the same profile
(clustertracking_amd.artificial.feat_inv_series; the reference has no drawing function for it).

Run in the build container only (needs /root/reference; see oracle/refshim.py):

    python tests/golden/make_golden_inv.py

Same file format as make_golden.py.  The reference has no ``dfunc`` for this profile: its SLSQP
differentiates the objective numerically, so its converged run B is only as converged as that
allows (as for the disc).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

import make_golden as mg  # noqa: E402  (loads the reference through oracle/refshim.py)
import make_golden_profiles as mp  # noqa: E402
from clustertracking_amd import artificial  # noqa: E402

NAMES = ['signal_mult'] + ['param_' + chr(97 + i) for i in range(6)]


def vals(p):
    return dict(zip(NAMES, [float(x) for x in p]))


def main(only=None):
    if only:
        save = mg.save_case
        mg.save_case = lambda name, *a, **k: save(name, *a, **k) if name in only else None
    # inv_series_2 with its defaults (1, 1, 1): 1 / (r^4 + r^2 + 1); 2D isotropic, clean and noisy
    p = [1., 1., 1.]
    for tag, noise in (('', 0), ('_noisy', 10)):
        im, truth, p0 = mp.frame((144, 144), 4, 4., 'inv_series', 71, noise, 0.3, dict(p=p))
        mg.save_case('inv2_2d' + tag, mg.table(p0, 4., 150., noise / 2., 2, True), im[None],
                     dict(diameter=16, fit_function='inv_series_2'))
    # inv_series_1 (1 / (r^2 + 1)), 3D isotropic
    p = [1., 1.]
    im, truth, p0 = mp.frame((72, 72, 72), 2, 4., 'inv_series', 72, 6, 0.3, dict(p=p))
    mg.save_case('inv1_3d', mg.table(p0, 4., 150., 3., 3, True), im[None],
                 dict(diameter=16, fit_function='inv_series_1'))
    # free size (isotropic), started 2.5 % off
    p = [1., 1., 1.]
    im, truth, p0 = mp.frame((144, 144), 4, 4., 'inv_series', 77, 6, 0.3, dict(p=p))
    mg.save_case('inv2_2d_sizevar', mg.table(p0, 4.1, 150., 3., 2, True), im[None],
                 dict(diameter=16, fit_function='inv_series_2', param_mode=dict(size='var')))
    # inv_series_3 with its own coefficients, 2D anisotropic, free sizes.  The reference's SLSQP (on
    # its numerical gradient) gives up on 13 of the 19 features here ("Inequality constraints
    # incompatible", also with the sizes started at their true values; with one shared size as
    # above, or constant sizes, it fits them all): a solver-specific fixture, compared per cluster
    p = [1.5, 0.5, 2., 1.5]
    im, truth, p0 = mp.frame((176, 112), 4, (5., 3.), 'inv_series', 73, 8, 0.25, dict(p=p))
    mg.save_case('inv3_2d_a_sizevar', mg.table(p0, (5.1, 2.95), 150., 4., 2, False), im[None],
                 dict(diameter=(20, 12), fit_function='inv_series_3', param_val=vals(p),
                      param_mode=dict(size='var')))
    # free profile parameters: param_a per cluster, param_b per feature (started 10 % off)
    p = [1., 1.2, 0.8]
    im, truth, p0 = mp.frame((144, 144), 4, 4., 'inv_series', 74, 6, 0.25, dict(p=p))
    mg.save_case('inv2_2d_free_params', mg.table(p0, 4., 150., 3., 2, True), im[None],
                 dict(diameter=16, fit_function='inv_series_2',
                      param_val=dict(signal_mult=1., param_a=1.3, param_b=0.75),
                      param_mode=dict(param_a='cluster', param_b='var')))
    # the widest tables the engine takes (12 columns): inv_series_3 in 3D anisotropic,
    # inv_series_6 in 2D isotropic
    p = [1., 0.5, 1., 1.]
    im, truth, p0 = mp.frame((56, 88, 88), 2, (3., 5., 5.), 'inv_series', 75, 6, 0.2, dict(p=p))
    mg.save_case('inv3_3d_a', mg.table(p0, (3., 5., 5.), 150., 3., 3, False), im[None],
                 dict(diameter=(12, 20, 20), fit_function='inv_series_3', param_val=vals(p)))
    p = [1., 0., 0., 0.5, 0., 1., 1.]
    im, truth, p0 = mp.frame((144, 144), 3, 4., 'inv_series', 76, 6, 0.3, dict(p=p))
    mg.save_case('inv6_2d', mg.table(p0, 4., 150., 3., 2, True), im[None],
                 dict(diameter=16, fit_function='inv_series_6', param_val=vals(p)))


if __name__ == '__main__':
    main(sys.argv[1:])
