"""Generate the ``args_*`` fixtures: arguments of refine_leastsq that the other fixtures leave at
their defaults -- param_val, max_shift, residual_factor, max_iter.

Run in the build container only (needs /root/reference; see oracle/refshim.py):

    python tests/golden/make_golden_args.py

Same file format as make_golden.py.
"""
import os
import sys

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

import make_golden as mg  # noqa: E402  (loads the reference through oracle/refshim.py)
from clustertracking_amd import artificial  # noqa: E402


def main():
    # param_val overrides the table (refine.py:293-297), max_shift 0.4 forces extra re-window
    # rounds, residual_factor rescales norm and cost (refine.py:354,379), max_iter caps the rounds
    im, truth, p0 = artificial.random_frame((180, 200), 45, 3., 100, 10, 51, margin=10)
    rng = np.random.RandomState(51)
    f0 = mg.table(p0 + rng.uniform(-0.8, 0.8, p0.shape), 2., 40., 1., 2, True)
    mg.save_case('args_param_val', f0, im[None],
                 dict(diameter=13, param_val=dict(signal=95., size=3.1, background=5.),
                      max_shift=0.4, residual_factor=25000., max_iter=4))
    # (a custom t_column cannot be pinned: the reference groups by the literal 'frame',
    #  refine.py:336, and raises KeyError for any other name)

if __name__ == '__main__':
    main()
