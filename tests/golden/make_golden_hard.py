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
}


def main():
    for name, seed in HARD.items():
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
    main()
