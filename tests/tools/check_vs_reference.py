"""One-off (build container only): random configurations through the REAL reference
(oracle/refshim.py, converged: tol=1e-14) and through the C oracle; prints the differences.
Complements the committed golden fixtures with inputs nobody looked at.
    python tests/tools/check_vs_reference.py [n_seeds] [first_seed] [lowpass]
With a third argument every configuration gets a random noise_size / threshold (refine.py:37-40;
the reference's lowpass with the restated trackpy taps of oracle/refshim.py)."""
import os, sys, warnings
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
for p in (ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'oracle')):
    sys.path.insert(0, p)
import numpy as np
import pandas as pd
import refshim
ct = refshim.load()
from clustertracking import constraints as ref_cons
import clustertracking_amd as cta
import _cases

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
first = int(sys.argv[2]) if len(sys.argv) > 2 else 9000
worst = 0.
for seed in range(first, first + n):
    f0, im, diameter, kw = _cases.random_case(seed)
    if len(sys.argv) > 3:
        r2 = np.random.RandomState(seed + 77)
        kw['noise_size'] = float(r2.choice([0.5, 1., 1.5, 2.])) if r2.rand() < 0.6 else \
            tuple(float(x) for x in r2.choice([0., 0.5, 1., 1.5], im.ndim))
        if not np.any(np.asarray(kw['noise_size']) > 0):
            kw['noise_size'] = 1.
        if r2.rand() < 0.4:
            kw['threshold'] = float(r2.uniform(1., 12.))
    kw_ref = dict(kw)
    if 'constraints' in kw:
        c = kw['constraints'][0]
        kw_ref['constraints'] = getattr(ref_cons, c['kind'])(c['args'][0], im.ndim)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        try:
            ref = ct.refine_leastsq(f0.copy(), im, diameter, tol=1e-14,
                                    options=dict(maxiter=1000), **kw_ref)
        except Exception as e:
            print('seed %d: reference raised %s' % (seed, type(e).__name__))
            continue
    ours = _cases.refine_leastsq(f0.copy(), im, diameter, _run_batch=_cases.oracle_runner(), **kw)
    pc = ['z', 'y', 'x'][-im.ndim:]
    nan_r, nan_o = np.isnan(ref['cost'].values), np.isnan(ours['cost'].values)
    both = ~nan_r & ~nan_o
    d = np.abs(ref[pc].values - ours[pc].values)[both]
    dc = np.abs(ref['cost'].values - ours['cost'].values)[both]
    same_cl = (ref['cluster'].values == ours['cluster'].values).all()
    flag = '' if (nan_r == nan_o).all() and (d.size == 0 or d.max() < 1e-5) else '   <<<'
    worst = max(worst, d.max() if d.size else 0.)
    print('seed %d: n=%d nan ref/ours %d/%d  max dpos %.1e  max dcost %.1e  labels equal %s  %s%s' % (
        seed, len(f0), nan_r.sum(), nan_o.sum(), d.max() if d.size else 0., dc.max() if dc.size else 0.,
        same_cl, {k: v for k, v in kw.items() if k in ('param_mode', 'max_iter', 'noise_size', 'threshold')}, flag))
print('worst', worst)
