"""Shared helpers for the test-suite: golden fixtures and backends."""
import glob
import json
import os
import sys

import numpy as np
import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')
for p in (ROOT, os.path.join(ROOT, 'oracle')):
    if p not in sys.path:
        sys.path.insert(0, p)

import clustertracking_amd as cta  # noqa: E402
from clustertracking_amd import constraints as cons  # noqa: E402


def case_names():
    # refine fixtures only (link_cases.npz belongs to tests/test_link.py)
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, '*.npz'))
                  if not os.path.basename(p).startswith('link_'))


class Case(object):
    def __init__(self, name):
        self.name = name
        self.z = np.load(os.path.join(GOLDEN, name + '.npz'))
        self.call = json.loads(str(self.z['call']))
        self.frames = self.z['frames']
        self.f0 = pd.DataFrame(self.z['f0_values'], columns=list(self.z['f0_columns']),
                               index=self.z['f0_index'])
        self.ref_aborts = bool(int(self.z['ref_aborts']))

    def ref(self, which):
        cols = list(self.z['ref%s_columns' % which])
        return pd.DataFrame(self.z['ref%s_values' % which], columns=cols,
                            index=self.z['ref%s_index' % which])

    @property
    def pos_columns(self):
        return ['z', 'y', 'x'][-(self.frames.ndim - 1):]

    def kwargs(self):
        kw = dict(self.call)
        diameter = kw.pop('diameter')
        c = kw.pop('constraints', None)
        if c is not None:
            kw['constraints'] = getattr(cons, c['kind'])(c['dist'], c['ndim'])
        return diameter, kw

    def reader(self):
        if self.frames.shape[0] == 1 and 'frame' not in self.f0:
            return self.frames[0]
        return cta.ArrayReader(self.frames)

    def prepare(self):
        diameter, kw = self.kwargs()
        return cta.prepare_batch(self.f0.copy(), self.reader(), diameter, **kw)

    def run(self, run_batch=None):
        """refine_leastsq through the host layer; run_batch=None -> HIP engine."""
        diameter, kw = self.kwargs()
        if run_batch is not None:
            kw['_run_batch'] = run_batch
        return cta.refine_leastsq(self.f0.copy(), self.reader(), diameter, **kw)


def oracle_runner(n_threads=1):
    import ctr_oracle
    return lambda p, b: ctr_oracle.run_batch(p, b, n_threads)


def compare(res, ref, pos_columns):
    """(rmse, max) of the position difference over rows where both succeeded,
    plus the per-row success masks."""
    ok_a = ~np.isnan(res['cost'].values)
    ok_b = ~np.isnan(ref['cost'].values)
    both = ok_a & ok_b
    d = (res[pos_columns].values - ref[pos_columns].values)[both]
    if d.size == 0:
        return 0., 0., ok_a, ok_b
    return float(np.sqrt(np.mean(d ** 2))), float(np.abs(d).max()), ok_a, ok_b


def random_case(seed):
    """A random configuration (dimension, dtype, modes, bounds, constraints, separation) from a
    seed: the soak tests, tests/tools/check_vs_reference.py and the `hard_*` fixtures share it."""
    rng = np.random.RandomState(seed)
    ndim = int(rng.choice([2, 3], p=[0.7, 0.3]))
    iso = bool(rng.rand() < 0.5)
    if ndim == 2:
        shape = tuple(rng.randint(60, 120, 2))
        size = rng.uniform(2.5, 4.5) if iso else tuple(rng.uniform(2.5, 4.5, 2))
        n = rng.randint(3, 30)
    else:
        shape = tuple(rng.randint(24, 44, 3))
        size = rng.uniform(2., 3.) if iso else tuple(rng.uniform(2., 3.2, 3))
        n = rng.randint(2, 9)
    sz = np.broadcast_to(size, (ndim,))
    diameter = int(4 * sz[0]) | 1 if iso else tuple(int(4 * s) | 1 for s in sz)
    if not iso and len(set(diameter)) == 1:   # equal diameters would mean isotropic (utils.py:52-56)
        diameter = (diameter[0] + 2,) + tuple(diameter[1:])
    margin = tuple(int(2 * s) + 2 for s in sz)
    dtype = [np.uint8, np.uint16, np.float32, np.float64][rng.randint(4)]
    im, truth, p0 = cta.artificial.random_frame(shape, n, size, 100, int(rng.choice([0, 10])),
                                                seed, margin=margin)
    im = im.astype(dtype)
    f0 = pd.DataFrame(p0 + rng.uniform(-0.7, 0.7, p0.shape), columns=['z', 'y', 'x'][-ndim:])
    f0['signal'] = 90.
    if iso:
        f0['size'] = float(size) * rng.uniform(0.9, 1.1)
    else:
        for c, s in zip(['size_z', 'size_y', 'size_x'][-ndim:], sz):
            f0[c] = s * rng.uniform(0.9, 1.1)
    if rng.rand() < 0.7:
        f0['background'] = 4.
    modes = {}
    r = rng.rand()
    if r < 0.25:
        modes['size'] = 'var'
    elif r < 0.4:
        modes['size'] = 'cluster'
    r = rng.rand()
    if r < 0.15:
        modes['signal'] = 'cluster'
    elif r < 0.25:
        modes['signal'] = 'const'
    if rng.rand() < 0.1:
        modes['background'] = 'const'
    kw = dict(param_mode=modes or None)
    if rng.rand() < 0.3:
        kw['bounds'] = dict(signal=(20., 400.), pos_diff=float(rng.uniform(1.5, 4.)),
                            size_rel_diff=0.4)
    if rng.rand() < 0.25:
        kind = ['dimer', 'trimer', 'tetramer'][rng.randint(3)]
        kw['constraints'] = getattr(cta.constraints, kind)(2. * np.asarray(sz, float), ndim)
    if rng.rand() < 0.3:
        kw['separation'] = tuple(float(d) * 1.6 for d in np.broadcast_to(diameter, (ndim,)))
    if rng.rand() < 0.2:
        kw['max_iter'] = int(rng.randint(1, 4))
    return f0, im, diameter, kw
