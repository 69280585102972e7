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
