"""Soak test: many random configurations, engine vs C oracle (tests/test_gpu_parity.py's
generator with more seeds).   python tests/tools/soak_random.py [n_seeds] [first_seed]"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
for p in (ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'oracle')):
    sys.path.insert(0, p)
import numpy as np
import pandas as pd
import clustertracking_amd as cta
from clustertracking_amd import _abi, _lib
import ctr_oracle
import _cases

n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 300
first = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
eng = _lib.default_engine(0)
bad = 0
n_clusters = 0
worst = 0.
for seed in range(first, first + n_seeds):
    f0, im, diameter, kw = _cases.random_case(seed)
    if os.environ.get('SOAK_CONSTRAINED_ONLY') and 'constraints' not in kw:
        continue
    prep = cta.prepare_batch(f0, im, diameter, **kw)
    if os.environ.get('SOAK_THROUGHPUT'):
        prep.problem.flags |= _abi.FLAG_THROUGHPUT   # scheduling flag: same results expected
    b = prep.batch
    ref = _abi.HostBatch(b.frames, b.frame_index, b.feat_offset, b.params, b.low, b.high)
    eng.refine_batch(prep.problem, b)
    ctr_oracle.run_batch(prep.problem, ref, 4)
    nd = im.ndim
    n_clusters += b.n_clusters
    same_status = (b.status == ref.status).all()
    ok = (ref.status == 0) & (b.status == 0)
    cost_ok = np.allclose(b.cost[ok], ref.cost[ok], rtol=1e-7, atol=1e-12)
    rows = np.repeat(ok, np.diff(ref.feat_offset))
    d = np.abs(b.params_out[:, 2:2 + nd] - ref.params_out[:, 2:2 + nd])[rows]
    dmax = d.max() if d.size else 0.
    worst = max(worst, dmax)
    if not same_status or not cost_ok or dmax > 1e-3:
        bad += 1
        print('seed %d: status-eq %s cost-ok %s dmax %.2e modes %s cons %s' % (
            seed, same_status, cost_ok, dmax, kw.get('param_mode'), 'constraints' in kw))
print('seeds %d clusters %d bad %d worst position difference %.2e px' % (n_seeds, n_clusters, bad, worst))
