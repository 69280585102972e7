"""params_std of engine and oracle side by side for soak seeds:  python tests/tools/std_debug.py seed [seed ...]"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
for p in (ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'oracle')):
    sys.path.insert(0, p)
import numpy as np
import clustertracking_amd as cta
from clustertracking_amd import _abi, _lib
import ctr_oracle
import _cases
eng = _lib.default_engine(0)
for seed in [int(x) for x in sys.argv[1:]]:
    f0, im, diameter, kw = _cases.random_case(seed)
    prep = cta.prepare_batch(f0, im, diameter, compute_error=True, **kw)
    prep.problem.flags |= _abi.FLAG_THROUGHPUT
    b = prep.batch
    ref = _abi.HostBatch(b.frames, b.frame_index, b.feat_offset, b.params, b.low, b.high, want_std=True)
    eng.refine_batch(prep.problem, b)
    ctr_oracle.run_batch(prep.problem, ref, 4)
    fa, fb = np.isfinite(b.params_std), np.isfinite(ref.params_std)
    print('seed', seed, kw.get('param_mode'), 'clusters', b.n_clusters, 'std entries', fa.size, 'finite here', fa.sum(), 'there', fb.sum(), 'mismatch', (fa != fb).sum())
    both = fa & fb
    rel = np.abs(b.params_std - ref.params_std)[both] / np.abs(ref.params_std[both])
    print('   rel max %.2e  q99 %.2e' % (rel.max() if rel.size else 0, np.quantile(rel, 0.99) if rel.size else 0))
    rows = np.flatnonzero((fa != fb).any(1) | ((np.abs(b.params_std - ref.params_std) > 1e-4 * np.abs(ref.params_std)) & both).any(1))
    cl = np.searchsorted(b.feat_offset, rows, side='right') - 1
    for r, c in list(zip(rows, cl))[:6]:
        print('   row', r, 'cluster', c, 'n', b.feat_offset[c + 1] - b.feat_offset[c], 'status', b.status[c], ref.status[c], 'iters', b.n_iter[c], ref.n_iter[c])
        print('      here ', b.params_std[r]); print('      there', ref.params_std[r])
