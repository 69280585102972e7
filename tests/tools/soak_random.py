"""Soak test: many random configurations, engine vs C oracle (tests/test_gpu_parity.py's
generator with more seeds).   python tests/tools/soak_random.py [n_seeds] [first_seed]
Environment: SOAK_CONSTRAINED_ONLY, SOAK_THROUGHPUT / SOAK_ISOLATE (scheduling flags), SOAK_LOWPASS (a random
noise_size / threshold per configuration), SOAK_STD (also compare params_std)."""
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
n_std = 0
for seed in range(first, first + n_seeds):
    f0, im, diameter, kw = _cases.random_case(seed)
    if os.environ.get('SOAK_CONSTRAINED_ONLY') and 'constraints' not in kw:
        continue
    if os.environ.get('SOAK_LOWPASS'):
        r2 = np.random.RandomState(seed + 77)
        nd_ = im.ndim
        kw['noise_size'] = float(r2.choice([0.5, 1., 1.5, 2.])) if r2.rand() < 0.6 else \
            tuple(float(x) for x in r2.choice([0., 0.5, 1., 1.5], nd_))
        if not np.any(np.asarray(kw['noise_size']) > 0):
            kw['noise_size'] = 1.
        if r2.rand() < 0.4:
            kw['threshold'] = float(r2.uniform(1., 12.))
    want_std = bool(os.environ.get('SOAK_STD'))
    prep = cta.prepare_batch(f0, im, diameter, compute_error=want_std, **kw)
    if os.environ.get('SOAK_THROUGHPUT'):
        prep.problem.flags |= _abi.FLAG_THROUGHPUT   # scheduling flag: same results expected
    if os.environ.get('SOAK_ISOLATE'):
        prep.problem.flags |= _abi.FLAG_ISOLATE_TAIL
    b = prep.batch
    ref = _abi.HostBatch(b.frames, b.frame_index, b.feat_offset, b.params, b.low, b.high, want_std=want_std)
    eng.refine_batch(prep.problem, b)
    ctr_oracle.run_batch(prep.problem, ref, 4)
    nd = im.ndim
    n_clusters += b.n_clusters
    # clusters of the large-cluster kernel (> 64 features or > 127 variables): no params_std there
    # (include/ctrefine.h) -- a documented limit, not a difference
    modes_ = np.array(list(prep.problem.modes)[:prep.problem.n_params])
    nvar = (modes_ == 3).sum() + np.diff(b.feat_offset) * (modes_ == 1).sum()
    large = (np.diff(b.feat_offset) > 64) | (nvar > 127)
    same_status = (b.status == ref.status).all()
    ok = (ref.status == 0) & (b.status == 0)
    cost_ok = np.allclose(b.cost[ok], ref.cost[ok], rtol=1e-7, atol=1e-12)
    rows = np.repeat(ok, np.diff(ref.feat_offset))
    d = np.abs(b.params_out[:, 2:2 + nd] - ref.params_out[:, 2:2 + nd])[rows]
    dmax = d.max() if d.size else 0.
    worst = max(worst, dmax)
    std_ok = True
    if want_std and same_status:
        small_rows = np.repeat(~large, np.diff(b.feat_offset))
        fa, fb = np.isfinite(b.params_std[small_rows]), np.isfinite(ref.params_std[small_rows])
        b_std, r_std = b.params_std[small_rows], ref.params_std[small_rows]
        both = fa & fb
        rel = np.abs(b_std - r_std)[both] / np.abs(r_std[both]) if both.any() else np.zeros(1)
        # (a Hessian on the edge of positive definiteness may be NaN on one side only)
        std_ok = fa.size == 0 or ((fa != fb).mean() < 0.02 and np.quantile(rel, 0.99) < 1e-4)
        n_std += int(both.sum())
    if not same_status or not cost_ok or dmax > 1e-3 or not std_ok:
        bad += 1
        print('seed %d: status-eq %s cost-ok %s std-ok %s dmax %.2e modes %s cons %s lowpass %s' % (
            seed, same_status, cost_ok, std_ok, dmax, kw.get('param_mode'), 'constraints' in kw, kw.get('noise_size')))
print('seeds %d clusters %d bad %d worst position difference %.2e px; std values compared %d' % (n_seeds, n_clusters, bad, worst, n_std))
