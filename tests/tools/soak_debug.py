"""Details of one soak seed: python tests/tools/soak_debug.py seed [lowpass] """
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
for p in (ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'oracle')):
    sys.path.insert(0, p)
import numpy as np
import clustertracking_amd as cta
from clustertracking_amd import _abi, _lib
import ctr_oracle
import _cases
np.set_printoptions(linewidth=200, precision=6)
eng = _lib.default_engine(0)
for seed in [int(x) for x in sys.argv[1].split(',')]:
    f0, im, diameter, kw = _cases.random_case(seed)
    if len(sys.argv) > 2:
        r2 = np.random.RandomState(seed + 77)
        nd_ = im.ndim
        kw['noise_size'] = float(r2.choice([0.5, 1., 1.5, 2.])) if r2.rand() < 0.6 else \
            tuple(float(x) for x in r2.choice([0., 0.5, 1., 1.5], nd_))
        if not np.any(np.asarray(kw['noise_size']) > 0):
            kw['noise_size'] = 1.
        if r2.rand() < 0.4:
            kw['threshold'] = float(r2.uniform(1., 12.))
    prep = cta.prepare_batch(f0, im, diameter, compute_error=True, **kw)
    b = prep.batch
    ref = _abi.HostBatch(b.frames, b.frame_index, b.feat_offset, b.params, b.low, b.high, want_std=True)
    eng.refine_batch(prep.problem, b)
    ctr_oracle.run_batch(prep.problem, ref, 4)
    print('seed', seed, 'ndim', im.ndim, 'dtype', im.dtype, 'diameter', diameter, {k: v for k, v in kw.items() if k != 'constraints'}, 'cons' if 'constraints' in kw else '')
    sz = np.diff(b.feat_offset)
    for c in range(b.n_clusters):
        rows = slice(b.feat_offset[c], b.feat_offset[c + 1])
        fa, fb = np.isfinite(b.params_std[rows]), np.isfinite(ref.params_std[rows])
        both = fa & fb
        rel = (np.abs(b.params_std[rows] - ref.params_std[rows])[both] / np.abs(ref.params_std[rows][both])).max() if both.any() else 0.
        nd_ = im.ndim
        dpos = np.abs(b.params_out[rows][:, 2:2 + nd_] - ref.params_out[rows][:, 2:2 + nd_]).max()
        flag = b.status[c] != ref.status[c] or (fa != fb).any() or rel > 1e-4 or dpos > 1e-6
        if flag:
            print(' cluster', c, 'n', sz[c], 'status', b.status[c], ref.status[c], 'iters', b.n_iter[c], ref.n_iter[c], 'rounds', b.n_rounds[c], ref.n_rounds[c],
                  'cost', b.cost[c], ref.cost[c], 'std finite', fa.sum(), fb.sum(), 'rel', rel, 'dpos %.2e' % dpos)
            if sz[c] <= 4:
                print('  engine std', b.params_std[rows]); print('  oracle std', ref.params_std[rows])
                print('  engine out', b.params_out[rows]); print('  oracle out', ref.params_out[rows])
