"""One golden fixture through the HIP engine and the C oracle, printed side by side.
    python tests/tools/run_fixture.py <name> [...]"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
for p in (ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'oracle')):
    sys.path.insert(0, p)
import numpy as np
import _cases, ctr_oracle
from clustertracking_amd import _abi, _lib

eng = _lib.default_engine(0)
for name in sys.argv[1:]:
    case = _cases.Case(name)
    prep = case.prepare()
    b = prep.batch
    ref = _abi.HostBatch(b.frames, b.frame_index, b.feat_offset, b.params, b.low, b.high)
    t0 = time.perf_counter(); eng.refine_batch(prep.problem, b); t1 = time.perf_counter()
    ctr_oracle.run_batch(prep.problem, ref, 8); t2 = time.perf_counter()
    nd = len(case.pos_columns)
    n_per = np.diff(b.feat_offset)
    print(name, 'engine %.3f s oracle %.3f s' % (t1 - t0, t2 - t1))
    for c in range(b.n_clusters):
        sl = slice(b.feat_offset[c], b.feat_offset[c + 1])
        d = np.abs(b.params_out[sl, 2:2 + nd] - ref.params_out[sl, 2:2 + nd]).max()
        if n_per[c] > 4 or b.status[c] != ref.status[c] or d > 1e-7:
            print('  cluster %d n=%d status %d/%d rounds %d/%d iters %d/%d cost %.10f/%.10f dpos %.2e' % (
                c, n_per[c], b.status[c], ref.status[c], b.n_rounds[c], ref.n_rounds[c], b.n_iter[c], ref.n_iter[c],
                b.cost[c], ref.cost[c], d))
