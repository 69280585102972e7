"""cfg 3 at its stated density: which clusters end at the iteration limit, and do more solver
iterations per round (options=dict(maxiter=...)) fit them?   python tests/tools/cfg3_failures.py [stacks]"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
for p in (ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'oracle')):
    sys.path.insert(0, p)
import numpy as np
import clustertracking_amd as cta
from clustertracking_amd import workloads, _lib

stacks = int(sys.argv[1]) if len(sys.argv) > 1 else 64
frames, f0, truth, opts = workloads.cfg3(stacks, 0)
eng = _lib.default_engine(0)
for maxiter in (100, 200, 400):
    prep = cta.prepare_batch(f0, cta.ArrayReader(frames), opts['diameter'], solver_maxiter=maxiter)
    b = prep.batch
    t0 = time.perf_counter(); eng.refine_batch(prep.problem, b); dt = time.perf_counter() - t0
    n_per = np.diff(b.feat_offset)
    bad = np.flatnonzero(b.status != 0)
    print('maxiter %d: %.2f s, failed %d of %d clusters' % (maxiter, dt, len(bad), b.n_clusters))
    for c in bad:
        print('   cluster %d: %d features, status %d, rounds %d, iterations %d' % (c, n_per[c], b.status[c], b.n_rounds[c], b.n_iter[c]))
    big = np.flatnonzero((n_per > 64) & (b.status == 0))
    print('   fitted large clusters: rounds median %d max %d, iterations median %d max %d' % (
        np.median(b.n_rounds[big]), b.n_rounds[big].max(), np.median(b.n_iter[big]), b.n_iter[big].max()))
    out = np.empty_like(b.params_out); out[prep.order] = b.params_out
    ok = np.empty(len(out), bool); ok[prep.order] = np.repeat(b.status == 0, n_per)
    print('   rms vs truth %.4f px over %d features' % (np.sqrt(np.mean((out[ok, 2:5] - truth[ok]) ** 2)), ok.sum()))
