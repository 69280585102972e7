"""BASELINE cfg 3 at its stated density through the HIP engine (host-buffer call): time, statuses,
rms vs truth; stack 0 against the stored oracle vector (tests/golden/cfg3_500_oracle.npz).
    python tests/tools/run_cfg3.py [stacks] [features] [sizevar|default] [maxiter]     (sizevar: param_mode=dict(size='var'))"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
for p in (ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'oracle')):
    sys.path.insert(0, p)
import numpy as np
import clustertracking_amd as cta
from clustertracking_amd import workloads, _lib

stacks = int(sys.argv[1]) if len(sys.argv) > 1 else 1
nf = int(sys.argv[2]) if len(sys.argv) > 2 else 500
frames, f0, truth, opts = workloads.cfg3(stacks, 0, n_features=nf)
sizevar = len(sys.argv) > 3 and sys.argv[3] == 'sizevar'
prep = cta.prepare_batch(f0, cta.ArrayReader(frames), opts['diameter'], param_mode=dict(size='var') if sizevar else None,
                         **(dict(solver_maxiter=int(sys.argv[4])) if len(sys.argv) > 4 else {}))
if os.environ.get('CFG3_THROUGHPUT'):
    from clustertracking_amd import _abi
    prep.problem.flags |= _abi.FLAG_THROUGHPUT      # one workgroup per large cluster (no helpers)
b = prep.batch
eng = _lib.default_engine(0)
eng.refine_batch(prep.problem, b)
t0 = time.perf_counter(); eng.refine_batch(prep.problem, b); dt = time.perf_counter() - t0
n_per = np.diff(b.feat_offset)
print('stacks %d features %d clusters %d largest %s: %.3f s (%.0f features/s)' % (
    stacks, len(f0), b.n_clusters, np.sort(n_per)[-3:], dt, len(f0) / dt))
import ctypes
dbg = (ctypes.c_ulonglong * 48)()
if hasattr(eng._lib, 'ctr_debug_large_counters') and eng._lib.ctr_debug_large_counters(dbg, 1) == 0:
    print('large path (both runs): solves %d, CG iterations %d (%.1f per solve), pixel passes %d' % (
        dbg[0], dbg[1], dbg[1] / max(dbg[0], 1), dbg[2]))
    print('  time in pixel passes %.3f s, in solves %.3f s of which matrix-vector products %.3f s (summed over clusters)' % (dbg[4] * 1e-8, dbg[5] * 1e-8, dbg[3] * 1e-8))
    print('  wave 0: feature tiles %.3f s, pair blocks %.3f s' % (dbg[6] * 1e-8, dbg[7] * 1e-8))
    for kq, name in enumerate(('exact Hessian, converged', 'exact Hessian, not positive definite', 'J^T J, converged', 'J^T J, failed')):
        ns, ni, ncap = dbg[8 + 4 * kq], dbg[9 + 4 * kq], dbg[10 + 4 * kq]
        print('  %-38s %6d solves, %8d CG iterations (%.1f per solve), %d at the iteration cap' % (name, ns, ni, ni / max(ns, 1), ncap))
if dbg[27]:
    print('  per feature tile of the leader\'s wave 0 (%d tiles; shader-clock ticks): neighbour visits (per tile of the feature) %.0f; own model %.0f; rows -> LDS, 16 MFMA %.0f' % (
        dbg[27], dbg[24] / dbg[27], dbg[25] / dbg[27], dbg[26] / dbg[27]))
if dbg[38]:
    print('  %d rounds: neighbour lists %.3f s, pixel / pair lists %.3f s, aggregates %.3f s; accept / tabulate %.3f s, factorisations %.3f s (feature blocks %.3f, aggregates %.3f, their inverses %.3f)' % (
        dbg[38], dbg[32] * 1e-8, dbg[33] * 1e-8, dbg[34] * 1e-8, dbg[35] * 1e-8, dbg[37] * 1e-8, dbg[39] * 1e-8, dbg[40] * 1e-8, (dbg[37] - dbg[39] - dbg[40]) * 1e-8))
if dbg[29]:
    print('  neighbour visits of wave 0: %d with a block (j > i): %.0f ticks each, of which MFMA %.0f, + flush %.0f; %d without: %.0f each' % (
        dbg[29], dbg[28] / dbg[29], dbg[15] / dbg[29], dbg[11] / dbg[29], dbg[31], dbg[30] / max(dbg[31], 1)))
print('status counts', np.bincount(b.status), 'rounds', b.n_rounds[:8], 'iters', b.n_iter[:8])
out = np.empty_like(b.params_out); out[prep.order] = b.params_out
ok = np.empty(len(out), bool); ok[prep.order] = np.repeat(b.status == 0, n_per)
print('rms vs truth %.4f px over %d features' % (np.sqrt(np.mean((out[ok, 2:5] - truth[ok]) ** 2)), ok.sum()))
gold = os.path.join(ROOT, 'tests', 'golden', 'cfg3_500_oracle.npz')
if nf == 500 and not sizevar and os.path.exists(gold):
    z = np.load(gold)
    n0 = int(z['feat_offset'][-1])
    d = np.abs(b.params_out[:n0, 2:5] - z['params_out'][:, 2:5]).max()
    print('stack 0 vs oracle vector: status %s/%s rounds %s/%s iters %s/%s cost %.12f/%.12f max dpos %.2e' % (
        b.status[0], z['status'][0], b.n_rounds[0], z['n_rounds'][0], b.n_iter[0], z['n_iter'][0],
        b.cost[0], z['cost'][0], d))
