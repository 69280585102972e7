"""Engine vs C oracle on a cfg2 block, cluster by cluster: status, solver iterations,
re-window rounds and parameters.  Equal iteration counts mean the two implementations
walk the same path, not only that they end in the same minimum.

    python tests/tools/iter_parity.py [frames] [workload]      # needs the MI355X
"""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
for p in (ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'oracle')):
    sys.path.insert(0, p)
import clustertracking_amd as cta  # noqa: E402
from clustertracking_amd import workloads, _lib  # noqa: E402
import ctr_oracle  # noqa: E402


def main():
    frames_n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    wl = sys.argv[2] if len(sys.argv) > 2 else 'cfg2'
    frames, f0, truth, opts = getattr(workloads, wl)(frames_n)
    extra = {k: v for k, v in opts.items() if k != 'diameter'}
    prep = cta.prepare_batch(f0, cta.ArrayReader(frames), opts['diameter'], **extra)
    b = prep.batch
    eng = _lib.Engine()
    eng.refine_batch(prep.problem, b)
    e = dict(status=b.status.copy(), it=b.n_iter.copy(), rounds=b.n_rounds.copy(),
             cost=b.cost.copy(), par=b.params_out.copy())
    ctr_oracle.run_batch(prep.problem, b, os.cpu_count() or 1)
    size = np.diff(b.feat_offset)
    print('%d clusters; status equal: %s; failed engine/oracle: %d/%d' % (
        len(size), bool((e['status'] == b.status).all()), (e['status'] != 0).sum(), (b.status != 0).sum()))
    print('mean iterations engine %.3f oracle %.3f; max %d / %d' % (
        e['it'].mean(), b.n_iter.mean(), e['it'].max(), b.n_iter.max()))
    for lo, hi in ((1, 1), (2, 2), (3, 4), (5, 1000)):
        sel = (size >= lo) & (size <= hi)
        if not sel.any():
            continue
        same_it = (e['it'][sel] == b.n_iter[sel]).mean()
        same_r = (e['rounds'][sel] == b.n_rounds[sel]).mean()
        rows = np.repeat(sel, size)
        ok = np.repeat((e['status'] == 0) & (b.status == 0), size) & rows
        dp = np.abs(e['par'][ok] - b.params_out[ok])
        print('size %d-%d: %6d clusters, same iterations %.4f, same rounds %.4f, max |dparam| %.2e, '
              'max |dcost| %.2e' % (lo, hi, sel.sum(), same_it, same_r, dp.max() if dp.size else 0.,
                                    np.nanmax(np.abs(e['cost'][sel] - b.cost[sel]))))
        bad = np.flatnonzero(sel & (e['it'] != b.n_iter))[:5]
        for c in bad:
            print('    cluster %d: it %d/%d rounds %d/%d status %d/%d' % (
                c, e['it'][c], b.n_iter[c], e['rounds'][c], b.n_rounds[c], e['status'][c], b.status[c]))


if __name__ == '__main__':
    main()
