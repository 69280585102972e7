import sys
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import clustertracking_amd as cta
from clustertracking_amd import workloads, _lib
from clustertracking_amd.device import DeviceBatch
frames, f0, truth, opts = workloads.cfg2(64, 0)
prep = cta.prepare_batch(f0, cta.ArrayReader(frames), 13)
db = DeviceBatch(prep.problem, prep.batch, device=0)
outs = []
for rep in range(4):
    db.run(); torch.cuda.synchronize()
    outs.append((db.t['params_out'].clone(), db.t['cost'].clone(), db.t['n_iter'].clone()))
sz = np.diff(prep.batch.feat_offset)
for rep in range(1, 4):
    dp = (outs[rep][0] != outs[0][0]).any(1).cpu().numpy()
    rows_cluster = np.repeat(np.arange(len(sz)), sz)
    bad = np.unique(rows_cluster[dp])
    print('run %d vs run 0: rows differing %d, clusters %d, sizes %s, max |d| %.3e, n_iter equal %s' % (
        rep, dp.sum(), len(bad), np.bincount(sz[bad])[:8] if len(bad) else [], (outs[rep][0] - outs[0][0]).abs().max().item(),
        bool((outs[rep][2] == outs[0][2]).all())))
