"""Refine-stage time of each size class of cfg2 when it runs alone on the GPU (diagnostic)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import clustertracking_amd as cta
from clustertracking_amd import workloads, _abi, _lib
from clustertracking_amd.device import DeviceBatch

frames, f0, truth, opts = workloads.cfg2(256, 0)
prep = cta.prepare_batch(f0, cta.ArrayReader(frames), 13)
hb = prep.batch
sz = np.diff(hb.feat_offset)
eng = _lib.default_engine(0)
import torch
import itertools
classes = [(1, 1), (2, 2), (3, 4), (5, 100)]
sets = [[c] for c in classes] + [list(x) for x in itertools.combinations(classes, 2)] + [list(x) for x in itertools.combinations(classes, 3)] + [classes]
for cls in sets:
    mask = np.zeros(len(sz), bool)
    for lo_n, hi_n in cls:
        mask |= (sz >= lo_n) & (sz <= hi_n)
    sel = np.flatnonzero(mask)
    rows = np.concatenate([np.arange(hb.feat_offset[c], hb.feat_offset[c + 1]) for c in sel])
    off = np.concatenate([[0], np.cumsum(sz[sel])])
    sub = _abi.HostBatch(hb.frames, hb.frame_index[sel], off, hb.params[rows], hb.low[rows], hb.high[rows])
    db = DeviceBatch(prep.problem, sub, device=0, engine=eng)
    ts = []
    for _ in range(6):
        db.run(); eng.synchronize(None); torch.cuda.synchronize()
        ts.append(eng.last_kernel_ms()[1])
    db.download()
    print('sizes %-40s %6d clusters, %7d iterations (max %d): refine stage %.3f ms' % (
        '+'.join('%d-%d' % c for c in cls) + ':', len(sel), sub.n_iter.sum(), sub.n_iter.max(), np.median(ts[1:])))
