"""Machine time of each size class of cfg 2: refine stage of the singles / pairs / 3-4 / 5+
feature clusters ALONE, with default scheduling and with CTR_FLAG_THROUGHPUT, plus the
wave-occupancy-limited sum.  (With batches in flight a step costs about the sum of the classes'
alone-times under the throughput flag.)"""
import os, sys, copy
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import clustertracking_amd as cta
from clustertracking_amd import workloads, _abi, _lib
from clustertracking_amd.device import DeviceBatch
import torch

frames, f0, truth, opts = workloads.cfg2(256, 0)
prep = cta.prepare_batch(f0, cta.ArrayReader(frames), 13)
hb = prep.batch
sz = np.diff(hb.feat_offset)
eng = _lib.default_engine(0)
for tp in (0, 1):
    prob = copy.copy(prep.problem)
    if tp:
        prob.flags |= _abi.FLAG_THROUGHPUT
    out = []
    for lo_n, hi_n in [(1, 1), (2, 2), (3, 4), (5, 100), (1, 100)]:
        sel = np.flatnonzero((sz >= lo_n) & (sz <= hi_n))
        rows = np.concatenate([np.arange(hb.feat_offset[c], hb.feat_offset[c + 1]) for c in sel])
        off = np.concatenate([[0], np.cumsum(sz[sel])])
        sub = _abi.HostBatch(hb.frames, hb.frame_index[sel], off, hb.params[rows], hb.low[rows], hb.high[rows])
        db = DeviceBatch(prob, sub, device=0, engine=eng)
        ts = []
        for _ in range(8):
            db.run(); eng.synchronize(None); torch.cuda.synchronize()
            ts.append(eng.last_kernel_ms()[1])
        db.download()
        out.append('%d-%d (%d clusters, %d its): %.3f ms' % (lo_n, hi_n, len(sel), sub.n_iter.sum(), np.median(ts[2:])))
    print('throughput flag %d:  ' % tp + '   '.join(out))
