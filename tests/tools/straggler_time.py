"""Latency of the slowest clusters when each runs alone on the GPU."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'oracle'))
import numpy as np
import clustertracking_amd as cta
from clustertracking_amd import workloads, _abi, _lib
import ctr_oracle

frames, f0, truth, opts = workloads.cfg2(256, 0)
prep = cta.prepare_batch(f0, cta.ArrayReader(frames), 13)
hb = prep.batch
ctr_oracle.run_batch(prep.problem, hb, 16)
sz = np.diff(hb.feat_offset)
eng = _lib.default_engine(0)
for lo_n, hi_n in ((1, 1), (2, 2), (3, 4), (5, 8)):
    cand = np.flatnonzero((sz >= lo_n) & (sz <= hi_n))
    c = cand[np.argmax(hb.n_iter[cand])]
    rows = np.arange(hb.feat_offset[c], hb.feat_offset[c + 1])
    fi = hb.frame_index[c]
    sub = _abi.HostBatch(hb.frames[fi:fi + 1], [0], [0, len(rows)], hb.params[rows], hb.low[rows], hb.high[rows])
    eng.refine_batch(prep.problem, sub)
    from clustertracking_amd.device import DeviceBatch
    db = DeviceBatch(prep.problem, sub, device=0, engine=eng)
    ts = []
    for _ in range(5):
        db.run(); eng.synchronize(None)
        import torch; torch.cuda.synchronize()
        ts.append(eng.last_kernel_ms()[1])
    db.download()
    print('n=%d cluster %d: %d iterations, %d rounds, status %d: refine stage %.3f ms -> %.2f us / iteration'
          % (sz[c], c, sub.n_iter[0], sub.n_rounds[0], sub.status[0], np.median(ts), 1e3 * np.median(ts) / sub.n_iter[0]))
