"""Machine time per size class of cfg 2: ms per step with eight batches of ONE class in flight
(CTR_FLAG_THROUGHPUT), i.e. with the tails of the slow clusters hidden by the overlap.  The sum
over the classes is about what a step of the whole workload costs in bench.py."""
import os, sys, copy, time
os.environ.setdefault('GPU_MAX_HW_QUEUES', '20')
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
prob = copy.copy(prep.problem)
prob.flags |= _abi.FLAG_THROUGHPUT
engines = [_lib.Engine(0) for _ in range(8)]
tot = 0.
for lo_n, hi_n in [(1, 1), (2, 2), (3, 3), (4, 4), (3, 4), (5, 100), (1, 100)]:
    sel = np.flatnonzero((sz >= lo_n) & (sz <= hi_n))
    rows = np.concatenate([np.arange(hb.feat_offset[c], hb.feat_offset[c + 1]) for c in sel])
    off = np.concatenate([[0], np.cumsum(sz[sel])])
    sub = _abi.HostBatch(hb.frames, hb.frame_index[sel], off, hb.params[rows], hb.low[rows], hb.high[rows])
    dbs = [DeviceBatch(prob, sub, device=0, engine=e) for e in engines]
    for d in dbs:
        d.engine.refine_batch_device(d.plan, d.struct, 0)
    torch.cuda.synchronize()
    for e in engines: e.synchronize()
    steps = 48
    t0 = time.perf_counter()
    for k in range(steps):
        d = dbs[k % 8]
        d.engine.refine_batch_device(d.plan, d.struct, 0)
    for e in engines: e.synchronize()
    dt = (time.perf_counter() - t0) / steps * 1e3
    if (hi_n < 100 or lo_n > 1) and (lo_n, hi_n) not in ((3, 3), (4, 4)): tot += dt
    print('%d-%d features: %6d clusters  %.3f ms per step  (%.1f ns per cluster)' % (lo_n, hi_n, len(sel), dt, dt * 1e6 / len(sel)), flush=True)
    del dbs
print('sum of the classes %.3f ms' % tot)
