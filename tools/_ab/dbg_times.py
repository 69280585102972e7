import ctypes, os, sys
sys.path.insert(0, '/root/repo')
import numpy as np
import clustertracking_amd as cta
from clustertracking_amd import workloads, _abi, _lib
from clustertracking_amd.device import DeviceBatch
import torch
frames, f0, truth, opts = workloads.cfg2(256, 0)
prep = cta.prepare_batch(f0, cta.ArrayReader(frames), 13)
eng = _lib.default_engine(0); lib = _lib.load()
db = DeviceBatch(prep.problem, prep.batch, device=0, engine=eng)
buf = (ctypes.c_ulonglong * 16)()
for rep in range(3):
    lib.ctr_debug_stamps(buf, 1)
    db.run(); eng.synchronize(None); torch.cuda.synchronize()
    ms = eng.last_kernel_ms()[1]
    lib.ctr_debug_stamps(buf, 0)
    print('refine stage %.3f ms; straggler block index %d; starts %.3f ms after the first block; runs %.3f ms; last block ends %.3f ms after the first block started' % (
        ms, buf[11], (buf[12]-buf[14])/1e5, (buf[13]-buf[12])/1e5, (buf[15]-buf[14])/1e5))

st = (ctypes.c_ulonglong * 4096)()
lib.ctr_debug_starts(st)
a = (np.array(st[:1645], dtype=np.float64) - float(buf[14])) / 1e5
print('NT=1 block start times (ms after first block) by block index:')
for lo in range(0, 1645, 100):
    seg = a[lo:lo + 100]
    print('  blocks %4d-%4d: min %.3f median %.3f max %.3f' % (lo, lo + len(seg) - 1, seg.min(), np.median(seg), seg.max()))
