"""Diagnostic: cycle shares of the block kernel's sections (needs the -DCTR_STAMPS build).
    CTREFINE_LIB=tools/_stamps/libctrefine_stamps.so python tests/tools/stamps_run.py [min_n] [max_n]
"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import clustertracking_amd as cta
from clustertracking_amd import workloads, _abi, _lib

lo_n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
hi_n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
wl = os.environ.get('WL', 'cfg2')
if wl == 'cfg2':
    frames, f0, truth, opts = workloads.cfg2(256, 0)
    prep = cta.prepare_batch(f0, cta.ArrayReader(frames), 13)
elif wl == 'cfg5':
    frames, f0, truth, opts = workloads.cfg5(64, 0)
    prep = cta.prepare_batch(f0, cta.ArrayReader(frames), 13, constraints=cta.constraints.dimer(6., 2))
else:
    frames, f0, truth, opts = workloads.cfg3(64, 0, n_features=40)
    prep = cta.prepare_batch(f0, cta.ArrayReader(frames), opts['diameter'])
hb = prep.batch
sz = np.diff(hb.feat_offset)
sel = np.flatnonzero((sz >= lo_n) & (sz <= hi_n))
if len(sys.argv) > 3:   # only the slow clusters: every block alone on its CU
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'oracle'))
    import ctr_oracle
    ctr_oracle.run_batch(prep.problem, hb, 16)
    sel = np.flatnonzero((sz >= lo_n) & (sz <= hi_n) & (hb.n_iter >= int(sys.argv[3])))
rows = np.concatenate([np.arange(hb.feat_offset[c], hb.feat_offset[c + 1]) for c in sel])
off = np.concatenate([[0], np.cumsum(sz[sel])])
sub = _abi.HostBatch(hb.frames, hb.frame_index[sel], off, hb.params[rows], hb.low[rows], hb.high[rows])
if os.environ.get('THROUGHPUT'):
    prep.problem.flags |= _abi.FLAG_THROUGHPUT
eng = _lib.default_engine(0)
lib = _lib.load()
buf = (ctypes.c_ulonglong * 16)()
eng.refine_batch(prep.problem, sub)
lib.ctr_debug_stamps(buf, 1)
eng.refine_batch(prep.problem, sub)
lib.ctr_debug_stamps(buf, 0)
tot = sum(buf[:15])
names = ['eval+park', 'barrier1', 'combine+accept', 'active set', 'solve', 'step/pred', 'round/fpar', 'barrier2']
print('clusters', len(sel), 'iterations', int(sub.n_iter.sum()), 'max', int(sub.n_iter.max()))
names += ['(s8) solve prologue', '(s9) factor+solve', '(s10) projected step', '(s11) model decrease',
          '(s12) build H (LDS path)', '(s13) factor (LDS path)', '(s14) substitutions (LDS path)']
for n, v in zip(names, list(buf[:8]) + list(buf[8:15])):
    print('%-16s %12d cycles  %5.1f %%   %8.0f cycles/iteration' % (n, v, 100. * v / tot, v / sub.n_iter.sum()))
print('total cycles/iteration %.0f' % (tot / sub.n_iter.sum()))
