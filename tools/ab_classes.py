"""A/B of several builds (tools/_ab/lib_*.so) per size class of cfg 2: refine-stage time of the
singles / pairs / 3-4 / 5+ feature clusters alone, one process per build (CTREFINE_LIB)."""
import os, subprocess, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
if len(sys.argv) > 1 and sys.argv[1] == 'child':
    sys.path.insert(0, ROOT)
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
    out = []
    for lo_n, hi_n in [(1, 1), (2, 2), (3, 4), (5, 100), (1, 100)]:
        sel = np.flatnonzero((sz >= lo_n) & (sz <= hi_n))
        rows = np.concatenate([np.arange(hb.feat_offset[c], hb.feat_offset[c + 1]) for c in sel])
        off = np.concatenate([[0], np.cumsum(sz[sel])])
        sub = _abi.HostBatch(hb.frames, hb.frame_index[sel], off, hb.params[rows], hb.low[rows], hb.high[rows])
        db = DeviceBatch(prep.problem, sub, device=0, engine=eng)
        ts = []
        for _ in range(8):
            db.run(); eng.synchronize(None); torch.cuda.synchronize()
            ts.append(eng.last_kernel_ms()[1])
        out.append('%d-%d: %.3f' % (lo_n, hi_n, np.median(ts[2:])))
    print('  '.join(out))
else:
    import glob
    libs = sorted(glob.glob(os.path.join(ROOT, 'tools', '_ab', 'lib_*.so')))
    for rep in range(2):
        for lib in libs:
            env = dict(os.environ, CTREFINE_LIB=lib)
            r = subprocess.run([sys.executable, os.path.abspath(__file__), 'child'], env=env, capture_output=True, text=True, timeout=300)
            print(os.path.basename(lib), r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:], flush=True)
