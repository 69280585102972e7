"""BASELINE cfg 4, one GPU's shard: 1250 frames of the 10 000-frame video through the
drop-in call (device cluster labels), then linking of the refined coordinates on the host.

    python tools/run_cfg4_shard.py [n_frames]
"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import clustertracking_amd as cta
from clustertracking_amd import workloads

n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 1250
t0 = time.perf_counter()
frames, f0, truth, opts = workloads.cfg2(n_frames, 0)
print('synthesised %d frames (%.1f MiB) in %.1f s' % (n_frames, frames.nbytes / 2**20, time.perf_counter() - t0))
reader = cta.ArrayReader(frames)
cta.refine_leastsq(f0.iloc[:400].copy(), reader, 13, cluster_labels='device')   # warm-up (library load, buffers)
for labels in ('reference', 'device'):
    t0 = time.perf_counter()
    res = cta.refine_leastsq(f0.copy(), reader, 13, cluster_labels=labels)
    dt = time.perf_counter() - t0
    n_cl = res.groupby(['frame', 'cluster']).ngroups
    ok = np.isfinite(res['cost'].values)
    err = np.sqrt(np.mean((res[['y', 'x']].values - truth)[ok] ** 2))
    print('refine_leastsq, cluster_labels=%-9s: %.3f s for %d features / %d clusters -> %.0f cluster-fits/s '
          'end to end; %d failed; rms error vs truth %.4f px' % (labels, dt, len(res), n_cl, n_cl / dt,
                                                                  int((~ok).sum()), err))
t0 = time.perf_counter()
tracks = cta.link_df(res[ok][['y', 'x', 'frame']], search_range=5.)
print('link: %.2f s, %d tracks' % (time.perf_counter() - t0, tracks['particle'].nunique()))
