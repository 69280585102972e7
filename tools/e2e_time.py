"""End-to-end wall time of the drop-in call (host prep + PCIe + kernels + write-back)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import clustertracking_amd as cta
from clustertracking_amd import workloads, _lib

frames, f0, truth, opts = workloads.cfg2(256, 0)
reader = cta.ArrayReader(frames)
eng = _lib.default_engine(0)
for rep in range(6):
    t0 = time.perf_counter()
    prep = cta.prepare_batch(f0.copy(), reader, 13, cluster_labels='device' if rep >= 3 else 'reference')
    t1 = time.perf_counter()
    eng.refine_batch(prep.problem, prep.batch)
    t2 = time.perf_counter()
    res = cta.write_back(prep)
    t3 = time.perf_counter()
    n = prep.batch.n_clusters
    print(('device labels ' if rep >= 3 else 'host labels   ') + 'rep %d: prepare %.1f ms (find_clusters + bounds + CSR), ctr_refine_batch incl. PCIe %.1f ms, '
          'write_back %.1f ms, total %.1f ms -> %.0f cluster-fits/s end to end; engine call alone %.0f fits/s'
          % (rep, 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), 1e3 * (t3 - t0), n / (t3 - t0), n / (t2 - t1)))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); prep = cta.prepare_batch(f0.copy(), reader, 13); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(12)
