import os, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
import clustertracking_amd as cta
from clustertracking_amd import workloads, _lib
frames, f0, truth, opts = workloads.cfg2(256, 0)
reader = cta.ArrayReader(frames)
eng = _lib.default_engine(0)
for rep in range(4):
    t0 = time.perf_counter()
    prep = cta.prepare_batch(f0.copy(), reader, 13, cluster_labels='device')
    t1 = time.perf_counter()
    eng.refine_batch(prep.problem, prep.batch)
    t2 = time.perf_counter()
    res = cta.write_back(prep)
    t3 = time.perf_counter()
    print('prepare %.2f refine %.2f write_back %.2f total %.2f ms' % (1e3*(t1-t0), 1e3*(t2-t1), 1e3*(t3-t2), 1e3*(t3-t0)))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(5): prep = cta.prepare_batch(f0.copy(), reader, 13, cluster_labels='device')
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(18)
pr = cProfile.Profile(); pr.enable()
for _ in range(5): eng.refine_batch(prep.problem, prep.batch)
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(10)
