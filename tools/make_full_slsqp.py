"""The reference's algorithm (oracle/ref_numpy.py: NumPy objective + SciPy SLSQP) over the WHOLE
bench workload (cfg 2: 256 frames, ~41 k clusters), with the default tolerance (tol=1e-6, run A)
and converged (tol=1e-14, maxiter=1000, run B).  CPU only, a pool of processes over frame blocks;
writes tests/golden/cfg2_full_slsqp.npz (positions, cost, status of both runs, per row / cluster
of the prepared batch).  tools/full_parity.py compares the engine with it on the GPU box.
    python tools/make_full_slsqp.py [frames] [workers]"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
for p in (ROOT, os.path.join(ROOT, 'oracle')):
    sys.path.insert(0, p)
import numpy as np


def work(args):
    first, count = args
    import clustertracking_amd as cta
    from clustertracking_amd import workloads, _abi
    import ref_numpy
    frames, f0, truth, opts = workloads.cfg2(count, first_seed=first)
    prep = cta.prepare_batch(f0, cta.ArrayReader(frames), opts['diameter'])
    b = prep.batch
    out = {}
    for tag, kw in (('A', {}), ('B', dict(tol=1e-14, maxiter=1000))):
        hb = _abi.HostBatch(b.frames, b.frame_index, b.feat_offset, b.params, b.low, b.high)
        ref_numpy.run_batch(prep.problem, hb, **kw)
        out[tag] = (hb.params_out[:, 2:4].copy(), hb.cost.copy(), hb.status.copy())
    return first, out, np.diff(b.feat_offset)


if __name__ == '__main__':
    n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    workers = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    import multiprocessing as mp
    block = 4
    jobs = [(f, min(block, n_frames - f)) for f in range(0, n_frames, block)]
    t0 = time.time()
    with mp.get_context('spawn').Pool(workers) as pool:
        res = sorted(pool.imap_unordered(work, jobs), key=lambda r: r[0])
    z = {}
    for tag in 'AB':
        z['pos_' + tag] = np.concatenate([r[1][tag][0] for r in res])
        z['cost_' + tag] = np.concatenate([r[1][tag][1] for r in res])
        z['status_' + tag] = np.concatenate([r[1][tag][2] for r in res]).astype(np.int8)
    z['n_per_cluster'] = np.concatenate([r[2] for r in res]).astype(np.int16)
    z['n_frames'] = np.int64(n_frames)
    np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'cfg2_full_slsqp.npz'), **z)
    print('%d clusters, %d rows, %.0f s' % (len(z['cost_A']), len(z['pos_A']), time.time() - t0))
