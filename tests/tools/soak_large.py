"""Soak test of the large-cluster path: random clusters of more than 64 features (2D and 3D, several
parameter modes, narrow and wide masks -- several segments of a feature's pixel list, overflowing
pair pools -- float frames with NaN pixels, a lowpass), engine vs C oracle.
    python tests/tools/soak_large.py [n_seeds] [first_seed]"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
for p in (ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'oracle')):
    sys.path.insert(0, p)
import numpy as np
import pandas as pd
import clustertracking_amd as cta
from clustertracking_amd import _abi, _lib, artificial
import ctr_oracle

MODES = [None, dict(size='var'), dict(signal='cluster'), dict(size='cluster', background='const'),
         dict(signal='cluster', size='var'), dict(background='const')]


def case(seed):
    rng = np.random.RandomState(seed)
    ndim = 3 if rng.rand() < 0.3 else 2
    if ndim == 2:
        size = float(rng.choice([2.5, 3., 4., 8.]))
        diameter = int(4 * size) | 1 if size < 8 else int(rng.choice([33, 51]))
        pitch = float(rng.uniform(1.6, 2.4)) * size
        ny, nx = int(rng.randint(8, 11)), int(rng.randint(9, 13))
        shape = (int(pitch * (ny + 1)), int(pitch * (nx + 1)))
        truth = np.array([[pitch * (1 + gy), pitch * (1 + gx)] for gy in range(ny) for gx in range(nx)])
        sizes = (size, size)
        iso = True
        cols = ['y', 'x']
    else:
        size = (2., 3., 3.)
        diameter = (9, 13, 13)
        pitch = np.array([4.5, 6.5, 6.5]) * float(rng.uniform(0.9, 1.1))
        nz, ny, nx = 3, int(rng.randint(5, 7)), int(rng.randint(5, 8))
        shape = tuple(int(p * (n + 1)) for p, n in zip(pitch, (nz, ny, nx)))
        truth = np.array([[pitch[0] * (1 + gz), pitch[1] * (1 + gy), pitch[2] * (1 + gx)]
                          for gz in range(nz) for gy in range(ny) for gx in range(nx)])
        sizes = size
        iso = False
        cols = ['z', 'y', 'x']
    truth = truth + rng.uniform(-0.15, 0.15, truth.shape) * np.asarray(pitch)
    im = np.zeros(shape, np.uint8)
    for p in truth:
        artificial.draw_gaussian(im, p, sizes if ndim == 3 else sizes[0], 100)
    im = artificial.add_poisson_noise(im, int(rng.choice([0, 6, 10])), rng)
    kw = {}
    what = rng.rand()
    if what < 0.25:
        im = im.astype(np.float64)
        k = int(rng.randint(3, 12))
        idx = tuple(rng.randint(0, s, k) for s in shape)
        im[idx] = np.nan                        # NaN pixels of the image are skipped (nansum)
    elif what < 0.4:
        im = im.astype(np.uint16) * 3
    elif what < 0.55 and ndim == 2:
        kw['noise_size'] = float(rng.choice([0.5, 1.]))
    f0 = pd.DataFrame(truth + rng.uniform(-0.4, 0.4, truth.shape), columns=cols)
    f0['signal'] = 90. * (3 if im.dtype == np.uint16 else 1)
    if iso:
        f0['size'] = sizes[0] * float(rng.uniform(0.97, 1.03))
    else:
        for c, s in zip(['size_z', 'size_y', 'size_x'], sizes):
            f0[c] = s
    f0['background'] = 4.
    mode = MODES[int(rng.randint(len(MODES)))]
    if mode:
        kw['param_mode'] = mode
    return f0, im, diameter, kw


def compare(eng, seed):
    """One configuration through engine and oracle -> (description, differs, max dpos) or None if the
    generator made no large cluster."""
    f0, im, diameter, kw = case(seed)
    prep = cta.prepare_batch(f0, im, diameter, **kw)
    b = prep.batch
    n_per = np.diff(b.feat_offset)
    if n_per.max() <= 64:
        return None
    ref = _abi.HostBatch(b.frames, b.frame_index, b.feat_offset, b.params, b.low, b.high)
    eng.refine_batch(prep.problem, b)
    ctr_oracle.run_batch(prep.problem, ref, 8)
    pos = slice(2, 2 + im.ndim)
    same_status = np.array_equal(b.status, ref.status)
    ok = (b.status == 0) & (ref.status == 0)
    rows = np.repeat(ok, n_per)
    d = np.abs(b.params_out[rows][:, pos] - ref.params_out[rows][:, pos]).max() if rows.any() else 0.
    # (a frame with NaN pixels has a NaN maximum: the cost is NaN in the reference, the oracle and the engine alike)
    same_nan = np.array_equal(np.isnan(b.cost), np.isnan(ref.cost))
    both = ok & ~np.isnan(b.cost) & ~np.isnan(ref.cost)
    dc = np.abs(b.cost[both] - ref.cost[both]).max() if both.any() else 0.
    differs = not (same_status and same_nan and d < 1e-6 and dc < 1e-9)
    text = 'seed %d: %dD %s features %s clusters %d dtype %s %s: status %s/%s iters %s/%s max dpos %.2e dcost %.1e' % (
        seed, im.ndim, str(diameter), n_per.max(), b.n_clusters, im.dtype, kw, np.bincount(b.status, minlength=4),
        np.bincount(ref.status, minlength=4), b.n_iter[n_per.argmax()], ref.n_iter[n_per.argmax()], d, dc)
    return text, differs, d


if __name__ == '__main__':
    n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    eng = _lib.default_engine(0)
    bad, worst, n_done = 0, 0., 0
    t_start = time.time()
    for seed in range(first, first + n_seeds):
        r = compare(eng, seed)
        if r is None:
            print('seed %d: no large cluster, skipped' % seed, flush=True)
            continue
        n_done += 1
        bad += 1 if r[1] else 0
        worst = max(worst, r[2])
        print(r[0] + ('   <-- DIFFERS' if r[1] else ''), flush=True)
    print('%d configurations with a large cluster, %d differ, worst dpos %.2e px, %.0f s' % (n_done, bad, worst, time.time() - t_start))
