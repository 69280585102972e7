"""The benchmark workloads of SURVEY.md section 8(d), generated from seeds.

All configurations are stated in the reference's ``size`` units (radius of
gyration; sigma = size / sqrt(ndim), reference fitfunc.py:112-113).
"""
import numpy as np
import pandas as pd

from . import artificial


def cfg2(n_frames=256, first_seed=0, shape=(512, 512), n_features=200, size=3.,
         diameter=13, signal=100, noise=10):
    """cfg 2: ``n_frames`` frames of 512x512 uint8 with 200 Gaussians each
    (size 3, signal 100, Poisson noise 10, seed = frame index), initial
    guesses = truth + U(-0.5, 0.5) px, signal 90, background noise/2.

    Returns (frames [T,H,W] uint8, f0 DataFrame, truth [T*n, 2], options dict).
    """
    frames = np.empty((n_frames,) + tuple(shape), dtype=np.uint8)
    tabs, truths = [], []
    for t in range(n_frames):
        im, truth, p0 = artificial.random_frame(shape, n_features, size, signal, noise,
                                                seed=first_seed + t, margin=diameter)
        frames[t] = im
        tab = pd.DataFrame(p0, columns=['y', 'x'])
        tab['frame'] = t
        tabs.append(tab)
        truths.append(truth)
    f0 = pd.concat(tabs, ignore_index=True)
    f0['signal'] = 0.9 * signal
    f0['size'] = float(size)
    f0['background'] = noise / 2.
    return frames, f0, np.concatenate(truths), dict(diameter=diameter)


def cfg3(n_stacks=4, first_seed=0, shape=(64, 128, 128), n_features=500,
         size=(2., 4., 4.), diameter=(9, 17, 17), signal=100, noise=10):
    """cfg 3: 3D stacks (z, y, x) = (64, 128, 128), anisotropic Gaussians."""
    frames = np.empty((n_stacks,) + tuple(shape), dtype=np.uint8)
    tabs, truths = [], []
    for t in range(n_stacks):
        im, truth, p0 = artificial.random_frame(shape, n_features, size, signal, noise,
                                                seed=first_seed + t, margin=diameter)
        frames[t] = im
        tab = pd.DataFrame(p0, columns=['z', 'y', 'x'])
        tab['frame'] = t
        tabs.append(tab)
        truths.append(truth)
    f0 = pd.concat(tabs, ignore_index=True)
    f0['signal'] = 0.9 * signal
    for c, s in zip(('size_z', 'size_y', 'size_x'), size):
        f0[c] = float(s)
    f0['background'] = noise / 2.
    return frames, f0, np.concatenate(truths), dict(diameter=diameter)


def cfg5(n_frames=8, first_seed=0, shape=(512, 512), n_clusters=36, size=3.,
         diameter=13, signal=60, noise=10):
    """cfg 5: compact clusters of 2 / 8-16 Gaussians at spacing 2*size; the
    dimers carry the ``constraints.dimer(2*size)`` constraint."""
    frames = np.empty((n_frames,) + tuple(shape), dtype=np.uint8)
    tabs, truths = [], []
    for t in range(n_frames):
        im, truth, p0 = artificial.cluster_frame(shape, n_clusters, [2, 8, 10, 12, 14, 16],
                                                 size, 2.0, signal, noise,
                                                 seed=first_seed + t)
        frames[t] = im
        tab = pd.DataFrame(p0, columns=['y', 'x'])
        tab['frame'] = t
        tabs.append(tab)
        truths.append(truth)
    f0 = pd.concat(tabs, ignore_index=True)
    f0['signal'] = 0.9 * signal
    f0['size'] = float(size)
    f0['background'] = noise / 2.
    return frames, f0, np.concatenate(truths), dict(diameter=diameter)
