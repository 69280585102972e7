"""Synthetic frames on the device (``ctr_draw_frames_device``) against the host
restatement of the reference's drawing rule (clustertracking_amd/artificial.py = reference
artificial.py:131-141, itself pinned by the golden fixtures it generated).  Noise-free frames are
compared bit for bit, including uint8 wrap-around where features pile up, clipped patches at the
frame edges and anisotropic 3D features; the Poisson noise (the engine's own generator) by its
statistics and the clip.  Needs a real MI355X."""
import numpy as np
import pytest
from numpy.testing import assert_equal

import clustertracking_amd as cta
from clustertracking_amd import artificial, workloads
from clustertracking_amd.device import draw_frames

pytestmark = pytest.mark.gpu


def host_frames(shape, frame_of, pos, size, max_value, n_frames, dtype):
    out = np.zeros((n_frames,) + tuple(shape), dtype)
    size = np.broadcast_to(np.asarray(size, float), pos.shape)
    max_value = np.broadcast_to(np.asarray(max_value, float), (len(pos),))
    for f, p, s, m in zip(frame_of, pos, size, max_value):
        artificial.draw_gaussian(out[f], p, tuple(s), m)
    return out


def to_numpy(t, dtype):
    a = t.cpu().numpy()
    return a.view(np.uint16) if dtype == np.uint16 else a


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16])
def test_noise_free_2d_bit_exact_with_wraparound_and_edges(engine, dtype):
    rng = np.random.RandomState(3)
    shape = (96, 120)
    n = 300                                   # dense: sums beyond 255 wrap around in uint8
    pos = np.column_stack([rng.uniform(0, shape[0] - 1e-9, n), rng.uniform(0, shape[1] - 1e-9, n)])
    pos[:4] = [[0.2, 0.3], [95.7, 119.6], [0.0, 60.5], [48.3, 119.99]]      # corners and edges
    frame_of = rng.randint(0, 3, n).astype(np.int32)
    size = rng.uniform(2., 4.5, (n, 2))
    mv = rng.uniform(40, 250 if dtype == np.uint8 else 4000, n)
    want = host_frames(shape, frame_of, pos, size, mv, 3, dtype)
    got = to_numpy(draw_frames(shape, frame_of, pos, size, mv, n_frames=3, dtype=dtype), dtype)
    assert_equal(got, want)
    if dtype == np.uint8:
        big = host_frames(shape, frame_of, pos, size, mv, 3, np.int64)
        assert (big > 255).any(), "the case is meant to wrap around"


def test_noise_free_3d_anisotropic_bit_exact(engine):
    rng = np.random.RandomState(4)
    shape = (24, 40, 48)
    n = 60
    pos = np.column_stack([rng.uniform(0, s - 1e-9, n) for s in shape])
    frame_of = rng.randint(0, 2, n).astype(np.int32)
    want = host_frames(shape, frame_of, pos, (2., 4., 4.), 100., 2, np.uint8)
    got = to_numpy(draw_frames(shape, frame_of, pos, (2., 4., 4.), 100., n_frames=2), np.uint8)
    assert_equal(got, want)


def test_cfg2_frames_from_seeds_match_host_generation(engine):
    """The noise-free part of the benchmark workload: same truth positions (host RNG, seed =
    frame index), frames drawn on the device == frames drawn on the host."""
    n_frames = 4
    pos, frame_of = [], []
    for t in range(n_frames):
        rs = np.random.RandomState(t)
        truth = np.stack([rs.uniform(13, s - 1 - 13, 200) for s in (512, 512)], axis=1)
        pos.append(truth)
        frame_of.append(np.full(200, t, np.int32))
    pos, frame_of = np.concatenate(pos), np.concatenate(frame_of)
    want = host_frames((512, 512), frame_of, pos, 3., 100., n_frames, np.uint8)
    got = to_numpy(draw_frames((512, 512), frame_of, pos, 3., 100., n_frames=n_frames), np.uint8)
    assert_equal(got, want)


def test_poisson_noise_statistics_and_clip(engine):
    shape = (256, 256)
    empty = np.zeros((0, 2))
    for level in (10., 48., 200.):
        a = to_numpy(draw_frames(shape, np.zeros(0, np.int32), empty, 3., 100., n_frames=2, noise=level,
                                 seed=5), np.uint8).astype(np.float64)
        if level < 200:
            assert abs(a.mean() - level) < 4 * np.sqrt(level / a.size) + 0.05
            assert abs(a.var() - level) < 0.05 * level
        else:
            assert a.max() == 255 and a.mean() > 190        # clipped to the pixel range
    # different seeds give different frames, the same seed the same frame
    f1 = draw_frames(shape, np.zeros(0, np.int32), empty, 3., 100., n_frames=1, noise=10., seed=1)
    f2 = draw_frames(shape, np.zeros(0, np.int32), empty, 3., 100., n_frames=1, noise=10., seed=2)
    f1b = draw_frames(shape, np.zeros(0, np.int32), empty, 3., 100., n_frames=1, noise=10., seed=1)
    assert (f1 != f2).any() and bool((f1 == f1b).all())


def test_device_frames_feed_the_refine_path(engine):
    """frames drawn on the device -> refine on the device -> rms vs truth at the reference's bar
    for S/N 10 (tests/test_refine.py:40)"""
    from clustertracking_amd.device import DeviceBatch
    frames, f0, truth, opts = workloads.cfg2(2, 0)
    dev = draw_frames((512, 512), f0['frame'].values.astype(np.int32), truth, 3., 100., n_frames=2,
                      noise=10., seed=9)
    host = dev.cpu().numpy()
    res = cta.refine_leastsq(f0, cta.ArrayReader(host), opts['diameter'])
    ok = ~np.isnan(res['cost'].values)
    assert ok.mean() > 0.99
    rms = np.sqrt(np.mean((res[['y', 'x']].values - truth)[ok] ** 2))
    assert rms < 0.05, rms
