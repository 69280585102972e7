"""Cluster labelling on the host (reference find.py:12-163), following the
reference's tests/test_find.py:34-125 with fixed seeds."""
import numpy as np
import pandas as pd
from numpy.testing import assert_equal

import _cases  # noqa: F401
import clustertracking_amd as cta


def dummy_cluster(rng, N, center, separation, ndim=2):
    devs = (rng.random_sample((N, ndim)) - 0.5) * separation / np.sqrt(ndim)
    return np.array(center)[np.newaxis, :] + devs


def dummy_clusters(rng, N, max_size, separation, ndim=2):
    center = np.array([separation] * ndim, dtype=float)
    sizes = rng.randint(1, max_size, N)
    displ = (rng.random_sample((N, ndim)) + 2) * separation
    res = []
    for i, size in enumerate(sizes):
        center = center + displ[i]
        res.append(dummy_cluster(rng, size, center, separation, ndim))
    return res


def pos_to_df(pos):
    pos_a = np.concatenate(pos)
    return pd.DataFrame(pos_a, columns=['z', 'y', 'x'][-pos_a.shape[1]:])


def test_single_and_multiple_clusters():
    rng = np.random.RandomState(0)
    for ndim in (2, 3):
        for sep in rng.random_sample(10) * 10:
            pos = dummy_clusters(rng, 1, 10, sep, ndim)
            df = cta.find_clusters(pos_to_df(pos), sep)
            assert_equal(df['cluster_size'].values, len(pos[0]))
        for number in rng.randint(1, 10, 10):
            pos = dummy_clusters(rng, number, 10, 1, ndim)
            df = cta.find_clusters(pos_to_df(pos), 1)
            assert_equal(df['cluster'].nunique(), number)


def test_line_clusters():
    rng = np.random.RandomState(1)
    for ndim in (2, 3):
        for _ in range(10):
            sep = rng.random_sample() * 10 + 0.1
            vec = rng.normal(size=ndim)
            vec /= np.linalg.norm(vec)
            ind = np.arange(10)
            for order in (ind, ind[::-1], rng.permutation(10)):
                pos = order[:, None] * vec[None, :] * sep
                df = cta.find_clusters(pos_to_df([pos]), sep * 1.1)
                assert_equal(df['cluster_size'].values, 10)
                df = cta.find_clusters(pos_to_df([pos]), sep * 0.9)
                assert_equal(df['cluster_size'].values, 1)


def test_anisotropic_separation_and_frames():
    pos = np.array([[0., 0.], [0., 5.], [5., 0.], [40., 40.]])
    df = pd.DataFrame(pos, columns=['y', 'x'])
    out = cta.find_clusters(df, (2., 6.))      # only the x-neighbour is within reach
    assert_equal(out['cluster_size'].values, [2, 2, 1, 1])
    assert 'frame' not in df                     # temporary column removed again (find.py:149-160)
    df['frame'] = [1, 1, 0, 0]
    out = cta.find_clusters(df, 10.)
    assert_equal(out['frame'].values, [0, 0, 1, 1])            # grouped by frame
    assert_equal(out['cluster_size'].values, [1, 1, 2, 2])
    assert out['cluster'].values[2] == out['cluster'].values[3]
    assert len(set(out['cluster'].values)) == 3
