"""N > 1 path on CPU: world_size-2 gloo processes shard a video by frames,
refine their blocks (oracle backend via the test hook) and gather the rows.
The gathered table must equal the single-process result row for row."""
import os
import socket
import sys

import numpy as np
import pandas as pd
import pytest
from numpy.testing import assert_allclose, assert_equal

import _cases
import clustertracking_amd as cta
from clustertracking_amd import parallel


def test_frame_blocks_partition():
    frames = np.repeat(np.arange(11), 3)
    for world in (1, 2, 3, 4, 8, 16):
        blocks = [parallel.frame_block(frames, world, r) for r in range(world)]
        assert_equal(np.concatenate(blocks), np.arange(11))
        lens = [len(b) for b in blocks]
        assert max(lens) - min(lens) <= 1


def test_frame_blocks_balanced_by_weight():
    """blocks balanced by feature count when the frames are uneven -- contiguous,
    a partition, and no rank more than one frame's weight away from its share."""
    rng = np.random.RandomState(3)
    per_frame = rng.randint(1, 400, 57)
    per_frame[10] = 3000                      # one frame as heavy as ten others
    frames = np.repeat(np.arange(57) * 2 + 5, per_frame)   # (frame numbers need not start at 0)
    for world in (1, 2, 3, 8):
        blocks = [parallel.frame_block(frames, world, r, np.ones(len(frames))) for r in range(world)]
        assert_equal(np.concatenate(blocks), np.unique(frames))
        share = len(frames) / world
        for b in blocks:
            assert abs(np.isin(frames, b).sum() - share) <= per_frame.max()
        # the same cut on every rank whatever the row order
        perm = rng.permutation(len(frames))
        again = [parallel.frame_block(frames[perm], world, r, np.ones(len(frames))) for r in range(world)]
        for a, b in zip(blocks, again):
            assert_equal(a, b)
    # equal weights per frame = the plain split up to rounding of the cut positions
    even = np.repeat(np.arange(12), 7)
    lens = [len(parallel.frame_block(even, 4, r, np.ones(len(even)))) for r in range(4)]
    assert lens == [3, 3, 3, 3]
    # likely slow fits weigh more: two start positions 0.5 px apart
    f = pd.DataFrame(dict(frame=[0, 0, 0, 1, 1, 1], y=[10., 10.4, 30., 10., 20., 30.], x=[10., 10.3, 30., 10., 20., 30.]))
    w = parallel.shard_weights(f, ['y', 'x'], (6, 6), slow_fit_weight=50.)
    assert_equal(w, [51., 51., 1., 1., 1., 1.])


def _video(n_frames=5):
    frames, tabs = [], []
    for t in range(n_frames):
        im, truth, p0 = cta.artificial.random_frame((96, 112), 14 + t, 3., 100, 10,
                                                    200 + t, margin=13)
        frames.append(im)
        tab = pd.DataFrame(p0, columns=['y', 'x'])
        tab['frame'] = t
        tabs.append(tab)
    f0 = pd.concat(tabs, ignore_index=True)
    f0['signal'] = 90.
    f0['size'] = 3.
    f0['background'] = 5.
    return np.stack(frames), f0


def _worker(rank, world, port, out_dir, extra=None):
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(_cases.ROOT, 'oracle'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    frames, f0 = _video()
    res = _cases.refine_leastsq_sharded(f0, cta.ArrayReader(frames), 13,
                                        _run_batch=_cases.oracle_runner(), **(extra or {}))
    res.to_pickle(os.path.join(out_dir, 'rank%d.pkl' % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("extra", [None, dict(compute_error=True, noise_size=1, threshold=2,
                                                  param_mode=dict(size='var'))],
                         ids=['defaults', 'compute_error+noise_size+sizevar'])
def test_two_rank_gloo_matches_single_process(tmp_path, oracle, extra):
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    world = 2
    mp.spawn(_worker, args=(world, port, str(tmp_path), extra), nprocs=world, join=True)
    frames, f0 = _video()
    single = _cases.refine_leastsq(f0.copy(), cta.ArrayReader(frames), 13,
                                _run_batch=_cases.oracle_runner(), **(extra or {}))
    if extra:
        assert 'size_std' in single and 'x_std' in single
    for rank in range(world):
        got = pd.read_pickle(os.path.join(str(tmp_path), 'rank%d.pkl' % rank))
        assert_equal(np.asarray(got.index), np.asarray(single.index))
        assert list(got.columns) == list(single.columns)
        assert_equal(got['cluster'].values, single['cluster'].values)
        assert_equal(got['cluster_size'].values, single['cluster_size'].values)
        for col in single.columns:
            assert_allclose(got[col].values.astype(float), single[col].values.astype(float),
                            rtol=0, atol=0, err_msg=col)
