"""BASELINE cfg 4, the part one GPU does: a 1250-frame shard of a 512x512 video (200 features per
frame) -> frames drawn on the device (ctr_draw_frames_device) -> refine through the drop-in call
-> linking of the refined coordinates on the host (clustertracking_amd.link).
Full size, so checked by size-independent properties: no failed cluster, rms error vs truth at
the reference's bar for S/N 10 (tests/test_refine.py:40), and the tracks: the features are 200
particles on slow random walks, so consecutive positions of one particle must carry one id.
(The 8-GPU run of the 10 000-frame video is the driver's; the sharding itself is covered by
tests/test_parallel_gloo.py.)  Needs a real MI355X."""
import numpy as np
import pandas as pd
import pytest

import clustertracking_amd as cta
from clustertracking_amd.device import draw_frames

pytestmark = pytest.mark.gpu


def test_cfg4_shard_refine_then_link(engine):
    n_frames, n_part, shape = 1250, 200, (512, 512)
    rng = np.random.RandomState(44)
    start = np.stack([rng.uniform(40, s - 41, n_part) for s in shape], axis=1)
    steps = rng.normal(0., 0.5, (n_frames, n_part, 2))
    steps[0] = 0.
    truth = np.clip(start[None] + np.cumsum(steps, axis=0), 14., 497.)       # [T, n, 2]
    frame_of = np.repeat(np.arange(n_frames, dtype=np.int32), n_part)
    pos = truth.reshape(-1, 2)
    dev = draw_frames(shape, frame_of, pos, 3., 100., n_frames=n_frames, noise=10., seed=4)
    frames = dev.cpu().numpy()
    assert frames.shape == (n_frames,) + shape and frames.dtype == np.uint8
    f0 = pd.DataFrame(pos + rng.uniform(-0.5, 0.5, pos.shape), columns=['y', 'x'])
    f0['frame'] = frame_of
    f0['signal'], f0['size'], f0['background'] = 90., 3., 5.
    f0['truth_id'] = np.tile(np.arange(n_part), n_frames)
    res = cta.refine_leastsq(f0, cta.ArrayReader(frames), 13, cluster_labels='device')
    assert len(res) == n_frames * n_part
    ok = np.isfinite(res['cost'].values)
    assert ok.mean() > 0.9995, (~ok).sum()
    tr = truth.reshape(-1, 2)[res.index.values]
    rms = np.sqrt(np.mean((res[['y', 'x']].values - tr)[ok] ** 2))
    assert rms < 0.05, rms
    # linking on the host (find_link.py:579-733 restated): steps of 0.5 px, search range 3 px
    tracks = cta.link_df(res[ok][['y', 'x', 'frame', 'truth_id']], search_range=3.)
    t = tracks.sort_values(['truth_id', 'frame'])
    same_truth = t['truth_id'].values[1:] == t['truth_id'].values[:-1]
    consecutive = same_truth & (t['frame'].values[1:] == t['frame'].values[:-1] + 1)
    linked = t['particle'].values[1:] == t['particle'].values[:-1]
    # two particles that cross within the search range may trade ids: allow 0.5 %
    assert (linked & consecutive).sum() >= 0.995 * consecutive.sum()
    assert tracks['particle'].nunique() < 3 * n_part
