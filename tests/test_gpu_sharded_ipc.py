"""The multi-GPU hand-over as PRODUCT code: clustertracking_amd.parallel.refine_leastsq_sharded
with transport='ipc' -- rank 0's inbox mapped by another process (ctr_ipc_open, owner checked by
PCI bus id), the engine writing its rows there.  Two processes share the one GPU of the box
(control plane over gloo); the table rank 0 assembles must equal the single-process table."""
import os
import socket
import sys

import numpy as np
import pandas as pd
import pytest
from numpy.testing import assert_equal

import _cases
import clustertracking_amd as cta

pytestmark = pytest.mark.gpu


def _video(n_frames=6):
    frames, tabs = [], []
    for t in range(n_frames):
        im, truth, p0 = cta.artificial.random_frame((128, 144), 10 + 6 * t, 3., 100, 10, 700 + t, margin=13)
        frames.append(im)
        tab = pd.DataFrame(p0, columns=['y', 'x'])
        tab['frame'] = t
        tabs.append(tab)
    f0 = pd.concat(tabs, ignore_index=True)
    f0['signal'] = 90.
    f0['size'] = 3.
    f0['background'] = 5.
    f0['note'] = np.arange(len(f0)) % 7          # a column that is not fitted travels with the table
    return np.stack(frames), f0


def _worker(rank, world, port, out_dir, transport):
    import torch.distributed as dist
    from clustertracking_amd import parallel
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    frames, f0 = _video()
    res = parallel.refine_leastsq_sharded(f0, cta.ArrayReader(frames), 13, device=0, transport=transport,
                                          balance='cost')
    res.to_pickle(os.path.join(out_dir, 'rank%d.pkl' % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("transport,no_ipc", [('ipc', False), ('ipc', True), ('collective', False)],
                         ids=['ipc', 'ipc-falls-back', 'collective'])
def test_two_processes_one_gpu(tmp_path, transport, no_ipc, monkeypatch):
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    if no_ipc:
        monkeypatch.setenv('CTR_NO_IPC', '1')
    world = 2
    mp.spawn(_worker, args=(world, port, str(tmp_path), transport), nprocs=world, join=True)
    frames, f0 = _video()
    single = cta.refine_leastsq(f0.copy(), cta.ArrayReader(frames), 13)
    got0 = pd.read_pickle(os.path.join(str(tmp_path), 'rank0.pkl'))
    assert_equal(np.asarray(got0.index), np.asarray(single.index))
    assert list(got0.columns) == list(single.columns)
    for col in single.columns:
        assert_equal(got0[col].values.astype(float), single[col].values.astype(float), err_msg=col)
    got1 = pd.read_pickle(os.path.join(str(tmp_path), 'rank1.pkl'))
    if transport == 'ipc' and not no_ipc:
        # the other rank keeps its own rows: the tail of the table (contiguous frame blocks)
        assert 0 < len(got1) < len(single)
        tail = single.loc[got1.index]
        for col in single.columns:
            assert_equal(got1[col].values.astype(float), tail[col].values.astype(float), err_msg=col)
    else:
        assert_equal(np.asarray(got1.index), np.asarray(single.index))
