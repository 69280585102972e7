"""Frame-parallel refinement across the GPUs of one node.

Clusters never span frames (reference find.py:122-129) and every
``(frame, cluster)`` group is an independent problem (refine.py:333-343), so the
video is cut into contiguous blocks of frames, one block per rank (one process
per GPU).  No collective touches the data path; the only exchange is the final
gather of the result rows (RCCL over xGMI on GPUs, gloo in the CPU tests), plus
one integer per rank to keep the running cluster-id offset of the reference
(find.py:120-128).
"""
import numpy as np
import pandas as pd

import os

from . import refine as _refine
from .refine import prepare_batch, write_back


def frame_block(frame_numbers, world_size, rank):
    """Contiguous block of the sorted unique frame numbers owned by ``rank``;
    blocks differ in length by at most one frame."""
    uniq = np.unique(np.asarray(frame_numbers))
    n = len(uniq)
    base, extra = divmod(n, world_size)
    start = rank * base + min(rank, extra)
    stop = start + base + (1 if rank < extra else 0)
    return uniq[start:stop]


def refine_leastsq_sharded(f, reader, diameter, group=None, device=None,
                           t_column='frame', gather=True, **kwargs):
    """``refine_leastsq`` with the frames sharded over the ranks of ``group``
    (default: the world group).  ``reader`` must serve every frame this rank
    owns.  Every rank returns the full result table when ``gather`` is true
    (frame-sorted, identical on all ranks), else only its own rows.
    Memory: a rank stacks the frames of ITS block that hold features into one host array and
    uploads it as one block (``prepare_batch``; the reference reads frame by frame) -- shard
    finer, or call per chunk of frames, when a rank's block does not fit host memory or HBM.
    ``device``: HIP device index of this process; default ``LOCAL_RANK`` (set by
    torch.distributed.run; the rank inside the node, not the group rank), else the current
    torch device.
    """
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        raise RuntimeError("torch.distributed is not initialised")
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if t_column not in f:
        raise ValueError("sharding needs a %r column" % t_column)
    mine = frame_block(f[t_column].values, world, rank)
    f_local = f[f[t_column].isin(mine)].copy()

    options = dict(maxiter=100)
    options.update(kwargs.pop('options', None) or {})
    kwargs.pop('method', None)
    kwargs.pop('tol', None)
    if kwargs.get('noise_size') is None:
        kwargs.pop('threshold', None)     # only used together with noise_size (refine.py:37-40)
    backend = dist.get_backend(group)
    if device is None:
        if 'LOCAL_RANK' in os.environ:
            device = int(os.environ['LOCAL_RANK'])
        elif backend == 'nccl':
            device = torch.cuda.current_device()
        else:
            device = 0
    if backend == 'nccl' and not 0 <= device < torch.cuda.device_count():
        raise ValueError("device %d is not one of the %d visible GPUs" % (device, torch.cuda.device_count()))
    prep = prepare_batch(f_local, reader, diameter, t_column=t_column,
                         solver_maxiter=int(options.get('maxiter', 100)), device=device, **kwargs)
    if prep.batch.n_clusters:
        _refine._run_on_engine(prep.problem, prep.batch, device)
    out = write_back(prep)

    dev = torch.device('cuda', device) if backend == 'nccl' else torch.device('cpu')

    # running cluster-id offset: ids of a frame start where the previous frame's ended
    next_id = int(out['cluster'].max()) + 1 if len(out) else 0
    ids = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(ids, torch.tensor([next_id], dtype=torch.int64, device=dev), group=group)
    offset = int(sum(int(x.item()) for x in ids[:rank]))
    out['cluster'] = out['cluster'] + offset
    if not gather:
        return out

    # final gather of the result rows: one padded f64 block per rank
    columns = list(out.columns)
    block = np.column_stack([np.asarray(out.index, dtype=np.float64),
                             out.values.astype(np.float64)]) if len(out) else \
        np.zeros((0, len(columns) + 1))
    counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([len(out)], dtype=torch.int64, device=dev), group=group)
    counts = [int(c.item()) for c in counts]
    pad = max(max(counts), 1)
    send = torch.zeros((pad, len(columns) + 1), dtype=torch.float64, device=dev)
    if len(out):
        send[:len(out)] = torch.from_numpy(block).to(dev)
    recv = [torch.empty_like(send) for _ in range(world)]
    dist.all_gather(recv, send, group=group)
    parts = []
    for r in range(world):
        arr = recv[r][:counts[r]].cpu().numpy()
        part = pd.DataFrame(arr[:, 1:], columns=columns, index=arr[:, 0].astype(np.int64))
        parts.append(part)
    full = pd.concat(parts)
    for col in columns:
        if out[col].dtype.kind in 'iu':
            full[col] = full[col].astype(out[col].dtype)
    return full
