"""Frame-parallel refinement across the GPUs of one node.

Clusters never span frames (reference find.py:122-129) and every
``(frame, cluster)`` group is an independent problem (refine.py:333-343), so the
video is cut into contiguous blocks of frames, one block per rank (one process
per GPU).  No collective touches the data path; the only exchange is the final
hand-over of the result rows, plus one integer per rank to keep the running
cluster-id offset of the reference (find.py:120-128).

Two transports for the result rows (``refine_leastsq_sharded(transport=...)``):

``'collective'``  one all-gather of padded f64 blocks (RCCL over xGMI on GPUs, gloo in the
                  CPU tests); every rank ends with the full table.
``'ipc'``         rank 0 owns an *inbox* in its HBM (``Inbox`` / ``open_inbox`` below), every
                  rank maps it and the ENGINE writes its rows there as it finishes (peer stores
                  over xGMI, ``ctr_batch.result_rows`` / ``done_flag``): no collective on the
                  data path, nothing packed on the host.  Rank 0 ends with the full table, the
                  other ranks with their own rows.  Falls back to the collective when the
                  inbox cannot be mapped (no peer access, one GPU hidden from another rank).
"""
import os
import sys

import numpy as np
import pandas as pd

from . import refine as _refine
from .refine import prepare_batch, write_back


def frame_block(frame_numbers, world_size, rank, weights=None):
    """Contiguous block of the sorted unique frame numbers owned by ``rank``.

    Without ``weights`` the blocks differ in length by at most one frame.  With ``weights``
    (one non-negative number per entry of ``frame_numbers``, e.g. 1 per feature row, summed per
    frame) the cuts sit where the cumulative weight passes ``k/world_size`` of the total, so
    that the ranks get about equal work when frames are uneven ("load balance by
    feature count per block if frames are uneven")."""
    frame_numbers = np.asarray(frame_numbers)
    if weights is None:
        uniq = np.unique(frame_numbers)
        n = len(uniq)
        base, extra = divmod(n, world_size)
        start = rank * base + min(rank, extra)
        stop = start + base + (1 if rank < extra else 0)
        return uniq[start:stop]
    uniq, inv = np.unique(frame_numbers, return_inverse=True)
    w = np.bincount(inv, weights=np.asarray(weights, dtype=np.float64), minlength=len(uniq))
    if np.any(w < 0):
        raise ValueError("weights must not be negative")
    cum = np.cumsum(w)
    total = cum[-1] if len(cum) else 0.
    if not total > 0:
        return frame_block(frame_numbers, world_size, rank)
    # frame j belongs to the rank whose share holds the midpoint of its weight interval
    mid = cum - 0.5 * w
    owner = np.minimum((mid * world_size / total).astype(np.int64), world_size - 1)
    return uniq[owner == rank]


def shard_weights(f, pos_columns, radius, t_column='frame', slow_fit_weight=0.):
    """Work estimate per row of ``f`` for ``frame_block``: 1 per feature, plus
    ``slow_fit_weight`` for every feature with another one closer than a quarter of the mask
    radius in the same frame (the engine's own key for likely slow fits, ``front_load_kernel``:
    two start positions on one real feature take hundreds of solver iterations)."""
    w = np.ones(len(f))
    if slow_fit_weight > 0 and len(f):
        from scipy.spatial import cKDTree
        scaled = f[list(pos_columns)].values / (0.25 * np.asarray(radius, dtype=np.float64))
        for _, idx in f.groupby(t_column).indices.items():
            if len(idx) < 2:
                continue
            pairs = cKDTree(scaled[idx]).query_pairs(1.0, output_type='ndarray')
            if len(pairs):
                w[idx[np.unique(pairs)]] += slow_fit_weight
    return w


class Inbox(object):
    """Rank 0's block for the result rows of every rank and batch in flight, mapped by all ranks:
    rows[world, nfl, pad_rows, width] f64, then seq[world, nfl] int64 (``ctr_ipc_*``,
    include/ctrefine.h).  ``nfl`` = batches a rank keeps in flight (1 for one call)."""

    def __init__(self, engine, base, world, nfl, pad_rows, width, owner):
        self.engine, self.base, self.owner = engine, base, owner
        self.world, self.nfl, self.pad_rows, self.width = world, nfl, pad_rows, width
        self.seq_off = world * nfl * pad_rows * width * 8

    @staticmethod
    def n_bytes(world, nfl, pad_rows, width):
        return world * nfl * pad_rows * width * 8 + world * nfl * 8

    def rows_addr(self, r, slot):
        return self.base + ((r * self.nfl + slot) * self.pad_rows) * self.width * 8

    def seq_addr(self, r, slot):
        return self.base + self.seq_off + (r * self.nfl + slot) * 8

    def read_rows(self, r, slot, n):
        return self.engine.ipc_read(self.rows_addr(r, slot), (n, self.width), np.float64)

    def read_seq(self):
        return self.engine.ipc_read(self.base + self.seq_off, (self.world, self.nfl), np.int64)

    def release(self):
        if self.base:
            (self.engine.ipc_free if self.owner else self.engine.ipc_close)(self.base)
            self.base = 0


def open_inbox(torch, dist, engine, rank, world, nfl, pad_rows, width, coll_dev, group=None):
    """Rank 0 allocates the inbox and hands its handle round (the blob names the owning device:
    ``ctr_ipc_open`` refuses a block on a device this rank's GPU has no peer access to BEFORE
    mapping it); every other rank maps it for its own device and proves with a store FROM A
    KERNEL of that device that it can write there (``ctr_ipc_probe``).  Returns (ok on every
    rank, Inbox); ok False -> the caller gathers with a collective instead."""
    ok, box, payload = True, None, [None]
    pad_rows = max(pad_rows, 1)
    try:
        if os.environ.get('CTR_NO_IPC') or os.environ.get('CTR_BENCH_NO_IPC'):   # (test switch: the fallback)
            raise RuntimeError("CTR_NO_IPC is set")
        if rank == 0:
            base, handle = engine.ipc_alloc(Inbox.n_bytes(world, nfl, pad_rows, width))
            box = Inbox(engine, base, world, nfl, pad_rows, width, owner=True)
            payload = [handle]
    except Exception as e:   # noqa: BLE001 (anything here means: no inbox)
        sys.stderr.write("rank 0: cannot export the inbox (%r)\n" % (e,))
        ok = False
    dist.broadcast_object_list(payload, src=dist.get_global_rank(group, 0) if group is not None else 0,
                               group=group)
    if rank != 0:
        try:
            if payload[0] is None:
                raise RuntimeError("rank 0 has no inbox")
            box = Inbox(engine, engine.ipc_open(payload[0]), world, nfl, pad_rows, width, owner=False)
            for slot in range(nfl):
                engine.ipc_probe(box.seq_addr(rank, slot), -1 - rank)   # a peer store into rank 0's memory
        except Exception as e:   # noqa: BLE001
            sys.stderr.write("rank %d: cannot map or write rank 0's inbox (%r)\n" % (rank, e))
            ok = False
    flag = torch.tensor([1 if ok else 0], dtype=torch.int64, device=coll_dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)   # (also orders the probes before rank 0's look)
    ok = bool(flag.item())
    if ok and rank == 0:
        seq = box.read_seq()
        ok = all((seq[r] == -1 - r).all() for r in range(1, world))
    flag = torch.tensor([1 if ok else 0], dtype=torch.int64, device=coll_dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    ok = bool(flag.item())
    if not ok:
        try:
            if box is not None and rank != 0:
                box.release()
        except Exception:   # noqa: BLE001
            pass
        dist.barrier(group=group)          # importers first, then the owner
        try:
            if box is not None and rank == 0:
                box.release()
        except Exception:   # noqa: BLE001
            pass
        if rank == 0:
            sys.stderr.write("no IPC inbox: the result rows are gathered with a collective instead\n")
        box = None
    return ok, box


def _gather_collective(out, world, rank, dev, dist, torch, group):
    """One all-gather of the result rows, a padded f64 block per rank; every rank gets the table."""
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


def refine_leastsq_sharded(f, reader, diameter, group=None, device=None,
                           t_column='frame', gather=True, transport='collective',
                           balance='features', **kwargs):
    """``refine_leastsq`` with the frames sharded over the ranks of ``group``
    (default: the world group).  ``reader`` must serve every frame this rank
    owns.  With ``gather`` and ``transport='collective'`` every rank returns the full result
    table (frame-sorted, identical on all ranks); with ``transport='ipc'`` rank 0 returns the
    full table and the other ranks their own rows (module docstring); without ``gather`` every
    rank returns its own rows.
    ``balance``: ``'features'`` (default) cuts the frames where the cumulative feature count
    passes k/world of the total, ``'cost'`` also weighs likely slow fits (``shard_weights``),
    ``'frames'`` gives every rank the same number of frames.
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
    if transport not in ('collective', 'ipc'):
        raise ValueError("transport must be 'collective' or 'ipc'")
    if balance not in ('features', 'cost', 'frames'):
        raise ValueError("balance must be 'features', 'cost' or 'frames'")
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if t_column not in f:
        raise ValueError("sharding needs a %r column" % t_column)
    weights = None
    if balance != 'frames':
        pos_columns = kwargs.get('pos_columns') or _refine.guess_pos_columns(f)
        ndim = len(pos_columns)
        radius = tuple(x // 2 for x in _refine.validate_tuple(diameter, ndim))
        weights = shard_weights(f, pos_columns, radius, t_column,
                                slow_fit_weight=50. if balance == 'cost' else 0.)
    mine = frame_block(f[t_column].values, world, rank, weights)
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
    dev = torch.device('cuda', device) if backend == 'nccl' else torch.device('cpu')
    prep = prepare_batch(f_local, reader, diameter, t_column=t_column,
                         solver_maxiter=int(options.get('maxiter', 100)), device=device, **kwargs)

    box = None
    use_ipc = transport == 'ipc' and gather and world > 1 and not kwargs.get('compute_error')
    if use_ipc:
        # the engine writes this rank's rows (params_out | cost) straight into rank 0's inbox
        from . import _lib
        from .device import DeviceBatch
        engine = _lib.default_engine(device)
        n_rows = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
        dist.all_gather(n_rows, torch.tensor([prep.batch.n_features], dtype=torch.int64, device=dev), group=group)
        n_rows = [int(c.item()) for c in n_rows]
        width = prep.batch.params.shape[1] + 1
        ok, box = open_inbox(torch, dist, engine, rank, world, 1, max(n_rows), width, dev, group)
        use_ipc = ok
    if use_ipc:
        if prep.batch.n_clusters:
            db = DeviceBatch(prep.problem, prep.batch, device=device, engine=engine,
                             result_rows=(box.rows_addr(rank, 0), box.pad_rows),
                             done_flag=box.seq_addr(rank, 0))
            db.struct.done_value = 1
            db.run()
            db.download()      # (synchronises: rows and flag have left this device)
        else:
            engine.ipc_probe(box.seq_addr(rank, 0), 1)
    elif prep.batch.n_clusters:
        _refine._run_on_engine(prep.problem, prep.batch, device)
    out = write_back(prep)

    # running cluster-id offset: ids of a frame start where the previous frame's ended
    next_id = int(out['cluster'].max()) + 1 if len(out) else 0
    ids = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(ids, torch.tensor([next_id], dtype=torch.int64, device=dev), group=group)
    offset = int(sum(int(x.item()) for x in ids[:rank]))
    out['cluster'] = out['cluster'] + offset
    if not gather:
        return out
    if not use_ipc:
        return _gather_collective(out, world, rank, dev, dist, torch, group)

    # ---- 'ipc': the fitted values are in rank 0's inbox; the tables they belong to (index, frame,
    # cluster ids, columns that are not fitted) travel once as host objects (control plane)
    fitted = list(prep.ff.params) + ['cost']
    skeleton = (out.drop(columns=fitted), np.asarray(prep.order), list(out.columns))
    gathered = [None] * world if rank == 0 else None
    dist.gather_object(skeleton, gathered, dst=dist.get_global_rank(group, 0) if group is not None else 0,
                       group=group)      # (also: every rank's download() is behind us)
    full = out
    try:
        if rank == 0:
            seq = box.read_seq()
            if not (seq[:, 0] == 1).all():
                raise RuntimeError("inbox: rows of rank(s) %s have not arrived" % np.flatnonzero(seq[:, 0] != 1).tolist())
            parts = []
            for r in range(world):
                skel, order, columns = gathered[r]
                rows = box.read_rows(r, 0, n_rows[r])
                vals = np.empty_like(rows)
                vals[order] = rows                  # batch order -> row order of the rank's table
                part = skel.copy()
                for k, col in enumerate(fitted):
                    part[col] = vals[:, k]
                parts.append(part[columns])
            full = pd.concat(parts)
    finally:
        dist.barrier(group=group)
        if rank != 0:
            box.release()
        dist.barrier(group=group)       # importers unmap, then the owner frees
        if rank == 0:
            box.release()
    return full
