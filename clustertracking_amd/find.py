"""Host-side cluster labelling: which features are fitted together.

Behavioural mirror of reference ``clustertracking/find.py:12-163``
(``Clusters``, ``_find``, ``find_iter``, ``find_clusters``): features closer
than ``separation`` (per-axis scaled Euclidean distance < 1) belong to one
cluster; clusters never span frames; ids carry a running per-frame offset.

The labels themselves follow the reference's merge rule (when a pair (a, b) is
joined, b's whole cluster takes a's current label; pairs are visited in the
iteration order of the set returned by ``cKDTree.query_pairs``), so that ids
are equal to the reference's and not merely partition-equivalent.
"""
import numpy as np
import pandas as pd
from scipy.spatial import cKDTree

from .utils import guess_pos_columns, validate_tuple


def label_points(pos, separation):
    """Cluster labels and sizes for one frame.

    pos : [n, ndim] array; separation : per-axis tuple.
    Returns (ids [n] int, sizes [n] int).  (reference find.py:72-93)"""
    pos = np.asarray(pos, dtype=np.float64)
    n = len(pos)
    label = list(range(n))
    if n > 1:
        pairs = cKDTree(pos / separation).query_pairs(1)
        members = {}
        for a, b in pairs:
            la, lb = label[a], label[b]
            if la == lb:
                continue
            grp_a = members.setdefault(la, [la])
            grp_b = members.pop(lb, [lb])
            for k in grp_b:
                label[k] = la
            grp_a.extend(grp_b)
    label = np.asarray(label, dtype=np.int64)
    if n == 0:
        return label, label.copy()
    sizes = np.bincount(label, minlength=n)[label]
    return label, sizes


def find_iter(f, separation, pos_columns=None, t_column='frame'):
    """Per-frame generator of ``(frame_no, DataFrame)`` with ``cluster`` and
    ``cluster_size`` columns added (reference find.py:96-129)."""
    if pos_columns is None:
        pos_columns = guess_pos_columns(f)
    next_id = 0
    for frame_no, f_frame in f.groupby(t_column):
        ids, sizes = label_points(f_frame[pos_columns].values, separation)
        result = f_frame.copy()
        result['cluster'] = ids + next_id
        result['cluster_size'] = sizes
        next_id = result['cluster'].max() + 1
        yield frame_no, result


def label_frames(pos, frames, separation):
    """Cluster ids and sizes for a whole table at once (NumPy only).

    pos [N, ndim], frames [N]; returns (order, ids, sizes) where ``order`` is
    the stable frame-sorted row order of find_clusters' output and ids/sizes
    are aligned with it.  Same labels as running :func:`label_points` frame by
    frame with the reference's running id offset (find.py:120-128)."""
    pos = np.asarray(pos, dtype=np.float64)
    frames = np.asarray(frames)
    order = np.argsort(frames, kind='stable')
    fs = frames[order]
    n = len(order)
    ids = np.empty(n, dtype=np.int64)
    sizes = np.empty(n, dtype=np.int64)
    if n == 0:
        return order, ids, sizes
    starts = np.flatnonzero(np.r_[True, fs[1:] != fs[:-1]])
    stops = np.r_[starts[1:], n]
    scaled = pos[order] / separation
    next_id = 0
    for a, b in zip(starts, stops):
        lab, siz = label_points(scaled[a:b], 1.)
        ids[a:b] = lab + next_id
        sizes[a:b] = siz
        next_id = ids[a:b].max() + 1
    return order, ids, sizes


def label_frames_device(pos, frames, separation, device=0):
    """Same contract as :func:`label_frames`, computed by the HIP engine
    (``ctr_find_clusters``).  The partition equals the reference's; the ids are
    canonical (smallest frame-sorted row index of the cluster) instead of the
    reference's set-order-dependent ones."""
    from . import _lib
    pos = np.asarray(pos, dtype=np.float64)
    frames = np.asarray(frames)
    order = np.argsort(frames, kind='stable')
    fs = frames[order]
    n = len(order)
    if n == 0:
        return order, np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64)
    starts = np.flatnonzero(np.r_[True, fs[1:] != fs[:-1]])
    offsets = np.r_[starts, n].astype(np.int32)
    labels, sizes = _lib.default_engine(device).find_clusters(pos[order], offsets, separation)
    return order, labels.astype(np.int64), sizes.astype(np.int64)


def find_clusters(f, separation, pos_columns=None, t_column='frame', labels='reference',
                  device=0):
    """Copy of ``f`` (rows grouped by frame) with ``cluster`` and
    ``cluster_size`` columns (reference find.py:132-163).  One pass over NumPy
    arrays instead of a DataFrame copy per frame; same rows, order and labels.

    ``labels='device'`` computes the same partition on the MI355X
    (``ctr_find_clusters``) with canonical ids (smallest row of the cluster)."""
    if pos_columns is None:
        pos_columns = guess_pos_columns(f)
    separation = np.array(validate_tuple(separation, len(pos_columns)),
                          dtype=np.float64)
    if t_column in f:
        frames = f[t_column].values
    else:
        frames = np.zeros(len(f), dtype=np.int64)
    if labels == 'reference':
        order, ids, sizes = label_frames(f[pos_columns].values, frames, separation)
    elif labels == 'device':
        order, ids, sizes = label_frames_device(f[pos_columns].values, frames, separation, device)
    else:
        raise ValueError("labels must be 'reference' or 'device'")
    result = f.iloc[order].copy()
    if t_column not in f:
        result[t_column] = 0   # the reference's output carries the temporary column (find.py:149-157)
    result['cluster'] = ids
    result['cluster_size'] = sizes
    return result
