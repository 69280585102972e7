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


def find_clusters(f, separation, pos_columns=None, t_column='frame'):
    """Copy of ``f`` (rows grouped by frame) with ``cluster`` and
    ``cluster_size`` columns (reference find.py:132-163)."""
    if pos_columns is None:
        pos_columns = guess_pos_columns(f)
    separation = np.array(validate_tuple(separation, len(pos_columns)),
                          dtype=np.float64)
    remove_t = t_column not in f
    if remove_t:
        f[t_column] = 0
    try:
        parts = [x[1] for x in find_iter(f, separation, pos_columns, t_column)]
        if parts:
            result = pd.concat(parts)
        else:  # empty table (the reference raises ValueError from pd.concat here)
            result = f.copy()
            result['cluster'] = np.zeros(0, dtype=np.int64)
            result['cluster_size'] = np.zeros(0, dtype=np.int64)
    finally:
        if remove_t:
            del f[t_column]
    return result
