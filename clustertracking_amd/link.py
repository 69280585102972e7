"""Frame-to-frame linking of refined coordinates on the host.

BASELINE cfg 4 links the refined coordinates after the GPU refine.  The
reference has no public "link these coordinates" function; its ``Linker`` base
class (reference ``clustertracking/find_link.py:579-733``, the Crocker-Grier
scheme with sub-network resolution, ``:236-376,507-576``) is what this module
restates, vectorised per frame pair:

* candidates: for every feature of the new frame its (up to 10) nearest
  features of the previous frame within ``search_range`` (per-axis scaled
  distance <= 1 + 1e-7), ``find_link.py:259-275``;
* sub-networks: connected components of the candidate graph
  (``find_link.py:320-343``);
* inside a sub-network the set of links minimising ``sum(d^2)`` plus one unit
  per missing link (``SubnetLinker``, ``find_link.py:507-576``); solved here as
  an assignment problem, which has the same optimum as the reference's
  exhaustive recursion;
* unmatched new features start tracks, numbered in lexicographic order of their
  position (``_sort_key_spl_dpl``, ``find_link.py:379-383,701-712``);
* unmatched old features stay candidates for ``memory`` more frames at their
  last position (``find_link.py:594-610,716-732``).
"""
import numpy as np
from scipy.optimize import linear_sum_assignment
from scipy.sparse import coo_matrix
from scipy.sparse.csgraph import connected_components
from scipy.spatial import cKDTree

from .utils import guess_pos_columns, validate_tuple

MAX_NEIGHBORS = 10      # find_link.py:586
MAX_SUB_NET_SIZE = 30   # find_link.py:582


class SubnetOversizeException(Exception):
    """A sub-network has more than 30 source features (find_link.py:531-533)."""


def _assign(n_src, n_dst, cand_src, cand_dst, cand_d):
    """Optimal links inside one level.  Returns link[dst] = src or -1."""
    link = np.full(n_dst, -1, dtype=np.int64)
    if len(cand_src) == 0:
        return link
    graph = coo_matrix((np.ones(len(cand_src)), (cand_src, cand_dst + n_src)),
                       shape=(n_src + n_dst, n_src + n_dst))
    _, comp = connected_components(graph, directed=False)
    comp_of_cand = comp[cand_src]
    order = np.argsort(comp_of_cand, kind='stable')
    bounds = np.flatnonzero(np.r_[True, np.diff(comp_of_cand[order]) != 0, True])
    for a, b in zip(bounds[:-1], bounds[1:]):
        idx = order[a:b]
        s_ids, s_loc = np.unique(cand_src[idx], return_inverse=True)
        d_ids, d_loc = np.unique(cand_dst[idx], return_inverse=True)
        if len(s_ids) == 1 and len(d_ids) == 1:
            link[d_ids[0]] = s_ids[0]
            continue
        if len(s_ids) > MAX_SUB_NET_SIZE:
            raise SubnetOversizeException("Subnetwork contains %d points" % len(s_ids))
        # minimise sum(d^2) + 1 per null link + 1 per missing link  ==  maximise sum(2 - d^2)
        big = 1e6
        cost = np.full((len(s_ids), len(d_ids) + len(s_ids)), big)
        cost[s_loc, d_loc] = cand_d[idx] ** 2 - 2.
        cost[:, len(d_ids):] = 0.          # "no link" for any source
        rows, cols = linear_sum_assignment(cost)
        for r, c in zip(rows, cols):
            if c < len(d_ids) and cost[r, c] < 0.5 * big:
                link[d_ids[c]] = s_ids[r]
    return link


def link_levels(levels, search_range, memory=0):
    """Link a sequence of coordinate arrays ``[n_t, ndim]`` (one per frame).
    Returns a list of integer id arrays aligned with the input."""
    levels = [np.asarray(c, dtype=np.float64) for c in levels]
    if len(levels) == 0:
        return []
    ndim = levels[0].shape[1] if levels[0].ndim == 2 else len(np.atleast_1d(search_range))
    sr = np.asarray(validate_tuple(search_range, ndim), dtype=np.float64)
    next_id = 0
    ids_out = []
    # sources of the next level: previous level + remembered lost features
    src_pos = np.zeros((0, ndim))
    src_id = np.zeros(0, dtype=np.int64)
    mem_pos = np.zeros((0, ndim))
    mem_id = np.zeros(0, dtype=np.int64)
    mem_age = np.zeros(0, dtype=np.int64)
    for t, pos in enumerate(levels):
        pos = pos.reshape(-1, ndim)
        n = len(pos)
        if t == 0:
            ids = np.arange(n, dtype=np.int64)      # find_link.py:620-623
            next_id = n
        else:
            all_src_pos = np.concatenate([src_pos, mem_pos])
            all_src_id = np.concatenate([src_id, mem_id])
            n_src = len(all_src_pos)
            ids = np.full(n, -1, dtype=np.int64)
            link = np.full(n, -1, dtype=np.int64)
            if n_src and n:
                tree = cKDTree(all_src_pos / sr, 15)
                k = min(MAX_NEIGHBORS, n_src)
                dists, inds = tree.query(pos / sr, k, distance_upper_bound=1 + 1e-7)
                dists = dists.reshape(n, -1)
                inds = inds.reshape(n, -1)
                ok = np.isfinite(dists)
                cand_dst = np.nonzero(ok)[0]
                link = _assign(n_src, n, inds[ok], cand_dst, dists[ok])
            linked = link >= 0
            ids[linked] = all_src_id[link[linked]]
            # new tracks in lexicographic order of position (find_link.py:379-383,703)
            new = np.flatnonzero(~linked)
            if len(new):
                order = np.lexsort(pos[new].T[::-1])
                ids[new[order]] = next_id + np.arange(len(new))
                next_id += len(new)
            # memory bookkeeping (find_link.py:716-732)
            if memory > 0:
                used = np.zeros(n_src, dtype=bool)
                used[link[linked]] = True
                lost_new = ~used[:len(src_pos)]
                keep_mem = ~used[len(src_pos):] & (mem_age + 1 < memory)
                mem_pos = np.concatenate([mem_pos[keep_mem], src_pos[lost_new]])
                mem_id = np.concatenate([mem_id[keep_mem], src_id[lost_new]])
                mem_age = np.concatenate([mem_age[keep_mem] + 1,
                                          np.zeros(int(lost_new.sum()), dtype=np.int64)])
        ids_out.append(ids)
        src_pos, src_id = pos, ids
    return ids_out


def link(f, search_range, memory=0, pos_columns=None, t_column='frame'):
    """Return a copy of ``f`` (sorted by frame) with a ``particle`` column."""
    if pos_columns is None:
        pos_columns = guess_pos_columns(f)
    result = f.sort_values(t_column, kind='stable').copy()
    frames = result[t_column].values
    pos = result[pos_columns].values
    uniq, starts = np.unique(frames, return_index=True)
    stops = np.r_[starts[1:], len(frames)]
    ids = link_levels([pos[a:b] for a, b in zip(starts, stops)], search_range, memory)
    result['particle'] = np.concatenate(ids) if len(ids) else np.zeros(0, dtype=np.int64)
    return result
