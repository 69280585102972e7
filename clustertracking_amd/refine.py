"""``refine_leastsq`` -- drop-in for reference ``clustertracking/refine.py:82-452``.

The host side keeps what the reference does before and after its per-cluster
loop (option / reader normalisation refine.py:242-289, cluster labelling :297,
parameter columns :299-305, bounds templates :315, and the DataFrame output
:419-430), vectorised.  The loop itself (refine.py:343-430: windows, masks,
objective, minimiser, re-window rounds, failure rules) is ONE batched call into
the HIP engine through the C-ABI of ``include/ctrefine.h``.
"""
import logging

import numpy as np

from . import _abi
from .constraints import engine_constraint
from .find import find_clusters
from .fitfunc import FitFunctions
from .utils import (ArrayReader, guess_pos_columns, is_isotropic,
                    validate_tuple)

logger = logging.getLogger(__name__)


class PreparedBatch(object):
    """Everything one engine call needs, plus how to scatter results back."""

    def __init__(self, f, ff, problem, batch, order, cluster_of_row):
        self.f = f                    # clustered copy of the input (output frame)
        self.ff = ff
        self.problem = problem
        self.batch = batch            # _abi.HostBatch
        self.order = order            # row positions of f in batch feature order
        self.cluster_of_row = cluster_of_row  # batch cluster index per batch feature


def _normalise_reader(f, reader, t_column):
    """refine.py:251-281.  Returns (get_frame, ndim, is_sequence)."""
    try:
        ndim = len(reader.frame_shape)
        return reader, ndim, True
    except AttributeError:
        pass
    try:
        ndim = reader.ndim
    except AttributeError:
        raise ValueError('For multiple frames, the reader should be a'
                         'FramesSequence object exposing the "frame_shape"'
                         'attribute')
    frame_no = getattr(reader, 'frame_no', None)
    if frame_no is not None:
        frame_no = int(frame_no)
    if frame_no is not None and t_column in f:
        assert np.all(f[t_column] == frame_no)
    elif frame_no is not None:
        f[t_column] = frame_no
    elif t_column in f:
        assert f[t_column].nunique() == 1
        frame_no = int(f[t_column].iloc[0])
    else:
        f[t_column] = 0
        frame_no = 0
    return {frame_no: reader}, ndim, False


def prepare_batch(f, reader, diameter, separation=None, fit_function='gauss',
                  param_mode=None, param_val=None, constraints=None, bounds=None,
                  pos_columns=None, t_column='frame', max_iter=10, max_shift=1,
                  max_rms_dev=1., residual_factor=100000., solver_maxiter=100,
                  xtol=0., ftol=0., cluster_labels='reference', device=0,
                  compute_error=False, noise_size=None, threshold=None):
    """Host-side set-up of one refine call (reference refine.py:242-341)."""
    if pos_columns is None:
        pos_columns = guess_pos_columns(f)
    frames_src, ndim, _ = _normalise_reader(f, reader, t_column)
    assert ndim == len(pos_columns)
    if int(max_iter) < 1:
        raise ValueError("max_iter must be at least 1")

    diameter = validate_tuple(diameter, ndim)
    radius = tuple([x // 2 for x in diameter])
    isotropic = is_isotropic(diameter)
    if separation is None:
        separation = diameter

    ff = FitFunctions(fit_function, ndim, isotropic, param_mode)
    modes = np.array(ff.modes)
    if np.any(modes == 2):
        raise NotImplementedError(
            "param_mode 'global' couples all features into one problem "
            "(reference refine.py:319-332) and is not supported by the MI355X "
            "engine")
    if not np.all(modes <= 3):
        raise NotImplementedError("param modes 'particle'/'frame' are not "
                                  "implemented (reference refine.py:339-340)")
    cons = engine_constraint(constraints, ndim)
    f = find_clusters(f, separation, pos_columns, t_column, labels=cluster_labels,
                      device=device)  # makes a copy
    if param_val is not None:
        for col in param_val:
            f[col] = param_val[col]
    for col in ff.params:
        if col not in f.columns:
            f[col] = ff.default[col]

    templates = ff.validate_bounds(bounds, radius=radius)
    params = f[ff.params].values.astype(np.float64)
    low, high = ff.feature_bounds(templates, params)

    # groupby([t_column, 'cluster']) order, rows keep their order inside a group
    frame_vals = f[t_column].values
    cluster_vals = f['cluster'].values
    if len(frame_vals) and np.issubdtype(frame_vals.dtype, np.integer) and \
            np.issubdtype(cluster_vals.dtype, np.integer) and frame_vals.min() >= 0 and cluster_vals.min() >= 0 \
            and (int(frame_vals.max()) + 1) * (int(cluster_vals.max()) + 1) < 2 ** 62:
        # one stable sort of a combined key (the same order as lexsort, at half its cost)
        key = frame_vals.astype(np.int64) * (int(cluster_vals.max()) + 1) + cluster_vals
        order = np.argsort(key, kind='stable')
    else:
        order = np.lexsort((cluster_vals, frame_vals))
    fr_s, cl_s = frame_vals[order], cluster_vals[order]
    n_rows = len(order)
    new = np.ones(n_rows, dtype=bool)
    new[1:] = (fr_s[1:] != fr_s[:-1]) | (cl_s[1:] != cl_s[:-1])
    starts = np.flatnonzero(new)
    feat_offset = np.append(starts, n_rows).astype(np.int32)
    cluster_of_row = np.cumsum(new) - 1
    cl_frames = fr_s[starts]

    # frames block
    if isinstance(frames_src, ArrayReader):
        frames = frames_src.array
        frame_index = cl_frames.astype(np.int64)
        if len(frame_index) and (frame_index.min() < 0 or frame_index.max() >= len(frames)):
            raise IndexError("frame number outside of the video")
    else:
        uniq, inv = np.unique(cl_frames, return_inverse=True)
        if len(uniq):
            frames = np.stack([np.asarray(frames_src[int(i)]) for i in uniq])
        else:
            frames = np.zeros((0,) + (1,) * ndim, dtype=np.uint8)
        frame_index = inv
    if frames.ndim != ndim + 1:
        raise ValueError("frames must have %d dimensions" % ndim)

    problem = _abi.make_problem(ndim, isotropic, ff.modes, radius, cons,
                                max_iter=max_iter, max_shift=max_shift,
                                max_rms_dev=max_rms_dev,
                                residual_factor=residual_factor,
                                solver_maxiter=solver_maxiter, xtol=xtol, ftol=ftol,
                                noise_size=None if noise_size is None else validate_tuple(noise_size, ndim),
                                threshold=threshold, fit_function=ff.fit_function)
    batch = _abi.HostBatch(frames, frame_index, feat_offset, params[order],
                           low[order], high[order], want_std=bool(compute_error))
    # SciPy raises ValueError for an infeasible box (lower > upper)
    n_per = np.diff(feat_offset)
    if n_rows:
        for k, m in enumerate(ff.modes):
            if m == 0:
                continue
            lo_k, hi_k = batch.low[:, k], batch.high[:, k]
            if m != 1:  # shared: loosest bound over the cluster (fitfunc.py:554-557)
                lo_k = np.minimum.reduceat(lo_k, starts)
                hi_k = np.maximum.reduceat(hi_k, starts)
            if np.any(lo_k > hi_k):
                raise ValueError("SLSQP Error: the lower bound exceeds the "
                                 "upper bound (parameter %r)" % ff.params[k])
    return PreparedBatch(f, ff, problem, batch, order, cluster_of_row)


def write_back(prep):
    """Vectorised equivalent of refine.py:408-430."""
    f, ff, batch = prep.f, prep.ff, prep.batch
    n_rows = len(prep.order)
    out = np.empty((n_rows, len(ff.params)), dtype=np.float64)
    out[prep.order] = batch.params_out
    cost = np.empty(n_rows, dtype=np.float64)
    cost[prep.order] = batch.cost[prep.cluster_of_row]
    for k, col in enumerate(ff.params):
        f[col] = out[:, k]
    f['cost'] = cost
    if batch.params_std is not None:      # refine.py:307-312,423-429: '<param>_std' of the fitted ones
        std = np.empty((n_rows, len(ff.params)), dtype=np.float64)
        std[prep.order] = batch.params_std
        for k, col in enumerate(ff.params):
            if ff.modes[k] > 0:
                f[col + '_std'] = std[:, k]
    failed = np.flatnonzero(batch.status != _abi.STATUS_OK)
    for c in failed:
        logger.warning('RefineException: ' + _abi.STATUS_TEXT.get(
            int(batch.status[c]), 'status %d' % batch.status[c]))
    return f


def _run_on_engine(problem, batch, device=0):
    from . import _lib
    return _lib.default_engine(device).refine_batch(problem, batch)


def refine_leastsq(f, reader, diameter, separation=None, fit_function='gauss',
                   param_mode=None, param_val=None, constraints=None,
                   bounds=None, pos_columns=None, t_column='frame',
                   noise_size=None, threshold=None, max_iter=10, max_shift=1,
                   max_rms_dev=1., residual_factor=100000.,
                   compute_error=False, **kwargs):
    """Refines cluster coordinates by least-squares fitting to radial model
    functions, on the MI355X engine.

    Signature, defaults, side effects and output columns follow reference
    ``refine_leastsq`` (refine.py:82-241); see there for the parameters.  This
    does not raise an error if minimization fails: coordinates are unchanged
    and the added column ``cost`` is NaN for that cluster.

    Differences, all deliberate:

    * The minimiser is the engine's bounded Levenberg-Marquardt, run to
      convergence, not SciPy SLSQP at ``tol=1e-6``: ``method`` and ``tol`` in
      ``kwargs`` are accepted and ignored; ``options['maxiter']`` (default 100)
      caps the solver iterations per re-window round.  Engine-specific keys:
      ``xtol``, ``ftol``, ``device``, ``cluster_labels`` ('reference': ids equal to
      the reference's, labelled on the host; 'device': same partition labelled on
      the GPU, canonical ids).
    * ``fit_function``: ``'gauss'``, ``'ring'``, ``'disc'`` and ``'inv_series_<N>'`` (fitfunc.py:112-154,
      with the profile parameters ``thickness`` / ``disc_size`` / ``signal_mult, param_a, ...`` as
      columns like any other: constant by default, ``param_val`` sets them); custom (dict)
      functions and the ``param_mode`` value ``'global'`` raise ``NotImplementedError`` (there is no
      CPU fallback to hand them to).  Fits of the other profiles iterate with the Gauss-Newton
      model, have no ``compute_error`` (NaN) and are limited to clusters of 64 features / 127
      variables and to 12 parameter columns.
    * ``noise_size`` (the lowpass of every window, refine.py:37-40) up to sigma 4.
    * ``compute_error``: the ``'<param>_std'`` columns come from the exact second
      derivatives of the objective (the reference differentiates numerically with
      numdifftools), for every ``param_mode``; clusters of more than 64 features or 127
      variables (the large-cluster kernel) get NaN there.
    * A cluster whose coordinates are all outside the frame, or that has
      non-finite parameters, gets ``cost = NaN`` (the reference means to do
      that, but crashes with IndexError at refine.py:417).
    """
    options = dict(maxiter=100)
    options.update(kwargs.pop('options', None) or {})
    kwargs.pop('method', None)
    kwargs.pop('tol', None)
    xtol = kwargs.pop('xtol', 0.)
    ftol = kwargs.pop('ftol', 0.)
    device = kwargs.pop('device', 0)
    cluster_labels = kwargs.pop('cluster_labels', 'reference')
    if kwargs:
        raise TypeError("unexpected keyword arguments: %s" % sorted(kwargs))
    prep = prepare_batch(f, reader, diameter, separation, fit_function,
                         param_mode, param_val, constraints, bounds,
                         pos_columns, t_column, max_iter, max_shift,
                         max_rms_dev, residual_factor,
                         solver_maxiter=int(options.get('maxiter', 100)),
                         xtol=xtol, ftol=ftol, cluster_labels=cluster_labels, device=device,
                         compute_error=compute_error, noise_size=noise_size, threshold=threshold)
    if prep.batch.n_clusters:
        _run_on_engine(prep.problem, prep.batch, device)
    return write_back(prep)
