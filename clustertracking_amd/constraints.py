"""Equality constraints on cluster geometry.

Same constructors and return shape as reference
``clustertracking/constraints.py:59-137`` (a 1-tuple holding a dict with
``type``, ``cluster_size``, ``fun``, ``args``), plus a ``kind`` key the engine
dispatches on: the constraint functions are evaluated *inside* the HIP kernel
(with analytic Jacobians; the reference lets SLSQP finite-difference them,
constraints.py:53-55), so arbitrary Python callables cannot be honoured.
``fun`` is kept as a NumPy callable so that results can be checked by hand.
"""
import numpy as np

from .utils import validate_tuple

_PAIRS = ((0, 1), (1, 2), (0, 2), (1, 3), (0, 3), (2, 3))


def _pair_dist2(x, dist, ndim, pairs):
    pos = np.asarray(x)[..., 2:2 + ndim]
    return np.stack([np.sum(((pos[:, a] - pos[:, b]) / dist) ** 2, axis=1)
                     for a, b in pairs])


def _dimer_fun(x, dist, ndim):
    """``1 - sum(((p0 - p1)/dist)^2)`` per cluster (constraints.py:59-61)."""
    return 1 - _pair_dist2(x, dist, ndim, _PAIRS[:1])[0]


def _trimer_fun(x, dist, ndim):
    """All three pair distances equal ``dist`` (constraints.py:79-83)."""
    return np.concatenate(1 - _pair_dist2(x, dist, ndim, _PAIRS[:3]))


def _tetramer_fun_2d(x, dist):
    """Square: the 4 smallest of the 6 pair distances (constraints.py:102-114)."""
    d2 = np.sort(_pair_dist2(x, dist, 2, _PAIRS), axis=0)[:4]
    return np.ravel(1 - d2)


def _tetramer_fun_3d(x, dist):
    """Tetrahedron: all 6 pair distances (constraints.py:117-123)."""
    return np.concatenate(1 - _pair_dist2(x, dist, 3, _PAIRS))


def dimer(dist, ndim=2):
    """Constrain clusters of 2 at given distance (per-axis tuple allowed)."""
    dist = np.array(validate_tuple(dist, ndim), dtype=np.float64)
    return (dict(type='eq', cluster_size=2, fun=_dimer_fun, args=(dist, ndim),
                 kind='dimer'),)


def trimer(dist, ndim=2):
    """Constrain clusters of 3: all three distances equal ``dist``."""
    dist = np.array(validate_tuple(dist, ndim), dtype=np.float64)
    return (dict(type='eq', cluster_size=3, fun=_trimer_fun, args=(dist, ndim),
                 kind='trimer'),)


def tetramer(dist, ndim=2):
    """Constrain clusters of 4: a square in 2D (4 constraints), a tetrahedron
    in 3D (6 constraints)."""
    dist = np.array(validate_tuple(dist, ndim), dtype=np.float64)
    if ndim == 2:
        fun = _tetramer_fun_2d
    elif ndim == 3:
        fun = _tetramer_fun_3d
    else:
        raise NotImplementedError
    return (dict(type='eq', cluster_size=4, fun=fun, args=(dist,), kind='tetramer'),)


def dimer_global(mpp, ndim=2):
    """Reference constraints.py:140-171 couples every cluster of the call into
    one optimisation problem; that does not shard and is outside the engine."""
    raise NotImplementedError(
        "dimer_global couples all clusters into one problem (reference "
        "refine.py:319-332) and is not supported by the MI355X engine")


def engine_constraint(constraints, ndim):
    """Translate a reference-style constraint list to ``(kind, dist)`` or None."""
    if constraints is None:
        return None
    if isinstance(constraints, dict):
        constraints = [constraints] if constraints else []
    constraints = list(constraints)
    if len(constraints) == 0:
        return None
    if len(constraints) > 1:
        # the reference itself mis-binds several constraints (late-bound loop
        # variable, constraints.py:25-31); refuse instead of guessing
        raise NotImplementedError("only one constraint per call is supported")
    cons = constraints[0]
    kind = cons.get('kind')
    if kind not in ('dimer', 'trimer', 'tetramer'):
        raise NotImplementedError(
            "only constraints built by clustertracking_amd.constraints."
            "dimer/trimer/tetramer are supported (custom Python constraint "
            "functions cannot run inside the device solver)")
    if cons.get('type', 'eq') != 'eq':
        raise NotImplementedError("only equality constraints are supported")
    dist = np.asarray(cons['args'][0], dtype=np.float64)
    if dist.shape != (ndim,):
        raise ValueError("constraint distance must have %d entries" % ndim)
    return kind, dist
