"""Faithful CPU restatement of the reference's per-cluster loop INCLUDING its
minimiser: NumPy for the objective, ``scipy.optimize.minimize(method='SLSQP')``
for the solve, exactly as reference ``clustertracking/refine.py:343-430`` does.

TEST INFRASTRUCTURE: imported only by tests/ and bench.py's parity leg.

SciPy is a third-party dependency of the reference (``setup.py:22``, unpinned;
this image has 1.15.3) and is importable on the GPU box, so this oracle can run
there where the reference itself cannot.  It is pinned to the reference by
``tests/test_ref_numpy_oracle.py``: on the golden fixtures it reproduces the
reference's default-tolerance output (oracle A) to ~1e-8 px.

Same batch contract as ``ctr_refine_batch`` (include/ctrefine.h): fills
``batch.params_out / cost / status / n_rounds / n_iter``.

Functions and the reference lines they follow:
  window()         masks.py:42-68
  subimage()       refine.py:28-58
  Layout           fitfunc.py:207-315 (groups=None)
  objective()      fitfunc.py:436-487 (+ r2/dr2 kernels :14-109, gauss :112-118)
  constraints()    constraints.py:17-137
  refine_cluster() refine.py:343-430
"""
import warnings

import numpy as np
from scipy.optimize import minimize

OK, OUT_OF_BOUNDS, NONFINITE, NO_CONVERGENCE, RMS_DEV = 0, 1, 2, 3, 4
PAIRS = ((0, 1), (1, 2), (0, 2), (1, 3), (0, 3), (2, 3))


def window(coords, shape, radius):
    """masks.py:42-68 -> (origin, stop) or None"""
    c = np.round(coords).astype(int)
    shape = np.asarray(shape)
    radius = np.asarray(radius)
    keep = np.all((c >= -radius) & (c < shape + radius), axis=1)
    c = c[keep]
    if len(c) == 0:
        return None
    lo = np.maximum(c.min(0) - radius, 0)
    hi = np.minimum(c.max(0) + radius + 1, shape)
    return lo, hi


def subimage(coords, image, radius):
    """refine.py:28-58 -> (pixels [P] f64, mesh [d, P] f64, masks [n, P] bool) or None"""
    win = window(coords, image.shape, radius)
    if win is None:
        return None
    lo, hi = win
    im = image[tuple(slice(a, b) for a, b in zip(lo, hi))]
    grid = np.indices(im.shape).T
    dist = [np.sum(((grid - (c - lo)) / radius) ** 2, -1) <= 1 for c in coords]
    total = np.any(dist, axis=0).T
    masks = np.array([d.T[total] for d in dist], dtype=bool).reshape(len(coords), -1)
    mesh = np.indices(im.shape, dtype=np.float64)[:, total]
    mesh += np.asarray(lo, dtype=np.float64)[:, None]
    return im[total].astype(np.float64), mesh, masks


class Layout(object):
    """parameter-major optimiser vector with groups=None (fitfunc.py:207-315)"""

    def __init__(self, modes, n):
        self.n = n
        self.base, self.per_feat = [], []
        nv = 0
        for m in modes:
            if m == 0:
                self.base.append(-1)
                self.per_feat.append(False)
            elif m == 1:
                self.base.append(nv)
                self.per_feat.append(True)
                nv += n
            else:
                self.base.append(nv)
                self.per_feat.append(False)
                nv += 1
        self.nv = nv

    def pack(self, params, op):
        out = np.empty(self.nv)
        for k, b in enumerate(self.base):
            if b < 0:
                continue
            if self.per_feat[k]:
                out[b:b + self.n] = params[:, k]
            else:
                out[b] = op(params[:, k])
        return out

    def unpack(self, vect, const):
        out = const.copy()
        for k, b in enumerate(self.base):
            if b < 0:
                continue
            out[:, k] = vect[b:b + self.n] if self.per_feat[k] else vect[b]
        return out


def make_objective(lay, const, pix, mesh, masks, ndim, isotropic, norm):
    """(residual, gradient) closures of fitfunc.py:436-487 for one cluster."""
    n = lay.n
    P = len(pix)
    nsz = 1 if isotropic else ndim

    def model(params, want_derivs):
        diff = pix - params[0, 0]
        derivs = np.zeros((n, 1 + ndim + nsz, P)) if want_derivs else None
        for i in range(n):
            mk = masks[i]
            d = mesh[:, mk] - params[i, 2:2 + ndim, None]          # x - c per axis
            sizes = params[i, 2 + ndim:2 + ndim + nsz]
            sz = np.broadcast_to(sizes, (ndim,)) if isotropic else sizes
            r2 = np.sum(d ** 2 / sz[:, None] ** 2, axis=0)
            g = np.exp(-0.5 * ndim * r2)
            s = params[i, 1]
            diff[mk] -= s * g
            if want_derivs:
                dg = -0.5 * ndim * g
                derivs[i, 0, mk] = g
                derivs[i, 1:1 + ndim, mk] = (s * dg * (-d) * (2. / sz[:, None] ** 2)).T
                if isotropic:
                    derivs[i, 1 + ndim, mk] = s * dg * np.sum(d ** 2, 0) * (-2. / sizes[0] ** 3)
                else:
                    derivs[i, 1 + ndim:, mk] = (s * dg * d ** 2 * (-2. / sz[:, None] ** 3)).T
        return diff, derivs

    def residual(vect):
        if np.any(np.isnan(vect)):
            raise FloatingPointError
        diff, _ = model(lay.unpack(vect, const), False)
        return np.nansum(diff ** 2) / P / norm

    def gradient(vect):
        if np.any(np.isnan(vect)):
            raise FloatingPointError
        params = lay.unpack(vect, const)
        diff, derivs = model(params, True)
        grad = np.empty_like(params)
        grad[:, 1:] = np.nansum(-2 * diff * derivs, axis=2) / P
        grad[:, 0] = np.nansum(-2 * diff) / (n * P)
        return lay.pack(grad, np.sum) / norm

    return residual, gradient


def make_constraints(kind, dist, ndim, lay, const):
    """constraints.py:17-137 for a cluster of matching size; SLSQP finite-differences
    the constraint Jacobian (constraints.py:53-55)."""
    dist = np.asarray(dist, dtype=np.float64)[:ndim]
    need = {1: 2, 2: 3, 3: 4}.get(kind)
    if need is None or lay.n != need:
        return []
    pairs = PAIRS[:1] if kind == 1 else PAIRS[:3] if kind == 2 else PAIRS

    def fun(vect):
        pos = lay.unpack(vect, const)[:, 2:2 + ndim]
        d2 = np.array([np.sum(((pos[a] - pos[b]) / dist) ** 2) for a, b in pairs])
        if kind == 3 and ndim == 2:
            d2 = np.sort(d2)[:4]
        return 1 - d2
    return [dict(type='eq', fun=fun)]


def refine_cluster(frame, fmax, params, low, high, problem, tol, maxiter):
    """-> (params_out, cost, status, n_rounds, n_iter)  (refine.py:343-430)"""
    ndim, isotropic = problem.ndim, bool(problem.isotropic)
    n_params = problem.n_params
    modes = [problem.modes[k] for k in range(n_params)]
    radius = np.array([problem.radius[a] for a in range(ndim)])
    n = len(params)
    if not np.isfinite(params).all():
        return params, np.nan, NONFINITE, 0, 0
    lay = Layout(modes, n)
    norm = float(fmax) ** 2 / problem.residual_factor
    vect = lay.pack(params, np.mean)
    bounds = np.array([lay.pack(low, np.min), lay.pack(high, np.max)]).T
    cons = make_constraints(problem.constraint_kind, problem.constraint_dist, ndim, lay, params)
    coords = params[:, 2:2 + ndim]
    cur = params
    rms = np.nan
    n_iter = 0
    rounds = 0
    for rounds in range(1, problem.max_iter + 1):
        sub = subimage(coords, frame, radius)
        if sub is None or len(sub[0]) == 0:
            return params, np.nan, OUT_OF_BOUNDS, rounds, n_iter
        residual, gradient = make_objective(lay, cur, sub[0], sub[1], sub[2], ndim, isotropic, norm)
        try:
            with warnings.catch_warnings():
                warnings.simplefilter('ignore')
                res = minimize(residual, vect, bounds=bounds, constraints=cons, jac=gradient,
                               method='SLSQP', tol=tol, options=dict(maxiter=maxiter, disp=False))
        except FloatingPointError:
            return params, np.nan, NO_CONVERGENCE, rounds, n_iter
        n_iter += int(res.get('nit', 0))
        if not res['success']:
            return params, np.nan, NO_CONVERGENCE, rounds, n_iter
        rms = np.sqrt(res['fun'] / problem.residual_factor)
        cur = lay.unpack(res['x'], cur)
        new_coords = cur[:, 2:2 + ndim]
        if np.all(np.sum((new_coords - coords) ** 2, 1) < problem.max_shift ** 2):
            break
        coords = new_coords
    if rms > problem.max_rms_dev:
        return params, np.nan, RMS_DEV, rounds, n_iter
    return cur, rms, OK, rounds, n_iter


def run_batch(problem, batch, tol=1e-6, maxiter=100, clusters=None):
    """Fill the outputs of ``batch`` (an ``_abi.HostBatch``) like ctr_refine_batch;
    ``clusters`` restricts the work to a subset (the others are left untouched)."""
    fmax = {}
    todo = range(batch.n_clusters) if clusters is None else clusters
    for c in todo:
        a, b = batch.feat_offset[c], batch.feat_offset[c + 1]
        fi = int(batch.frame_index[c])
        if fi not in fmax:
            fmax[fi] = batch.frames[fi].max()
        out, cost, status, rounds, iters = refine_cluster(
            batch.frames[fi], fmax[fi], batch.params[a:b], batch.low[a:b], batch.high[a:b],
            problem, tol, maxiter)
        batch.params_out[a:b] = out
        batch.cost[c] = cost
        batch.status[c] = status
        batch.n_rounds[c] = rounds
        batch.n_iter[c] = iters
    return batch
