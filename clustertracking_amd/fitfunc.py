"""Host-side description of the fit model: parameter layout, modes, bounds.

Mirrors the *metadata* half of reference ``clustertracking/fitfunc.py``
(``FitFunctions.__init__`` :325-413, ``validate_bounds`` :492-533,
``compute_bounds`` :535-558).  The arithmetic half (residual / gradient
closures, :421-489) is what the HIP engine replaces and is not present here.
"""
import warnings

import numpy as np

from .utils import default_pos_columns, default_size_columns

# reference fitfunc.py:9-11
MODE_DICT = {0: 0, 1: 1, 2: 2, 3: 3, 4: 4, 5: 5, 6: 6,
             'const': 0, 'var': 1, 'global': 2, 'cluster': 3,
             'particle': 4, 'frame': 5}

# enum shared with include/ctrefine.h (CTR_FIT_*)
FIT_FUNCTION_CODES = {'gauss': 0, 'ring': 1, 'disc': 2, 'inv_series': 3}
# reference fitfunc.py:195-204 (function_templates) for the profiles the engine implements
FIT_TEMPLATES = dict(gauss=dict(params=[], default={}, continuous=True),
                     ring=dict(params=['thickness'], default=dict(thickness=0.5), continuous=False),
                     disc=dict(params=['disc_size'], default=dict(disc_size=0.5), continuous=False))


class FitFunctions(object):
    """Parameter names/order and modes for one refine call.

    Column order of the per-feature parameter matrix (reference
    fitfunc.py:353-354): ``[background, signal, (z,) y, x, size | size_(z,)y,x]``.
    ``'ring'`` and ``'disc'`` add one profile parameter after the sizes (``thickness`` /
    ``disc_size``, fitfunc.py:195-204), ``'inv_series_<N>'`` N + 1 (``signal_mult``, ``param_a``,
    ..., all 1 by default; fitfunc.py:148-154,334-343); custom (dict) functions are rejected with a
    clear message.
    """

    def __init__(self, fit_function='gauss', ndim=2, isotropic=True,
                 param_mode=None):
        if isinstance(fit_function, dict):
            raise NotImplementedError(
                "custom (dict) fit functions need Python callbacks per "
                "evaluation and are not supported by the MI355X engine")
        if fit_function in FIT_TEMPLATES:
            tmpl = FIT_TEMPLATES[fit_function]
        else:
            # fitfunc.py:334-343: '<name>_<order>' = a template with generated parameter names
            head, _, order = str(fit_function).rpartition('_')
            if head != 'inv_series' or not order.isdigit():
                raise ValueError("Unknown fit function {}".format(fit_function))
            names = ['signal_mult'] + ['param_' + chr(i) for i in range(97, 97 + int(order))]
            tmpl = dict(params=names, default={p: 1. for p in names}, continuous=True)
        self.fit_function = fit_function
        self.ndim = int(ndim)
        self.isotropic = bool(isotropic)
        self.pos_columns = default_pos_columns(ndim)
        self.size_columns = default_size_columns(ndim, isotropic)
        # reference fitfunc.py:195-204: the profile's own parameters and their defaults
        self._params = list(tmpl['params'])
        self.default = dict(background=0., **tmpl['default'])
        self.continuous = tmpl['continuous']
        self.params = ['background', 'signal'] + self.pos_columns + \
            self.size_columns + self._params

        # fitfunc.py:356-387
        mode = dict(signal='var', background='cluster')
        if param_mode is not None:
            mode.update(param_mode)
        if 'pos' in mode:
            for col in self.pos_columns:
                mode.setdefault(col, mode['pos'])
            del mode['pos']
        if (not isotropic) and ('size' in mode):
            for col in self.size_columns:
                mode.setdefault(col, mode['size'])
            del mode['size']
        mode = {key: MODE_DICT[val] for key, val in mode.items()}
        for col in self.pos_columns:
            mode.setdefault(col, 1)
        for col in self.params:
            mode.setdefault(col, 0)
        # fitfunc.py:389-392
        if mode['background'] == 1:
            warnings.warn('The background param mode cannot vary per feature. '
                          'Varying per cluster now.')
            mode['background'] = 3
        self.param_mode = mode
        self.modes = [int(mode[p]) for p in self.params]

    @property
    def n_params(self):
        return len(self.params)

    def validate_bounds(self, bounds=None, radius=None):
        """Three ``[2, n_params]`` templates: absolute, +-difference and
        +-relative difference; NaN = no bound (reference fitfunc.py:492-533)."""
        if bounds is None:
            bounds = dict()
        n = len(self.params)
        tmpl_abs = np.empty((2, n), dtype=np.float64)
        tmpl_diff = np.empty((2, n), dtype=np.float64)
        tmpl_rel = np.empty((2, n), dtype=np.float64)
        nan = np.nan
        for i, name in enumerate(self.params):
            b_abs = bounds.get(name, nan)
            b_diff = bounds.get(name + '_diff', nan)
            b_rel = bounds.get(name + '_rel_diff', nan)
            for group, key in ((self.pos_columns, 'pos'), (self.size_columns, 'size')):
                if name in group:
                    if b_abs is nan:
                        b_abs = bounds.get(key, nan)
                    if b_diff is nan:
                        b_diff = bounds.get(key + '_diff', nan)
                    if b_rel is nan:
                        b_rel = bounds.get(key + '_rel_diff', nan)
            if b_abs is nan and name in ['background', 'signal'] + self.size_columns:
                b_abs = (0., nan)          # positive by default (:518-521)
            if b_diff is nan and name in self.pos_columns:
                r = float(radius[self.pos_columns.index(name)])
                b_diff = (r, r)            # stay inside the mask (:523-527)
            tmpl_abs[:, i] = b_abs
            tmpl_diff[:, i] = b_diff
            tmpl_rel[:, i] = b_rel
        return tmpl_abs, tmpl_diff, tmpl_rel

    def feature_bounds(self, templates, params):
        """Per-feature ``low, high [N, n_params]`` (reference
        fitfunc.py:541-550, the part of ``compute_bounds`` before packing).
        Packing to the per-cluster vector (min of lows / max of highs for
        shared parameters, :554-557) happens inside the engine."""
        b_abs, b_diff, b_rel = templates
        params = np.asarray(params, dtype=np.float64)
        n, npar = params.shape
        low = np.full((n, npar), -np.inf)
        high = np.full((n, npar), np.inf)
        # column by column, only the terms whose template is set (NaN = none):
        # np.nanmax([p - diff, p * (1 - rel), abs], axis=0) of the reference, -inf where all are NaN
        for k in range(npar):
            p = params[:, k]
            lo_terms, hi_terms = [], []
            if not np.isnan(b_diff[0][k]):
                lo_terms.append(p - b_diff[0][k])
            if not np.isnan(b_rel[0][k]):
                lo_terms.append(p * (1 - b_rel[0][k]))
            if not np.isnan(b_diff[1][k]):
                hi_terms.append(p + b_diff[1][k])
            if not np.isnan(b_rel[1][k]):
                hi_terms.append(p * (1 + b_rel[1][k]))
            for terms, out, absb, red in ((lo_terms, low, b_abs[0][k], np.fmax), (hi_terms, high, b_abs[1][k], np.fmin)):
                col = None
                for t in terms:
                    col = t if col is None else red(col, t)
                if col is None:
                    if not np.isnan(absb):
                        out[:, k] = absb
                    continue
                if not np.isnan(absb):
                    col = red(col, absb)
                nanmask = np.isnan(col)          # a NaN parameter: no bound from it
                if nanmask.any():
                    col = np.where(nanmask, -np.inf if out is low else np.inf, col)
                out[:, k] = col
        return low, high
