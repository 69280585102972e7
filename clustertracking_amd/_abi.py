"""ctypes mirror of ``include/ctrefine.h`` and the host-side batch container.

Pure description of the C-ABI (structs, enums) plus a NumPy container for one
batch of clusters.  No arithmetic.
"""
import ctypes as C

import numpy as np

ABI_VERSION = 7
IPC_HANDLE_BYTES = 128
MAX_NDIM = 3
MAX_NOISE_SIZE = 4.0
MAX_PARAMS = 12
MAX_VARS = 127

OK, ERR_INVALID, ERR_UNSUPPORTED, ERR_DEVICE, ERR_NOMEM = range(5)

DTYPE_CODES = {np.dtype(np.uint8): 0, np.dtype(np.uint16): 1, np.dtype(np.int16): 2,
               np.dtype(np.int32): 3, np.dtype(np.float32): 4, np.dtype(np.float64): 5}

FIT_GAUSS, FIT_RING, FIT_DISC, FIT_INV_SERIES = 0, 1, 2, 3
FIT_CODES = {'gauss': FIT_GAUSS, 'ring': FIT_RING, 'disc': FIT_DISC}
FIT_EXTRAS = {FIT_GAUSS: 0, FIT_RING: 1, FIT_DISC: 1}   # profile parameters after the sizes


def fit_code(fit_function):
    """(CTR_FIT_* code, number of profile parameters) of a fit function name; ``'inv_series_<N>'``
    has N + 1 of them (fitfunc.py:334-343)."""
    if fit_function in FIT_CODES:
        code = FIT_CODES[fit_function]
        return code, FIT_EXTRAS[code]
    head, _, order = str(fit_function).rpartition('_')
    if head == 'inv_series' and order.isdigit():
        return FIT_INV_SERIES, int(order) + 1
    raise ValueError("fit_function must be one of %s or 'inv_series_<N>'" % sorted(FIT_CODES))
MODE_CONST, MODE_VAR, MODE_GLOBAL, MODE_CLUSTER = 0, 1, 2, 3
CONS_NONE, CONS_DIMER, CONS_TRIMER, CONS_TETRAMER = 0, 1, 2, 3
CONS_CODES = {None: 0, 'dimer': 1, 'trimer': 2, 'tetramer': 3}

STATUS_OK = 0
STATUS_OUT_OF_BOUNDS = 1
STATUS_NONFINITE = 2
STATUS_NO_CONVERGENCE = 3
STATUS_RMS_DEV = 4
STATUS_TOO_LARGE = 5
STATUS_TEXT = {
    0: 'ok',
    1: 'coordinates are out of image bounds',
    2: 'non-finite initial parameters',
    3: 'solver did not converge',
    4: 'rms deviation of the fit is more than max_rms_dev',
    5: 'cluster is beyond the engine (a feature with more than 48 overlapping neighbours)',
}


FLAG_THROUGHPUT = 1   # ctr_problem.flags (include/ctrefine.h): scheduling hint, results unchanged
FLAG_ISOLATE_TAIL = 2  # only the kernel of the likely slow fits beside the main stream
FLAG_WINDOW_FILTER = 4  # noise_size was given: the window is thresholded even with every sigma 0


class Problem(C.Structure):
    _fields_ = [
        ('ndim', C.c_int32), ('isotropic', C.c_int32), ('fit_function', C.c_int32),
        ('n_params', C.c_int32), ('modes', C.c_int32 * MAX_PARAMS),
        ('radius', C.c_int32 * MAX_NDIM), ('constraint_kind', C.c_int32),
        ('max_iter', C.c_int32), ('solver_maxiter', C.c_int32), ('flags', C.c_int32),
        ('constraint_dist', C.c_double * MAX_NDIM), ('max_shift', C.c_double),
        ('max_rms_dev', C.c_double), ('residual_factor', C.c_double),
        ('xtol', C.c_double), ('ftol', C.c_double), ('threshold', C.c_double),
        ('noise_size', C.c_double * MAX_NDIM),
    ]


class Batch(C.Structure):
    _fields_ = [
        ('frames', C.c_void_p), ('frame_dtype', C.c_int32), ('reserved0', C.c_int32),
        ('n_frames', C.c_int64), ('shape', C.c_int64 * MAX_NDIM),
        ('n_clusters', C.c_int64), ('n_features', C.c_int64),
        ('frame_index', C.c_void_p), ('feat_offset', C.c_void_p),
        ('params', C.c_void_p), ('low', C.c_void_p), ('high', C.c_void_p),
        ('params_out', C.c_void_p), ('cost', C.c_void_p), ('status', C.c_void_p),
        ('n_rounds', C.c_void_p), ('n_iter', C.c_void_p), ('params_std', C.c_void_p),
        ('result_rows', C.c_void_p), ('done_flag', C.c_void_p), ('done_value', C.c_int64),
    ]


class Synth(C.Structure):
    """``ctr_synth`` (include/ctrefine.h): synthetic frames on the device."""
    _fields_ = [
        ('ndim', C.c_int32), ('frame_dtype', C.c_int32), ('n_frames', C.c_int64),
        ('shape', C.c_int64 * MAX_NDIM), ('n_features', C.c_int64),
        ('frame_of', C.c_void_p), ('pos', C.c_void_p), ('size', C.c_void_p),
        ('max_value', C.c_void_p), ('noise', C.c_double), ('seed', C.c_uint64),
    ]


def make_problem(ndim, isotropic, modes, radius, constraint=None, max_iter=10,
                 max_shift=1., max_rms_dev=1., residual_factor=100000.,
                 solver_maxiter=100, xtol=0., ftol=0., noise_size=None, threshold=None,
                 fit_function='gauss'):
    """Fill a ``ctr_problem``.  ``constraint`` = None or (kind, dist[ndim]); ``noise_size`` = None
    or one sigma per axis (refine.py:37-40)."""
    p = Problem()
    p.ndim = int(ndim)
    p.isotropic = int(bool(isotropic))
    p.fit_function, n_extra = fit_code(fit_function)
    p.n_params = 2 + ndim + (1 if isotropic else ndim) + n_extra
    if p.n_params > MAX_PARAMS:
        raise NotImplementedError("%s in %dD%s: %d parameter columns, the engine takes %d"
                                  % (fit_function, ndim, '' if isotropic else ' anisotropic', p.n_params, MAX_PARAMS))
    if len(modes) != p.n_params:
        raise ValueError("modes must have %d entries" % p.n_params)
    for i, m in enumerate(modes):
        p.modes[i] = int(m)
    for i, r in enumerate(radius):
        p.radius[i] = int(r)
    if constraint is not None:
        kind, dist = constraint
        p.constraint_kind = CONS_CODES[kind]
        for i, d in enumerate(dist):
            p.constraint_dist[i] = float(d)
    p.max_iter = int(max_iter)
    p.solver_maxiter = int(solver_maxiter)
    p.max_shift = float(max_shift)
    p.max_rms_dev = float(max_rms_dev)
    p.residual_factor = float(residual_factor)
    p.xtol = float(xtol)
    p.ftol = float(ftol)
    if noise_size is not None:
        for i, sgm in enumerate(noise_size):
            if not (0 <= float(sgm) <= MAX_NOISE_SIZE):
                raise ValueError("noise_size must be between 0 and %g" % MAX_NOISE_SIZE)
            p.noise_size[i] = float(sgm)
        p.threshold = 0. if threshold is None else float(threshold)   # refine.py:38-39
        p.flags |= FLAG_WINDOW_FILTER                                 # refine.py:37: `is not None`
    return p


class HostBatch(object):
    """NumPy-owned buffers of one batch + the ``ctr_batch`` view onto them."""

    def __init__(self, frames, frame_index, feat_offset, params, low, high, want_std=False):
        frames = np.asarray(frames)
        if frames.dtype not in DTYPE_CODES:
            # every other pixel type is widened to f64 exactly like
            # ``.astype(np.float64)`` in the reference (refine.py:58)
            frames = frames.astype(np.float64)
        self.frames = np.ascontiguousarray(frames)
        self.frame_index = np.ascontiguousarray(frame_index, dtype=np.int32)
        self.feat_offset = np.ascontiguousarray(feat_offset, dtype=np.int32)
        self.params = np.ascontiguousarray(params, dtype=np.float64)
        self.low = np.ascontiguousarray(low, dtype=np.float64)
        self.high = np.ascontiguousarray(high, dtype=np.float64)
        n_cl = len(self.frame_index)
        if len(self.feat_offset) != n_cl + 1:
            raise ValueError("feat_offset must have n_clusters + 1 entries")
        n_feat = int(self.feat_offset[-1]) if n_cl else 0
        if self.params.ndim != 2 or self.params.shape[0] != n_feat:
            raise ValueError("params must be [n_features, n_params]")
        if self.low.shape != self.params.shape or self.high.shape != self.params.shape:
            raise ValueError("low/high must match params")
        if n_cl and (self.frame_index.min() < 0 or
                     self.frame_index.max() >= self.frames.shape[0]):
            raise ValueError("frame_index out of range")
        if n_cl and np.any(np.diff(self.feat_offset) < 0):
            raise ValueError("feat_offset must be non-decreasing")
        self.params_out = np.empty_like(self.params)
        self.cost = np.full(n_cl, np.nan)
        self.status = np.zeros(n_cl, dtype=np.int32)
        self.n_rounds = np.zeros(n_cl, dtype=np.int32)
        self.n_iter = np.zeros(n_cl, dtype=np.int32)
        # refine.py:400-406 (compute_error): one standard deviation per fitted parameter
        self.params_std = np.full_like(self.params, np.nan) if want_std else None

    @property
    def n_clusters(self):
        return len(self.frame_index)

    @property
    def n_features(self):
        return self.params.shape[0]

    def as_struct(self):
        b = Batch()
        b.frames = self.frames.ctypes.data
        b.frame_dtype = DTYPE_CODES[self.frames.dtype]
        b.n_frames = self.frames.shape[0]
        for i, s in enumerate(self.frames.shape[1:]):
            b.shape[i] = int(s)
        b.n_clusters = self.n_clusters
        b.n_features = self.n_features
        for name in ('frame_index', 'feat_offset', 'params', 'low', 'high',
                     'params_out', 'cost', 'status', 'n_rounds', 'n_iter'):
            setattr(b, name, getattr(self, name).ctypes.data)
        b.params_std = self.params_std.ctypes.data if self.params_std is not None else None
        return b
