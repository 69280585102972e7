"""ctypes wrapper of oracle/_build/libctr_oracle.so (the C oracle).

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, '_build', 'libctr_oracle.so')
_lib = None


def build(force=False):
    if force or not os.path.exists(LIB_PATH) or \
            os.path.getmtime(LIB_PATH) < os.path.getmtime(os.path.join(HERE, 'ctr_oracle.c')):
        subprocess.check_call(['make', '-s', '-C', HERE])
    return LIB_PATH


def load():
    global _lib
    if _lib is None:
        from clustertracking_amd import _abi
        build()
        lib = C.CDLL(LIB_PATH)
        P = C.POINTER
        lib.ctro_refine_batch.argtypes = [P(_abi.Problem), P(_abi.Batch), C.c_int]
        lib.ctro_refine_batch.restype = C.c_int
        lib.ctro_cluster_n_vars.argtypes = [P(_abi.Problem), C.c_int]
        lib.ctro_window.argtypes = [C.c_int] + [C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 2
        lib.ctro_mask_counts.argtypes = [C.c_int] + [C.c_void_p] * 3 + [C.c_int, C.c_void_p]
        lib.ctro_mask_counts.restype = C.c_long
        lib.ctro_objective.argtypes = [P(_abi.Problem), P(_abi.Batch), C.c_int64] + \
            [C.c_void_p] * 8
        lib.ctro_objective.restype = C.c_int
        _lib = lib
    return _lib


def run_batch(problem, batch, n_threads=1):
    """Same contract as ctr_refine_batch: fills batch.params_out/cost/status..."""
    lib = load()
    b = batch.as_struct()
    rc = lib.ctro_refine_batch(C.byref(problem), C.byref(b), int(n_threads))
    if rc != 0:
        raise ValueError("ctro_refine_batch rejected the descriptor (%d)" % rc)
    return batch


def window(shape, radius, coords):
    lib = load()
    coords = np.ascontiguousarray(np.atleast_2d(coords), dtype=np.float64)
    nd = coords.shape[1]
    shape = np.asarray(shape, dtype=np.int64)
    radius = np.asarray(radius, dtype=np.int32)
    origin = np.zeros(3, np.int32)
    wshape = np.zeros(3, np.int32)
    ok = lib.ctro_window(nd, shape.ctypes.data, radius.ctypes.data, coords.ctypes.data,
                         len(coords), origin.ctypes.data, wshape.ctypes.data)
    if not ok:
        return None, (0,) * nd
    return tuple(origin[:nd]), tuple(wshape[:nd])


def mask_counts(shape, radius, coords):
    lib = load()
    coords = np.ascontiguousarray(np.atleast_2d(coords), dtype=np.float64)
    nd = coords.shape[1]
    shape = np.asarray(shape, dtype=np.int64)
    radius = np.asarray(radius, dtype=np.int32)
    per = np.zeros(len(coords), np.int64)
    P = lib.ctro_mask_counts(nd, shape.ctypes.data, radius.ctypes.data,
                             coords.ctypes.data, len(coords), per.ctypes.data)
    return P, per


def objective(problem, batch, cluster, v_in=None):
    """(F, vect, grad, bounds[nv,2], origin, wshape, P) at the packed start vector
    (or at ``v_in`` with the masks kept at the start coordinates)."""
    from clustertracking_amd import _abi
    lib = load()
    b = batch.as_struct()
    F = C.c_double()
    P = C.c_int64()
    vect = np.zeros(_abi.MAX_VARS)
    grad = np.zeros(_abi.MAX_VARS)
    bounds = np.zeros((_abi.MAX_VARS, 2))
    origin = np.zeros(3, np.int32)
    wshape = np.zeros(3, np.int32)
    nv = lib.ctro_objective(C.byref(problem), C.byref(b), int(cluster),
                            C.addressof(F), vect.ctypes.data, grad.ctypes.data,
                            bounds.ctypes.data, origin.ctypes.data, wshape.ctypes.data,
                            C.addressof(P),
                            None if v_in is None else
                            np.ascontiguousarray(v_in, dtype=np.float64).ctypes.data)
    if nv < 0:
        raise ValueError("ctro_objective failed (%d)" % nv)
    nd = problem.ndim
    return (F.value, vect[:nv], grad[:nv], bounds[:nv], tuple(origin[:nd]),
            tuple(wshape[:nd]), P.value)


def hessian(problem, batch, cluster, v_in=None, exact=True):
    """Model Hessian of F at ``v_in`` (or the packed start vector): 2 (J^T J + Q) / (P norm),
    Q = the exact second-order part of solve() for (signal, positions); exact=False -> J^T J only;
    exact='full' -> the second-order part in ALL variables, any modes (what compute_error uses)."""
    from clustertracking_amd import _abi
    lib = load()
    b = batch.as_struct()
    H = np.zeros((_abi.MAX_VARS, _abi.MAX_VARS))
    lib.ctro_hessian.argtypes = [C.POINTER(_abi.Problem), C.POINTER(_abi.Batch), C.c_int64, C.c_void_p,
                                 C.c_int, C.c_void_p]
    lib.ctro_hessian.restype = C.c_int
    nv = lib.ctro_hessian(C.byref(problem), C.byref(b), int(cluster),
                          None if v_in is None else
                          np.ascontiguousarray(v_in, dtype=np.float64).ctypes.data,
                          2 if exact == 'full' else int(bool(exact)), H.ctypes.data)
    if nv < 0:
        raise ValueError("ctro_hessian failed (%d)" % nv)
    return H.reshape(-1)[:nv * nv].reshape(nv, nv).copy()


def gaussian_kernel(sigma):
    """The restated trackpy.masks.gaussian_kernel(sigma, 4) of the oracle (parity unpinned)."""
    lib = load()
    w = np.zeros(2 * 16 + 1 + 8)
    lib.ctro_gaussian_kernel.argtypes = [C.c_double, C.c_void_p]
    lib.ctro_gaussian_kernel.restype = C.c_int
    lw = lib.ctro_gaussian_kernel(float(sigma), w.ctypes.data)
    return w[:2 * lw + 1].copy()


def lowpass(window_pixels, noise_size, threshold=0.):
    """preprocessing.py:12-49 on one window (refine.py:37-40): the oracle's filter."""
    lib = load()
    win = np.array(window_pixels, dtype=np.float64, order='C')
    nd = win.ndim
    ns = np.zeros(3)
    ns[:nd] = noise_size
    shape = np.asarray(win.shape, dtype=np.int32)
    lib.ctro_lowpass.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p]
    lib.ctro_lowpass.restype = None
    lib.ctro_lowpass(nd, shape.ctypes.data, ns.ctypes.data, float(threshold), win.ctypes.data)
    return win
