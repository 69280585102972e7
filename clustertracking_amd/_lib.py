"""ctypes binding of ``libctrefine.so`` (the HIP engine, C-ABI ``include/ctrefine.h``).

There is no CPU fallback: if the library is missing, or no MI355X is visible,
the calls raise.
"""
import ctypes as C
import os
import threading

from . import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('CTREFINE_LIB') or os.path.join(_HERE, 'csrc', 'libctrefine.so')

# every symbol include/ctrefine.h declares
EXPORTS = ('ctr_abi_version', 'ctr_create', 'ctr_destroy', 'ctr_last_error',
           'ctr_validate_problem', 'ctr_cluster_n_vars', 'ctr_refine_batch',
           'ctr_plan_create', 'ctr_plan_destroy', 'ctr_refine_batch_device',
           'ctr_frame_max_device', 'ctr_synchronize', 'ctr_last_kernel_ms',
           'ctr_find_clusters', 'ctr_engine_wait_stream', 'ctr_stream_wait_engine',
           'ctr_draw_frames_device', 'ctr_query_done', 'ctr_ipc_alloc', 'ctr_ipc_open',
           'ctr_ipc_probe', 'ctr_ipc_read', 'ctr_ipc_close', 'ctr_ipc_free')

_lib = None
_lock = threading.Lock()


class EngineError(RuntimeError):
    pass


def _share_hip_runtime_with_torch():
    """One HIP runtime per process: PyTorch-ROCm wheels bundle their own
    libamdhip64.so.7 (+ HSA runtime).  If libctrefine.so pulled in the system
    copy first, a later ``import torch`` would bind to that one and then fail to
    see the GPU.  So when torch is installed, its copy is loaded first and
    libctrefine.so (DT_NEEDED libamdhip64.so.7) binds to it."""
    import importlib.util
    import sys
    if 'torch' in sys.modules:
        return
    try:
        spec = importlib.util.find_spec('torch')
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], 'lib', 'libamdhip64.so')
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load():
    """Load the shared library once; raises EngineError when it is absent."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        _share_hip_runtime_with_torch()
        if not os.path.exists(LIB_PATH):
            raise EngineError(
                "HIP engine not built: %s is missing. Build it with "
                "`python -c 'import __graft_entry__ as g; g.build()'` or "
                "`make -C clustertracking_amd/csrc`. There is no CPU fallback."
                % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        P = C.POINTER
        lib.ctr_abi_version.restype = C.c_int
        lib.ctr_create.argtypes = [P(C.c_void_p), C.c_int]
        lib.ctr_create.restype = C.c_int
        lib.ctr_destroy.argtypes = [C.c_void_p]
        lib.ctr_destroy.restype = None
        lib.ctr_last_error.argtypes = [C.c_void_p]
        lib.ctr_last_error.restype = C.c_char_p
        lib.ctr_validate_problem.argtypes = [P(_abi.Problem), C.c_char_p, C.c_int]
        lib.ctr_validate_problem.restype = C.c_int
        lib.ctr_cluster_n_vars.argtypes = [P(_abi.Problem), C.c_int]
        lib.ctr_cluster_n_vars.restype = C.c_int
        lib.ctr_refine_batch.argtypes = [C.c_void_p, P(_abi.Problem), P(_abi.Batch)]
        lib.ctr_refine_batch.restype = C.c_int
        lib.ctr_plan_create.argtypes = [C.c_void_p, P(_abi.Problem), C.c_int64,
                                        C.c_void_p, P(C.c_void_p)]
        lib.ctr_plan_create.restype = C.c_int
        lib.ctr_plan_destroy.argtypes = [C.c_void_p]
        lib.ctr_plan_destroy.restype = None
        lib.ctr_refine_batch_device.argtypes = [C.c_void_p, C.c_void_p,
                                                P(_abi.Batch), C.c_void_p]
        lib.ctr_refine_batch_device.restype = C.c_int
        lib.ctr_frame_max_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int32,
                                             C.c_int64, C.c_int64, C.c_void_p,
                                             C.c_void_p]
        lib.ctr_frame_max_device.restype = C.c_int
        if hasattr(lib, 'ctr_find_clusters'):   # absent only in older diagnostic builds
            lib.ctr_find_clusters.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                              C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
            lib.ctr_find_clusters.restype = C.c_int
        lib.ctr_synchronize.argtypes = [C.c_void_p, C.c_void_p]
        lib.ctr_synchronize.restype = C.c_int
        for name in ('ctr_engine_wait_stream', 'ctr_stream_wait_engine'):
            if hasattr(lib, name):   # (absent from libraries built before they existed)
                getattr(lib, name).argtypes = [C.c_void_p, C.c_void_p]
                getattr(lib, name).restype = C.c_int
        lib.ctr_draw_frames_device.argtypes = [C.c_void_p, P(_abi.Synth), C.c_void_p, C.c_void_p]
        lib.ctr_draw_frames_device.restype = C.c_int
        lib.ctr_query_done.argtypes = [C.c_void_p]
        lib.ctr_query_done.restype = C.c_int
        lib.ctr_ipc_alloc.argtypes = [C.c_void_p, C.c_int64, P(C.c_void_p), C.c_void_p]
        lib.ctr_ipc_open.argtypes = [C.c_void_p, C.c_void_p, P(C.c_void_p)]
        lib.ctr_ipc_probe.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        lib.ctr_ipc_read.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
        lib.ctr_ipc_close.argtypes = [C.c_void_p, C.c_void_p]
        lib.ctr_ipc_free.argtypes = [C.c_void_p, C.c_void_p]
        for fn in (lib.ctr_ipc_alloc, lib.ctr_ipc_open, lib.ctr_ipc_probe, lib.ctr_ipc_read,
                   lib.ctr_ipc_close, lib.ctr_ipc_free):
            fn.restype = C.c_int
        lib.ctr_last_kernel_ms.argtypes = [C.c_void_p, P(C.c_double), P(C.c_double)]
        lib.ctr_last_kernel_ms.restype = C.c_int
        if lib.ctr_abi_version() != _abi.ABI_VERSION:
            raise EngineError("libctrefine.so ABI version mismatch")
        _lib = lib
        return lib


class Engine(object):
    """One engine handle bound to one GPU (``ctr_create`` / ``ctr_destroy``)."""

    def __init__(self, device=0):
        self._lib = load()
        self._h = C.c_void_p()
        rc = self._lib.ctr_create(C.byref(self._h), int(device))
        if rc != _abi.OK:
            msg = self._lib.ctr_last_error(None)
            self._h = None
            raise EngineError("ctr_create(device=%d) failed (%d): %s" % (
                device, rc, (msg or b'').decode()))
        self.device = int(device)

    def close(self):
        if getattr(self, '_h', None):
            self._lib.ctr_destroy(self._h)
            self._h = None

    __del__ = close

    def _check(self, rc, what):
        if rc != _abi.OK:
            msg = (self._lib.ctr_last_error(self._h) or b'').decode()
            if rc == _abi.ERR_INVALID:
                raise ValueError("%s: %s" % (what, msg))
            if rc == _abi.ERR_UNSUPPORTED:
                raise NotImplementedError("%s: %s" % (what, msg))
            raise EngineError("%s failed (%d): %s" % (what, rc, msg))

    def refine_batch(self, problem, batch):
        """Host-buffer call: ``batch`` is an ``_abi.HostBatch``; outputs are
        written into its arrays."""
        b = batch.as_struct()
        self._check(self._lib.ctr_refine_batch(self._h, C.byref(problem), C.byref(b)),
                    'ctr_refine_batch')
        return batch

    # ---- device-resident path (bench, multi-GPU driver) ---------------------
    def plan(self, problem, feat_offset_host):
        import numpy as np
        off = np.ascontiguousarray(feat_offset_host, dtype=np.int32)
        plan = C.c_void_p()
        self._check(self._lib.ctr_plan_create(self._h, C.byref(problem), len(off) - 1,
                                              off.ctypes.data, C.byref(plan)),
                    'ctr_plan_create')
        return Plan(self._lib, plan)

    def refine_batch_device(self, plan, batch_struct, stream=None):
        self._check(self._lib.ctr_refine_batch_device(
            self._h, plan._p, C.byref(batch_struct), C.c_void_p(stream or 0)),
            'ctr_refine_batch_device')

    def frame_max_device(self, frames_ptr, dtype_code, n_frames, frame_elems, out_ptr,
                         stream=None):
        self._check(self._lib.ctr_frame_max_device(
            self._h, C.c_void_p(frames_ptr), dtype_code, n_frames, frame_elems,
            C.c_void_p(out_ptr), C.c_void_p(stream or 0)), 'ctr_frame_max_device')

    def find_clusters(self, pos, frame_offset, separation):
        """Labels (smallest row index of the cluster) and sizes for a frame-sorted
        position table; ``ctr_find_clusters``."""
        import numpy as np
        pos = np.ascontiguousarray(pos, dtype=np.float64)
        off = np.ascontiguousarray(frame_offset, dtype=np.int32)
        sep = np.ascontiguousarray(separation, dtype=np.float64)
        n, nd = pos.shape
        labels = np.empty(n, dtype=np.int32)
        sizes = np.empty(n, dtype=np.int32)
        self._check(self._lib.ctr_find_clusters(self._h, nd, pos.ctypes.data, off.ctypes.data,
                                                len(off) - 1, sep.ctypes.data,
                                                labels.ctypes.data, sizes.ctypes.data),
                    'ctr_find_clusters')
        return labels, sizes

    def draw_frames_device(self, synth, frames_ptr, stream=None):
        """``ctr_draw_frames_device``: ``synth`` is an ``_abi.Synth`` with device pointers."""
        self._check(self._lib.ctr_draw_frames_device(self._h, C.byref(synth), C.c_void_p(frames_ptr),
                                                     C.c_void_p(stream or 0)), 'ctr_draw_frames_device')

    def query_done(self):
        """True when the last ``refine_batch_device`` call of this engine has finished on the
        device (``ctr_query_done``; never blocks)."""
        rc = self._lib.ctr_query_done(self._h)
        if rc < 0:
            raise EngineError("ctr_query_done failed: %s" % (self._lib.ctr_last_error(self._h) or b'').decode())
        return rc == 1

    def synchronize(self, stream=None):
        self._check(self._lib.ctr_synchronize(self._h, C.c_void_p(stream or 0)),
                    'ctr_synchronize')

    # ---- the inbox of a multi-GPU pipeline (include/ctrefine.h: ctr_ipc_*) ---------------------
    def ipc_alloc(self, n_bytes):
        """(device pointer, handle blob) of a zeroed block on this engine's device that other
        processes can map (the blob names the owning device, include/ctrefine.h)."""
        ptr = C.c_void_p()
        handle = (C.c_ubyte * _abi.IPC_HANDLE_BYTES)()
        self._check(self._lib.ctr_ipc_alloc(self._h, C.c_int64(int(n_bytes)), C.byref(ptr), handle), 'ctr_ipc_alloc')
        return int(ptr.value), bytes(handle)

    def ipc_open(self, handle):
        ptr = C.c_void_p()
        if len(handle) != _abi.IPC_HANDLE_BYTES:
            raise ValueError("an inbox handle has %d bytes" % _abi.IPC_HANDLE_BYTES)
        buf = (C.c_ubyte * _abi.IPC_HANDLE_BYTES).from_buffer_copy(handle)
        self._check(self._lib.ctr_ipc_open(self._h, buf, C.byref(ptr)), 'ctr_ipc_open')
        return int(ptr.value)

    def ipc_probe(self, ptr, value):
        self._check(self._lib.ctr_ipc_probe(self._h, C.c_void_p(int(ptr)), C.c_int64(int(value))), 'ctr_ipc_probe')

    def ipc_read(self, ptr, shape, dtype):
        import numpy as np
        out = np.empty(shape, dtype=dtype)
        self._check(self._lib.ctr_ipc_read(self._h, C.c_void_p(out.ctypes.data), C.c_void_p(int(ptr)),
                                           C.c_int64(out.nbytes)), 'ctr_ipc_read')
        return out

    def ipc_close(self, ptr):
        self._check(self._lib.ctr_ipc_close(self._h, C.c_void_p(int(ptr))), 'ctr_ipc_close')

    def ipc_free(self, ptr):
        self._check(self._lib.ctr_ipc_free(self._h, C.c_void_p(int(ptr))), 'ctr_ipc_free')

    def engine_wait_stream(self, stream=0):
        """The engine's own stream waits (on the device) for what is queued on ``stream``
        (raw handle; 0 = the legacy default stream)."""
        self._check(self._lib.ctr_engine_wait_stream(self._h, C.c_void_p(stream or 0)),
                    'ctr_engine_wait_stream')

    def stream_wait_engine(self, stream=0):
        """``stream`` waits (on the device) for what is queued on the engine's own stream."""
        self._check(self._lib.ctr_stream_wait_engine(self._h, C.c_void_p(stream or 0)),
                    'ctr_stream_wait_engine')

    def last_kernel_ms(self):
        a, b = C.c_double(), C.c_double()
        self._check(self._lib.ctr_last_kernel_ms(self._h, C.byref(a), C.byref(b)),
                    'ctr_last_kernel_ms')
        return a.value, b.value


class Plan(object):
    def __init__(self, lib, p):
        self._lib, self._p = lib, p

    def close(self):
        if self._p:
            self._lib.ctr_plan_destroy(self._p)
            self._p = None

    __del__ = close


_default_engines = {}


def default_engine(device=0):
    eng = _default_engines.get(device)
    if eng is None:
        eng = _default_engines[device] = Engine(device)
    return eng
