"""Import shim for the *reference* package (this container only).

TEST INFRASTRUCTURE - never imported by the product path.

The reference (``/root/reference/clustertracking``) depends on ``trackpy`` and
``pims`` which are not installed, and uses NumPy aliases removed in NumPy 2.
This module registers the minimal stand-ins that the *refine hot path* needs
 and then imports the reference from where it lies.  No reference
arithmetic is altered:

* ``trackpy.utils.validate_tuple``: scalar -> (v,)*ndim, len-ndim iterable ->
  tuple, else ValueError (behaviour inferred from refine.py:30,285 and
  masks.py:11,53 call sites).
* ``trackpy.masks.gaussian_kernel`` (needed by the reference's ``lowpass``,
  preprocessing.py:43, only when ``noise_size`` is given): restated from trackpy's published
  source (trackpy 0.3/0.4 ``masks.py``: ``lw = int(truncate*sigma + 0.5); x = arange(-lw, lw+1);
  exp(x**2/(-2*sigma**2)) / sum``) -- PARITY UNPINNED for this one function; the correlation
  itself is SciPy's ``correlate1d``, which is installed.
* ``np.bool / np.int / np.float / np.Inf`` aliases (used at refine.py:49,
  masks.py:54, find_link.py:527).
* ``masks.slice_image`` indexes with a *list* of slices (masks.py:68), an
  IndexError on NumPy >= 1.23; rebound to index with ``tuple(slices)``.

Nothing here runs on the GPU box: ``/root/reference`` does not exist there.
Only ``tests/golden/make_golden.py`` (fixture generation) and ``-m "not gpu"``
validation tests that skip when the reference is absent use it.
"""
import os
import sys
import types

import numpy as np

REFERENCE_ROOT = os.environ.get("CTR_REFERENCE_ROOT", "/root/reference")


def available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "clustertracking"))


def _validate_tuple(value, ndim):
    if not hasattr(value, '__iter__'):
        return (value,) * ndim
    if len(value) == ndim:
        return tuple(value)
    raise ValueError("List length should have same length as image dimensions.")


def _gaussian_kernel(sigma, truncate=4.0):
    "1D discretized gaussian (restated trackpy.masks.gaussian_kernel; see the module docstring)"
    lw = int(truncate * sigma + 0.5)
    x = np.arange(-lw, lw + 1)
    result = np.exp(x ** 2 / (-2 * sigma ** 2))
    return result / np.sum(result)


def _not_available(*args, **kwargs):
    raise NotImplementedError("trackpy/pims is not installed; this code path "
                              "is outside the refine hot path")


_loaded = None


def load():
    """Return the imported reference package (cached)."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not available():
        raise ImportError("reference not present at %s" % REFERENCE_ROOT)

    for alias, typ in (("bool", bool), ("int", int), ("float", float), ("Inf", np.inf)):
        if not hasattr(np, alias):   # removed in NumPy 2 (np.Inf: find_link.py:527)
            setattr(np, alias, typ)

    tp = types.ModuleType("trackpy")
    tp_utils = types.ModuleType("trackpy.utils")
    tp_utils.validate_tuple = _validate_tuple
    tp_pre = types.ModuleType("trackpy.preprocessing")
    tp_pre.bandpass = _not_available
    tp_pre.scalefactor_to_gamut = _not_available
    tp_pre.scale_to_gamut = _not_available
    tp_masks = types.ModuleType("trackpy.masks")
    tp_masks.gaussian_kernel = _gaussian_kernel
    tp_masks.r_squared_mask = _not_available
    tp_masks.x_squared_masks = _not_available
    tp_masks.binary_mask = _not_available
    tp_masks.N_binary_mask = _not_available
    tp_find = types.ModuleType("trackpy.find")
    tp_find.grey_dilation = _not_available
    tp.utils = tp_utils
    tp.preprocessing = tp_pre
    tp.masks = tp_masks
    tp.find = tp_find
    tp.refine = _not_available
    tp.annotate = _not_available
    tp.annotate3d = _not_available

    class Frame(np.ndarray):
        def __new__(cls, arr, frame_no=None, metadata=None):
            obj = np.asarray(arr).view(cls)
            obj.frame_no = frame_no
            obj.metadata = metadata or {}
            return obj

        def __array_finalize__(self, obj):
            self.frame_no = getattr(obj, 'frame_no', None)
            self.metadata = getattr(obj, 'metadata', {})

    class FramesSequence(object):
        def __getitem__(self, i):
            return self.get_frame(i)

        def __iter__(self):
            return (self.get_frame(i) for i in range(len(self)))

    pims = types.ModuleType("pims")
    pims.Frame = Frame
    pims.FramesSequence = FramesSequence
    pims.pipeline = lambda f: f

    for name, mod in (("trackpy", tp), ("trackpy.utils", tp_utils),
                      ("trackpy.preprocessing", tp_pre),
                      ("trackpy.masks", tp_masks), ("trackpy.find", tp_find),
                      ("pims", pims)):
        sys.modules.setdefault(name, mod)

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import clustertracking as ct
    import clustertracking.masks as ct_masks
    import clustertracking.refine as ct_refine

    def slice_image(coords, image, radius):
        slices, origin = ct_masks.slices_multiple(coords, image.shape, radius)
        return image[tuple(slices)], origin

    ct_masks.slice_image = slice_image
    ct_refine.slice_image = slice_image
    ct.quiet()
    _loaded = ct
    return ct


class _NumpyCompat(object):
    """``np`` as the reference's ``artificial.draw_feature`` needs it under NumPy 2: the one call
    that fails there, ``np.array(coords)`` on the SPARSE meshgrid (artificial.py:139: arrays of
    shapes (n,1) and (1,m), an inhomogeneous list since NumPy 1.24), gets the grids broadcast
    against each other first -- the same values the old object-array arithmetic summed.  Every
    other attribute is NumPy's."""

    def __getattr__(self, name):
        return getattr(np, name)

    @staticmethod
    def array(obj, *args, **kwargs):
        if isinstance(obj, (list, tuple)) and len(obj) > 1 and all(isinstance(o, np.ndarray) for o in obj) \
                and len({o.shape for o in obj}) > 1:
            obj = np.broadcast_arrays(*obj)
        return np.array(obj, *args, **kwargs)


class ListIndexArray(np.ndarray):
    """An image the reference can index with a LIST of slices (artificial.py:141 ``image[rect] +=``;
    an IndexError on NumPy >= 1.23): the list is turned into the tuple it meant."""

    def __getitem__(self, key):
        return np.ndarray.__getitem__(self, tuple(key) if isinstance(key, list) else key)

    def __setitem__(self, key, value):
        np.ndarray.__setitem__(self, tuple(key) if isinstance(key, list) else key, value)


def reference_draw_feature(image, position, size, max_value, feat_func='gauss', **kwargs):
    """The REFERENCE's ``artificial.draw_feature`` (artificial.py:81-141) run as it is on a copy of
    ``image`` (see the two classes above for what NumPy 2 needs); returns the drawn image."""
    ct = load()
    import clustertracking.artificial as ref_art
    ref_art.np = _NumpyCompat()
    try:
        im = np.array(image).view(ListIndexArray)
        ref_art.draw_feature(im, position, size, max_value, feat_func, **kwargs)
    finally:
        ref_art.np = np
    return np.asarray(im)

