"""Small host-side helpers mirroring reference ``clustertracking/utils.py``."""
import numpy as np


class RefineException(Exception):
    """Per-cluster refinement failure (reference utils.py:96-97).  The engine
    reports these as data (status code + NaN cost), never as a raised error."""


def validate_tuple(value, ndim):
    """scalar -> (v,)*ndim; iterable of length ndim -> tuple; else ValueError.
    (``trackpy.utils.validate_tuple`` semantics as used at reference
    refine.py:30,285 and masks.py:11,53.)"""
    if not hasattr(value, '__iter__'):
        return (value,) * ndim
    if len(value) == ndim:
        return tuple(value)
    raise ValueError("List length should have same length as image dimensions.")


def guess_pos_columns(f):
    """reference utils.py:24-29"""
    return ['z', 'y', 'x'] if 'z' in f else ['y', 'x']


def default_pos_columns(ndim):
    """reference utils.py:40-41"""
    return ['z', 'y', 'x'][-ndim:]


def default_size_columns(ndim, isotropic):
    """reference utils.py:44-49"""
    if isotropic:
        return ['size']
    return ['size_z', 'size_y', 'size_x'][-ndim:]


def is_isotropic(value):
    """reference utils.py:52-56 (all entries equal, or a scalar)"""
    if hasattr(value, '__iter__'):
        value = tuple(value)
        return bool(np.all(value[1:] == value[:-1]))
    return True


class ArrayReader(object):
    """A video held as one C-contiguous array ``[T, *frame_shape]``.

    Behaves like the ``FramesSequence`` the reference expects
    (``frame_shape`` attribute + integer indexing, reference refine.py:252-255)
    and additionally exposes ``.array`` so the engine can hand the whole block
    to the device without restacking frames."""

    def __init__(self, array):
        self.array = np.ascontiguousarray(array)
        self.frame_shape = self.array.shape[1:]

    def __len__(self):
        return self.array.shape[0]

    def __getitem__(self, i):
        return self.array[i]
