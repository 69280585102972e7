"""Pins oracle/ref_numpy.py (NumPy objective + SciPy SLSQP, the reference's own
minimiser) to the reference: on the golden fixtures it reproduces the
reference's default-tolerance output (oracle A) and, with tol=1e-14, its
converged output (oracle B)."""
import numpy as np
import pytest
from numpy.testing import assert_allclose, assert_equal

import _cases
import ref_numpy

CASES = ['cfg1_triple', 'edges', 'rms_threshold', 'dimer_constrained', 'trimer_constrained',
         'iso2d_signal_cluster', 'aniso2d_sizevar', 'iso3d_default', 'dtype_f32',
         'video_2frames', 'overlap_d21_bigshift']


@pytest.mark.parametrize("name", CASES)
def test_reproduces_reference_default_tolerance(name):
    case = _cases.Case(name)
    res = case.run(lambda p, b: ref_numpy.run_batch(p, b))
    A = case.ref('A')
    pc = case.pos_columns
    assert_equal(np.isnan(res['cost'].values), np.isnan(A['cost'].values))
    ok = ~np.isnan(A['cost'].values)
    # same objective, same SLSQP: agreement far below the solver tolerance
    assert np.abs(res[pc].values - A[pc].values)[ok].max() < 1e-6
    assert_allclose(res['cost'].values[ok], A['cost'].values[ok], rtol=0, atol=1e-8)
    assert_allclose(res['signal'].values[ok], A['signal'].values[ok], rtol=1e-5)


@pytest.mark.parametrize("name", ['cfg1_triple', 'edges', 'dimer_constrained'])
def test_reproduces_reference_converged(name):
    case = _cases.Case(name)
    res = case.run(lambda p, b: ref_numpy.run_batch(p, b, tol=1e-14, maxiter=1000))
    B = case.ref('B')
    pc = case.pos_columns
    ok = ~np.isnan(B['cost'].values)
    assert np.abs(res[pc].values - B[pc].values)[ok].max() < 1e-6
