"""Host-side behaviour of the drop-in API (no GPU): the C-ABI library loads
and exports every symbol of include/ctrefine.h, argument handling mirrors the
reference (refine.py:242-341), unsupported combinations fail loudly."""
import ctypes
import os
import re

import numpy as np
import pandas as pd
import pytest
from numpy.testing import assert_allclose, assert_equal

import _cases
import clustertracking_amd as cta
from clustertracking_amd import _abi, _lib, constraints


def small_problem(seed=0, n=12, shape=(96, 96)):
    im, truth, p0 = cta.artificial.random_frame(shape, n, 3., 100, 10, seed, margin=13)
    f0 = pd.DataFrame(p0, columns=['y', 'x'])
    f0['signal'] = 90.
    f0['size'] = 3.
    f0['background'] = 5.
    return im, truth, f0


def test_header_symbols_exported():
    """every function include/ctrefine.h declares is exported by libctrefine.so"""
    header = open(os.path.join(_cases.ROOT, 'include', 'ctrefine.h')).read()
    declared = set(re.findall(r'\b(ctr_[a-z_]+)\s*\(', header))
    assert declared == set(_lib.EXPORTS)
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.ctr_abi_version() == _abi.ABI_VERSION


def test_struct_layout_matches_header(tmp_path):
    """ctypes mirror vs the C compiler's view of include/ctrefine.h"""
    import subprocess
    fields_p = [f[0] for f in _abi.Problem._fields_]
    fields_b = [f[0] for f in _abi.Batch._fields_]
    src = '#include <stdio.h>\n#include <stddef.h>\n#include "ctrefine.h"\nint main(void){\n'
    src += 'printf("%zu %zu\\n", sizeof(ctr_problem), sizeof(ctr_batch));\n'
    for f in fields_p:
        src += 'printf("%%zu\\n", offsetof(ctr_problem, %s));\n' % f
    for f in fields_b:
        src += 'printf("%%zu\\n", offsetof(ctr_batch, %s));\n' % f
    src += 'return 0;}\n'
    c = tmp_path / 'layout.c'
    c.write_text(src)
    exe = tmp_path / 'layout'
    subprocess.check_call(['gcc', '-I', os.path.join(_cases.ROOT, 'include'), str(c), '-o', str(exe)])
    out = subprocess.check_output([str(exe)]).decode().split()
    assert int(out[0]) == ctypes.sizeof(_abi.Problem)
    assert int(out[1]) == ctypes.sizeof(_abi.Batch)
    offs = [int(x) for x in out[2:]]
    expect = [getattr(_abi.Problem, f).offset for f in fields_p] + \
             [getattr(_abi.Batch, f).offset for f in fields_b]
    assert offs == expect


def test_validate_problem_without_device():
    lib = _lib.load()
    msg = ctypes.create_string_buffer(256)
    p = _abi.make_problem(2, True, [3, 1, 1, 1, 0], (6, 6))
    assert lib.ctr_validate_problem(ctypes.byref(p), msg, 256) == _abi.OK
    assert lib.ctr_cluster_n_vars(ctypes.byref(p), 3) == 10   # [bg, s0..2, y0..2, x0..2]
    p.modes[1] = _abi.MODE_GLOBAL
    assert lib.ctr_validate_problem(ctypes.byref(p), msg, 256) == _abi.ERR_UNSUPPORTED
    assert b'global' in msg.value
    p = _abi.make_problem(2, True, [3, 1, 1, 1, 0], (6, 6))
    p.fit_function = _abi.FIT_RING          # a ring has one more column (thickness)
    assert lib.ctr_validate_problem(ctypes.byref(p), msg, 256) == _abi.ERR_INVALID
    for fit in ('ring', 'disc'):
        p = _abi.make_problem(2, True, [3, 1, 1, 1, 0, 0], (6, 6), fit_function=fit)
        assert p.n_params == 6
        assert lib.ctr_validate_problem(ctypes.byref(p), msg, 256) == _abi.OK
        p = _abi.make_problem(3, False, [3, 1, 1, 1, 1, 1, 1, 1, 3], (4, 6, 6), fit_function=fit)
        assert p.n_params == 9 and lib.ctr_cluster_n_vars(ctypes.byref(p), 2) == 2 + 2 * 7
        assert lib.ctr_validate_problem(ctypes.byref(p), msg, 256) == _abi.OK
        p = _abi.make_problem(2, True, [3, 1, 1, 1, 0, 0], (6, 6), fit_function=fit, noise_size=(1, 1))
        assert lib.ctr_validate_problem(ctypes.byref(p), msg, 256) == _abi.ERR_UNSUPPORTED
    # inv_series_<N>: N + 1 columns behind the sizes, as many as CTR_MAX_PARAMS leaves room for
    p = _abi.make_problem(2, True, [3, 1, 1, 1, 0], (6, 6))
    p.fit_function = _abi.FIT_INV_SERIES    # ... at least 'signal_mult'
    assert lib.ctr_validate_problem(ctypes.byref(p), msg, 256) == _abi.ERR_INVALID
    p = _abi.make_problem(2, True, [3, 1, 1, 1, 0] + [0] * 7, (6, 6), fit_function='inv_series_6')
    assert p.n_params == 12 == _abi.MAX_PARAMS and lib.ctr_validate_problem(ctypes.byref(p), msg, 256) == _abi.OK
    p = _abi.make_problem(3, False, [3, 1, 1, 1, 1, 0, 0, 0, 0, 3, 1, 0], (4, 6, 6), fit_function='inv_series_3')
    assert lib.ctr_validate_problem(ctypes.byref(p), msg, 256) == _abi.OK
    assert lib.ctr_cluster_n_vars(ctypes.byref(p), 3) == 2 + 3 * 5
    with pytest.raises(NotImplementedError):
        _abi.make_problem(3, False, [0] * 13, (4, 6, 6), fit_function='inv_series_4')
    p.n_params = 13
    assert lib.ctr_validate_problem(ctypes.byref(p), msg, 256) == _abi.ERR_UNSUPPORTED
    p.fit_function = 4
    assert lib.ctr_validate_problem(ctypes.byref(p), msg, 256) == _abi.ERR_INVALID
    p = _abi.make_problem(2, True, [3, 1, 1, 1, 0], (6, 6), max_iter=0)
    assert lib.ctr_validate_problem(ctypes.byref(p), msg, 256) == _abi.ERR_INVALID
    p = _abi.make_problem(2, True, [1, 1, 1, 1, 0], (6, 6))
    assert lib.ctr_validate_problem(ctypes.byref(p), msg, 256) == _abi.ERR_INVALID


def test_no_silent_cpu_fallback():
    """Without a GPU the product path raises; it never computes on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    im, truth, f0 = small_problem()
    with pytest.raises(_lib.EngineError):
        cta.refine_leastsq(f0, im, 13)


def test_product_never_imports_oracle():
    pkg = os.path.join(_cases.ROOT, 'clustertracking_amd')
    for fn in os.listdir(pkg):
        if fn.endswith('.py'):
            src = open(os.path.join(pkg, fn)).read()
            assert 'ctr_oracle' not in src and 'refshim' not in src, fn
            assert 'import oracle' not in src and 'from oracle' not in src, fn


def test_default_modes_and_layout():
    ff = cta.FitFunctions('gauss', 2, True)
    assert ff.params == ['background', 'signal', 'y', 'x', 'size']
    assert ff.modes == [3, 1, 1, 1, 0]
    ff = cta.FitFunctions('gauss', 3, False, dict(size='var', signal='cluster'))
    assert ff.params == ['background', 'signal', 'z', 'y', 'x', 'size_z', 'size_y', 'size_x']
    assert ff.modes == [3, 3, 1, 1, 1, 1, 1, 1]
    ff = cta.FitFunctions('gauss', 2, True, dict(pos='const', size='var'))
    assert ff.modes == [3, 1, 0, 0, 1]
    # the profiles with a parameter of their own (reference fitfunc.py:195-204)
    ff = cta.FitFunctions('ring', 2, False, dict(thickness='cluster'))
    assert ff.params == ['background', 'signal', 'y', 'x', 'size_y', 'size_x', 'thickness']
    assert ff.modes == [3, 1, 1, 1, 0, 0, 3] and ff.default['thickness'] == 0.5 and not ff.continuous
    ff = cta.FitFunctions('disc', 3, True)
    assert ff.params[-1] == 'disc_size' and ff.modes[-1] == 0 and ff.default['disc_size'] == 0.5
    # fitfunc.py:334-343: generated names, all 1 by default, "continuous"
    ff = cta.FitFunctions('inv_series_3', 2, True, dict(param_b='var'))
    assert ff.params[-4:] == ['signal_mult', 'param_a', 'param_b', 'param_c'] and ff.continuous
    assert ff.modes == [3, 1, 1, 1, 0, 0, 0, 1, 0] and ff.default['param_c'] == 1.
    for bad in ('lorentz', 'inv_series', 'inv_series_x', 'gauss_2'):
        with pytest.raises(ValueError):
            cta.FitFunctions(bad, 2, True)
    with pytest.warns(UserWarning):
        ff = cta.FitFunctions('gauss', 2, True, dict(background='var'))
    assert ff.modes[0] == 3


def test_unsupported_is_loud(oracle):
    im, truth, f0 = small_problem()
    run = _cases.oracle_runner()
    for kw in (dict(param_mode=dict(signal='global')), dict(fit_function=dict(params=[], func=None))):
        with pytest.raises(NotImplementedError):
            _cases.refine_leastsq(f0.copy(), im, 13, _run_batch=run, **kw)
    # compute_error goes with every param_mode (second derivatives in all variables)
    for mode, col in ((dict(size='var'), 'size_std'), (dict(signal='cluster'), 'signal_std')):
        res = _cases.refine_leastsq(f0.copy(), im, 13, compute_error=True, param_mode=mode, _run_batch=run)
        assert np.isfinite(res[col]).all() and (res[col] > 0).all()
    with pytest.raises(ValueError):      # lowpass sigma beyond CTR_MAX_NOISE_SIZE
        _cases.refine_leastsq(f0.copy(), im, 13, noise_size=5, _run_batch=run)
    with pytest.raises(ValueError):
        _cases.refine_leastsq(f0.copy(), im, 13, fit_function='nonsense', _run_batch=run)
    with pytest.raises(ValueError):
        _cases.refine_leastsq(f0.copy(), im, 13, max_iter=0, _run_batch=run)
    with pytest.raises(TypeError):
        _cases.refine_leastsq(f0.copy(), im, 13, bogus=1, _run_batch=run)
    with pytest.raises(NotImplementedError):
        constraints.dimer_global(1.)
    with pytest.raises(NotImplementedError):
        _cases.refine_leastsq(f0.copy(), im, 13, _run_batch=run,
                           constraints=constraints.dimer(6.) + constraints.trimer(6.))
    with pytest.raises(NotImplementedError):
        _cases.refine_leastsq(f0.copy(), im, 13, _run_batch=run,
                           constraints=[dict(type='eq', fun=lambda x: 0., cluster_size=2)])
    with pytest.raises(ValueError):   # SciPy: lower bound exceeds upper bound
        _cases.refine_leastsq(f0.copy(), im, 13, _run_batch=run, bounds=dict(signal=(500, 1000),
                           signal_rel_diff=0.1))
    with pytest.raises(AssertionError):
        _cases.refine_leastsq(f0.copy(), np.zeros((4, 8, 8)), 13, _run_batch=run)


def test_side_effects_and_output_shape(oracle):
    """refine.py:274-281,296-305,426-427: t_column is added to the INPUT in place,
    the result is a copy with cluster, cluster_size, missing param columns, cost."""
    im, truth, f0 = small_problem()
    f0 = f0.drop(columns=['background'])
    f_in = f0.copy()
    res = _cases.refine_leastsq(f_in, im, 13, _run_batch=_cases.oracle_runner())
    assert 'frame' in f_in and (f_in['frame'] == 0).all()
    assert_equal(f_in[['y', 'x']].values, f0[['y', 'x']].values)   # input rows untouched
    for col in ('cluster', 'cluster_size', 'background', 'cost', 'frame'):
        assert col in res
    assert len(res) == len(f0)
    assert np.isfinite(res['cost']).all()
    d = res[['y', 'x']].values - truth
    assert np.sqrt(np.mean(d ** 2)) < 0.1
    # accepted-and-ignored SciPy kwargs
    res2 = _cases.refine_leastsq(f0.copy(), im, 13, method='SLSQP', tol=1e-6,
                              options=dict(maxiter=100, disp=False),
                              _run_batch=_cases.oracle_runner())
    assert_allclose(res2[['y', 'x']].values, res[['y', 'x']].values, atol=1e-12)


def test_frame_no_attribute_and_video_order(oracle):
    """Frame objects with frame_no (refine.py:264-274) and unsorted videos."""
    class Frame(np.ndarray):
        pass
    im, truth, f0 = small_problem(3)
    fr = im.view(Frame)
    fr.frame_no = 7
    res = _cases.refine_leastsq(f0.copy(), fr, 13, _run_batch=_cases.oracle_runner())
    assert (res['frame'] == 7).all()
    # rows of a two-frame video come back grouped by frame (find.py:157)
    im2, truth2, f2 = small_problem(4)
    f0a, f0b = f0.copy(), f2.copy()
    f0a['frame'] = 1
    f0b['frame'] = 0
    video = cta.ArrayReader(np.stack([im2, im]))
    both = pd.concat([f0a, f0b], ignore_index=True)
    res = _cases.refine_leastsq(both, video, 13, _run_batch=_cases.oracle_runner())
    assert_equal(res['frame'].values, np.sort(both['frame'].values))
    single = _cases.refine_leastsq(f0.copy(), im, 13, _run_batch=_cases.oracle_runner())
    sub = res[res['frame'] == 1]
    assert_allclose(sub[['y', 'x']].values, single[['y', 'x']].values, atol=1e-12)
    # cluster ids keep running across frames (find.py:120-128)
    assert res.loc[res['frame'] == 1, 'cluster'].min() > res.loc[res['frame'] == 0, 'cluster'].max()


def test_accuracy_vs_truth_like_reference_suite(oracle):
    """tests/test_refine.py:598-688 thresholds with a fixed seed: 20 features on a
    grid, size 4, signal 160, uint8; RMS position error < 0.01 px noise-free,
    < 0.05 px at S/N 10, < 0.1 px at S/N 3; signal within 1 % noise-free."""
    rng = np.random.RandomState(42)
    size, signal, sep = 4., 160, 32
    pos = np.array([[32 + sep * (i // 5), 32 + sep * (i % 5)] for i in range(20)], float)
    pos += rng.uniform(-.5, .5, pos.shape)
    shape = (int(pos[:, 0].max()) + 32, int(pos[:, 1].max()) + 32)
    clean = np.zeros(shape, np.uint8)
    sig = signal * rng.uniform(0.8, 1.2, 20)
    for p, s in zip(pos, sig):
        cta.artificial.draw_gaussian(clean, p, size, s)
    for noise, tol in ((0, 0.01), (16, 0.05), (48, 0.1)):
        im = cta.artificial.add_poisson_noise(clean, noise, rng) if noise else clean
        dev = rng.uniform(-1, 1, pos.shape)
        dev *= rng.uniform(0, 1, (20, 1)) * 0.5 * size / np.sqrt((dev ** 2).sum(1))[:, None]
        f0 = pd.DataFrame(pos + dev, columns=['y', 'x'])
        f0['signal'] = float(signal)
        f0['size'] = size
        f0['background'] = noise / 2.
        res = _cases.refine_leastsq(f0, im, 16, _run_batch=_cases.oracle_runner())
        assert not np.isnan(res['cost']).any()
        rms = np.sqrt(np.mean((res[['y', 'x']].values - pos) ** 2))
        assert rms < tol, (noise, rms)
        if noise == 0:
            assert np.sqrt(np.mean((1 - res['signal'].values / sig) ** 2)) < 0.01
