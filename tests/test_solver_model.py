"""The solver's model Hessian and its stopping rules, on the CPU oracle (the engine runs the
same sequence: tests/tools/iter_parity.py shows equal iteration counts cluster by cluster, and
tests/test_gpu_parity.py compares the results).

The reference leaves the optimisation to SciPy's SLSQP (refine.py:373-375); the engine's
bounded Levenberg-Marquardt loop is its own, so these tests pin its pieces:
  * the exact second-order part of the model Hessian against finite differences of the
    gradient that the reference's known answers already pin (tests/test_golden_oracle.py);
  * regressions of the two stopping-rule faults found with tests/tools/check_vs_reference.py.
"""
import numpy as np
import pytest

import _cases
import ctr_oracle


def _numeric_hessian(problem, batch, cluster, v, h=1e-5):
    nv = len(v)
    H = np.zeros((nv, nv))
    for j in range(nv):
        e = np.zeros(nv)
        e[j] = h * max(1., abs(v[j]))
        gp = ctr_oracle.objective(problem, batch, cluster, v + e)[2]
        gm = ctr_oracle.objective(problem, batch, cluster, v - e)[2]
        H[:, j] = (gp - gm) / (2. * e[j])
    return H


@pytest.mark.parametrize("name", ['cfg1_triple', 'iso3d_default', 'aniso2d_default'])
def test_exact_hessian_equals_finite_differences_of_the_gradient(name):
    # default modes (sizes constant): every second derivative of the residual is in the model
    case = _cases.Case(name)
    prep = case.prepare()
    b = prep.batch
    cluster = int(np.argmax(np.diff(b.feat_offset)))
    F, v, g, bounds, origin, wshape, P = ctr_oracle.objective(prep.problem, b, cluster)
    v = v + 0.05 * np.cos(np.arange(len(v)))       # away from the start, still inside the masks
    H = ctr_oracle.hessian(prep.problem, b, cluster, v, exact=True)
    Hn = _numeric_hessian(prep.problem, b, cluster, v)
    scale = np.sqrt(np.outer(np.abs(np.diag(Hn)), np.abs(np.diag(Hn)))) + 1e-12
    assert np.abs(H - Hn).max() / scale.max() < 1e-6
    assert (np.abs(H - Hn) / scale).max() < 1e-4
    # and the Gauss-Newton part alone is NOT the Hessian (the test would be vacuous otherwise)
    Hgn = ctr_oracle.hessian(prep.problem, b, cluster, v, exact=False)
    assert (np.abs(Hgn - Hn) / scale).max() > 1e-3


@pytest.mark.parametrize("name", ['cfg1_triple', 'iso2d_sizevar', 'aniso2d_sizevar', 'aniso3d_sizevar',
                                  'iso2d_signal_cluster', 'iso2d_signal_const', 'hard_bg_at_bound_modes',
                                  'hard_cons_trimer_sizecluster'])
def test_full_second_order_equals_finite_differences_for_any_modes(name):
    """The Hessian that compute_error inverts (solution_std -> full_second_order): every second
    derivative of the residual, whatever the parameter modes (free / shared sizes, shared /
    constant signal)."""
    case = _cases.Case(name)
    prep = case.prepare()
    b = prep.batch
    cluster = int(np.argmax(np.diff(b.feat_offset)))
    F, v, g, bounds, origin, wshape, P = ctr_oracle.objective(prep.problem, b, cluster)
    v = v + 0.05 * np.cos(np.arange(len(v)))
    H = ctr_oracle.hessian(prep.problem, b, cluster, v, exact='full')
    Hn = _numeric_hessian(prep.problem, b, cluster, v)
    scale = np.sqrt(np.outer(np.abs(np.diag(Hn)), np.abs(np.diag(Hn)))) + 1e-12
    assert np.abs(H - Hn).max() / scale.max() < 1e-6
    assert (np.abs(H - Hn) / scale).max() < 1e-4
    assert np.array_equal(H, H.T)


def test_exact_part_is_limited_to_signal_and_positions_when_sizes_vary():
    case = _cases.Case('iso2d_sizevar')
    prep = case.prepare()
    b = prep.batch
    cluster = int(np.argmax(np.diff(b.feat_offset)))
    n = int(np.diff(b.feat_offset)[cluster])
    F, v, g, bounds, origin, wshape, P = ctr_oracle.objective(prep.problem, b, cluster)
    H = ctr_oracle.hessian(prep.problem, b, cluster, v, exact=True)
    Hgn = ctr_oracle.hessian(prep.problem, b, cluster, v, exact=False)
    Hn = _numeric_hessian(prep.problem, b, cluster, v)
    # vect layout (fitfunc.py:207-263): [bg, signal x n, y x n, x x n, size x n]
    sp = np.arange(1, 1 + 3 * n)
    same_feature = (sp[:, None] - 1) % n == (sp[None, :] - 1) % n
    blk = np.ix_(sp, sp)
    scale = np.abs(np.diag(Hn)).max()
    assert (np.abs(H[blk] - Hn[blk])[same_feature]).max() / scale < 1e-6
    size_cols = np.arange(1 + 3 * n, 1 + 4 * n)
    assert np.array_equal(H[np.ix_(size_cols, size_cols)], Hgn[np.ix_(size_cols, size_cols)])


@pytest.mark.parametrize("name", ['hard_bg_at_bound', 'hard_bg_at_bound_modes'])
def test_projected_uphill_step_is_not_convergence(name):
    """Background on its lower bound, gradient pointing inward: the projected Newton step had
    a NEGATIVE predicted decrease, which the stopping test `pred <= ftol * S` took for
    convergence after two iterations (cost 0.0152 instead of the reference's 0.0120)."""
    case = _cases.Case(name)
    res = case.run(_cases.oracle_runner())
    ref = case.ref('B')
    rm, mx, ok_a, ok_b = _cases.compare(res, ref, case.pos_columns)
    assert (ok_a == ok_b).all()
    assert mx < 1e-6
    np.testing.assert_allclose(res['cost'].values, ref['cost'].values, rtol=1e-6)


@pytest.mark.parametrize("name", ['hard_valley_triple', 'hard_valley_pair'])
def test_valleys_converge_within_the_iteration_limit(name):
    """Nearly coincident features: with J^T J alone the loop needs 100..1900 iterations (the
    reference's SLSQP converges within its 100); with the exact Hessian a few tens."""
    case = _cases.Case(name)
    prep = case.prepare()
    ctr_oracle.run_batch(prep.problem, prep.batch, 1)
    assert (prep.batch.status == 0).all()
    assert prep.batch.n_iter.max() < 60
    res = case.run(_cases.oracle_runner())
    rm, mx, ok_a, ok_b = _cases.compare(res, case.ref('B'), case.pos_columns)
    assert (ok_a == ok_b).all() and ok_a.all()
    assert mx < 1e-6


def test_gauss_newton_only_switch_reproduces_the_stall():
    # the diagnostic switch of the oracle: without the exact part the valley pair hits the limit
    case = _cases.Case('hard_valley_pair')
    prep = case.prepare()
    lib = ctr_oracle.load()
    lib.ctro_set_newton(0)
    try:
        ctr_oracle.run_batch(prep.problem, prep.batch, 1)
        assert prep.batch.n_iter.max() >= 100
    finally:
        lib.ctro_set_newton(1)


def _pack(problem, batch, cluster, table):
    """Packed vector (fitfunc.py:207-263, groups=None) of a parameter table of one cluster."""
    F, v, g, bounds, origin, wshape, P = ctr_oracle.objective(problem, batch, cluster)
    n = int(np.diff(batch.feat_offset)[cluster])
    out = []
    for k in range(problem.n_params):
        mode = problem.modes[k]
        if mode == 0:
            continue
        out.extend(table[:, k] if mode == 1 else [table[:, k].mean()])
    out = np.array(out, dtype=np.float64)
    assert len(out) == len(v)
    return out


@pytest.mark.parametrize("name", ['cfg1_triple', 'cfg2_frame_noisy', 'aniso3d_default', 'iso2d_sizevar',
                                  'aniso2d_sizevar', 'iso2d_signal_cluster', 'iso2d_signal_const'])
def test_parameter_standard_deviations(name):
    """compute_error (refine.py:400-406): sqrt(2 diag(inv(Hessian of F))) at the solution.  The
    reference takes the Hessian by finite differences (numdifftools, absent here: parity with
    the reference itself is unpinned); the oracle's values are checked against the same
    formula on a finite-difference Hessian of the oracle's own gradient."""
    from clustertracking_amd import _abi
    case = _cases.Case(name)
    prep = case.prepare()
    b0 = prep.batch
    b = _abi.HostBatch(b0.frames, b0.frame_index, b0.feat_offset, b0.params, b0.low, b0.high,
                       want_std=True)
    ctr_oracle.run_batch(prep.problem, b, 1)
    size = np.diff(b.feat_offset)
    checked = 0
    for c in np.flatnonzero((b.status == 0) & (b.n_rounds == 1))[:6]:   # masks still at p0
        rows = slice(b.feat_offset[c], b.feat_offset[c + 1])
        v = _pack(prep.problem, b, c, b.params_out[rows])
        Hn = _numeric_hessian(prep.problem, b, c, v)
        std = np.sqrt(2. * np.diag(np.linalg.inv(Hn)))
        got = _pack(prep.problem, b, c, b.params_std[rows])
        np.testing.assert_allclose(got, std, rtol=2e-4)
        checked += 1
    assert checked > 0
    # constant parameters carry no error; failed clusters none at all
    const_cols = [k for k in range(prep.problem.n_params) if prep.problem.modes[k] == 0]
    assert np.isnan(b.params_std[:, const_cols]).all()
    assert np.isnan(b.params_std[np.repeat(b.status != 0, size)]).all()
