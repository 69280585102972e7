"""Generate the golden fixtures under tests/golden/ from the REFERENCE.

Run in the build container only (needs /root/reference; see oracle/refshim.py):

    python tests/golden/make_golden.py

Every fixture is a small ``.npz`` holding *data only*: the input frames, the
initial feature table, the call options (JSON) and what the reference
returned -- with its defaults (oracle A: SLSQP tol=1e-6, maxiter=100) and
converged (oracle B: tol=1e-14, maxiter=1000; legal kwargs pass-through,
reference refine.py:225-228,242-244) -- plus intermediate known answers of the
reference's own helpers for a few clusters (window origin/shape, mask pixel
counts, objective and gradient at the start vector, packed bounds).

Where the reference itself aborts (bare ``raise RefineException`` makes
``e.args[0]`` an IndexError at refine.py:417) the fixture
records ``ref_aborts=1`` and carries no reference output for that call.
"""
import json
import os
import sys
import warnings

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle'))

import refshim  # noqa: E402
from clustertracking_amd import artificial  # noqa: E402

ct = refshim.load()
from clustertracking import constraints as ct_constraints  # noqa: E402
from clustertracking import refine as ct_refine  # noqa: E402
from clustertracking.fitfunc import FitFunctions, vect_from_params  # noqa: E402
from clustertracking.find import find_clusters  # noqa: E402
from clustertracking.masks import slices_multiple  # noqa: E402

TIGHT = dict(tol=1e-14, options=dict(maxiter=1000, disp=False))


def _constraints(spec):
    if spec is None:
        return None
    kind, dist, ndim = spec['kind'], spec['dist'], spec['ndim']
    return getattr(ct_constraints, kind)(np.array(dist, dtype=float), ndim)


def _df_to_arrays(df, prefix, out):
    out[prefix + 'columns'] = np.array(list(df.columns))
    out[prefix + 'values'] = df.values.astype(np.float64)
    out[prefix + 'index'] = np.asarray(df.index)


def run_reference(f0, frames, call, extra):
    """Call the reference; frames [T, ...]; returns DataFrame or None if it aborts."""
    kwargs = dict(call)
    kwargs['constraints'] = _constraints(kwargs.pop('constraints', None))
    kwargs.update(extra)
    diameter = kwargs.pop('diameter')
    f = f0.copy()
    if frames.shape[0] == 1 and 'frame' not in f0:
        reader = frames[0]
    else:
        class Reader(object):
            frame_shape = frames.shape[1:]

            def __getitem__(self, i):
                return frames[i]
        reader = Reader()
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        try:
            return ct.refine_leastsq(f, reader, diameter, **kwargs)
        except IndexError:
            return None


def intermediates(f0, frames, call, n_clusters=6):
    """Known answers of the reference's helpers on the first few clusters."""
    out = {}
    ndim = frames.ndim - 1
    diameter = call['diameter']
    if not hasattr(diameter, '__iter__'):
        diameter = (diameter,) * ndim
    radius = tuple(int(d) // 2 for d in diameter)
    isotropic = all(d == diameter[0] for d in diameter)
    ff = FitFunctions(call.get('fit_function', 'gauss'), ndim, isotropic, call.get('param_mode'))
    out['im_params'] = np.array(ff.params)
    out['im_modes'] = np.array(ff.modes)
    tmpl = ff.validate_bounds(call.get('bounds'), radius=radius)
    out['im_bounds_templates'] = np.array(tmpl)
    f = f0.copy()
    if 'frame' not in f:
        f['frame'] = 0
    sep = call.get('separation') or diameter
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        f = find_clusters(f, sep, ff.pos_columns if hasattr(ff, 'pos_columns') else None)
    pv = call.get('param_val') or {}
    for col in pv:
        f[col] = pv[col]
    for col in set(ff.params) - set(f.columns):
        f[col] = ff.default[col]
    out['im_cluster'] = f['cluster'].values
    out['im_cluster_size'] = f['cluster_size'].values
    out['im_row_index'] = np.asarray(f.index)
    k = 0
    for (frame_no, cl), f_iter in f.groupby(['frame', 'cluster']):
        if k >= n_clusters:
            break
        params = f_iter[ff.params].values.astype(np.float64)
        if not np.isfinite(params).all():
            continue
        image = frames[int(frame_no)]
        coords = params[:, 2:2 + ndim]
        slices, origin = slices_multiple(coords, image.shape, radius)
        if origin is None:
            continue
        sub, mesh, masks = ct_refine.prepare_subimage(coords, image, radius, call.get('noise_size'),
                                                      call.get('threshold'))
        norm = float(image.max()) ** 2 / float(call.get('residual_factor', 100000.))
        residual, jacobian = ff.get_residual([sub], [mesh], [masks], params, None, norm)
        vect = vect_from_params(params, ff.modes, None, operation=np.mean)
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            bnds = ff.compute_bounds(tmpl, params, None)
        out['im%d_rows' % k] = np.asarray(f_iter.index)
        out['im%d_origin' % k] = np.array(origin)
        out['im%d_shape' % k] = np.array([s.stop - s.start for s in slices])
        out['im%d_P' % k] = np.array(len(sub))
        out['im%d_mask_counts' % k] = masks.sum(1)
        out['im%d_pix_sum' % k] = np.array(sub.sum())
        out['im%d_vect' % k] = vect
        out['im%d_bounds' % k] = bnds
        out['im%d_F' % k] = np.array(residual(vect))
        if jacobian is not None:     # (the disc profile has none: fitfunc.py:451-452)
            out['im%d_grad' % k] = jacobian(vect)
        k += 1
    out['im_count'] = np.array(k)
    return out


def save_case(name, f0, frames, call, do_intermediates=True, tight=True):
    frames = np.asarray(frames)
    out = {'frames': frames, 'call': np.array(json.dumps(call))}
    _df_to_arrays(f0, 'f0_', out)
    res_a = run_reference(f0, frames, call, {})
    out['ref_aborts'] = np.array(int(res_a is None))
    if res_a is not None:
        _df_to_arrays(res_a, 'refA_', out)
        if tight:
            res_b = run_reference(f0, frames, call, TIGHT)
            _df_to_arrays(res_b, 'refB_', out)
    if do_intermediates:
        out.update(intermediates(f0, frames, call))
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **out)
    n_nan = -1 if res_a is None else int(np.isnan(res_a['cost']).sum())
    print('%-28s %7.1f KB  features=%d  nan_cost=%d' % (
        name, os.path.getsize(path) / 1024., len(f0), n_nan))


def table(p0, size, signal, background, ndim, isotropic, frame=None):
    cols = ['z', 'y', 'x'][-ndim:]
    f0 = pd.DataFrame(p0, columns=cols)
    f0['signal'] = float(signal)
    if isotropic:
        f0['size'] = float(size if not hasattr(size, '__iter__') else size[0])
    else:
        for c, s in zip(['size_z', 'size_y', 'size_x'][-ndim:], size):
            f0[c] = float(s)
    if background is not None:
        f0['background'] = float(background)
    if frame is not None:
        f0['frame'] = frame
    return f0


def main():
    # --- cfg 1: one 64x64 frame, one cluster of three -------------
    truth = np.array([[30., 28.], [33.5, 34.2], [27.1, 35.3]])
    im = np.zeros((64, 64), np.uint8)
    for p in truth:
        artificial.draw_gaussian(im, p, 3., 100)
    rng = np.random.RandomState(1)
    p0 = truth + rng.uniform(-0.5, 0.5, truth.shape)
    save_case('cfg1_triple', table(p0, 3., 90., 0., 2, True), im[None],
              dict(diameter=13))

    # --- cfg 2 frames (seed = frame index), one noisy + one noise-free -------
    for seed, noise, tag in ((0, 10, 'noisy'), (1, 0, 'clean')):
        im, truth, p0 = artificial.random_frame((512, 512), 200, 3., 100, noise, seed,
                                                margin=13)
        save_case('cfg2_frame_%s' % tag, table(p0, 3., 90., noise / 2., 2, True),
                  im[None], dict(diameter=13))

    # --- two-frame video through a FramesSequence-like reader ---------------
    frames, tabs = [], []
    for t in range(2):
        im, truth, p0 = artificial.random_frame((128, 160), 30, 3., 100, 10, 100 + t,
                                                margin=13)
        frames.append(im)
        tabs.append(table(p0, 3., 90., 5., 2, True, frame=t))
    f0 = pd.concat(tabs[::-1], ignore_index=True)  # unsorted frames on purpose
    save_case('video_2frames', f0, np.stack(frames), dict(diameter=13))

    # --- sigma=3 variant (size 4.243, diameter 17) ---------------------------
    im, truth, p0 = artificial.random_frame((256, 256), 60, 3 * np.sqrt(2), 100, 10, 7,
                                            margin=17)
    save_case('sigma3_d17', table(p0, 3 * np.sqrt(2), 90., 5., 2, True), im[None],
              dict(diameter=17))

    # --- 2D anisotropic, sizes free -----------------------------------------
    im, truth, p0 = artificial.random_frame((160, 160), 24, (5., 3.), 160, 16, 11,
                                            margin=(20, 12))
    save_case('aniso2d_sizevar', table(p0, (5.2, 2.9), 150., 8., 2, False), im[None],
              dict(diameter=(20, 12), param_mode=dict(size='var')))
    save_case('aniso2d_default', table(p0, (5., 3.), 150., 8., 2, False), im[None],
              dict(diameter=(20, 12)))

    # --- 2D isotropic, size free, signal per cluster ------------------------
    im, truth, p0 = artificial.random_frame((200, 200), 40, 4., 160, 16, 12, margin=16)
    save_case('iso2d_sizevar', table(p0, 4.3, 150., 8., 2, True), im[None],
              dict(diameter=16, param_mode=dict(size='var')))
    save_case('iso2d_signal_cluster', table(p0, 4., 150., 8., 2, True), im[None],
              dict(diameter=16, param_mode=dict(signal='cluster', size='cluster')))
    save_case('iso2d_signal_const', table(p0, 4., 158., 8., 2, True), im[None],
              dict(diameter=16, param_mode=dict(signal='const', background='const')))

    # --- custom bounds -------------------------------------------------------
    save_case('iso2d_bounds', table(p0, 4.3, 150., 8., 2, True), im[None],
              dict(diameter=16, param_mode=dict(size='var'),
                   bounds=dict(signal=(20, 2000), size=(.9, 9), pos_diff=2.,
                               signal_rel_diff=0.5, background=(1., 50.))))

    # --- 3D anisotropic (cfg 3 shape, smaller) -------------------------------
    im, truth, p0 = artificial.random_frame((32, 64, 64), 14, (2., 4., 4.), 100, 10, 3,
                                            margin=(9, 17, 17))
    save_case('aniso3d_bigcluster', table(p0, (2., 4., 4.), 90., 5., 3, False), im[None],
              dict(diameter=(9, 17, 17)))
    im, truth, p0 = artificial.random_frame((36, 84, 84), 9, (2., 4., 4.), 100, 10, 13,
                                            margin=(9, 17, 17))
    save_case('aniso3d_default', table(p0, (2., 4., 4.), 90., 5., 3, False), im[None],
              dict(diameter=(9, 17, 17)))
    save_case('aniso3d_sizevar', table(p0, (2.1, 3.9, 4.1), 90., 5., 3, False), im[None],
              dict(diameter=(9, 17, 17), param_mode=dict(size='var')))

    # --- 3D isotropic ---------------------------------------------------------
    im, truth, p0 = artificial.random_frame((48, 64, 72), 7, 3., 120, 10, 4, margin=13)
    save_case('iso3d_default', table(p0, 3., 100., 5., 3, True), im[None],
              dict(diameter=13))

    # --- constrained dimers / trimers (2D) and tetramers (3D) ---------------
    rng = np.random.RandomState(21)
    size = 4.
    im = np.zeros((160, 200), np.uint8)
    truth = []
    for gy in range(3):
        for gx in range(4):
            c = np.array([30. + gy * 48, 28. + gx * 46]) + rng.uniform(-.5, .5, 2)
            a = rng.uniform(0, 2 * np.pi)
            d = np.array([np.sin(a), np.cos(a)]) * size
            truth += [c - d, c + d]
    truth = np.array(truth)
    for p in truth:
        artificial.draw_gaussian(im, p, size, 160)
    p0 = truth + rng.uniform(-0.7, 0.7, truth.shape)
    f0 = table(p0, size, 150., 0., 2, True)
    save_case('dimer_free', f0, im[None], dict(diameter=16, separation=32))
    save_case('dimer_constrained', f0, im[None],
              dict(diameter=16, separation=32,
                   constraints=dict(kind='dimer', dist=[2 * size] * 2, ndim=2)))
    imn = artificial.add_poisson_noise(im, 16, rng)
    f0n = f0.copy()
    f0n['background'] = 8.
    save_case('dimer_constrained_noisy', f0n, imn[None],
              dict(diameter=16, separation=32,
                   constraints=dict(kind='dimer', dist=[2 * size] * 2, ndim=2)))

    im = np.zeros((150, 150), np.uint8)
    truth = []
    tri = np.array([[0, 1], [-0.5 * np.sqrt(3), -0.5], [0.5 * np.sqrt(3), -0.5]]) \
        * 2 / 3 * np.sqrt(3)
    for gy in range(2):
        for gx in range(2):
            c = np.array([40. + gy * 70, 40. + gx * 70]) + rng.uniform(-.5, .5, 2)
            a = rng.uniform(0, 2 * np.pi)
            rot = np.array([[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]])
            truth.append(c + tri.dot(rot.T) * size)
    truth = np.concatenate(truth)
    for p in truth:
        artificial.draw_gaussian(im, p, size, 120)
    p0 = truth + rng.uniform(-0.7, 0.7, truth.shape)
    save_case('trimer_constrained', table(p0, size, 110., 0., 2, True), im[None],
              dict(diameter=16, separation=32,
                   constraints=dict(kind='trimer', dist=[2 * size] * 2, ndim=2)))

    tet = np.array([[0, 0, 0.5 * np.sqrt(6)],
                    [0, -(2 / 3.) * np.sqrt(3), -(1 / 6.) * np.sqrt(6)],
                    [1, (1 / 3.) * np.sqrt(3), -(1 / 6.) * np.sqrt(6)],
                    [-1, (1 / 3.) * np.sqrt(3), -(1 / 6.) * np.sqrt(6)]])
    size3 = 3.
    im = np.zeros((48, 48, 96), np.uint8)
    truth = []
    for gx in range(2):
        c = np.array([24., 24., 24. + 48 * gx]) + rng.uniform(-.5, .5, 3)
        truth.append(c + tet * size3)
    truth = np.concatenate(truth)
    for p in truth:
        artificial.draw_gaussian(im, p, size3, 60)
    p0 = truth + rng.uniform(-0.5, 0.5, truth.shape)
    save_case('tetramer3d_constrained', table(p0, size3, 55., 0., 3, True), im[None],
              dict(diameter=13, separation=26,
                   constraints=dict(kind='tetramer', dist=[2 * size3] * 3, ndim=3)))

    # --- overlapping features, large p0 error (re-window rounds) ------------
    #     reference tests/test_refine.py:884-922: 256x256, diameter 21, sep 24
    rng = np.random.RandomState(5)
    size = 21 / 4.
    pos = np.stack([rng.uniform(21, 256 - 21, 100), rng.uniform(21, 256 - 21, 100)], 1)
    keep = []
    for i, p in enumerate(pos):
        if all(np.sum((p - pos[j]) ** 2) > 15 ** 2 for j in keep):
            keep.append(i)
    pos = pos[keep]
    im = np.zeros((256, 256), np.uint8)
    for p in pos:
        artificial.draw_gaussian(im, p, size, 200)
    p0 = pos + rng.uniform(0, 4, pos.shape)
    save_case('overlap_d21_bigshift', table(p0, size, 200., None, 2, True), im[None],
              dict(diameter=21, separation=24))

    # --- cfg 5: dense clusters of 8-16 + dimers ------------------------------
    im, truth, p0 = artificial.cluster_frame((192, 192), 6, [2, 8, 12, 16], 3., 2.0, 60,
                                             10, 9)
    save_case('cfg5_dense', table(p0, 3., 55., 5., 2, True), im[None],
              dict(diameter=13,
                   constraints=dict(kind='dimer', dist=[6., 6.], ndim=2)),
              tight=True)

    # --- edges / failures -----------------------------------------------------
    im, truth, p0 = artificial.random_frame((64, 64), 6, 3., 100, 10, 31, margin=13)
    p0e = np.concatenate([p0, [[2.3, 30.2], [61.5, 62.1], [30.4, -3.2]]])
    for p in [[2., 30.], [61., 62.], [30., 0.]]:
        artificial.draw_gaussian(im, p, 3., 100)
    save_case('edges', table(p0e, 3., 90., 5., 2, True), im[None], dict(diameter=13))
    # fully out-of-bounds feature and a NaN row: the reference aborts (IndexError)
    p0f = np.concatenate([p0, [[-20., 30.]]])
    save_case('oob_feature', table(p0f, 3., 90., 5., 2, True), im[None],
              dict(diameter=13), do_intermediates=False)
    f0 = table(p0, 3., 90., 5., 2, True)
    f0.loc[2, 'signal'] = np.nan
    save_case('nan_feature', f0, im[None], dict(diameter=13), do_intermediates=False)
    # rms threshold: message-carrying RefineException -> NaN cost, no abort
    save_case('rms_threshold', table(p0, 3., 90., 5., 2, True), im[None],
              dict(diameter=13, max_rms_dev=0.028))
    # uint16 and float frames
    im16 = im.astype(np.uint16) * 40
    save_case('dtype_u16', table(p0, 3., 3600., 200., 2, True), im16[None],
              dict(diameter=13))
    imf = (im / 255.).astype(np.float32)
    save_case('dtype_f32', table(p0, 3., 0.35, 0.02, 2, True), imf[None],
              dict(diameter=13))


if __name__ == '__main__':
    main()
