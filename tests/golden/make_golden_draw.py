"""Generate tests/golden/draw_cases.npz: features drawn by the REFERENCE's ``artificial.draw_feature``
(artificial.py:81-141, run through oracle/refshim.py:reference_draw_feature) -- what pins
``clustertracking_amd.artificial.draw_feature`` / ``draw_gaussian`` (the generator of every synthetic
frame of the tests and benchmarks, restated because the reference's module does not run on NumPy 2).

    python tests/golden/make_golden_draw.py        (build container only: needs /root/reference)
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import refshim  # noqa: E402

CASES = [
    # (name, shape, dtype, [(position, size, max_value, feat_func, kwargs), ...])
    ('gauss2d_u8', (40, 48), 'uint8', [((20.3, 24.7), 3., 100, 'gauss', {}), ((22.9, 27.1), 3., 100, 'gauss', {}),
                                      ((1.2, 46.9), 3., 200, 'gauss', {})]),           # overlap, edge, corner
    ('gauss2d_wrap', (24, 24), 'uint8', [((12.4, 11.6), 2.5, 200, 'gauss', {})] * 3),   # integer wrap-around
    ('gauss2d_aniso_u16', (50, 36), 'uint16', [((25.5, 18.2), (5., 3.), 3000, 'gauss', {})]),
    ('gauss3d_u8', (20, 30, 30), 'uint8', [((10.2, 15.7, 14.1), (2., 4., 4.), 120, 'gauss', {}),
                                           ((0.4, 3.3, 28.8), (2., 4., 4.), 120, 'gauss', {})]),
    ('gauss2d_f64', (30, 30), 'float64', [((15.1, 14.9), 3., 1., 'gauss', {})]),
    ('ring2d_u8', (48, 48), 'uint8', [((24.3, 23.6), 4., 160, 'ring', dict(thickness=0.2)),
                                      ((2.1, 40.4), 4., 160, 'ring', dict(thickness=0.2))]),
    ('ring3d_a_u8', (28, 44, 44), 'uint8', [((14.2, 22.7, 21.4), (3., 5., 5.), 160, 'ring', dict(thickness=0.3))]),
    ('disc2d_u8', (48, 48), 'uint8', [((24.3, 23.6), 4., 160, 'disc', dict(disc_size=0.5)),
                                      ((45.8, 5.2), 4., 160, 'disc', dict(disc_size=0.5))]),
    ('disc3d_u16', (30, 30, 30), 'uint16', [((15.4, 14.8, 15.1), 4., 1600, 'disc', dict(disc_size=0.5))]),
]


def main():
    out = {'cases': np.array(json.dumps([(n, s, d, [(list(p), sz if not hasattr(sz, '__iter__') else list(sz), mv, ff, kw)
                                                     for p, sz, mv, ff, kw in feats])
                                         for n, s, d, feats in CASES]))}
    for name, shape, dtype, feats in CASES:
        im = np.zeros(shape, dtype=dtype)
        for pos, size, mv, ff, kw in feats:
            im = refshim.reference_draw_feature(im, pos, size, mv, ff, **kw)
        out[name] = im
        print('%-20s sum %d  max %s' % (name, int(im.astype(np.float64).sum()), im.max()))
    np.savez_compressed(os.path.join(HERE, 'draw_cases.npz'), **out)


if __name__ == '__main__':
    main()
