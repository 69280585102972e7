"""The reference's accuracy matrix (clustertracking/tests/test_refine.py:598-765) restated for
the HIP engine with fixed seeds: Gaussian, disc and ring features (test_refine.py:768-881), 2D / 3D, isotropic / anisotropic, Poisson
noise 0 / 16 / 48 (perfect, S/N 10, S/N 3), parameter modes const / var signal / var size / var,
and dimers / trimers / tetramers with and without constraints.  Same image construction
(test_refine.py:82-122,124-186: 20 features on a grid of twice the diameter with random sub-pixel
offsets, signal 160 +- 20 %, size +- 20 %, uint8), same start guesses (truth + a random offset of
at most pos_diff x size, signal 160, nominal size, background noise / 2), same bounds and the
reference's own thresholds (test_refine.py:39-49):

    rms position error   < 0.01 px (noise 0)   < 0.05 px (noise 16)   < 0.1 px (noise 48)
    relative rms error of signal / size  < 1 % (noise 0),  < 50 % (noise 16)
    constrained dimers: bias along the bond < 0.001 x size, trimers / tetramers < 0.01 px

The reference draws its images with an unseeded RNG; here the seed is fixed per case.
Needs a real MI355X.
"""
import numpy as np
import pandas as pd
import pytest

import clustertracking_amd as cta
from clustertracking_amd import artificial

pytestmark = pytest.mark.gpu

SIGNAL = 160
NOISE = {'perfect': 0, 'imperfect': 16, 'noisy': 48}
PRECISION = {'perfect': 0.01, 'imperfect': 0.05, 'noisy': 0.1}       # px
SIGNAL_RTOL = {'perfect': 0.01, 'imperfect': 0.5}
SIZE_RTOL = {'perfect': 0.01, 'imperfect': 0.5}
REPEATS = 20
POS_DIFF = 0.5          # start offset in units of size
SIGNAL_DEV = SIZE_DEV = 0.2
BOUNDS = dict(signal=(20, 2000), size=(.9, 9))

GEOMETRIES = {
    'gauss2D': (2, (4., 4.)),
    'gauss2D_a': (2, (5., 3.)),
    'gauss3D': (3, (4., 4., 4.)),
    'gauss3D_a': (3, (3., 5., 5.)),
}
# the other profiles (test_refine.py:797-881: TestFit_disc* / TestFit_ring*): features drawn with the
# profile (artificial.py:17-28), fitted with it, its parameter given by param_val; rings are sharp:
# start offsets of a quarter of the size and sizes 5 % off (test_refine.py:839-843)
PROFILES = {
    'disc': dict(feat_kwargs=dict(disc_size=0.5), pos_diff=POS_DIFF, size_dev=SIZE_DEV),
    'ring': dict(feat_kwargs=dict(thickness=0.2), pos_diff=0.25, size_dev=0.05),
}
for _prof in PROFILES:
    for _g, _v in list(GEOMETRIES.items()):
        if _g.startswith('gauss'):
            GEOMETRIES[_g.replace('gauss', _prof)] = _v
MODES = {
    'const': (dict(signal='const', size='const'), 0., 0.),
    'var_signal': (dict(signal='var', size='const'), SIGNAL_DEV, 0.),
    'var_size': (dict(signal='const', size='var'), 0., SIZE_DEV),
    'var': (dict(signal='var', size='var'), SIGNAL_DEV, SIZE_DEV),
}


class Geometry(object):
    def __init__(self, name):
        self.ndim, self.size = GEOMETRIES[name]
        self.fit_function = name[:name.index('D') - 1]
        prof = PROFILES.get(self.fit_function, dict(feat_kwargs={}, pos_diff=POS_DIFF, size_dev=SIZE_DEV))
        self.feat_kwargs, self.pos_diff, self.size_dev = prof['feat_kwargs'], prof['pos_diff'], prof['size_dev']
        # what refine_leastsq gets besides the table: the profile and its parameter
        self.fit_kwargs = dict(fit_function=self.fit_function, param_val=dict(self.feat_kwargs)) \
            if self.feat_kwargs else {}
        self.diameter = tuple(int(s * 4) for s in self.size)
        self.separation = tuple(d * 2 for d in self.diameter)
        self.isotropic = len(set(self.diameter)) == 1
        self.pos_columns = ['z', 'y', 'x'][-self.ndim:]
        self.size_columns = ['size'] if self.isotropic else ['size_z', 'size_y', 'size_x'][-self.ndim:]

    def grid(self, rng, separation):
        n_side = int(REPEATS ** (1. / self.ndim) + 0.9999)
        pos = np.meshgrid(*[np.arange(0, s * n_side, s) for s in separation], indexing='ij')
        pos = np.array([p.ravel() for p in pos], dtype=float).T[:REPEATS] + self.separation
        pos += rng.random_sample(pos.shape) - 0.5
        return pos

    def draw(self, rng, pos, signal, size, noise):
        shape = tuple(np.max(pos, axis=0).astype(int) + np.array(self.separation))
        image = np.zeros(shape, dtype=np.uint8)
        for p, s, sz in zip(pos, signal, size):
            artificial.draw_feature(image, p, tuple(sz), s, self.fit_function, **self.feat_kwargs)
        if noise > 0:
            image = image + rng.poisson(noise, shape)
            if image.max() <= 255:
                image = image.astype(np.uint8)
        return image

    def signals_sizes(self, rng, n, signal_dev, size_dev):
        signal = SIGNAL * rng.uniform(1 - signal_dev, 1 + signal_dev, n) if signal_dev > 0 else np.repeat(float(SIGNAL), n)
        size = np.array([self.size]) * rng.uniform(1 - size_dev, 1 + size_dev, (n, 1)) if size_dev > 0 \
            else np.repeat([self.size], n, axis=0)
        return signal, size

    def p0(self, rng, expected_pos):
        n = expected_pos.shape[0]
        box = np.array([self.size]) * self.pos_diff
        dev = (rng.random_sample((10 * n, self.ndim)) - 0.5) * box * 2
        dev = dev[np.sum((dev / box) ** 2, axis=1) <= 1][:n]
        return expected_pos + dev

    def table(self, p0, noise):
        f0 = pd.DataFrame(p0, columns=self.pos_columns)
        f0['signal'] = float(SIGNAL)
        if self.isotropic:
            f0['size'] = float(self.size[0])
        else:
            for col, s in zip(self.size_columns, self.size):
                f0[col] = float(s)
        f0['background'] = noise / 2.
        return f0


@pytest.mark.parametrize("geometry", sorted(GEOMETRIES))
@pytest.mark.parametrize("level", ['perfect', 'imperfect', 'noisy'])
@pytest.mark.parametrize("mode", ['const', 'var_signal', 'var_size', 'var'])
def test_accuracy_matrix(engine, geometry, level, mode):
    """test_refine.py:598-688 (test_perfect_* / test_imperfect_* / test_noisy_*), for the gaussian,
    disc and ring profiles (test_refine.py:768-881)"""
    g = Geometry(geometry)
    param_mode, signal_dev, size_dev = MODES[mode]
    size_dev = g.size_dev if size_dev > 0 else 0.
    noise = NOISE[level]
    rng = np.random.RandomState(sum(map(ord, geometry + level + mode)))
    pos = g.grid(rng, g.separation)
    signal, size = g.signals_sizes(rng, REPEATS, signal_dev, size_dev)
    image = g.draw(rng, pos, signal, size, noise)
    f0 = g.table(g.p0(rng, pos), noise)
    res = cta.refine_leastsq(f0, image, g.diameter, param_mode=param_mode, bounds=BOUNDS,
                             pos_columns=g.pos_columns, **g.fit_kwargs)
    assert not np.any(np.isnan(res['cost']))
    dev = pos - res[g.pos_columns].values
    assert np.sqrt(np.mean(dev ** 2)) < PRECISION[level]
    if mode == 'const' and level == 'perfect':      # constant means constant
        assert np.abs(res['signal'].values / SIGNAL - 1).max() < 1e-7
        assert np.abs(res[g.size_columns].values / np.array(g.size if not g.isotropic else g.size[:1]) - 1).max() < 1e-7
    if level in SIGNAL_RTOL and param_mode['signal'] == 'var':
        assert np.sqrt(np.mean((1 - res['signal'].values / signal) ** 2)) < SIGNAL_RTOL[level]
    if level in SIZE_RTOL and param_mode['size'] == 'var':
        got = res[g.size_columns].values
        want = size[:, :1] if g.isotropic else size
        assert np.sqrt(np.mean((1 - got / want) ** 2)) < SIZE_RTOL[level]


# ---- clusters (test_refine.py:124-186,690-765) -----------------------------------------------

def _rot_2d(angle):
    return np.array([[np.cos(angle), -np.sin(angle)], [np.sin(angle), np.cos(angle)]])


def _rot_3d(angles):
    # Tait-Bryan rotation; any proper rotation serves (the reference: artificial.py:162-177)
    a, b, c = angles
    rx = np.array([[1, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]])
    ry = np.array([[np.cos(b), 0, np.sin(b)], [0, 1, 0], [-np.sin(b), 0, np.cos(b)]])
    rz = np.array([[np.cos(c), -np.sin(c), 0], [np.sin(c), np.cos(c), 0], [0, 0, 1]])
    return rx.dot(ry).dot(rz)


CLUSTER_GEOMETRY = {     # bond length 2, to be scaled by hard_radius x size (artificial.py:180-194)
    (2, 2): np.array([[0, -1], [0, 1]], float),
    (2, 3): np.array([[0, 1], [-0.5 * np.sqrt(3), -0.5], [0.5 * np.sqrt(3), -0.5]], float) * 2 / 3 * np.sqrt(3),
    (3, 2): np.array([[0, 0, -1], [0, 0, 1]], float),
    (3, 3): np.array([[0, 0, 2 / np.sqrt(3)], [-1, 0, -1 / np.sqrt(3)], [1, 0, -1 / np.sqrt(3)]], float),
    (3, 4): np.array([[0, 0, 0.5 * np.sqrt(6)], [0, -(2 / 3.) * np.sqrt(3), -(1 / 6.) * np.sqrt(6)],
                      [1, (1 / 3.) * np.sqrt(3), -(1 / 6.) * np.sqrt(6)],
                      [-1, (1 / 3.) * np.sqrt(3), -(1 / 6.) * np.sqrt(6)]], float),
}


def _cluster_case(g, cluster_size, noise, signal_dev, seed):
    """get_image_clusters: REPEATS clusters of `cluster_size` features at centre distance
    2 x size (hard_radius 1), random orientation"""
    rng = np.random.RandomState(seed)
    separation = [int(sep + 2 * s) for sep, s in zip(g.separation, g.size)]
    centres = g.grid(rng, separation)
    unit = CLUSTER_GEOMETRY[(g.ndim, cluster_size)]
    signal0, size0 = g.signals_sizes(rng, REPEATS, signal_dev, 0.)
    coords = []
    for c, sz in zip(centres, size0):
        rot = _rot_2d(rng.uniform(0, 2 * np.pi)) if g.ndim == 2 else _rot_3d(rng.uniform(0, 2 * np.pi, 3))
        coords.append(c + unit.dot(rot.T) * np.array(g.size))
    coords = np.concatenate(coords)
    signal = np.repeat(signal0, cluster_size)
    size = np.repeat(size0, cluster_size, axis=0)
    image = g.draw(rng, coords, signal, size, noise)
    f0 = g.table(g.p0(rng, coords), noise)
    return image, coords, signal, f0


@pytest.mark.parametrize("geometry", ['gauss2D', 'gauss3D'])
@pytest.mark.parametrize("level", ['perfect', 'imperfect', 'noisy'])
def test_dimer_unconstrained(engine, geometry, level):
    """test_dimer_perfect / _imperfect / _noisy (test_refine.py:690-727): rms error < precision"""
    g = Geometry(geometry)
    image, coords, signal, f0 = _cluster_case(g, 2, NOISE[level], SIGNAL_DEV, 11 + len(level))
    res = cta.refine_leastsq(f0, image, g.diameter, param_mode=dict(signal='var', size='const'),
                             bounds=BOUNDS, pos_columns=g.pos_columns)
    assert not np.any(np.isnan(res['cost']))
    assert np.all(res['cluster_size'] <= 2)
    assert np.sqrt(np.mean((coords - res[g.pos_columns].values) ** 2)) < PRECISION[level]


@pytest.mark.parametrize("geometry,cluster_size,kind", [
    ('gauss2D', 2, 'dimer'), ('gauss3D', 2, 'dimer'), ('gauss2D_a', 2, 'dimer'),
    ('gauss2D', 3, 'trimer'), ('gauss3D', 3, 'trimer'), ('gauss3D', 4, 'tetramer')])
def test_constrained_clusters(engine, geometry, cluster_size, kind):
    """test_dimer_constrained / test_trimer_constrained / test_tetramer_constrained
    (test_refine.py:729-765): noise-free, bond length 2 x size fixed by the constraint"""
    g = Geometry(geometry)
    if (g.ndim, cluster_size) not in CLUSTER_GEOMETRY:
        pytest.skip("geometry not defined")
    image, coords, signal, f0 = _cluster_case(g, cluster_size, 0, SIGNAL_DEV, 23 + cluster_size)
    cons = getattr(cta.constraints, kind)(2 * np.array(g.size), g.ndim)
    res = cta.refine_leastsq(f0, image, g.diameter, param_mode=dict(signal='var', size='const'),
                             bounds=BOUNDS, pos_columns=g.pos_columns, constraints=cons)
    assert not np.any(np.isnan(res['cost']))
    got = res[g.pos_columns].values
    assert np.sqrt(np.mean((coords - got) ** 2)) < PRECISION['perfect']
    # the constraint holds: every constrained pair at scaled distance 1 (constraints.py:59-137)
    for c in range(REPEATS):
        p = got[c * cluster_size:(c + 1) * cluster_size] / (2 * np.array(g.size))
        d = np.sqrt(((p[:, None, :] - p[None, :, :]) ** 2).sum(-1))[np.triu_indices(cluster_size, 1)]
        assert np.abs(d - 1).max() < 1e-9
    if kind == 'dimer':
        # bias along the bond < 0.001 x size (accuracy_constrained, test_refine.py:45,738)
        bias = []
        for c in range(REPEATS):
            a, b = coords[2 * c], coords[2 * c + 1]
            u = (b - a) / np.linalg.norm(b - a)
            bias += [np.dot(got[2 * c] - a, u), np.dot(got[2 * c + 1] - b, u)]
        assert abs(np.mean(bias)) < 0.001 * max(g.size)
