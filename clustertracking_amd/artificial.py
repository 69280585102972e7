"""Synthetic frames for tests and benchmarks (host side, NumPy).

Restates the drawing rule of the reference's ``artificial.draw_feature``
(reference ``clustertracking/artificial.py:81-141``) and the Poisson-noise rule
of ``SimulatedImage.noisy_image`` (``artificial.py:368-378``) for the Gaussian
feature only.  The reference module itself does not run on NumPy 2
(``artificial.py:139,141``), so this is a restatement, not an import:

* per axis the patch is ``[max(floor(c - 4*size), 0), min(ceil(c + 4*size + 1), lim))``
* ``r^2 = sum(((idx - c) / size)^2)``, ``spot = max_value * exp(-r^2 * ndim / 2)``
* the spot is *truncated* to the image dtype and *added with integer
  wrap-around* (``image[rect] += spot.astype(image.dtype)``)

The benchmark workloads of SURVEY.md 8(d) (cfg 1-5) are generated here from
seeds so that the GPU box regenerates identical inputs without any data files.
"""
import numpy as np


def _as_tuple(value, ndim):
    if not hasattr(value, '__iter__'):
        return (value,) * ndim
    value = tuple(value)
    if len(value) != ndim:
        raise ValueError("expected a scalar or %d values" % ndim)
    return value


def draw_gaussian(image, position, size, max_value):
    """Add one Gaussian feature to ``image`` in place (artificial.py:81-141)."""
    ndim = image.ndim
    size = _as_tuple(size, ndim)
    sl = []
    r2 = 0.
    for ax, (c, s, lim) in enumerate(zip(position, size, image.shape)):
        if c >= lim or c < 0:
            raise ValueError("Position outside of image.")
        lo = max(int(np.floor(c - 4. * s)), 0)
        hi = min(int(np.ceil(c + 4. * s + 1)), lim)
        sl.append(slice(lo, hi))
        t = (np.arange(lo, hi, dtype=np.float64) - c) / s
        shape = [1] * ndim
        shape[ax] = -1
        r2 = r2 + (t * t).reshape(shape)
    spot = max_value * np.exp(r2 * (ndim / -2.))
    with np.errstate(over='ignore'):
        image[tuple(sl)] += spot.astype(image.dtype)
    return image


def feat_ring(r, ndim, thickness):
    """Ring with a Gaussian cross-section (reference artificial.py:17-19)."""
    return np.exp(((r - 1 + thickness) / thickness) ** 2 * ndim / -2)


def feat_disc(r, ndim, disc_size):
    """Solid disc with Gaussian-smoothed border (reference artificial.py:22-28, ``feat_hat``)."""
    result = np.ones_like(r)
    mask = r > disc_size
    result[mask] = np.exp(((r[mask] - disc_size) / (1 - disc_size)) ** 2 * ndim / -2)
    return result


def feat_inv_series(r, ndim, p):
    """The profile the ``'inv_series_<N>'`` fit function evaluates (reference fitfunc.py:148-154):
    ``p[0] / polyval([1, p[1], ..., p[N]], r**2)``.  (The reference's artificial.py has no drawing
    function for it; this one serves the fixtures of that fit function.)"""
    c = np.array(p, dtype=np.float64)
    c[0] = 1.
    return p[0] / np.polyval(c, r ** 2)


def draw_feature(image, position, size, max_value, feat_func='gauss', **kwargs):
    """Add one radially symmetric feature in place: the reference's ``draw_feature``
    (artificial.py:81-141; patch of 8 x size per axis, ``r = sqrt(sum(((idx - c)/size)^2))``, the
    spot truncated to the image dtype and added with integer wrap-around) for ``feat_func`` =
    'gauss', 'ring' (``thickness=``), 'disc' (``disc_size=``), 'inv_series' (``p=``) or a callable
    ``feat_func(r, ndim=..., **kwargs)`` as there."""
    if feat_func == 'gauss':
        return draw_gaussian(image, position, size, max_value)
    func = feat_func if callable(feat_func) else \
        dict(ring=feat_ring, disc=feat_disc, inv_series=feat_inv_series)[feat_func]
    ndim = image.ndim
    size = _as_tuple(size, ndim)
    sl = []
    r2 = 0.
    for ax, (c, s, lim) in enumerate(zip(position, size, image.shape)):
        if c >= lim or c < 0:
            raise ValueError("Position outside of image.")
        lo = max(int(np.floor(c - 4. * s)), 0)
        hi = min(int(np.ceil(c + 4. * s + 1)), lim)
        sl.append(slice(lo, hi))
        t = (np.arange(lo, hi, dtype=np.float64) - c) / s
        shape = [1] * ndim
        shape[ax] = -1
        r2 = r2 + (t * t).reshape(shape)
    spot = max_value * func(np.sqrt(r2), ndim=ndim, **kwargs)
    with np.errstate(over='ignore'):
        image[tuple(sl)] += spot.astype(image.dtype)
    return image


def add_poisson_noise(image, level, rng, saturation=None):
    """Poisson noise then clip to the dtype range (artificial.py:368-378)."""
    if level <= 0:
        return image
    if saturation is None:
        saturation = np.iinfo(image.dtype).max
    noise = rng.poisson(level, image.shape)
    return np.clip(image.astype(np.int64) + noise, 0, saturation).astype(image.dtype)


def random_frame(shape, n_features, size, signal=100, noise=10, seed=0,
                 margin=None, dtype=np.uint8, p0_jitter=0.5):
    """One synthetic frame of SURVEY.md 8(d): ``n_features`` Gaussians at
    uniform random positions inside ``margin``, Poisson noise ``noise``.

    Returns (image, truth[n, ndim], p0[n, ndim]); p0 = truth + U(-jitter, jitter).
    """
    ndim = len(shape)
    size = _as_tuple(size, ndim)
    if margin is None:
        margin = tuple(int(np.ceil(4 * s)) + 1 for s in size)
    margin = _as_tuple(margin, ndim)
    rng = np.random.RandomState(seed)
    truth = np.stack([rng.uniform(m, s - 1 - m, n_features)
                      for s, m in zip(shape, margin)], axis=1)
    image = np.zeros(shape, dtype=dtype)
    for pos in truth:
        draw_gaussian(image, pos, size, signal)
    image = add_poisson_noise(image, noise, rng)
    p0 = truth + rng.uniform(-p0_jitter, p0_jitter, truth.shape)
    return image, truth, p0


def cluster_frame(shape, n_clusters, cluster_sizes, size, spacing=2.0,
                  signal=60, noise=10, seed=0, dtype=np.uint8, p0_jitter=0.5):
    """Frame seeded with compact clusters (cfg 5): each cluster is a random
    walk-free lattice blob of ``k`` Gaussians at centre distance
    ``spacing*size`` (k drawn from ``cluster_sizes``), clusters placed on a
    jittered grid so that they do not merge with each other."""
    ndim = len(shape)
    assert ndim == 2
    rng = np.random.RandomState(seed)
    size = float(size)
    step = spacing * size
    n_side = int(np.ceil(np.sqrt(n_clusters)))
    cell = min(shape) / float(n_side)
    truth = []
    for ci in range(n_clusters):
        gy, gx = divmod(ci, n_side)
        centre = np.array([(gy + 0.5) * cell, (gx + 0.5) * cell])
        centre += rng.uniform(-0.5, 0.5, 2)
        k = int(cluster_sizes[rng.randint(len(cluster_sizes))])
        # grow a compact blob on a triangular lattice, nearest-first
        n_ring = int(np.ceil(np.sqrt(k))) + 1
        ii, jj = np.meshgrid(np.arange(-n_ring, n_ring + 1),
                             np.arange(-n_ring, n_ring + 1), indexing='ij')
        lat = np.stack([ii.ravel() * step * np.sqrt(3) / 2,
                        (jj.ravel() + 0.5 * (ii.ravel() % 2)) * step], axis=1)
        order = np.argsort((lat ** 2).sum(1), kind='stable')[:k]
        ang = rng.uniform(0, 2 * np.pi)
        rot = np.array([[np.cos(ang), -np.sin(ang)], [np.sin(ang), np.cos(ang)]])
        truth.append(centre + lat[order].dot(rot.T))
    truth = np.concatenate(truth)
    keep = np.all((truth >= 4 * size + 1) &
                  (truth < np.array(shape) - 4 * size - 2), axis=1)
    truth = truth[keep]
    image = np.zeros(shape, dtype=dtype)
    for pos in truth:
        draw_gaussian(image, pos, size, signal)
    image = add_poisson_noise(image, noise, rng)
    p0 = truth + rng.uniform(-p0_jitter, p0_jitter, truth.shape)
    return image, truth, p0
