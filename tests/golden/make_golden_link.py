"""Golden vectors for the host-side linker from the REFERENCE's ``Linker`` class
(clustertracking/find_link.py:579-733), run through oracle/refshim.py in the
build container:  python tests/golden/make_golden_link.py

Each case stores the per-frame coordinate arrays (flattened with offsets), the
search range, the memory and the particle ids the reference assigned."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import refshim  # noqa: E402

refshim.load()
from clustertracking.find_link import Linker  # noqa: E402


def reference_ids(levels, search_range, memory):
    linker = Linker(search_range, memory)
    out = []
    for t, coords in enumerate(levels):
        if t == 0:
            linker.init_level(coords, t)
        else:
            linker.next_level(coords, t)
        out.append(np.array(linker.particle_ids, dtype=np.int64))
    return out


def random_walkers(rng, n, n_frames, ndim, box, step, p_drop, p_birth):
    pos = rng.uniform(0, box, (n, ndim))
    levels = []
    for t in range(n_frames):
        pos = pos + rng.normal(0, step, pos.shape)
        seen = rng.rand(len(pos)) >= p_drop          # missed detections (memory cases)
        lvl = pos[seen]
        n_new = rng.poisson(p_birth)
        if n_new:
            born = rng.uniform(0, box, (n_new, ndim))
            pos = np.concatenate([pos, born])
            lvl = np.concatenate([lvl, born])
        levels.append(lvl[rng.permutation(len(lvl))])
    return levels


def main():
    rng = np.random.RandomState(123)
    cases = {
        'sparse2d': (random_walkers(rng, 30, 12, 2, 200., 1.0, 0., 0.5), (5., 5.), 0),
        'dense2d': (random_walkers(rng, 150, 10, 2, 100., 1.5, 0., 1.0), (5., 5.), 0),
        'aniso3d': (random_walkers(rng, 60, 8, 3, 60., 1.0, 0., 0.5), (3., 6., 6.), 0),
        'memory1': (random_walkers(rng, 40, 12, 2, 300., 1.0, 0.1, 0.5), (5., 5.), 1),
        'memory3': (random_walkers(rng, 40, 14, 2, 300., 1.0, 0.15, 0.5), (5., 5.), 3),
        'memory2_3d': (random_walkers(rng, 40, 10, 3, 80., 0.8, 0.1, 0.5), (4., 4., 4.), 2),
    }
    out = {}
    for name, (levels, sr, memory) in cases.items():
        ids = reference_ids(levels, sr, memory)
        counts = np.array([len(l) for l in levels])
        out[name + '_pos'] = np.concatenate(levels)
        out[name + '_counts'] = counts
        out[name + '_sr'] = np.array(sr)
        out[name + '_memory'] = np.array(memory)
        out[name + '_ids'] = np.concatenate(ids)
        print(name, 'frames', len(levels), 'points', counts.sum(), 'tracks', out[name + '_ids'].max() + 1)
    np.savez_compressed(os.path.join(HERE, 'link_cases.npz'), **out)


if __name__ == '__main__':
    main()
