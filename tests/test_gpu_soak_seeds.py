"""The seeds on which the soak (tests/tools/soak_random.py) found engine != oracle, kept as a test.

Round 3, constrained fits (6 000 seeds from 300 000 = 6 355 clusters): 7 clusters differ, all with
equal status.  They are ill-conditioned problems, not a difference between the two programs: the
ORACLE ITSELF lands in another minimum when its input positions change by 1e-13 (relative) -- 2D
tetramers at the kink of their sort-based constraint (constraints.py:102-114), dimers / trimers on
one real feature.  Asserted per cluster: the engine agrees with the oracle to 1e-6 px, or the
engine's result is one the oracle reaches from an input 1e-13 away, or the oracle is shown to be
unstable there (moves by more than 1e-4 px under such a perturbation)."""
import numpy as np
import pytest

import _cases
import clustertracking_amd as cta
from clustertracking_amd import _abi, _lib

pytestmark = pytest.mark.gpu

SEEDS = [300279, 300733, 303063, 304636, 305143, 305359, 305365,
         # the final build's soaks (gpurun_out/r3_soak_*.log: 12 150 + 5 925 + 4 104 + 5 898 clusters; every
         # differing cluster is a constrained fit of cost 0.02-0.22 that wanders for 50-300 iterations)
         401359, 402708, 411174, 411337, 430763, 430892, 432582, 433968, 434721, 435356, 435783,
         # after the inv_series / large-kernel work (gpurun_out/r3_soak_final_*.log: 16 037 + 12 169 clusters)
         20059, 22661, 23138, 23638, 23713, 31972, 32840, 32883]
EPS = (1e-13, -1e-13, 3e-13, -3e-13, 1e-12, -1e-12)


def _oracle(f0, im, diameter, kw, eps=0.):
    import ctr_oracle
    f = f0.copy()
    for c in ('z', 'y', 'x'):
        if c in f:
            f[c] = f[c] * (1. + eps)
    prep = cta.prepare_batch(f, im, diameter, **kw)
    ctr_oracle.run_batch(prep.problem, prep.batch, 4)
    return prep.batch


@pytest.mark.parametrize("seed", SEEDS)
def test_soak_seed(seed):
    f0, im, diameter, kw = _cases.random_case(seed)
    nd = im.ndim
    prep = cta.prepare_batch(f0.copy(), im, diameter, **kw)
    b = prep.batch
    _lib.default_engine(0).refine_batch(prep.problem, b)
    ref = _oracle(f0, im, diameter, kw)
    assert (b.status == ref.status).all()
    off = b.feat_offset
    d = np.abs(b.params_out[:, 2:2 + nd] - ref.params_out[:, 2:2 + nd])
    per = np.array([d[off[c]:off[c + 1]].max() if b.status[c] == 0 else 0. for c in range(b.n_clusters)])
    differing = np.flatnonzero(per > 1e-6)
    if len(differing) == 0:
        return
    alts = [_oracle(f0, im, diameter, kw, e) for e in EPS]
    for c in differing:
        sl = slice(off[c], off[c + 1])
        reached = any(a.status[c] == 0 and np.abs(a.params_out[sl, 2:2 + nd] - b.params_out[sl, 2:2 + nd]).max() < 1e-5
                      for a in alts)
        unstable = any(a.status[c] != ref.status[c] or
                       np.abs(a.params_out[sl, 2:2 + nd] - ref.params_out[sl, 2:2 + nd]).max() > 1e-4 for a in alts)
        assert reached or unstable, (seed, int(c), float(per[c]), float(b.cost[c]), float(ref.cost[c]))
