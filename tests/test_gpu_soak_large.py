"""The large-cluster path on random configurations (tests/tools/soak_large.py: 2D / 3D, parameter
modes with shared and free sizes / signals, narrow and wide masks, uint16 and float frames with
NaN pixels, a lowpass): engine = oracle.  400 seeds run in 45 s on the GPU box (one differs by
2e-6 px: a 105-iteration fit with free sizes on 51-pixel masks); a slice of them here."""
import importlib.util
import os

import pytest

import _cases  # noqa: F401  (paths)

pytestmark = pytest.mark.gpu

_spec = importlib.util.spec_from_file_location(
    'soak_large', os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tools', 'soak_large.py'))
soak_large = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(soak_large)


@pytest.mark.parametrize("seed", list(range(100, 148)))
def test_random_large_cluster_engine_vs_oracle(engine, oracle, seed):
    r = soak_large.compare(engine, seed)
    if r is None:
        pytest.skip("no cluster beyond 64 features from this seed")
    text, differs, d = r
    assert not differs, text
