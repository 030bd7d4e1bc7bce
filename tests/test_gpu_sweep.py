"""GPU: a slice of tools/prove_sweep.py in the suite - random layer sizes, table packings, FRI
parameters, cap heights, proof-of-work bits, folding schedules (fitting and blind), proof layouts and
the LogUp packing switch, both fields, D = 4 and D = 5 circuits with their table variants.  Per draw: preprocessed commitment and proof bytes equal the
oracle's and both verifiers accept - or both sides refuse the configuration."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _sweep():
    spec = importlib.util.spec_from_file_location("prove_sweep", os.path.join(ROOT, "tools", "prove_sweep.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("first", [21000, 21008, 21016])
def test_random_draws_agree_with_the_oracle(oracle, first):
    sweep = _sweep()
    proved = 0
    for seed in range(first, first + 8):
        desc, n = sweep.one(oracle, seed, 9)
        proved += n > 0
    assert proved >= 4, "most draws must be provable configurations"
