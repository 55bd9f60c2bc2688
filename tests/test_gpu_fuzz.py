"""A slice of tests/gpu_fuzz.py in the GPU suite: random designs (factor / continuous / mixed, p up to 24), weights,
ridge, QR / LU, prior, Cox-Reid on / off -- the HIP library against the CPU checker, every output bit for bit."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", range(4000, 4048))
def test_fuzzed_configuration(oracle, seed):
    from tests import gpu_fuzz
    gpu_fuzz.one(seed)
