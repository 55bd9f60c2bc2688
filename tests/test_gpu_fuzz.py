"""A slice of tests/gpu_fuzz.py in the GPU suite: random designs (factor / continuous / mixed, p up to 24), weights,
ridge, QR / LU, prior, Cox-Reid on / off -- the HIP library against the CPU checker, every output bit for bit."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", range(4000, 4048))
def test_fuzzed_configuration(oracle, seed):
    from tests import gpu_fuzz
    gpu_fuzz.one(seed)


@pytest.fixture(scope="module")
def engine():
    from deseq2_amd.engine import DeviceEngine
    return DeviceEngine("cuda:0")


@pytest.mark.parametrize("seed", range(9000, 9032))
def test_fuzzed_analysis_fused_equals_call_by_call(engine, seed):
    """a slice of tests/gpu_fuzz_chain.py: the fused device-driven DESeq() against the call-by-call chain"""
    from tests import gpu_fuzz_chain
    gpu_fuzz_chain.one(engine, seed)
