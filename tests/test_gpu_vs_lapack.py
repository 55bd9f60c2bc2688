"""-m gpu: the HIP path against the independently written LAPACK restatement (oracle/lapack_oracle.py) -- the same
comparison and budgets the C oracle is held to in tests/test_oracle_vs_lapack.py, through the host-pointer C ABI.  The
bit-for-bit comparisons with the C oracle are tests/test_gpu_parity.py and its neighbours; this file shows the HIP results
next to an implementation that shares no arithmetic with them (LAPACK, cephes, numpy's summation order)."""
import numpy as np
import pytest

from deseq2_amd import native
from oracle import lapack_oracle
from tests.test_oracle_vs_lapack import SHAPE_NAMES, _case, compare, golden_cases, run_all, shape_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", sorted(golden_cases()))
def test_hip_vs_lapack_small(name):
    d = golden_cases()[name]
    compare(run_all(native, d), run_all(lapack_oracle, d), d, name)


@pytest.mark.parametrize("n,m,design,kw", [
    (1500, 50, "batch_condition", {}),
    (400, 120, ("factor", 8), {}),
    (600, 30, "two_group", {"weights": True}),
    (300, 64, "batch_condition", {"weights": True, "zero_w": True, "useQR": False}),
])
def test_hip_vs_lapack(n, m, design, kw):
    d = _case(n, m, design, seed=n + m, **kw)
    compare(run_all(native, d), run_all(lapack_oracle, d), d, "%dx%d" % (n, m))


def test_hip_vs_lapack_beyond_the_resident_waves():
    """9 000 genes x 100 samples, ~ batch + condition: three times the 3 072 wave slots the persistent fit kernels keep
    resident, so every wave fits several genes in a row: fitBeta$iter equal on every gene, fitDisp
    iterations equal outside <= 1 % ulp-level ties, values within 1e-7 / 1e-8 (about a minute of numpy)."""
    d = _case(9000, 100, "batch_condition", seed=9100)
    st = compare(run_all(native, d), run_all(lapack_oracle, d), d, "9000x100")
    assert st["fitBeta"]["iter_mismatch"] == 0 and st["fitDispGrid"]["same"] >= 0.99


@pytest.mark.parametrize("name", SHAPE_NAMES)
def test_hip_vs_lapack_at_baseline_shapes(name):
    """BASELINE.json configs C2..C5 at full shape (C3: 1000 x 500 p=4; C4: 200 x 2000, 10-level factor, QR; C5:
    1000 x 200, weights with zeros + the betaPrior pass on the expanded p=3 design; C2: 1000 x 100): iteration counts
    equal (ties <= 1 %), well-conditioned share >= 0.95, grid agreement >= 0.95, values within 1e-7 / 1e-8."""
    d = shape_case(name)
    compare(run_all(native, d), run_all(lapack_oracle, d), d, name, min_well=0.95)


@pytest.mark.parametrize("seed", (5, 6, 7))
def test_hip_floor_regime_vs_lapack(seed):
    """the dispersion-floor regime (Poisson / NB mixture, > 25 % of the genes start at alpha_0 = 1e-8): everything R's
    callers see -- fitBeta$iter, clamped dispGeneEst, dispGeneEstConv / refitDisp, MAP dispConv and dispMAP
    (tests/floor_regime.py)"""
    from tests.floor_regime import assert_visible_parity, floor_case, visible_chain
    d = floor_case(seed)
    assert_visible_parity(visible_chain(native, d), visible_chain(lapack_oracle, d), "hip seed %d" % seed)
