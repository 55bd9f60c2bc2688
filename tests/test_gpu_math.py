"""-m gpu: the device math library (csrc/dsq_math.hpp) must reproduce the oracle's scalar
primitives (oracle/orc_nmath.c) BIT FOR BIT -- that is what makes iteration counts of the
kernels identical to the oracle's by construction."""
import numpy as np
import pytest

from tests.helpers import assert_same

pytestmark = pytest.mark.gpu


def _inputs(rng):
    pos = np.concatenate([np.exp(rng.uniform(-30, 30, 40000)), rng.uniform(0, 20, 40000),
                          np.arange(0, 40, 0.5), [1e-300, 5e-324, 1e300, np.inf, 0.0, np.nan]])
    anyx = np.concatenate([rng.uniform(-750, 720, 40000), rng.uniform(-2, 2, 40000), -pos[:2000],
                           [np.inf, -np.inf, np.nan, 0.0, -0.0, 709.8, -745.2]])
    return pos, anyx


@pytest.mark.parametrize("name,op", [("exp", 0), ("log", 1), ("log1p", 2), ("lgamma", 3), ("digamma", 4),
                                     ("trigamma", 5), ("stirlerr", 6)])
def test_unary_bit_exact(oracle, name, op):
    from deseq2_amd import native
    rng = np.random.default_rng(op + 11)
    pos, anyx = _inputs(rng)
    if name == "exp":
        x = anyx
    elif name == "log":
        x = np.concatenate([pos, [-1.0]])
    elif name == "log1p":
        x = np.concatenate([rng.uniform(-1, 3, 50000), np.exp(rng.uniform(-60, 60, 20000)),
                            -np.exp(rng.uniform(-60, 0, 20000)), [-1.0, -2.0, np.inf, np.nan, 0.0]])
    elif name == "stirlerr":
        x = pos[(pos > 0) & np.isfinite(pos)]
    else:
        x = pos
    got = native.test_math(op, x)
    want = oracle.unary(name, x)
    assert_same(got, want, name)


def test_bd0_and_dnbinom_bit_exact(oracle):
    from deseq2_amd import native
    rng = np.random.default_rng(5)
    n = 200000
    x = rng.integers(0, 5000, n).astype(float)
    x[rng.uniform(size=n) < 0.2] = 0.0
    size = np.exp(rng.uniform(np.log(1e-3), np.log(1e9), n))
    mu = np.exp(rng.uniform(np.log(1e-6), np.log(1e6), n))
    mu[:2000] = x[:2000] * (1 + rng.normal(0, 1e-3, 2000)) + 1e-9   # x ~ np branch of bd0
    got = native.test_math(8, x, size, mu)
    want = oracle.dnbinom_mu_log(x, size, mu)
    assert_same(got, want, "dnbinom_mu_log")
    xx = np.exp(rng.uniform(-5, 12, n)); npp = xx * np.exp(rng.normal(0, 0.2, n))
    got = native.test_math(7, xx, npp)
    L = oracle.lib()
    want = np.array([L.orc_bd0(a, b) for a, b in zip(xx[:20000], npp[:20000])])
    assert_same(got[:20000], want, "bd0")
    # edge cases: infinite size (Poisson limit), zero mu, huge mu
    xe = np.array([0, 3, 7, 0, 5, 2, 4], float)
    se = np.array([np.inf, np.inf, 1e300, 0.0, 0.5, 10, 10], float)
    me = np.array([2.5, 2.5, 3.0, 1.0, 0.0, np.inf, 1e-300], float)
    assert_same(native.test_math(8, xe, se, me), oracle.dnbinom_mu_log(xe, se, me), "dnbinom edge")
