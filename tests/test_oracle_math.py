"""CPU: the oracle's scalar primitives (oracle/orc_nmath.c) against the committed mpmath
golden vectors (tests/golden/nmath_golden.json, made by tests/golden/make_golden.py)."""
import json
import os

import numpy as np
import pytest

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "nmath_golden.json")))
EPS = 2.220446049250313e-16


def _ulps(got, want):
    got, want = np.asarray(got), np.asarray(want)
    ok = want != 0
    return np.max(np.abs(got[ok] - want[ok]) / np.spacing(np.abs(want[ok])))


@pytest.mark.parametrize("name,max_ulp", [("exp", 1.5), ("log", 1.5), ("log1p", 1.5), ("trigamma", 8.0)])
def test_relative_accuracy(oracle, name, max_ulp):
    g = GOLD[name]
    got = oracle.unary(name, np.array(g["x"]))
    # golden values are mpmath results rounded to double: allow 0.5 ulp for that rounding
    assert _ulps(got, g["y"]) <= max_ulp + 0.5


@pytest.mark.parametrize("name,max_eps", [("lgamma", 50.0), ("digamma", 16.0), ("stirlerr", 80.0)])
def test_absolute_accuracy(oracle, name, max_eps):
    """absolute error in units of eps*max(1,|value|): what enters the likelihood sums.
    (R's own lgammafn / pre-4.4 stirlerr lose the same digits near their zeros.)"""
    g = GOLD[name]
    got = oracle.unary(name, np.array(g["x"]))
    want = np.array(g["y"])
    err = np.abs(got - want) / np.maximum(1.0, np.abs(want)) / EPS
    assert err.max() <= max_eps


def test_bd0(oracle):
    g = GOLD["bd0"]
    L = oracle.lib()
    got = np.array([L.orc_bd0(a, b) for a, b in zip(g["x"], g["np"])])
    want = np.array(g["y"])
    # bd0 is a cancellation-free evaluation of a quantity that cancels: compare absolutely
    assert np.max(np.abs(got - want) / np.maximum(np.abs(want), 1e-300)) < 1e-13


def test_dnbinom_mu_log(oracle):
    g = GOLD["dnbinom_mu_log"]
    got = oracle.dnbinom_mu_log(np.array(g["x"]), np.array(g["size"]), np.array(g["mu"]))
    want = np.array(g["y"])
    rel = np.abs(got - want) / np.maximum(1.0, np.abs(want))
    assert rel.max() < 2e-13
    # and against scipy's independent implementation
    from scipy.stats import nbinom
    x, r, mu = map(np.array, (g["x"], g["size"], g["mu"]))
    sp = nbinom.logpmf(x, r, r / (r + mu))
    # (scipy's gammaln-difference form loses digits for size >> x; mpmath above is the tight check)
    assert np.max(np.abs(got - sp) / np.maximum(1.0, np.abs(sp))) < 1e-6


def test_special_values(oracle):
    inf, nan = np.inf, np.nan
    assert oracle.unary("exp", [inf, -inf, 710.0, -746.0, 0.0]).tolist() == [inf, 0.0, inf, 0.0, 1.0]
    r = oracle.unary("log", [0.0, inf, 1.0, -1.0])
    assert r[0] == -inf and r[1] == inf and r[2] == 0.0 and np.isnan(r[3])
    r = oracle.unary("log1p", [-1.0, 0.0, 1e-20, -2.0])
    assert r[0] == -inf and r[1] == 0.0 and r[2] == 1e-20 and np.isnan(r[3])
    assert oracle.unary("lgamma", [1.0, 2.0])[0] == pytest.approx(0.0, abs=1e-14)
    assert np.isnan(oracle.unary("exp", [nan])[0])
    # size = Inf is the Poisson limit (nmath/dnbinom.c)
    from scipy.stats import poisson
    got = oracle.dnbinom_mu_log([0.0, 3.0, 40.0], [inf, inf, inf], [2.5, 2.5, 30.0])
    np.testing.assert_allclose(got, poisson.logpmf([0, 3, 40], [2.5, 2.5, 30.0]), rtol=1e-12)
