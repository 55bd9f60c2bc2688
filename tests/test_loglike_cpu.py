"""nbinomLogLike (R/core.R:2208-2217) on the closed split of the density -- the CPU checker's
restatement pinned against what it replaces and against independent arithmetic:
  * R's own formulation: rowSums([w *] dnbinom(y, mu = mu, size = 1/alpha, log = TRUE)) with nmath's dnbinom_mu, sample by
    sample (orc_dnbinom_mu_log, the restatement of the nmath source the engine used until round 4);
  * scipy's negative-binomial log pmf;
  * mpmath at 40 digits on a few rows: away from the dispersion floor the split loses four to five digits to cancellation
    that bd0 does not (~ 1e-11 absolute on a row); AT the floor (alpha = 1e-8, size = 1e8) it is the other way round --
    dnbinom_mu's `log1p(-x/n)` (dbinom_raw's lf term, x = size, n = size + y) cancels nine digits, 6e-10 per sample, the
    same amount in every model of a gene (it depends on the count and the size only, so it drops out of an LRT statistic)
    -- and the split is exact to 1e-13 there: the two formulations differ by ~ 6e-8 on such a row, R's own error;
and on the corners: zero counts at vanishing means, counts outside the split (below 1e-10 / alpha: the full density),
dispersions at the 1e-8 floor and at 10, observation weights with zeros, means below the 0.5 clamp of the fits."""
import numpy as np
import pytest

from oracle import oracle as O


def _faithful(counts, mu, disp, w=None):
    n, m = counts.shape
    size = np.repeat(1.0 / disp, m)
    d = O.dnbinom_mu_log(counts.astype(np.float64).ravel(), size, mu.ravel()).reshape(n, m)
    return (d if w is None else w * d).sum(axis=1)


def _case(seed, n=60, m=96, alpha=None, mean_scale=3.0):
    rng = np.random.default_rng(seed)
    mu = np.exp(rng.normal(mean_scale, 2.0, (n, 1)) + rng.normal(0, 0.5, (n, m)))
    disp = np.exp(rng.uniform(np.log(1e-3), np.log(2.0), n)) if alpha is None else np.full(n, alpha)
    size = 1.0 / disp
    counts = rng.negative_binomial(np.broadcast_to(size[:, None], mu.shape), size[:, None] / (size[:, None] + mu)).astype(np.int32)
    return counts, mu, disp


@pytest.mark.parametrize("seed,alpha", [(1, None), (2, 1e-8), (3, 10.0), (4, 0.05)])
def test_split_equals_the_sample_by_sample_density(seed, alpha):
    counts, mu, disp = _case(seed, alpha=alpha)
    got = O.nbinomLogLike(counts, mu, disp, np.ones(counts.shape), False)
    want = _faithful(counts, mu, disp)
    if alpha == 1e-8:
        # the dispersion floor: dnbinom_mu itself is off by ~ 6e-10 per sample (module docstring); the split is the exact one
        import mpmath as mp
        mp.mp.dps = 40
        np.testing.assert_allclose(got, want, rtol=2e-9)
        for g in np.argsort(-np.abs(got - want))[:3]:
            s = mp.mpf(1) / mp.mpf(float(disp[g]))
            t = mp.mpf(0)
            for y, mm in zip(counts[g].tolist(), mu[g].tolist()):
                mm = mp.mpf(mm)
                t += mp.loggamma(y + s) - mp.loggamma(s) - mp.loggamma(y + 1) + s * mp.log(s / (s + mm)) + y * mp.log(mm / (s + mm))
            assert abs(mp.mpf(float(got[g])) - t) < mp.mpf("1e-11") and abs(mp.mpf(float(want[g])) - t) > mp.mpf("1e-9")
        return
    # both carry rounding of ~ m eps |terms|; the split's terms are larger than the density by y log(.) ~ 1e3 .. 1e6
    scale = np.maximum(np.abs(want), (counts * np.log1p(counts)).sum(axis=1) + 1.0)
    assert np.all(np.abs(got - want) <= 2e-13 * scale), np.max(np.abs(got - want) / scale)
    np.testing.assert_allclose(got, want, rtol=5e-11)


def test_against_scipy_and_mpmath():
    from scipy.stats import nbinom
    import mpmath as mp
    counts, mu, disp = _case(7, n=24, m=64)
    got = O.nbinomLogLike(counts, mu, disp, np.ones(counts.shape), False)
    size = 1.0 / disp[:, None]
    np.testing.assert_allclose(got, nbinom.logpmf(counts, size, size / (size + mu)).sum(axis=1), rtol=1e-11)
    mp.mp.dps = 40
    for g in range(6):
        s = mp.mpf(1) / mp.mpf(float(disp[g]))
        t = mp.mpf(0)
        for y, mm in zip(counts[g].tolist(), mu[g].tolist()):
            mm = mp.mpf(mm)
            t += mp.loggamma(y + s) - mp.loggamma(s) - mp.loggamma(y + 1) + s * mp.log(s / (s + mm)) + y * mp.log(mm / (s + mm))
        assert abs(mp.mpf(float(got[g])) - t) < mp.mpf("5e-11") * (1 + abs(t) / 1000), (g, float(got[g]), t)


def test_weights_zero_counts_and_samples_outside_the_split():
    rng = np.random.default_rng(11)
    counts, mu, disp = _case(12, n=40, m=80)
    w = rng.uniform(0.0, 1.0, counts.shape)
    w[rng.uniform(size=counts.shape) < 0.1] = 0.0
    got = O.nbinomLogLike(counts, mu, disp, w, True)
    np.testing.assert_allclose(got, _faithful(counts, mu, disp, w), rtol=5e-11)
    # all-zero rows at vanishing means (the fits' unclamped mu of a gene with a dead group), means below the 0.5 clamp
    z = np.zeros((5, 40), np.int32)
    muz = np.full((5, 40), 1e-300); muz[1] = 0.0; muz[2] = 1e-3; muz[3] = 0.49; muz[4] = 5e-324
    gz = O.nbinomLogLike(z, muz, np.full(5, 0.3), np.ones(z.shape), False)
    np.testing.assert_allclose(gz, _faithful(z, muz, np.full(5, 0.3)), rtol=1e-12, atol=1e-300)
    assert gz[1] == 0.0
    # counts outside the split: y < 1e-10 size (dispersion at 1e-12: size 1e12, counts of 1 .. 50) -- the full density
    y = rng.integers(1, 50, (6, 30)).astype(np.int32)
    m2 = np.exp(rng.normal(2.5, 0.5, y.shape))
    d2 = np.full(6, 1e-12)
    np.testing.assert_allclose(O.nbinomLogLike(y, m2, d2, np.ones(y.shape), False), _faithful(y, m2, d2), rtol=1e-12)
    # a positive count at mean zero is impossible: -inf, as dnbinom_mu says
    y3 = np.array([[3, 0, 1]], np.int32); m3 = np.array([[0.0, 0.0, 2.0]])
    assert O.nbinomLogLike(y3, m3, np.array([0.2]), np.ones((1, 3)), False)[0] == -np.inf


def test_the_reduced_models_constants_cancel_in_the_lrt_statistic():
    """full and reduced model share counts, dispersions and weights: their K' is the same number, so the statistic
    2 (logLik_full - logLik_reduced) is the difference of the two sweeps alone -- exact zero for equal means"""
    counts, mu, disp = _case(21, n=30, m=64)
    a = O.nbinomLogLike(counts, mu, disp, np.ones(counts.shape), False)
    b = O.nbinomLogLike(counts, mu.copy(), disp, np.ones(counts.shape), False)
    assert np.array_equal(a, b)
    red = np.repeat(mu.mean(axis=1, keepdims=True), mu.shape[1], axis=1)
    stat = 2 * (a - O.nbinomLogLike(counts, red, disp, np.ones(counts.shape), False))
    want = 2 * (_faithful(counts, mu, disp) - _faithful(counts, red, disp))
    np.testing.assert_allclose(stat, want, rtol=1e-9, atol=1e-9)
