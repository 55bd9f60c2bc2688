"""stats::optim(method = "L-BFGS-B") as the reference calls it for the rows the IRLS leaves
(R/fitNbinomGLMs.R:340-407), through scipy -- TEST helper: the cross-check of the engine's own optimiser
(optim_rows_kernel, damped Fisher scoring on the same objective and box).  Not part of the product package."""
import numpy as np
from scipy import special as sps


def _dnbinom_mu_log(k, size, mu):
    """dnbinom(k, mu = mu, size = size, log = TRUE) for the host-side optim fallback, in a form that
    stays finite for the extreme mu an L-BFGS-B line search visits (beta up to +-30 on the log2 scale)"""
    k, mu = np.asarray(k, np.float64), np.asarray(mu, np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        tail = np.where(k > 0, k * (np.log(mu) - np.log(size + mu)), 0.0)
    return sps.gammaln(k + size) - sps.gammaln(size) - sps.gammaln(k + 1.0) - size * np.log1p(mu / size) + tail


def fitNbinomGLMsOptim_scipy(yh, nfh, x, lam, alpha, wh, useWeights, start):
    """the reference's own route for ONE row -- L-BFGS-B with optim's numerical gradient and stopping parameters
    (R/fitNbinomGLMs.R:359-371) -- kept as the cross-check of the engine's optimiser (tests), not used by the chain"""
    from scipy.optimize import minimize
    from scipy.stats import norm
    large = 30.0

    def objectiveFn(pv):                                                           # :359-370
        mu_row = nfh * 2.0 ** (x @ pv)
        with np.errstate(all="ignore"):
            ll = _dnbinom_mu_log(yh, 1.0 / alpha, mu_row)
            logLike_ = np.sum(wh * ll) if useWeights else np.sum(ll)
            logPrior = np.sum(norm.logpdf(pv, 0.0, np.sqrt(1.0 / lam)))
        v = -1.0 * (logLike_ + logPrior)
        return v if np.isfinite(v) else 1e300

    def gradFn(pv):                      # stats::optim's numerical gradient: central, ndeps = 1e-3, clipped
        g = np.empty_like(pv)
        for i in range(pv.size):
            hi, lo = pv.copy(), pv.copy()
            hi[i], lo[i] = min(pv[i] + 1e-3, large), max(pv[i] - 1e-3, -large)
            g[i] = (objectiveFn(hi) - objectiveFn(lo)) / (hi[i] - lo[i])
        return g
    o = minimize(objectiveFn, np.asarray(start, float), jac=gradFn, method="L-BFGS-B", bounds=[(-large, large)] * len(start),
                 options=dict(maxcor=5, ftol=1e7 * np.finfo(float).eps, gtol=0.0, maxiter=100))   # :371
    return o.x, bool(o.success), objectiveFn
