"""SECOND-OPINION ORACLE -- test infrastructure only (tests/ may import it; the product never does).

A numpy restatement of the three native routines of /root/reference/src/DESeq2.cpp, written matrix expression
by matrix expression the way the reference writes them (Armadillo), with REAL LAPACK underneath
(numpy.linalg: qr -> dgeqrf/dorgqr, solve -> dgesv, inv -> dgetrf/dgetri, det -> dgetrf -- the routines
Armadillo itself dispatches to for qr_econ / solve / inv / det) and scipy.special's gammaln / digamma /
polygamma for R's lgamma / digamma / trigamma.  Rf_dnbinom_mu is the C oracle's restatement of R nmath's
algorithm (oracle/orc_nmath.c; held < 1 ulp-level against mpmath by tests/test_oracle_math.py) -- scipy's
nbinom.logpmf cancels badly at small dispersions and would not do.

Why it exists: the reference itself cannot be built in this image.  src/DESeq2.cpp:16,23-25 needs
RcppArmadillo.h, R.h, Rmath.h and R_ext/Utils.h -- external libraries (R, Rcpp, RcppArmadillo, Armadillo)
that are absent -- and a build against stand-ins for them is not a reference build.  So the C oracle
(oracle/deseq2_oracle.c) is pinned by the reference's OWN tests' known answers and properties
(tests/test_oracle_properties.py), and this module gives it an independently written
counterpart with different linear algebra (LAPACK instead of the oracle's own LU / Householder), different
special functions (cephes instead of the nmath restatement) and a different summation order (numpy pairwise
instead of wave order): tests/test_oracle_vs_lapack.py holds the two to equal iteration counts (outside
ulp-level Armijo ties) and values within 1e-7 / 1e-8 -- evidence that the control flow was read the same way
twice and that the results are not artefacts of one arithmetic.  It is NOT the reference and no file calls it so.

Plain per-gene Python loops: use at sizes of hundreds of genes.  Signatures, argument order and returned
keys follow R/RcppExports.R:4,8,12 like oracle/oracle.py.
"""
import numpy as np
from scipy import special

from oracle import oracle as _C


def _cr_matrices(x, mu, alpha, wts, useWeights, weightThreshold, orders):
    """src/DESeq2.cpp:36-45, 73-84, 118-132: w_diag and its derivatives, the row / column subsetting under
    observation weights, b = x' diag(w) x (and db, d2b)"""
    base = 1.0 / mu + alpha
    diags = [base ** -1.0, -1.0 * base ** -2.0, 2.0 * base ** -3.0][:orders]
    if useWeights:
        keep = wts > weightThreshold
        x = x[keep]
        x = x[:, np.abs(x).sum(axis=0) > 0.0]
        diags = [d[keep] for d in diags]
    return [x.T @ (x * d[:, None]) for d in diags]


def _det(b):
    return np.linalg.det(b) if b.shape[0] else 1.0          # arma::det of a 0 x 0 matrix is 1


def log_posterior(log_alpha, y, mu, x, prior_mean, prior_sigmasq, usePrior, wts, useWeights, weightThreshold, useCR):
    """src/DESeq2.cpp:31-64"""
    alpha = np.exp(log_alpha)
    cr_term = 0.0
    if useCR:
        (b,) = _cr_matrices(x, mu, alpha, wts, useWeights, weightThreshold, 1)
        with np.errstate(all="ignore"):
            cr_term = -0.5 * np.log(_det(b))
    an1 = 1.0 / alpha
    terms = special.gammaln(y + an1) - special.gammaln(an1) - y * np.log(mu + an1) - an1 * np.log(1.0 + mu * alpha)
    ll_part = np.sum(wts * terms) if useWeights else np.sum(terms)
    prior_part = -0.5 * (log_alpha - prior_mean) ** 2 / prior_sigmasq if usePrior else 0.0
    return ll_part + prior_part + cr_term


def _dll_terms(y, mu, alpha):
    an1 = 1.0 / alpha
    return (special.digamma(an1) + np.log(1 + mu * alpha) - mu * alpha / (1.0 + mu * alpha)
            - special.digamma(y + an1) + y / (mu + an1))


def dlog_posterior(log_alpha, y, mu, x, prior_mean, prior_sigmasq, usePrior, wts, useWeights, weightThreshold, useCR):
    """src/DESeq2.cpp:68-107"""
    alpha = np.exp(log_alpha)
    cr_term = 0.0
    if useCR:
        b, db = _cr_matrices(x, mu, alpha, wts, useWeights, weightThreshold, 2)
        if b.shape[0]:
            detb = _det(b)
            ddetb = detb * np.trace(np.linalg.inv(b) @ db)
            cr_term = -0.5 * ddetb / detb
    t = _dll_terms(y, mu, alpha)
    ll_part = alpha ** -2.0 * (np.sum(wts * t) if useWeights else np.sum(t))
    prior_part = -1.0 * (log_alpha - prior_mean) / prior_sigmasq if usePrior else 0.0
    return (ll_part + cr_term) * alpha + prior_part


def d2log_posterior(log_alpha, y, mu, x, prior_mean, prior_sigmasq, usePrior, wts, useWeights, weightThreshold, useCR):
    """src/DESeq2.cpp:111-158"""
    alpha = np.exp(log_alpha)
    cr_term = 0.0
    if useCR:
        b, db, d2b = _cr_matrices(x, mu, alpha, wts, useWeights, weightThreshold, 3)
        if b.shape[0]:
            b_i = np.linalg.inv(b)
            detb = _det(b)
            ddetb = detb * np.trace(b_i @ db)
            d2detb = detb * (np.trace(b_i @ db) ** 2 - np.trace(b_i @ db @ b_i @ db) + np.trace(b_i @ d2b))
            cr_term = 0.5 * (ddetb / detb) ** 2 - 0.5 * d2detb / detb
    an1 = 1.0 / alpha
    an2 = alpha ** -2.0
    t1 = _dll_terms(y, mu, alpha)
    t2 = (-1 * an2 * special.polygamma(1, an1) + mu ** 2 * alpha * (1 + mu * alpha) ** -2.0
          + an2 * special.polygamma(1, y + an1) + an2 * y * (mu + an1) ** -2.0)
    if useWeights:
        ll_part = -2 * alpha ** -3.0 * np.sum(wts * t1) + an2 * np.sum(wts * t2)
    else:
        ll_part = -2 * alpha ** -3.0 * np.sum(t1) + an2 * np.sum(t2)
    prior_part = -1.0 / prior_sigmasq if usePrior else 0.0
    inner = dlog_posterior(log_alpha, y, mu, x, prior_mean, prior_sigmasq, False, wts, useWeights, weightThreshold,
                           useCR)                                   # :156 -- usePrior = false, the un-subset x
    return ((ll_part + cr_term) * alpha ** 2 + inner) + prior_part


def fitDisp(ySEXP, xSEXP, mu_hatSEXP, log_alphaSEXP, log_alpha_prior_meanSEXP, log_alpha_prior_sigmasqSEXP,
            min_log_alphaSEXP, kappa_0SEXP, tolSEXP, maxitSEXP, usePriorSEXP, weightsSEXP, useWeightsSEXP,
            weightThresholdSEXP, useCRSEXP):
    """src/DESeq2.cpp:164-277"""
    y = np.asarray(ySEXP, float); x = np.asarray(xSEXP, float); mu_hat = np.asarray(mu_hatSEXP, float)
    weights = np.asarray(weightsSEXP, float)
    n = y.shape[0]
    log_alpha = np.array(np.broadcast_to(np.asarray(log_alphaSEXP, float), (n,)))
    pmean = np.broadcast_to(np.asarray(log_alpha_prior_meanSEXP, float), (n,))
    sig, min_la, kappa_0 = float(log_alpha_prior_sigmasqSEXP), float(min_log_alphaSEXP), float(kappa_0SEXP)
    tol, maxit = float(tolSEXP), int(maxitSEXP)
    usePrior, useWeights, useCR = bool(usePriorSEXP), bool(useWeightsSEXP), bool(useCRSEXP)
    thr = float(weightThresholdSEXP)
    epsilon = 1.0e-4
    out = {k: np.zeros(n) for k in ("last_change", "initial_lp", "initial_dlp", "last_lp", "last_dlp", "last_d2lp")}
    it = np.zeros(n, np.int32); ita = np.zeros(n, np.int32)
    with np.errstate(all="ignore"):
        for i in range(n):
            rest = (y[i], mu_hat[i], x, pmean[i], sig, usePrior, weights[i], useWeights, thr, useCR)
            a = log_alpha[i]
            lp = log_posterior(a, *rest)
            dlp = dlog_posterior(a, *rest)
            kappa = kappa_0
            out["initial_lp"][i] = lp; out["initial_dlp"][i] = dlp
            change = -1.0
            for _t in range(maxit):
                it[i] += 1
                a_propose = a + kappa * dlp
                if a_propose < -30.0:
                    kappa = (-30.0 - a) / dlp
                if a_propose > 10.0:
                    kappa = (10.0 - a) / dlp
                theta_kappa = -1.0 * log_posterior(a + kappa * dlp, *rest)
                theta_hat_kappa = -1.0 * lp - kappa * epsilon * dlp ** 2
                if theta_kappa <= theta_hat_kappa:
                    ita[i] += 1
                    a = a + kappa * dlp
                    lpnew = log_posterior(a, *rest)
                    change = lpnew - lp
                    if change < tol:
                        lp = lpnew
                        break
                    if a < min_la:
                        break
                    lp = lpnew
                    dlp = dlog_posterior(a, *rest)
                    kappa = min(kappa * 1.1, kappa_0)
                    if ita[i] % 5 == 0:
                        kappa = kappa / 2.0
                else:
                    kappa = kappa / 2.0
            out["last_lp"][i] = lp; out["last_dlp"][i] = dlp
            out["last_d2lp"][i] = d2log_posterior(a, *rest)
            log_alpha[i] = a
            out["last_change"][i] = change
    out.update(log_alpha=log_alpha, iter=it, iter_accept=ita)
    return out


def fitBeta(ySEXP, xSEXP, nfSEXP, alpha_hatSEXP, contrastSEXP, beta_matSEXP, lambdaSEXP, weightsSEXP, useWeightsSEXP,
            tolSEXP, maxitSEXP, useQRSEXP, minmuSEXP):
    """src/DESeq2.cpp:283-465"""
    y = np.asarray(ySEXP, float); x = np.asarray(xSEXP, float); nf = np.asarray(nfSEXP, float)
    weights = np.asarray(weightsSEXP, float)
    n, m = y.shape; p = x.shape[1]
    alpha_hat = np.broadcast_to(np.asarray(alpha_hatSEXP, float), (n,))
    beta_mat = np.array(beta_matSEXP, float).reshape(n, p)
    lam = np.asarray(lambdaSEXP, float); contrast = np.asarray(contrastSEXP, float)
    tol, maxit, minmu = float(tolSEXP), int(maxitSEXP), float(minmuSEXP)
    useWeights, useQR = bool(useWeightsSEXP), bool(useQRSEXP)
    large = 30.0
    ridge = np.diag(lam)
    beta_var = np.zeros((n, p)); cnum = np.zeros((n, 1)); cden = np.zeros((n, 1)); hat = np.zeros((n, m))
    it = np.zeros(n); deviance = np.zeros(n)

    def wvec(i, mu):
        return (weights[i] * mu if useWeights else mu) / (1.0 + alpha_hat[i] * mu)

    with np.errstate(all="ignore"):
        for i in range(n):
            nfrow, yrow = nf[i], y[i]
            beta_hat = beta_mat[i].copy()
            mu_hat = np.fmax(nfrow * np.exp(x @ beta_hat), minmu)
            dev = dev_old = 0.0
            for t in range(maxit):
                it[i] += 1
                w_vec = wvec(i, mu_hat)
                z = np.log(mu_hat / nfrow) + (yrow - mu_hat) / mu_hat
                try:
                    if useQR:
                        wxr = np.vstack([x * np.sqrt(w_vec)[:, None], np.sqrt(ridge)])
                        q, r = np.linalg.qr(wxr)
                        big_z = np.concatenate([z * np.sqrt(w_vec), np.zeros(p)])
                        beta_hat = np.linalg.solve(r, q.T @ big_z)
                    else:
                        beta_hat = np.linalg.solve(x.T @ (x * w_vec[:, None]) + ridge, x.T @ (z * w_vec))
                except np.linalg.LinAlgError:
                    beta_hat = np.full(p, np.nan)
                if np.sum(np.abs(beta_hat) > large) > 0:
                    it[i] = maxit
                    break
                mu_hat = np.fmax(nfrow * np.exp(x @ beta_hat), minmu)
                ld = _C.dnbinom_mu_log(yrow, 1.0 / alpha_hat[i], mu_hat)
                dev = 0.0
                for j in range(m):                       # the reference's sequential accumulation, :366-373
                    dev = dev + -2.0 * (weights[i, j] * ld[j] if useWeights else ld[j])
                conv_test = abs(dev - dev_old) / (abs(dev) + 0.1)
                if np.isnan(conv_test):
                    it[i] = maxit
                    break
                if t > 0 and conv_test < tol:
                    break
                dev_old = dev
            deviance[i] = dev
            beta_mat[i] = beta_hat
            w_vec = wvec(i, mu_hat)
            xw = x * np.sqrt(w_vec)[:, None]
            xtwx = x.T @ (x * w_vec[:, None])
            try:
                inv = np.linalg.inv(xtwx + ridge)
            except np.linalg.LinAlgError:
                inv = np.full((p, p), np.nan)
            hat[i] = np.einsum("ja,jb,ba->j", xw, xw, inv)
            sigma = inv @ xtwx @ inv
            cnum[i, 0] = contrast @ beta_hat
            cden[i, 0] = np.sqrt(contrast @ sigma @ contrast)
            beta_var[i] = np.diag(sigma)
    return dict(beta_mat=beta_mat, beta_var_mat=beta_var, iter=it, hat_diagonals=hat, contrast_num=cnum,
                contrast_denom=cden, deviance=deviance)


def fitDispGrid(ySEXP, xSEXP, mu_hatSEXP, disp_gridSEXP, log_alpha_prior_meanSEXP, log_alpha_prior_sigmasqSEXP,
                usePriorSEXP, weightsSEXP, useWeightsSEXP, weightThresholdSEXP, useCRSEXP):
    """src/DESeq2.cpp:469-513"""
    y = np.asarray(ySEXP, float); x = np.asarray(xSEXP, float); mu_hat = np.asarray(mu_hatSEXP, float)
    weights = np.asarray(weightsSEXP, float)
    grid = np.asarray(disp_gridSEXP, float)
    n = y.shape[0]
    pmean = np.broadcast_to(np.asarray(log_alpha_prior_meanSEXP, float), (n,))
    delta = grid[1] - grid[0]
    out = np.zeros(n)
    with np.errstate(all="ignore"):
        for i in range(n):
            rest = (y[i], mu_hat[i], x, pmean[i], float(log_alpha_prior_sigmasqSEXP), bool(usePriorSEXP), weights[i],
                    bool(useWeightsSEXP), float(weightThresholdSEXP), bool(useCRSEXP))
            lpv = np.array([log_posterior(a, *rest) for a in grid])
            a_hat = grid[_first_max(lpv)]
            fine = np.linspace(a_hat - delta, a_hat + delta, grid.size)
            lpv = np.array([log_posterior(a, *rest) for a in fine])
            out[i] = fine[_first_max(lpv)]
    return {"log_alpha": out}


def _first_max(v):
    """arma::vec::max(idx): first index of the maximum; NaN entries never win (comparisons are false)"""
    best, idx = -np.inf, 0
    for k, val in enumerate(v):
        if val > best:
            best, idx = val, k
    return idx
