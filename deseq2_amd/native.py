"""Python mirror of the reference's R/RcppExports.R:4,8,12 -- the three `.Call` stubs
`fitDisp`, `fitBeta`, `fitDispGrid` -- bound to the MI355X engine instead of
src/DESeq2.cpp.  Same argument names, same order, same return-list member names and
types (`fitBeta$iter` is double, `fitDisp$iter` is integer; `contrast_num/denom` are
n x 1 matrices; src/DESeq2.cpp:268-276, 458-464, 512).

Two flavours:
  fitBeta / fitDisp / fitDispGrid             numpy in, numpy out (host pointers through
                                              dsq_fit_*; what the R shim does)
  fitBeta_dev / fitDisp_dev / fitDispGrid_dev torch CUDA tensors in/out through
                                              dsq_fit_*_dev on the current stream; matrices
                                              may be R-layout or gene-major.
"""
import ctypes as C

import numpy as np

from . import _lib as L


IGNORES_UNUSED_WEIGHTS = True      # weightsSEXP may be None when useWeightsSEXP is FALSE (HostEngine builds no all-ones matrix)


def _fcol(a, dtype=np.float64):
    """numpy array in R memory order (column-major)"""
    return np.asfortranarray(np.asarray(a, dtype=dtype))


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _counts(y):
    y = np.asarray(y)
    if y.dtype.kind in "iu" or y.dtype == np.bool_:
        return np.asfortranarray(y.astype(np.int32, copy=False)), L.DSQ_Y_INT32
    return np.asfortranarray(y.astype(np.float64, copy=False)), L.DSQ_Y_FLOAT64


def _scalar_len1(v, name):
    a = np.asarray(v, dtype=np.float64).reshape(-1)
    if a.size != 1:
        raise ValueError("%s must be a single value" % name)
    return float(a[0])


def fitBeta(ySEXP, xSEXP, nfSEXP, alpha_hatSEXP, contrastSEXP, beta_matSEXP, lambdaSEXP, weightsSEXP,
            useWeightsSEXP, tolSEXP, maxitSEXP, useQRSEXP, minmuSEXP, want_mu=False, mu_floor=0.0,
            want_hat=True, row_ranges=None):
    """row_ranges: [(lo, cnt), ...] walks the genes range by range through dsq_fit_beta_rows, as the R shim does
    between its R_CheckUserInterrupt() polls (the outputs of rows outside the ranges stay zero)"""
    y, ytype = _counts(ySEXP)
    x = _fcol(xSEXP); nf = _fcol(nfSEXP); b0 = _fcol(beta_matSEXP)
    if y.ndim != 2 or x.ndim != 2:
        raise ValueError("ySEXP and xSEXP must be matrices")
    n, m = y.shape
    p = x.shape[1]
    if x.shape[0] != m or nf.shape != (n, m) or b0.shape != (n, p):
        raise ValueError("non-conformable arguments")
    useW = bool(useWeightsSEXP)
    w = _fcol(weightsSEXP) if useW else None          # unused weights are neither converted nor uploaded
    if w is not None and w.shape != (n, m):
        raise ValueError("weights must be n x m")
    alpha = np.ascontiguousarray(np.broadcast_to(np.asarray(alpha_hatSEXP, np.float64).reshape(-1), (n,)))
    contrast = np.ascontiguousarray(np.asarray(contrastSEXP, np.float64).reshape(-1))
    lam = np.ascontiguousarray(np.asarray(lambdaSEXP, np.float64).reshape(-1))
    if contrast.size != p or lam.size != p:
        raise ValueError("contrast and lambda must have length ncol(x)")
    out = {"beta_mat": np.zeros((n, p), order="F"), "beta_var_mat": np.zeros((n, p), order="F"),
           "iter": np.zeros(n), "hat_diagonals": np.zeros((n, m), order="F") if want_hat else None,
           "contrast_num": np.zeros((n, 1)), "contrast_denom": np.zeros((n, 1)), "deviance": np.zeros(n)}
    mu = np.zeros((n, m), order="F") if want_mu else None
    a = L.DsqFitBetaArgs(n=n, m=m, p=p, layout=L.DSQ_LAYOUT_R, ld=0, y=_ptr(y), y_type=ytype, x=_ptr(x),
                         nf=_ptr(nf), nf_is_vector=0, alpha_hat=_ptr(alpha), contrast=_ptr(contrast),
                         beta_mat=_ptr(b0), lambda_=_ptr(lam), weights=_ptr(w) if useW else None,
                         useWeights=int(useW), tol=_scalar_len1(tolSEXP, "tol"),
                         maxit=int(_scalar_len1(maxitSEXP, "maxit")), useQR=int(bool(useQRSEXP)),
                         minmu=_scalar_len1(minmuSEXP, "minmu"))
    o = L.DsqFitBetaOut(beta_mat=_ptr(out["beta_mat"]), beta_var_mat=_ptr(out["beta_var_mat"]),
                        iter=_ptr(out["iter"]), hat_diagonals=_ptr(out["hat_diagonals"]),
                        contrast_num=_ptr(out["contrast_num"]), contrast_denom=_ptr(out["contrast_denom"]),
                        deviance=_ptr(out["deviance"]), mu=_ptr(mu), mu_floor=float(mu_floor))
    if row_ranges is None:
        L.check(L.lib().dsq_fit_beta(C.byref(a), C.byref(o)))
    else:
        for lo, cnt in row_ranges:
            L.check(L.lib().dsq_fit_beta_rows(C.byref(a), C.byref(o), int(lo), int(cnt)))
    if want_mu:
        out["mu"] = mu
    return out


def fitDisp(ySEXP, xSEXP, mu_hatSEXP, log_alphaSEXP, log_alpha_prior_meanSEXP, log_alpha_prior_sigmasqSEXP,
            min_log_alphaSEXP, kappa_0SEXP, tolSEXP, maxitSEXP, usePriorSEXP, weightsSEXP, useWeightsSEXP,
            weightThresholdSEXP, useCRSEXP):
    y, ytype = _counts(ySEXP)
    x = _fcol(xSEXP); mu = _fcol(mu_hatSEXP)
    n, m = y.shape
    p = x.shape[1]
    if x.shape[0] != m or mu.shape != (n, m):
        raise ValueError("non-conformable arguments")
    useW = bool(useWeightsSEXP)
    w = _fcol(weightsSEXP) if useW else None
    if w is not None and w.shape != (n, m):
        raise ValueError("weights must be n x m")
    la = np.ascontiguousarray(np.broadcast_to(np.asarray(log_alphaSEXP, np.float64).reshape(-1), (n,)))
    pm = np.ascontiguousarray(np.broadcast_to(np.asarray(log_alpha_prior_meanSEXP, np.float64).reshape(-1), (n,)))
    out = {k: np.zeros(n) for k in ("log_alpha", "last_change", "initial_lp", "initial_dlp", "last_lp",
                                     "last_dlp", "last_d2lp")}
    out["iter"] = np.zeros(n, dtype=np.int32)
    out["iter_accept"] = np.zeros(n, dtype=np.int32)
    a = L.DsqFitDispArgs(n=n, m=m, p=p, layout=L.DSQ_LAYOUT_R, ld=0, y=_ptr(y), y_type=ytype, x=_ptr(x),
                         mu_hat=_ptr(mu), log_alpha=_ptr(la), log_alpha_prior_mean=_ptr(pm),
                         log_alpha_prior_sigmasq=_scalar_len1(log_alpha_prior_sigmasqSEXP, "sigmasq"),
                         min_log_alpha=_scalar_len1(min_log_alphaSEXP, "min_log_alpha"),
                         kappa_0=_scalar_len1(kappa_0SEXP, "kappa_0"), tol=_scalar_len1(tolSEXP, "tol"),
                         maxit=int(_scalar_len1(maxitSEXP, "maxit")), usePrior=int(bool(usePriorSEXP)),
                         weights=_ptr(w), useWeights=int(useW),
                         weightThreshold=_scalar_len1(weightThresholdSEXP, "weightThreshold"),
                         useCR=int(bool(useCRSEXP)))
    o = L.DsqFitDispOut(**{k: _ptr(v) for k, v in out.items()})
    L.check(L.lib().dsq_fit_disp(C.byref(a), C.byref(o)))
    return out


def fitDispGrid(ySEXP, xSEXP, mu_hatSEXP, disp_gridSEXP, log_alpha_prior_meanSEXP,
                log_alpha_prior_sigmasqSEXP, usePriorSEXP, weightsSEXP, useWeightsSEXP, weightThresholdSEXP,
                useCRSEXP):
    y, ytype = _counts(ySEXP)
    x = _fcol(xSEXP); mu = _fcol(mu_hatSEXP)
    n, m = y.shape
    p = x.shape[1]
    if x.shape[0] != m or mu.shape != (n, m):
        raise ValueError("non-conformable arguments")
    useW = bool(useWeightsSEXP)
    w = _fcol(weightsSEXP) if useW else None
    grid = np.ascontiguousarray(np.asarray(disp_gridSEXP, np.float64).reshape(-1))
    pm = np.ascontiguousarray(np.broadcast_to(np.asarray(log_alpha_prior_meanSEXP, np.float64).reshape(-1), (n,)))
    la = np.zeros(n)
    a = L.DsqFitDispGridArgs(n=n, m=m, p=p, layout=L.DSQ_LAYOUT_R, ld=0, y=_ptr(y), y_type=ytype, x=_ptr(x),
                             mu_hat=_ptr(mu), disp_grid=_ptr(grid), ngrid=grid.size,
                             log_alpha_prior_mean=_ptr(pm),
                             log_alpha_prior_sigmasq=_scalar_len1(log_alpha_prior_sigmasqSEXP, "sigmasq"),
                             usePrior=int(bool(usePriorSEXP)), weights=_ptr(w), useWeights=int(useW),
                             weightThreshold=_scalar_len1(weightThresholdSEXP, "weightThreshold"),
                             useCR=int(bool(useCRSEXP)))
    o = L.DsqFitDispGridOut(log_alpha=_ptr(la))
    L.check(L.lib().dsq_fit_disp_grid(C.byref(a), C.byref(o)))
    return {"log_alpha": la}


_QR_CACHE = {}


def design_qr(x):
    """memoised thin QR of a model matrix (asked for by every prefit / linear_mu call of a DESeq())"""
    x = np.ascontiguousarray(x, dtype=np.float64)
    key = (x.shape, x.tobytes())
    v = _QR_CACHE.get(key)
    if v is None:
        if len(_QR_CACHE) > 64:
            _QR_CACHE.clear()
        v = _QR_CACHE[key] = _design_qr(x)
    return v


def _design_qr(x):
    """thin QR of the model matrix, taken on the host like the reference does with stats::qr
    (R/fitNbinomGLMs.R:139-143, R/core.R:2455-2457): Q (m x p), A = X R^-1 (m x p), R (p x p)"""
    x = np.asarray(x, np.float64)
    q, r = np.linalg.qr(x)
    a = x @ np.linalg.inv(r)
    return np.asfortranarray(q), np.asfortranarray(a), np.asfortranarray(r)


def prefitMoments(counts, nf, x, weights=None, useWeights=False):
    """baseMean / baseVar / allZero (R/core.R:2138-2146), roughDispEstimate (:2422-2437) and the
    QR least-squares start values (R/fitNbinomGLMs.R:139-145) in one kernel launch (dsq_prefit_moments)."""
    y, ytype = _counts(counts)
    nf = _fcol(nf)
    n, m = y.shape
    q, a, r = design_qr(x)
    p = q.shape[1]
    w = _fcol(weights) if useWeights else None
    bm = np.zeros(n); bv = np.zeros(n); az = np.zeros(n, dtype=np.int32); rd = np.zeros(n)
    b0 = np.zeros((n, p), order="F")
    args = L.DsqPrefitArgs(n=n, m=m, p=p, layout=L.DSQ_LAYOUT_R, ld=0, y=_ptr(y), y_type=ytype, nf=_ptr(nf),
                           nf_is_vector=0, weights=_ptr(w), useWeights=int(bool(useWeights)), q=_ptr(q), a=_ptr(a),
                           r=_ptr(r))
    out = L.DsqPrefitOut(baseMean=_ptr(bm), baseVar=_ptr(bv), allZero=_ptr(az), roughDisp=_ptr(rd),
                         beta_init=_ptr(b0))
    L.check(L.lib().dsq_prefit_moments(C.byref(args), C.byref(out)))
    return {"baseMean": bm, "baseVar": bv, "allZero": az.astype(bool), "roughDisp": rd, "beta_init": b0}


def linearMu(counts, nf, x, mu_floor=0.0):
    """linearModelMuNormalized (R/core.R:2454-2471) through dsq_linear_mu"""
    y, ytype = _counts(counts)
    nf = _fcol(nf)
    n, m = y.shape
    q, a, r = design_qr(x)
    q, a = _fcol(q), _fcol(a)
    mu = np.zeros((n, m), order="F")
    args = L.DsqPrefitArgs(n=n, m=m, p=q.shape[1], layout=L.DSQ_LAYOUT_R, ld=0, y=_ptr(y), y_type=ytype, nf=_ptr(nf),
                           nf_is_vector=0, weights=None, useWeights=0, q=_ptr(q), a=_ptr(a), r=None)
    L.check(L.lib().dsq_linear_mu(C.byref(args), float(mu_floor), _ptr(mu)))
    return mu


def nbinomLogLike(counts, mu, disp, weights, useWeights):
    """R/core.R:2208-2217 through dsq_nbinom_loglike"""
    y, ytype = _counts(counts)
    mu = _fcol(mu)
    n, m = y.shape
    w = _fcol(weights) if useWeights else None
    d = np.ascontiguousarray(np.broadcast_to(np.asarray(disp, np.float64).reshape(-1), (n,)))
    out = np.zeros(n)
    args = L.DsqLogLikeArgs(n=n, m=m, layout=L.DSQ_LAYOUT_R, ld=0, y=_ptr(y), y_type=ytype, mu=_ptr(mu),
                            disp=_ptr(d), weights=_ptr(w), useWeights=int(bool(useWeights)))
    L.check(L.lib().dsq_nbinom_loglike(C.byref(args), _ptr(out)))
    return out


def interceptFit(counts, nf, alpha, weights=None, useWeights=False, mu_floor=0.0, want_hat=True):
    """closed form of R/fitNbinomGLMs.R:99-137 (design ~ 1) through dsq_intercept_fit"""
    y, ytype = _counts(counts)
    nf = _fcol(nf)
    n, m = y.shape
    w = _fcol(weights) if useWeights else None
    a = np.ascontiguousarray(np.broadcast_to(np.asarray(alpha, np.float64).reshape(-1), (n,)))
    b, se = np.zeros(n), np.zeros(n)
    mu = np.zeros((n, m), order="F")
    hat = np.zeros((n, m), order="F") if want_hat else None
    args = L.DsqInterceptArgs(n=n, m=m, layout=L.DSQ_LAYOUT_R, ld=0, y=_ptr(y), y_type=ytype, nf=_ptr(nf),
                              nf_is_vector=0, weights=_ptr(w), useWeights=int(bool(useWeights)), alpha=_ptr(a),
                              mu_floor=float(mu_floor))
    out = L.DsqInterceptOut(beta_log2=_ptr(b), betaSE=_ptr(se), mu=_ptr(mu), hat=_ptr(hat))
    L.check(L.lib().dsq_intercept_fit(C.byref(args), C.byref(out)))
    return {"beta": b, "betaSE": se, "mu": mu, "hat_diagonals": hat}


def optimRows(counts, x, nf, alpha, lam, weights, useWeights, beta_start, minmu=0.5):
    """fitNbinomGLMsOptim (R/fitNbinomGLMs.R:340-407) on the given rows through dsq_optim_rows.  lam: prior
    precisions on the log2 scale; beta_start: log2-scale start values."""
    y, ytype = _counts(counts)
    x = _fcol(x); nf = _fcol(nf)
    n, m = y.shape
    p = x.shape[1]
    w = _fcol(weights) if useWeights else None
    a = np.ascontiguousarray(np.broadcast_to(np.asarray(alpha, np.float64).reshape(-1), (n,)))
    lam = np.ascontiguousarray(np.asarray(lam, np.float64).reshape(-1))
    b0 = _fcol(np.asarray(beta_start, np.float64).reshape(n, p))
    beta = np.zeros((n, p), order="F"); se = np.zeros((n, p), order="F")
    conv = np.zeros(n, dtype=np.int32); mu = np.zeros((n, m), order="F"); ll = np.zeros(n)
    args = L.DsqOptimArgs(n=n, m=m, p=p, layout=L.DSQ_LAYOUT_R, ld=0, y=_ptr(y), y_type=ytype, x=_ptr(x), nf=_ptr(nf),
                          nf_is_vector=0, alpha_hat=_ptr(a), lambda_=_ptr(lam), weights=_ptr(w),
                          useWeights=int(bool(useWeights)), beta_start=_ptr(b0), minmu=float(minmu))
    out = L.DsqOptimOut(beta=_ptr(beta), betaSE=_ptr(se), conv=_ptr(conv), mu=_ptr(mu), logLike=_ptr(ll))
    L.check(L.lib().dsq_optim_rows(C.byref(args), C.byref(out)))
    return {"beta": beta, "betaSE": se, "conv": conv.astype(bool), "mu": mu, "logLike": ll}


def coef_factor_codes(factors, expanded=False):
    """the column coding dsq_deseq takes for the beta prior: for model.matrix(~ f1 + f2 + ...) of the design `factors`
    (ordered dict name -> integer level codes, 0 = reference level) 0 for the intercept and f >= 1 for a level
    indicator of the f-th factor; `expanded` = the same for the expanded model matrix (R/expanded.R:1-18: every level)"""
    codes = [0]
    for f, (_, lv) in enumerate(factors.items(), start=1):
        k = int(np.asarray(lv).max())
        codes += [f] * (k + 1 if expanded else k)
    return np.ascontiguousarray(codes, dtype=np.int32)


def estimateBetaPriorVarHost(mle_beta, baseMean, dispFit, allZero, coef_factor, prior_coef_factor=None, prior_coef_src=None):
    """dsq_beta_prior_var: estimateBetaPriorVar (R/core.R:1601-1689) on host arrays, no device work"""
    mle = _fcol(mle_beta)
    n, p = mle.shape
    bm, df = np.ascontiguousarray(baseMean, np.float64), np.ascontiguousarray(dispFit, np.float64)
    az = np.ascontiguousarray(np.asarray(allZero).astype(np.int32))
    cf = np.ascontiguousarray(coef_factor, dtype=np.int32)
    pcf = None if prior_coef_factor is None else np.ascontiguousarray(prior_coef_factor, dtype=np.int32)
    pcs = None if prior_coef_src is None else np.ascontiguousarray(prior_coef_src, dtype=np.int32)
    pp = p if pcf is None else pcf.size
    out = np.zeros(pp)
    args = L.DsqBetaPriorArgs(n=n, p=p, mle_beta=_ptr(mle), baseMean=_ptr(bm), dispFit=_ptr(df), allZero=_ptr(az),
                              coef_factor=_ptr(cf), expanded=int(pcf is not None), p_prior=int(pp),
                              prior_coef_factor=_ptr(pcf), prior_coef_src=_ptr(pcs), upperQuantile=0.05)
    L.check(L.lib().dsq_beta_prior_var(C.byref(args), _ptr(out)))
    return out


def DESeq(counts, x, sizeFactors=None, test="Wald", reduced=None, normalizationFactors=None, weights=None,
          minReplicatesForReplace=7, betaTol=1e-8, maxit=100, useQR=True,
          minmu=0.5, disp_maxit=100, useCR=True, assays=("mu", "H", "cooks"),
          betaPrior=False, factors=None, modelMatrixType=None, betaPriorVar=None, coef_factor=None,
          fitType="parametric", dispFit=None, geneEstOnly=False, dispPriorVar=None):
    """dsq_deseq: DESeq() behind ONE host-pointer call (what r_shim.c binds as _DESeq2_mi355x_DESeq; the R-side glue is
    in INTEGRATION.md).  counts: n x m integer matrix in R orientation; x: m x p model matrix; sizeFactors: m.  The three
    design-only quantities the R caller computes with qr() / qf() / trigamma() come from numpy / scipy here.  Returns the
    per-gene columns (NA = NaN; integer columns as float64 with NaN), the requested n x m assays, the dispersion
    function and the status counters.  fitType: "parametric" (a trend that does not fit is an error, DSQ_ERR_FIT: the R
    caller then takes the reference's route to locfit), "mean", or "parametric_or_mean" (the mean substituted on the device).
    A trend the library does not fit (fitType = "local", dispersionFunction<-): geneEstOnly = True returns after
    estimateDispersionsGeneEst; the caller evaluates its trend at res["baseMean"] and calls again with dispFit = those values
    (count outliers are then flagged but not replaced: the refit is the caller's, R/core.R:2484-2563).  dispPriorVar: the
    argument of estimateDispersionsMAP (R/core.R:989-994); required for m - p <= 3, where R's own estimate is a seeded
    Monte-Carlo match (:1155-1190) the library leaves to the caller."""
    from scipy import special as sps
    if fitType not in L.DSQ_FIT:
        raise ValueError("fitType should be one of %s" % sorted(L.DSQ_FIT))
    from scipy.stats import f as fdist
    y, ytype = _counts(counts)
    x = _fcol(x)
    n, m = y.shape
    p = x.shape[1]
    sf = None if sizeFactors is None or normalizationFactors is not None else np.ascontiguousarray(sizeFactors, dtype=np.float64)
    nfm = None if normalizationFactors is None else _fcol(normalizationFactors)
    wts = None if weights is None else _fcol(weights)
    q, a, r = design_qr(x)
    if m - p > 0:
        cutoff, evld = float(fdist.ppf(.99, p, m - p)), float(sps.polygamma(1, (m - p) / 2.0))
    else:
        cutoff, evld = 1.0, 1.0           # (the library rejects m <= p itself)
    wald = test == "Wald"
    if not wald and test != "LRT":
        raise ValueError("test should be either 'Wald' or 'LRT'")
    grid = np.ascontiguousarray(np.linspace(np.log(1e-8), np.log(max(10, m)), 20))          # R/wrappers.R:70-72
    xr = qr_ = rr_ = None
    p_red = 0
    if not wald and reduced is not None:
        red = _fcol(reduced)
        if not (red.shape[1] == 1 and (red == 1).all()):
            xr, p_red = red, red.shape[1]
            qr_, _, rr_ = design_qr(red)
    # nbinomWaldTest(betaPrior = TRUE): the model matrix of the prior pass and what its columns are (R/core.R:1374-1380)
    pcol, xe, cf, pcf = p, None, None, None
    if betaPrior:
        if not wald:
            raise ValueError("betaPrior: the Wald test only")
        mmt = modelMatrixType or ("expanded" if factors is not None else "standard")
        if factors is not None:
            cf = coef_factor_codes(factors)
        else:
            cf = np.ascontiguousarray(coef_factor if coef_factor is not None else [0] + [-1] * (p - 1), dtype=np.int32)
        if cf.size != p:
            raise ValueError("the design factors do not describe the %d columns of the model matrix" % p)
        if mmt == "expanded":
            if factors is None:
                raise ValueError("an expanded model matrix needs the design factors")
            from . import core
            xe = _fcol(core.makeExpandedModelMatrix(factors)[0])
            pcf = coef_factor_codes(factors, expanded=True)
            pcol = xe.shape[1]
    fit_in = None if dispFit is None else np.ascontiguousarray(dispFit, dtype=np.float64)
    if fit_in is not None and fit_in.shape != (n,):
        raise ValueError("dispFit needs one value per gene")
    bpv_in = None if betaPriorVar is None else np.ascontiguousarray(betaPriorVar, dtype=np.float64)
    if bpv_in is not None and bpv_in.size != pcol:
        raise ValueError("betaPriorVar needs one value per column of the (expanded) model matrix")
    f64 = lambda *sh: np.full(sh, np.nan, order="F")                    # noqa: E731
    i32 = lambda: np.full(n, -1, dtype=np.int32)                         # noqa: E731
    d = {k: f64(n) for k in ("baseMean", "baseVar", "dispGeneEst", "dispFit", "dispMAP", "dispersion", "betaIter",
                             "logLike", "maxCooks")}
    d.update({k: i32() for k in ("allZero", "dispGeneIter", "dispIter", "dispOutlier", "betaConv", "replace", "weightsFail")})
    d.update(beta=f64(n, pcol), betaSE=f64(n, pcol))
    if betaPrior:
        d["mle_beta"] = f64(n, p)
    if wald:
        d.update(stat=f64(n, pcol), pvalue=f64(n, pcol))
    else:
        d["logLikeReduced"] = f64(n)
    for k in assays:
        d[k] = np.zeros((n, m), order="F", dtype=np.int32 if k == "replaceCounts" else np.float64)
    args = L.DsqDeseqHostArgs(
        n=n, m=m, p=p, counts=_ptr(y), y_type=ytype, x=_ptr(x), sizeFactors=_ptr(sf), normalizationFactors=_ptr(nfm),
        weights=_ptr(wts), q=_ptr(q), r=_ptr(r), xrinv=_ptr(a),
        test=0 if wald else 1, x_reduced=_ptr(xr), q_reduced=_ptr(qr_), r_reduced=_ptr(rr_), p_reduced=int(p_red),
        minReplicatesForReplace=float(minReplicatesForReplace), cooksCutoff=cutoff,
        expVarLogDisp=evld, betaTol=float(betaTol), minmu=float(minmu), maxit=int(maxit), useQR=int(bool(useQR)),
        disp_maxit=int(disp_maxit), useCR=int(bool(useCR)), disp_grid=_ptr(grid), ngrid=int(grid.size),
        betaPrior=int(bool(betaPrior)), x_prior=_ptr(xe), p_prior=int(pcol), coef_factor=_ptr(cf),
        prior_coef_factor=_ptr(pcf), prior_coef_src=None, betaPriorVar=_ptr(bpv_in), fitType=L.DSQ_FIT[fitType],
        dispFit=_ptr(fit_in), geneEstOnly=int(bool(geneEstOnly)),
        dispPriorVar=0.0 if dispPriorVar is None else float(dispPriorVar))
    out = L.DsqDeseqHostOut(**{k: _ptr(v) for k, v in d.items()})
    L.check(L.lib().dsq_deseq(C.byref(args), C.byref(out)))
    res = {}
    for k, v in d.items():
        if v.dtype == np.int32 and k != "replaceCounts":
            f = v.astype(np.float64)
            f[v < 0] = np.nan
            res[k] = f
        else:
            res[k] = v
    res["status"] = {k: int(out.status[i]) for k, i in L.DSQ_ST.items()}
    mean_used = out.dispersionFunction[L.DSQ_SC_FIT_USED] == L.DSQ_FIT["mean"]
    given = out.dispersionFunction[L.DSQ_SC_FIT_USED] == L.DSQ_FIT["given"]
    res["dispersionFunction"] = {"fitType": "given" if given else "mean" if mean_used else "parametric",
                                 "coefficients": float(out.dispersionFunction[0]) if mean_used else np.array(out.dispersionFunction[0:2]),
                                 "varLogDispEsts": float(out.dispersionFunction[2]),
                                 "dispPriorVar": float(out.dispersionFunction[3])}
    if betaPrior:
        res["betaPriorVar"] = np.array(out.betaPriorVar[:pcol])
    res["cooksCutoff"] = cutoff
    res["df"] = p - (p_red if p_red else 1)
    return res


_CELL_CACHE = {}


def cell_index(x):
    """design cells (nOrMoreInCell, R/core.R:2366-2371): samples with identical model-matrix rows"""
    x = np.ascontiguousarray(x, dtype=np.float64)
    key = (x.shape, x.tobytes())
    v = _CELL_CACHE.get(key)
    if v is None:
        if len(_CELL_CACHE) > 64:
            _CELL_CACHE.clear()
        _, inv = np.unique(x, axis=0, return_inverse=True)
        v = _CELL_CACHE[key] = np.ascontiguousarray(inv.reshape(-1), dtype=np.int32)
    return v


def cooksDistance(counts, nf, mu, H, x):
    """calculateCooksDistance + recordMaxCooks (R/core.R:2333-2359) through dsq_cooks_distance;
    `x` is the dispersion model matrix."""
    y, ytype = _counts(counts)
    nf, mu, H = _fcol(nf), _fcol(mu), _fcol(H)
    n, m = y.shape
    cells = cell_index(x)
    ck = np.zeros((n, m), order="F")
    mx, rd = np.zeros(n), np.zeros(n)
    args = L.DsqCooksArgs(n=n, m=m, p=np.asarray(x).shape[1], layout=L.DSQ_LAYOUT_R, ld=0, y=_ptr(y), y_type=ytype,
                          nf=_ptr(nf), nf_is_vector=0, mu=_ptr(mu), H=_ptr(H), cell_of=_ptr(cells),
                          ncell=int(cells.max()) + 1)
    out = L.DsqCooksOut(cooks=_ptr(ck), maxCooks=_ptr(mx), robustDisp=_ptr(rd))
    L.check(L.lib().dsq_cooks_distance(C.byref(args), C.byref(out)))
    return {"cooks": ck, "maxCooks": mx, "robustDisp": rd}


def replaceOutliers(counts, nf, cooks, cooksCutoff, replaceable, trim=0.2):
    """replaceOutliers (R/core.R:2069-2115) through dsq_replace_outliers"""
    y, ytype = _counts(counts)
    nf, ck = _fcol(nf), _fcol(cooks)
    n, m = y.shape
    rep = np.ascontiguousarray(np.asarray(replaceable).astype(np.int32))
    newc = np.zeros((n, m), dtype=np.int32, order="F")
    flag = np.zeros(n, dtype=np.int32)
    args = L.DsqReplaceArgs(n=n, m=m, layout=L.DSQ_LAYOUT_R, ld=0, y=_ptr(y), y_type=ytype, nf=_ptr(nf),
                            nf_is_vector=0, cooks=_ptr(ck), cooksCutoff=float(cooksCutoff), trim=float(trim),
                            replaceable=_ptr(rep))
    out = L.DsqReplaceOut(newCounts=_ptr(newc), replace=_ptr(flag))
    L.check(L.lib().dsq_replace_outliers(C.byref(args), C.byref(out)))
    return {"counts": newc, "replace": flag.astype(bool)}


_FIT_ERRORS = {1: "parametric dispersion fit failed", 2: "dispersion fit did not converge"}


def parametricDispersionFit(means, disps):
    """R/core.R:2166-2190 through dsq_parametric_dispersion_fit; raises RuntimeError with the
    reference's messages so the caller can fall back like R/core.R:885-893"""
    means = np.ascontiguousarray(means, dtype=np.float64)
    disps = np.ascontiguousarray(disps, dtype=np.float64)
    coefs = np.zeros(2)
    st = np.zeros(1, dtype=np.int32)
    L.check(L.lib().dsq_parametric_dispersion_fit(_ptr(means), _ptr(disps), means.size, _ptr(coefs), _ptr(st)))
    if st[0] != 0:
        raise RuntimeError(_FIT_ERRORS.get(int(st[0]), "parametric dispersion fit failed"))
    return coefs


def test_math(op, a, b=None, c=None):
    """Evaluate one device-math primitive on the GPU (parity hook, see dsq_test_math)."""
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = None if b is None else np.ascontiguousarray(np.broadcast_to(np.asarray(b, np.float64), a.shape))
    c = None if c is None else np.ascontiguousarray(np.broadcast_to(np.asarray(c, np.float64), a.shape))
    out = np.empty_like(a)
    L.check(L.lib().dsq_test_math(int(op), _ptr(a), _ptr(b), _ptr(c), _ptr(out), a.size))
    return out


UNARY_OPS = {"exp": 0, "log": 1, "log1p": 2, "lgamma": 3, "digamma": 4, "trigamma": 5, "stirlerr": 6, "pnorm_upper2": 9}


def unary(name, x):
    """n-vector log / exp / ... in the engine's pinned f64 arithmetic (csrc/dsq_math.hpp)"""
    x = np.asarray(x, np.float64)
    return test_math(UNARY_OPS[name], x.reshape(-1)).reshape(x.shape)


# ------------------------------------------------------------------------------------
# device-resident flavour: torch CUDA tensors, current stream, no host round trip
# ------------------------------------------------------------------------------------
def _t_ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class GeneMajor:
    """n x m device matrix in the engine's native layout (row-major, leading dim ld)."""

    def __init__(self, tensor, m):
        assert tensor.dim() == 2 and tensor.is_contiguous()
        self.t = tensor
        self.n = tensor.shape[0]
        self.ld = tensor.shape[1]
        self.m = m

    def view(self):
        return self.t[:, : self.m]


def gene_major_ld(m):
    return (m + 7) & ~7


def fitBeta_dev(y, x, nf, alpha_hat, contrast, beta_mat, lambda_, weights, useWeights, tol, maxit, useQR,
                minmu, want_hat=True, want_mu=False, mu_floor=0.0, nf_is_vector=False, cells=None):
    """All array arguments are torch CUDA tensors.  y / nf / weights: GeneMajor (y int32) --
    outputs hat_diagonals / mu are GeneMajor too; x: (m, p) column-major i.e. a (p, m)
    contiguous tensor is passed as `x` (see `design_to_device`)."""
    import torch
    assert isinstance(y, GeneMajor)
    n, m, ld = y.n, y.m, y.ld
    p = x.shape[0]
    dev = y.t.device
    f64 = dict(dtype=torch.float64, device=dev)
    # all per-gene outputs live in ONE buffer so the caller brings them to the host with one copy
    pack = torch.empty((2 * p + 4, n), **f64)
    out = {"beta_mat": pack[:p], "beta_var_mat": pack[p:2 * p], "iter": pack[2 * p], "contrast_num": pack[2 * p + 1],
           "contrast_denom": pack[2 * p + 2], "deviance": pack[2 * p + 3]}
    hat = GeneMajor(torch.empty((n, ld), **f64), m) if want_hat else None
    mu = GeneMajor(torch.empty((n, ld), **f64), m) if want_mu else None
    if nf_is_vector:
        nf_ptr = _t_ptr(nf)
    else:
        assert isinstance(nf, GeneMajor) and nf.ld == ld
        nf_ptr = _t_ptr(nf.t)
    if useWeights:
        assert isinstance(weights, GeneMajor) and weights.ld == ld
    a = L.DsqFitBetaArgs(n=n, m=m, p=p, layout=L.DSQ_LAYOUT_GENE_MAJOR, ld=ld, y=_t_ptr(y.t),
                         y_type=L.DSQ_Y_INT32, x=_t_ptr(x), nf=nf_ptr, nf_is_vector=int(nf_is_vector),
                         alpha_hat=_t_ptr(alpha_hat), contrast=_t_ptr(contrast), beta_mat=_t_ptr(beta_mat),
                         lambda_=_t_ptr(lambda_), weights=_t_ptr(weights.t) if useWeights else None,
                         useWeights=int(bool(useWeights)), tol=float(tol), maxit=int(maxit),
                         useQR=int(bool(useQR)), minmu=float(minmu),
                         cell_of=_ptr(cells) if cells is not None else None,
                         ncell=(int(cells.max()) + 1) if cells is not None else 0)
    o = L.DsqFitBetaOut(beta_mat=_t_ptr(out["beta_mat"]), beta_var_mat=_t_ptr(out["beta_var_mat"]),
                        iter=_t_ptr(out["iter"]), hat_diagonals=_t_ptr(hat.t) if hat else None,
                        contrast_num=_t_ptr(out["contrast_num"]), contrast_denom=_t_ptr(out["contrast_denom"]),
                        deviance=_t_ptr(out["deviance"]), mu=_t_ptr(mu.t) if mu else None,
                        mu_floor=float(mu_floor))
    L.check(L.lib().dsq_fit_beta_dev(C.byref(a), C.byref(o), _stream()))
    out["hat_diagonals"] = hat
    out["mu"] = mu
    out["_pack"] = pack
    return out


def fitDisp_dev(y, x, mu_hat, log_alpha, log_alpha_prior_mean, log_alpha_prior_sigmasq, min_log_alpha,
                kappa_0, tol, maxit, usePrior, weights, useWeights, weightThreshold, useCR, want_d2lp=True, cells=None):
    import torch
    assert isinstance(y, GeneMajor) and isinstance(mu_hat, GeneMajor) and mu_hat.ld == y.ld
    n, m, ld = y.n, y.m, y.ld
    p = x.shape[0]
    dev = y.t.device
    f64 = dict(dtype=torch.float64, device=dev)
    keys = ("log_alpha", "last_change", "initial_lp", "initial_dlp", "last_lp", "last_dlp", "last_d2lp")
    pack = torch.empty((len(keys) + 1, n), **f64)       # one buffer, one device-to-host copy for the caller
    out = {k: pack[i] for i, k in enumerate(keys)}
    if not want_d2lp:
        out["last_d2lp"] = None
    ints = pack[len(keys)].view(torch.int32)            # the last row holds the two int32 counters
    out["iter"], out["iter_accept"] = ints[:n], ints[n:]
    a = L.DsqFitDispArgs(n=n, m=m, p=p, layout=L.DSQ_LAYOUT_GENE_MAJOR, ld=ld, y=_t_ptr(y.t),
                         y_type=L.DSQ_Y_INT32, x=_t_ptr(x), mu_hat=_t_ptr(mu_hat.t), log_alpha=_t_ptr(log_alpha),
                         log_alpha_prior_mean=_t_ptr(log_alpha_prior_mean),
                         log_alpha_prior_sigmasq=float(log_alpha_prior_sigmasq),
                         min_log_alpha=float(min_log_alpha), kappa_0=float(kappa_0), tol=float(tol),
                         maxit=int(maxit), usePrior=int(bool(usePrior)),
                         weights=_t_ptr(weights.t) if useWeights else None, useWeights=int(bool(useWeights)),
                         weightThreshold=float(weightThreshold), useCR=int(bool(useCR)),
                         cell_of=_ptr(cells) if cells is not None else None,
                         ncell=(int(cells.max()) + 1) if cells is not None else 0)
    o = L.DsqFitDispOut(**{k: _t_ptr(v) for k, v in out.items()})
    L.check(L.lib().dsq_fit_disp_dev(C.byref(a), C.byref(o), _stream()))
    out = {k: v for k, v in out.items() if v is not None}
    out["_pack"] = pack
    return out


def fitDispGrid_dev(y, x, mu_hat, disp_grid, log_alpha_prior_mean, log_alpha_prior_sigmasq, usePrior, weights,
                    useWeights, weightThreshold, useCR, cells=None):
    import torch
    assert isinstance(y, GeneMajor) and isinstance(mu_hat, GeneMajor) and mu_hat.ld == y.ld
    n, m, ld = y.n, y.m, y.ld
    p = x.shape[0]
    la = torch.empty(n, dtype=torch.float64, device=y.t.device)
    a = L.DsqFitDispGridArgs(n=n, m=m, p=p, layout=L.DSQ_LAYOUT_GENE_MAJOR, ld=ld, y=_t_ptr(y.t),
                             y_type=L.DSQ_Y_INT32, x=_t_ptr(x), mu_hat=_t_ptr(mu_hat.t),
                             disp_grid=_t_ptr(disp_grid), ngrid=int(disp_grid.numel()),
                             log_alpha_prior_mean=_t_ptr(log_alpha_prior_mean),
                             log_alpha_prior_sigmasq=float(log_alpha_prior_sigmasq), usePrior=int(bool(usePrior)),
                             weights=_t_ptr(weights.t) if useWeights else None, useWeights=int(bool(useWeights)),
                             weightThreshold=float(weightThreshold), useCR=int(bool(useCR)),
                             cell_of=_ptr(cells) if cells is not None else None,
                             ncell=(int(cells.max()) + 1) if cells is not None else 0)
    o = L.DsqFitDispGridOut(log_alpha=_t_ptr(la))
    L.check(L.lib().dsq_fit_disp_grid_dev(C.byref(a), C.byref(o), _stream()))
    return {"log_alpha": la}


def to_gene_major(t_r, dtype=None):
    """(n, m) torch CUDA tensor holding an R-layout matrix, given as its TRANSPOSE storage:
    pass a contiguous (m, n) tensor (column-major n x m).  Returns GeneMajor."""
    import torch
    m, n = t_r.shape
    ld = gene_major_ld(m)
    if t_r.dtype == torch.int32:
        dst = torch.zeros((n, ld), dtype=torch.int32, device=t_r.device)
        L.check(L.lib().dsq_to_gene_major_i32(_t_ptr(t_r), _t_ptr(dst), n, m, ld, _stream()))
    else:
        assert t_r.dtype == torch.float64
        dst = torch.zeros((n, ld), dtype=torch.float64, device=t_r.device)
        L.check(L.lib().dsq_to_gene_major_f64(_t_ptr(t_r), _t_ptr(dst), n, m, ld, _stream()))
    return GeneMajor(dst, m)


def from_gene_major(gm):
    """GeneMajor f64 -> contiguous (m, n) tensor = column-major n x m (R layout)."""
    import torch
    dst = torch.empty((gm.m, gm.n), dtype=torch.float64, device=gm.t.device)
    L.check(L.lib().dsq_from_gene_major_f64(_t_ptr(gm.t), _t_ptr(dst), gm.n, gm.m, gm.ld, _stream()))
    return dst


def prefitMoments_dev(y, nf, q, a, r, weights=None, useWeights=False, nf_is_vector=False):
    """device flavour of prefitMoments: y / nf / weights GeneMajor, q / a (p, m) and r (p, p)
    contiguous CUDA tensors (= column-major m x p / p x p)."""
    import torch
    n, m, ld = y.n, y.m, y.ld
    p = q.shape[0]
    dev = y.t.device
    f64 = dict(dtype=torch.float64, device=dev)
    pack = torch.empty((4, n), **f64)
    out = {"baseMean": pack[0], "baseVar": pack[1], "allZero": pack[3].view(torch.int32)[:n], "roughDisp": pack[2],
           "beta_init": torch.empty((p, n), **f64)}
    args = L.DsqPrefitArgs(n=n, m=m, p=p, layout=L.DSQ_LAYOUT_GENE_MAJOR, ld=ld, y=_t_ptr(y.t),
                           y_type=L.DSQ_Y_INT32, nf=_t_ptr(nf if nf_is_vector else nf.t),
                           nf_is_vector=int(nf_is_vector), weights=_t_ptr(weights.t) if useWeights else None,
                           useWeights=int(bool(useWeights)), q=_t_ptr(q), a=_t_ptr(a), r=_t_ptr(r))
    o = L.DsqPrefitOut(**{k: _t_ptr(v) for k, v in out.items()})
    L.check(L.lib().dsq_prefit_moments_dev(C.byref(args), C.byref(o), _stream()))
    out["_pack"] = pack
    return out


def linearMu_dev(y, nf, q, a, mu_floor=0.0, nf_is_vector=False):
    """y / nf GeneMajor, q / a (p, m) contiguous CUDA tensors; returns GeneMajor mu"""
    import torch
    n, m, ld = y.n, y.m, y.ld
    mu = torch.zeros((n, ld), dtype=torch.float64, device=y.t.device)
    args = L.DsqPrefitArgs(n=n, m=m, p=q.shape[0], layout=L.DSQ_LAYOUT_GENE_MAJOR, ld=ld, y=_t_ptr(y.t),
                           y_type=L.DSQ_Y_INT32, nf=_t_ptr(nf if nf_is_vector else nf.t),
                           nf_is_vector=int(nf_is_vector), weights=None, useWeights=0, q=_t_ptr(q), a=_t_ptr(a), r=None)
    L.check(L.lib().dsq_linear_mu_dev(C.byref(args), float(mu_floor), _t_ptr(mu), _stream()))
    return GeneMajor(mu, m)


def nbinomLogLike_dev(y, mu, disp, weights=None, useWeights=False):
    import torch
    n, m, ld = y.n, y.m, y.ld
    out = torch.empty(n, dtype=torch.float64, device=y.t.device)
    args = L.DsqLogLikeArgs(n=n, m=m, layout=L.DSQ_LAYOUT_GENE_MAJOR, ld=ld, y=_t_ptr(y.t), y_type=L.DSQ_Y_INT32,
                            mu=_t_ptr(mu.t), disp=_t_ptr(disp), weights=_t_ptr(weights.t) if useWeights else None,
                            useWeights=int(bool(useWeights)))
    L.check(L.lib().dsq_nbinom_loglike_dev(C.byref(args), _t_ptr(out), _stream()))
    return out


def interceptFit_dev(y, nf, alpha, weights=None, useWeights=False, mu_floor=0.0, want_hat=True, nf_is_vector=False):
    """y / nf / weights GeneMajor, alpha a CUDA vector; mu / hat come back GeneMajor, beta / betaSE in `_pack`"""
    import torch
    n, m, ld = y.n, y.m, y.ld
    dev = y.t.device
    pack = torch.empty((2, n), dtype=torch.float64, device=dev)
    mu = torch.zeros((n, ld), dtype=torch.float64, device=dev)
    hat = torch.zeros((n, ld), dtype=torch.float64, device=dev) if want_hat else None
    args = L.DsqInterceptArgs(n=n, m=m, layout=L.DSQ_LAYOUT_GENE_MAJOR, ld=ld, y=_t_ptr(y.t), y_type=L.DSQ_Y_INT32,
                              nf=_t_ptr(nf if nf_is_vector else nf.t), nf_is_vector=int(nf_is_vector),
                              weights=_t_ptr(weights.t) if useWeights else None, useWeights=int(bool(useWeights)),
                              alpha=_t_ptr(alpha), mu_floor=float(mu_floor))
    out = L.DsqInterceptOut(beta_log2=_t_ptr(pack[0]), betaSE=_t_ptr(pack[1]), mu=_t_ptr(mu), hat=_t_ptr(hat))
    L.check(L.lib().dsq_intercept_fit_dev(C.byref(args), C.byref(out), _stream()))
    return {"_pack": pack, "mu": GeneMajor(mu, m), "hat_diagonals": GeneMajor(hat, m) if want_hat else None}


def parametricDispersionFit_dev(means, disps):
    """means / disps: float64 CUDA tensors; returns the two coefficients as a host array"""
    import torch
    out = torch.zeros(3, dtype=torch.float64, device=means.device)      # coefs[2] + status word
    L.check(L.lib().dsq_parametric_dispersion_fit_dev(_t_ptr(means), _t_ptr(disps), means.numel(), _t_ptr(out),
                                                      C.c_void_p(out.data_ptr() + 16), _stream()))
    h = out.cpu()
    st = int(h[2:3].view(torch.int32)[0])
    if st != 0:
        raise RuntimeError(_FIT_ERRORS.get(st, "parametric dispersion fit failed"))
    return h[:2].numpy().copy()


def cooksDistance_dev(y, nf, mu, H, cell_of, p, nf_is_vector=False):
    """y / nf / mu / H GeneMajor; cell_of: host int32 array of design-cell ids.  cooks comes back GeneMajor."""
    import torch
    n, m, ld = y.n, y.m, y.ld
    dev = y.t.device
    cells = np.ascontiguousarray(cell_of, dtype=np.int32)
    ck = torch.zeros((n, ld), dtype=torch.float64, device=dev)
    pack = torch.empty((2, n), dtype=torch.float64, device=dev)
    mx, rd = pack[0], pack[1]
    args = L.DsqCooksArgs(n=n, m=m, p=int(p), layout=L.DSQ_LAYOUT_GENE_MAJOR, ld=ld, y=_t_ptr(y.t),
                          y_type=L.DSQ_Y_INT32, nf=_t_ptr(nf if nf_is_vector else nf.t),
                          nf_is_vector=int(nf_is_vector), mu=_t_ptr(mu.t), H=_t_ptr(H.t), cell_of=_ptr(cells),
                          ncell=int(cells.max()) + 1)
    out = L.DsqCooksOut(cooks=_t_ptr(ck), maxCooks=_t_ptr(mx), robustDisp=_t_ptr(rd))
    L.check(L.lib().dsq_cooks_distance_dev(C.byref(args), C.byref(out), _stream()))
    return {"cooks": GeneMajor(ck, m), "maxCooks": mx, "robustDisp": rd, "_pack": pack}


def replaceOutliers_dev(y, nf, cooks, cooksCutoff, replaceable, trim=0.2, nf_is_vector=False):
    """y / nf / cooks GeneMajor; replaceable: host flags.  The new count matrix comes back GeneMajor int32."""
    import torch
    n, m, ld = y.n, y.m, y.ld
    dev = y.t.device
    rep = np.ascontiguousarray(np.asarray(replaceable).astype(np.int32))
    newc = torch.zeros((n, ld), dtype=torch.int32, device=dev)
    flag = torch.empty(n, dtype=torch.int32, device=dev)
    args = L.DsqReplaceArgs(n=n, m=m, layout=L.DSQ_LAYOUT_GENE_MAJOR, ld=ld, y=_t_ptr(y.t), y_type=L.DSQ_Y_INT32,
                            nf=_t_ptr(nf if nf_is_vector else nf.t), nf_is_vector=int(nf_is_vector),
                            cooks=_t_ptr(cooks.t), cooksCutoff=float(cooksCutoff), trim=float(trim),
                            replaceable=_ptr(rep))
    out = L.DsqReplaceOut(newCounts=_t_ptr(newc), replace=_t_ptr(flag))
    L.check(L.lib().dsq_replace_outliers_dev(C.byref(args), C.byref(out), _stream()))
    return {"counts": GeneMajor(newc, m), "replace": flag}
