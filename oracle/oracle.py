"""ORACLE binding -- test infrastructure only.

ctypes front-end of oracle/_build/libdeseq2_oracle.so (the plain-C restatement of
/root/reference/src/DESeq2.cpp).  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this module; the product package
(deseq2_amd) never does.

Function names, argument names/order and returned dict keys mirror the reference's
Rcpp exports (R/RcppExports.R:4,8,12; return lists at src/DESeq2.cpp:268-276,
458-464, 512).  All matrices are numpy arrays in R orientation (genes x samples,
samples x coefficients); they are passed to C in column-major order like R does.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libdeseq2_oracle.so")
_lib = None


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("deseq2_oracle.c", "orc_nmath.c", "orc_nmath.h", "Makefile")]
    stale = force or not os.path.exists(_SO) or any(
        os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        try:
            _lib = ctypes.CDLL(_SO)
        except OSError:
            build(force=True)
            _lib = ctypes.CDLL(_SO)
        _lib.orc_vec_unary.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long]
        _lib.orc_vec_dnbinom_mu_log.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_long]
        _lib.orc_bd0.restype = ctypes.c_double
        _lib.orc_bd0.argtypes = [ctypes.c_double, ctypes.c_double]
    return _lib


def _f(a):
    """column-major float64 copy (R layout)"""
    return np.asfortranarray(np.asarray(a, dtype=np.float64))


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


UNARY_OPS = {"exp": 0, "log": 1, "log1p": 2, "lgamma": 3, "digamma": 4, "trigamma": 5, "stirlerr": 6, "pnorm_upper2": 9}


def unary(name, x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.empty_like(x)
    lib().orc_vec_unary(UNARY_OPS[name], _p(x), _p(out), x.size)
    return out


def dnbinom_mu_log(x, size, mu):
    x, size, mu = np.broadcast_arrays(np.asarray(x, float), np.asarray(size, float), np.asarray(mu, float))
    x = np.ascontiguousarray(x); size = np.ascontiguousarray(size); mu = np.ascontiguousarray(mu)
    out = np.empty_like(x)
    lib().orc_vec_dnbinom_mu_log(_p(x), _p(size), _p(mu), _p(out), x.size)
    return out


def set_threads(k):
    os.environ["OMP_NUM_THREADS"] = str(int(k))
    try:
        omp = ctypes.CDLL("libgomp.so.1")
        omp.omp_set_num_threads(int(k))
    except OSError:
        pass


def fitDisp(ySEXP, xSEXP, mu_hatSEXP, log_alphaSEXP, log_alpha_prior_meanSEXP,
            log_alpha_prior_sigmasqSEXP, min_log_alphaSEXP, kappa_0SEXP, tolSEXP, maxitSEXP,
            usePriorSEXP, weightsSEXP, useWeightsSEXP, weightThresholdSEXP, useCRSEXP,
            sum_mode=0, cell_mode=1):
    y = _f(ySEXP); x = _f(xSEXP); mu = _f(mu_hatSEXP); w = _f(weightsSEXP)
    n, m = y.shape; p = x.shape[1]
    assert x.shape[0] == m and mu.shape == (n, m) and w.shape == (n, m)
    la = np.ascontiguousarray(np.broadcast_to(np.asarray(log_alphaSEXP, float), (n,)))
    pm = np.ascontiguousarray(np.broadcast_to(np.asarray(log_alpha_prior_meanSEXP, float), (n,)))
    out = {k: np.zeros(n) for k in ("log_alpha", "last_change", "initial_lp", "initial_dlp",
                                     "last_lp", "last_dlp", "last_d2lp")}
    it = np.zeros(n, dtype=np.int32); ita = np.zeros(n, dtype=np.int32)
    rc = lib().orc_fit_disp(
        ctypes.c_int(n), ctypes.c_int(m), ctypes.c_int(p), _p(y), _p(x), _p(mu), _p(la), _p(pm),
        ctypes.c_double(float(log_alpha_prior_sigmasqSEXP)), ctypes.c_double(float(min_log_alphaSEXP)),
        ctypes.c_double(float(kappa_0SEXP)), ctypes.c_double(float(tolSEXP)), ctypes.c_int(int(maxitSEXP)),
        ctypes.c_int(int(bool(usePriorSEXP))), _p(w), ctypes.c_int(int(bool(useWeightsSEXP))),
        ctypes.c_double(float(weightThresholdSEXP)), ctypes.c_int(int(bool(useCRSEXP))),
        _p(out["log_alpha"]), _p(it), _p(ita), _p(out["last_change"]), _p(out["initial_lp"]),
        _p(out["initial_dlp"]), _p(out["last_lp"]), _p(out["last_dlp"]), _p(out["last_d2lp"]),
        ctypes.c_int(sum_mode), ctypes.c_int(int(cell_mode)))
    if rc != 0:
        raise RuntimeError("orc_fit_disp failed: %d" % rc)
    out["iter"] = it; out["iter_accept"] = ita
    return out


def fitBeta(ySEXP, xSEXP, nfSEXP, alpha_hatSEXP, contrastSEXP, beta_matSEXP, lambdaSEXP,
            weightsSEXP, useWeightsSEXP, tolSEXP, maxitSEXP, useQRSEXP, minmuSEXP, sum_mode=0,
            want_mu=False, mu_floor=0.0, want_hat=True, cell_mode=1):
    """cell_mode = 1 (the engine's default): designs with at most 32 distinct rows take the cell-collapsed path
    (see fit_beta_gene_cells); 0 forces the general per-sample path (what a continuous covariate takes)"""
    y = _f(ySEXP); x = _f(xSEXP); nf = _f(nfSEXP); w = _f(weightsSEXP); b0 = _f(beta_matSEXP)
    n, m = y.shape; p = x.shape[1]
    assert x.shape[0] == m and nf.shape == (n, m) and w.shape == (n, m) and b0.shape == (n, p)
    alpha = np.ascontiguousarray(np.broadcast_to(np.asarray(alpha_hatSEXP, float), (n,)))
    contrast = np.ascontiguousarray(contrastSEXP, dtype=np.float64)
    lam = np.ascontiguousarray(lambdaSEXP, dtype=np.float64)
    assert contrast.shape == (p,) and lam.shape == (p,)
    beta_mat = np.zeros((n, p), order="F"); beta_var = np.zeros((n, p), order="F")
    H = np.zeros((n, m), order="F")
    it = np.zeros(n); cn = np.zeros(n); cd = np.zeros(n); dev = np.zeros(n)
    rc = lib().orc_fit_beta(
        ctypes.c_int(n), ctypes.c_int(m), ctypes.c_int(p), _p(y), _p(x), _p(nf), _p(alpha),
        _p(contrast), _p(b0), _p(lam), _p(w), ctypes.c_int(int(bool(useWeightsSEXP))),
        ctypes.c_double(float(tolSEXP)), ctypes.c_int(int(maxitSEXP)), ctypes.c_int(int(bool(useQRSEXP))),
        ctypes.c_double(float(minmuSEXP)),
        _p(beta_mat), _p(beta_var), _p(it), _p(H), _p(cn), _p(cd), _p(dev), ctypes.c_int(sum_mode),
        ctypes.c_int(int(cell_mode)))
    if rc != 0:
        raise RuntimeError("orc_fit_beta failed: %d" % rc)
    out = {"beta_mat": beta_mat, "beta_var_mat": beta_var, "iter": it, "hat_diagonals": H,
           "contrast_num": cn.reshape(n, 1), "contrast_denom": cd.reshape(n, 1), "deviance": dev}
    if want_mu:
        mu = np.zeros((n, m), order="F")
        lib().orc_fitted_mu(ctypes.c_int(n), ctypes.c_int(m), ctypes.c_int(p), _p(x), _p(nf), _p(beta_mat),
                            ctypes.c_double(float(mu_floor)), _p(mu))
        out["mu"] = mu
    return out


def fittedMu(xSEXP, nfSEXP, beta_mat, mu_floor=0.0):
    """mu = max(nf * exp(x beta), mu_floor) with the oracle's exp (R/fitNbinomGLMs.R:180, R/core.R:763)"""
    x = _f(xSEXP); nf = _f(nfSEXP); b = _f(beta_mat)
    n, m = nf.shape
    mu = np.zeros((n, m), order="F")
    lib().orc_fitted_mu(ctypes.c_int(n), ctypes.c_int(m), ctypes.c_int(x.shape[1]), _p(x), _p(nf), _p(b),
                        ctypes.c_double(float(mu_floor)), _p(mu))
    return mu


def fitDispGrid(ySEXP, xSEXP, mu_hatSEXP, disp_gridSEXP, log_alpha_prior_meanSEXP,
                log_alpha_prior_sigmasqSEXP, usePriorSEXP, weightsSEXP, useWeightsSEXP,
                weightThresholdSEXP, useCRSEXP, sum_mode=0, cell_mode=1):
    y = _f(ySEXP); x = _f(xSEXP); mu = _f(mu_hatSEXP); w = _f(weightsSEXP)
    n, m = y.shape; p = x.shape[1]
    grid = np.ascontiguousarray(disp_gridSEXP, dtype=np.float64)
    pm = np.ascontiguousarray(np.broadcast_to(np.asarray(log_alpha_prior_meanSEXP, float), (n,)))
    la = np.zeros(n)
    rc = lib().orc_fit_disp_grid(
        ctypes.c_int(n), ctypes.c_int(m), ctypes.c_int(p), _p(y), _p(x), _p(mu), _p(grid),
        ctypes.c_int(grid.size), _p(pm), ctypes.c_double(float(log_alpha_prior_sigmasqSEXP)),
        ctypes.c_int(int(bool(usePriorSEXP))), _p(w), ctypes.c_int(int(bool(useWeightsSEXP))),
        ctypes.c_double(float(weightThresholdSEXP)), ctypes.c_int(int(bool(useCRSEXP))), _p(la),
        ctypes.c_int(sum_mode), ctypes.c_int(int(cell_mode)))
    if rc != 0:
        raise RuntimeError("orc_fit_disp_grid failed: %d" % rc)
    return {"log_alpha": la}


def nbinomLogLike(counts, mu, disp, weights, useWeights, sum_mode=0):
    """R/core.R:2208-2217"""
    y = _f(counts); mu = _f(mu)
    n, m = y.shape
    w = _f(weights) if useWeights else None
    d = np.ascontiguousarray(np.broadcast_to(np.asarray(disp, float), (n,)))
    out = np.zeros(n)
    rc = lib().orc_nbinom_loglike(ctypes.c_int(n), ctypes.c_int(m), _p(y), _p(mu), _p(d), _p(w),
                                  ctypes.c_int(int(bool(useWeights))), _p(out), ctypes.c_int(sum_mode))
    if rc != 0:
        raise RuntimeError("orc_nbinom_loglike failed")
    return out


def design_qr(x):
    """thin QR of the model matrix as the reference takes it on the host (qr(), qr.Q, qr.R):
    returns Q (m x p), A = X R^-1 (m x p), R (p x p)"""
    x = np.asarray(x, np.float64)
    q, r = np.linalg.qr(x)
    a = x @ np.linalg.inv(r)
    return np.asfortranarray(q), np.asfortranarray(a), np.asfortranarray(r)


def prefitMoments(counts, nf, x, weights=None, useWeights=False, sum_mode=0):
    """baseMean / baseVar / allZero (R/core.R:2138-2146), roughDispEstimate (:2422-2437) and the
    QR least-squares start values of R/fitNbinomGLMs.R:139-145, one pass per gene."""
    y = _f(counts); nf = _f(nf)
    n, m = y.shape
    q, a, r = design_qr(x)
    p = q.shape[1]
    w = _f(weights) if useWeights else None
    bm = np.zeros(n); bv = np.zeros(n); az = np.zeros(n, dtype=np.int32); rd = np.zeros(n)
    b0 = np.zeros((n, p), order="F")
    rc = lib().orc_prefit_moments(ctypes.c_int(n), ctypes.c_int(m), ctypes.c_int(p), _p(y), _p(nf), _p(w),
                                  ctypes.c_int(int(bool(useWeights))), _p(q), _p(a), _p(r), _p(bm), _p(bv),
                                  _p(az), _p(rd), _p(b0), ctypes.c_int(sum_mode))
    if rc != 0:
        raise RuntimeError("orc_prefit_moments failed")
    return {"baseMean": bm, "baseVar": bv, "allZero": az.astype(bool), "roughDisp": rd, "beta_init": b0}


def linearMu(counts, nf, x, mu_floor=0.0, sum_mode=0):
    """linearModelMuNormalized (R/core.R:2454-2471), optionally floored (R/core.R:763)"""
    y = _f(counts); nf = _f(nf)
    n, m = y.shape
    q, a, r = design_qr(x)
    q = _f(q); a = _f(a)
    mu = np.zeros((n, m), order="F")
    rc = lib().orc_linear_mu(ctypes.c_int(n), ctypes.c_int(m), ctypes.c_int(q.shape[1]), _p(y), _p(nf), _p(q), _p(a),
                             ctypes.c_double(float(mu_floor)), _p(mu), ctypes.c_int(sum_mode))
    if rc != 0:
        raise RuntimeError("orc_linear_mu failed: %d" % rc)
    return mu


def parametricDispersionFit(means, disps):
    """R/core.R:2166-2190; raises RuntimeError with the reference's messages on failure"""
    means = np.ascontiguousarray(means, dtype=np.float64); disps = np.ascontiguousarray(disps, dtype=np.float64)
    coefs = np.zeros(2); st = ctypes.c_int(0)
    lib().orc_parametric_dispersion_fit(ctypes.c_long(means.size), _p(means), _p(disps), _p(coefs), ctypes.byref(st))
    if st.value == 1:
        raise RuntimeError("parametric dispersion fit failed")
    if st.value == 2:
        raise RuntimeError("dispersion fit did not converge")
    return coefs


def trimmedMeanFit(disps, minDisp=1e-8):
    """fitType = "mean", R/core.R:894-899: the trimmed mean of the gene-wise estimates above 10 minDisp"""
    disps = np.ascontiguousarray(disps, dtype=np.float64)
    out = ctypes.c_double(0.0)
    lib().orc_trimmed_mean_fit.restype = ctypes.c_long
    kept = lib().orc_trimmed_mean_fit(ctypes.c_long(disps.size), _p(disps), ctypes.c_double(float(minDisp)), ctypes.byref(out))
    if kept == 0:
        raise RuntimeError("no gene-wise dispersion estimate above 10 minDisp")
    return float(out.value)


def cell_index(x):
    """cells of identical model-matrix rows (nOrMoreInCell, R/core.R:2366-2371): cell id per sample"""
    _, inv = np.unique(np.asarray(x, np.float64), axis=0, return_inverse=True)
    return np.ascontiguousarray(inv.reshape(-1), dtype=np.int32)


def cooksDistance(counts, nf, mu, H, x, sum_mode=0):
    """calculateCooksDistance + robustMethodOfMomentsDisp + recordMaxCooks (R/core.R:2333-2359, 2277-2331)"""
    y = _f(counts); nf = _f(nf); mu = _f(mu); H = _f(H)
    n, m = y.shape
    p = np.asarray(x).shape[1]
    cells = cell_index(x)
    ck = np.zeros((n, m), order="F"); mx = np.zeros(n); rd = np.zeros(n)
    lib().orc_cooks_distance(ctypes.c_int(n), ctypes.c_int(m), ctypes.c_int(p), _p(y), _p(nf), _p(mu), _p(H),
                             _p(cells), ctypes.c_int(int(cells.max()) + 1), _p(ck), _p(mx), _p(rd), ctypes.c_int(sum_mode))
    return {"cooks": ck, "maxCooks": mx, "robustDisp": rd}


def replaceOutliers(counts, nf, cooks, cooksCutoff, replaceable, trim=0.2, sum_mode=0):
    """replaceOutliers (R/core.R:2069-2115): new count matrix and the per-gene `replace` flag"""
    y = _f(counts); nf = _f(nf); ck = _f(cooks)
    n, m = y.shape
    rep = np.ascontiguousarray(np.asarray(replaceable).astype(np.int32))
    newc = np.zeros((n, m), dtype=np.int32, order="F"); flag = np.zeros(n, dtype=np.int32)
    lib().orc_replace_outliers(ctypes.c_int(n), ctypes.c_int(m), _p(y), _p(nf), _p(ck), ctypes.c_double(float(cooksCutoff)),
                               _p(rep), ctypes.c_double(float(trim)), _p(newc), _p(flag), ctypes.c_int(sum_mode))
    return {"counts": newc, "replace": flag.astype(bool)}


def interceptFit(counts, nf, alpha, weights=None, useWeights=False, mu_floor=0.0, want_hat=True, sum_mode=0):
    """closed form of R/fitNbinomGLMs.R:99-137 (design ~ 1): betaMatrix (log2), betaSE, mu, hat_diagonals"""
    y = _f(counts); nf = _f(nf)
    n, m = y.shape
    w = _f(weights) if useWeights else None
    a = np.ascontiguousarray(np.broadcast_to(np.asarray(alpha, float), (n,)))
    b, se = np.zeros(n), np.zeros(n)
    mu = np.zeros((n, m), order="F")
    hat = np.zeros((n, m), order="F") if want_hat else None
    lib().orc_intercept_fit(ctypes.c_int(n), ctypes.c_int(m), _p(y), _p(nf), _p(w), ctypes.c_int(int(bool(useWeights))),
                            _p(a), ctypes.c_double(float(mu_floor)), _p(b), _p(se), _p(mu), _p(hat),
                            ctypes.c_int(sum_mode))
    return {"beta": b, "betaSE": se, "mu": mu, "hat_diagonals": hat}


def optimRows(counts, x, nf, alpha, lam, weights, useWeights, beta_start, minmu=0.5, sum_mode=0):
    """fitNbinomGLMsOptim (R/fitNbinomGLMs.R:340-407) on the given rows: damped Fisher scoring on the penalised NB
    posterior over [-30, 30]^p.  lam: prior precisions on the log2 scale; beta_start: log2-scale start values."""
    y = _f(counts); x = _f(x); nf = _f(nf)
    n, m = y.shape; p = x.shape[1]
    w = _f(weights) if useWeights else None
    a = np.ascontiguousarray(np.broadcast_to(np.asarray(alpha, float), (n,)))
    lamnat = np.ascontiguousarray(np.asarray(lam, float) / np.log(2) ** 2)
    b0 = _f(np.asarray(beta_start, float).reshape(n, p))
    beta = np.zeros((n, p), order="F"); se = np.zeros((n, p), order="F")
    conv = np.zeros(n, dtype=np.int32); mu = np.zeros((n, m), order="F"); ll = np.zeros(n)
    rc = lib().orc_optim_rows(ctypes.c_int(n), ctypes.c_int(m), ctypes.c_int(p), _p(y), _p(x), _p(nf), _p(a), _p(lamnat),
                              _p(w), ctypes.c_int(int(bool(useWeights))), _p(b0), ctypes.c_double(float(minmu)),
                              _p(beta), _p(se), _p(conv), _p(mu), _p(ll), ctypes.c_int(sum_mode))
    if rc != 0:
        raise RuntimeError("orc_optim_rows failed: %d" % rc)
    return {"beta": beta, "betaSE": se, "conv": conv.astype(bool), "mu": mu, "logLike": ll}
