"""REFERENCE binding -- test infrastructure only.

ctypes front-end of oracle/_ref/libdeseq2_ref.so: the reference's own src/DESeq2.cpp compiled
against the stand-in headers of oracle/shim/ (oracle/Makefile, target `ref`).  Same function
names, argument order and returned keys as the Rcpp exports (R/RcppExports.R:4,8,12) and as
oracle/oracle.py, so a test can swap one for the other.  The .so is built in the development
container (where /root/reference exists) and travels to the GPU box as a prebuilt file;
`available()` says whether it is there.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_ref", "libdeseq2_ref.so")
_SO_FAST = os.path.join(_HERE, "_ref", "libdeseq2_ref_fast.so")   # special functions in double: timing only
_REF_SRC = "/root/reference/src/DESeq2.cpp"
_lib = None
_fast = False


def build():
    """(re)build when the reference tree is present; otherwise keep whatever prebuilt file exists"""
    if os.path.exists(_REF_SRC):
        subprocess.check_call(["make", "-C", _HERE, "-s", "ref"])
    return _SO if os.path.exists(_SO) else None


def available():
    return os.path.exists(_SO) or (os.path.exists(_REF_SRC) and build() is not None)


def use_fast(flag=True):
    """switch to the timing build (bench.py's cpu_baseline); parity tests use the binary128 build"""
    global _lib, _fast
    if flag and not os.path.exists(_SO_FAST):
        raise FileNotFoundError(_SO_FAST)
    _fast, _lib = bool(flag), None


def lib():
    global _lib
    if _lib is None:
        if not available():
            raise FileNotFoundError(_SO)
        _lib = ctypes.CDLL(_SO_FAST if _fast else _SO)
    return _lib


def _f(a):
    return np.asfortranarray(np.asarray(a, dtype=np.float64))


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def _d(v):
    return ctypes.c_double(float(v))


def _i(v):
    return ctypes.c_int(int(v))


def fitDisp(ySEXP, xSEXP, mu_hatSEXP, log_alphaSEXP, log_alpha_prior_meanSEXP, log_alpha_prior_sigmasqSEXP,
            min_log_alphaSEXP, kappa_0SEXP, tolSEXP, maxitSEXP, usePriorSEXP, weightsSEXP, useWeightsSEXP,
            weightThresholdSEXP, useCRSEXP):
    y, x, mu, w = _f(ySEXP), _f(xSEXP), _f(mu_hatSEXP), _f(weightsSEXP)
    n, m = y.shape
    p = x.shape[1]
    la = np.ascontiguousarray(np.broadcast_to(np.asarray(log_alphaSEXP, float), (n,)))
    pm = np.ascontiguousarray(np.broadcast_to(np.asarray(log_alpha_prior_meanSEXP, float), (n,)))
    keys = ("log_alpha", "iter", "iter_accept", "last_change", "initial_lp", "initial_dlp", "last_lp", "last_dlp",
            "last_d2lp")
    out = {k: np.zeros(n) for k in keys}
    lib().ref_fit_disp(_i(n), _i(m), _i(p), _p(y), _p(x), _p(mu), _p(la), _p(pm), _d(log_alpha_prior_sigmasqSEXP),
                       _d(min_log_alphaSEXP), _d(kappa_0SEXP), _d(tolSEXP), _i(maxitSEXP), _i(bool(usePriorSEXP)),
                       _p(w), _i(bool(useWeightsSEXP)), _d(weightThresholdSEXP), _i(bool(useCRSEXP)),
                       *[_p(out[k]) for k in keys])
    out["iter"] = out["iter"].astype(np.int32)
    out["iter_accept"] = out["iter_accept"].astype(np.int32)
    return out


def fitBeta(ySEXP, xSEXP, nfSEXP, alpha_hatSEXP, contrastSEXP, beta_matSEXP, lambdaSEXP, weightsSEXP,
            useWeightsSEXP, tolSEXP, maxitSEXP, useQRSEXP, minmuSEXP):
    y, x, nf, w, b0 = _f(ySEXP), _f(xSEXP), _f(nfSEXP), _f(weightsSEXP), _f(beta_matSEXP)
    n, m = y.shape
    p = x.shape[1]
    alpha = np.ascontiguousarray(np.broadcast_to(np.asarray(alpha_hatSEXP, float), (n,)))
    contrast = np.ascontiguousarray(contrastSEXP, dtype=np.float64)
    lam = np.ascontiguousarray(lambdaSEXP, dtype=np.float64)
    beta_mat, beta_var = np.zeros((n, p), order="F"), np.zeros((n, p), order="F")
    H = np.zeros((n, m), order="F")
    it, cn, cd, dev = np.zeros(n), np.zeros(n), np.zeros(n), np.zeros(n)
    lib().ref_fit_beta(_i(n), _i(m), _i(p), _p(y), _p(x), _p(nf), _p(alpha), _p(contrast), _p(b0), _p(lam), _p(w),
                       _i(bool(useWeightsSEXP)), _d(tolSEXP), _i(maxitSEXP), _i(bool(useQRSEXP)), _d(minmuSEXP),
                       _p(beta_mat), _p(beta_var), _p(it), _p(H), _p(cn), _p(cd), _p(dev))
    return {"beta_mat": beta_mat, "beta_var_mat": beta_var, "iter": it, "hat_diagonals": H,
            "contrast_num": cn.reshape(n, 1), "contrast_denom": cd.reshape(n, 1), "deviance": dev}


def fitDispGrid(ySEXP, xSEXP, mu_hatSEXP, disp_gridSEXP, log_alpha_prior_meanSEXP, log_alpha_prior_sigmasqSEXP,
                usePriorSEXP, weightsSEXP, useWeightsSEXP, weightThresholdSEXP, useCRSEXP):
    y, x, mu, w = _f(ySEXP), _f(xSEXP), _f(mu_hatSEXP), _f(weightsSEXP)
    n, m = y.shape
    p = x.shape[1]
    grid = np.ascontiguousarray(disp_gridSEXP, dtype=np.float64)
    pm = np.ascontiguousarray(np.broadcast_to(np.asarray(log_alpha_prior_meanSEXP, float), (n,)))
    la = np.zeros(n)
    lib().ref_fit_disp_grid(_i(n), _i(m), _i(p), _p(y), _p(x), _p(mu), _p(grid), _i(grid.size), _p(pm),
                            _d(log_alpha_prior_sigmasqSEXP), _i(bool(usePriorSEXP)), _p(w), _i(bool(useWeightsSEXP)),
                            _d(weightThresholdSEXP), _i(bool(useCRSEXP)), _p(la))
    return {"log_alpha": la}
