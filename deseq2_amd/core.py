"""Host-side mirror of the reference's callers of the native boundary (R/core.R,
R/fitNbinomGLMs.R, R/wrappers.R): same function names, same argument meaning and defaults,
same decision rules, operating on a minimal DESeqDataSet stand-in.  The three native
routines go to the MI355X engine (engine.py); the all-gene "global" steps between them
(dispersion trend, prior variance) stay on the host exactly as in DESeqParallel
(R/parallel.R:27-28) -- they exchange only n-vectors.

Cook's distances, outlier replacement and the refit loop are mirrored too; their
O(n m) parts run in the engine.

Not mirrored (out of the hot-path scope): size-factor estimation,
local/glmGamPoi dispersion fits, results(), lfcShrink().
"""
import numpy as np
from scipy import special as sps

from .engine import HostEngine

LOG2E = np.log2(np.e)

# facts about a model matrix (rank, design cells) are asked for several times per DESeq(); they depend
# on the m x p matrix only, so they are memoised on its bytes
_DESIGN_CACHE = {}


def _design_fact(kind, x, fn):
    x = np.ascontiguousarray(x, dtype=np.float64)
    key = (kind, x.shape, x.tobytes())
    v = _DESIGN_CACHE.get(key)
    if v is None:
        if len(_DESIGN_CACHE) > 256:
            _DESIGN_CACHE.clear()
        v = _DESIGN_CACHE[key] = fn(x)
    return v


def _rank(x):
    return _design_fact("rank", x, lambda a: int(np.linalg.matrix_rank(a)))


def _cells(x):
    """(cell id per sample, size of that sample's cell): samples with identical model-matrix rows"""
    def f(a):
        _, inv, cnt = np.unique(a, axis=0, return_inverse=True, return_counts=True)
        inv = inv.reshape(-1)
        return inv, cnt[inv]
    return _design_fact("cells", x, f)


class DESeqDataSet:
    """counts (n x m int), model matrix x (m x p), size factors (m) or normalization
    factors (n x m), optional observation weights (n x m).  `mcols` holds per-gene vectors
    (R: mcols(object)), `assays` the n x m engine handles (mu, H)."""

    def __init__(self, counts, x, sizeFactors=None, normalizationFactors=None, weights=None, engine=None):
        counts = np.asarray(counts)
        if counts.ndim != 2 or (counts < 0).any():
            raise ValueError("counts must be a non-negative integer matrix")          # R/AllClasses.R:9-20
        self.n, self.m = counts.shape
        self.x = np.asarray(x, dtype=np.float64)
        if self.x.shape[0] != self.m:
            raise ValueError("model matrix rows must match samples")
        self.engine = engine if engine is not None else HostEngine()
        if normalizationFactors is not None:
            nf = np.asarray(normalizationFactors, np.float64)
            self.sizeFactors = None
        else:
            sf = np.ones(self.m) if sizeFactors is None else np.asarray(sizeFactors, np.float64)
            nf = np.broadcast_to(sf[None, :], counts.shape)                          # getSizeOrNormFactors :2221
            self.sizeFactors = sf
        self.counts_host = counts.astype(np.int32)
        E = self.engine
        self.y = E.counts(self.counts_host)
        self.nf = E.matrix(nf)
        self.has_weights = weights is not None
        # assays(object)[["weights"]] as given (an engine handle); normalised by getAndCheckWeights
        self.weights_h = None if weights is None else E.matrix(np.asarray(weights, np.float64))
        self.xh = E.design(self.x)
        self.mcols = {}
        self.assays = {}
        self.attrs = {}
        self.dispersionFunction = None

    @classmethod
    def from_device(cls, engine, counts_r, nf_r, x, weights=None, sizeFactors=None, weights_r=None):
        """Build from matrices ALREADY RESIDENT in HBM in R layout: `counts_r` (int32), `nf_r` and the optional
        `weights_r` (float64) are contiguous (m, n) torch tensors, i.e. column-major n x m exactly as
        R holds them.  Converts to the engine's gene-major layout on the device (no host copy).  (`weights`, a
        host array, is uploaded when no `weights_r` is given.)  `sizeFactors`: pass them only when `nf_r` is the matrix
        R builds FROM them (`getSizeOrNormFactors`, R/core.R:2221-2227: no normalizationFactors) -- the chain may
        then read the m-vector instead of the matrix."""
        self = cls.__new__(cls)
        self.m, self.n = counts_r.shape
        self.x = np.asarray(x, dtype=np.float64)
        self.engine = engine
        self.counts_host = None
        self.sizeFactors = None if sizeFactors is None else np.asarray(sizeFactors, np.float64)
        self.y = engine.native.to_gene_major(counts_r)
        # with size factors the chain can run on the m-vector (nf[i, j] = s_j for every gene): the n x m matrix R
        # passes is then converted only if some step asks for it
        if self.sizeFactors is not None:
            self._nf, self._nf_src = None, nf_r
        else:
            self.nf = engine.native.to_gene_major(nf_r)
        self.has_weights = weights is not None or weights_r is not None
        if weights_r is not None:
            self.weights_h = engine.native.to_gene_major(weights_r)
        else:
            self.weights_h = None if weights is None else engine.matrix(np.asarray(weights, np.float64))
        self.xh = engine.design(self.x)
        self.mcols, self.assays, self.attrs = {}, {}, {}
        self.dispersionFunction = None
        return self

    def subset(self, idx, counts_handle=None):
        """object[idx, ] (optionally over another count matrix handle with the same rows as self)"""
        E = self.engine
        sub = DESeqDataSet.__new__(DESeqDataSet)
        idx = np.asarray(idx)
        sub.n, sub.m, sub.x, sub.engine = int(idx.size), self.m, self.x, E
        sub.sizeFactors = self.sizeFactors
        sub.counts_host = None
        sub.y = E.take_rows(self.y if counts_handle is None else counts_handle, idx)
        sub.nf = E.take_rows(self.nf, idx)
        sub.has_weights = self.has_weights
        sub.weights_h = E.take_rows(self.weights_h, idx)
        sub.xh = self.xh
        sub.mcols, sub.assays, sub.attrs = {}, {}, {}
        if "weightsOK" in self.attrs:          # attr(object, "weightsOK"): the rank checks ran on the parent
            sub.attrs["weightsOK"] = True
        sub.dispersionFunction = None if self.dispersionFunction is None else dict(self.dispersionFunction)
        return sub

    @property
    def p(self):
        return self.x.shape[1]

    # normalization factors in the engine's gene-major layout; from_device() defers the conversion (see there)
    _nf = None
    _nf_src = None

    @property
    def nf(self):
        if self._nf is None and self._nf_src is not None:
            self._nf, self._nf_src = self.engine.native.to_gene_major(self._nf_src), None
        return self._nf

    @nf.setter
    def nf(self, v):
        self._nf, self._nf_src = v, None


# ------------------------------------------------------------------ R/wrappers.R
def _na_guard(fname, **args):
    """R/wrappers.R:31-34,110-113: error naming the arguments that contain NA"""
    bad = [k for k, v in args.items() if v is not None and isinstance(v, np.ndarray) and np.isnan(v).any()]
    if bad:
        raise ValueError("in call to %s, the following arguments contain NA: %s" % (fname, ", ".join(bad)))


def fitDispGridWrapper(E, y, x, mu, logAlphaPriorMean, logAlphaPriorSigmaSq, usePrior, weights, useWeights,
                       weightThreshold, useCR, ncol_y):
    """R/wrappers.R:63-82"""
    _na_guard("fitDispGridWrapper", logAlphaPriorMean=np.asarray(logAlphaPriorMean, float))
    minLogAlpha = np.log(1e-8)
    maxLogAlpha = np.log(max(10, ncol_y))
    dispGrid = np.linspace(minLogAlpha, maxLogAlpha, 20)
    la = E.fit_disp_grid(y, x, mu, dispGrid, logAlphaPriorMean, logAlphaPriorSigmaSq, usePrior, weights,
                         useWeights, weightThreshold, useCR)["log_alpha"]
    return E.vexp(la)


# ------------------------------------------------------------------ weights
def getAndCheckWeights(dds, weightThreshold=1e-2, modelMatrix=None):
    """R/core.R:2697-2751: row-max normalisation; once per analysis the per-gene rank checks, whose failures are
    flagged in mcols weightsFail and treated as all-zero rows (:2736-2747).  Returns (engine handle of the
    normalised weights, useWeights); the handle is cached on the object."""
    E = dds.engine
    if not dds.has_weights:
        return None, False
    w = dds.attrs.get("weights_norm")
    if w is None or dds.attrs.get("weights_norm_of") is not dds.weights_h:
        if E.any_negative(dds.weights_h):
            raise ValueError("all(weights >= 0) is not TRUE")
        w = E.row_max_normalize(dds.weights_h)                                        # :2702
        dds.attrs["weights_norm"], dds.attrs["weights_norm_of"] = w, dds.weights_h
        dds.attrs.pop("weights_floor", None)
    if "weightsOK" not in dds.attrs:
        x = dds.x if modelMatrix is None else np.asarray(modelMatrix, np.float64)
        ok = E.weights_ok(w, x, weightThreshold, _rank(x) == x.shape[1])              # :2706-2734
        if not ok.all():
            dds.mcols["weightsFail"] = ~ok
            if "allZero" in dds.mcols:
                dds.mcols["allZero"] = dds.mcols["allZero"] | ~ok                     # :2737
        dds.attrs["weightsOK"] = True
    return w, True


def _floored_weights(dds, w):
    """weights <- pmax(weights, 1e-6)   (R/core.R:702), cached next to the normalised handle"""
    f = dds.attrs.get("weights_floor")
    if f is None:
        f = dds.attrs["weights_floor"] = dds.engine.clamp_min(w, 1e-6)
    return f


def getBaseMeansAndVariances(dds):
    """R/core.R:2138-2157"""
    E = dds.engine
    # one pass also yields the rough dispersion and the IRLS start values used right after
    # (roughDispEstimate :2422, fitNbinomGLMs.R:139-145); they are cached on the object
    if dds.attrs.get("prefit_for") is dds.y and "prefit" in dds.attrs:   # same count handle: nothing changed
        pf = dds.attrs["prefit"]
    else:
        w = dds.weights_h if dds.has_weights else None
        pf = E.prefit(dds.y, dds.nf, dds.x, w)
        dds.attrs["prefit"], dds.attrs["prefit_for"] = pf, dds.y
    dds.mcols["baseMean"], dds.mcols["baseVar"], dds.mcols["allZero"] = pf["baseMean"], pf["baseVar"], pf["allZero"]
    return dds


def xim_size_factors(sf):
    """momentsDispEstimate's xim = mean(1 / sizeFactors) (R/core.R:2440-2444), summed in sample order (R's mean() is a
    sequential sum too): ONE definition for the call-by-call chain, the fused chain and the library's own host entry
    (csrc/deseq_host.hip), so that all three start the dispersion search from the same bits"""
    sf = np.asarray(sf, np.float64)
    return float(np.cumsum(1.0 / sf)[-1] / sf.size)


def modelMatrixGroups(x):
    """R/core.R:2450-2452"""
    return _cells(x)[0]


# ------------------------------------------------------------------ R/fitNbinomGLMs.R
def fitNbinomGLMsOptim(E, y, nf, x, lam, rowsForOptim, rowStable, alpha_hat, weights, useWeights, betaMatrix,
                       betaSE, betaConv, beta_mat_init, logLike, minmu=0.5):
    """R/fitNbinomGLMs.R:340-407: the rows the IRLS did not fit are re-fitted by maximising the penalised NB log
    posterior over beta in [-30, 30]^p.  The reference loops over them in R with optim(method = "L-BFGS-B"); the
    engine fits them in one launch (damped Fisher scoring on the same objective and box, E.optim_rows), everything
    after the optimum -- mu, betaSE, logLike -- as :382-400."""
    rows = np.asarray(rowsForOptim)
    large = 30.0
    start = np.empty((rows.size, x.shape[1]))
    for r, row in enumerate(rows):
        if rowStable[row] and (np.abs(betaMatrix[row]) < large).all():
            start[r] = betaMatrix[row]                                             # :351-352
        else:
            start[r] = np.asarray(beta_mat_init[row], float)                       # :354 (the natural-log start, as in R)
    o = E.optim_rows(E.take_rows(y, rows), E.take_rows(nf, rows), x, np.asarray(alpha_hat)[rows], lam,
                     E.take_rows(weights, rows) if useWeights else None, useWeights, start, minmu)
    betaConv[rows[np.asarray(o["conv"], bool)]] = True                             # :378-380
    betaMatrix[rows] = o["beta"]                                                   # :382
    betaSE[rows] = o["betaSE"]                                                     # :397
    logLike[rows] = o["logLike"]                                                   # :398-399
    return betaMatrix, betaSE, betaConv, rows, np.asarray(o["mu"]), logLike


def _host_vector(v):
    """an engine may hand back an n-vector that is still on the device (engine.LaunchedVector)"""
    return v.host() if hasattr(v, "host") else v


class PendingFit:
    """fitNbinomGLMs(defer = TRUE): the fit has been launched; `mu` (and `hat_diagonals`) are engine handles
    usable at once, finish() runs the host half of R/fitNbinomGLMs.R:184-235 (row checks, optim fallback)."""

    def __init__(self, mu, hat_diagonals, finish):
        self.mu, self.hat_diagonals, self.finish = mu, hat_diagonals, finish


def fitNbinomGLMs(dds, rows=None, modelMatrix=None, alpha_hat=None, lam=None, betaTol=1e-8, maxit=100,
                  useOptim=True, useQR=True, minmu=0.5, weights=None, useWeights=False, mu_floor=0.0,
                  want_hat=True, forceOptim=False, want_loglike=False, defer=False):
    """R/fitNbinomGLMs.R:29-236.  Rows the IRLS does not fit (`rowsForOptim`, :203-211) go through the
    reference's L-BFGS-B fallback on the host (fitNbinomGLMsOptim, :213-227), as in R.  logLike
    (:182) is computed only when the caller reads it (want_loglike)."""
    E = dds.engine
    x = dds.x if modelMatrix is None else np.asarray(modelMatrix, np.float64)
    xh = dds.xh if modelMatrix is None else E.design(x)
    y, nf = dds.y, dds.nf
    if rows is not None:
        y, nf = E.take_rows(y, rows), E.take_rows(nf, rows)
        weights = E.take_rows(weights, rows) if weights is not None else None
    n = E.nrow(y)
    p = x.shape[1]
    if alpha_hat is None:
        alpha_hat = dds.mcols["dispersion"] if rows is None else dds.mcols["dispersion"][rows]
    alpha_hat = np.asarray(alpha_hat, np.float64)
    if alpha_hat.shape[0] != n:
        raise ValueError("alpha_hat needs to be the same length as nrows(object)")
    if lam is None:
        lam = np.full(p, 1e-6)                                                     # :73
    lam = np.asarray(lam, np.float64)
    if not (np.abs(x).sum(axis=0) > 0).all():
        raise ValueError("all(colSums(abs(modelMatrix)) > 0) is not TRUE")          # :45
    # intercept-only model with the wide prior: closed form, no native call (:99-137)
    if p == 1 and (x == 1).all() and (lam <= 1e-6).all():
        # one engine pass (the reference does this in vectorised R): beta = log2 of the [weighted] mean normalized
        # count, mu = nf 2^beta, betaSE / hat from w = [weights] / (1/mu + alpha).  The mu handed back is floored at
        # mu_floor (the gene-wise dispersion fit reads fitMu[fitMu < minmu] <- minmu, R/core.R:763); betaSE, hat and
        # logLike are of the unfloored fit, as in R.
        r = E.intercept_fit(y, nf, alpha_hat, weights, useWeights, mu_floor=0.0, want_hat=want_hat)
        mu_h = r["mu"]
        res = {"betaConv": np.ones(n, bool), "betaMatrix": r["beta"][:, None], "betaSE": r["betaSE"][:, None],
                "mu": E.clamp_min(mu_h, mu_floor) if mu_floor > 0 else mu_h, "betaIter": np.ones(n), "modelMatrix": x,
                "nterms": 1, "hat_diagonals": r["hat_diagonals"], "deviance_native": None,
                "rowsForOptim": np.array([], int), "beta_natlog": r["beta"][:, None] / LOG2E, "optimRows": None,
                "logLike": (_host_vector(E.nbinom_loglike(y, mu_h, alpha_hat, weights, useWeights))
                            if want_loglike else None)}
        return PendingFit(res["mu"], res["hat_diagonals"], lambda: res) if defer else res
    # initial betas by QR least squares when full rank (:139-155)
    if _rank(x) == p:
        if rows is None and modelMatrix is None and "prefit" in dds.attrs:
            beta_mat = dds.attrs["prefit"]["beta_init"]
        else:
            beta_mat = E.prefit(y, nf, x)["beta_init"]
    else:
        beta0 = np.zeros((n, p))
        bm = E.prefit(y, nf, np.ones((x.shape[0], 1)))["baseMean"]
        if (x[:, 0] == 1).all():
            beta0[:, 0] = E.vlog(bm)
        else:
            beta0[:] = 1.0
        beta_mat = beta0
    lambdaNatLogScale = lam / np.log(2) ** 2                                       # :162
    contrast = np.r_[1.0, np.zeros(p - 1)]                                         # R/wrappers.R:105-108
    _na_guard("fitBeta", alpha_hatSEXP=alpha_hat, lambdaSEXP=lambdaNatLogScale)
    betaRes = E.fit_beta(y, xh, nf, alpha_hat, contrast, beta_mat, lambdaNatLogScale, weights, useWeights,
                         betaTol, maxit, useQR, minmu, want_mu=True, mu_floor=mu_floor, want_hat=want_hat)
    mu0 = betaRes["mu"]                                                            # :180
    # launched before the host reads anything back, so the row checks below overlap it (:182)
    logLike0 = E.nbinom_loglike(y, mu0, alpha_hat, weights, useWeights) if want_loglike else None

    def finish():
        mu = mu0
        rowStable = ~np.isnan(betaRes["beta_mat"]).any(axis=1)                     # :185
        rowVarPositive = ~(betaRes["beta_var_mat"] <= 0).any(axis=1)               # :188
        betaConv = betaRes["iter"] < maxit                                         # :191
        betaMatrix = LOG2E * betaRes["beta_mat"]                                   # :194
        betaSE = LOG2E * np.sqrt(np.maximum(betaRes["beta_var_mat"], 0))           # :198
        if useOptim:
            rowsForOptim = np.where(~betaConv | ~rowStable | ~rowVarPositive)[0]   # :203-207
        else:
            rowsForOptim = np.where(~rowStable | ~rowVarPositive)[0]
        if forceOptim:
            rowsForOptim = np.arange(n)                                            # :209-211
        logLike = _host_vector(logLike0)
        optimRows = None
        if len(rowsForOptim) > 0:                                                  # :213-227
            ll = np.full(n, np.nan)
            b0 = E.to_numpy_np(beta_mat) if hasattr(E, "to_numpy_np") else beta_mat
            betaMatrix, betaSE, betaConv, optimRows, optimMu, ll = fitNbinomGLMsOptim(
                E, y, nf, x, lam, rowsForOptim, rowStable, alpha_hat, weights, useWeights, betaMatrix.copy(),
                betaSE.copy(), betaConv.copy(), b0, ll, minmu=minmu)
            mu = E.put_rows(mu, optimRows, np.maximum(optimMu, mu_floor))          # mu[row,] <- mu_row  (:386)
            if logLike is not None:
                logLike = np.array(logLike, copy=True)
                logLike[optimRows] = ll[optimRows]                                 # :399
        return {"betaConv": betaConv, "betaMatrix": betaMatrix, "betaSE": betaSE, "mu": mu, "logLike": logLike,
                "optimRows": optimRows,
                "betaIter": betaRes["iter"], "modelMatrix": x, "nterms": p,
                "hat_diagonals": betaRes.get("hat_diagonals"), "deviance_native": betaRes["deviance"],
                "rowsForOptim": rowsForOptim, "beta_natlog": betaRes["beta_mat"]}
    return PendingFit(mu0, betaRes.get("hat_diagonals") if want_hat else None, finish) if defer else finish()


# ------------------------------------------------------------------ dispersions
def estimateDispersionsGeneEst(dds, minDisp=1e-8, kappa_0=1.0, dispTol=1e-6, maxit=100, useCR=True,
                               weightThreshold=1e-2, modelMatrix=None, niter=1, linearMu=None, minmu=0.5,
                               alphaInit=None):
    """R/core.R:657-860"""
    if np.log(minDisp / 10) <= -30:
        raise ValueError("for computational stability, log(minDisp/10) should be above -30")
    E = dds.engine
    x = dds.x if modelMatrix is None else np.asarray(modelMatrix, np.float64)
    if _rank(x) < x.shape[1]:
        raise ValueError("the model matrix is not full rank")                       # checkFullRank :2624
    if x.shape[0] == x.shape[1]:
        raise ValueError("the number of samples and the number of model coefficients are equal")
    if modelMatrix is not None:
        dds.attrs["geneEstModelMatrix"] = x           # attr(object, "dispModelMatrix"): estimateDispersionsMAP's default (R/core.R:850, 1003)
    if not (int(niter) == niter and niter > 0):
        raise ValueError("length(niter) == 1 & niter > 0 is not TRUE")              # :730
    getBaseMeansAndVariances(dds)
    w_norm, useWeights = getAndCheckWeights(dds, weightThreshold, modelMatrix=modelMatrix)   # :698
    weights_glm = w_norm                                            # fitNbinomGLMs re-reads them unfloored (:77)
    weights = _floored_weights(dds, w_norm) if useWeights else None                 # :702
    nz = ~dds.mcols["allZero"]
    if not nz.all():
        raise ValueError("all-zero rows must be removed before fitting (the engine fits objectNZ)")
    m, n = dds.m, dds.n
    xh = dds.xh if modelMatrix is None else E.design(x)
    if alphaInit is None:
        if modelMatrix is None:
            roughDisp = dds.attrs["prefit"]["roughDisp"]                            # :713
        else:
            roughDisp = E.prefit(dds.y, dds.nf, x, dds.weights_h if dds.has_weights else None)["roughDisp"]
        bm, bv = dds.mcols["baseMean"], dds.mcols["baseVar"]
        xim = xim_size_factors(dds.sizeFactors) if dds.sizeFactors is not None else E.xim(dds.nf)
        momentsDisp = (bv - xim * bm) / (bm * bm)                                   # :2439-2448
        alpha_hat = np.minimum(roughDisp, momentsDisp)
    else:
        alpha_hat = np.broadcast_to(np.asarray(alphaInit, float), (n,)).copy()
    maxDisp = max(10, m)
    alpha_hat = alpha_init = np.minimum(np.maximum(minDisp, alpha_hat), maxDisp)    # :727-728
    alpha_hat_new = alpha_hat.copy()
    if linearMu is None:
        linearMu = (len(np.unique(modelMatrixGroups(x))) == x.shape[1]) and not useWeights   # :735-742

    def fit_disp(y, mu, la, w):
        return E.fit_disp(y, xh, mu, la, la, 1.0, np.log(minDisp / 10), kappa_0, dispTol, maxit, False,
                          w, useWeights, weightThreshold, useCR)                    # :771-782

    fitidx = np.ones(n, bool)
    mu = None
    dispIter = np.zeros(n, dtype=np.int32)
    last_lp = initial_lp = None
    for it in range(int(niter)):                                                    # :751
        every = bool(fitidx.all())
        idx = None if every else np.where(fitidx)[0]
        y_f = dds.y if every else E.take_rows(dds.y, idx)
        la_f = E.vlog(alpha_hat if every else alpha_hat[idx])
        # `weightsSEXP = weights` (:778) is NOT subset by fitidx: from the second pass on fitDisp pairs the rows
        # of y[fitidx, ] with the FIRST sum(fitidx) rows of the weight matrix.  Mirrored as written.
        w_f = weights if (every or weights is None) else E.take_rows(weights, np.arange(idx.size))
        if not linearMu:
            # The GLM fit is launched, the dispersion search is launched on its mu right behind it, and only then
            # does the host half of fitNbinomGLMs (row checks, optim fallback) run -- overlapping the search.  Genes
            # are independent: rows whose mu the optim fallback replaces (:386) get their search redone below.
            pend = fitNbinomGLMs(dds, rows=idx, alpha_hat=alpha_hat if every else alpha_hat[idx], modelMatrix=modelMatrix,
                                 weights=weights_glm, useWeights=useWeights, mu_floor=minmu, want_hat=False,
                                 defer=True)                                        # :755-757 (the IRLS keeps ITS default
                                                                                    # minmu = 0.5; `minmu` is the floor of :763)
            dispRes = fit_disp(y_f, pend.mu, la_f, w_f)
            fit = pend.finish()
            fitMu = fit["mu"]                                                       # clamped at minmu (:763)
            if fit["optimRows"] is not None and len(fit["optimRows"]) > 0:
                oi = np.asarray(fit["optimRows"])
                sub = fit_disp(E.take_rows(y_f, oi), E.take_rows(fitMu, oi), la_f[oi], E.take_rows(w_f, oi))
                dispRes = {k: np.array(dispRes[k], copy=True) for k in sub.keys()}
                for k in dispRes:
                    dispRes[k][oi] = sub[k]
        else:
            nf_f = dds.nf if every else E.take_rows(dds.nf, idx)
            fitMu = E.clamp_min(E.linear_mu(y_f, nf_f, xh), minmu)                  # :760,763
            dispRes = fit_disp(y_f, fitMu, la_f, w_f)
        mu = fitMu if every else E.set_rows(mu, idx, fitMu)                         # :764
        new = np.minimum(E.vexp(dispRes["log_alpha"]), maxDisp)                     # :785
        if every:
            dispIter = np.asarray(dispRes["iter"]).copy()
            alpha_hat_new = new
        else:
            dispIter[idx] = dispRes["iter"]
            alpha_hat_new = alpha_hat_new.copy()
            alpha_hat_new[idx] = new
        last_lp, initial_lp = dispRes["last_lp"], dispRes["initial_lp"]
        with np.errstate(invalid="ignore"):
            fitidx = np.abs(E.vlog(alpha_hat_new) - E.vlog(alpha_hat)) > .05        # :822
        fitidx[np.isnan(fitidx)] = False
        alpha_hat = alpha_hat_new
        if fitidx.sum() == 0:
            break
    dispGeneEst = alpha_hat.copy()
    if niter == 1:
        noIncrease = last_lp < initial_lp + np.abs(initial_lp) / 1e6               # :828
        dispGeneEst[noIncrease] = alpha_init[noIncrease]
    dispGeneEstConv = (dispIter < maxit) & ~(dispIter == 1)                         # :832
    refitDisp = ~dispGeneEstConv & (dispGeneEst > minDisp * 10)                     # :835
    if refitDisp.sum() > 0:
        idx = np.where(refitDisp)[0]
        dispGrid = fitDispGridWrapper(E, E.take_rows(dds.y, idx), xh, E.take_rows(mu, idx),
                                      np.zeros(idx.size), 1.0, False, E.take_rows(weights, idx), useWeights,
                                      weightThreshold, useCR, m)                    # :837-846
        dispGeneEst[refitDisp] = dispGrid
    dispGeneEst = np.minimum(np.maximum(dispGeneEst, minDisp), maxDisp)             # :848
    dds.mcols["dispGeneEst"] = dispGeneEst
    dds.mcols["dispGeneIter"] = dispIter
    dds.assays["mu"] = mu
    dds.attrs["disp_weights"] = weights
    dds.attrs["useWeights"] = useWeights
    return dds


def parametricDispersionFit(means, disps):
    """R/core.R:2166-2190 in numpy -- kept as an independent cross-check of the engine's
    `parametric_fit` (tests); the pipeline itself calls the engine.  disp ~ asymptDisp +
    extraPois/mean by a Gamma GLM with identity link (stats::glm's IRLS restated: working weights
    mu^-2, working response = disps)."""
    means = np.asarray(means, np.float64)
    disps = np.asarray(disps, np.float64)
    coefs = np.array([0.1, 1.0])
    it = 0
    while True:
        residuals = disps / (coefs[0] + coefs[1] / means)
        good = (residuals > 1e-4) & (residuals < 15)
        yg, xg = disps[good], 1.0 / means[good]
        b = coefs.copy()
        converged = False
        r0 = yg / (b[0] + b[1] * xg)
        dev_old = -2.0 * (np.log(r0).sum() - (r0 - 1.0).sum())     # glm.fit: devold from the start values
        for _ in range(25):                                   # glm.control(maxit = 25, epsilon = 1e-8)
            mu = b[0] + b[1] * xg
            if mu.min() <= 0:
                raise RuntimeError("parametric dispersion fit failed")
            wgt = 1.0 / (mu * mu)
            wx = wgt * xg
            s0, s1, s2 = wgt.sum(), wx.sum(), (wx * xg).sum()
            t0, t1 = (wgt * yg).sum(), (wx * yg).sum()
            det = s0 * s2 - s1 * s1
            b = np.array([(s2 * t0 - s1 * t1) / det, (s0 * t1 - s1 * t0) / det])
            mu = b[0] + b[1] * xg
            if mu.min() <= 0:
                raise RuntimeError("parametric dispersion fit failed")
            r = yg / mu
            dev = -2.0 * (np.log(r).sum() - (r - 1.0).sum())
            if abs(dev - dev_old) / (abs(dev) + 0.1) < 1e-8:
                converged = True
                break
            dev_old = dev
        oldcoefs = coefs
        coefs = b
        if not (coefs > 0).all():
            raise RuntimeError("parametric dispersion fit failed")
        if np.sum(np.log(coefs / oldcoefs) ** 2) < 1e-6 and converged:
            break
        it += 1
        if it > 10:
            raise RuntimeError("dispersion fit did not converge")
    return coefs


def _mad(v):
    med = np.median(v)
    return 1.4826 * np.median(np.abs(v - med))


def trimmed_mean_fit(dge, minDisp=1e-8, trim=0.001):
    """fitType = "mean" (R/core.R:894-899): mean(dispGeneEst[dispGeneEst > 10 minDisp], na.rm = TRUE, trim = 0.001).
    base::mean.default keeps the order statistics floor(N trim) + 1 ... N - floor(N trim) and takes their long-double
    mean with a correction pass -- to double precision the correctly rounded mean.  The specification shared with the
    chain's kernel (csrc/pipeline.hip: trend_mean_kernel): every kept value as the integer floor(x 2^128), the exact
    integer sum, the quotient rounded once to nearest-even -- Python integers here."""
    import math
    with np.errstate(invalid="ignore"):
        v = np.sort(dge[dge > 10 * minDisp])
    k = int(np.floor(v.size * trim))
    kept = v[k: v.size - k]
    fr, ex = np.frexp(kept)
    M = np.ldexp(fr, 53).astype(np.int64)
    S = 0
    for mnt, sh in zip(M.tolist(), (ex.astype(np.int64) - 53 + 128).tolist()):
        S += (mnt << sh) if sh >= 0 else (mnt >> -sh)
    Q, rem = divmod(S, int(kept.size))
    h = Q.bit_length() - 1
    if h <= 52:
        return math.ldexp(float(Q), -128)
    shift = h - 52
    mant, rest, half = Q >> shift, Q & ((1 << shift) - 1), 1 << (shift - 1)
    if rest > half or (rest == half and (rem != 0 or (mant & 1))):
        mant += 1
    return math.ldexp(float(mant), shift - 128)


def estimateDispersionsFit(dds, fitType="parametric", minDisp=1e-8, engine=None):
    """R/core.R:864-939 + `dispersionFunction<-` (R/methods.R:142-190)"""
    E = engine if engine is not None else dds.engine
    dge, bm = dds.mcols["dispGeneEst"], dds.mcols["baseMean"]
    useForFit = dge > 100 * minDisp
    if useForFit.sum() == 0:
        raise RuntimeError("all gene-wise dispersion estimates are within 2 orders of magnitude from the minimum value")
    if fitType == "parametric":
        try:
            coefs = E.parametric_fit(bm[useForFit], dge[useForFit])
            fn = ("parametric", coefs)
        except RuntimeError:
            fitType = "mean"       # the reference falls back to locfit (not available here)
    if fitType == "mean":
        fn = ("mean", trimmed_mean_fit(dge, minDisp))                                  # mean(trim = 0.001)
    if callable(fitType):
        # the caller's trend, a function of the normalized mean: what R reaches with fitType = "local" (locfit, :889-893:
        # `dispFunction <- localDispersionFit(means[useForFit], disps[useForFit], minDisp)`) or `dispersionFunction(dds) <- f`
        # (R/methods.R:142-190); it is given the vectors of the genes that enter the fit, returns the function of the mean
        fn = ("custom", fitType(bm[useForFit], dge[useForFit]))
    if fn[0] == "parametric":
        dispFit = fn[1][0] + fn[1][1] / bm
    elif fn[0] == "custom":
        with np.errstate(invalid="ignore", divide="ignore"):
            dispFit = np.asarray(fn[1](bm), dtype=np.float64)
    else:
        dispFit = np.full(bm.shape, fn[1])
    dds.mcols["dispFit"] = dispFit
    aboveMinDisp = dge >= minDisp * 100
    varLogDispEsts = None
    if aboveMinDisp.sum() > 0:
        res = E.vlog(dge) - E.vlog(dispFit)
        md = E.mad(res[aboveMinDisp])
        varLogDispEsts = md * md                                                       # mad(...)^2, methods.R:180
    dds.dispersionFunction = {"fitType": fn[0], "coefficients": fn[1], "varLogDispEsts": varLogDispEsts}
    return dds


def estimateDispersionsPriorVar(dds, minDisp=1e-8):
    """R/core.R:1135-1208 (the (m-p) <= 3 Monte-Carlo branch relies on R's RNG stream and is not
    mirrored)"""
    dge = dds.mcols["dispGeneEst"]
    aboveMinDisp = dge >= minDisp * 100
    if aboveMinDisp.sum() == 0:
        raise RuntimeError("no data found which is greater than minDisp")
    varLogDispEsts = dds.dispersionFunction["varLogDispEsts"]
    m, p = dds.x.shape
    if (m - p) <= 3 and m > p:
        raise NotImplementedError("residual df <= 3: the reference's seeded Monte-Carlo matching is not mirrored")
    if m > p:
        expVarLogDisp = sps.polygamma(1, (m - p) / 2.0)
        return float(max(varLogDispEsts - expVarLogDisp, 0.25))                       # :1200
    return float(varLogDispEsts)


def estimateDispersionsMAP(dds, outlierSD=2, dispPriorVar=None, minDisp=1e-8, kappa_0=1.0, dispTol=1e-6,
                           maxit=100, useCR=True, weightThreshold=1e-2, modelMatrix=None):
    """R/core.R:943-1131.  modelMatrix: the caller's design (R/core.R:945, 1003-1005: the Cox-Reid term of the MAP search runs
    on it); default: the one the gene-wise estimate ran on, else the object's"""
    E = dds.engine
    if modelMatrix is None:
        modelMatrix = dds.attrs.get("geneEstModelMatrix")
    xh = dds.xh if modelMatrix is None else E.design(np.asarray(modelMatrix, np.float64))
    if dispPriorVar is None:
        dispPriorVar = estimateDispersionsPriorVar(dds, minDisp)                     # :986
    dds.dispersionFunction["dispPriorVar"] = dispPriorVar
    weights, useWeights = getAndCheckWeights(dds, weightThreshold, modelMatrix=modelMatrix)      # :999 (no 1e-6 floor here)
    dge, dfit = dds.mcols["dispGeneEst"], dds.mcols["dispFit"]
    mu = dds.assays["mu"]
    dispInit = np.where(dge > 0.1 * dfit, dge, dfit)                                 # :1019-1021
    dispInit = np.where(np.isnan(dispInit), dfit, dispInit)
    log_dfit = E.vlog(dfit)
    res = E.fit_disp(dds.y, xh, mu, E.vlog(dispInit), log_dfit, dispPriorVar, np.log(minDisp / 10),
                     kappa_0, dispTol, maxit, True, weights, useWeights, weightThreshold, useCR)   # :1027-1039
    dispMAP = E.vexp(res["log_alpha"])
    dispIter = res["iter"]
    dispConv = dispIter < maxit                                                      # :1048
    refitDisp = ~dispConv
    if refitDisp.sum() > 0:
        idx = np.where(refitDisp)[0]
        dispGrid = fitDispGridWrapper(E, E.take_rows(dds.y, idx), xh, E.take_rows(mu, idx),
                                      log_dfit[idx], dispPriorVar, True, E.take_rows(weights, idx),
                                      useWeights, weightThreshold, True, dds.m)      # :1051-1061
        dispMAP[refitDisp] = dispGrid
    maxDisp = max(10, dds.m)
    dispMAP = np.minimum(np.maximum(dispMAP, minDisp), maxDisp)                      # :1100-1101
    dispersionFinal = dispMAP.copy()
    varLogDispEsts = dds.dispersionFunction["varLogDispEsts"]
    dispOutlier = E.vlog(dge) > log_dfit + outlierSD * np.sqrt(varLogDispEsts)      # :1111-1113
    dispOutlier[np.isnan(dispOutlier)] = False
    dispersionFinal[dispOutlier] = dge[dispOutlier]
    dds.mcols.update(dispersion=dispersionFinal, dispIter=dispIter, dispOutlier=dispOutlier, dispMAP=dispMAP)
    return dds


def estimateDispersions(dds, fitType="parametric", maxit=100, dispPriorVar=None, **kw):
    """R/methods.R:500-563 (maxit is handed to both dispersion searches, :520,:546).  dispPriorVar: the argument of
    estimateDispersionsMAP (R/core.R:989-994) for callers that run the three steps themselves -- the way an analysis with
    residual df <= 3 gets its prior variance (R's estimate there needs R's RNG, :1155-1190)."""
    estimateDispersionsGeneEst(dds, maxit=maxit, **kw)
    estimateDispersionsFit(dds, fitType=fitType)
    estimateDispersionsMAP(dds, maxit=maxit, dispPriorVar=dispPriorVar, modelMatrix=kw.get("modelMatrix"))   # (R/methods.R:546)
    return dds


# ------------------------------------------------------------------ beta prior (R/core.R:1601-1689, R/expanded.R)
def standard_model_matrix(factors):
    """model.matrix(~ f1 + f2 + ...) with treatment contrasts; `factors` is an ordered dict
    name -> integer level codes (0 = reference level).  Returns (x, column names)."""
    m = len(next(iter(factors.values())))
    cols, names = [np.ones(m)], ["Intercept"]
    for f, codes in factors.items():
        codes = np.asarray(codes)
        for lv in range(1, int(codes.max()) + 1):
            cols.append((codes == lv).astype(np.float64)); names.append("%s%d" % (f, lv))
    return np.column_stack(cols), names


def makeExpandedModelMatrix(factors):
    """R/expanded.R:1-18: intercept + one indicator per level of every design factor
    (rank deficient; made solvable by the ridge)."""
    m = len(next(iter(factors.values())))
    cols, names = [np.ones(m)], ["Intercept"]
    for f, codes in factors.items():
        codes = np.asarray(codes)
        for lv in range(0, int(codes.max()) + 1):
            cols.append((codes == lv).astype(np.float64)); names.append("%s%d" % (f, lv))
    return np.column_stack(cols), names


def Hmisc_wtd_quantile(x, weights, prob, normwt=True, sorter=None):
    """R/core.R:2762-2800 (type = 'quantile'), one probability.  `sorter(values)` may supply the stable ascending
    order of `values` (an engine can sort on the device; the stable order of a vector is unique, so nothing changes)."""
    x = np.asarray(x, float); w = np.asarray(weights, float)
    keep = ~(np.isnan(w) | (w == 0))
    x, w = x[keep], w[keep]
    if normwt:
        w = w * x.size / np.cumsum(w)[-1]          # (every sum here is sequential: the order csrc/beta_prior.hip follows)
    o = np.argsort(x, kind="stable") if sorter is None else np.asarray(sorter(x))
    x, w = x[o], w[o]
    # the distinct values of the sorted vector and, per value, the sum of its weights in index order (what
    # unique(return_inverse) + bincount give, without sorting a second time)
    head = np.empty(x.size, bool)
    head[:1] = True
    np.not_equal(x[1:], x[:-1], out=head[1:])
    ux = x[head]
    inv = np.cumsum(head) - 1
    wts = np.bincount(inv, weights=w)
    cs = np.cumsum(wts)
    n = cs[-1]
    order = 1 + (n - 1) * prob
    low = max(np.floor(order), 1.0)
    high = min(low + 1, n)
    frac = order % 1

    def stepq(q):                      # approx(cumsum(wts), x, method='constant', f=1, rule=2)
        i = np.searchsorted(cs, q, side="left")
        return ux[min(i, ux.size - 1)]
    return (1 - frac) * stepq(low) + frac * stepq(high)


def matchWeightedUpperQuantileForVariance(x, weights, upperQuantile=0.05, sorter=None):
    """R/core.R:2416-2419"""
    qn = 1.959963984540054 if upperQuantile == 0.05 else sps.ndtri(1 - upperQuantile / 2)       # qnorm(1 - 0.05 / 2)
    sdEst = float(Hmisc_wtd_quantile(np.abs(x), weights, 1 - upperQuantile, normwt=True, sorter=sorter) / qn)
    return sdEst * sdEst


def matchUpperQuantileForVariance(x, upperQuantile=0.05):
    """R/core.R:2411-2414"""
    return float(np.quantile(np.abs(x), 1 - upperQuantile) / sps.ndtri(1 - upperQuantile / 2)) ** 2


def estimateBetaPriorVar(dds, mleBetaMatrix, names, betaPriorMethod="weighted", upperQuantile=0.05,
                         modelMatrixType="standard", factors=None, sorter=None):
    """R/core.R:1601-1689.  `mleBetaMatrix` (log2 scale) are the MLE coefficients of the standard
    design whose column names are `names`."""
    beta = np.asarray(mleBetaMatrix, float)
    names = list(names)
    if modelMatrixType == "expanded":                                  # addAllContrasts, expanded.R:76-98
        for f, codes in factors.items():
            idx = [i for i, nm in enumerate(names) if nm.startswith(f) and nm != "Intercept"]
            k = len(idx)
            if k > 1:
                for j in range(k - 1):
                    for i in range(j + 1, k):
                        beta = np.column_stack([beta, beta[:, idx[i]] - beta[:, idx[j]]])
                        names.append(f + "Cntrst")
    dispFit = dds.mcols.get("dispFit")
    if dispFit is None:
        dispFit = np.mean(dds.mcols["dispersion"])
    weights = 1.0 / (1.0 / dds.mcols["baseMean"] + dispFit)            # :1641-1642
    pv = np.empty(beta.shape[1])
    for c in range(beta.shape[1]):
        xcol = beta[:, c]
        if beta.shape[0] == 1:                                         # a one-gene object: (betaMatrix)^2, :1647,1662
            pv[c] = float(xcol[0]) ** 2
            continue
        use = np.abs(xcol) < 10
        if use.sum() == 0:
            pv[c] = 1e6
        elif betaPriorMethod == "quantile":
            pv[c] = matchUpperQuantileForVariance(xcol[use], upperQuantile)
        else:
            pv[c] = matchWeightedUpperQuantileForVariance(xcol[use], weights[use], upperQuantile, sorter=sorter)
    for c, nm in enumerate(names):
        if nm == "Intercept":
            pv[c] = 1e6                                               # :1669-1671
    if modelMatrixType == "expanded":                                  # averagePriorsOverLevels, expanded.R:20-73
        xe, enames = makeExpandedModelMatrix(factors)
        out = np.zeros(len(enames))
        for c, nm in enumerate(names):
            if nm in enames:
                out[enames.index(nm)] = pv[c]
        for f in factors:
            mmset = {nm for nm in enames if nm.startswith(f)} | {f + "Cntrst"}
            vals = [pv[c] for c, nm in enumerate(names) if nm in mmset]
            meanvar = float(np.cumsum(vals)[-1] / len(vals))
            for i, nm in enumerate(enames):
                if nm.startswith(f) and nm != "Intercept":
                    out[i] = meanvar
        if not (out > 0).all():
            raise RuntimeError("beta prior is not greater than 0")
        return out, enames
    return pv, names


def fitGLMsWithPrior(dds, betaTol=1e-8, maxit=100, useOptim=True, useQR=True, betaPriorVar=None,
                     modelMatrixType="standard", factors=None, minmu=0.5, weights=None, useWeights=False):
    """R/fitNbinomGLMs.R:242-337: (1) MLE fit with the wide prior, (2) the all-gene beta prior
    variance, (3) refit with lambda = 1/betaPriorVar on the standard or expanded design."""
    if modelMatrixType == "expanded" and factors is None:
        raise ValueError("an expanded model matrix needs the design factors")
    names = standard_model_matrix(factors)[1] if factors is not None else ["Intercept"] + [
        "V%d" % i for i in range(1, dds.p)]
    fit = fitNbinomGLMs(dds, betaTol=betaTol, maxit=maxit, useOptim=useOptim, useQR=useQR, minmu=minmu,
                        weights=weights, useWeights=useWeights)                       # :256-260
    H, mu, mle = fit["hat_diagonals"], fit["mu"], fit["betaMatrix"]
    if betaPriorVar is None:
        betaPriorVar, pnames = estimateBetaPriorVar(dds, mle, names, modelMatrixType=modelMatrixType,
                                                    factors=factors)                  # :293
    if (np.asarray(betaPriorVar) == 0).any():
        raise ValueError("beta prior variances are equal to zero for some variables")
    lam = 1.0 / np.asarray(betaPriorVar, float)                                       # :311
    if modelMatrixType == "expanded":
        xe, _ = makeExpandedModelMatrix(factors)
        fit2 = fitNbinomGLMs(dds, lam=lam, betaTol=betaTol, maxit=maxit, useOptim=useOptim, useQR=useQR,
                             modelMatrix=xe, minmu=minmu, weights=weights, useWeights=useWeights,
                             want_loglike=True)                                        # :319-325
    else:
        fit2 = fitNbinomGLMs(dds, lam=lam, betaTol=betaTol, maxit=maxit, useOptim=useOptim, useQR=useQR,
                             minmu=minmu, weights=weights, useWeights=useWeights, want_loglike=True)   # :314-317
    return {"fit": fit2, "H": H, "betaPriorVar": np.asarray(betaPriorVar), "mu": mu,
            "modelMatrix": fit2["modelMatrix"], "mleBetaMatrix": mle}


# ------------------------------------------------------------------ tests
def nbinomWaldTest(dds, betaTol=1e-8, maxit=100, useOptim=True, useT=False, df=None, useQR=True, minmu=0.5,
                   modelMatrix=None, betaPrior=False, betaPriorVar=None, modelMatrixType=None, factors=None):
    """R/core.R:1332-1565.  betaPrior = TRUE goes through
    fitGLMsWithPrior (:1416-1432), by default on the expanded model matrix (:1374-1380)."""
    if "dispersion" not in dds.mcols:
        raise RuntimeError("testing requires dispersion estimates, first call estimateDispersions()")
    E = dds.engine
    weights, useWeights = getAndCheckWeights(dds)
    if not betaPrior:
        fit = fitNbinomGLMs(dds, betaTol=betaTol, maxit=maxit, useOptim=useOptim, useQR=useQR, minmu=minmu,
                            modelMatrix=modelMatrix, weights=weights, useWeights=useWeights,
                            want_loglike=True)                                                   # :1403-1408
        H, mu_fit = fit["hat_diagonals"], fit["mu"]
        bpv = np.full(fit["nterms"], 1e6)
    else:
        if modelMatrixType is None:
            modelMatrixType = "expanded" if factors is not None else "standard"
        pf = fitGLMsWithPrior(dds, betaTol=betaTol, maxit=maxit, useOptim=useOptim, useQR=useQR,
                              betaPriorVar=betaPriorVar, modelMatrixType=modelMatrixType, factors=factors,
                              minmu=minmu, weights=weights, useWeights=useWeights)       # :1416-1421
        fit, H, mu_fit, bpv = pf["fit"], pf["H"], pf["mu"], pf["betaPriorVar"]
        dds.mcols["MLE_beta"] = pf["mleBetaMatrix"]
    fit = dict(fit); fit["mu"] = mu_fit; fit["hat_diagonals"] = H
    dds.assays["mu"] = fit["mu"]
    dds.assays["H"] = fit["hat_diagonals"]
    dds.attrs.update(betaPrior=bool(betaPrior), betaPriorVar=bpv, test="Wald",
                     modelMatrixType=modelMatrixType, factors=factors)
    calculateCooksDistance(dds, H, dds.x if modelMatrix is None else np.asarray(modelMatrix, float))   # :1451-1463
    betaMatrix, betaSE = fit["betaMatrix"], fit["betaSE"]
    with np.errstate(divide="ignore", invalid="ignore"):
        WaldStatistic = betaMatrix / betaSE                                         # :1471
    if useT:
        from scipy.stats import t as tdist
        if df is None:
            num = E.to_numpy(weights).sum(axis=1) if useWeights else np.full(dds.n, dds.m)
            df = num - dds.p
        df = np.where(np.asarray(df, float) > 0, df, np.nan)
        WaldPvalue = 2 * tdist.sf(np.abs(WaldStatistic), df=np.asarray(df)[:, None])
    else:
        WaldPvalue = E.two_sided_normal_p(WaldStatistic)                            # :1507
    logLike = fit["logLike"]                                                    # fitNbinomGLMs.R:182,399
    dds.mcols.update(beta=betaMatrix, betaSE=betaSE, WaldStatistic=WaldStatistic, WaldPvalue=WaldPvalue,
                     betaConv=fit["betaConv"], betaIter=fit["betaIter"], deviance=-2 * logLike,
                     rowsForOptim=fit["rowsForOptim"])
    return dds


def pchisq_upper(stat, df):
    """pchisq(stat, df, lower.tail = FALSE) (R/core.R:1878) -- the value scipy.stats.chi2.sf returns (its _sf IS
    scipy.special.chdtrc) without the distribution machinery around it (argument broadcasting / masking: a quarter of
    its 4 ms per 60 000 genes, which shows in a C4 step on the device chain)"""
    stat = np.asarray(stat, np.float64)
    if not (np.ndim(df) == 0 and df > 0):
        from scipy.stats import chi2
        return chi2.sf(stat, df=df)
    from scipy.special import chdtrc
    out = chdtrc(float(df), stat)
    out[np.isnan(stat)] = np.nan
    return out


def nbinomLRT(dds, reduced, betaTol=1e-8, maxit=100, useOptim=True, useQR=True, minmu=0.5):
    """R/core.R:1787-2012: full vs reduced model matrices (betaPrior = FALSE).  `reduced` is a
    model matrix whose column space is nested in dds.x; an intercept-only reduced model takes
    the closed form of R/fitNbinomGLMs.R:99-137."""
    E = dds.engine
    reduced = np.asarray(reduced, np.float64)
    weights, useWeights = getAndCheckWeights(dds)
    disp = dds.mcols["dispersion"]
    full = fitNbinomGLMs(dds, betaTol=betaTol, maxit=maxit, useOptim=useOptim, useQR=useQR, minmu=minmu,
                         weights=weights, useWeights=useWeights, want_loglike=True)
    ll_full = full["logLike"]
    red = fitNbinomGLMs(dds, modelMatrix=reduced, betaTol=betaTol, maxit=maxit, useOptim=useOptim,
                        useQR=useQR, minmu=minmu, weights=weights, useWeights=useWeights, want_hat=False,
                        want_loglike=True)                        # closed form when reduced is ~1 (:99-137)
    ll_red = red["logLike"]
    LRTStatistic = 2 * (ll_full - ll_red)                                            # :1877
    LRTPvalue = pchisq_upper(LRTStatistic, full["nterms"] - red["nterms"])            # :1878
    dds.assays["mu"] = full["mu"]
    dds.assays["H"] = full["hat_diagonals"]
    dds.attrs.update(betaPrior=False, test="LRT", reduced=reduced)
    calculateCooksDistance(dds, full["hat_diagonals"], dds.x)                         # :1886-1891
    dds.mcols.update(beta=full["betaMatrix"], betaSE=full["betaSE"], LRTStatistic=LRTStatistic,
                     LRTPvalue=LRTPvalue, fullBetaConv=full["betaConv"], betaIter=full["betaIter"],
                     deviance=-2 * ll_full)
    return dds


# ------------------------------------------------------------------ count outliers
def nOrMoreInCell(modelMatrix, n):
    """R/core.R:2366-2371: per sample, are there n or more samples with the same model-matrix row"""
    return _cells(modelMatrix)[1] >= n


def calculateCooksDistance(dds, H, modelMatrix):
    """calculateCooksDistance (R/core.R:2333-2340) on the robust method-of-moments dispersion
    (:2277-2331) + recordMaxCooks (:2349-2359); one engine pass over assays mu / H."""
    r = dds.engine.cooks_distance(dds.y, dds.nf, dds.assays["mu"], H, modelMatrix)
    dds.assays["cooks"] = r["cooks"]
    dds.mcols["maxCooks"] = r["maxCooks"]
    dds.attrs["dispModelMatrix"] = np.asarray(modelMatrix, np.float64)
    return dds


def replaceOutliers(dds, trim=.2, cooksCutoff=None, minReplicates=7, whichSamples=None):
    """R/core.R:2069-2115.  The replaced count matrix goes to assays["replaceCounts"] (the handle
    the refit reads); dds.y keeps the original counts, as counts(dds) does at the end of
    refitWithoutOutliers (:2553-2557)."""
    from scipy.stats import f as fdist
    if "cooks" not in dds.assays:
        raise RuntimeError("first run DESeq, nbinomWaldTest, or nbinomLRT to identify outliers")
    if minReplicates < 3:
        raise ValueError("at least 3 replicates are necessary in order to indentify a sample as a count outlier")
    x = dds.attrs["dispModelMatrix"]
    p, m = x.shape[1], dds.m
    if m <= p:
        return dds
    if cooksCutoff is None:
        cooksCutoff = float(fdist.ppf(.99, p, m - p))                                # :2081
    if whichSamples is None:
        whichSamples = nOrMoreInCell(x, minReplicates)                               # :2101
    whichSamples = np.asarray(whichSamples, bool)
    dds.attrs["replaceable"] = whichSamples
    r = dds.engine.replace_outliers(dds.y, dds.nf, dds.assays["cooks"], cooksCutoff, whichSamples, trim)
    dds.mcols["replace"] = r["replace"]
    dds.assays["replaceCounts"] = r["counts"] if whichSamples.any() else dds.y       # :2108-2112
    return dds


_RESULT_COLS = ("beta", "betaSE", "WaldStatistic", "WaldPvalue", "LRTStatistic", "LRTPvalue", "betaConv",
                "fullBetaConv", "betaIter", "deviance", "maxCooks")


def _dispersion_function(dds, baseMean):
    fn = dds.dispersionFunction
    if fn["fitType"] == "parametric":
        return fn["coefficients"][0] + fn["coefficients"][1] / baseMean
    if fn["fitType"] == "custom":
        return np.asarray(fn["coefficients"](baseMean), dtype=np.float64)
    return np.full(baseMean.shape, fn["coefficients"])


def refitWithoutOutliers(dds, test="Wald", reduced=None, minReplicatesForReplace=7, disp_maxit=100, count_all=None, **kw):
    """R/core.R:2484-2563: replace count outliers by the trimmed mean, then re-estimate the
    dispersion and refit the rows that had a replacement (all through the same engine entry
    points, on the row subset).  `count_all`: when `dds` is one shard of a gene-sharded analysis, a function that adds
    a count up over all shards (called exactly once) -- the closing steps (NA results on the rows that became all zero,
    maxCooks, :2535-2546) run when ANY row of the whole object was refitted (:2496), as on the unsharded object."""
    E = dds.engine
    # the refit runs estimateDispersionsGeneEst / MAP and nbinomWaldTest / nbinomLRT on their DEFAULTS: DESeq()'s betaTol,
    # maxit, useQR, minmu, useT, df are not handed on (:2509-2531) -- refitted rows carry normal-distribution p-values
    # even in a useT analysis
    # ... but the caller's model matrix IS (modelMatrix = modelMatrix in the gene-wise estimate, the MAP estimate and the test of
    # the refit, :2509-2527)
    kw = {k: v for k, v in kw.items() if k == "modelMatrix" and v is not None}
    replaceOutliers(dds, minReplicates=minReplicatesForReplace)
    if "replace" not in dds.mcols:
        if count_all is not None:
            count_all(0)
        return dds
    replace = dds.mcols["replace"]
    nrefit = int(replace.sum())
    newAllZero = np.array([], int)
    refitReplace = np.array([], int)
    whole = None
    if nrefit > 0:
        idx_rep = np.where(replace)[0]
        whole = dds.subset(idx_rep, dds.assays["replaceCounts"])
        getBaseMeansAndVariances(whole)                                               # :2491
        for k in ("baseMean", "baseVar", "allZero"):
            dds.mcols[k] = dds.mcols[k].copy()
            dds.mcols[k][idx_rep] = whole.mcols[k]
        newAllZero = idx_rep[whole.mcols["allZero"]]
        refitReplace = idx_rep[~whole.mcols["allZero"]]
    n_refit_all = refitReplace.size if count_all is None else int(count_all(int(refitReplace.size)))
    if refitReplace.size > 0:                                                         # :2496
        keep = ~whole.mcols["allZero"]
        sub = whole if keep.all() else dds.subset(refitReplace, dds.assays["replaceCounts"])
        estimateDispersionsGeneEst(sub, maxit=disp_maxit, **kw)                       # :2509
        sub.mcols["dispFit"] = _dispersion_function(dds, sub.mcols["baseMean"])       # :2512
        estimateDispersionsMAP(sub, dispPriorVar=dds.dispersionFunction["dispPriorVar"], maxit=disp_maxit, **kw)   # :2518-2519
        if test == "Wald":
            nbinomWaldTest(sub, betaPrior=dds.attrs.get("betaPrior", False), betaPriorVar=(
                dds.attrs["betaPriorVar"] if dds.attrs.get("betaPrior", False) else None),
                modelMatrixType=dds.attrs.get("modelMatrixType"), factors=dds.attrs.get("factors"), **kw)
        else:
            nbinomLRT(sub, reduced)         # (full / reduced are the matrices themselves in this mirror)
        for k, v in sub.mcols.items():                                                # :2533-2534
            if k not in dds.mcols or k == "rowsForOptim" or np.shape(v)[:1] != (sub.n,):
                continue
            dst = np.array(dds.mcols[k], copy=True)
            dst[refitReplace] = v
            dds.mcols[k] = dst
    if n_refit_all > 0:
        for k in _RESULT_COLS:                                                        # :2535
            if k in dds.mcols and newAllZero.size:
                a = dds.mcols[k]
                if a.dtype.kind != "f":
                    a = a.astype(np.float64)
                a[newAllZero] = np.nan
                dds.mcols[k] = a
        replaceable = dds.attrs["replaceable"]
        if replaceable.all():                                                         # :2538-2546
            dds.mcols["maxCooks"] = np.full(dds.n, np.nan)
        else:
            x = dds.attrs["dispModelMatrix"]
            if x.shape[0] > x.shape[1] and nOrMoreInCell(x, 3).any():
                dds.mcols["maxCooks"] = E.masked_row_max(dds.assays["cooks"], nOrMoreInCell(x, 3), replaceable)
            else:
                dds.mcols["maxCooks"] = np.full(dds.n, np.nan)
    if nrefit == 0:
        dds.assays.pop("replaceCounts", None)
        return dds
    dds.assays["replaceCooks"] = dds.assays["cooks"]                                   # :2551
    return dds


def cooksOutlier(dds, cooksCutoff=None):
    """The Cook's-distance filter results() applies to the p-values (R/results.R:520-564): genes whose
    maxCooks exceeds qf(.99, p, m - p); for a two-group design a gene is kept when three or more
    counts are larger than the count with the largest Cook's distance (:541-560).  Returns the flags
    (results() sets pvalue <- NA there)."""
    from scipy.stats import f as fdist
    x = dds.attrs["dispModelMatrix"]
    m, p = x.shape
    if cooksCutoff is None:
        cooksCutoff = float(fdist.ppf(.99, p, m - p))
    mx = dds.mcols["maxCooks"]
    with np.errstate(invalid="ignore"):
        out = np.where(np.isnan(mx), False, mx > cooksCutoff)
    two_group = p == 2 and (x[:, 0] == 1).all() and set(np.unique(x[:, 1])) <= {0.0, 1.0}
    if out.any() and two_group:
        E = dds.engine
        ii = np.where(out)[0]
        cnt = E.to_numpy(E.take_rows(dds.y, ii))
        ck = E.to_numpy(E.take_rows(dds.assays["cooks"], ii))
        outCount = cnt[np.arange(ii.size), np.argmax(ck, axis=1)]
        out[ii[(cnt > outCount[:, None]).sum(axis=1) >= 3]] = False
    return out


def _DESeqNZ(dds, test, fitType, reduced, minReplicatesForReplace, disp_maxit=100, **kw):
    dpv = kw.pop("dispPriorVar", None)      # (not an argument of R's DESeq(): of estimateDispersionsMAP, R/core.R:989-994)
    estimateDispersions(dds, fitType=fitType, maxit=disp_maxit, minmu=kw.get("minmu", 0.5), dispPriorVar=dpv)      # R/core.R:393
    if test == "Wald":
        nbinomWaldTest(dds, **kw)
    elif test == "LRT":
        nbinomLRT(dds, reduced, **kw)
    else:
        raise ValueError("test should be either 'Wald' or 'LRT'")
    if np.isfinite(minReplicatesForReplace) and nOrMoreInCell(dds.x, minReplicatesForReplace).any():   # :419-426
        refitWithoutOutliers(dds, test=test, reduced=reduced, minReplicatesForReplace=minReplicatesForReplace,
                             disp_maxit=disp_maxit, **kw)
    return dds


def DESeq(dds, test="Wald", fitType="parametric", reduced=None, minReplicatesForReplace=7, **kw):
    """R/core.R:280-432, serial path (size factors are taken as given: estimateSizeFactors is
    outside the hot path).  minReplicatesForReplace = np.inf switches the outlier refit off.
    Rows whose counts are all zero are set aside as every step of the reference does (objectNZ,
    e.g. R/core.R:705,1351) and come back as NA (NaN) rows of mcols (buildDataFrameWithNARows);
    the n x m assays then cover the non-zero rows, listed in attrs["nz_rows"]."""
    getBaseMeansAndVariances(dds)
    if dds.has_weights:
        getAndCheckWeights(dds)       # rows whose weights leave a degenerate design count as all-zero (R/core.R:2737)
    allZero = dds.mcols["allZero"]
    if not allZero.any():
        return _DESeqNZ(dds, test, fitType, reduced, minReplicatesForReplace, **kw)
    nz = np.where(~allZero)[0]
    if nz.size == 0:
        raise ValueError("all genes have zero counts in every sample")
    keep = {k: dds.mcols[k] for k in ("baseMean", "baseVar", "allZero", "weightsFail") if k in dds.mcols}
    sub = _DESeqNZ(dds.subset(nz), test, fitType, reduced, minReplicatesForReplace, **kw)
    out = {}
    for k, v in sub.mcols.items():
        v = np.asarray(v)
        if v.shape[:1] != (nz.size,) or k == "rowsForOptim":
            continue
        full = np.full((dds.n,) + v.shape[1:], np.nan)
        full[nz] = v
        out[k] = full
    for k, v in keep.items():
        # the refit may have updated baseMean / baseVar / allZero of its rows (R/core.R:2491-2494): take the
        # non-zero rows from the sub-analysis, the all-zero rows as first computed
        full = np.array(v, copy=True)
        if k in sub.mcols and np.shape(sub.mcols[k])[:1] == (nz.size,):
            full[nz] = sub.mcols[k]
        out[k] = full
    dds.mcols = out
    dds.assays, dds.attrs = sub.assays, dict(sub.attrs, nz_rows=nz)
    dds.dispersionFunction = sub.dispersionFunction
    return dds
