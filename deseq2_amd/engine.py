"""Engines: where the n x m matrices of one analysis live and who runs the three native
routines on them.

The host-side mirror of the reference's R callers (core.py, fit_nbinom_glms.py) is written
once against this small interface.  n-vectors and n x p matrices are always host numpy
arrays (they are what R keeps in mcols()); n x m matrices are opaque handles:

  HostEngine    handles are numpy arrays in R orientation (genes x samples); every call
                goes through the host-pointer C ABI (dsq_fit_*), i.e. exactly what the
                .Call shim does -- upload, kernels, download.  Constructed with the module
                providing fitBeta/fitDisp/fitDispGrid, so the parity tests can run the very
                same host code over the CPU oracle.
  DeviceEngine  handles are gene-major torch CUDA tensors resident in HBM; calls go through
                dsq_fit_*_dev on the current stream.  Y / nf / weights are uploaded and
                transposed once, mu-hat produced by fitBeta is consumed in place by fitDisp
                (SURVEY 8f-2).  torch is used for memory, streams and the O(n*m)
                elementwise glue (row gathers, clamps); the fits are the HIP kernels.
"""
import threading

import numpy as np

_PROF_LOCK = threading.Lock()


class Launched(dict):
    """Results of a kernel that has been LAUNCHED: n x m device handles are in the dict at once; the per-gene
    host arrays arrive (one device-to-host copy + stream sync) the first time one of them is read.  A caller
    that launches its next kernel before touching the host arrays overlaps its own host code with the GPU."""

    def __init__(self, ready, fetch):
        super().__init__(ready)
        self._fetch = fetch

    def _materialize(self):
        if self._fetch is not None:
            f, self._fetch = self._fetch, None
            super().update(f())

    def __getitem__(self, k):
        if not super().__contains__(k):
            self._materialize()
        return super().__getitem__(k)

    def get(self, k, default=None):
        if not super().__contains__(k):
            self._materialize()
        return super().get(k, default)

    def __contains__(self, k):
        self._materialize()
        return super().__contains__(k)

    def items(self):
        self._materialize()
        return super().items()

    def keys(self):
        self._materialize()
        return super().keys()


class LaunchedVector:
    """an n-vector still on the device; .host() copies it (once)"""

    def __init__(self, fetch):
        self._fetch, self._v = fetch, None

    def host(self):
        if self._fetch is not None:
            self._v, self._fetch = self._fetch(), None
        return self._v


def _gram_rank(G, tol=1e-7):
    """qr()$rank of the matrices A whose Gram matrices A'A are stacked in G (n, p, p), as R computes it: LINPACK
    dqrdc2's limited column pivoting -- going through the columns in order, a column counts when the norm of its part
    orthogonal to the columns already counted is at least tol (1e-7, qr()'s default) times ITS OWN original norm (so
    rescaling a column never changes its decision), and an exactly zero column never counts.  Evaluated on the Gram
    matrix (a Cholesky factorisation that skips the rejected columns); the squared threshold 1e-14 is a hundred
    roundings of a Gram entry."""
    G = np.asarray(G, np.float64)
    n, p, _ = G.shape
    d0 = np.einsum("nii->ni", G)
    C = np.zeros((n, p, p))                      # row k: <q_k, a_j> for every column j; zero when column k was rejected
    rank = np.zeros(n, dtype=np.int64)
    for j in range(p):
        r = d0[:, j] - (C[:, :, j] ** 2).sum(axis=1)
        ok = (d0[:, j] > 0) & (r >= (tol * tol) * d0[:, j])
        num = G[:, j, :] - np.einsum("nk,nkj->nj", C[:, :, j], C)
        C[:, j, :] = np.where(ok[:, None], num / np.sqrt(np.where(ok, r, 1.0))[:, None], 0.0)
        rank += ok
    return rank


def _weights_ok_host(w, x, thr, full_rank):
    """the per-gene checks of getAndCheckWeights (R/core.R:2711-2734), all genes at once: test1 = rank(w_i * X) == p;
    test2 = the rows with w_i > thr, minus their all-zero columns, have full column rank.  Not full rank (expanded
    designs): no design column may be weighted out entirely."""
    n, m = w.shape
    p = x.shape[1]
    ok = np.ones(n, bool)
    if full_rank:
        w2 = w * w
        G1 = np.einsum("nm,ma,mb->nab", w2, x, x)
        t1 = _gram_rank(G1) == p
        keep = (w > thr).astype(np.float64)
        G2 = np.einsum("nm,ma,mb->nab", keep, x, x)
        ncol = (np.einsum("nm,ma->na", keep, np.abs(x)) > 0).sum(axis=1)
        t2 = _gram_rank(G2) == ncol
        ok = t1 & t2
    else:
        for j in range(p):
            allzero = ((w * x[None, :, j]) == 0).all(axis=1)
            ok &= ~allzero
    return ok


class HostEngine:
    name = "host"

    def __init__(self, fns=None):
        if fns is None:
            from . import native as fns
        self.fns = fns

    # ---- handles
    # n x m handles are held column-major, as R holds them: the C ABI then takes them without a layout copy
    def counts(self, K):
        return np.asfortranarray(K, dtype=np.int32)

    def matrix(self, A):
        return None if A is None else np.asfortranarray(A, dtype=np.float64)

    def design(self, x):
        return np.ascontiguousarray(x, dtype=np.float64)

    def to_numpy(self, h):
        return h

    def take_rows(self, h, idx):
        return None if h is None else h[idx]

    def clamp_min(self, h, v):
        return np.maximum(h, v)

    def put_rows(self, h, idx, values):
        """h[idx, ] <- values (a few rows patched from the host, R/fitNbinomGLMs.R:386)"""
        h = np.array(h, copy=True)
        h[np.asarray(idx)] = values
        return h

    def set_rows(self, h, idx, sub):
        """h[idx, ] <- sub, both handles (mu[fitidx, ] <- fitMu, R/core.R:764)"""
        h = np.array(h, copy=True, order="F")
        h[np.asarray(idx)] = sub
        return h

    def nrow(self, h):
        return h.shape[0]

    # ---- n-vector log / exp in the engine's pinned arithmetic (csrc/dsq_math.hpp; the test suite's CPU checker restates it), so that
    # the host-side decision rules give the same bits whichever engine runs them and whether they run here or in
    # the fused device pipeline (R: log(), exp())
    def vlog(self, v):
        return self.fns.unary("log", np.asarray(v, np.float64))

    def vexp(self, v):
        return self.fns.unary("exp", np.asarray(v, np.float64))

    # ---- observation weights (getAndCheckWeights, R/core.R:2697-2751)
    def any_negative(self, h):
        return bool((h < 0).any())

    def row_max_normalize(self, h):
        """weights / apply(weights, 1, max)"""
        return np.asfortranarray(h / h.max(axis=1, keepdims=True))

    def weights_ok(self, w, x, thr, full_rank):
        return _weights_ok_host(np.asarray(w), x, thr, full_rank)

    # ---- O(n*m) steps the reference does in R around the native calls
    def prefit(self, y, nf, x, weights=None):
        """baseMean / baseVar / allZero, roughDispEstimate and the QR start values in one pass
        (R/core.R:2138-2146, 2422-2437; R/fitNbinomGLMs.R:139-145)"""
        return self.fns.prefitMoments(y, nf, x, weights, weights is not None)

    def xim(self, nf):
        """momentsDispEstimate's xim, R/core.R:2440-2444: mean over the samples of 1 / colMeans(nf) -- every column summed
        down the genes in gene order, the m reciprocals in sample order (the order contract of csrc/aux.hip xim_kernel
        and of the library's host entry; numpy's own mean() sums pairwise)"""
        nf = np.asarray(nf, np.float64)
        n, m = nf.shape
        rec = np.empty(m)
        for j0 in range(0, m, 64):
            rec[j0:j0 + 64] = 1.0 / (np.cumsum(nf[:, j0:j0 + 64], axis=0)[-1] / n)
        return float(np.cumsum(rec)[-1] / m)

    def linear_mu(self, y, nf, x):
        """linearModelMuNormalized, R/core.R:2465-2471 (engine kernel: a BLAS product on the host would make
        a gene's value depend on how many rows are in the call)"""
        return self.fns.linearMu(y, nf, x)

    def nbinom_loglike(self, y, mu, disp, weights, useWeights):
        """nbinomLogLike, R/core.R:2208-2217"""
        return self.fns.nbinomLogLike(y, mu, disp, weights if useWeights else None, useWeights)

    def optim_rows(self, y, nf, x, alpha, lam, weights, useWeights, beta_start, minmu):
        """fitNbinomGLMsOptim on the rows handed over (R/fitNbinomGLMs.R:340-407); mu comes back as a host array"""
        return self.fns.optimRows(y, x, nf, alpha, lam, weights if useWeights else None, useWeights, beta_start, minmu)

    def intercept_fit(self, y, nf, alpha, weights, useWeights, mu_floor=0.0, want_hat=True):
        """closed form of the intercept-only model, R/fitNbinomGLMs.R:99-137"""
        return self.fns.interceptFit(y, nf, alpha, weights if useWeights else None, useWeights, mu_floor, want_hat)

    def parametric_fit(self, means, disps):
        """parametricDispersionFit, R/core.R:2166-2190"""
        return self.fns.parametricDispersionFit(means, disps)

    def mad(self, v):
        """stats::mad (R/methods.R:180)"""
        v = np.asarray(v, np.float64)
        med = np.median(v)
        return 1.4826 * np.median(np.abs(v - med))

    def two_sided_normal_p(self, z):
        """2 * pnorm(abs(z), lower.tail = FALSE), R/core.R:1507 (Cody's algorithm as in R's pnorm, engine arithmetic)"""
        return self.fns.unary("pnorm_upper2", np.asarray(z, np.float64))

    # ---- count outliers (R/core.R:2333-2359, 2069-2115)
    def cooks_distance(self, y, nf, mu, H, x):
        """calculateCooksDistance + recordMaxCooks; x = the dispersion model matrix"""
        return self.fns.cooksDistance(y, nf, mu, H, x)

    def replace_outliers(self, y, nf, cooks, cooksCutoff, replaceable, trim=0.2):
        """replaceOutliers: new counts handle + per-gene `replace` flag"""
        r = self.fns.replaceOutliers(y, nf, cooks, cooksCutoff, replaceable, trim)
        return {"counts": np.asfortranarray(r["counts"], dtype=np.int32), "replace": r["replace"]}     # (stays column-major)

    def masked_row_max(self, h, use, zero):
        """apply(h[, use], 1, max) after h[, zero] <- 0   (refitWithoutOutliers, R/core.R:2542-2545)"""
        a = np.where(np.asarray(zero, bool)[None, :], 0.0, h)
        return a[:, np.asarray(use, bool)].max(axis=1)

    def _ones(self, n, m):
        """the all-ones weight matrix R passes when there are no weights (R/fitNbinomGLMs.R:85): one per shape.  The
        engine library never reads unused weights (useWeights = FALSE), so for it no matrix is built at all."""
        if getattr(self.fns, "IGNORES_UNUSED_WEIGHTS", False):
            return None
        if getattr(self, "_ones_cache", None) is None or self._ones_cache.shape != (n, m):
            self._ones_cache = np.ones((n, m), order="F")
        return self._ones_cache

    # ---- the three native routines
    def fit_beta(self, y, x, nf, alpha_hat, contrast, beta_mat, lam, weights, useWeights, tol, maxit, useQR,
                 minmu, want_mu=True, mu_floor=0.0, want_hat=True):
        n, m = y.shape
        w = weights if weights is not None else self._ones(n, m)
        # mu = nf * exp(x beta) comes back from the engine (extension of the fitBeta entry point)
        # instead of being recomputed on the host as R/fitNbinomGLMs.R:180 does
        return self.fns.fitBeta(y, x, nf, alpha_hat, contrast, beta_mat, lam, w, useWeights, tol, maxit, useQR, minmu,
                                want_mu=want_mu, mu_floor=mu_floor, want_hat=want_hat)

    def fit_disp(self, y, x, mu_hat, log_alpha, prior_mean, prior_sigmasq, min_log_alpha, kappa_0, tol, maxit,
                 usePrior, weights, useWeights, weightThreshold, useCR):
        n, m = y.shape
        w = weights if weights is not None else self._ones(n, m)
        return self.fns.fitDisp(y, x, mu_hat, log_alpha, prior_mean, prior_sigmasq, min_log_alpha, kappa_0, tol,
                                maxit, usePrior, w, useWeights, weightThreshold, useCR)

    def fit_disp_grid(self, y, x, mu_hat, disp_grid, prior_mean, prior_sigmasq, usePrior, weights, useWeights,
                      weightThreshold, useCR):
        n, m = y.shape
        w = weights if weights is not None else self._ones(n, m)
        return self.fns.fitDispGrid(y, x, mu_hat, disp_grid, prior_mean, prior_sigmasq, usePrior, w, useWeights,
                                    weightThreshold, useCR)


class DeviceEngine:
    name = "device"

    def __init__(self, device="cuda:0"):
        import torch
        from . import native
        self.torch = torch
        self.native = native
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("DeviceEngine needs a CUDA/HIP device; deseq2_amd has no CPU compute path")
        torch.cuda.set_device(self.device)
        from . import _lib
        _lib.check(_lib.lib().dsq_set_device(self.device.index or 0))
        self._cache = {}
        self._cells = {}
        self._tls = threading.local()
        self.record = None          # set to a list to collect (name, n, ms) per fit kernel
        self.want_d2lp = False      # estimateDispersions* never read fitDisp$last_d2lp (R/core.R:784-787,1042)

    def _timed(self, name, n, fn):
        """profiling pass only: the C library brackets the fit kernel with HIP events on the
        launch stream (dsq_profile_enable); read the duration back after the call."""
        if self.record is None:
            return fn()
        from . import _lib
        L = _lib.lib()
        with _PROF_LOCK:            # the library keeps ONE pair of events: chunk threads take turns here
            L.dsq_profile_enable(1)
            r = fn()
            ms = L.dsq_profile_last_ms()
            L.dsq_profile_enable(0)
        self.record.append((name, n, ms))
        return r

    def _host(self, t):
        """device tensor -> host tensor through PINNED memory (torch's caching host allocator recycles
        the blocks, so this is an async DMA + one stream sync instead of a staged pageable copy)"""
        h = self.torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        h.copy_(t, non_blocking=True)
        # cooperative chunk pipeline (parallel.Pipeline): this is the point where the host would block on the
        # GPU, so the chunk hands the interpreter to the next chunk first and synchronises when it is back
        wait = getattr(self._tls, "before_sync", None)
        if wait is not None:
            wait()
        self.torch.cuda.current_stream().synchronize()
        return h

    def _vec(self, a):
        """host vector -> device, through pinned memory and WITHOUT blocking the host: a pageable copy on the
        (null) stream would wait for every kernel launched before it, i.e. serialise launch and host code"""
        t = self.torch
        a = np.ascontiguousarray(a, dtype=np.float64)
        h = t.empty(a.shape, dtype=t.float64, pin_memory=True)
        h.numpy()[...] = a
        return h.to(self.device, non_blocking=True)

    # ---- handles
    def counts(self, K):
        t = self.torch.as_tensor(np.ascontiguousarray(np.asarray(K, dtype=np.int32).T), device=self.device)
        return self.native.to_gene_major(t)

    def matrix(self, A):
        if A is None:
            return None
        t = self.torch.as_tensor(np.ascontiguousarray(np.asarray(A, dtype=np.float64).T), device=self.device)
        return self.native.to_gene_major(t)

    def design(self, x):
        """(p, m) device copy of the model matrix, memoised on its bytes (asked for by every fit of a DESeq())"""
        x = np.ascontiguousarray(x, dtype=np.float64)
        key = ("x", x.shape, x.tobytes())
        v = self._cache.get(key)
        if v is None:
            v = self._cache[key] = self.torch.as_tensor(np.ascontiguousarray(x.T), device=self.device)
            self._cells[v.data_ptr()] = self.native.cell_index(x)       # design cells, for the cell-collapsed fitBeta
        return v

    def _design_qr_dev(self, x):
        """device copies of Q, X R^-1, R of the (memoised) thin QR of the model matrix"""
        x = np.ascontiguousarray(x, dtype=np.float64)
        key = ("qr", x.shape, x.tobytes())
        v = self._cache.get(key)
        if v is None:
            t = self.torch
            q, a, r = self.native.design_qr(x)
            v = self._cache[key] = tuple(t.as_tensor(np.ascontiguousarray(z.T), device=self.device) for z in (q, a, r))
        return v

    def to_numpy(self, h):
        return self._host(h.view()).numpy()

    def to_numpy_np(self, b):
        """n x p host matrix from either a host array or the (p, n) device tensor of start values"""
        return self._host(b.t()).numpy() if self.torch.is_tensor(b) else np.asarray(b)

    def take_rows(self, h, idx):
        if h is None:
            return None
        ii = self.torch.as_tensor(np.asarray(idx), device=self.device)
        if ii.dtype == self.torch.bool:
            ii = ii.nonzero().squeeze(1)
        return self.native.GeneMajor(h.t.index_select(0, ii).contiguous(), h.m)

    def clamp_min(self, h, v):
        return self.native.GeneMajor(self.torch.clamp_min(h.t, v), h.m)

    def put_rows(self, h, idx, values):
        t = self.torch
        ii = t.as_tensor(np.asarray(idx), device=self.device)
        out = h.t.clone()
        out[ii, : h.m] = t.as_tensor(np.ascontiguousarray(values, dtype=np.float64), device=self.device)
        return self.native.GeneMajor(out, h.m)

    def set_rows(self, h, idx, sub):
        t = self.torch
        ii = t.as_tensor(np.asarray(idx), device=self.device)
        out = h.t.clone()
        out[ii] = sub.t
        return self.native.GeneMajor(out, h.m)

    def nrow(self, h):
        return h.n

    def vlog(self, v):
        return self.native.unary("log", np.asarray(v, np.float64))

    def vexp(self, v):
        return self.native.unary("exp", np.asarray(v, np.float64))

    # ---- observation weights: elementwise / flag work on the resident handle
    def any_negative(self, h):
        return bool((h.view() < 0).any())

    def row_max_normalize(self, h):
        t = self.torch
        out = t.zeros_like(h.t)
        out[:, : h.m] = h.view() / h.view().max(dim=1, keepdim=True).values
        return self.native.GeneMajor(out, h.m)

    def _prof_lock(self):
        return _PROF_LOCK

    def weights_ok(self, w, x, thr, full_rank):
        return self._host(self.weights_ok_dev(w, x, thr, full_rank)).numpy()

    def weights_ok_dev(self, w, x, thr, full_rank):
        """the flags of weights_ok as a device tensor (no host round trip)"""
        t = self.torch
        xd = t.as_tensor(np.ascontiguousarray(x, dtype=np.float64), device=self.device)
        wv = w.view()
        p = xd.shape[1]
        if not full_rank:
            ok = t.ones(w.n, dtype=t.bool, device=self.device)
            for j in range(p):
                ok &= ~((wv * xd[None, :, j]) == 0).all(dim=1)
            return ok

        def rank(G, tol=1e-7):               # _gram_rank (dqrdc2's column-relative test) on the device
            d0 = t.diagonal(G, dim1=1, dim2=2)
            C = t.zeros_like(G)
            rk = t.zeros(G.shape[0], dtype=t.int64, device=self.device)
            for j in range(p):
                r = d0[:, j] - (C[:, :, j] ** 2).sum(dim=1)
                ok = (d0[:, j] > 0) & (r >= (tol * tol) * d0[:, j])
                num = G[:, j, :] - t.einsum("nk,nkj->nj", C[:, :, j], C)
                den = t.sqrt(t.where(ok, r, t.ones_like(r)))[:, None]
                C[:, j, :] = t.where(ok[:, None], num / den, t.zeros_like(num))
                rk += ok
            return rk
        xx = (xd[:, :, None] * xd[:, None, :]).reshape(xd.shape[0], p * p)          # m x p^2
        G1 = ((wv * wv) @ xx).reshape(-1, p, p)
        keep = (wv > thr).to(t.float64)
        G2 = (keep @ xx).reshape(-1, p, p)
        ncol = ((keep @ xd.abs()) > 0).sum(dim=1)
        return (rank(G1) == p) & (rank(G2) == ncol)

    # ---- O(n*m) steps around the fits: HIP kernels too (csrc/aux.hip)
    def prefit(self, y, nf, x, weights=None):
        t = self.torch
        dq, da, dr = self._design_qr_dev(x)          # m x p design: thin QR on the host, like stats::qr
        o = self._timed("prefit_moments", y.n, lambda: self.native.prefitMoments_dev(
            y, nf, dq, da, dr, weights, weights is not None))
        h = self._host(o["_pack"])                        # one copy for the four n-vectors
        n = y.n
        return {"baseMean": h[0].numpy(), "baseVar": h[1].numpy(),
                "allZero": h[3].view(t.int32)[:n].numpy().astype(bool), "roughDisp": h[2].numpy(),
                "beta_init": o["beta_init"]}        # (p, n) device tensor, consumed by fit_beta in place

    def xim(self, nf):
        """momentsDispEstimate's xim (R/core.R:2440-2444) by the library's kernel: columns summed down the genes in gene
        order, the same definition the one-call host entry uses"""
        from . import _lib
        import ctypes as C
        t = self.torch
        buf = t.empty(nf.m + 1, dtype=t.float64, device=self.device)
        st = C.c_void_p(t.cuda.current_stream().cuda_stream)
        _lib.check(_lib.lib().dsq_xim_dev(C.c_void_p(nf.t.data_ptr()), nf.n, nf.m, nf.ld, C.c_void_p(buf.data_ptr()),
                                          C.c_void_p(buf[nf.m:].data_ptr()), st))
        return float(self._host(buf[nf.m:])[0])

    def weights_prep(self, w, x, thr=1e-2):
        """getAndCheckWeights (R/core.R:2697-2751) in ONE kernel on the resident weights: (w / rowmax, its 1e-6 floor,
        weightsFail flags (int32 device tensor), any-negative flag (int32 device tensor of one element))"""
        from . import _lib
        import ctypes as C
        t = self.torch
        xd = self.design(x)
        wn, wf = t.zeros_like(w.t), t.zeros_like(w.t)
        fz = t.empty(w.n, dtype=t.int32, device=self.device)
        neg = t.zeros(1, dtype=t.int32, device=self.device)
        st = C.c_void_p(t.cuda.current_stream().cuda_stream)
        _lib.check(_lib.lib().dsq_weights_prep_dev(C.c_void_p(w.t.data_ptr()), C.c_void_p(xd.data_ptr()), w.n, w.m,
                                                   int(np.asarray(x).shape[1]), w.ld, float(thr), C.c_void_p(wn.data_ptr()),
                                                   C.c_void_p(wf.data_ptr()), C.c_void_p(fz.data_ptr()),
                                                   C.c_void_p(neg.data_ptr()), st))
        GM = self.native.GeneMajor
        return GM(wn, w.m), GM(wf, w.m), fz, neg

    def linear_mu(self, y, nf, x_dev):
        dq, da, _ = self._design_qr_dev(x_dev.t().cpu().numpy())
        return self._timed("linear_mu", y.n, lambda: self.native.linearMu_dev(y, nf, dq, da))

    def nbinom_loglike(self, y, mu, disp, weights, useWeights):
        dv = self._vec(disp)
        o = self._timed("nbinom_loglike", y.n, lambda: self.native.nbinomLogLike_dev(y, mu, dv, weights, useWeights))
        return LaunchedVector(lambda: self._host(o).numpy())

    def optim_rows(self, y, nf, x, alpha, lam, weights, useWeights, beta_start, minmu):
        """the handful of rows the IRLS left: gathered rows go through the host-pointer entry point"""
        return self.native.optimRows(self.to_numpy(y), x, self.to_numpy(nf), alpha, lam,
                                     self.to_numpy(weights) if useWeights else None, useWeights, beta_start, minmu)

    def intercept_fit(self, y, nf, alpha, weights, useWeights, mu_floor=0.0, want_hat=True):
        av = self._vec(np.broadcast_to(np.asarray(alpha, float), (y.n,)))
        o = self._timed("intercept_fit", y.n, lambda: self.native.interceptFit_dev(
            y, nf, av, weights, useWeights, mu_floor, want_hat))
        h = self._host(o["_pack"]).numpy()
        return {"beta": h[0], "betaSE": h[1], "mu": o["mu"], "hat_diagonals": o["hat_diagonals"]}

    def parametric_fit(self, means, disps):
        dm, dd = self._vec(means), self._vec(disps)
        return self._timed("trend_fit", int(dm.numel()), lambda: self.native.parametricDispersionFit_dev(dm, dd))

    def mad(self, v):
        """stats::mad on the device (a sort; selecting order statistics is exact, so the value equals numpy's)"""
        t = self.torch
        x = self._vec(v)

        def med(z):
            s, _ = t.sort(z)
            k = s.numel()
            return s[k // 2] if k % 2 else (s[k // 2 - 1] + s[k // 2]) * 0.5
        m0 = med(x)
        return float(1.4826 * med((x - m0).abs()))

    def two_sided_normal_p(self, z):
        return self.native.unary("pnorm_upper2", np.asarray(z, np.float64))

    # ---- count outliers: HIP kernels (csrc/outlier.hip)
    def cooks_distance(self, y, nf, mu, H, x):
        cells = self.native.cell_index(x)
        p = np.asarray(x).shape[1]
        r = self._timed("cooks_distance", y.n, lambda: self.native.cooksDistance_dev(y, nf, mu, H, cells, p))
        h = self._host(r["_pack"]).numpy()
        return {"cooks": r["cooks"], "maxCooks": h[0], "robustDisp": h[1]}

    def replace_outliers(self, y, nf, cooks, cooksCutoff, replaceable, trim=0.2):
        r = self._timed("replace_outliers", y.n, lambda: self.native.replaceOutliers_dev(
            y, nf, cooks, cooksCutoff, replaceable, trim))
        return {"counts": r["counts"], "replace": self._host(r["replace"]).numpy().astype(bool)}

    def masked_row_max(self, h, use, zero):
        t = self.torch
        z = t.as_tensor(np.asarray(zero, bool), device=self.device)
        u = t.as_tensor(np.asarray(use, bool), device=self.device)
        a = t.where(z[None, :], t.zeros((), dtype=t.float64, device=self.device), h.view())
        return self._host(a[:, u].max(dim=1).values).numpy()      # max is exact: glue, not arithmetic

    # ---- the three native routines
    def fit_beta(self, y, x, nf, alpha_hat, contrast, beta_mat, lam, weights, useWeights, tol, maxit, useQR,
                 minmu, want_mu=True, mu_floor=0.0, want_hat=True):
        t = self.torch
        b0 = beta_mat if t.is_tensor(beta_mat) else self._vec(np.asarray(beta_mat).T)
        n, pp = y.n, len(lam)
        dv = self._vec(np.concatenate([np.broadcast_to(np.asarray(alpha_hat, float), (n,)), contrast, lam]))   # one upload
        av, cv, lv = dv[:n], dv[n:n + pp], dv[n + pp:]
        r = self._timed("fit_beta", y.n, lambda: self.native.fitBeta_dev(
            y, x, nf, av, cv, b0, lv, weights, useWeights, tol, maxit, useQR, minmu, want_hat=want_hat,
            want_mu=want_mu, mu_floor=mu_floor, cells=self._cells.get(x.data_ptr())))
        def fetch():
            h = self._host(r["_pack"]).numpy()           # one device-to-host copy for all per-gene outputs
            p = (h.shape[0] - 4) // 2
            # n x p matrices as column-major VIEWS of the host buffer (what R holds): no transpose copy, and the
            # row scans of fitNbinomGLMs (is.na / <= 0 per row) run along contiguous memory
            return {"beta_mat": h[:p].T, "beta_var_mat": h[p:2 * p].T, "iter": h[2 * p], "deviance": h[2 * p + 3],
                    "contrast_num": h[2 * p + 1].reshape(-1, 1), "contrast_denom": h[2 * p + 2].reshape(-1, 1)}
        return Launched({"hat_diagonals": r["hat_diagonals"], "mu": r["mu"]}, fetch)

    def fit_disp(self, y, x, mu_hat, log_alpha, prior_mean, prior_sigmasq, min_log_alpha, kappa_0, tol, maxit,
                 usePrior, weights, useWeights, weightThreshold, useCR):
        n = y.n
        dv = self._vec(np.concatenate([np.broadcast_to(np.asarray(log_alpha, float), (n,)),
                                       np.broadcast_to(np.asarray(prior_mean, float), (n,))]))    # one upload
        la, pm = dv[:n], dv[n:]
        r = self._timed("fit_disp", n, lambda: self.native.fitDisp_dev(
            y, x, mu_hat, la, pm, prior_sigmasq, min_log_alpha, kappa_0, tol, maxit, usePrior, weights, useWeights,
            weightThreshold, useCR, want_d2lp=self.want_d2lp, cells=self._cells.get(x.data_ptr())))
        def fetch():
            h = self._host(r.pop("_pack"))               # one copy; the views below index the host buffer
            keys = ("log_alpha", "last_change", "initial_lp", "initial_dlp", "last_lp", "last_dlp", "last_d2lp")
            out = {k: h[i].numpy() for i, k in enumerate(keys) if k in r}
            ints = h[len(keys)].view(self.torch.int32)
            out["iter"], out["iter_accept"] = ints[:n].numpy(), ints[n:].numpy()
            return out
        return Launched({}, fetch)

    def fit_disp_grid(self, y, x, mu_hat, disp_grid, prior_mean, prior_sigmasq, usePrior, weights, useWeights,
                      weightThreshold, useCR):
        n = y.n
        pm = self._vec(np.broadcast_to(np.asarray(prior_mean, float), (n,)))
        gv = self._vec(disp_grid)
        r = self._timed("fit_disp_grid", n, lambda: self.native.fitDispGrid_dev(
            y, x, mu_hat, gv, pm, prior_sigmasq, usePrior, weights, useWeights, weightThreshold, useCR,
            cells=self._cells.get(x.data_ptr())))
        return {"log_alpha": self._host(r["log_alpha"]).numpy()}
