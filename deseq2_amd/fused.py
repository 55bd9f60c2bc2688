"""DESeq() as ONE device-driven chain (dsq_deseq_dev, csrc/pipeline.hip).

core.DESeq() mirrors the reference's R callers one call at a time: every decision rule between two native calls
is host code on n-vectors, i.e. a device round trip.  Here the same rules run as kernels and the rows a rule sends
on (fitDispGrid stragglers, replaced-outlier rows) are compacted on the device, so a whole phase is enqueued
without a host decision -- including the rows that go to the reference's host-side fallback (fitNbinomGLMsOptim,
R/fitNbinomGLMs.R:340-407), which the library re-fits by a row-listed launch of its optim kernel.  The host looks at
the device ONCE per analysis, at the end: counters and the dispersion-trend scalars.  Results are bit-identical to core.DESeq() (tests/test_gpu_fused.py).

Supported: DeviceEngine, p <= 64, fitType = "parametric" / "mean", test = "Wald" (also with betaPrior = TRUE on the standard or
the expanded model matrix, and with useT) or "LRT" (any full-rank reduced model matrix), niter = 1, more than 3
residual degrees of freedom.  Anything else falls back to core.DESeq() / parallel.DESeqParallel().
"""
import ctypes as C

import numpy as np

from . import _lib as L
from . import core


def supported(dds, test="Wald", reduced=None, fitType="parametric", **kw):
    E = dds.engine
    if getattr(E, "name", "") != "device" or not (fitType in ("parametric", "mean") or callable(fitType)):
        return False
    if callable(fitType) and kw.get("betaPrior"):
        return False
    if dds.p > L.DSQ_MAX_P or dds.m <= dds.p:
        return False
    # (wide designs, 10 < p <= 64 on the zero-padded kernel builds, take everything the narrow ones do since round 5:
    #  observation weights -- csrc/aux.hip weights_prep_wide_kernel --, reduced models and beta-prior passes of more than 10
    #  columns, each design at its own padded width in csrc/pipeline.hip)
    # the preconditions core.estimateDispersionsGeneEst raises on (rank, R/core.R:2624) and the residual-df <= 3
    # branch of estimateDispersionsPriorVar (seeded Monte-Carlo matching, R/core.R:1155-1190: not mirrored, core raises
    # NotImplementedError) are left to the call-by-call code, which reports them
    # ... unless the caller brings the prior variance (estimateDispersionsMAP(dispPriorVar = x), R/core.R:989-994)
    if (dds.m - dds.p <= 3 and not kw.get("dispPriorVar")) or dds.m <= dds.p or core._rank(dds.x) < dds.p:
        return False
    if kw.get("modelMatrix") is not None or not kw.get("useOptim", True):
        return False
    if kw.get("betaPrior"):
        # nbinomWaldTest(betaPrior = TRUE): the MLE pass, the all-gene prior variance (host), the pass with the ridge
        from . import parallel
        if test != "Wald" or _prior_design(dds, kw) is None:
            return False
        # gene shards: R/parallel.R:34-40 takes the MLE coefficients for the prior variance from a fit WITHOUT the
        # observation weights (estimateMLEForBetaPriorVar) -- not the chain's MLE pass: left to DESeqParallel
        # (... and on that function's defaults: minmu 0.5, betaTol 1e-8, maxit 100, QR)
        if parallel.world_size() > 1 and (dds.has_weights or kw.get("minmu", 0.5) != 0.5 or kw.get("betaTol", 1e-8) != 1e-8 or
                                          kw.get("maxit", 100) != 100 or not kw.get("useQR", True)):
            return False
    if kw.get("useT") and test != "Wald":
        return False
    if set(kw) - {"betaPrior", "betaPriorVar", "modelMatrixType", "factors", "modelMatrix", "useT", "df", "useOptim", "betaTol",
                  "maxit", "useQR", "minmu", "disp_maxit", "minReplicatesForReplace", "dispPriorVar"}:
        return False
    if test == "LRT":
        # reduced = ~1 takes the closed form (R/fitNbinomGLMs.R:99-137); any other reduced model matrix is fitted by the
        # IRLS like the full one -- full rank (the rank-deficient start values of :146-155 are left to core.py)
        r = None if reduced is None else np.asarray(reduced, np.float64)
        if r is None or r.ndim != 2 or r.shape[0] != dds.m or not (1 <= r.shape[1] < dds.p):
            return False
        if not (np.abs(r).sum(axis=0) > 0).all() or (not _intercept_only(r) and core._rank(r) < r.shape[1]):
            return False
    elif test != "Wald":
        return False
    return True


def _prior_design(dds, kw):
    """(model matrix of the beta-prior pass, its type, coefficient names) -- R/core.R:1374-1380, R/fitNbinomGLMs.R:311-325 --
    or None when the chain does not take it (more than DSQ_MAX_P columns; cells that differ from the design's)"""
    factors = kw.get("factors")
    mmt = kw.get("modelMatrixType") or ("expanded" if factors is not None else "standard")
    if mmt == "expanded":
        if factors is None:
            return None
        xe, _ = core.makeExpandedModelMatrix(factors)
    else:
        xe = dds.x
    names = core.standard_model_matrix(factors)[1] if factors is not None else ["Intercept"] + ["V%d" % i for i in range(1, dds.p)]
    xe = np.ascontiguousarray(xe, dtype=np.float64)
    if xe.shape[0] != dds.m or xe.shape[1] > L.DSQ_MAX_P or len(names) != dds.p:
        return None
    ca, cb = core._cells(dds.x)[0], core._cells(xe)[0]
    if len(set(zip(ca.tolist(), cb.tolist()))) != len(set(ca.tolist())) or len(set(cb.tolist())) != len(set(ca.tolist())):
        return None
    return xe, mmt, names


def _intercept_only(r):
    return r.shape[1] == 1 and bool((r == 1).all())


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


_FACTS = {}


def _design_facts(E, x, minReplicatesForReplace):
    """what the chain needs to know about the DESIGN alone (cells, replaceable samples, the Cook's cutoff quantile,
    whether the design is one group per column, trigamma((m - p) / 2)): host work, memoised per design like the QR"""
    key = (x.shape, x.tobytes(), float(minReplicatesForReplace))
    f = _FACTS.get(key)
    if f is None:
        from scipy.stats import f as fdist
        from scipy import special as sps
        m, p = x.shape
        finite = bool(np.isfinite(minReplicatesForReplace))
        rep = core.nOrMoreInCell(x, minReplicatesForReplace) if finite else np.zeros(m, bool)
        f = dict(cells=np.ascontiguousarray(E.native.cell_index(x), dtype=np.int32),
                 do_replace=bool(finite and rep.any()),
                 replaceable=np.ascontiguousarray(rep.astype(np.int32)),
                 cutoff=float(fdist.ppf(.99, p, m - p)),
                 groups_eq_p=bool(len(np.unique(core.modelMatrixGroups(x))) == p),
                 expVarLogDisp=float(sps.polygamma(1, (m - p) / 2.0)) if m > p else 0.0)
        if len(_FACTS) > 16:
            _FACTS.clear()
        _FACTS[key] = f
    return f


class _Run:
    """buffers + argument block of one analysis"""

    def __init__(self, dds, test, minReplicatesForReplace, n_trend, kw, reduced=None):
        E = dds.engine
        t = self.t = E.torch
        # (no reference back to `dds`: DESeq() below hangs this object on dds._fused_run, and a dds <-> run cycle would
        #  keep the ~26 n x m bytes per gene of device buffers -- and the pinned result block -- alive until Python's
        #  CYCLIC collector gets to them, so that every step of a loop would miss torch's caching allocators and pay
        #  fresh hipMalloc / hipHostMalloc calls; with the plain reference the buffers go back to the caches the moment
        #  the caller drops the object)
        self.E = E
        dev = E.device
        n, m, p, ld = dds.n, dds.m, dds.p, dds.y.ld
        self.n, self.m, self.p, self.ld = n, m, p, ld
        self.test = test
        f64 = dict(dtype=t.float64, device=dev)
        i32 = dict(dtype=t.int32, device=dev)
        # ---- weights (getAndCheckWeights, R/core.R:2697-2751), all on the device
        self.useWeights = bool(dds.has_weights)
        self.neg = None
        self.force_zero = None
        if self.useWeights:
            self.w_norm, self.w_floor, self.force_zero, self.neg = E.weights_prep(dds.weights_h, dds.x, 1e-2)
        # ---- outputs: every per-gene column in ONE device block (a single copy brings them all to the host):
        #      vec (10 x n f64) | mat (4 x pcol x n f64) | [mle (p x n f64)] | scalars | ivec (9 x n i32) | status
        self.prior = _prior_design(dds, kw) if kw.get("betaPrior") else None
        pcol = self.prior[0].shape[1] if self.prior is not None else p      # columns of beta / betaSE / stat / pvalue
        nmle = p * n if self.prior is not None else 0
        nm_ = self._nmat = 4 if test == "Wald" else 2                      # the LRT has no per-coefficient stat / pvalue
        nd = 10 * n + nm_ * pcol * n + nmle + L.DSQ_SC_COUNT
        ni = 9 * n + L.DSQ_ST_COUNT + 2
        self.blob = t.empty(nd * 8 + ni * 4, dtype=t.uint8, device=dev)
        dpart, ipart = self.blob[: nd * 8].view(t.float64), self.blob[nd * 8:].view(t.int32)
        self._nd, self._ni, self._pcol, self._nmle = nd, ni, pcol, nmle
        self.vec = dpart[: 10 * n].view(10, n)
        (self.baseMean, self.baseVar, self.dispGeneEst, self.dispFit, self.dispMAP, self.dispersion, self.betaIter,
         self.logLike, self.logLikeReduced, self.maxCooks) = self.vec
        self.mat = dpart[10 * n: 10 * n + nm_ * pcol * n].view(nm_, pcol, n)   # beta, betaSE, stat, pvalue: (p, n) = column-major n x p
        self.mle = dpart[10 * n + nm_ * pcol * n: 10 * n + nm_ * pcol * n + nmle].view(p, n) if nmle else None
        self.scalars = dpart[nd - L.DSQ_SC_COUNT:]
        self.ivec = ipart[: 9 * n].view(9, n)
        self.status = ipart[9 * n: 9 * n + L.DSQ_ST_COUNT]
        self.negflag = ipart[9 * n + L.DSQ_ST_COUNT:]
        (self.allZero, self.dispGeneIter, self.dispIter, self.dispOutlier, self.betaConv, self.replace,
         self.optim_geneest, self.optim_test, _) = self.ivec
        if self.neg is not None:
            self.negflag[:1] = self.neg
        self.mu_hat = t.empty((n, ld), **f64)
        self.mu = t.empty((n, ld), **f64)
        self.H = t.empty((n, ld), **f64)
        self.cooks = t.empty((n, ld), **f64)
        self.replaceCounts = t.empty((n, ld), **i32)
        lib = L.lib()
        wsb = int(lib.dsq_deseq_workspace_bytes(n, m, max(p, pcol), int(n_trend)))
        self.workspace = t.empty(wsb, dtype=t.uint8, device=dev)
        # ---- design facts (host, memoised per design)
        x = dds.x
        dq, da, dr = E._design_qr_dev(x)
        self.keep = [dq, da, dr]
        minDisp = 1e-8
        self.minDisp = minDisp
        self.grid = E._cache.get(("grid", m))                                       # R/wrappers.R:70-72
        if self.grid is None:
            self.grid = E._cache[("grid", m)] = E._vec(np.linspace(np.log(1e-8), np.log(max(10, m)), 20))
        self.lam = np.ascontiguousarray(np.full(p, 1e-6) / np.log(2) ** 2)          # R/fitNbinomGLMs.R:73,162
        # (a normalization-factor matrix: the chain takes mean(1 / colMeans(nf)) over its non-zero rows itself)
        xim = core.xim_size_factors(dds.sizeFactors) if dds.sizeFactors is not None else 0.0
        facts = _design_facts(E, x, minReplicatesForReplace)
        cells = facts["cells"]
        self.cells = cells
        do_replace = facts["do_replace"]
        self.do_replace = do_replace
        self.replaceable = facts["replaceable"]
        cutoff = facts["cutoff"]                                                     # R/core.R:2081
        linearMu = facts["groups_eq_p"] and not self.useWeights                      # :735-742
        # size factors: every kernel of the chain reads the m-vector (same values as the rows of the n x m matrix R
        # builds from them, R/core.R:2221-2227, so the same bits) -- 8 B less per sample and pass, no layout conversion
        if dds.sizeFactors is not None:
            sfh = np.ascontiguousarray(dds.sizeFactors, np.float64)
            key = ("sf", sfh.tobytes())
            self.sf_dev = E._cache.get(key)
            if self.sf_dev is None:
                if sum(1 for k in E._cache if k[0] == "sf") > 8:
                    for k in [k for k in E._cache if k[0] == "sf"]:
                        del E._cache[k]
                self.sf_dev = E._cache[key] = E._vec(sfh)
            nf_ptr, nf_vec = _ptr(self.sf_dev), 1
        else:
            nf_ptr, nf_vec = _ptr(dds.nf.t), 0
        self.args = L.DsqDeseqArgs(
            n=n, m=m, p=p, ld=ld, phases=0, y=_ptr(dds.y.t), nf=nf_ptr, nf_is_vector=nf_vec,
            useWeights=int(self.useWeights),
            weights_raw=_ptr(dds.weights_h.t) if self.useWeights else None,
            weights_norm=_ptr(self.w_norm.t) if self.useWeights else None,
            weights_floor=_ptr(self.w_floor.t) if self.useWeights else None,
            force_zero=_ptr(self.force_zero), x=_ptr(dds.xh), q=_ptr(dq), a=_ptr(da), r=_ptr(dr), xim=xim,
            linearMu=int(bool(linearMu)), minDisp=minDisp, kappa_0=1.0, dispTol=1e-6, weightThreshold=1e-2, outlierSD=2.0,
            betaTol=kw.get("betaTol", 1e-8), minmu=kw.get("minmu", 0.5), maxit=int(kw.get("disp_maxit", 100)),
            useCR=int(kw.get("useCR", True)), useQR=int(kw.get("useQR", True)), betaMaxit=int(kw.get("maxit", 100)),
            disp_grid=_ptr(self.grid), ngrid=20,
            expVarLogDisp=facts["expVarLogDisp"],
            trend_mean=None, trend_disp=None, n_trend=int(n_trend), lambda_=self.lam.ctypes.data_as(C.c_void_p),
            min_log_alpha=float(np.log(minDisp / 10)), workspace=_ptr(self.workspace), workspace_bytes=wsb,
            test=0 if test == "Wald" else 1, cell_of=self.cells.ctypes.data_as(C.c_void_p),
            ncell=int(cells.max()) + 1, replaceable=self.replaceable.ctypes.data_as(C.c_void_p), cooksCutoff=cutoff,
            trim=0.2, do_replace=int(do_replace))
        self.cooksCutoff = cutoff
        if self.prior is not None:
            xe = self.prior[0]
            xp = E.design(xe)
            self.keep.append(xp)
            self.args.betaPrior, self.args.x_prior, self.args.p_prior = 1, _ptr(xp), int(xe.shape[1])
            self.args.prior_expanded = int(core._rank(xe) < xe.shape[1])
            self.args.prior_intercept = int(bool((xe[:, 0] == 1).all()))
        self.p_red = 1
        if test == "LRT" and reduced is not None and not _intercept_only(np.asarray(reduced, np.float64)):
            red = np.ascontiguousarray(reduced, dtype=np.float64)
            xr = E.design(red)
            rq, ra, rr = E._design_qr_dev(red)
            self.red_cells = np.ascontiguousarray(E.native.cell_index(red), dtype=np.int32)
            self.keep += [xr, rq, ra, rr]
            self.p_red = red.shape[1]
            self.args.x_red, self.args.q_red, self.args.a_red, self.args.r_red = _ptr(xr), _ptr(rq), _ptr(ra), _ptr(rr)
            self.args.p_red = int(self.p_red)
            self.args.cell_of_red = self.red_cells.ctypes.data_as(C.c_void_p)
            self.args.ncell_red = int(self.red_cells.max()) + 1
        self.out = L.DsqDeseqOut(
            baseMean=_ptr(self.baseMean), baseVar=_ptr(self.baseVar), allZero=_ptr(self.allZero),
            dispGeneEst=_ptr(self.dispGeneEst), dispGeneIter=_ptr(self.dispGeneIter), dispFit=_ptr(self.dispFit),
            dispMAP=_ptr(self.dispMAP), dispersion=_ptr(self.dispersion), dispIter=_ptr(self.dispIter),
            dispOutlier=_ptr(self.dispOutlier), beta=_ptr(self.mat[0]), betaSE=_ptr(self.mat[1]),
            stat=_ptr(self.mat[2]) if test == "Wald" else None, pvalue=_ptr(self.mat[3]) if test == "Wald" else None,
            betaConv=_ptr(self.betaConv), betaIter=_ptr(self.betaIter), logLike=_ptr(self.logLike),
            logLikeReduced=_ptr(self.logLikeReduced), maxCooks=_ptr(self.maxCooks), replace=_ptr(self.replace),
            optim_geneest=_ptr(self.optim_geneest), optim_test=_ptr(self.optim_test), mu_hat=_ptr(self.mu_hat),
            mu=_ptr(self.mu), H=_ptr(self.H), cooks=_ptr(self.cooks), replaceCounts=_ptr(self.replaceCounts),
            status=_ptr(self.status), scalars=_ptr(self.scalars), mle_beta=_ptr(self.mle))

    def launch(self, phases, trend=None):
        self.args.phases = int(phases)
        if trend is not None:
            assert int(trend[0].numel()) == int(self.args.n_trend)
            self.args.trend_mean, self.args.trend_disp = _ptr(trend[0]), _ptr(trend[1])
            self._trend = trend
        stream = C.c_void_p(self.t.cuda.current_stream().cuda_stream)
        E = self.E
        if E.record is not None:
            lib = L.lib()
            with E._prof_lock():
                lib.dsq_profile_enable(1)
                L.check(lib.dsq_deseq_dev(C.byref(self.args), C.byref(self.out), stream))
                nm = C.create_string_buffer(40)
                g, ms = C.c_int32(0), C.c_double(0.0)
                for i in range(lib.dsq_profile_count()):
                    lib.dsq_profile_get(i, nm, 40, C.byref(g), C.byref(ms))
                    E.record.append((nm.value.decode(), int(g.value), ms.value))
                lib.dsq_profile_enable(0)
            return
        L.check(L.lib().dsq_deseq_dev(C.byref(self.args), C.byref(self.out), stream))

    _side = {}          # device -> side stream of the early copy of the log likelihoods

    def early_loglike(self):
        """nbinomLRT: start bringing logLike / logLikeReduced to the host as soon as the test's fits are enqueued, on a
        side stream, so that pchisq over all genes (host code, R/core.R:1878; ~3 ms per 60 000 genes) runs while the
        device works through the outlier phase.  The rows that phase refits are still being rewritten while this copy
        is in flight: whatever the copy saw of them is checked against the final values and recomputed (DESeq below)."""
        t = self.t
        key = str(self.E.device)
        stream = _Run._side.get(key)
        if stream is None:
            stream = _Run._side[key] = t.cuda.Stream(device=self.E.device)
        # (a pinned buffer per analysis, from torch's caching host allocator: with wait = False two analyses are in flight)
        host = t.empty(2 * self.n, dtype=t.float64, pin_memory=True)
        ready = t.cuda.Event()
        ready.record()
        stream.wait_event(ready)
        with t.cuda.stream(stream):
            host.copy_(self.vec[7:9].reshape(-1), non_blocking=True)
            done = t.cuda.Event()
            done.record(stream)
        self._early_done = done
        return host, done

    def read_status(self):
        """ONE small device-to-host copy + stream sync: counters and scalars of the phases run so far"""
        t = self.t
        parts = [self.status.to(t.float64), self.scalars]
        if self.neg is not None:
            parts.append(self.neg.to(t.float64))
        h = self.E._host(t.cat(parts)).numpy()
        st = {k: int(h[i]) for k, i in L.DSQ_ST.items()}
        sc = h[L.DSQ_ST_COUNT: L.DSQ_ST_COUNT + L.DSQ_SC_COUNT]
        if self.neg is not None and h[-1] != 0:
            raise ValueError("all(weights >= 0) is not TRUE")
        return st, sc

    _copy_stream = {}   # device -> the stream the result block travels on

    def start_read(self):
        """enqueue the ONE device-to-host copy of the result block (counters, scalars, every per-gene column) behind
        everything enqueued so far -- on a side stream, into pinned memory, without blocking the host: whatever the
        caller enqueues next on the chain's stream (the next analysis of a pipelined loop) runs beside the copy"""
        t = self.t
        key = str(self.E.device)
        side = _Run._copy_stream.get(key)
        if side is None:
            side = _Run._copy_stream[key] = t.cuda.Stream(device=self.E.device)
        ready = getattr(self, "_ready", None)
        if ready is None:
            ready = t.cuda.Event()
            ready.record()
        side.wait_event(ready)
        host = t.empty(self.blob.shape, dtype=t.uint8, pin_memory=True)
        with t.cuda.stream(side):
            host.copy_(self.blob, non_blocking=True)
            done = t.cuda.Event()
            done.record(side)
        self._pending = (host, done, ready)

    def mark_ready(self):
        """wait = False: the point of the chain's stream behind which the result block is complete.  The copy itself is
        enqueued by read_all() -- at that time, in a loop of analyses, this point has usually been passed, so the side stream
        never holds more than one copy, and none that waits (measured: with the copy of analysis k enqueued behind a
        pending event while that of k - 1 had not run, hipMemcpyAsync once blocked for 5 ms -- a whole step at the small
        configurations)."""
        self._ready = self.t.cuda.Event()
        self._ready.record()

    def __del__(self):
        # an analysis dropped between start_read() and read_all(): its result block goes back to torch's allocator (and
        # to the next analysis on the chain's stream) only when the copy on the side stream has read it.  (Waiting here
        # instead of blob.record_stream(side) keeps the block's reuse deterministic: a loop over analyses does not grow
        # the pool by a block whenever a copy happens to be still in flight.)
        try:
            pend = getattr(self, "_pending", None)
            if pend is not None:
                pend[1].synchronize()
            early = getattr(self, "_early_done", None)
            if early is not None:
                early.synchronize()
        except Exception:                                            # noqa: BLE001  (interpreter shutdown)
            pass

    def read_all(self):
        """the whole result block on the host: waits for start_read()'s copy (enqueues it first if nobody has)"""
        n, pc = self.n, self._pcol
        if getattr(self, "_pending", None) is None:
            self.start_read()
        host, done, ready = self._pending
        self._pending = None
        wait = getattr(self.E._tls, "before_sync", None)    # (cooperative chunk pipeline: see DeviceEngine._host)
        if wait is not None:
            wait()
        done.synchronize()
        ready.synchronize()     # (complete by now: lets the runtime retire the chain's commands on ITS stream as well -- a loop
                                #  of wait = False analyses never synchronises that stream otherwise)
        h = host.numpy()
        hd, hi = h[: self._nd * 8].view(np.float64), h[self._nd * 8:].view(np.int32)
        st = {k: int(hi[9 * n + i]) for k, i in L.DSQ_ST.items()}
        if self.neg is not None and hi[9 * n + L.DSQ_ST_COUNT] != 0:
            raise ValueError("all(weights >= 0) is not TRUE")
        hv = hd[: 10 * n].reshape(10, n)
        nm_ = self._nmat
        hm = hd[10 * n: 10 * n + nm_ * pc * n].reshape(nm_, pc, n)
        hmle = hd[10 * n + nm_ * pc * n: 10 * n + nm_ * pc * n + self._nmle].reshape(self.p, n) if self._nmle else None
        return st, hd[self._nd - L.DSQ_SC_COUNT:], hv, hm, hi[: 9 * n].reshape(9, n), hmle


_DEVICE_REDUCE_OK = None      # the device all-reduce of N_REFIT: None = not yet checked against the host exchange
_DEVICE_REDUCE_FOR = None     # ... and the communicator that verdict was reached on


def _global_refit_count(run, comm_device, t):
    """refitWithoutOutliers' closing steps ask whether ANY row of the whole object was refitted (R/core.R:2496): the
    shards add up their counts.  With a device communicator (RCCL) the sum is an all-reduce of the one device counter,
    enqueued behind the chain -- no host look at the device in the middle of the analysis.  The route is decided ONCE per
    communicator, by collectives every rank executes: the first call runs both exchanges and the ranks agree on whether
    the device sum matched the host sum.  A collective that fails raises on the rank it fails on (no per-rank change of
    route: the other ranks would be left waiting in a collective this one never joins).  Ranks sharing a device (tests):
    through the host."""
    global _DEVICE_REDUCE_OK, _DEVICE_REDUCE_FOR
    from . import parallel
    if comm_device is not None:
        import torch.distributed as dist
        key = (id(dist.distributed_c10d._get_default_group()), str(comm_device))
        if _DEVICE_REDUCE_FOR != key:
            _DEVICE_REDUCE_OK, _DEVICE_REDUCE_FOR = None, key
        if _DEVICE_REDUCE_OK is not False:
            tot = run.status[L.DSQ_ST["N_REFIT"]: L.DSQ_ST["N_REFIT"] + 1].to(t.int64)
            dist.all_reduce(tot)
            dev_total = tot.clamp(max=2 ** 31 - 1).to(t.int32)
            if _DEVICE_REDUCE_OK is None:
                st, _ = run.read_status()
                host_total = sum(parallel.allgather_sizes(st["N_REFIT"], comm_device))
                mine = int(dev_total.item()) == min(host_total, 2 ** 31 - 1)
                _DEVICE_REDUCE_OK = all(parallel.allgather_sizes(int(mine), comm_device))
            if _DEVICE_REDUCE_OK:
                return dev_total
    st, _ = run.read_status()
    total = sum(parallel.allgather_sizes(st["N_REFIT"], comm_device))
    return t.tensor([min(total, 2 ** 31 - 1)], dtype=t.int32, device=run.E.device)


def finish(dds):
    """the second half of DESeq(dds, wait=False): waits for the result block and fills in mcols / assays / attrs"""
    f = dds.__dict__.pop("_fused_pending", None)
    return f(dds) if f is not None else dds


def DESeq(dds, test="Wald", fitType="parametric", reduced=None, minReplicatesForReplace=7, comm_device=None,
          shard_sizes=None, wait=True, **kw):
    """core.DESeq() / parallel.DESeqParallel() semantics (R/core.R:280-432, R/parallel.R:6-74) on the fused device
    chain.  With torch.distributed initialised, `dds` is this rank's gene shard and the dispersion trend is fitted
    over the gathered (baseMean, dispGeneEst) of all ranks.

    wait = False (single process): returns as soon as the chain is ENQUEUED; the object is complete after
    fused.finish(dds), which enqueues the copy of the result block (side stream), waits for it and builds the columns --
    input errors the device detects (negative weights, all-zero counts, a trend that does not fit) surface THERE.  A loop over analyses then keeps the device busy while the host
    prepares the next one and post-processes the previous one, and the result copy (a side stream) runs beside the
    next chain's kernels -- bench.py's pipelined steps."""
    if not supported(dds, test=test, reduced=reduced, fitType=fitType, minReplicatesForReplace=minReplicatesForReplace, **kw):
        from . import parallel
        if parallel.world_size() > 1:
            return parallel.DESeqParallel(dds, test=test, fitType=fitType, reduced=reduced, comm_device=comm_device,
                                          minReplicatesForReplace=minReplicatesForReplace, **kw)
        return core.DESeq(dds, test=test, fitType=fitType, reduced=reduced,
                          minReplicatesForReplace=minReplicatesForReplace, **kw)
    from . import parallel
    E = dds.engine
    t = E.torch
    world = parallel.world_size()
    n = dds.n
    n_all = n
    if world > 1:
        # (the caller may pass the shard sizes of all ranks -- bench.py cuts the shards itself -- and save the exchange)
        sizes = list(shard_sizes) if shard_sizes is not None else parallel.allgather_sizes(n, comm_device)
        assert len(sizes) == world and sizes[parallel.rank()] == n, "shard_sizes: one entry per rank, this rank's = n"
        n_all = int(max(sizes)) * world
    run = _Run(dds, test, minReplicatesForReplace, n_all if world > 1 else 0, kw, reduced=reduced)
    # estimateDispersionsFit (R/core.R:864-939): "mean" on the device; a parametric trend that does not fit is replaced
    # by the mean there as well (core.estimateDispersionsFit's substitute for the reference's locfit fallback)
    run.args.fitType = L.DSQ_FIT["mean" if fitType == "mean" else "parametric_or_mean"]
    if kw.get("dispPriorVar"):
        run.args.dispPriorVar_in = float(kw["dispPriorVar"])
    custom = None
    if callable(fitType):
        # the caller's trend (core.estimateDispersionsFit: what R has after fitType = "local" or dispersionFunction<-):
        # the gene-wise estimates come up, the function is evaluated on the host, its values go down as dispFit_in.  The
        # refit of replaced rows needs the function at their NEW means (R/core.R:2512): the outlier phase then runs in its two
        # halves with one more look at the device in between (below).  Gene shards go call by call.
        if world > 1:
            return parallel.DESeqParallel(dds, test=test, fitType=fitType, reduced=reduced, comm_device=comm_device,
                                          minReplicatesForReplace=minReplicatesForReplace, **kw)
        run.launch(L.DSQ_PH_GENE_EST)
        hb = E._host(t.stack([run.baseMean, run.dispGeneEst])).numpy()
        with np.errstate(invalid="ignore"):
            use = hb[1] > 100 * 1e-8
        if not use.any():
            raise RuntimeError("all gene-wise dispersion estimates are within 2 orders of magnitude from the minimum value")
        custom = fitType(hb[0][use], hb[1][use])
        with np.errstate(invalid="ignore", divide="ignore"):
            run._fit_in = t.as_tensor(np.ascontiguousarray(custom(hb[0]), dtype=np.float64), device=E.device)
        run.args.dispFit_in = _ptr(run._fit_in)

    # Everything is enqueued without a host decision: the rows a rule sends on -- fitDispGrid stragglers, rows for the
    # optim fallback (R/fitNbinomGLMs.R:203-227), replaced-outlier rows -- are row-listed launches whose lengths live
    # on the device.  ONE look at the counters at the end (multi-GPU: one more before the all-gather of the trend's
    # input vectors).
    bpv = None
    early = None
    if run.prior is not None:
        # betaPrior: the MLE pass, then ONE extra look at the device -- estimateBetaPriorVar (R/core.R:1601-1689) is an
        # all-gene weighted quantile of the MLE coefficients, host code on n x p values -- then the pass with the ridge.
        # Gene shards (R/parallel.R:30-48): the shards hand over their MLE coefficients, baseMean and dispFit and every
        # rank computes the same prior variance over all rows.
        if world == 1:
            run.launch(L.DSQ_PH_GENE_EST | L.DSQ_PH_TREND | L.DSQ_PH_MAP_TEST)
        else:
            run.launch(L.DSQ_PH_GENE_EST)
            trend = parallel.allgather_device_pairs(run.baseMean, run.dispGeneEst, max(sizes), comm_device, t)
            run.args.defer_finish = 1
            run.launch(L.DSQ_PH_TREND | L.DSQ_PH_MAP_TEST, trend=trend)
        st0, sc0 = run.read_status()
        nnz0 = st0["N_NONZERO"] if world == 1 else sum(parallel.allgather_sizes(st0["N_NONZERO"], comm_device))
        if nnz0 == 0:
            raise ValueError("all genes have zero counts in every sample")
        if st0["N_TREND"] == 0 or st0["TREND_STATUS"] != 0 or st0["N_ABOVE_MIN"] == 0:
            if world > 1:
                return parallel.DESeqParallel(dds, test=test, fitType=fitType, reduced=reduced, comm_device=comm_device,
                                              minReplicatesForReplace=minReplicatesForReplace, **kw)
            return core.DESeq(dds, test=test, fitType=fitType, reduced=reduced,
                              minReplicatesForReplace=minReplicatesForReplace, **kw)
        bpv = kw.get("betaPriorVar")
        if bpv is None:
            h = E._host(t.cat([run.mle, run.baseMean[None], run.dispFit[None], run.allZero[None].to(t.float64)])).numpy()
            if world > 1:
                h = np.stack([parallel._allgather_vec(np.ascontiguousarray(r), comm_device) for r in h])
            nzr = h[-1] == 0
            view = type("V", (), {"mcols": {"baseMean": h[dds.p][nzr], "dispFit": h[dds.p + 1][nzr]}})()
            def dev_sort(v):           # the (unique) stable order, sorted on the device: the host has nothing else to do
                return E._host(t.sort(t.as_tensor(v, device=E.device), stable=True).indices).numpy()
            bpv, _ = core.estimateBetaPriorVar(view, h[:dds.p].T[nzr], run.prior[2], modelMatrixType=run.prior[1],
                                               factors=kw.get("factors"), sorter=dev_sort)
        bpv = np.asarray(bpv, np.float64)
        if (bpv == 0).any():
            raise ValueError("beta prior variances are equal to zero for some variables")
        run.lam_prior = np.ascontiguousarray((1.0 / bpv) / np.log(2) ** 2)                 # R/fitNbinomGLMs.R:311,162
        run.args.lambda_prior = run.lam_prior.ctypes.data_as(C.c_void_p)
        run.launch(L.DSQ_PH_PRIOR | L.DSQ_PH_OUTLIERS)
    elif custom is not None and run.do_replace:
        run.launch(L.DSQ_PH_TREND | L.DSQ_PH_MAP_TEST | L.DSQ_PH_OUTLIERS_DETECT)
        h2 = E._host(t.stack([run.baseMean, run.replace.to(t.float64), run.allZero.to(t.float64)])).numpy()
        rows = np.flatnonzero((h2[1] != 0) & (h2[2] == 0))                # refitReplace, R/core.R:2496-2498
        if rows.size:
            with np.errstate(invalid="ignore", divide="ignore"):
                vals = np.ascontiguousarray(custom(h2[0][rows]), dtype=np.float64)
            run._fit_in[t.as_tensor(rows, device=E.device)] = t.as_tensor(vals, device=E.device)
        run.launch(L.DSQ_PH_OUTLIERS_REFIT)
    elif custom is not None:
        run.launch(L.DSQ_PH_TREND | L.DSQ_PH_MAP_TEST | L.DSQ_PH_OUTLIERS)
    elif world == 1 and test == "LRT" and E.record is None and dds.n >= 4096:
        run.launch(L.DSQ_PH_GENE_EST | L.DSQ_PH_TREND | L.DSQ_PH_MAP_TEST)
        early_host, early_done = run.early_loglike()
        run.launch(L.DSQ_PH_OUTLIERS)
        p_full = dds.p                        # (the closure below must not hold `dds`: see _finish)

        def early():                          # (first thing of _finish below)
            early_done.synchronize()
            eh = early_host.numpy()
            e = 2 * (eh[:n] - eh[n:])
            return e, core.pchisq_upper(e, p_full - run.p_red)            # (the device is in the outlier phase meanwhile)
    elif world == 1:
        run.launch(L.DSQ_PH_GENE_EST | L.DSQ_PH_TREND | L.DSQ_PH_MAP_TEST | L.DSQ_PH_OUTLIERS)
    else:
        run.launch(L.DSQ_PH_GENE_EST)
        trend = parallel.allgather_device_pairs(run.baseMean, run.dispGeneEst, max(sizes), comm_device, t)
        run.args.defer_finish = 1
        run.launch(L.DSQ_PH_TREND | L.DSQ_PH_MAP_TEST | L.DSQ_PH_OUTLIERS, trend=trend)
    if world > 1 and run.do_replace:
        # refitWithoutOutliers' closing steps (NA results on rows that became all zero, maxCooks) need the GLOBAL count of
        # refitted rows; then each shard finishes its own rows
        run.n_refit_all = _global_refit_count(run, comm_device, t)
        run.args.n_refit_global = _ptr(run.n_refit_all)
        run.launch(L.DSQ_PH_FINISH)
    def _finish(dds):
        # (`dds` is a PARAMETER: the closure holds the run and the call's settings, not the data set -- an analysis enqueued
        #  with wait = False hangs this function on dds._fused_pending, and a dds -> closure -> dds cycle would keep its
        #  n x m device buffers until the cyclic collector runs)
        early_v = early() if early is not None else None
        st, sc, hv, hm, hi, hmle = run.read_all()
        st2 = st
        # R/parallel.R fits the trend on the gathered object: a shard whose rows are all zero is legal as long as some
        # rank holds counts (N_TREND / TREND_STATUS / N_ABOVE_MIN below come from the gathered vectors: equal on all ranks)
        # (N_TREND is the same on every rank, so all ranks take this branch -- and its exchange -- together; an analysis that
        # fits a trend has non-zero rows somewhere and needs no exchange to know it)
        if st["N_TREND"] == 0:
            nnz = st["N_NONZERO"] if world == 1 else sum(parallel.allgather_sizes(st["N_NONZERO"], comm_device))
            if nnz == 0:
                raise ValueError("all genes have zero counts in every sample")
            raise RuntimeError("all gene-wise dispersion estimates are within 2 orders of magnitude from the minimum value")
        if st["TREND_STATUS"] != 0 or st["N_ABOVE_MIN"] == 0:
            # the reference falls back to a local / mean fit here (R/core.R:885-893): not on the fused path
            if world > 1:
                return parallel.DESeqParallel(dds, test=test, fitType=fitType, reduced=reduced, comm_device=comm_device,
                                              minReplicatesForReplace=minReplicatesForReplace, **kw)
            return core.DESeq(dds, test=test, fitType=fitType, reduced=reduced,
                              minReplicatesForReplace=minReplicatesForReplace, **kw)
        if custom is not None:
            fn = {"fitType": "custom", "coefficients": custom, "varLogDispEsts": float(sc[2]), "dispPriorVar": float(sc[3])}
        elif sc[L.DSQ_SC_FIT_USED] == L.DSQ_FIT["mean"]:
            fn = {"fitType": "mean", "coefficients": float(sc[0]), "varLogDispEsts": float(sc[2]), "dispPriorVar": float(sc[3])}
        else:
            fn = {"fitType": "parametric", "coefficients": np.array([sc[0], sc[1]]), "varLogDispEsts": float(sc[2]),
                  "dispPriorVar": float(sc[3])}
        allZero = hi[0].astype(bool)
        # rows that were all zero from the start carry NA in every column; a row that only BECAME all zero when its outlier
        # was replaced (newAllZero, R/core.R:2492) keeps its "intermediate" columns and gets NA in the "results" columns
        # only (:2534-2536) -- the device has already written those
        zero0 = allZero & (hi[5] == 0) if run.do_replace else allZero
        anyz = bool(zero0.any())

        def icol(v, as_bool=False):
            if anyz:
                out = v.astype(np.float64)
                out[zero0] = np.nan
                return out
            return v.astype(bool) if as_bool else v.copy()
        mc = {"baseMean": hv[0], "baseVar": hv[1], "allZero": allZero, "dispGeneEst": hv[2], "dispGeneIter": icol(hi[1]),
              "dispFit": hv[3], "dispMAP": hv[4], "dispersion": hv[5], "dispIter": icol(hi[2]),
              "dispOutlier": icol(hi[3], True), "beta": hm[0].T, "betaSE": hm[1].T, "betaIter": hv[6],
              "deviance": -2 * hv[7], "maxCooks": hv[9]}
        if run.force_zero is not None:
            wf = E._host(run.force_zero).numpy().astype(bool)
            if wf.any():
                mc["weightsFail"] = wf
        conv = hi[4].astype(np.float64)
        conv[hi[4] < 0] = np.nan
        if test == "Wald":
            pval = hm[3].T
            if kw.get("useT"):
                # t-distribution p-values (R/core.R:1474-1503): a function of the statistic and the residual degrees of
                # freedom alone, evaluated on the host from the downloaded column as core.nbinomWaldTest does
                from scipy.stats import t as tdist
                df = kw.get("df")
                if df is None:
                    num = E.to_numpy(run.w_norm).sum(axis=1) if run.useWeights else np.full(dds.n, dds.m)   # (core's own sum)
                    df = num - dds.p
                df = np.broadcast_to(np.asarray(df, float), (dds.n,))
                df = np.where(df > 0, df, np.nan)
                pval_t = 2 * tdist.sf(np.abs(hm[2].T), df=df[:, None])
                if run.do_replace and st["N_REFIT"] > 0:
                    # refitWithoutOutliers calls nbinomWaldTest WITHOUT useT (R/core.R:2524-2527): the refitted rows keep the
                    # normal-distribution p-values the device wrote
                    refit = (hi[5] != 0) & ~allZero
                    pval_t[refit] = pval[refit]
                pval = pval_t
            mc.update(WaldStatistic=hm[2].T, WaldPvalue=pval, betaConv=conv if np.isnan(conv).any() else hi[4].astype(bool))
        else:
            stat = 2 * (hv[7] - hv[8])                                                    # R/core.R:1877-1878
            if early_v is None:
                pval = core.pchisq_upper(stat, dds.p - run.p_red)
            else:
                # p-values computed during the outlier phase from the early copy: keep them where the statistic is what it
                # was then (bit for bit), recompute the rest (the refitted rows)
                changed = ~((stat == early_v[0]) | (np.isnan(stat) & np.isnan(early_v[0])))
                pval = early_v[1]
                if changed.any():
                    pval[changed] = core.pchisq_upper(stat[changed], dds.p - run.p_red)
            mc.update(LRTStatistic=stat, LRTPvalue=pval,
                      fullBetaConv=conv if np.isnan(conv).any() else hi[4].astype(bool))
        if run.do_replace:
            mc["replace"] = icol(hi[5], True)
        dds.mcols = mc
        GM = E.native.GeneMajor
        dds.assays = {"mu": GM(run.mu, dds.m), "H": GM(run.H, dds.m), "cooks": GM(run.cooks, dds.m)}
        if run.prior is not None:
            mc["MLE_beta"] = hmle.T
            dds.attrs.update(betaPriorVar=bpv, modelMatrixType=run.prior[1], factors=kw.get("factors"))
        dds.attrs.update(betaPrior=run.prior is not None, test=test, dispModelMatrix=np.asarray(dds.x, np.float64), fused=True,
                         status={**st, **{k: v for k, v in st2.items() if k.startswith(("N_REPLACE", "N_REFIT")) or k.endswith("_REFIT")}})
        if run.do_replace:
            dds.attrs["replaceable"] = run.replaceable.astype(bool)
            if st2["N_REPLACE"] > 0:
                dds.assays["replaceCounts"] = GM(run.replaceCounts, dds.m)
                dds.assays["replaceCooks"] = dds.assays["cooks"]
        if anyz:
            dds.attrs["nz_rows"] = np.where(~zero0)[0]
        dds.dispersionFunction = fn
        dds._fused_run = run           # keeps the device buffers of the assays alive
        return dds

    if wait or world > 1:
        return _finish(dds)
    run.mark_ready()
    dds._fused_pending = _finish
    return dds
