"""Gene sharding across the GPUs of a node -- the mirror of the reference's only
parallelism strategy, DESeqParallel (R/parallel.R:6-74): contiguous gene ranges per worker
(`idx`, :10), gene-wise estimation per shard, the all-gene steps (dispersion trend and
prior variance, :27-28) computed from n-vectors only, then MAP + test per shard (:54-66).

One process per GPU.  The native fits need no collective at all (genes are independent,
src/DESeq2.cpp:194,319,492).  The only exchange is the pair of n-vectors the global trend
needs (baseMean, dispGeneEst): one small all-gather over torch.distributed (RCCL when the
backend is "nccl", gloo in the CPU tests); every rank then fits the identical trend.
"""
import os

import numpy as np

from . import core


def shard_ranges(n, nworkers):
    """idx <- factor(sort(rep(seq_len(nworkers), length.out = n)))   R/parallel.R:10"""
    idx = np.sort(np.resize(np.arange(nworkers), n))
    return [np.where(idx == w)[0] for w in range(nworkers)]


def _allgather_vec(v, device=None):
    """concatenate a per-rank 1-d float64 vector over all ranks, in rank order"""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return np.asarray(v, np.float64)
    ws = dist.get_world_size()
    dev = device if device is not None else torch.device("cpu")
    n_local = torch.tensor([len(v)], dtype=torch.int64, device=dev)
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(ws)]
    dist.all_gather(sizes, n_local)
    sizes = [int(s.item()) for s in sizes]
    nmax = max(sizes)
    buf = torch.full((nmax,), float("nan"), dtype=torch.float64, device=dev)
    buf[: len(v)] = torch.as_tensor(np.asarray(v, np.float64), device=dev)
    parts = [torch.empty(nmax, dtype=torch.float64, device=dev) for _ in range(ws)]
    dist.all_gather(parts, buf)
    return np.concatenate([p[:s].cpu().numpy() for p, s in zip(parts, sizes)])


def world_size():
    try:
        import torch.distributed as dist
        return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
    except ImportError:
        return 1


def rank():
    try:
        import torch.distributed as dist
        return dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0
    except ImportError:
        return 0


def allgather_sizes(n, device=None):
    """the shard sizes of all ranks, in rank order"""
    import torch
    import torch.distributed as dist
    dev = device if device is not None else torch.device("cpu")
    mine = torch.tensor([int(n)], dtype=torch.int64, device=dev)
    out = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return [int(v.item()) for v in out]


def allgather_device_pairs(a, b, nmax, comm_device, torch):
    """(baseMean, dispGeneEst) of all ranks as two device vectors of world * nmax values in rank order, each shard
    padded with NaN to nmax: the trend fit skips NaN rows, so the padded sequence fits like the concatenation"""
    import torch.distributed as dist
    ws = dist.get_world_size()
    dev = a.device
    buf = torch.full((2, nmax), float("nan"), dtype=torch.float64, device=dev)
    buf[0, : a.numel()] = a
    buf[1, : b.numel()] = b
    if comm_device is None:                       # ranks sharing one device (tests): exchange through the host
        hb = buf.cpu()
        parts = [torch.empty_like(hb) for _ in range(ws)]
        dist.all_gather(parts, hb)
        g = torch.stack(parts).to(dev)            # (ws, 2, nmax)
    else:
        g = torch.empty((ws, 2, nmax), dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(g.view(-1), buf.view(-1))
    return g[:, 0, :].contiguous().view(-1), g[:, 1, :].contiguous().view(-1)


class Baton:
    """Round-robin token for the cooperative chunk pipeline: exactly one chunk thread runs host code at a time
    (no interpreter-lock contention); a chunk passes the token on wherever it would block on the GPU or on the
    other chunks, and continues when the token comes round again."""

    def __init__(self, k):
        import threading
        self.k, self.turn = k, 0
        self.cv = threading.Condition()
        self.alive = [True] * k

    def _advance(self, i):
        for step in range(1, self.k + 1):
            j = (i + step) % self.k
            if self.alive[j]:
                self.turn = j
                return
        self.turn = -1

    def acquire(self, i):
        with self.cv:
            while self.turn != i:
                self.cv.wait()

    def release(self, i):
        with self.cv:
            self._advance(i)
            self.cv.notify_all()

    def handoff(self, i):
        self.release(i)
        self.acquire(i)

    def retire(self, i):
        with self.cv:
            self.alive[i] = False
            if self.turn == i:
                self._advance(i)
            self.cv.notify_all()


class LocalGroup:
    """The chunk workers of ONE process: k threads, each driving its own HIP stream over a contiguous
    gene range of this rank's shard (DESeqPipelined).  allgather() concatenates the chunk vectors in chunk
    order and, when torch.distributed is initialised, lets chunk 0 extend that over the ranks -- the
    global gene order is (rank, chunk), i.e. the contiguous ranges of R/parallel.R:10."""

    def __init__(self, k, comm_device=None, baton=None):
        import threading
        self.k, self.comm_device, self.baton = k, comm_device, baton
        self.barrier = threading.Barrier(k)
        self.slots = [None] * k
        self.result = None

    def _wait(self, idx):
        if self.baton is not None:          # never hold the token while waiting for the other chunks
            self.baton.release(idx)
        self.barrier.wait()
        if self.baton is not None:
            self.baton.acquire(idx)

    def allgather(self, idx, vec):
        self.slots[idx] = np.asarray(vec, np.float64)
        self._wait(idx)
        if idx == 0:
            self.result = _allgather_vec(np.concatenate(self.slots), self.comm_device)
        self._wait(idx)
        res = self.result
        self._wait(idx)              # everybody has read before the slots are reused
        return res


def DESeqParallel(dds, test="Wald", fitType="parametric", reduced=None, comm_device=None,
                  minReplicatesForReplace=7, group=None, chunk=0, **kw):
    """`dds` is THIS worker's shard (a rank's, or one chunk of a rank's when `group` is given).
    R/parallel.R:6-74, both branches (betaPrior = TRUE: :30-48)."""
    betaPrior = bool(kw.get("betaPrior"))
    if betaPrior and test != "Wald":
        raise ValueError("betaPrior: the Wald test only")
    # round 1: gene-wise estimates on the shard                                 (:18-20; minmu is handed on, :19)
    core.estimateDispersionsGeneEst(dds, minmu=kw.get("minmu", 0.5))
    # global steps on the gathered n-vectors                                    (:27-28)
    if group is not None:
        bm_all = group.allgather(chunk, dds.mcols["baseMean"])
        dge_all = group.allgather(chunk, dds.mcols["dispGeneEst"])
    else:
        bm_all = _allgather_vec(dds.mcols["baseMean"], comm_device)
        dge_all = _allgather_vec(dds.mcols["dispGeneEst"], comm_device)
    glob = _GlobalView(bm_all, dge_all, dds.x)
    core.estimateDispersionsFit(glob, fitType=fitType, engine=dds.engine)
    dispPriorVar = core.estimateDispersionsPriorVar(glob)
    # bring the global dispersion function back to the shard
    fn = glob.dispersionFunction
    bm = dds.mcols["baseMean"]
    dds.dispersionFunction = dict(fn)
    dds.mcols["dispFit"] = core._dispersion_function(dds, bm)      # parametric / mean / a caller's function ('custom')
    # round 2: MAP + test on the shard                                          (:54-66)
    core.estimateDispersionsMAP(dds, dispPriorVar=dispPriorVar)
    if betaPrior and kw.get("betaPriorVar") is None:
        # (:34-40) the MLE coefficients of every shard -- estimateMLEForBetaPriorVar (R/core.R:1693-1730): fitNbinomGLMs on
        # its DEFAULTS and, as written there, without the observation weights -- then the beta prior variance over ALL
        # rows (estimateBetaPriorVar on the recombined object)
        gather = (lambda v: group.allgather(chunk, v)) if group is not None else (lambda v: _allgather_vec(v, comm_device))
        factors = kw.get("factors")
        mmt = kw.get("modelMatrixType") or ("expanded" if factors is not None else "standard")
        names = core.standard_model_matrix(factors)[1] if factors is not None else ["Intercept"] + ["V%d" % i for i in range(1, dds.p)]
        mle = np.asarray(core.fitNbinomGLMs(dds)["betaMatrix"], np.float64)
        mle_all = np.column_stack([gather(np.ascontiguousarray(mle[:, c])) for c in range(mle.shape[1])])
        view = type("V", (), {"mcols": {"baseMean": bm_all, "dispFit": gather(dds.mcols["dispFit"])}})()
        bpv, _ = core.estimateBetaPriorVar(view, mle_all, names, modelMatrixType=mmt, factors=factors)
        kw = dict(kw, betaPriorVar=bpv)
    if test == "Wald":
        core.nbinomWaldTest(dds, **kw)                                          # (:42-46 / :54-60)
    else:
        core.nbinomLRT(dds, reduced, **kw)
    # outlier replacement + refit (R/core.R:419-426 after the parallel branch): per gene, so per shard;
    # the refit reads the global dispersion function / prior variance already on the shard
    if np.isfinite(minReplicatesForReplace) and core.nOrMoreInCell(dds.x, minReplicatesForReplace).any():
        # ... except its closing steps, which ask whether ANY row of the whole object was refitted (R/core.R:2496)
        if group is not None:
            count_all = lambda k: float(group.allgather(chunk, np.array([float(k)])).sum())      # noqa: E731
        elif world_size() > 1:
            count_all = lambda k: sum(allgather_sizes(k, comm_device))                          # noqa: E731
        else:
            count_all = None
        core.refitWithoutOutliers(dds, test=test, reduced=reduced, count_all=count_all,
                                  minReplicatesForReplace=minReplicatesForReplace, **kw)
    return dds


class _GlobalView:
    """just enough of a DESeqDataSet for the all-gene steps: they read only n-vectors"""

    def __init__(self, baseMean, dispGeneEst, x):
        self.mcols = {"baseMean": baseMean, "dispGeneEst": dispGeneEst}
        self.x = x
        self.dispersionFunction = None


class Pipeline:
    """DESeq() of one rank's genes as k chunks, each in its own thread on its own HIP stream: while one
    chunk's host code (the R-side decision rules between the native calls) runs, the other chunks' kernels
    keep the GPU busy.  Same arithmetic per gene and the same all-gene steps as DESeqParallel, so the
    results equal the serial ones (tests/test_gpu_pipeline.py).  Streams and their engine workspaces
    persist across calls."""

    def __init__(self, engine, n_chunks=2, comm_device=None, cooperative=True):
        self.engine, self.k, self.comm_device = engine, int(n_chunks), comm_device
        self.cooperative = bool(cooperative)      # chunks take turns on the host (Baton) instead of free-running
        t = engine.torch
        self.streams = [t.cuda.Stream(device=engine.device) for _ in range(self.k)]

    def run(self, make_dds, n, **kw):
        """make_dds(lo, hi) builds the DESeqDataSet of genes [lo, hi) (called inside the chunk's stream)."""
        overlap = True
        import threading
        t = self.engine.torch
        bounds = [r[[0, -1]] + np.array([0, 1]) for r in shard_ranges(n, self.k)]
        out, errs = [None] * self.k, []
        baton = Baton(self.k) if (self.cooperative and self.k > 1) else None
        group = LocalGroup(self.k if overlap else 1, self.comm_device, baton)
        cur = t.cuda.current_stream(self.engine.device)

        def work(c):
            try:
                t.cuda.set_device(self.engine.device)
                if baton is not None:
                    baton.acquire(c)
                    self.engine._tls.before_sync = lambda: baton.handoff(c)
                self.streams[c].wait_stream(cur)
                with t.cuda.stream(self.streams[c]):
                    dds = make_dds(int(bounds[c][0]), int(bounds[c][1]))
                    out[c] = DESeqParallel(dds, comm_device=self.comm_device, group=group, chunk=c, **kw)
                    self.engine._tls.before_sync = None
                    if baton is not None:
                        baton.retire(c)
                    self.streams[c].synchronize()
            except BaseException as e:          # noqa: BLE001 -- re-raised in the caller's thread
                errs.append(e)
                if baton is not None:
                    baton.retire(c)
                group.barrier.abort()
        if overlap and self.k > 1:
            import sys
            # a chunk thread coming back from a kernel wait must not sit out the interpreter's default 5 ms
            # switch interval while another chunk runs host code
            old = sys.getswitchinterval()
            sys.setswitchinterval(float(os.environ.get("DSQ_SWITCH_INTERVAL", "1e-4")))
            try:
                th = [threading.Thread(target=work, args=(c,)) for c in range(self.k)]
                for x in th:
                    x.start()
                for x in th:
                    x.join()
            finally:
                sys.setswitchinterval(old)
        else:
            work(0)
        if errs:
            raise errs[0]
        return out


def concat_mcols(shards, keys=None):
    """per-gene columns of the chunk results, back in gene order"""
    keys = keys if keys is not None else [k for k in shards[0].mcols if k != "rowsForOptim"]
    return {k: np.concatenate([np.asarray(d.mcols[k]) for d in shards]) for k in keys}
