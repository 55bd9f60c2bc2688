"""Gene sharding across the GPUs of a node -- the mirror of the reference's only
parallelism strategy, DESeqParallel (R/parallel.R:6-74): contiguous gene ranges per worker
(`idx`, :10), gene-wise estimation per shard, the all-gene steps (dispersion trend and
prior variance, :27-28) computed from n-vectors only, then MAP + test per shard (:54-66).

One process per GPU.  The native fits need no collective at all (genes are independent,
src/DESeq2.cpp:194,319,492).  The only exchange is the pair of n-vectors the global trend
needs (baseMean, dispGeneEst): one small all-gather over torch.distributed (RCCL when the
backend is "nccl", gloo in the CPU tests); every rank then fits the identical trend.
"""
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


def DESeqParallel(dds, test="Wald", fitType="parametric", reduced=None, comm_device=None,
                  minReplicatesForReplace=7, **kw):
    """`dds` is THIS rank's shard.  R/parallel.R:6-74 (betaPrior = FALSE branch)."""
    # round 1: gene-wise estimates on the shard                                 (:18-20)
    core.estimateDispersionsGeneEst(dds)
    # global steps on the gathered n-vectors                                    (:27-28)
    bm_all = _allgather_vec(dds.mcols["baseMean"], comm_device)
    dge_all = _allgather_vec(dds.mcols["dispGeneEst"], comm_device)
    glob = _GlobalView(bm_all, dge_all, dds.x)
    core.estimateDispersionsFit(glob, fitType=fitType, engine=dds.engine)
    dispPriorVar = core.estimateDispersionsPriorVar(glob)
    # bring the global dispersion function back to the shard
    fn = glob.dispersionFunction
    bm = dds.mcols["baseMean"]
    dds.mcols["dispFit"] = (fn["coefficients"][0] + fn["coefficients"][1] / bm
                            if fn["fitType"] == "parametric" else np.full(bm.shape, fn["coefficients"]))
    dds.dispersionFunction = dict(fn)
    # round 2: MAP + test on the shard                                          (:54-66)
    core.estimateDispersionsMAP(dds, dispPriorVar=dispPriorVar)
    if test == "Wald":
        core.nbinomWaldTest(dds, **kw)
    else:
        core.nbinomLRT(dds, reduced, **kw)
    # outlier replacement + refit (R/core.R:419-426 after the parallel branch): per gene, so per shard;
    # the refit reads the global dispersion function / prior variance already on the shard
    if np.isfinite(minReplicatesForReplace) and core.nOrMoreInCell(dds.x, minReplicatesForReplace).any():
        core.refitWithoutOutliers(dds, test=test, reduced=reduced,
                                  minReplicatesForReplace=minReplicatesForReplace, **kw)
    return dds


class _GlobalView:
    """just enough of a DESeqDataSet for the all-gene steps: they read only n-vectors"""

    def __init__(self, baseMean, dispGeneEst, x):
        self.mcols = {"baseMean": baseMean, "dispGeneEst": dispGeneEst}
        self.x = x
        self.dispersionFunction = None
