"""one-off (round 5): what would longest-first scheduling of the dispersion searches buy?  C3 workload, the genes of the
input permuted by DESCENDING iteration count of a first run (dispGeneIter / dispIter), per-kernel times of both orders.
The results are per gene, so only the launch times move."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import bench
from deseq2_amd import core, fused, simulate
from deseq2_amd.engine import DeviceEngine
cfg = dict(bench.CONFIGS["C3"]); m = cfg["samples"]; x = bench.make_design(cfg["design"], m)
genes = int(os.environ.get("GENES", cfg["genes"]))
sf = np.exp(np.random.Generator(np.random.PCG64(1001)).normal(0.0, 0.25, m))
d = simulate.make_counts(genes, x, seed=1, size_factors=sf)
dev = torch.device("cuda", 0); E = DeviceEngine(dev)

def run(counts, tag):
    n = counts.shape[0]
    counts_r = torch.as_tensor(np.ascontiguousarray(counts.T), device=dev)
    nf_r = torch.ones((m, n), dtype=torch.float64, device=dev) * torch.as_tensor(sf, device=dev)[:, None]
    def step():
        dds = core.DESeqDataSet.from_device(E, counts_r, nf_r, x, sizeFactors=sf)
        fused.DESeq(dds); return dds
    for _ in range(3): step()
    E.record = []
    for _ in range(3): dds = step()
    rec, E.record = E.record, None
    per = {}
    for name, ng, ms in rec:
        if ng >= n // 2: per.setdefault(name, []).append(ms)
    print(tag, {k: [round(v, 3) for v in vs[-2:]] for k, vs in per.items() if k in ("fit_disp", "fit_beta")})
    return dds

dds = run(d["counts"], "natural order      ")
gi, mi = np.asarray(dds.mcols["dispGeneIter"], float), np.asarray(dds.mcols["dispIter"], float)
print("dispGeneIter: mean %.1f, >=30: %.2f%%, >=60: %.2f%%, ==100: %.2f%%" % (np.nanmean(gi), 100 * np.mean(gi >= 30), 100 * np.mean(gi >= 60), 100 * np.mean(gi >= 100)))
print("dispIter    : mean %.1f, >=30: %.2f%%, >=60: %.2f%%, ==100: %.2f%%" % (np.nanmean(mi), 100 * np.mean(mi >= 30), 100 * np.mean(mi >= 60), 100 * np.mean(mi >= 100)))
print("corr(dispGeneIter, dispIter) = %.3f" % np.corrcoef(np.nan_to_num(gi), np.nan_to_num(mi))[0, 1])
bm = np.asarray(dds.mcols["baseMean"]); dge = np.asarray(dds.mcols["dispGeneEst"])
for thr in (30, 60):
    long_ = gi >= thr
    print("genes with dispGeneIter >= %d: median baseMean %.2f, median dispGeneEst %.3g, frac at the floor (<= 1e-7): %.2f" %
          (thr, np.median(bm[long_]), np.median(dge[long_]), np.mean(dge[long_] <= 1e-7)))
run(d["counts"][np.argsort(-np.nan_to_num(gi), kind="stable")], "by dispGeneIter desc")
run(d["counts"][np.argsort(-np.nan_to_num(mi), kind="stable")], "by dispIter desc    ")
run(d["counts"][np.argsort(np.nan_to_num(gi), kind="stable")], "by dispGeneIter ASC ")
