#!/usr/bin/env python
"""cProfile of bench.py's step() on the GPU box (host-side hot spots).  STEPPROF_CFG = C2 | C3 | C4 | C4R | C5 (bench.py's
configs), STEPPROF_GENES overrides the gene count, STEPPROF_CALLBYCALL=1 profiles core.DESeq() instead of the fused chain."""
import cProfile, pstats, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from deseq2_amd import core, fused, simulate
from deseq2_amd.engine import DeviceEngine
cfg = dict(bench.CONFIGS[os.environ.get("STEPPROF_CFG", "C3")])
NG = int(os.environ.get("STEPPROF_GENES", cfg["genes"]))
m = cfg["samples"]; x = bench.make_design(cfg["design"], m)
d = simulate.make_counts(NG, x, seed=1, intercept_mean=cfg.get("intercept_mean", 4.0)); counts = d["counts"]; n = counts.shape[0]
dev = torch.device("cuda", 0); E = DeviceEngine(dev)
counts_r = torch.as_tensor(np.ascontiguousarray(counts.T), device=dev)
nf_r = torch.ones((m, n), dtype=torch.float64, device=dev)
w = bench.make_weights(n, m, 78) if cfg.get("weights") else None
w_r = None if w is None else torch.as_tensor(np.ascontiguousarray(w.T), device=dev)
kw = dict(test=cfg["test"], reduced=np.ones((m, 1)) if cfg["test"] == "LRT" else None)
if cfg.get("reduced2"):
    kw["reduced"] = np.column_stack([np.ones(m), (np.arange(m) >= m // 2).astype(np.float64)])
if cfg.get("betaPrior"):
    kw.update(betaPrior=True, factors={"condition": x[:, 1].astype(int)})
if cfg.get("minmu"):
    kw["minmu"] = cfg["minmu"]
def step():
    dds = core.DESeqDataSet.from_device(E, counts_r, nf_r, x, weights=w, sizeFactors=np.ones(m), weights_r=w_r)
    (core.DESeq if os.environ.get('STEPPROF_CALLBYCALL') else fused.DESeq)(dds, **kw); return dds
for _ in range(2): step()
for i in range(4):
    torch.cuda.synchronize(); t = time.perf_counter(); dds = step(); torch.cuda.synchronize()
    print("STEP %d: %.2f ms (fused %s)" % (i, (time.perf_counter() - t) * 1e3, dds.attrs.get("fused")))
pr = cProfile.Profile(); pr.enable(); step(); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(int(os.environ.get("STEPPROF_TOP", "22")))
if os.environ.get("STEPPROF_CUM"): pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
