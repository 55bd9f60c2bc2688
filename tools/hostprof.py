#!/usr/bin/env python
"""cProfile of the HOST side of one fused step (bench.py's step at --genes N, one call at a time): where the Python time
between two chains goes.  usage: python tools/hostprof.py [genes] [config]"""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np   # noqa: E402
import torch         # noqa: E402

import bench         # noqa: E402
from deseq2_amd import core, fused, simulate   # noqa: E402
from deseq2_amd.engine import DeviceEngine     # noqa: E402

genes = int(sys.argv[1]) if len(sys.argv) > 1 else 6250
cfg = bench.CONFIGS[sys.argv[2] if len(sys.argv) > 2 else "C3"]
m = cfg["samples"]
x = bench.make_design(cfg["design"], m)
dev = torch.device("cuda", 0)
E = DeviceEngine(dev)
sf = np.exp(np.random.Generator(np.random.PCG64(1001)).normal(0.0, 0.25, m))
d = simulate.make_counts(genes, x, seed=1, intercept_mean=cfg.get("intercept_mean", 4.0), size_factors=sf)
counts_r = torch.as_tensor(np.ascontiguousarray(d["counts"].T), device=dev)
nf_r = torch.ones((m, d["counts"].shape[0]), dtype=torch.float64, device=dev) * torch.as_tensor(sf, device=dev)[:, None]


def step():
    dds = core.DESeqDataSet.from_device(E, counts_r, nf_r, x, sizeFactors=d["size_factors"])
    fused.DESeq(dds, test=cfg["test"])
    return dds


for _ in range(5):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    step()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
