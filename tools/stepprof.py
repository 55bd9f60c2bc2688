#!/usr/bin/env python
"""cProfile of bench.py's step() on the GPU box (host-side hot spots)."""
import cProfile, pstats, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deseq2_amd import core, fused, simulate
from deseq2_amd.engine import DeviceEngine
NG = int(os.environ.get("STEPPROF_GENES", "50000"))
m = 500; x = simulate.design_batch_condition(m)
d = simulate.make_counts(NG, x, seed=1); counts = d["counts"]; n = counts.shape[0]
dev = torch.device("cuda", 0); E = DeviceEngine(dev)
counts_r = torch.as_tensor(np.ascontiguousarray(counts.T), device=dev)
nf_r = torch.ones((m, n), dtype=torch.float64, device=dev)
def step():
    dds = core.DESeqDataSet.from_device(E, counts_r, nf_r, x, sizeFactors=np.ones(m)); (core.DESeq if os.environ.get('STEPPROF_CALLBYCALL') else fused.DESeq)(dds); return dds
for _ in range(2): step()
for i in range(4):
    torch.cuda.synchronize(); t = time.perf_counter(); step(); torch.cuda.synchronize()
    print("STEP %d: %.2f ms" % (i, (time.perf_counter() - t) * 1e3))
pr = cProfile.Profile(); pr.enable(); step(); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(int(os.environ.get("STEPPROF_TOP", "22")))
if os.environ.get("STEPPROF_CUM"): pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
