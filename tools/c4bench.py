#!/usr/bin/env python
"""BASELINE configs[3] shape (scaled in genes): 10-level factor, m = 2000, nbinomLRT vs intercept"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deseq2_amd import core, simulate
from deseq2_amd.engine import DeviceEngine
E = DeviceEngine("cuda:0")
n, m, levels = int(os.environ.get("GENES", "12000")), int(os.environ.get("SAMPLES", "2000")), int(os.environ.get("LEVELS", "10"))
x = simulate.design_factor(m, levels)
d = simulate.make_counts(n, x, seed=3)
def run(rec):
    dds = core.DESeqDataSet(d["counts"], x, engine=E)
    E.record = [] if rec else None
    torch.cuda.synchronize(); t = time.perf_counter()
    core.DESeq(dds, test="LRT", reduced=np.ones((m, 1)))
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    r, E.record = E.record, None
    return dt, r, dds
run(False)
dt, _, dds = run(False)
_, rec, _ = run(True)
big = {}
for name, g, ms in rec:
    if g > n // 2:
        big.setdefault(name, []).append(ms)
print("C4 n=%d m=%d p=%d: %.1f ms per DESeq(LRT) = %.0f genes/s;  " % (dds.n, m, levels, dt * 1e3, dds.n / dt) +
      "  ".join("%s %s" % (k, ["%.1f" % v for v in vs]) for k, vs in big.items()))
