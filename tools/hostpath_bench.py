#!/usr/bin/env python
"""PCIe-inclusive rate: the same DESeq() chain through the HOST-pointer C ABI (what an unmodified R session
pays: every call uploads its n x m inputs from pageable host memory and downloads its outputs)."""
import os, sys, time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401  (loads the HIP runtime torch bundles before the engine library)
from deseq2_amd import core, simulate
from deseq2_amd.engine import HostEngine
n, m = int(os.environ.get("GENES", "50000")), 500
x = simulate.design_batch_condition(m)
d = simulate.make_counts(n, x, seed=1)
E = HostEngine()
def step():
    dds = core.DESeqDataSet(d["counts"], x, sizeFactors=d["size_factors"], engine=E)
    return core.DESeq(dds)
step()
t = time.perf_counter(); k = 3
for _ in range(k):
    step()
dt = (time.perf_counter() - t) / k
print("HOSTPATH %d genes x %d samples: %.1f ms per DESeq() = %.0f genes/s (host-pointer ABI, PCIe inclusive)" % (d["counts"].shape[0], m, dt * 1e3, d["counts"].shape[0] / dt))
