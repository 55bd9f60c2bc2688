#!/usr/bin/env python
"""times the general (per-sample) kernels: a C3-shaped analysis whose design carries a continuous covariate"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
from deseq2_amd import core, simulate
from deseq2_amd.engine import DeviceEngine
E = DeviceEngine("cuda:0")
SHAPES = [(20000, 500, 1, None), (20000, 500, 3, None), (20000, 200, 6, None),
          # a factor with p - 1 levels and ONE continuous covariate, 20 000 x 200
          (20000, 200, 1, 9), (20000, 200, 1, 15), (20000, 240, 1, 23),
          # threshold sweep for the wide-style build of the general kernels (DSQ_BETA_WIDE_MIN)
          (20000, 200, 1, 6), (20000, 200, 1, 7), (20000, 200, 1, 8), (20000, 500, 1, 8), (20000, 500, 1, 9), (20000, 100, 1, 9)]
if os.environ.get("CONTBENCH_ONLY"):
    SHAPES = [SHAPES[int(k)] for k in os.environ["CONTBENCH_ONLY"].split(",")]
for n, m, extra, levels in SHAPES:
    x0 = simulate.design_batch_condition(m) if levels is None else simulate.design_factor(m, levels)
    rng = np.random.default_rng(5)
    x = np.column_stack([x0] + [rng.normal(size=m) for _ in range(extra)])
    d = simulate.make_counts(n, x, seed=3, beta_sd=np.array([0.5] * (x.shape[1] - 2) + [1.0]) * 0.3)
    dds = core.DESeqDataSet(d["counts"], x, engine=E)
    for rep in range(2):
        E.record = []
        core.DESeq(dds, minReplicatesForReplace=np.inf)
        rec, E.record = E.record, None
    big = {}
    for name, g, ms in rec:
        if g > n // 2:
            big.setdefault(name, []).append(ms)
    print("p=%2d n=%d m=%d: " % (x.shape[1], dds.n, m) + "  ".join("%s %.2f ms" % (k, np.mean(v)) for k, v in big.items()), flush=True)
