#!/usr/bin/env python
"""times the wide-design (generic, scratch-resident) kernels next to a register-resident width"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401  (loads the HIP runtime torch bundles before the engine library)
from deseq2_amd import core, simulate
from deseq2_amd.engine import DeviceEngine
E = DeviceEngine("cuda:0")
for levels, n, m in ((10, 20000, 200), (12, 20000, 200), (16, 20000, 200), (20, 20000, 200), (24, 20000, 240)):
    x = simulate.design_factor(m, levels)
    d = simulate.make_counts(n, x, seed=3)
    dds = core.DESeqDataSet(d["counts"], x, engine=E)
    E.record = []
    core.DESeq(dds, minReplicatesForReplace=np.inf)
    rec, E.record = E.record, None
    big = {}
    for name, g, ms in rec:
        if g > n // 2:
            big.setdefault(name, []).append(ms)
    print("p=%2d n=%d m=%d: " % (levels, dds.n, m) + "  ".join("%s %.1f ms" % (k, np.mean(v)) for k, v in big.items()))
