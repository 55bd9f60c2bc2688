#!/usr/bin/env python
"""times the dispersion-trend kernel (dsq_parametric_dispersion_fit_dev) for 1 and 8 ranks' worth of genes"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deseq2_amd import native
from oracle import oracle as O
rng = np.random.default_rng(0)
for n in (1000, 50000, 400000):
    means = np.exp(rng.normal(4, 2, n)); disps = (0.1 + 4 / means) * np.exp(rng.normal(0, 0.5, n))
    dm, dd = torch.as_tensor(means, device="cuda"), torch.as_tensor(disps, device="cuda")
    native.parametricDispersionFit_dev(dm, dd)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        c = native.parametricDispersionFit_dev(dm, dd)
    e1.record(); torch.cuda.synchronize()
    ref = O.parametricDispersionFit(means, disps)
    print("n=%d  %.3f ms per fit  coefs %s  equal to oracle: %s" % (n, e0.elapsed_time(e1) / 5, c, (c == ref).all()))
