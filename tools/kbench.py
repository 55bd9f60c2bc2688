#!/usr/bin/env python
"""Kernel micro-benchmark for tuning sweeps (GPU box).  Times one fitBeta (QR, final-fit
style) and one fitDisp (no prior) launch on gene-major resident data with HIP events.
usage: python tools/kbench.py [--genes N] [--samples M] [--reps R]   (env DSQ_* knobs apply)"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genes", type=int, default=50000)
    ap.add_argument("--samples", type=int, default=500)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--cache", default="/tmp/kbench_data.npz")
    ap.add_argument("--design", default="bc")
    ap.add_argument("--nocr", action="store_true")
    args = ap.parse_args()
    import torch
    from deseq2_amd import simulate
    from deseq2_amd.engine import DeviceEngine
    from tests.helpers import rough_alpha, beta_init_qr
    m = args.samples
    if args.design == "bc":
        x = simulate.design_batch_condition(m)
    elif args.design.startswith("f"):
        x = simulate.design_factor(m, int(args.design[1:]))
    else:
        x = simulate.design_two_group(m)
    key = "%d_%d_%s" % (args.genes, m, args.design)
    if os.path.exists(args.cache) and str(np.load(args.cache)["key"]) == key:
        z = np.load(args.cache)
        counts, alpha, b0 = z["counts"], z["alpha"], z["b0"]
    else:
        d = simulate.make_counts(args.genes, x, seed=1)
        counts = d["counts"]
        nf = np.ones(counts.shape)
        alpha = rough_alpha(counts.astype(float), nf, x)
        b0 = beta_init_qr(counts.astype(float), nf, x)
        np.savez(args.cache, key=key, counts=counts, alpha=alpha, b0=b0)
    n = counts.shape[0]
    p = x.shape[1]
    E = DeviceEngine("cuda:0")
    y = E.counts(counts)
    nf = E.matrix(np.ones(counts.shape))
    xh = E.design(x)
    lam = np.full(p, 1e-6) / np.log(2) ** 2
    contrast = np.r_[1.0, np.zeros(p - 1)]
    E.record = []
    for _ in range(args.reps + 1):
        fb = E.fit_beta(y, xh, nf, alpha, contrast, b0, lam, None, False, 1e-8, 100, True, 0.5, want_mu=True,
                        mu_floor=0.5, want_hat=True)
        fd = E.fit_disp(y, xh, fb["mu"], np.log(alpha), np.log(alpha), 1.0, np.log(1e-9), 1.0, 1e-6, 100, False, None,
                        False, 1e-2, not args.nocr)
    torch.cuda.synchronize()
    tb = [ms for nm, _, ms in E.record if nm == "fit_beta"][1:]
    td = [ms for nm, _, ms in E.record if nm == "fit_disp"][1:]
    knobs = {k: v for k, v in os.environ.items() if k.startswith("DSQ_")}
    print("KBENCH n=%d m=%d p=%d  fit_beta %.3f ms  fit_disp %.3f ms  (beta iters %.2f, disp iters %.2f) %s" %
          (n, m, p, np.mean(tb), np.mean(td), fb["iter"].mean(), fd["iter"].mean(), knobs))


if __name__ == "__main__":
    main()
