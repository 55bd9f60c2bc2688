#!/usr/bin/env python
"""PCIe-inclusive rates of one DESeq() from PAGEABLE host memory (what an R session pays), C3 shape by default:
  classic : the chain through the per-call host-pointer entry points (dsq_fit_* + the SURVEY 8f extensions; every call
            uploads its n x m inputs and downloads its outputs) -- core.DESeq over HostEngine
  fused   : ONE dsq_deseq call (counts up once, device-driven chain, per-gene columns down) -- native.DESeq
Honours the staging knobs (DSQ_STAGE, DSQ_COPY_THREADS, DSQ_STAGE_MB) and DSQ_HOST_SHARDS; WHICH=classic|fused|both."""
import os, sys, time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401  (loads the HIP runtime torch bundles before the engine library)
from deseq2_amd import core, native, simulate
from deseq2_amd.engine import HostEngine
n, m = int(os.environ.get("GENES", "50000")), int(os.environ.get("SAMPLES", "500"))
which = os.environ.get("WHICH", "both")
x = simulate.design_batch_condition(m)
d = simulate.make_counts(n, x, seed=1)
counts_r = np.asfortranarray(d["counts"])
knobs = {k: v for k, v in os.environ.items() if k.startswith("DSQ_")}
E = HostEngine()


def classic():
    dds = core.DESeqDataSet(d["counts"], x, sizeFactors=d["size_factors"], engine=E)
    t = time.perf_counter()
    core.DESeq(dds)
    return time.perf_counter() - t


def fused(assays=()):
    t = time.perf_counter()
    native.DESeq(counts_r, x, d["size_factors"], assays=assays)
    return time.perf_counter() - t


if which in ("classic", "both"):
    classic()
    print("HOSTPATH classic %d x %d: %.1f ms %s" % (counts_r.shape[0], m, min(classic() for _ in range(3)) * 1e3, knobs))
if which in ("fused", "both"):
    fused()
    print("HOSTPATH fused   %d x %d: %.1f ms (with mu/H/cooks %.1f ms) %s" % (
        counts_r.shape[0], m, min(fused() for _ in range(4)) * 1e3, min(fused(("mu", "H", "cooks")) for _ in range(3)) * 1e3, knobs))
if os.environ.get("PROFILE"):
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable(); classic(); pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(30)
