"""GPU fuzz of the whole chain (run by hand on a GPU box): the fused device-driven DESeq() against the call-by-call
chain of core.py on the same engine, random small analyses -- designs (factor, factor + factor, two-group; with a
continuous covariate the general kernels), samples per cell, size factors, weights, spiked outliers, all-zero rows,
Wald / LRT -- every per-gene column, the assays and the trend bit for bit (tests/test_gpu_fused.py's comparison).

    python tests/gpu_fuzz_chain.py [first_seed] [n_seeds]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402
from deseq2_amd import core, fused, simulate  # noqa: E402
from deseq2_amd.engine import DeviceEngine  # noqa: E402
from tests.test_gpu_fused import _compare, _spike_outliers  # noqa: E402


def one(E, seed):
    rng = np.random.default_rng(70000 + seed)
    kind = int(rng.integers(4))
    if kind == 0:
        m = int(rng.integers(3, 9)) * 2
        x = simulate.design_two_group(m)
    elif kind == 1:
        m = int(rng.integers(2, 7)) * 6
        x = simulate.design_batch_condition(m)
    elif kind == 2:
        levels = int(rng.integers(3, 9))
        m = levels * int(rng.integers(2, 9))
        x = simulate.design_factor(m, levels)
    else:
        m = int(rng.integers(3, 7)) * 6
        x = np.column_stack([simulate.design_batch_condition(m), rng.normal(size=m)])
    n = int(rng.integers(120, 600))
    sf = np.exp(rng.normal(0, 0.25, m)) if rng.uniform() < 0.6 else np.ones(m)
    d = simulate.make_counts(n, x, seed=int(rng.integers(1 << 30)), size_factors=sf, drop_all_zero=bool(rng.uniform() < 0.5),
                             intercept_mean=float(rng.uniform(1.0, 6.0)))
    counts = d["counts"].copy()
    if rng.uniform() < 0.5:
        counts = _spike_outliers(counts, rng, k=int(rng.integers(1, 6)))
    if rng.uniform() < 0.3:
        counts[:: int(rng.integers(17, 60))] = 0
    weights = None
    if rng.uniform() < 0.3:
        weights = rng.uniform(0.05, 1.0, counts.shape)
        weights[rng.uniform(size=counts.shape) < 0.02] = 0.0
    kw = {}
    if rng.uniform() < 0.3:
        kw.update(test="LRT", reduced=np.ones((m, 1)))
    tag = "seed %d: kind=%d n=%d m=%d p=%d weights=%d %s" % (seed, kind, counts.shape[0], m, x.shape[1], weights is not None, kw.get("test", "Wald"))
    a = core.DESeqDataSet(counts, x, sizeFactors=d["size_factors"], weights=weights, engine=E)
    try:
        core.DESeq(a, **kw)
    except NotImplementedError as e:       # e.g. residual df <= 3: the prior-variance branch that needs R's RNG
        return tag + " SKIP " + str(e)[:60]
    except ValueError as e:                # the wrappers' NA guard (R/wrappers.R:31-34), e.g. a gene whose only non-zero
        if "contain NA" in str(e):         # counts carry weight 0: R stops there as well
            return tag + " SKIP NA guard"
        raise
    b = core.DESeqDataSet(counts, x, sizeFactors=d["size_factors"], weights=weights, engine=E)
    fused.DESeq(b, **kw)
    if not b.attrs.get("fused"):           # (a failed parametric trend hands the analysis to core.DESeq)
        return tag + " SKIP not fused"
    _compare(a, b, tag)
    return tag


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    E = DeviceEngine("cuda:0")
    bad = skipped = 0
    for s in range(first, first + count):
        try:
            r = one(E, s)
            skipped += " SKIP " in r
        except AssertionError as e:
            bad += 1
            import traceback
            tb = traceback.extract_tb(e.__traceback__)[-1]
            print("FAIL seed %d at %s:%d %s | %s" % (s, os.path.basename(tb.filename), tb.lineno, tb.line, str(e)[:400]), flush=True)
        except Exception as e:                       # noqa: BLE001
            bad += 1
            print("ERROR seed %d: %r" % (s, e), flush=True)
    print("chain fuzz: %d seeds, %d failures, %d skipped" % (count, bad, skipped))


if __name__ == "__main__":
    main()
