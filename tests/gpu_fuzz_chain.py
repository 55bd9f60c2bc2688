"""GPU fuzz of the whole chain (run by hand on a GPU box): the fused device-driven DESeq() against the call-by-call
chain of core.py on the same engine, random small analyses -- designs (factor, factor + factor, two-group; with a
continuous covariate the general kernels), samples per cell, size factors or a normalization-factor matrix, weights,
spiked outliers, all-zero rows, Wald (useT, betaPrior) / LRT against ~1 or a nested reduced model -- every per-gene
column, the assays and the trend bit for bit (tests/test_gpu_fused.py's comparison); where the one-call host entry
covers the analysis (dsq_deseq: no useT) it is run too, over a random number of in-library gene ranges, and compared
with the fused chain column by column.  Round 4: wide factor designs (11 ... 20 levels, the zero-padded kernel builds
inside the chain), the beta prior THROUGH the host entry (prior variance estimated inside the library), a
normalization-factor matrix together with the outlier refit, minmu != 0.5 on Wald analyses, fitType = "mean" (and the mean
substituted on the device where the parametric trend does not fit).  Round 5: wide designs to 36 levels and paired designs
(~ patient + treatment) WITH weights and reduced models of any width below p on the chain, the caller's trend function with
count outliers (the outlier phase in its two halves), and analyses enqueued with wait = False and finished later.

    python tests/gpu_fuzz_chain.py [first_seed] [n_seeds]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402
from deseq2_amd import core, fused, native, simulate  # noqa: E402
from deseq2_amd.engine import DeviceEngine  # noqa: E402
from tests.helpers import assert_same  # noqa: E402
from tests.test_gpu_fused import _compare, _smooth_trend, _spike_outliers  # noqa: E402


def _host_entry_check(b, counts, x, sf, nfm, weights, kw, rng, tag):
    """dsq_deseq on the same analysis: every column it returns against the fused chain's"""
    shards = int(rng.integers(1, 5))
    old = os.environ.get("DSQ_HOST_SHARDS")
    try:
        os.environ["DSQ_HOST_SHARDS"] = str(shards)
        res = native.DESeq(counts, x, sf, test=kw.get("test", "Wald"), reduced=kw.get("reduced"),
                           normalizationFactors=nfm, weights=weights, minmu=kw.get("minmu", 0.5),
                           minReplicatesForReplace=kw.get("minReplicatesForReplace", 7), assays=("mu", "cooks"),
                           betaPrior=kw.get("betaPrior", False), factors=kw.get("factors"),
                           fitType="mean" if kw.get("fitType") == "mean" else "parametric_or_mean")
    finally:
        if old is None:
            os.environ.pop("DSQ_HOST_SHARDS", None)
        else:
            os.environ["DSQ_HOST_SHARDS"] = old
    f = lambda v: np.asarray(v, dtype=np.float64)      # noqa: E731
    pairs = {"baseMean": "baseMean", "dispGeneEst": "dispGeneEst", "dispFit": "dispFit", "dispMAP": "dispMAP",
             "dispersion": "dispersion", "dispIter": "dispIter", "beta": "beta", "betaSE": "betaSE", "betaIter": "betaIter",
             "maxCooks": "maxCooks", "allZero": "allZero"}
    for k, kb in pairs.items():
        assert_same(f(res[k]), f(b.mcols[kb]), "%s host entry (%d ranges): %s" % (tag, shards, k))
    assert_same(-2 * f(res["logLike"]), f(b.mcols["deviance"]), tag + " host entry: deviance")
    if kw.get("test") == "LRT":
        assert_same(2 * (f(res["logLike"]) - f(res["logLikeReduced"])), f(b.mcols["LRTStatistic"]), tag + " host entry: LRT")
    else:
        assert_same(f(res["stat"]), f(b.mcols["WaldStatistic"]), tag + " host entry: Wald statistic")
        assert_same(f(res["pvalue"]), f(b.mcols["WaldPvalue"]), tag + " host entry: Wald p-value")
    if kw.get("betaPrior"):
        assert_same(res["betaPriorVar"], np.asarray(b.attrs["betaPriorVar"]), tag + " host entry: betaPriorVar")
    nz = ~np.asarray(b.mcols["allZero"], bool) | (np.nan_to_num(f(b.mcols.get("replace", np.zeros(b.n)))) == 1)
    E = b.engine
    for k in ("mu", "cooks"):
        if k in b.assays:
            assert_same(res[k][nz], E.to_numpy(b.assays[k])[nz], tag + " host entry: assays$" + k)


def one(E, seed):
    rng = np.random.default_rng(70000 + seed)
    kind = int(rng.integers(6))
    if kind == 5:
        patients = int(rng.integers(5, 25))                      # ~ patient + treatment: p = 6 ... 25, 2 * patients cells
        reps = int(rng.integers(1, 4))
        m = 2 * reps * patients
        pat = np.repeat(np.arange(patients), 2 * reps)
        x = np.column_stack([np.ones(m)] + [(pat == k).astype(float) for k in range(1, patients)] +
                            [np.tile(np.repeat([0.0, 1.0], reps), patients)])
    elif kind == 4:
        levels = int(rng.integers(11, 37))                       # wide: p = 11 ... 36
        m = levels * int(rng.integers(2, 9))
        x = simulate.design_factor(m, levels)
    elif kind == 0:
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
    if rng.uniform() < 0.3:                                      # (round 5: on wide designs too)
        weights = rng.uniform(0.05, 1.0, counts.shape)
        weights[rng.uniform(size=counts.shape) < 0.02] = 0.0
    kw = {}
    p = x.shape[1]
    u = rng.uniform()
    if u < 0.3:
        q = 1 if (p == 1 or rng.uniform() < 0.4) else (p - 1 if rng.uniform() < 0.4 else int(rng.integers(1, p)))   # (round 5: any width < p)
        kw.update(test="LRT", reduced=np.ones((m, 1)) if q == 1 else np.ascontiguousarray(x[:, :q]))
        if q > 1 and rng.uniform() < 0.5:
            kw["minmu"] = 1e-6                                   # R/core.R:1856-1868 (glmGamPoi-style floor)
    elif u < 0.45:
        kw.update(useT=True)
        if rng.uniform() < 0.3:
            kw["minmu"] = float(rng.choice([1e-6, 0.1, 2.0]))
    elif u < 0.6 and kind in (0, 2, 4):                           # (round 5: wide factors too -- the prior pass at its padded width)
        # betaPrior on the expanded model matrix of a one-factor design (R/core.R:1374-1380)
        lev = (x[:, 1:] @ np.arange(1, p)).astype(int) if p > 1 else np.zeros(m, int)
        kw.update(betaPrior=True, factors={"condition": lev})
    u2 = rng.uniform()
    if u2 < 0.15:
        kw["fitType"] = "mean"                                   # R/core.R:894-899 on the device
    elif u2 < 0.27 and not kw.get("betaPrior"):
        kw["fitType"] = _smooth_trend                            # the caller's trend; with replaceable samples the outlier phase in two halves
    nfm = None
    if rng.uniform() < 0.15 and not kw.get("betaPrior"):
        nfm = np.exp(rng.normal(0, 0.2, counts.shape)) * d["size_factors"][None, :]
        if rng.uniform() < 0.4:
            kw["minReplicatesForReplace"] = np.inf
    sfv = None if nfm is not None else d["size_factors"]
    tag = "seed %d: kind=%d n=%d m=%d p=%d weights=%d nf=%d %s%s%s" % (
        seed, kind, counts.shape[0], m, p, weights is not None, nfm is not None, kw.get("test", "Wald"),
        " reduced=%d" % kw["reduced"].shape[1] if "reduced" in kw else "", (" useT" if kw.get("useT") else " betaPrior" if kw.get("betaPrior") else "")
        + (" fitType=custom" if callable(kw.get("fitType")) else " fitType=mean" if kw.get("fitType") else ""))
    a = core.DESeqDataSet(counts, x, sizeFactors=sfv, normalizationFactors=nfm, weights=weights, engine=E)
    try:
        core.DESeq(a, **kw)
    except NotImplementedError as e:       # e.g. residual df <= 3: the prior-variance branch that needs R's RNG
        return tag + " SKIP " + str(e)[:60]
    except ValueError as e:                # the wrappers' NA guard (R/wrappers.R:31-34), e.g. a gene whose only non-zero
        if "contain NA" in str(e):         # counts carry weight 0: R stops there as well
            return tag + " SKIP NA guard"
        raise
    b = core.DESeqDataSet(counts, x, sizeFactors=sfv, normalizationFactors=nfm, weights=weights, engine=E)
    if rng.uniform() < 0.4:                                      # enqueue now, finish later (another analysis in between)
        try:
            fused.DESeq(b, wait=False, **kw)
            other = core.DESeqDataSet(counts[: max(8, counts.shape[0] // 3)], x, sizeFactors=sfv,
                                      normalizationFactors=None if nfm is None else nfm[: max(8, counts.shape[0] // 3)],
                                      weights=None if weights is None else weights[: max(8, counts.shape[0] // 3)], engine=E)
            try:
                fused.DESeq(other, wait=False, **kw)
            except Exception:                                    # noqa: BLE001  (the third of the rows may be degenerate)
                other = None
            fused.finish(b)
            if other is not None:
                try:
                    fused.finish(other)
                except Exception:                                # noqa: BLE001
                    pass
            tag += " pipelined"
        except Exception:
            raise
    else:
        fused.DESeq(b, **kw)
    if not b.attrs.get("fused"):           # (a failed parametric trend hands the analysis to core.DESeq)
        return tag + " SKIP not fused"
    _compare(a, b, tag)
    if not kw.get("useT") and not callable(kw.get("fitType")):
        _host_entry_check(b, counts, x, sfv, nfm, weights, kw, rng, tag)
        tag += " +host"
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
