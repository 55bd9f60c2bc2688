"""-m gpu: the whole DESeq() chain (host mirror of the R callers + the three HIP routines).
  * HostEngine(native) vs HostEngine(oracle): identical host glue, so every column of the
    result must be IDENTICAL (flags, iteration counts, dispersions, beta, SE, Wald, p).
  * DeviceEngine (HBM-resident, gene-major, torch glue for the O(n m) pre-steps) vs the
    oracle chain: the start values differ in the last bits (torch vs numpy QR), so the bar
    is north_star's: values within 1e-6 relative, flags equal."""
import numpy as np
import pytest

from deseq2_amd import core, simulate
from deseq2_amd.engine import DeviceEngine, HostEngine
from tests.helpers import assert_same

pytestmark = pytest.mark.gpu

COLS = ["dispGeneEst", "dispGeneIter", "dispFit", "dispMAP", "dispIter", "dispOutlier", "dispersion", "beta",
        "betaSE", "WaldStatistic", "WaldPvalue", "betaConv", "betaIter", "deviance"]


def _run(engine, d, x, weights=None):
    dds = core.DESeqDataSet(d["counts"], x, sizeFactors=d["size_factors"], weights=weights, engine=engine)
    return core.DESeq(dds)


@pytest.mark.parametrize("n,m,design", [(1000, 6, "two"), (400, 60, "bc")])
def test_chain_host_engine_identical_to_oracle(oracle, n, m, design):
    x = simulate.design_two_group(m) if design == "two" else simulate.design_batch_condition(m)
    d = simulate.make_counts(n, x, seed=21)
    a = _run(HostEngine(), d, x)
    b = _run(HostEngine(oracle), d, x)
    for k in COLS:
        assert_same(a.mcols[k], b.mcols[k], "DESeq()$" + k)
    assert a.dispersionFunction["coefficients"][0] == b.dispersionFunction["coefficients"][0]


def test_chain_with_weights_identical_to_oracle(oracle):
    m = 40
    x = simulate.design_two_group(m)
    d = simulate.make_counts(300, x, seed=22)
    rng = np.random.default_rng(3)
    w = rng.uniform(0.05, 1.0, d["counts"].shape)
    w[rng.uniform(size=w.shape) < 0.02] = 0.0
    a = _run(HostEngine(), d, x, weights=w)
    b = _run(HostEngine(oracle), d, x, weights=w)
    for k in COLS:
        assert_same(a.mcols[k], b.mcols[k], "DESeq(weights)$" + k)


def test_chain_device_resident_matches_oracle(oracle):
    """HBM-resident chain: every O(n m) step is now a HIP kernel with the oracle's arithmetic, so
    the whole result is identical -- only the Wald p-values go through a different erfc."""
    m = 100
    x = simulate.design_batch_condition(m)
    d = simulate.make_counts(600, x, seed=23)
    a = _run(DeviceEngine("cuda:0"), d, x)
    b = _run(HostEngine(oracle), d, x)
    for k in COLS:
        if k == "WaldPvalue":
            np.testing.assert_allclose(a.mcols[k], b.mcols[k], rtol=1e-10, atol=1e-300)
        else:
            assert_same(a.mcols[k], b.mcols[k], "device DESeq()$" + k)


def test_chain_beta_prior_weights_identical_to_oracle(oracle):
    """BASELINE configs[4] shape family: weights + betaPrior on the expanded design (p = 3,
    rank deficient, lambda = 1/sigma^2) -- every column identical to the oracle chain."""
    factors = {"condition": np.repeat([0, 1], 20)}
    x, _ = core.standard_model_matrix(factors)
    d = simulate.make_counts(300, x, seed=24)
    w = np.random.default_rng(4).uniform(0.05, 1.0, d["counts"].shape)
    w[np.random.default_rng(5).uniform(size=w.shape) < 0.02] = 0.0
    res = []
    for eng in (HostEngine(), HostEngine(oracle)):
        dds = core.DESeqDataSet(d["counts"], x, sizeFactors=d["size_factors"], weights=w, engine=eng)
        core.estimateDispersions(dds)
        core.nbinomWaldTest(dds, betaPrior=True, factors=factors)
        res.append(dds)
    for k in ("dispersion", "beta", "betaSE", "WaldStatistic", "betaIter", "betaConv", "MLE_beta"):
        assert_same(res[0].mcols[k], res[1].mcols[k], "betaPrior DESeq()$" + k)
    assert_same(res[0].attrs["betaPriorVar"], res[1].attrs["betaPriorVar"], "betaPriorVar")


def test_chain_lrt_identical_to_oracle(oracle):
    """BASELINE configs[3] family: nbinomLRT, full (6-level factor) vs a 2-column reduced model"""
    m = 36
    x = simulate.design_factor(m, 6)
    d = simulate.make_counts(250, x, seed=25)
    red = np.column_stack([np.ones(m), (np.arange(m) >= m // 2).astype(float)])
    res = []
    for eng in (HostEngine(), HostEngine(oracle)):
        dds = core.DESeqDataSet(d["counts"], x, sizeFactors=d["size_factors"], engine=eng)
        core.DESeq(dds, test="LRT", reduced=red)
        res.append(dds)
    for k in ("dispersion", "beta", "betaSE", "LRTStatistic", "LRTPvalue", "betaIter"):
        assert_same(res[0].mcols[k], res[1].mcols[k], "LRT DESeq()$" + k)


@pytest.mark.parametrize("k", [2, 3])
def test_pipelined_chunks_equal_serial(k):
    """DESeq() as k chunk threads on k HIP streams (parallel.Pipeline) == the serial device chain: per-gene
    fits are independent and the all-gene steps see the same vectors in the same order."""
    from deseq2_amd import parallel
    m = 60
    x = simulate.design_batch_condition(m)
    d = simulate.make_counts(900, x, seed=41)
    c = d["counts"]
    c[::45, 2] = 120000                      # some outliers: the per-chunk refit path runs too
    E = DeviceEngine("cuda:0")
    serial = core.DESeq(core.DESeqDataSet(c, x, sizeFactors=d["size_factors"], engine=E))
    pipe = parallel.Pipeline(E, n_chunks=k)
    for _ in range(2):                       # streams / workspaces are reused across calls
        shards = pipe.run(lambda lo, hi: core.DESeqDataSet(c[lo:hi], x, sizeFactors=d["size_factors"], engine=E),
                          c.shape[0])
    got = parallel.concat_mcols(shards, COLS + ["maxCooks", "replace"])
    for kk in COLS + ["maxCooks", "replace"]:
        assert_same(got[kk], serial.mcols[kk], "pipelined DESeq()$" + kk)
    assert serial.mcols["replace"].sum() >= 10
    assert shards[0].dispersionFunction["coefficients"][0] == serial.dispersionFunction["coefficients"][0]
