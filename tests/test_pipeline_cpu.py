"""CPU (no GPU): the host-side mirror of the reference's R callers, driven over the CPU
oracle instead of the MI355X engine (same function signatures).  This is BASELINE.json
configs[0]: makeExampleDESeqDataSet(n=1000, m=6) ~condition, full DESeq() -- plumbing."""
import numpy as np
import pytest

from deseq2_amd import core, simulate
from deseq2_amd.engine import HostEngine


@pytest.fixture(scope="module")
def fitted(oracle):
    x = simulate.design_two_group(6)
    d = simulate.make_counts(1000, x, seed=1)
    dds = core.DESeqDataSet(d["counts"], x, sizeFactors=d["size_factors"], engine=HostEngine(oracle))
    core.DESeq(dds)
    return dds, d


def test_deseq_c1_plumbing(fitted):
    dds, d = fitted
    mc = dds.mcols
    for k in ("dispGeneEst", "dispFit", "dispMAP", "dispersion", "beta", "betaSE", "WaldStatistic", "WaldPvalue"):
        assert np.isfinite(mc[k]).all(), k
    assert dds.dispersionFunction["fitType"] == "parametric"
    a0, a1 = dds.dispersionFunction["coefficients"]
    assert 0.02 < a0 < 0.4 and 1.0 < a1 < 10.0          # true trend: 0.1 + 4/mean
    assert (mc["dispersion"] >= 1e-8).all() and (mc["dispersion"] <= 10).all()
    assert mc["betaConv"].mean() > 0.95
    # log2 fold changes recover the simulated ones for well-expressed genes
    hi = mc["baseMean"] > 50
    assert np.corrcoef(mc["beta"][hi, 1], d["beta"][hi, 1])[0, 1] > 0.8
    assert ((mc["WaldPvalue"] >= 0) & (mc["WaldPvalue"] <= 1)).all()
    np.testing.assert_allclose(mc["WaldStatistic"], mc["beta"] / mc["betaSE"])


def test_gene_est_rules(fitted):
    """accept/reject and clamp rules of R/core.R:826-848"""
    dds, _ = fitted
    mc = dds.mcols
    assert mc["dispGeneIter"].min() >= 1 and mc["dispGeneIter"].max() <= 100
    assert (mc["dispGeneEst"] >= 1e-8).all() and (mc["dispGeneEst"] <= 10).all()
    out = mc["dispOutlier"]
    np.testing.assert_array_equal(mc["dispersion"][out], mc["dispGeneEst"][out])
    np.testing.assert_array_equal(mc["dispersion"][~out], mc["dispMAP"][~out])


def test_lrt_reduced_intercept(oracle):
    x = simulate.design_batch_condition(12)
    d = simulate.make_counts(300, x, seed=2)
    dds = core.DESeqDataSet(d["counts"], x, sizeFactors=d["size_factors"], engine=HostEngine(oracle))
    core.DESeq(dds, test="LRT", reduced=x[:, :3])
    assert (dds.mcols["LRTStatistic"] > -0.05).all()   # minmu clamp in IRLS vs unclamped logLike (as in R)
    assert ((dds.mcols["LRTPvalue"] >= 0) & (dds.mcols["LRTPvalue"] <= 1)).all()
    dds2 = core.DESeqDataSet(d["counts"], x, sizeFactors=d["size_factors"], engine=HostEngine(oracle))
    core.DESeq(dds2, test="LRT", reduced=np.ones((12, 1)))
    assert (dds2.mcols["LRTStatistic"] >= dds.mcols["LRTStatistic"] - 0.05).all()   # nested models


def test_na_guard():
    with pytest.raises(ValueError, match="contain NA"):
        core._na_guard("fitBeta", alpha_hatSEXP=np.array([1.0, np.nan]))


def test_beta_prior_expanded_with_weights(oracle):
    """BASELINE configs[4] path: observation weights + betaPrior ridge on the expanded
    (rank-deficient) model matrix, R/fitNbinomGLMs.R:242-337 + R/expanded.R."""
    factors = {"condition": np.repeat([0, 1], 8)}
    x, names = core.standard_model_matrix(factors)
    d = simulate.make_counts(400, x, seed=5)
    w = np.random.default_rng(1).uniform(0.05, 1, d["counts"].shape)
    dds = core.DESeqDataSet(d["counts"], x, sizeFactors=d["size_factors"], weights=w, engine=HostEngine(oracle))
    core.estimateDispersions(dds)
    core.nbinomWaldTest(dds, betaPrior=True, factors=factors)
    bpv = dds.attrs["betaPriorVar"]
    assert bpv[0] == 1e6 and bpv[1] == bpv[2] and 0.1 < bpv[1] < 10          # levels share one prior variance
    b = dds.mcols["beta"]
    assert b.shape[1] == 3 and np.isfinite(b).all()
    # symmetric shrinkage: the two level effects are (nearly) opposite, and shrunk w.r.t. the MLE
    conv = dds.mcols["betaConv"]
    np.testing.assert_allclose(b[conv, 1], -b[conv, 2], atol=1e-4)   # equal up to the 1e-6 ridge on the intercept
    lfc = b[:, 2] - b[:, 1]
    mle = dds.mcols["MLE_beta"][:, 1]
    assert np.abs(lfc).mean() < np.abs(mle).mean()
    assert np.corrcoef(lfc[conv], mle[conv])[0, 1] > 0.9


def test_weighted_quantile_matches_unweighted():
    x = np.random.default_rng(2).normal(size=2000)
    assert core.Hmisc_wtd_quantile(np.abs(x), np.ones(x.size), 0.95) == pytest.approx(np.quantile(np.abs(x), 0.95))
    v = core.matchWeightedUpperQuantileForVariance(x, np.ones(x.size))
    assert 0.8 < v < 1.25


def test_optim_fallback_rows(oracle):
    """tests/testthat/test_optim.R:30-39: the IRLS reports iter == 100 for this row and the
    L-BFGS-B fallback (R/fitNbinomGLMs.R:340-407) then fits it; :2-27: IRLS == forced optim."""
    x = simulate.design_two_group(10)
    d = simulate.make_counts(100, x, seed=1)
    counts = d["counts"].copy()
    counts[0] = [0, 0, 0, 0, 0, 1000, 1000, 0, 0, 0]
    dds = core.DESeqDataSet(counts, x, sizeFactors=np.ones(10), engine=HostEngine(oracle))
    core.DESeq(dds)
    assert dds.mcols["betaIter"][0] == 100                       # IRLS gave up ...
    assert 0 in dds.mcols["rowsForOptim"]
    assert np.isfinite(dds.mcols["beta"][0]).all() and np.abs(dds.mcols["beta"][0]).max() <= 30   # ... optim fitted it
    assert np.isfinite(dds.mcols["betaSE"][0]).all() and np.isfinite(dds.mcols["deviance"][0])
    # forced optim agrees with IRLS on well-behaved rows (tolerance of the reference test: 1e-6 on a
    # scaled problem; L-BFGS-B's default stopping gives ~1e-4 here)
    dds2 = core.DESeqDataSet(d["counts"], x, sizeFactors=np.ones(10), engine=HostEngine(oracle))
    core.estimateDispersions(dds2)
    a = core.fitNbinomGLMs(dds2, lam=np.array([2.0, 2.0]))
    b = core.fitNbinomGLMs(dds2, lam=np.array([2.0, 2.0]), forceOptim=True)
    ok = a["betaConv"] & b["betaConv"]
    assert ok.mean() > 0.9
    np.testing.assert_allclose(a["betaMatrix"][ok], b["betaMatrix"][ok], atol=2e-3)
    np.testing.assert_allclose(a["betaSE"][ok], b["betaSE"][ok], rtol=2e-3)


def test_edge_cases_single_gene_and_intercept_only(oracle):
    """tests/testthat/test_edge_case.R: one gene; design ~1 (closed-form intercept fit)"""
    x = simulate.design_two_group(8)
    d = simulate.make_counts(50, x, seed=3)
    one = core.DESeqDataSet(d["counts"][:1], x, sizeFactors=np.ones(8), engine=HostEngine(oracle))
    core.estimateDispersionsGeneEst(one)
    one.mcols["dispersion"] = one.mcols["dispGeneEst"]       # the reference's advice when no trend can be fit
    core.nbinomWaldTest(one)
    assert one.mcols["beta"].shape == (1, 2) and np.isfinite(one.mcols["WaldStatistic"]).all()
    x1 = np.ones((8, 1))
    dds = core.DESeqDataSet(d["counts"], x1, sizeFactors=np.ones(8), engine=HostEngine(oracle))
    core.DESeq(dds)
    np.testing.assert_allclose(dds.mcols["beta"][:, 0], np.log2(d["counts"].mean(axis=1)), rtol=1e-12)
    assert (dds.mcols["betaIter"] == 1).all() and dds.mcols["betaConv"].all()
    with pytest.raises(ValueError, match="equal"):
        core.estimateDispersionsGeneEst(core.DESeqDataSet(d["counts"][:, :2], x[[0, 7]], engine=HostEngine(oracle)))


def test_all_zero_rows_are_set_aside(oracle):
    """objectNZ: genes without a single count get NA results, the others are what they are without them"""
    x = simulate.design_two_group(8)
    d = simulate.make_counts(120, x, seed=14)
    counts = d["counts"].copy()
    zero = np.array([0, 17, 63, counts.shape[0] - 1])
    counts[zero] = 0
    E = HostEngine(oracle)
    full = core.DESeq(core.DESeqDataSet(counts, x, sizeFactors=d["size_factors"], engine=E))
    nz = np.setdiff1d(np.arange(counts.shape[0]), zero)
    ref = core.DESeq(core.DESeqDataSet(counts[nz], x, sizeFactors=d["size_factors"], engine=E))
    for k in ("dispGeneEst", "dispersion", "beta", "betaSE", "WaldStatistic", "WaldPvalue", "deviance"):
        assert np.isnan(full.mcols[k][zero]).all(), k
        np.testing.assert_array_equal(full.mcols[k][nz], ref.mcols[k], err_msg=k)
    assert full.mcols["allZero"][zero].all() and (full.mcols["baseMean"][zero] == 0).all()
    np.testing.assert_array_equal(full.attrs["nz_rows"], nz)
