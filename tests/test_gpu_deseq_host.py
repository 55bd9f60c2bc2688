"""-m gpu: dsq_deseq -- DESeq() behind ONE host-pointer call (include/deseq2_mi355x.h, csrc/deseq_host.hip; what
r_shim.c binds as _DESeq2_mi355x_DESeq) -- against

  * the fused device chain driven from Python (deseq2_amd/fused.py): every per-gene column, the assays and the
    dispersion function BIT FOR BIT, also with the genes cut into 3 ranges inside the library (DSQ_HOST_SHARDS: the
    ranges exchange the trend's n-vectors through host memory, R/parallel.R:27-40);
  * the ORACLE chain: core.DESeq() over HostEngine(oracle) -- the CPU restatement of src/DESeq2.cpp under the Python
    mirror of the R callers -- directly (not through the HIP engine): every column identical, p-values to 1e-10.

and fused.DESeq() on the device directly against that oracle chain (VERDICT r2 #1b)."""
import os

import numpy as np
import pytest

from deseq2_amd import core, fused, native, simulate
from deseq2_amd.engine import DeviceEngine, HostEngine
from tests.helpers import assert_same

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def E():
    return DeviceEngine("cuda:0")


def _spike(counts, seed, k=6):
    rng = np.random.default_rng(seed)
    counts = counts.copy()
    for r in rng.choice(counts.shape[0], k, replace=False):
        counts[r, rng.integers(counts.shape[1])] = int(counts[r].max() * 40 + 1000)
    return counts


def _cases():
    x1 = simulate.design_batch_condition(48)                         # cells of 8 >= 7: replaceOutliers + refit
    d1 = simulate.make_counts(700, x1, seed=3, size_factors=np.exp(np.random.default_rng(1).normal(0, .2, 48)))
    c1 = _spike(d1["counts"], 5)
    c1[::53] = 0                                                     # all-zero rows
    x2 = simulate.design_two_group(12)                               # linear mu, no replacement (cells of 6)
    d2 = simulate.make_counts(500, x2, seed=7)
    c2 = d2["counts"].copy()
    c2[11] = [0, 0, 0, 0, 0, 0, 1000, 1000, 0, 0, 0, 0]              # IRLS does not converge: optim fallback rows
    c2[40] = [0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 3]
    x3 = simulate.design_factor(40, 5)                               # cells of 8, GLM mu (5 cells = p: linear) ...
    d3 = simulate.make_counts(400, x3, seed=9)
    x4 = simulate.design_factor(48, 6)                               # LRT against a 2-column reduced model (C4's second
    d4 = simulate.make_counts(450, x4, seed=25)                      # variant): cells of 8, refit included
    red4 = np.column_stack([np.ones(48), (np.arange(48) >= 24).astype(float)])
    c4 = _spike(d4["counts"], 4, 5)
    c4[7] = 0
    c4[7, 30:32] = 2500                                              # a row for the optim fallback in both fits
    # observation weights (incl. a gene whose weights leave a degenerate design -> weightsFail, treated as all zero)
    rng = np.random.default_rng(21)
    w5 = rng.uniform(0.05, 1.0, c2.shape)
    w5[rng.uniform(size=c2.shape) < 0.03] = 0.0
    w5[17, x2[:, 1] == 1] = 0.0
    # a normalization-factor matrix
    nf6 = np.exp(rng.normal(0, 0.2, d1["counts"].shape))
    c6 = d1["counts"].copy()
    c6[::41] = 0                 # all-zero rows: momentsDispEstimate averages the factors over the OTHER rows (objectNZ)
    # nbinomWaldTest(betaPrior = TRUE): the prior variance is estimated INSIDE the one call (csrc/beta_prior.hip); expanded
    # model matrix with weights (BASELINE configs[4]'s shape), a 5-level factor on the expanded and on the standard matrix
    # with replaced outliers (the refit reuses the prior variance, R/core.R:2521-2527)
    f2 = {"condition": x2[:, 1].astype(int)}
    f3 = {"group": (np.arange(40) * 5) // 40}
    c3s = _spike(d3["counts"], 12, 5)
    # WIDE designs (10 < p <= 24: the zero-padded kernel builds) inside the chain -- round 4
    x7 = simulate.design_factor(48, 12)                              # p = 12 on the 16-column build, cells of 4: no replacement
    d7 = simulate.make_counts(300, x7, seed=31)
    x8 = simulate.design_factor(136, 17)                             # p = 17 on the 24-column build, cells of 8: refit
    d8 = simulate.make_counts(260, x8, seed=32)
    c8 = _spike(d8["counts"], 33, 5)
    c8[5] = 0
    c8[5, 60:62] = 3000                                              # a row for the optim fallback
    # estimateDispersionsFit: fitType = "mean" asked for, and an analysis whose parametric trend does not fit (the mean
    # substituted on the device: DSQ_FIT_PARAMETRIC_OR_MEAN, what core.estimateDispersionsFit does in the reference's
    # locfit branch) -- round 4
    c9 = simulate.make_counts_trend_fails(420, x2, seed=5)
    return {"bc_outliers": (c1, x1, d1["size_factors"], {}),
            "bc_outliers_fit_mean": (c1, x1, d1["size_factors"], {"fitType": "mean"}),
            "two_group_trend_fails": (c9, x2, np.ones(12), {"host_fitType": "parametric_or_mean"}),
            "wide12_wald": (d7["counts"], x7, d7["size_factors"], {}),
            "wide12_lrt": (d7["counts"], x7, d7["size_factors"], {"test": "LRT"}),
            "wide17_outliers_optim": (c8, x8, d8["size_factors"], {}),
            "bp_two_group_expanded_weights": (c2, x2, d2["size_factors"], {"weights": w5, "betaPrior": True, "factors": f2}),
            "bp_factor5_expanded_outliers": (c3s, x3, d3["size_factors"], {"betaPrior": True, "factors": f3}),
            "bp_factor5_standard_outliers": (c3s, x3, d3["size_factors"], {"betaPrior": True, "factors": f3,
                                                                        "modelMatrixType": "standard"}),
            "two_group_weights": (c2, x2, d2["size_factors"], {"weights": w5}),
            "bc_nf_matrix": (c6, x1, None, {"normalizationFactors": nf6, "minReplicatesForReplace": np.inf}),
            # ... and with the outlier refit: momentsDispEstimate of the refitted subset re-averages the factors over ITS rows
            "bc_nf_matrix_outliers": (_spike(c6, 8), x1, None, {"normalizationFactors": nf6}),
            "factor6_lrt_reduced2": (c4, x4, d4["size_factors"], {"test": "LRT", "reduced": red4, "minmu": 1e-6}),
            "two_group_optim_rows": (c2, x2, d2["size_factors"], {}),
            "factor5_lrt": (_spike(d3["counts"], 2, 4), x3, d3["size_factors"], {"test": "LRT"}),
            "bc_no_replace": (d1["counts"], x1, d1["size_factors"], {"minReplicatesForReplace": np.inf})}


CASES = _cases()


def _host_entry(counts, x, sf, kw, assays=("mu", "H", "cooks")):
    return native.DESeq(counts, x, sf, test=kw.get("test", "Wald"), reduced=kw.get("reduced"), minmu=kw.get("minmu", 0.5),
                        normalizationFactors=kw.get("normalizationFactors"), weights=kw.get("weights"),
                        minReplicatesForReplace=kw.get("minReplicatesForReplace", 7), assays=assays,
                        betaPrior=kw.get("betaPrior", False), factors=kw.get("factors"), modelMatrixType=kw.get("modelMatrixType"),
                        fitType=kw.get("host_fitType", kw.get("fitType", "parametric")))


def _dataset(counts, x, sf, kw, engine):
    return core.DESeqDataSet(counts, x, sizeFactors=sf, normalizationFactors=kw.get("normalizationFactors"),
                             weights=kw.get("weights"), engine=engine)


def _chain_kw(kw, x):
    kw = {k: v for k, v in kw.items() if k not in ("weights", "normalizationFactors", "host_fitType")}
    if kw.get("test") == "LRT" and "reduced" not in kw:
        kw["reduced"] = np.ones((x.shape[0], 1))
    return kw


def _mcols_of(res, test):
    """the host entry's columns under the names core.DESeq() / fused.DESeq() use"""
    mc = {k: res[k] for k in ("baseMean", "baseVar", "dispGeneEst", "dispGeneIter", "dispFit", "dispMAP", "dispersion",
                              "dispIter", "dispOutlier", "beta", "betaSE", "betaIter", "maxCooks")}
    mc["allZero"] = res["allZero"]
    mc["deviance"] = -2 * res["logLike"]
    if test == "Wald":
        mc.update(WaldStatistic=res["stat"], WaldPvalue=res["pvalue"], betaConv=res["betaConv"])
    else:
        mc.update(LRTStatistic=2 * (res["logLike"] - res["logLikeReduced"]), fullBetaConv=res["betaConv"])
    if not np.isnan(res["replace"]).all():
        mc["replace"] = res["replace"]
    if np.nan_to_num(res["weightsFail"]).any():
        mc["weightsFail"] = res["weightsFail"]
    return mc


def _f(v):
    return np.asarray(v, dtype=np.float64)


@pytest.mark.parametrize("shards", [0, 3])
@pytest.mark.parametrize("name", sorted(CASES))
def test_host_entry_equals_fused_chain(E, name, shards):
    counts, x, sf, kw = CASES[name]
    test = kw.get("test", "Wald")
    b = _dataset(counts, x, sf, kw, E)
    fkw = _chain_kw(kw, x)
    assert fused.supported(b, **fkw)
    fused.DESeq(b, **fkw)
    assert b.attrs.get("fused")
    old = os.environ.get("DSQ_HOST_SHARDS")
    try:
        if shards:
            os.environ["DSQ_HOST_SHARDS"] = str(shards)
        res = _host_entry(counts, x, sf, kw, assays=("mu", "H", "cooks", "replaceCounts"))
    finally:
        if old is None:
            os.environ.pop("DSQ_HOST_SHARDS", None)
        else:
            os.environ["DSQ_HOST_SHARDS"] = old
    mc = _mcols_of(res, test)
    for k in sorted(mc):
        if k in b.mcols:
            assert_same(_f(mc[k]), _f(b.mcols[k]), "%s (%d ranges): %s" % (name, shards, k))
    for k in ("baseMean", "dispGeneEst", "dispersion", "beta", "betaSE", "maxCooks"):
        assert k in b.mcols
    if "replace" in b.mcols:
        assert_same(_f(mc["replace"]), _f(b.mcols["replace"]), name + ": replace")
    fa, fb = res["dispersionFunction"], b.dispersionFunction
    assert_same(fa["coefficients"], np.asarray(fb["coefficients"]), name + ": trend coefficients")
    assert fa["varLogDispEsts"] == fb["varLogDispEsts"] and fa["dispPriorVar"] == fb["dispPriorVar"]
    # assays: the fused chain keeps all rows; all-zero rows hold whatever the kernels left (never read): compare the rest
    nz = ~np.asarray(b.mcols["allZero"], bool) | (np.nan_to_num(_f(b.mcols.get("replace", np.zeros(b.n)))) == 1)
    for k in ("mu", "H", "cooks"):
        assert_same(res[k][nz], E.to_numpy(b.assays[k])[nz], "%s: assays$%s" % (name, k))
    if "replaceCounts" in b.assays:
        assert_same(res["replaceCounts"][nz], E.to_numpy(b.assays["replaceCounts"])[nz], name + ": replaceCounts")
    for k in ("N_NONZERO", "N_REPLACE", "N_REFIT", "N_OPTIM_GENEEST", "N_OPTIM_TEST", "N_GRID_GENEEST", "N_GRID_MAP"):
        assert res["status"][k] == b.attrs["status"][k], (name, k)
    if kw.get("betaPrior"):
        assert_same(res["betaPriorVar"], np.asarray(b.attrs["betaPriorVar"]), name + ": betaPriorVar")
        live = ~np.asarray(b.mcols["allZero"], bool)
        assert_same(_f(res["mle_beta"])[live], _f(b.mcols["MLE_beta"])[live], name + ": MLE_beta")
        assert res["beta"].shape[1] == (x.shape[1] + 1 if kw.get("modelMatrixType") != "standard" else x.shape[1])
    if name == "two_group_optim_rows":
        assert res["status"]["N_OPTIM_TEST"] >= 1
    if name == "bc_outliers":
        assert res["status"]["N_REFIT"] >= 3 and (~np.isnan(res["replace"])).any()


ORACLE_COLS = ["baseMean", "baseVar", "dispGeneEst", "dispGeneIter", "dispFit", "dispMAP", "dispIter", "dispOutlier",
               "dispersion", "beta", "betaSE", "betaIter", "deviance", "maxCooks", "replace"]


def _oracle_chain(oracle, counts, x, sf, kw):
    dds = _dataset(counts, x, sf, kw, HostEngine(oracle))
    core.DESeq(dds, **_chain_kw(kw, x))
    return dds


def _against_oracle(mc, o, name, test):
    cols = ORACLE_COLS + (["WaldStatistic", "betaConv"] if test == "Wald" else ["LRTStatistic", "fullBetaConv"])
    for k in cols:
        if k not in o.mcols:
            continue
        assert_same(_f(mc[k]), _f(o.mcols[k]), "%s vs the oracle chain: %s" % (name, k))
    if test == "Wald":      # the p-value goes through the engine's own pnorm on both sides: identical too
        np.testing.assert_allclose(_f(mc["WaldPvalue"]), _f(o.mcols["WaldPvalue"]), rtol=1e-10, atol=1e-300)


@pytest.mark.parametrize("name", sorted(CASES))
def test_host_entry_equals_oracle_chain(oracle, name):
    counts, x, sf, kw = CASES[name]
    test = kw.get("test", "Wald")
    res = _host_entry(counts, x, sf, kw, assays=())
    o = _oracle_chain(oracle, counts, x, sf, kw)
    _against_oracle(_mcols_of(res, test), o, name, test)
    assert o.dispersionFunction["fitType"] == res["dispersionFunction"]["fitType"]
    assert_same(np.asarray(o.dispersionFunction["coefficients"]), np.asarray(res["dispersionFunction"]["coefficients"]), name + ": trend")
    if "fit_mean" in name or "trend_fails" in name:
        assert res["dispersionFunction"]["fitType"] == "mean"
    assert o.dispersionFunction["dispPriorVar"] == res["dispersionFunction"]["dispPriorVar"]


@pytest.mark.parametrize("name", sorted(CASES))
def test_fused_chain_equals_oracle_chain_directly(E, oracle, name):
    """VERDICT r2 #1b: fused.DESeq() on the device against core.DESeq() over HostEngine(ORACLE) -- no HIP kernel on the
    reference side of the comparison"""
    counts, x, sf, kw = CASES[name]
    test = kw.get("test", "Wald")
    b = _dataset(counts, x, sf, kw, E)
    fused.DESeq(b, **_chain_kw(kw, x))
    assert b.attrs.get("fused")
    o = _oracle_chain(oracle, counts, x, sf, kw)
    _against_oracle(b.mcols, o, name, test)


def test_fused_chain_equals_oracle_chain_on_9000_genes(E, oracle):
    """an analysis with three times as many genes as the fit kernels keep resident waves (3 072), so
    that every persistent wave fits several genes in a row: ~ batch + condition on 48 samples (cells of 8: count outliers
    are replaced and their rows refitted), log-normal size factors, spiked counts and all-zero rows -- fused.DESeq() on the device against
    core.DESeq() over HostEngine(oracle), every column bit for bit; and the one-call host entry on the same analysis"""
    x = simulate.design_batch_condition(48)
    sf = np.exp(np.random.Generator(np.random.PCG64(90)).normal(0, 0.25, 48))
    d = simulate.make_counts(9000, x, seed=91, size_factors=sf)
    counts = _spike(d["counts"], 40, 4)
    counts[::997] = 0
    assert counts.shape[0] > 8192
    b = _dataset(counts, x, sf, {}, E)
    fused.DESeq(b)
    assert b.attrs.get("fused")
    o = _oracle_chain(oracle, counts, x, sf, {})
    _against_oracle(b.mcols, o, "9000 genes", "Wald")
    res = _host_entry(counts, x, sf, {}, assays=())
    mc = _mcols_of(res, "Wald")
    for k in sorted(mc):
        if k in b.mcols:
            assert_same(_f(mc[k]), _f(b.mcols[k]), "9000 genes, host entry: " + k)


@pytest.mark.parametrize("name", ["bc_outliers", "factor6_lrt_reduced2", "bp_factor5_expanded_outliers"])
def test_host_entry_in_eight_ranges(E, name):
    """dsq_deseq with the genes cut into EIGHT ranges inside the library (DSQ_HOST_SHARDS=8: the multi-device walk of the
    host entry -- per-range chains, the trend over the gathered vectors, the global refit count -- on one device):
    identical to the single-range call, column by column"""
    counts, x, sf, kw = CASES[name]
    one = _host_entry(counts, x, sf, kw, assays=("mu", "cooks"))
    old = os.environ.get("DSQ_HOST_SHARDS")
    try:
        os.environ["DSQ_HOST_SHARDS"] = "8"
        eight = _host_entry(counts, x, sf, kw, assays=("mu", "cooks"))
    finally:
        if old is None:
            os.environ.pop("DSQ_HOST_SHARDS", None)
        else:
            os.environ["DSQ_HOST_SHARDS"] = old
    for k in sorted(one):
        if isinstance(one[k], np.ndarray):
            assert_same(_f(eight[k]), _f(one[k]), "%s in 8 ranges: %s" % (name, k))
    assert eight["dispersionFunction"]["dispPriorVar"] == one["dispersionFunction"]["dispPriorVar"]


@pytest.mark.parametrize("shards", [0, 3])
def test_the_callers_trend_in_two_host_calls(E, shards):
    """what the R patch does for fitType = "local" (INTEGRATION.md section 4): dsq_deseq(geneEstOnly) -> the caller's trend at
    baseMean -> dsq_deseq(dispFit = ...) -- against the fused chain with the same function; every other column of the first
    call is NA; count outliers are flagged (Cook's distances) but not replaced"""
    from tests.test_gpu_fused import _smooth_trend
    counts, x, sf, _ = CASES["bc_outliers"]                    # cells of 8: replaceable samples, were the refit the library's
    old = os.environ.get("DSQ_HOST_SHARDS")
    try:
        if shards:
            os.environ["DSQ_HOST_SHARDS"] = str(shards)
        first = native.DESeq(counts, x, sf, assays=(), geneEstOnly=True)
        bm, dge = first["baseMean"], first["dispGeneEst"]
        assert np.isnan(first["dispFit"]).all() and np.isnan(first["dispersion"]).all() and np.isnan(first["beta"]).all()
        use = dge > 1e-6
        f = _smooth_trend(bm[use], dge[use])
        with np.errstate(invalid="ignore", divide="ignore"):
            fit = f(bm)
        res = native.DESeq(counts, x, sf, assays=("mu", "cooks"), dispFit=fit)
    finally:
        if old is None:
            os.environ.pop("DSQ_HOST_SHARDS", None)
        else:
            os.environ["DSQ_HOST_SHARDS"] = old
    assert res["dispersionFunction"]["fitType"] == "given" and np.isnan(res["replace"]).all()
    b = _dataset(counts, x, sf, {}, E)
    fused.DESeq(b, fitType=_smooth_trend, minReplicatesForReplace=np.inf)
    assert b.attrs.get("fused")
    assert_same(_f(first["baseMean"]), _f(b.mcols["baseMean"]), "first call: baseMean")
    assert_same(_f(first["dispGeneEst"]), _f(b.mcols["dispGeneEst"]), "first call: dispGeneEst")
    mc = _mcols_of(res, "Wald")
    for k in sorted(mc):
        if k in b.mcols:
            assert_same(_f(mc[k]), _f(b.mcols[k]), "second call (%d ranges): %s" % (shards, k))
    assert res["dispersionFunction"]["varLogDispEsts"] == b.dispersionFunction["varLogDispEsts"]
    assert res["dispersionFunction"]["dispPriorVar"] == b.dispersionFunction["dispPriorVar"]
    nz = ~np.asarray(b.mcols["allZero"], bool)
    for k in ("mu", "cooks"):
        assert_same(res[k][nz], E.to_numpy(b.assays[k])[nz], "second call: assays$" + k)
    assert (res["maxCooks"][nz] > res["cooksCutoff"]).any()     # the outliers are there for the caller's refitWithoutOutliers


def test_a_parametric_trend_that_does_not_fit_is_reported():
    """fitType = "parametric" (DSQ_FIT_PARAMETRIC): the library does not substitute anything -- DSQ_ERR_FIT, and the R caller
    takes the reference's own route (locfit, R/core.R:885-893)"""
    counts, x, sf, _ = CASES["two_group_trend_fails"]
    with pytest.raises(Exception, match="did not fit"):
        native.DESeq(counts, x, sf, assays=())
    with pytest.raises(ValueError, match="fitType"):
        native.DESeq(counts, x, sf, assays=(), fitType="local")


def test_a_gene_range_of_all_zero_rows_is_legal():
    """R/parallel.R fits the trend on the gathered object: a shard whose rows are all zero joins the exchange with
    nothing to contribute (the per-shard N_NONZERO check must be global)"""
    counts, x, sf, kw = CASES["bc_outliers"]
    counts = counts.copy()
    n = counts.shape[0]
    counts[n // 3: 2 * (n // 3) + 5] = 0                       # the whole middle range (and a bit of the last)
    one = _host_entry(counts, x, sf, kw, assays=())
    old = os.environ.get("DSQ_HOST_SHARDS")
    try:
        os.environ["DSQ_HOST_SHARDS"] = "3"
        three = _host_entry(counts, x, sf, kw, assays=())
    finally:
        if old is None:
            os.environ.pop("DSQ_HOST_SHARDS", None)
        else:
            os.environ["DSQ_HOST_SHARDS"] = old
    for k in ("baseMean", "dispGeneEst", "dispFit", "dispersion", "beta", "betaSE", "stat", "pvalue", "maxCooks", "allZero",
              "betaConv", "replace", "logLike"):
        assert_same(_f(three[k]), _f(one[k]), "all-zero range: " + k)
    assert three["status"]["N_NONZERO"] == one["status"]["N_NONZERO"] < n - n // 3


def test_host_entry_argument_errors():
    counts, x, sf, kw = CASES["two_group_optim_rows"]
    from deseq2_amd import _lib as L
    with pytest.raises(L.DsqError, match="residual degrees of freedom"):
        native.DESeq(counts[:, :4], simulate.design_two_group(4), sf[:4])
    with pytest.raises(L.DsqError, match="design columns"):
        native.DESeq(np.ones((5, 80), dtype=np.int32), np.column_stack([np.ones(80)] + [np.cos(np.arange(80.0) * k) for k in range(1, 65)]),
                     np.ones(80))                                   # 65 columns: beyond the widest (64-column) build
    with pytest.raises(L.DsqError, match="zero counts"):
        native.DESeq(np.zeros((20, 12), dtype=np.int32), x, sf)
    bad = counts.astype(np.float64)
    bad[3, 2] = 1.5
    with pytest.raises(L.DsqError, match="non-integer"):
        native.DESeq(bad, x, sf)
