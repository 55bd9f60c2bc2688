"""-m gpu: the fused device-driven DESeq() chain (dsq_deseq_dev, deseq2_amd/fused.py) against the call-by-call chain
of core.py on the same engine -- every per-gene column, the assays and the dispersion function BIT FOR BIT -- over
the branches the reference's callers take: linear mu / GLM mu, fitDispGrid stragglers, optim-fallback rows,
all-zero rows, observation weights incl. rows whose weights fail, count outliers + refit, LRT against ~1."""
import numpy as np
import pytest

from deseq2_amd import core, fused, simulate
from deseq2_amd.engine import DeviceEngine
from tests.helpers import assert_same

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def E():
    return DeviceEngine("cuda:0")


def _both(E, counts, x, sf, weights=None, **kw):
    a = core.DESeqDataSet(counts, x, sizeFactors=sf, weights=weights, engine=E)
    core.DESeq(a, **kw)
    b = core.DESeqDataSet(counts, x, sizeFactors=sf, weights=weights, engine=E)
    assert fused.supported(b, **{k: v for k, v in kw.items() if k != "minReplicatesForReplace"})
    fused.DESeq(b, **kw)
    assert b.attrs.get("fused")
    return a, b


def _compare(a, b, what):
    E = a.engine
    keys = [k for k in a.mcols if k != "rowsForOptim"]
    assert set(keys) <= set(b.mcols) | {"rowsForOptim"}, (sorted(keys), sorted(b.mcols))
    for k in keys:
        va, vb = np.asarray(a.mcols[k]), np.asarray(b.mcols[k])
        assert_same(va.astype(np.float64), vb.astype(np.float64), "%s: mcols$%s" % (what, k))
    fa, fb = a.dispersionFunction, b.dispersionFunction
    if callable(fa["coefficients"]):              # the caller's trend: the function itself, on both sides
        assert fa["fitType"] == fb["fitType"] == "custom"
    else:
        assert_same(np.asarray(fa["coefficients"]), np.asarray(fb["coefficients"]), what + ": trend coefficients")
    assert fa["varLogDispEsts"] == fb["varLogDispEsts"] and fa["dispPriorVar"] == fb["dispPriorVar"], what
    nz = a.attrs.get("nz_rows")
    for k in ("mu", "H", "cooks"):
        ha, hb = E.to_numpy(a.assays[k]), E.to_numpy(b.assays[k])
        if nz is not None:
            hb = hb[nz]                      # the call-by-call chain keeps the assays of the non-zero rows
        assert_same(ha, hb, "%s: assays$%s" % (what, k))


def _spike_outliers(counts, rng, k=6):
    counts = counts.copy()
    rows = rng.choice(counts.shape[0], k, replace=False)
    for r in rows:
        counts[r, rng.integers(counts.shape[1])] = int(counts[r].max() * 40 + 1000)
    return counts


def test_wald_batch_condition_with_outlier_refit(E):
    x = simulate.design_batch_condition(48)                 # cells of 8 >= 7: replaceOutliers + refit
    d = simulate.make_counts(900, x, seed=3, size_factors=np.exp(np.random.default_rng(1).normal(0, .2, 48)))
    counts = _spike_outliers(d["counts"], np.random.default_rng(5))
    a, b = _both(E, counts, x, d["size_factors"])
    assert b.attrs["status"]["N_REPLACE"] >= 3 and b.attrs["status"]["N_REFIT"] >= 3
    assert a.mcols["replace"].sum() == b.attrs["status"]["N_REPLACE"]
    _compare(a, b, "batch+condition, outliers")
    rep = np.asarray(a.mcols["replace"], bool)
    assert_same(E.to_numpy(a.assays["replaceCounts"])[rep], E.to_numpy(b.assays["replaceCounts"])[rep], "replaceCounts")


def test_wald_two_group_linear_mu_no_replace(E):
    x = simulate.design_two_group(10)                       # groups == columns: linearModelMu; cells of 5: no refit
    d = simulate.make_counts(700, x, seed=4)
    a, b = _both(E, d["counts"], x, d["size_factors"])
    _compare(a, b, "two-group")
    assert "replace" not in b.mcols


def test_all_zero_rows_and_stragglers(E):
    x = simulate.design_batch_condition(24)
    d = simulate.make_counts(500, x, seed=6, drop_all_zero=False)
    counts = d["counts"].copy()
    counts[::41] = 0
    a, b = _both(E, counts, x, d["size_factors"], disp_maxit=4)       # 4 line-search steps: many grid refits
    assert b.attrs["status"]["N_GRID_GENEEST"] > 10 and b.attrs["status"]["N_GRID_MAP"] > 10
    assert b.mcols["allZero"].sum() >= 12
    _compare(a, b, "all-zero rows + stragglers")


def test_optim_fallback_rows(E):
    x = simulate.design_two_group(16)
    xb = np.column_stack([x, (np.arange(16) % 2).astype(float)])      # p = 3, groups != columns: GLM mu
    d = simulate.make_counts(400, xb, seed=8)
    counts = d["counts"].copy()
    # the reference's own non-convergence example (tests/testthat/test_optim.R:30-39), several times
    for r in (3, 77, 200):
        counts[r] = np.array([0, 0, 0, 0, 0, 1000, 1000, 0, 0, 0, 0, 0, 0, 0, 0, 2])
    a, b = _both(E, counts, xb, d["size_factors"], minReplicatesForReplace=np.inf)
    st = b.attrs["status"]
    assert st["N_OPTIM_GENEEST"] + st["N_OPTIM_TEST"] >= 1
    assert (~np.asarray(a.mcols["betaConv"], bool)).sum() == (~np.asarray(b.mcols["betaConv"], bool)).sum()
    _compare(a, b, "optim rows")


def test_weights_with_failing_rows(E):
    x = simulate.design_batch_condition(42)                 # cells of 7
    d = simulate.make_counts(500, x, seed=9)
    n = d["counts"].shape[0]
    rng = np.random.default_rng(2)
    w = rng.uniform(0.05, 1.0, (n, 42))
    w[rng.uniform(size=w.shape) < 0.02] = 0.0
    w[7, x[:, 3] == 1] = 0.0                                # gene 7 loses a whole condition: weightsFail
    counts = _spike_outliers(d["counts"], np.random.default_rng(6), k=4)
    a, b = _both(E, counts, x, d["size_factors"], weights=w)
    assert b.mcols["weightsFail"][7] and np.isnan(b.mcols["dispersion"][7])
    _compare(a, b, "weights")


def test_lrt_against_intercept(E):
    x = simulate.design_factor(60, 5)                       # cells of 12
    d = simulate.make_counts(600, x, seed=10, intercept_mean=2.0)
    counts = _spike_outliers(d["counts"], np.random.default_rng(7), k=5)
    red = np.ones((60, 1))
    a, b = _both(E, counts, x, d["size_factors"], test="LRT", reduced=red)
    _compare(a, b, "LRT")
    assert np.isfinite(b.mcols["LRTPvalue"]).all()


def test_lrt_against_a_reduced_model_matrix(E):
    """nbinomLRT with a reduced model that is not ~1 (R/core.R:1856-1868; BASELINE configs[3]'s second variant,
    SURVEY 8d: 2-column reduced model, minmu = 1e-6): the reduced fit runs inside the chain -- QR start values, IRLS,
    logLik, its own optim-fallback rows -- for the main pass and for the refit of the replaced rows"""
    x = simulate.design_factor(48, 6)
    d = simulate.make_counts(500, x, seed=31)
    counts = _spike_outliers(d["counts"], np.random.default_rng(2), k=5)
    counts[9] = 0
    counts[9, 28:30] = 3000
    red = np.column_stack([np.ones(48), (np.arange(48) >= 24).astype(float)])
    a, b = _both(E, counts, x, d["size_factors"], test="LRT", reduced=red, minmu=1e-6)
    _compare(a, b, "LRT vs 2-column reduced")
    assert np.isfinite(b.mcols["LRTPvalue"][~np.asarray(b.mcols["allZero"], bool)]).all()
    assert b.attrs["status"]["N_REFIT"] >= 2
    red3 = np.column_stack([red, (np.arange(48) % 2).astype(float)])          # not nested in x: still a valid fit
    a, b = _both(E, counts, x, d["size_factors"], test="LRT", reduced=red3, minReplicatesForReplace=np.inf)
    _compare(a, b, "LRT vs 3-column reduced")


def test_lrt_pvalues_computed_during_the_outlier_phase(E):
    """from 4096 genes up the LRT p-values (host pchisq, R/core.R:1878) are computed from an early side-stream copy of
    the log likelihoods while the device runs the outlier phase; the rows that phase refits are recomputed: same
    columns as the call-by-call chain, refitted rows included"""
    x = simulate.design_factor(32, 4)                       # cells of 8: replacement + refit
    d = simulate.make_counts(5000, x, seed=77)
    counts = _spike_outliers(d["counts"], np.random.default_rng(3), k=12)
    counts[::97] = 0
    for red in (np.ones((32, 1)), np.column_stack([np.ones(32), (np.arange(32) % 4 == 1).astype(float)])):
        a, b = _both(E, counts, x, d["size_factors"], test="LRT", reduced=red)
        _compare(a, b, "LRT, early p-values, reduced p=%d" % red.shape[1])
        assert b.attrs["status"]["N_REFIT"] >= 3
        refit = np.nan_to_num(np.asarray(b.mcols["replace"], float)) == 1
        assert refit.sum() >= 3 and np.isfinite(np.asarray(b.mcols["LRTPvalue"])[refit]).all()


def test_wald_t_distribution_pvalues(E):
    """useT = TRUE (R/core.R:1474-1503): p-values from the t distribution with m - p (or sum(weights) - p) degrees of
    freedom -- same statistic, the p-value column evaluated on the host from it"""
    x = simulate.design_batch_condition(24)
    d = simulate.make_counts(300, x, seed=17)
    a, b = _both(E, d["counts"], x, d["size_factors"], useT=True, minReplicatesForReplace=np.inf)
    _compare(a, b, "useT")
    w = np.random.default_rng(3).uniform(0.3, 1.0, d["counts"].shape)
    a, b = _both(E, d["counts"], x, d["size_factors"], weights=w, useT=True, minReplicatesForReplace=np.inf)
    _compare(a, b, "useT + weights")
    from scipy.stats import norm
    assert not np.allclose(b.mcols["WaldPvalue"], 2 * norm.sf(np.abs(b.mcols["WaldStatistic"])), rtol=1e-3)


@pytest.mark.parametrize("case", ["mean_asked_for", "mean_with_ties_and_outliers", "parametric_fails", "parametric_fails_lrt"])
def test_fit_type_mean_on_the_device(E, case):
    """estimateDispersionsFit(fitType = "mean") (R/core.R:894-899) inside the chain -- the trimmed mean by selection and an
    exact integer sum (trend_mean_kernel) against the mirror's Python integers -- and the analysis whose parametric trend
    does not fit (:885-893): the mean substituted on the device, no bounce to the call-by-call chain."""
    x = simulate.design_batch_condition(48) if "outliers" in case else simulate.design_two_group(12)
    kw = {}
    if case.startswith("mean"):
        d = simulate.make_counts(900, x, seed=31)
        counts, sf = d["counts"], d["size_factors"]
        kw["fitType"] = "mean"
        if "ties" in case:
            counts = _spike_outliers(counts, np.random.default_rng(2))
            counts[100:400] = counts[100]                    # 300 equal rows: equal estimates at the cut points' side
    else:
        counts, sf = simulate.make_counts_trend_fails(500, x, seed=5), np.ones(12)
        if case.endswith("lrt"):
            kw.update(test="LRT", reduced=np.ones((12, 1)))
    a, b = _both(E, counts, x, sf, **kw)
    assert a.dispersionFunction["fitType"] == b.dispersionFunction["fitType"] == "mean"
    _compare(a, b, case)
    if "outliers" not in case:         # (the refit re-estimates the replaced rows: their dispGeneEst is no longer the trend's input)
        dge = np.asarray(b.mcols["dispGeneEst"], float)
        assert b.dispersionFunction["coefficients"] == core.trimmed_mean_fit(dge[~np.isnan(dge)])
    fit = np.asarray(b.mcols["dispFit"], float)
    assert (fit[~np.isnan(fit)] == b.dispersionFunction["coefficients"]).all()


def _smooth_trend(means, disps):
    """a stand-in for localDispersionFit (locfit is R code): a running median of the log dispersions over the log means,
    returned as a function of the mean -- what matters to the chain is only that it is NOT of the parametric form"""
    o = np.argsort(means)
    lx, ly = np.log(means[o]), np.log(disps[o])
    k = max(5, lx.size // 25)
    knots = np.array([np.median(lx[i: i + k]) for i in range(0, lx.size - k + 1, k)])
    vals = np.array([np.median(ly[i: i + k]) for i in range(0, lx.size - k + 1, k)])
    return lambda q: np.exp(np.interp(np.log(q), knots, vals))


@pytest.mark.parametrize("case", ["wald", "lrt", "weights"])
def test_the_callers_trend_inside_the_chain(E, case):
    """a dispersion trend the library does not fit (fitType = "local": locfit, R/core.R:889-893; dispersionFunction<-,
    R/methods.R:142-190): gene-wise estimates up, the caller's function on the host, its values down as dispFit_in -- the
    prior variance from the residuals against them, the MAP search around them -- against the call-by-call chain with the
    same function.  No sample is replaceable in the first analysis; the second one replaces count outliers and refits
    their rows (the function is asked again, at the new means)."""
    x = simulate.design_two_group(12)
    d = simulate.make_counts(700, x, seed=41)
    counts = d["counts"].copy()
    counts[::61] = 0
    kw = dict(fitType=_smooth_trend)
    w = None
    if case == "lrt":
        kw.update(test="LRT", reduced=np.ones((12, 1)))
    if case == "weights":
        w = np.random.default_rng(3).uniform(0.2, 1.0, counts.shape)
    a, b = _both(E, counts, x, d["size_factors"], weights=w, **kw)
    assert a.dispersionFunction["fitType"] == b.dispersionFunction["fitType"] == "custom"
    for k in [k for k in a.mcols if k != "rowsForOptim"]:
        assert_same(np.asarray(a.mcols[k], np.float64), np.asarray(b.mcols[k], np.float64), "%s: mcols$%s" % (case, k))
    assert a.dispersionFunction["varLogDispEsts"] == b.dispersionFunction["varLogDispEsts"]
    assert a.dispersionFunction["dispPriorVar"] == b.dispersionFunction["dispPriorVar"]
    fit = np.asarray(b.mcols["dispFit"], float)
    bm = np.asarray(b.mcols["baseMean"], float)
    live = ~np.isnan(fit)
    assert_same(fit[live], b.dispersionFunction["coefficients"](bm[live]), "dispFit = the function at baseMean")
    for k in ("mu", "H", "cooks"):
        ha, hb = E.to_numpy(a.assays[k]), E.to_numpy(b.assays[k])
        nz = a.attrs.get("nz_rows")
        assert_same(ha, hb[nz] if nz is not None else hb, "%s: assays$%s" % (case, k))
    # with replaceable samples (round 5): the outlier phase in its two halves -- DSQ_PH_OUTLIERS_DETECT, the function at the NEW
    # means of the replaced rows (R/core.R:2512), DSQ_PH_OUTLIERS_REFIT -- against the call-by-call chain
    x2 = simulate.design_two_group(16)
    d2 = simulate.make_counts(500, x2, seed=42)
    c2 = _spike_outliers(d2["counts"], np.random.default_rng(8), k=6)
    kw2 = dict(kw)
    if case == "lrt":
        kw2.update(reduced=np.ones((16, 1)))
    w2 = None if w is None else np.random.default_rng(4).uniform(0.2, 1.0, c2.shape)
    a3, b3 = _both(E, c2, x2, d2["size_factors"], weights=w2, **kw2)
    assert b3.attrs["status"]["N_REFIT"] >= 1 and b3.dispersionFunction["fitType"] == "custom"
    for k in [k for k in a3.mcols if k != "rowsForOptim"]:
        assert_same(np.asarray(a3.mcols[k], np.float64), np.asarray(b3.mcols[k], np.float64), "%s + refit: mcols$%s" % (case, k))
    rf = np.asarray(b3.mcols["replace"], float) == 1
    fit3, bm3 = np.asarray(b3.mcols["dispFit"], float), np.asarray(b3.mcols["baseMean"], float)
    assert_same(fit3[rf], b3.dispersionFunction["coefficients"](bm3[rf]), "dispFit of the refitted rows = the function at their new means")


@pytest.mark.parametrize("m", [4, 5])
def test_residual_df_of_at_most_three_with_the_callers_prior_variance(E, m):
    """m - p <= 3 (two against two, three against two): R's estimateDispersionsPriorVar is a seeded Monte-Carlo match there
    (R/core.R:1155-1190, R's RNG and loess) -- the caller's job; with estimateDispersionsMAP's dispPriorVar argument
    (:989-994) the analysis runs on the chain, against the call-by-call chain and through the host entry"""
    from deseq2_amd import native
    x = simulate.design_two_group(m) if m % 2 == 0 else np.column_stack([np.ones(m), (np.arange(m) >= 3).astype(float)])
    d = simulate.make_counts(600, x, seed=51, intercept_mean=6.0)
    dds = core.DESeqDataSet(d["counts"], x, sizeFactors=d["size_factors"], engine=E)
    assert not fused.supported(dds)                               # without the prior variance: left to the caller
    with pytest.raises(NotImplementedError):
        core.DESeq(core.DESeqDataSet(d["counts"], x, sizeFactors=d["size_factors"], engine=E))
    a, b = _both(E, d["counts"], x, d["size_factors"], dispPriorVar=0.7)
    assert a.dispersionFunction["dispPriorVar"] == b.dispersionFunction["dispPriorVar"] == 0.7
    _compare(a, b, "df = %d" % (m - 2))
    with pytest.raises(Exception, match="residual degrees of freedom"):
        native.DESeq(d["counts"], x, d["size_factors"], assays=())
    first = native.DESeq(d["counts"], x, d["size_factors"], assays=(), geneEstOnly=True)
    assert_same(np.asarray(first["dispGeneEst"], float), np.asarray(b.mcols["dispGeneEst"], float), "geneEstOnly: dispGeneEst")
    res = native.DESeq(d["counts"], x, d["size_factors"], assays=(), dispPriorVar=0.7)
    for k, kb in (("dispMAP", "dispMAP"), ("dispersion", "dispersion"), ("beta", "beta"), ("stat", "WaldStatistic"), ("maxCooks", "maxCooks")):
        assert_same(np.asarray(res[k], float), np.asarray(b.mcols[kb], float), "host entry: " + k)
    assert res["dispersionFunction"]["dispPriorVar"] == 0.7


def test_all_gene_kernels_above_their_size_thresholds(E):
    """20 000 genes: the prior variance of the trend runs on sixteen workgroups from 16 384 genes (prior_var_grid_kernel:
    per-pass histograms in a global table, grid barriers), the ordered compactions take several rounds of four tiles --
    against the call-by-call chain, whose MAD and row lists come from other code (engine.mad, host masks): trend
    coefficients, varLogDispEsts, dispPriorVar and every per-gene column bit for bit"""
    x = simulate.design_two_group(12)
    d = simulate.make_counts(20000, x, seed=61, drop_all_zero=False)
    counts = d["counts"].copy()
    counts[::13] = 0                                        # all-zero rows inside every tile of the compaction
    a, b = _both(E, counts, x, d["size_factors"])
    assert b.n >= 16384 and int(np.asarray(b.mcols["allZero"], bool).sum()) > 1500
    _compare(a, b, "20 000 genes")
    # ... and the caller's-trend form of the same kernel (residuals against given values)
    a2, b2 = _both(E, counts, x, d["size_factors"], fitType=_smooth_trend)
    assert a2.dispersionFunction["varLogDispEsts"] == b2.dispersionFunction["varLogDispEsts"]
    assert a2.dispersionFunction["dispPriorVar"] == b2.dispersionFunction["dispPriorVar"]
    assert_same(np.asarray(a2.mcols["dispersion"], float), np.asarray(b2.mcols["dispersion"], float), "custom trend, 20 000 genes")


def test_unsupported_settings_fall_back(E):
    x = simulate.design_two_group(12)
    d = simulate.make_counts(200, x, seed=11)
    dds = core.DESeqDataSet(d["counts"], x, sizeFactors=d["size_factors"], engine=E)
    assert not fused.supported(dds, useOptim=False)
    assert not fused.supported(dds, betaPrior=True, test="LRT", reduced=np.ones((12, 1)))
    assert not fused.supported(dds, test="LRT", reduced=np.column_stack([x[:, 0], x[:, 0] * 2, x[:, 1]]))   # p_red >= p
    assert not fused.supported(dds, test="LRT", reduced=np.zeros((12, 1)))
    fused.DESeq(dds, useOptim=False)
    assert "WaldPvalue" in dds.mcols and not dds.attrs.get("fused")


@pytest.mark.parametrize("mode", ["expanded_weights", "expanded_three_levels", "standard", "given_variance"])
def test_beta_prior(E, mode):
    """nbinomWaldTest(betaPrior = TRUE) (R/core.R:1416-1432, R/fitNbinomGLMs.R:242-337; BASELINE configs[4]): MLE pass ->
    estimateBetaPriorVar on the host -> the pass with lambda = 1 / betaPriorVar on the expanded (rank-deficient) or the
    standard model matrix, for the main rows and for the refit of the replaced rows; MLE_beta kept"""
    rng = np.random.default_rng(12)
    weights = None
    kw = dict(betaPrior=True)
    if mode == "expanded_three_levels":
        factors = {"condition": np.repeat([0, 1, 2], 8)}
    else:
        factors = {"condition": np.repeat([0, 1], 10)}
    x, _ = core.standard_model_matrix(factors)
    d = simulate.make_counts(500, x, seed=44)
    counts = _spike_outliers(d["counts"], np.random.default_rng(6), k=5)
    counts[5] = 0
    counts[5, -2:] = 4000                                    # a row for the optim fallback
    if mode == "expanded_weights":
        weights = rng.uniform(0.05, 1.0, counts.shape)
        weights[rng.uniform(size=counts.shape) < 0.02] = 0.0
    if mode != "standard":
        kw["factors"] = factors
    if mode == "given_variance":
        kw["betaPriorVar"] = np.array([1e6, 0.8, 0.8])
    a, b = _both(E, counts, x, d["size_factors"], weights=weights, **kw)
    _compare(a, b, "betaPrior " + mode)
    assert_same(np.asarray(a.attrs["betaPriorVar"]), np.asarray(b.attrs["betaPriorVar"]), "betaPriorVar")
    assert b.mcols["beta"].shape[1] == (2 if mode == "standard" else x.shape[1] + 1)
    assert b.mcols["MLE_beta"].shape[1] == x.shape[1]
    assert b.attrs["status"]["N_REFIT"] >= 1 or weights is not None


def test_low_residual_df_and_rank_are_left_to_core(E):
    """with 1..3 residual degrees of freedom the reference's prior variance is the seeded Monte-Carlo
    matching of R/core.R:1155-1190 (not mirrored): the fused chain must not substitute the trigamma formula -- it hands
    the analysis to core.DESeq(), which raises; the device entry point refuses the trend phase; a rank-deficient
    design reports core's error."""
    import ctypes as C
    from deseq2_amd import _lib as L
    x = simulate.design_two_group(4)                       # 2 vs 2: m - p = 2
    d = simulate.make_counts(150, x, seed=4)
    dds = core.DESeqDataSet(d["counts"], x, sizeFactors=d["size_factors"], engine=E)
    assert not fused.supported(dds)
    with pytest.raises(NotImplementedError):
        fused.DESeq(dds)
    with pytest.raises(NotImplementedError):
        core.DESeq(core.DESeqDataSet(d["counts"], x, sizeFactors=d["size_factors"], engine=E))
    run = fused._Run(dds, "Wald", 7, 0, {})
    run.args.phases = L.DSQ_PH_TREND
    rc = L.lib().dsq_deseq_dev(C.byref(run.args), C.byref(run.out), None)
    assert rc == L.DSQ_ERR_UNSUPPORTED and b"residual degrees of freedom" in L.lib().dsq_last_error()
    xr = np.column_stack([simulate.design_two_group(12), simulate.design_two_group(12)[:, 1]])   # duplicated column
    d = simulate.make_counts(100, xr[:, :2], seed=5)
    dds = core.DESeqDataSet(d["counts"], xr, sizeFactors=d["size_factors"], engine=E)
    assert not fused.supported(dds)
    with pytest.raises(ValueError, match="not full rank"):
        fused.DESeq(dds)


@pytest.mark.parametrize("others_refit", [True, False])
def test_row_that_becomes_all_zero_by_replacement(E, others_refit):
    """a gene whose single non-zero count is an outlier: replaceOutliers turns the row into zeros (newAllZero,
    R/core.R:2492) -- it keeps its dispersion ("intermediate") columns and gets NA in the "results" columns only
    (:2534-2536); found by tests/gpu_fuzz_chain.py"""
    x = simulate.design_factor(32, 4)                       # cells of 8 >= 7
    d = simulate.make_counts(400, x, seed=21)
    # with other replaced rows the refit runs and the new all-zero rows get NA results; when EVERY replaced row became
    # all zero nothing is refit and nothing is overwritten (:2496, "handled by results()")
    counts = _spike_outliers(d["counts"], np.random.default_rng(8), k=4) if others_refit else d["counts"].copy()
    for g, j, v in ((17, 31, 21), (230, 5, 400)):
        counts[g] = 0
        counts[g, j] = v
    a, b = _both(E, counts, x, d["size_factors"])
    new_zero = np.asarray(a.mcols["allZero"], bool) & (np.nan_to_num(np.asarray(a.mcols["replace"], float)) == 1)
    assert new_zero.sum() >= 1
    assert np.isfinite(np.asarray(a.mcols["dispGeneIter"], float)[new_zero]).all()
    assert np.isnan(np.asarray(a.mcols["betaIter"], float)[new_zero]).all() == others_refit
    _compare(a, b, "newAllZero, others refit = %s" % others_refit)


def test_a_loop_of_analyses_allocates_nothing_once_warm(E):
    """the buffers of an analysis go back to the caching allocators by REFERENCE COUNT when the caller drops the object
    (no dds <-> run cycle waiting for the cyclic collector), so a loop of same-shape analyses -- bench.py's timed
    region -- makes no hipMalloc / hipHostMalloc call once warm (round 3: the driver measured 27.8 ms against 13.0)"""
    import gc
    import weakref
    import torch
    x = simulate.design_batch_condition(48)
    d = simulate.make_counts(600, x, seed=8)
    cr = torch.as_tensor(np.ascontiguousarray(d["counts"].T), device=E.device)
    nf = torch.ones((48, d["counts"].shape[0]), dtype=torch.float64, device=E.device)

    def step():
        dds = core.DESeqDataSet.from_device(E, cr, nf, x, sizeFactors=np.ones(48))
        fused.DESeq(dds)
        assert dds.attrs.get("fused")
        return dds
    gc.collect()
    gc.disable()
    try:
        dds = step()
        ref = weakref.ref(dds._fused_run)
        ptr = dds._fused_run.mu.data_ptr()
        dds = None
        assert ref() is None, "the run outlived its dataset: reference cycle"
        for _ in range(3):
            dds = None
            dds = step()
        assert dds._fused_run.mu.data_ptr() == ptr, "the n x m buffers were not recycled"
        torch.cuda.synchronize()
        a0 = torch.cuda.memory_stats(E.device)["num_device_alloc"]
        h0 = torch.cuda.host_memory_stats().get("num_host_alloc") if hasattr(torch.cuda, "host_memory_stats") else None
        for _ in range(5):
            dds = None
            dds = step()
        torch.cuda.synchronize()
        assert torch.cuda.memory_stats(E.device)["num_device_alloc"] == a0
        if h0 is not None:
            assert torch.cuda.host_memory_stats().get("num_host_alloc") == h0
    finally:
        gc.enable()


def test_pipelined_analyses_equal_the_waiting_ones(E):
    """fused.DESeq(wait=False) + fused.finish(): three analyses enqueued back to back (each one's result block copied on
    the side stream while the next chain runs), finished out of order -- every column, the trend and the assays equal
    the one-call-at-a-time results; an object dropped without finish() leaves nothing behind"""
    x = simulate.design_batch_condition(48)
    jobs = []
    for seed in (21, 22, 23):
        d = simulate.make_counts(700, x, seed=seed, size_factors=np.exp(np.random.default_rng(seed).normal(0, .2, 48)))
        jobs.append((_spike_outliers(d["counts"], np.random.default_rng(seed)), d["size_factors"]))
    want = []
    for counts, sf in jobs:
        w = core.DESeqDataSet(counts, x, sizeFactors=sf, engine=E)
        fused.DESeq(w)
        want.append(w)
    got = []
    for counts, sf in jobs:
        g = core.DESeqDataSet(counts, x, sizeFactors=sf, engine=E)
        assert fused.DESeq(g, wait=False) is g and "_fused_pending" in g.__dict__ and not g.mcols
        got.append(g)
    dropped = core.DESeqDataSet(jobs[0][0], x, sizeFactors=jobs[0][1], engine=E)
    fused.DESeq(dropped, wait=False)
    del dropped
    for k in (2, 0, 1):
        assert fused.finish(got[k]) is got[k] and "_fused_pending" not in got[k].__dict__
        assert fused.finish(got[k]) is got[k]                     # (a second call finds nothing to do)
        _compare(want[k], got[k], "pipelined analysis %d" % k)
    lrt = core.DESeqDataSet(jobs[1][0], x, sizeFactors=jobs[1][1], engine=E)
    fused.DESeq(lrt, test="LRT", reduced=np.ones((48, 1)), wait=False)
    fused.finish(lrt)
    ref = core.DESeqDataSet(jobs[1][0], x, sizeFactors=jobs[1][1], engine=E)
    fused.DESeq(ref, test="LRT", reduced=np.ones((48, 1)))
    _compare(ref, lrt, "pipelined LRT")


def test_pipelined_analyses_of_different_designs(E):
    """six analyses with DIFFERENT designs (p = 4 / 4 / 6 / 2 / 5 / 4, other cells, other sample counts) enqueued back to back before
    any is finished: each chain's small host-side tables (ridge, design cells, outlier metadata) must have left the host
    by the time the call returns -- the next call overwrites them -- and the library's device scratch is shared
    stream-ordered; results equal the one-call-at-a-time ones"""
    # (round 6: the tables go up through a pinned ring and only when a slot's bytes change -- designs 0, 1 and 5 have the same
    #  shape (48 samples, 4 columns) and different cells, 0 and 5 are the same design: A, B, ..., A in flight together)
    designs = [simulate.design_batch_condition(48), simulate.design_factor(48, 4), simulate.design_factor(42, 6), simulate.design_two_group(16),
               np.column_stack([simulate.design_batch_condition(36), np.random.default_rng(1).normal(size=36)]),
               simulate.design_batch_condition(48)]
    jobs = []
    for i, x in enumerate(designs):
        d = simulate.make_counts(400 + 150 * i, x, seed=90 + i, size_factors=np.exp(np.random.default_rng(i).normal(0, .2, x.shape[0])))
        jobs.append((_spike_outliers(d["counts"], np.random.default_rng(i), k=4), x, d["size_factors"]))
    want = []
    for counts, x, sf in jobs:
        w = core.DESeqDataSet(counts, x, sizeFactors=sf, engine=E)
        fused.DESeq(w)
        want.append(w)
    for _ in range(2):
        got = []
        for counts, x, sf in jobs:
            g = core.DESeqDataSet(counts, x, sizeFactors=sf, engine=E)
            fused.DESeq(g, wait=False)
            got.append(g)
        for k in range(len(jobs)):
            fused.finish(got[k])
            _compare(want[k], got[k], "pipelined, design %d" % k)
