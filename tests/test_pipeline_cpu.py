"""CPU (no GPU): the host-side mirror of the reference's R callers, driven over the CPU
oracle instead of the MI355X engine (same function signatures).  This is BASELINE.json
configs[0]: makeExampleDESeqDataSet(n=1000, m=6) ~condition, full DESeq() -- plumbing."""
import numpy as np
import pytest

from deseq2_amd import core, simulate
from deseq2_amd.engine import HostEngine
from tests.optim_lbfgsb import fitNbinomGLMsOptim_scipy


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


# ---------------------------------------------------------------- round 2: the remaining a9 branches
def test_gene_est_model_matrix_override(oracle):
    """estimateDispersionsGeneEst(modelMatrix = ...) uses the SAME matrix for the GLM fit, the linear mu, the
    dispersion search and the grid refit (R/core.R:755-846): the override must equal a data set built on it"""
    m = 16
    x_own = simulate.design_two_group(m)
    x_alt = simulate.design_batch_condition(m)           # another width (p = 4 vs 2)
    d = simulate.make_counts(150, x_own, seed=5)
    E = HostEngine(oracle)
    for linearMu in (None, False):
        a = core.DESeqDataSet(d["counts"], x_own, sizeFactors=d["size_factors"], engine=E)
        core.estimateDispersionsGeneEst(a, modelMatrix=x_alt, maxit=3, linearMu=linearMu)     # maxit 3: grid refits
        b = core.DESeqDataSet(d["counts"], x_alt, sizeFactors=d["size_factors"], engine=E)
        core.estimateDispersionsGeneEst(b, maxit=3, linearMu=linearMu)
        np.testing.assert_array_equal(a.mcols["dispGeneIter"], b.mcols["dispGeneIter"])
        np.testing.assert_array_equal(a.mcols["dispGeneEst"], b.mcols["dispGeneEst"])
        np.testing.assert_array_equal(E.to_numpy(a.assays["mu"]), E.to_numpy(b.assays["mu"]))
        assert ((a.mcols["dispGeneIter"] >= 3) & (a.mcols["dispGeneEst"] > 1e-7)).any()        # the grid branch ran


def test_intercept_only_closed_form_clamps_mu(oracle):
    """~1 design with weights (or linearMu = FALSE): the closed-form branch of fitNbinomGLMs hands the dispersion
    search a mu clamped at minmu (fitMu[fitMu < minmu] <- minmu, R/core.R:763)"""
    m = 8
    rng = np.random.default_rng(3)
    counts = rng.poisson(0.3, size=(40, m)).astype(np.int32)
    counts[:, 0] += 1                                    # no all-zero rows
    x = np.ones((m, 1))
    E = HostEngine(oracle)
    dds = core.DESeqDataSet(counts, x, engine=E)
    core.estimateDispersionsGeneEst(dds, linearMu=False)
    mu = E.to_numpy(dds.assays["mu"])
    assert (mu >= 0.5).all() and (mu == 0.5).any()
    # the search started from that clamped mu: same result as handing fitDisp the clamped closed form directly
    cn = counts / 1.0
    mu_ref = np.maximum(np.repeat(cn.mean(1)[:, None], m, 1), 0.5)
    a0 = np.minimum(np.maximum(1e-8, np.minimum(dds.attrs["prefit"]["roughDisp"],
                    (dds.mcols["baseVar"] - dds.mcols["baseMean"]) / dds.mcols["baseMean"] ** 2)), 10)
    la = E.vlog(a0)
    r = oracle.fitDisp(counts, x, mu_ref, la, la, 1.0, np.log(1e-9), 1.0, 1e-6, 100, False, np.ones((40, m)), False,
                       1e-2, True)
    np.testing.assert_array_equal(r["iter"], dds.mcols["dispGeneIter"])


def test_gene_est_niter(oracle):
    """niter > 1 (R/core.R:751-847): second pass only over the rows whose dispersion moved by more than 0.05 on the
    log scale, with the new dispersion in the GLM weights; the noIncrease rule applies to niter == 1 only"""
    x = simulate.design_batch_condition(18)
    d = simulate.make_counts(200, x, seed=8)
    E = HostEngine(oracle)
    one = core.DESeqDataSet(d["counts"], x, sizeFactors=d["size_factors"], engine=E)
    core.estimateDispersionsGeneEst(one, niter=1)
    two = core.DESeqDataSet(d["counts"], x, sizeFactors=d["size_factors"], engine=E)
    core.estimateDispersionsGeneEst(two, niter=2)
    # restate the two passes by hand on the engine entry points
    pf = two.attrs["prefit"]
    a0 = np.minimum(np.maximum(1e-8, np.minimum(pf["roughDisp"], (two.mcols["baseVar"] - np.mean(1 / d["size_factors"]) *
                    two.mcols["baseMean"]) / two.mcols["baseMean"] ** 2)), 18)
    y, nf = two.y, two.nf
    lam = np.full(4, 1e-6) / np.log(2) ** 2
    con = np.r_[1.0, 0, 0, 0]

    optim = np.zeros(y.shape[0], bool)       # rows the IRLS leaves to the L-BFGS-B fallback: not restated here

    def glm_disp(rows, alpha):
        yy, nn = y[rows], nf[rows]
        fb = oracle.fitBeta(yy, x, nn, alpha, con, pf["beta_init"][rows], lam, np.ones_like(nn), False, 1e-8, 100, True, 0.5)
        optim[rows] |= (fb["iter"] >= 100) | (fb["beta_var_mat"] <= 0).any(axis=1) | np.isnan(fb["beta_mat"]).any(axis=1)
        mu = oracle.fittedMu(x, nn, fb["beta_mat"], 0.5)          # the engine's own mu = max(nf exp(x beta), minmu)
        la = E.vlog(alpha)
        return oracle.fitDisp(yy, x, mu, la, la, 1.0, np.log(1e-9), 1.0, 1e-6, 100, False, np.ones_like(nn), False, 1e-2, True)
    allrows = np.arange(y.shape[0])
    r1 = glm_disp(allrows, a0)
    a1 = np.minimum(E.vexp(r1["log_alpha"]), 18)
    moved = np.abs(E.vlog(a1) - E.vlog(a0)) > .05
    assert 0 < moved.sum() < moved.size
    r2 = glm_disp(np.where(moved)[0], a1[moved])
    a2 = a1.copy()
    a2[moved] = np.minimum(E.vexp(r2["log_alpha"]), 18)
    it = r1["iter"].copy()
    it[moved] = r2["iter"]
    assert optim.sum() < 5
    np.testing.assert_array_equal(two.mcols["dispGeneIter"][~optim], it[~optim])
    conv = (it < 100) & (it != 1)
    keep = (conv | ~(a2 > 1e-7)) & ~optim
    np.testing.assert_array_equal(two.mcols["dispGeneEst"][keep], np.minimum(np.maximum(a2, 1e-8), 18)[keep])
    assert (one.mcols["dispGeneEst"] != two.mcols["dispGeneEst"]).any()


def test_weights_rank_checks_match_per_gene_loop(oracle):
    """getAndCheckWeights' per-gene qr()$rank tests (R/core.R:2711-2722), vectorised over genes, against the loop"""
    from deseq2_amd.engine import _weights_ok_host
    rng = np.random.default_rng(11)
    for x in (simulate.design_two_group(10), simulate.design_batch_condition(12), simulate.design_factor(12, 4)):
        m, p = x.shape
        w = rng.uniform(0.0, 1.0, (60, m))
        w[rng.uniform(size=w.shape) < 0.4] = 0.0
        w[0] = 0.0
        w[1, 1:] = 0.0
        w[2] = 1.0
        w[3, x[:, -1] == 1] = 0.0
        w = w / np.maximum(w.max(axis=1, keepdims=True), 1e-300)
        want = np.ones(60, bool)
        for i in range(60):
            t1 = np.linalg.matrix_rank(w[i][:, None] * x) == p
            sub = x[w[i] > 1e-2]
            sub = sub[:, np.abs(sub).sum(axis=0) > 0]
            t2 = np.linalg.matrix_rank(sub) == sub.shape[1] if sub.size else True
            want[i] = t1 and t2
        np.testing.assert_array_equal(_weights_ok_host(w, x, 1e-2, True), want)
        assert 0 < want.sum() < 60


def test_weights_failing_rows_count_as_all_zero(oracle):
    """rows whose weights leave a degenerate design are flagged weightsFail and treated as all-zero (R/core.R:2736-2747)"""
    x = simulate.design_two_group(12)
    d = simulate.make_counts(80, x, seed=9)
    n = d["counts"].shape[0]
    w = np.ones((n, 12))
    w[5, x[:, 1] == 1] = 0.0                 # gene 5 loses one whole group
    dds = core.DESeqDataSet(d["counts"], x, sizeFactors=d["size_factors"], weights=w, engine=HostEngine(oracle))
    core.DESeq(dds, minReplicatesForReplace=np.inf)
    assert dds.mcols["weightsFail"][5] and dds.mcols["weightsFail"].sum() == 1
    assert dds.mcols["allZero"][5] and np.isnan(dds.mcols["dispersion"][5])
    assert np.isfinite(np.delete(dds.mcols["dispersion"], 5)).all()


def test_weights_rank_check_is_column_relative_like_dqrdc2():
    """qr() (LINPACK dqrdc2, tol = 1e-7) compares a column's residual norm with the column's OWN norm, so a
    design level whose weights are tiny but nonzero (1e-6 of the row maximum, zinbwave-style) keeps full rank in
    test1 (R/core.R:2716) -- only an exactly zero column is deficient -- while test2 (:2719-2721) sees the level
    dropped by the 1e-2 threshold together with its column and passes too."""
    from deseq2_amd.engine import _gram_rank, _weights_ok_host
    x = simulate.design_batch_condition(24)
    rng = np.random.default_rng(3)
    w = rng.uniform(0.2, 1.0, (6, 24))
    cond = x[:, 3] == 1
    w[1, cond] = 1e-6                      # tiny but nonzero: passes in R
    w[2, cond] = 0.0                       # a whole level weighted out: test1 fails
    w[3, cond] *= 1e-9
    w[4] = 0.0                             # nothing left
    w[5, x[:, 1] == 1] = 5e-3              # below the threshold: column dropped in test2, still fine
    G = np.einsum("nm,ma,mb->nab", w * w, x, x)
    assert _gram_rank(G).tolist() == [4, 4, 3, 4, 0, 4]
    assert _weights_ok_host(w, x, 1e-2, True).tolist() == [True, True, False, True, False, True]
    # rescaling a column of the design never changes the decision
    xs = x.copy(); xs[:, 2] *= 1e-5
    assert _gram_rank(np.einsum("nm,ma,mb->nab", w * w, xs, xs)).tolist() == [4, 4, 3, 4, 0, 4]
    # a numerically dependent column (beyond 1e-7 of its own norm) is rejected
    xd = np.column_stack([x, x[:, 1] + x[:, 2] + 1e-9 * rng.normal(size=24)])
    assert _gram_rank(np.einsum("ma,mb->ab", xd, xd)[None])[0] == 4


def test_optim_rows_against_lbfgsb(oracle):
    """the rows the IRLS leaves (R/fitNbinomGLMs.R:340-407): the engine's damped Fisher scoring reaches an objective
    value no worse than L-BFGS-B with optim's settings on the same rows, and the same coefficients where the optimum
    is well determined (the reference's own non-convergence example is flat along the separating direction)"""
    x = simulate.design_two_group(10)
    y = np.array([[0, 0, 0, 0, 0, 1000, 1000, 0, 0, 0],            # tests/testthat/test_optim.R:30-39
                  [0, 0, 0, 0, 0, 0, 0, 0, 0, 3],
                  [5, 0, 0, 0, 900, 0, 1, 0, 0, 2000],
                  [10, 12, 9, 11, 10, 100, 120, 90, 110, 95]], dtype=float)
    nf = np.ones_like(y)
    alpha = np.array([0.5, 0.2, 1.0, 0.05])
    lam = np.full(2, 1e-6)
    w = np.random.default_rng(1).uniform(0.2, 1.0, y.shape)
    for useW in (False, True):
        r = oracle.optimRows(y, x, nf, alpha, lam, w, useW, np.zeros((4, 2)))
        assert r["conv"].all()
        for i in range(4):
            xs, ok, obj = fitNbinomGLMsOptim_scipy(y[i], nf[i], x, lam, alpha[i], w[i], useW, np.zeros(2))
            assert obj(r["beta"][i]) <= obj(xs) + 1e-7 * abs(obj(xs))
        np.testing.assert_allclose(r["beta"][3], fitNbinomGLMsOptim_scipy(y[3], nf[3], x, lam, alpha[3], w[3], useW,
                                                                               np.zeros(2))[0], rtol=1e-4)
        mu = nf * 2.0 ** (r["beta"] @ x.T)
        np.testing.assert_allclose(r["mu"], mu, rtol=1e-12)
    # and through the chain: the reference's example gene ends with betaConv = TRUE after the fallback
    d = simulate.make_counts(60, x, seed=2)
    counts = d["counts"].copy()
    counts[7] = y[0]
    dds = core.DESeqDataSet(counts, x, sizeFactors=d["size_factors"], engine=HostEngine(oracle))
    core.DESeq(dds, minReplicatesForReplace=np.inf)
    assert 7 in set(dds.mcols["rowsForOptim"]) and dds.mcols["betaConv"][7]
    assert np.isfinite(dds.mcols["beta"][7]).all() and np.isfinite(dds.mcols["betaSE"][7]).all()


def test_optim_rows_conv_flag_against_lbfgsb_on_a_corpus(oracle):
    """`betaConv` of the rows the IRLS leaves comes from the engine's own stopping rule, not from
    L-BFGS-B's factr / maxit (lbfgsb.c is not in /root/reference, so these rows are PARITY-UNPINNED against R, see
    DESIGN.md / INTEGRATION.md).  Pinned here against scipy's L-BFGS-B run with optim's settings (tests/optim_lbfgsb.py)
    on a corpus of rows the IRLS does not converge on -- separated groups, single huge counts, near-empty rows, with and
    without weights: the convergence flag agrees on every row, the objective reached is no worse, and the coefficients
    agree where the optimum is determined (|beta| < 10: a separating direction is flat, there the optimum sits at an
    arbitrary large value in both)."""
    x = simulate.design_two_group(10)
    rng = np.random.default_rng(17)
    rows = [[0, 0, 0, 0, 0, 1000, 1000, 0, 0, 0],                    # tests/testthat/test_optim.R:30-39
            [0, 0, 0, 0, 0, 0, 0, 0, 0, 3],
            [0, 0, 0, 0, 0, 50, 40, 60, 55, 45],                      # complete separation
            [5, 0, 0, 0, 900, 0, 1, 0, 0, 2000],
            [1, 0, 0, 0, 0, 0, 0, 0, 0, 0],
            [100000, 2, 1, 0, 3, 1, 0, 2, 90000, 1],
            [3, 5, 2, 4, 3, 0, 0, 0, 0, 0],
            [0, 0, 0, 0, 700, 0, 0, 0, 0, 0]]
    for _ in range(8):                                                # heavy-tailed random rows
        r = rng.negative_binomial(0.05, 0.001, 10)
        if r.sum() > 0:
            rows.append(r.tolist())
    y = np.array(rows, dtype=float)
    n = y.shape[0]
    nf = np.exp(rng.normal(0, 0.2, (1, 10))) * np.ones_like(y)
    alpha = rng.uniform(0.05, 2.0, n)
    lam = np.full(2, 1e-6)
    w = rng.uniform(0.2, 1.0, y.shape)
    checked = 0
    for useW in (False, True):
        r = oracle.optimRows(y, x, nf, alpha, lam, w, useW, np.zeros((n, 2)))
        for i in range(n):
            xs, ok, obj = fitNbinomGLMsOptim_scipy(y[i], nf[i], x, lam, alpha[i], w[i], useW, np.zeros(2))
            assert bool(r["conv"][i]) == ok, "row %d (weights %s): conv %s vs L-BFGS-B %s" % (i, useW, r["conv"][i], ok)
            assert obj(r["beta"][i]) <= obj(xs) + 1e-6 * abs(obj(xs)), "row %d: objective" % i
            if (np.abs(xs) < 10).all() and (np.abs(r["beta"][i]) < 10).all():
                np.testing.assert_allclose(r["beta"][i], xs, rtol=2e-3, atol=2e-3, err_msg="row %d" % i)
                checked += 1
    assert checked >= 10       # a third of the (row, weights) pairs have a determined optimum


def test_from_device_defers_the_nf_matrix_when_size_factors_are_known():
    """DESeqDataSet.from_device: with size factors the n x m normalization-factor matrix is converted to the engine's
    layout only when a step asks for `dds.nf` (the fused chain reads the m-vector instead)"""
    from deseq2_amd import core

    class _Native:
        def __init__(self):
            self.calls = []

        def to_gene_major(self, t):
            self.calls.append(t)
            return ("gene-major", t)

    class _Engine:
        def __init__(self):
            self.native = _Native()

        def design(self, x):
            return x

    class _T:
        def __init__(self, name, shape):
            self.name, self.shape = name, shape

    m, n = 6, 11
    x = np.column_stack([np.ones(m), np.repeat([0, 1], 3)]).astype(float)
    E = _Engine()
    dds = core.DESeqDataSet.from_device(E, _T("counts", (m, n)), _T("nf", (m, n)), x, sizeFactors=np.ones(m))
    assert [t.name for t in E.native.calls] == ["counts"]
    assert dds.nf == ("gene-major", E.native.calls[-1]) and [t.name for t in E.native.calls] == ["counts", "nf"]
    assert dds.nf[0] == "gene-major" and len(E.native.calls) == 2          # converted once
    dds.nf = "override"
    assert dds.nf == "override"
    E2 = _Engine()
    core.DESeqDataSet.from_device(E2, _T("counts", (m, n)), _T("nf", (m, n)), x)   # no size factors: converted at once
    assert [t.name for t in E2.native.calls] == ["counts", "nf"]


class _Spy:
    """oracle fns with the fitBeta / fitDisp arguments DESeq() hands down recorded in call order"""

    def __init__(self, O):
        self._O, self.calls = O, []

    def __getattr__(self, k):
        return getattr(self._O, k)

    def fitBeta(self, y, x, nf, alpha_hat, contrast, beta_mat, lam, w, useWeights, tol, maxit, useQR, minmu, **kw):
        self.calls.append(("fitBeta", np.asarray(y).shape[0], np.asarray(x).shape[1], float(tol), int(maxit), float(minmu),
                           float(kw.get("mu_floor", 0.0))))
        return self._O.fitBeta(y, x, nf, alpha_hat, contrast, beta_mat, lam, w, useWeights, tol, maxit, useQR, minmu, **kw)

    def fitDisp(self, y, x, mu, *rest):
        self.calls.append(("fitDisp", np.asarray(y).shape[0], float(np.asarray(mu).min())))
        return self._O.fitDisp(y, x, mu, *rest)


def test_minmu_is_carried_the_way_the_reference_carries_it(oracle):
    """DESeq(minmu = v): estimateDispersions -> estimateDispersionsGeneEst floors the fitted means it hands to the
    search at v (R/core.R:393, R/methods.R:552, R/core.R:763) while the IRLS of that fitNbinomGLMs call keeps its default
    minmu = 0.5 (:755-757); nbinomWaldTest's fit clamps at v (:400, :1408); refitWithoutOutliers runs every step on
    its defaults (0.5, betaTol 1e-8, maxit 100; :2509-2531)."""
    x = simulate.design_batch_condition(48)                      # cells of 8: outlier replacement + refit; GLM mu (4 cells... p=4)
    x = np.column_stack([x, (np.arange(48) % 5 == 0).astype(float)])          # more cells than columns: no linear mu
    d = simulate.make_counts(200, x, seed=5, intercept_mean=1.0)   # low counts: fitted means below 0.5 exist
    counts = d["counts"].copy()
    counts[3, 7] = 50000                                           # a count outlier
    spy = _Spy(oracle)
    dds = core.DESeqDataSet(counts, x, sizeFactors=d["size_factors"], engine=HostEngine(spy))
    core.DESeq(dds, minmu=1e-3, betaTol=1e-6, maxit=50, minReplicatesForReplace=4)
    assert dds.mcols["replace"].sum() >= 1
    n = dds.n
    fb = [c for c in spy.calls if c[0] == "fitBeta"]
    fd = [c for c in spy.calls if c[0] == "fitDisp"]
    main = [c for c in fb if c[1] >= n - 1]
    refit = [c for c in fb if c[1] < n - 1]
    # gene-wise estimate: defaults inside the IRLS, the caller's minmu as the floor of mu-hat
    assert main[0][3:] == (1e-8, 100, 0.5, 1e-3)
    # the test's fit: the caller's settings
    assert main[-1][3:6] == (1e-6, 50, 1e-3)
    # the search saw means below 0.5 but not below the floor
    assert 1e-3 <= min(c[2] for c in fd if c[1] >= n - 1) < 0.5
    # the refit of the replaced rows: defaults everywhere
    assert refit and all(c[3:6] == (1e-8, 100, 0.5) for c in refit)
    assert all(c[6] in (0.0, 0.5) for c in refit)
    assert min(c[2] for c in fd if c[1] < n - 1) >= 0.5


def test_trimmed_mean_fit_is_the_correctly_rounded_mean_of_the_kept_values():
    """fitType = "mean" (R/core.R:894-899): mean(dispGeneEst[dispGeneEst > 10 minDisp], na.rm = TRUE, trim = 0.001).  The
    oracle's C restatement, core.py's Python-integer mirror and exact rational arithmetic agree bit for bit (the shared
    specification has no order of summation); base::mean.default's long-double two-pass mean -- restated with numpy's
    80-bit long double -- gives the same double up to the last bit."""
    from fractions import Fraction
    from oracle import oracle as O
    rng = np.random.default_rng(1)
    equal_to_r = 0
    for trial in range(40):
        n = int(rng.integers(5, 6000))
        d = np.exp(rng.normal(-3, 2, n))
        if trial % 3 == 0:
            d[rng.integers(0, n, n // 3)] = d[0]               # ties, also across the cut points
        if trial % 4 == 0:
            d[rng.integers(0, n, n // 10)] = np.nan            # all-zero rows
        if trial % 5 == 0:
            d[rng.integers(0, n, n // 10)] = 1e-8              # at the floor: not used
        a, b = O.trimmedMeanFit(d), core.trimmed_mean_fit(d)
        v = np.sort(d[d > 1e-7])
        k = int(np.floor(v.size * 0.001))
        kept = v[k: v.size - k]
        exact = float(sum(Fraction(x) for x in kept.tolist()) / kept.size)
        assert a == b == exact, (trial, a, b, exact)
        ld = np.asarray(kept, np.longdouble)                   # mean.default -> .Internal(mean(x)): summary.c real_mean
        s = ld.sum() / kept.size
        s = s + (ld - s).sum() / kept.size
        assert abs(float(s) - a) <= np.spacing(a)
        equal_to_r += float(s) == a
    assert equal_to_r >= 38
    # all kept values equal; a single value
    assert O.trimmedMeanFit(np.full(3000, 0.25)) == core.trimmed_mean_fit(np.full(3000, 0.25)) == 0.25
    assert O.trimmedMeanFit(np.array([0.3, np.nan, 1e-9])) == core.trimmed_mean_fit(np.array([0.3, np.nan, 1e-9])) == 0.3


def test_bench_parity_sample_on_the_oracle_chain(oracle):
    """bench.py's "parity" block (a sample of the step's own result re-fitted per gene by the oracle under the run's trend and
    prior variance) run on a result the oracle chain itself produced: every sampled row identical, iterations equal; and
    a perturbed result is reported as such"""
    import bench
    from deseq2_amd import core, simulate
    from deseq2_amd.engine import HostEngine
    x = simulate.design_batch_condition(18)
    sf = np.exp(np.random.Generator(np.random.PCG64(5)).normal(0, 0.25, 18))
    d = simulate.make_counts(500, x, seed=3, size_factors=sf)
    dds = core.DESeq(core.DESeqDataSet(d["counts"], x, sizeFactors=sf, engine=HostEngine(oracle)))
    W = {"counts": d["counts"], "sf": sf, "w": None}
    cfg = {"test": "Wald"}
    par = bench.parity_sample(dds, W, x, cfg, None, None, rows=64)
    assert par["rows"] == 64 and par["iter_equal"] == 1.0 and par["max_rel"] == 0.0, par
    dds.mcols["beta"] = dds.mcols["beta"] * (1 + 1e-9)
    dds.mcols["dispIter"] = dds.mcols["dispIter"] + 1
    par = bench.parity_sample(dds, W, x, cfg, None, None, rows=64)
    assert par["iter_equal"] == 0.0 and 0.5e-9 < par["max_rel"] < 2e-9 and par["worst_column"] == "beta", par
