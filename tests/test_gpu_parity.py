"""-m gpu: HIP kernels vs the CPU oracle through the C ABI (host-pointer entry points,
R layout, exactly what the .Call shim passes).  Bar (BASELINE.json north_star):
convergence flags and iteration counts bit-exact; beta / SE / dispersions / statistics
within 1e-6 relative.  Because the kernels implement the oracle's arithmetic operation
for operation, the tests assert the stronger property: every output identical."""
import numpy as np
import pytest

from tests.helpers import assert_same, make_case

pytestmark = pytest.mark.gpu

BETA_KEYS = ["iter", "beta_mat", "beta_var_mat", "deviance", "contrast_num", "contrast_denom", "hat_diagonals"]
DISP_KEYS = ["iter", "iter_accept", "log_alpha", "last_change", "initial_lp", "initial_dlp", "last_lp",
             "last_dlp", "last_d2lp"]


def _fit_beta_both(oracle, d, alpha, lam, useW, useQR, maxit=100, tol=1e-8, minmu=0.5, contrast=None):
    from deseq2_amd import native
    p = d["x"].shape[1]
    contrast = np.r_[1.0, np.zeros(p - 1)] if contrast is None else contrast
    args = (d["counts"], d["x"], d["nf"], alpha, contrast, d["beta_init"], lam, d["weights"], useW, tol, maxit,
            useQR, minmu)
    return native.fitBeta(*args), oracle.fitBeta(*args)


CASES = [
    # n, m, design, weights, useQR
    (600, 6, "two_group", False, True),        # config C1 shape
    (500, 100, "two_group", False, True),      # config C2 shape
    (300, 500, "batch_condition", False, True),  # config C3 shape
    (300, 200, "two_group", True, True),       # config C5 pass 1 (weights)
    (300, 100, "two_group", False, False),     # useQR = FALSE (tests/testthat/test_QR.R)
    (200, 37, "batch_condition", True, False),
    (200, 70, ("factor", 6), False, True),     # p = 6, ragged m (not a multiple of 64)
    (100, 90, ("factor", 8), True, False),     # p = 8
    (120, 130, ("factor", 10), False, True),   # config C4 design width (p = 10)
    (48, 2000, ("factor", 10), False, True),   # config C4 shape: m = 2000, rows/X no longer fit LDS
    (64, 1500, "batch_condition", True, True),  # large m with weights: slab falls back to global scratch
]


@pytest.mark.parametrize("n,m,design,useW,useQR", CASES)
def test_fit_beta_matches_oracle(oracle, n, m, design, useW, useQR):
    d = make_case(n, m, design, seed=3, weights=useW, sf_random=True)
    p = d["x"].shape[1]
    lam = np.full(p, 1e-6) / np.log(2) ** 2           # R/fitNbinomGLMs.R:73,162
    got, want = _fit_beta_both(oracle, d, d["alpha_init"], lam, useW, useQR)
    for k in BETA_KEYS:
        assert_same(got[k], want[k], "fitBeta$" + k)
    assert (want["iter"] < 100).mean() > 0.9


@pytest.mark.parametrize("n,m,base,useW,useQR", [(300, 100, "two_group", False, True), (200, 500, "batch_condition", False, True),
                                                  (150, 70, ("factor", 6), True, True), (150, 90, "two_group", True, False),
                                                  (60, 1500, "batch_condition", False, True)])
def test_fit_beta_general_path_matches_oracle(oracle, n, m, base, useW, useQR):
    """a design with a CONTINUOUS covariate has one cell per sample: the general per-sample kernel (Householder QR by
    replay / normal equations) against the general path of the oracle -- the factor designs above all take the
    cell-collapsed kernel"""
    d = make_case(n, m, base, seed=5, weights=useW, sf_random=True)
    rng = np.random.default_rng(m)
    x = np.column_stack([d["x"], rng.normal(0.0, 0.3, m)])
    d["x"] = x
    d["beta_init"] = np.column_stack([d["beta_init"], np.zeros(d["beta_init"].shape[0])])
    p = x.shape[1]
    lam = np.full(p, 1e-6) / np.log(2) ** 2
    got, want = _fit_beta_both(oracle, d, d["alpha_init"], lam, useW, useQR)
    for k in BETA_KEYS:
        assert_same(got[k], want[k], "fitBeta(general)$" + k)
    want0 = oracle.fitBeta(d["counts"], x, d["nf"], d["alpha_init"], np.r_[1.0, np.zeros(p - 1)], d["beta_init"], lam,
                           d["weights"], useW, 1e-8, 100, useQR, 0.5, cell_mode=0)
    assert_same(want["iter"], want0["iter"], "more than 32 cells = the general path")


@pytest.mark.parametrize("n,m,base,useW", [(200, 100, "two_group", False), (150, 500, "batch_condition", False),
                                            (150, 70, ("factor", 6), True), (40, 1500, "batch_condition", False)])
def test_fit_disp_general_path_matches_oracle(oracle, n, m, base, useW):
    """a continuous covariate (one design cell per sample): the per-sample Cox-Reid Gram accumulation"""
    from deseq2_amd import native
    d = make_case(n, m, base, seed=6, weights=useW)
    rng = np.random.default_rng(m + 1)
    x = np.column_stack([d["x"], rng.normal(0.0, 0.3, m)])
    mu = np.maximum(d["nf"] * np.exp(d["beta_init"] @ d["x"].T), 0.5)
    w = np.maximum(d["weights"], 1e-6)
    la0 = np.log(d["alpha_init"])
    args = (d["counts"], x, mu, la0, la0 + 0.2, 0.8, np.log(1e-8 / 10), 1.0, 1e-6, 100, True, w, useW, 1e-2, True)
    got, want = native.fitDisp(*args), oracle.fitDisp(*args)
    for k in DISP_KEYS:
        assert_same(got[k], want[k], "fitDisp(general)$" + k)
    grid = np.linspace(np.log(1e-8), np.log(max(10, m)), 20)
    gargs = (d["counts"][:24], x, mu[:24], grid, la0[:24], 1.0, True, w[:24], useW, 1e-2, True)
    assert_same(native.fitDispGrid(*gargs)["log_alpha"], oracle.fitDispGrid(*gargs)["log_alpha"], "fitDispGrid(general)")


@pytest.mark.parametrize("n,m,design,useW", [(500, 100, "two_group", False), (300, 500, "batch_condition", False),
                                              (300, 200, "two_group", True), (600, 6, "two_group", False),
                                              (200, 70, ("factor", 6), True), (100, 130, ("factor", 10), False),
                                              (48, 2000, ("factor", 10), False), (64, 1500, "batch_condition", True)])
@pytest.mark.parametrize("usePrior", [False, True])
def test_fit_disp_matches_oracle(oracle, n, m, design, useW, usePrior):
    from deseq2_amd import native
    d = make_case(n, m, design, seed=4, weights=useW)
    p = d["x"].shape[1]
    lam = np.full(p, 1e-6) / np.log(2) ** 2
    fb = oracle.fitBeta(d["counts"], d["x"], d["nf"], d["alpha_init"], np.r_[1.0, np.zeros(p - 1)], d["beta_init"],
                        lam, d["weights"], useW, 1e-8, 100, True, 0.5)
    mu = np.maximum(d["nf"] * np.exp(fb["beta_mat"] @ d["x"].T), 0.5)     # R/core.R:763
    w = np.maximum(d["weights"], 1e-6)                                     # R/core.R:702
    la0 = np.log(d["alpha_init"])
    prior_mean = la0 + 0.3 if usePrior else la0
    args = (d["counts"], d["x"], mu, la0, prior_mean, 0.7 if usePrior else 1.0, np.log(1e-8 / 10), 1.0, 1e-6, 100,
            usePrior, w, useW, 1e-2, True)
    got = native.fitDisp(*args)
    want = oracle.fitDisp(*args)
    for k in DISP_KEYS:
        assert_same(got[k], want[k], "fitDisp$" + k)


def test_fit_disp_grid_matches_oracle(oracle):
    from deseq2_amd import native
    for (n, m, design, useW) in [(200, 100, "two_group", False), (150, 60, "batch_condition", True)]:
        d = make_case(n, m, design, seed=5, weights=useW)
        mu = np.maximum(d["nf"] * np.exp(d["beta_init"] @ d["x"].T), 0.5)
        grid = np.linspace(np.log(1e-8), np.log(max(10, m)), 20)            # R/wrappers.R:70-72
        nn = d["counts"].shape[0]
        for usePrior in (False, True):
            args = (d["counts"], d["x"], mu, grid, np.zeros(nn) - 1.0, 1.0, usePrior, d["weights"], useW, 1e-2, True)
            assert_same(native.fitDispGrid(*args)["log_alpha"], oracle.fitDispGrid(*args)["log_alpha"],
                        "fitDispGrid$log_alpha")


def test_known_answers_from_reference_tests(oracle):
    """tests/testthat/test_results.R:9,43-50 and test_optim.R:30-39 through the GPU."""
    from deseq2_amd import native
    yk = np.repeat([100, 200, 800], 4).astype(np.int32)[None, :]
    group = np.tile([1, 2], 6); cond = np.repeat([1, 2, 3], 4)
    X = np.column_stack([np.ones(12), group == 2, cond == 2, cond == 3]).astype(float)
    binit = np.linalg.lstsq(X, np.log(yk[0] + .1), rcond=None)[0][None, :]
    lam = np.full(4, 1e-6) / np.log(2) ** 2
    r = native.fitBeta(yk, X, np.ones((1, 12)), [0.05], [1, 0, 0, 0], binit, lam, np.ones((1, 12)), False, 1e-8,
                       100, True, 0.5)
    np.testing.assert_allclose(r["beta_mat"][0] / np.log(2), [np.log2(100), 0, 1, 3], atol=1e-6)
    yo = np.array([0, 0, 0, 0, 0, 1000, 1000, 0, 0, 0], np.int32)[None, :]
    Xo = np.column_stack([np.ones(10), np.repeat([0, 1], 5)]).astype(float)
    binit = np.linalg.lstsq(Xo, np.log(yo[0] + .1), rcond=None)[0][None, :]
    r = native.fitBeta(yo, Xo, np.ones((1, 10)), [2.0], [1, 0], binit, lam[:2], np.ones((1, 10)), False, 1e-8, 100,
                       True, 0.5)
    assert r["iter"][0] == 100


def test_maxit_zero_contrast_mode(oracle):
    """R/results.R:797-807: getContrast calls fitBeta with maxit = 0, useQR = FALSE"""
    d = make_case(200, 50, "batch_condition", seed=6)
    lam = np.full(4, 1e-6) / np.log(2) ** 2
    got, want = _fit_beta_both(oracle, d, d["alpha_init"], lam, False, False, maxit=0,
                               contrast=np.array([0.0, 1.0, -1.0, 0.5]))
    for k in BETA_KEYS:
        assert_same(got[k], want[k], "fitBeta(maxit=0)$" + k)
    assert (got["iter"] == 0).all()


def test_double_counts_and_bad_counts():
    from deseq2_amd import native, _lib
    d = make_case(50, 20, "two_group", seed=7)
    lam = np.full(2, 1e-6)
    a = native.fitBeta(d["counts"], d["x"], d["nf"], d["alpha_init"], [1, 0], d["beta_init"], lam, d["weights"],
                       False, 1e-8, 100, True, 0.5)
    b = native.fitBeta(d["counts"].astype(np.float64), d["x"], d["nf"], d["alpha_init"], [1, 0], d["beta_init"],
                       lam, d["weights"], False, 1e-8, 100, True, 0.5)
    assert_same(a["beta_mat"], b["beta_mat"], "REALSXP counts")
    bad = d["counts"].astype(np.float64); bad[3, 4] = 2.5
    with pytest.raises(_lib.DsqError):
        native.fitBeta(bad, d["x"], d["nf"], d["alpha_init"], [1, 0], d["beta_init"], lam, d["weights"], False,
                       1e-8, 100, True, 0.5)
    with pytest.raises(_lib.DsqError):   # p beyond compiled kernels must fail loudly, never fall back
        d60 = make_case(20, 80, "two_group", seed=7)
        x9 = np.column_stack([d60["x"]] + [np.random.default_rng(i).normal(size=80) for i in range(63)])   # p = 65 > DSQ_MAX_P
        native.fitBeta(d60["counts"], x9, d60["nf"], d60["alpha_init"], np.r_[1, np.zeros(64)],
                       np.zeros((d60["counts"].shape[0], 65)), np.full(65, 1e-6), d60["weights"], False, 1e-8, 100,
                       True, 0.5)


@pytest.mark.parametrize("n,m,design,useW", [(400, 100, "two_group", False), (300, 500, "batch_condition", True),
                                              (100, 130, ("factor", 10), False), (200, 7, "two_group", False),
                                              # long rows (m p >= 8192): the waves of a block share the Q / A tiles through LDS
                                              # (round 5); 61 genes: a last block with idle waves; ragged last tile
                                              (61, 900, ("factor", 10), False), (59, 700, ("factor", 12), True),
                                              (130, 2000, ("factor", 10), False), (33, 2100, "batch_condition", False)])
def test_prefit_moments_matches_oracle(oracle, n, m, design, useW):
    """extension: baseMean/baseVar/allZero, roughDispEstimate, IRLS start values"""
    from deseq2_amd import native
    d = make_case(n, m, design, seed=13, weights=useW, sf_random=True, drop_all_zero=False)
    got = native.prefitMoments(d["counts"], d["nf"], d["x"], d["weights"], useW)
    want = oracle.prefitMoments(d["counts"], d["nf"], d["x"], d["weights"], useW)
    for k in ("baseMean", "baseVar", "allZero", "roughDisp", "beta_init"):
        assert_same(got[k], want[k], "prefitMoments$" + k)


def test_nbinom_loglike_matches_oracle(oracle):
    """extension: nbinomLogLike, R/core.R:2208-2217"""
    from deseq2_amd import native
    for useW in (False, True):
        d = make_case(300, 90, "batch_condition", seed=14, weights=useW)
        mu = d["nf"] * np.exp(d["beta_init"] @ d["x"].T)           # unclamped, as R/fitNbinomGLMs.R:180
        got = native.nbinomLogLike(d["counts"], mu, d["alpha_init"], d["weights"], useW)
        want = oracle.nbinomLogLike(d["counts"], mu, d["alpha_init"], d["weights"], useW)
        assert_same(got, want, "nbinomLogLike")
        from scipy.stats import nbinom
        size = 1 / d["alpha_init"][:, None]
        ref = nbinom.logpmf(d["counts"], size, size / (size + mu))
        ref = (d["weights"] * ref if useW else ref).sum(axis=1)
        np.testing.assert_allclose(got, ref, rtol=1e-9)


def test_parametric_dispersion_fit_matches_oracle(oracle):
    """extension: the all-gene trend fit (R/core.R:2166-2190) as a one-workgroup kernel"""
    from deseq2_amd import native
    rng = np.random.default_rng(8)
    for n in (300, 5000, 70001):
        bm = np.exp(rng.normal(3, 1.5, n)); disp = (0.1 + 4 / bm) * np.exp(rng.normal(0, 0.5, n))
        assert_same(native.parametricDispersionFit(bm, disp), oracle.parametricDispersionFit(bm, disp),
                    "parametricDispersionFit n=%d" % n)
    with pytest.raises(RuntimeError):
        native.parametricDispersionFit(bm, 1e-3 + 0.5 * bm / bm.max())


@pytest.mark.parametrize("design,useW", [("two_group", False), ("batch_condition", True), (("factor", 3), False)])
def test_optim_rows_matches_oracle(oracle, design, useW):
    """dsq_optim_rows (fitNbinomGLMsOptim's rows, one wavefront each) == the oracle's iteration, every output"""
    from deseq2_amd import native
    d = make_case(40, 24, design, seed=12, weights=useW)
    y = d["counts"].copy()
    y[3] = 0; y[3, 5:9] = 1000                      # rows the IRLS cannot fit
    y[11] = 0; y[11, -1] = 7
    y[20, :12] = 0
    p = d["x"].shape[1]
    lam = np.full(p, 1e-6)
    lam[-1] = 0.5                                   # a coefficient with a real prior
    start = np.random.default_rng(3).normal(0, 1.0, (y.shape[0], p))
    args = (y, d["x"], d["nf"], d["alpha_init"], lam, d["weights"], useW, start, 0.5)
    got, want = native.optimRows(*args), oracle.optimRows(*args)
    for k in ("beta", "betaSE", "conv", "mu", "logLike"):
        assert_same(np.asarray(got[k], float), np.asarray(want[k], float), "optimRows$" + k)
    assert want["conv"].mean() > 0.9
