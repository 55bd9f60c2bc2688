"""The dispersion-floor regime (Poisson-like genes whose start value is the minDisp clamp, alpha_0 = 1e-8).

There the reference's dlog_posterior multiplies a cancelling digamma sum by alpha^-2 = 1e16
(src/DESeq2.cpp:90-96): the number of line-search steps is rounding noise of whichever lgamma / digamma sits
underneath -- two correct implementations with different special functions (the C oracle's nmath restatement and
scipy's cephes under oracle/lapack_oracle.py) disagree on most such genes.  So the iteration count of a floor gene
cannot be pinned; what R's callers can SEE can, and that is what this module extracts and compares:

  fitBeta$iter                                                   (R/fitNbinomGLMs.R:191)
  dispGeneEst after the noIncrease rule and the [minDisp, maxDisp] clamp  (R/core.R:785, 826-830, 848)
  dispGeneEstConv = iter < maxit & iter != 1, refitDisp          (R/core.R:832-835)
  MAP dispConv = iter < maxit and the clamped dispMAP            (R/core.R:1048, 1100-1101)

Shared by tests/test_floor_regime.py (CPU: oracle) and tests/test_gpu_vs_lapack.py (HIP path).
"""
import numpy as np

from deseq2_amd import simulate
from tests.helpers import beta_init_qr, rough_alpha

SEEDS = (5, 6, 7)
NOISE = 1e-5


def floor_case(seed, n=500, m=60, frac_pois=0.6):
    """NB / Poisson mixture, ~batch + condition (p = 4): more than a quarter of the genes start at alpha_0 = 1e-8"""
    x = simulate.design_batch_condition(m)
    rng = np.random.Generator(np.random.PCG64(seed))
    d = simulate.make_counts(n, x, seed=seed)
    counts = d["counts"].copy()
    mu = 2.0 ** (d["beta"] @ x.T)
    pois = rng.uniform(size=counts.shape[0]) < frac_pois
    counts[pois] = rng.poisson(mu[pois]).astype(np.int32)
    counts = counts[counts.sum(axis=1) > 0]
    nf = np.ones(counts.shape)
    with np.errstate(all="ignore"):
        b0 = beta_init_qr(counts.astype(float), nf, x)
        a0 = rough_alpha(counts.astype(float), nf, x)
    return dict(counts=counts, x=x, nf=nf, beta_init=b0, alpha_init=a0, weights=np.ones(counts.shape))


def visible_chain(F, d, minDisp=1e-8, maxit=100):
    """estimateDispersionsGeneEst's calls and rules (R/core.R:755-848), then the MAP fitDisp (:1019-1063) on a fixed
    synthetic trend, through the three native routines of `F`"""
    y, x, nf, w = d["counts"].astype(float), d["x"], d["nf"], d["weights"]
    m = y.shape[1]
    p = x.shape[1]
    lam = np.full(p, 1e-6) / np.log(2) ** 2
    a0 = d["alpha_init"]
    fb = F.fitBeta(y, x, nf, a0, np.r_[1.0, np.zeros(p - 1)], d["beta_init"], lam, w, False, 1e-8, 100, True, 0.5)
    mu = np.maximum(nf * np.exp(np.asarray(fb["beta_mat"]) @ x.T), 0.5)
    la0 = np.log(a0)
    r = F.fitDisp(y, x, mu, la0, la0, 1.0, np.log(minDisp / 10), 1.0, 1e-6, maxit, False, w, False, 1e-2, True)
    maxDisp = max(10, m)
    it = np.asarray(r["iter"])
    dge = np.minimum(np.exp(np.asarray(r["log_alpha"])), maxDisp)                       # :785
    noInc = np.asarray(r["last_lp"]) < np.asarray(r["initial_lp"]) + np.abs(np.asarray(r["initial_lp"])) / 1e6
    dge[noInc] = a0[noInc]                                                               # :826-830
    conv = (it < maxit) & ~(it == 1)                                                     # :832
    refit = ~conv & (dge > minDisp * 10)                                                 # :835
    dge = np.minimum(np.maximum(dge, minDisp), maxDisp)                                  # :848
    bm = (y / nf).mean(axis=1)
    dfit = 0.1 + 2.0 / bm
    init = np.where(dge > 0.1 * dfit, dge, dfit)                                         # :1019-1021
    r2 = F.fitDisp(y, x, mu, np.log(init), np.log(dfit), 0.6, np.log(minDisp / 10), 1.0, 1e-6, maxit, True, w, False,
                   1e-2, True)
    it2 = np.asarray(r2["iter"])
    return dict(alpha_init=a0, beta_iter=np.asarray(fb["iter"]), iter=it, dispGeneEst=dge,
                dispGeneEstConv=conv, refitDisp=refit, map_iter=it2, dispConv=it2 < maxit,
                dispMAP=np.minimum(np.maximum(np.exp(np.asarray(r2["log_alpha"])), minDisp), maxDisp))


def rates(a, b):
    """agreement of two implementations on what R sees"""
    # the regime: start OR end in the noise region (step counts of two correct implementations differ up to final
    # dispersions of ~5e-6)
    fl = (b["alpha_init"] <= NOISE) | (a["dispGeneEst"] <= NOISE) | (b["dispGeneEst"] <= NOISE)
    rel = np.abs(a["dispGeneEst"] - b["dispGeneEst"]) / np.maximum(np.abs(b["dispGeneEst"]), 1e-300)
    return dict(n=int(fl.size), floor_start=int((b["alpha_init"] <= 1e-8).sum()), floor=int(fl.sum()),
                beta_iter_mismatch=int((a["beta_iter"] != b["beta_iter"]).sum()),
                iter_equal_floor=float((a["iter"][fl] == b["iter"][fl]).mean()) if fl.any() else 1.0,
                iter_equal_rest=float((a["iter"][~fl] == b["iter"][~fl]).mean()),
                conv_differs=int((a["dispGeneEstConv"] != b["dispGeneEstConv"]).sum()),
                refit_differs=int((a["refitDisp"] != b["refitDisp"]).sum()),
                dge_rel_gt_1e6=int((rel > 1e-6).sum()),
                dge_abs_max=float(np.abs(a["dispGeneEst"] - b["dispGeneEst"]).max()),
                map_conv_differs=int((a["dispConv"] != b["dispConv"]).sum()),
                map_iter_equal=float((a["map_iter"] == b["map_iter"]).mean()),
                map_rel_max=float((np.abs(a["dispMAP"] - b["dispMAP"]) / b["dispMAP"]).max()))


def assert_visible_parity(got, ref, name):
    """the budgets"""
    s = rates(got, ref)
    assert s["floor_start"] >= 0.25 * s["n"], "%s: only %d of %d genes start at the floor" % (name, s["floor_start"], s["n"])
    assert s["beta_iter_mismatch"] == 0, "%s: fitBeta$iter differs on %d genes" % (name, s["beta_iter_mismatch"])
    # clamped dispGeneEst: the floor genes end within rounding noise of the clamp
    assert s["dge_abs_max"] <= 1e-7, "%s: clamped dispGeneEst off by %.3g" % (name, s["dge_abs_max"])
    assert s["dge_rel_gt_1e6"] <= 0.03 * s["n"], "%s: %d genes beyond 1e-6 relative" % (name, s["dge_rel_gt_1e6"])
    # the decisions R takes on them
    assert s["refit_differs"] <= max(2, 0.005 * s["n"]), "%s: refitDisp differs on %d genes" % (name, s["refit_differs"])
    assert s["conv_differs"] <= 0.08 * s["n"], "%s: dispGeneEstConv differs on %d genes" % (name, s["conv_differs"])
    assert s["map_conv_differs"] == 0, "%s: MAP dispConv differs on %d genes" % (name, s["map_conv_differs"])
    assert s["map_iter_equal"] >= 0.99 and s["map_rel_max"] <= 1e-5, "%s: MAP %r" % (name, s)
    # away from the floor the strict claim holds: iteration counts equal (bar ulp-level ties, <= 1 %)
    assert s["iter_equal_rest"] >= 0.99, "%s: fitDisp$iter equal on %.3f of the genes above the floor" % (name, s["iter_equal_rest"])
    return s
