"""CPU (no GPU): the count-outlier machinery -- the oracle's Cook's distance /
replaceOutliers restatements against an independent numpy statement of R/core.R:2277-2359 and
2069-2115, and the host mirror (calculateCooksDistance, replaceOutliers, refitWithoutOutliers)
driven over the oracle, following tests/testthat/test_outlier.R."""
import numpy as np
import pytest
from scipy.stats import f as fdist

from deseq2_amd import core, simulate
from deseq2_amd.engine import HostEngine


def r_trim_mean(v, trim):
    """R's mean(x, trim): order statistics floor(n trim)+1 .. n - floor(n trim)"""
    v = np.sort(v)
    lo = int(np.floor(v.size * trim))
    return v[lo: v.size - lo].mean()


def np_cooks(counts, nf, mu, H, x):
    counts = counts.astype(float)
    n, m = counts.shape
    p = x.shape[1]
    _, inv, cnt = np.unique(x, axis=0, return_inverse=True, return_counts=True)
    inv = inv.reshape(-1)
    cn = counts / nf
    trimratio, scale = (1 / 3, 1 / 4, 1 / 8), (2.04, 1.86, 1.51)

    def tf(k):
        return 0 if k <= 3 else (1 if k <= 23 else 2)
    if (cnt >= 3).any():
        vs = []
        for c in np.where(cnt >= 3)[0]:
            sub = cn[:, inv == c]
            k = tf(sub.shape[1])
            cm = np.array([r_trim_mean(r, trimratio[k]) for r in sub])
            vs.append(scale[k] * np.array([r_trim_mean(r, trimratio[k]) for r in (sub - cm[:, None]) ** 2]))
        v = np.max(vs, axis=0)
    else:
        rm = np.array([r_trim_mean(r, 1 / 8) for r in cn])
        v = 1.51 * np.array([r_trim_mean(r, 1 / 8) for r in (cn - rm[:, None]) ** 2])
    mean = cn.mean(axis=1)
    alpha = np.maximum((v - mean) / mean ** 2, 0.04)
    V = mu + alpha[:, None] * mu ** 2
    ck = (counts - mu) ** 2 / V / p * H / (1 - H) ** 2
    use = (cnt >= 3)[inv]
    mx = ck[:, use].max(axis=1) if (m > p and use.any()) else np.full(n, np.nan)
    return ck, mx, alpha


def _inputs(n, x, seed):
    d = simulate.make_counts(n, x, seed=seed)
    c = d["counts"]
    rng = np.random.default_rng(seed)
    sf = np.exp(rng.normal(0, 0.2, x.shape[0]))
    nf = np.broadcast_to(sf[None, :], c.shape).copy()
    mu = np.maximum(c.mean(axis=1, keepdims=True) * nf, 0.5)
    H = rng.uniform(0.01, 0.6, c.shape)
    return c, nf, mu, H


@pytest.mark.parametrize("m,design", [(6, "two"), (12, "bc"), (24, "bc"), (70, "two"), (5, "unrep"), (130, "two")])
def test_oracle_cooks_vs_numpy(oracle, m, design):
    """trim classes n<=3 (1/3), <=23 (1/4), >23 (1/8); a design without replicates takes trimmedVariance"""
    if design == "two":
        x = simulate.design_two_group(m)
    elif design == "bc":
        x = simulate.design_batch_condition(m)
    else:
        x = np.column_stack([np.ones(m), np.arange(m, dtype=float) / m])
    c, nf, mu, H = _inputs(150, x, seed=m)
    c[3, 0] = 50000                                       # an outlier
    got = oracle.cooksDistance(c, nf, mu, H, x)
    ck, mx, alpha = np_cooks(c, nf, mu, H, x)
    np.testing.assert_allclose(got["robustDisp"], alpha, rtol=1e-12)
    np.testing.assert_allclose(got["cooks"], ck, rtol=1e-12)
    np.testing.assert_allclose(got["maxCooks"], mx, rtol=1e-12, equal_nan=True)
    if design == "unrep":
        assert np.isnan(got["maxCooks"]).all()            # recordMaxCooks: no cell with 3 samples -> NA


def test_oracle_replace_vs_numpy(oracle):
    m = 30
    x = simulate.design_two_group(m)
    c, nf, mu, H = _inputs(200, x, seed=9)
    ck = np.random.default_rng(1).gamma(0.5, 1.0, c.shape)
    replaceable = np.arange(m) < 20
    got = oracle.replaceOutliers(c, nf, ck, 2.5, replaceable, trim=0.2)
    tbm = np.array([r_trim_mean(r, 0.2) for r in c / nf])
    rep = (tbm[:, None] * nf).astype(np.int64)             # as.integer truncates
    exp = np.where((ck > 2.5) & replaceable[None, :], rep, c)
    np.testing.assert_array_equal(got["counts"], exp)
    np.testing.assert_array_equal(got["replace"], (ck > 2.5).any(axis=1))


def _example(n, m, seed, disp=None, intercept=None):
    x = simulate.design_two_group(m)
    rng = np.random.default_rng(seed)
    b0 = rng.normal(4, 2, n) if intercept is None else np.asarray(intercept, float)
    alpha = 4.0 / 2.0 ** b0 + 0.5 if disp is None else np.full(n, disp)
    mu = np.broadcast_to(2.0 ** b0[:, None], (n, m))
    size = 1.0 / alpha[:, None]
    counts = rng.negative_binomial(np.broadcast_to(size, mu.shape), size / (size + mu)).astype(np.int32)
    return counts, x


def test_outlier_replacement_like_reference(oracle):
    """tests/testthat/test_outlier.R:2-33 (rows that are all zero are dropped up front: the engine fits
    objectNZ; results()'s p-value filtering is outside the path, so the Cook's flags are checked)"""
    counts, x = _example(100, 12, seed=1)
    counts[counts.sum(axis=1) == 0, 0] = 1
    counts[1] = [100000] + [10] * 11
    counts[2] = [100000] + [0] * 11
    E = HostEngine(oracle)
    dds0 = core.DESeq(core.DESeqDataSet(counts, x, engine=E), minReplicatesForReplace=np.inf)
    dds1 = core.DESeq(core.DESeqDataSet(counts, x, engine=E), minReplicatesForReplace=6)
    cutoff = fdist.ppf(.99, 2, 10)
    assert (dds0.mcols["maxCooks"][1:3] > cutoff).all() and core.cooksOutlier(dds0)[1:3].all()   # filtered
    assert np.isnan(dds1.mcols["maxCooks"]).all()                       # all samples replaceable -> NA (:2538)
    np.testing.assert_array_equal(dds1.counts_host, counts)            # counts still the same
    assert dds1.mcols["replace"][1] and dds1.mcols["replace"][2]
    lfc0, lfc1 = dds0.mcols["beta"][:, 1], dds1.mcols["beta"][:, 1]
    assert abs(lfc1[1]) < abs(lfc0[1])                                  # replaced, reduced LFC
    assert lfc1[2] == 0 or np.isnan(lfc1[2])                            # replaced: now all zero -> NA results
    keep = ~dds1.mcols["replace"]
    for k in ("WaldPvalue", "beta", "dispersion"):
        np.testing.assert_array_equal(dds1.mcols[k][keep], dds0.mcols[k][keep])   # untouched rows are equal
    rc = oracle.replaceOutliers(counts, np.ones(counts.shape), dds0.assays["cooks"], cutoff, np.ones(12, bool))
    np.testing.assert_array_equal(np.asarray(dds1.assays["replaceCounts"]), rc["counts"])


@pytest.mark.parametrize("disp0", [.01, .1])
@pytest.mark.parametrize("m", [10, 20, 80])
def test_cooks_catches_outliers_across_mu(oracle, disp0, m):
    """tests/testthat/test_outlier.R:35-56"""
    beta0 = np.linspace(1, 16, 100)
    idx = np.tile(np.r_[True, np.zeros(9, bool)], 10)
    counts, x = _example(100, m, seed=int(100 * disp0) + m, disp=disp0, intercept=beta0)
    counts[counts.sum(axis=1) == 0, 1] = 1
    counts[idx, 0] = (1000 * 2 ** beta0[idx]).astype(np.int64).clip(max=2 ** 31 - 1)
    dds = core.DESeq(core.DESeqDataSet(counts, x, engine=HostEngine(oracle)), fitType="mean",
                     minReplicatesForReplace=np.inf)
    cutoff = fdist.ppf(.99, 2, m - 2)
    assert (np.asarray(dds.assays["cooks"])[idx, 0] > cutoff).all()
    assert (dds.mcols["maxCooks"][~idx] < cutoff).all()
    np.testing.assert_array_equal(core.cooksOutlier(dds), idx)          # res$pvalue NA exactly there


def test_replace_errors_and_lrt(oracle):
    """tests/testthat/test_outlier.R:58-66"""
    counts, x = _example(100, 12, seed=5)
    counts[counts.sum(axis=1) == 0, 0] = 1
    counts[0, 0] = 1000000
    E = HostEngine(oracle)
    dds = core.DESeq(core.DESeqDataSet(counts, x, engine=E), test="LRT", reduced=np.ones((12, 1)),
                     minReplicatesForReplace=6)
    assert dds.mcols["replace"][0] and np.isfinite(dds.mcols["LRTPvalue"][0])
    c4, x4 = _example(100, 6, seed=6)   # (m = 4 in the reference test; residual df <= 3 is not mirrored)
    c4[c4.sum(axis=1) == 0, 0] = 1
    d4 = core.DESeqDataSet(c4, x4, engine=E)
    with pytest.raises(RuntimeError, match="first run DESeq"):
        core.replaceOutliers(d4)
    core.DESeq(d4)
    with pytest.raises(ValueError, match="at least 3 replicates"):
        core.replaceOutliers(d4, minReplicates=2)


def test_small_counts_not_flagged(oracle):
    """tests/testthat/test_outlier.R:74-86"""
    counts, x = _example(100, 8, seed=7, disp=0.01)
    counts[counts.sum(axis=1) == 0, 0] = 1
    counts[0] = [0, 0, 0, 100, 2100, 2200, 2300, 2400]
    counts[1:3, 0] = 100000
    dds = core.DESeq(core.DESeqDataSet(counts, x, engine=HostEngine(oracle)), fitType="mean")
    flt = core.cooksOutlier(dds)                        # results()'s filter incl. the low-count heuristic
    assert not flt[0]
    assert flt[1:3].all()
