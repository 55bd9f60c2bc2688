"""-m gpu: the count-outlier kernels (csrc/outlier.hip) against the oracle, bit for bit, through the
host-pointer C ABI and the device-pointer one; then the DESeq() chain with planted outliers
(Cook's distances -> replaceOutliers -> refit of the replaced rows) on both engines."""
import numpy as np
import pytest

from deseq2_amd import core, native, simulate
from deseq2_amd.engine import DeviceEngine, HostEngine
from tests.helpers import assert_same

pytestmark = pytest.mark.gpu


def _design(kind, m):
    if kind == "two":
        return simulate.design_two_group(m)
    if kind == "bc":
        return simulate.design_batch_condition(m)
    if kind == "unrep":
        return np.column_stack([np.ones(m), np.arange(m, dtype=float) / m])
    return simulate.design_factor(m, int(kind[1:]))


def _inputs(n, x, seed, nf_matrix=True):
    d = simulate.make_counts(n, x, seed=seed)
    c = d["counts"]
    rng = np.random.default_rng(seed)
    sf = np.exp(rng.normal(0, 0.2, x.shape[0]))
    nf = np.broadcast_to(sf[None, :], c.shape).copy()
    if nf_matrix:
        nf *= np.exp(rng.normal(0, 0.05, c.shape))
    mu = np.maximum(c.mean(axis=1, keepdims=True) * nf * np.exp(rng.normal(0, 0.1, c.shape)), 0.5)
    H = rng.uniform(0.001, 0.7, c.shape)
    c[::17, 0] = 40000
    c[5::23, -1] += 9000
    return c, nf, mu, H


CASES = [(300, 6, "two"), (300, 12, "bc"), (257, 24, "bc"), (200, 70, "two"), (100, 5, "unrep"), (150, 130, "two"),
         (120, 500, "bc"), (64, 1000, "two"), (40, 2000, "f10"), (90, 37, "unrep"), (1, 9, "two"),
         # cell sizes on every branch of the sort: registers with 1 ... 32 values per lane (cells of <= 64 ... 2048
         # samples), the LDS network above that
         (40, 600, "two"), (30, 2400, "two"), (50, 128, "two"), (50, 258, "two"), (20, 4200, "two"), (30, 1300, "two")]


@pytest.mark.parametrize("n,m,kind", CASES)
def test_cooks_distance_host_abi_vs_oracle(oracle, n, m, kind):
    x = _design(kind, m)
    c, nf, mu, H = _inputs(n, x, seed=m + n)
    a = native.cooksDistance(c, nf, mu, H, x)
    b = oracle.cooksDistance(c, nf, mu, H, x)
    for k in ("robustDisp", "cooks", "maxCooks"):
        assert_same(a[k], b[k], "cooksDistance$" + k)


@pytest.mark.parametrize("n,m,kind", [(300, 12, "bc"), (120, 500, "bc"), (64, 1000, "two"), (50, 37, "unrep")])
def test_cooks_distance_device_abi_vs_oracle(oracle, n, m, kind):
    E = DeviceEngine("cuda:0")
    x = _design(kind, m)
    c, nf, mu, H = _inputs(n, x, seed=2 * m + n)
    r = E.cooks_distance(E.counts(c), E.matrix(nf), E.matrix(mu), E.matrix(H), x)
    b = oracle.cooksDistance(c, nf, mu, H, x)
    assert_same(E.to_numpy(r["cooks"]), b["cooks"], "cooks (device)")
    assert_same(r["maxCooks"], b["maxCooks"], "maxCooks (device)")
    assert_same(r["robustDisp"], b["robustDisp"], "robustDisp (device)")


@pytest.mark.parametrize("n,m,trim", [(300, 12, .2), (200, 30, .2), (100, 500, .2), (33, 2000, .2), (50, 7, 0.0),
                                      (50, 64, .49)])
def test_replace_outliers_vs_oracle(oracle, n, m, trim):
    x = simulate.design_two_group(m)
    c, nf, _, _ = _inputs(n, x, seed=3 * m + n)
    ck = np.random.default_rng(m).gamma(0.3, 2.0, c.shape)
    replaceable = np.random.default_rng(n).uniform(size=m) < 0.7
    b = oracle.replaceOutliers(c, nf, ck, 2.0, replaceable, trim)
    a = native.replaceOutliers(c, nf, ck, 2.0, replaceable, trim)
    assert_same(a["counts"], b["counts"], "replaceOutliers$counts")
    assert_same(a["replace"], b["replace"], "replaceOutliers$replace")
    assert (a["counts"] != c).any()
    E = DeviceEngine("cuda:0")
    r = E.replace_outliers(E.counts(c), E.matrix(nf), E.matrix(ck), 2.0, replaceable, trim)
    assert_same(r["counts"].view().cpu().numpy(), b["counts"], "replaceOutliers$counts (device)")
    assert_same(r["replace"], b["replace"], "replaceOutliers$replace (device)")


def test_cooks_bad_arguments():
    x = simulate.design_two_group(8)
    c, nf, mu, H = _inputs(10, x, seed=1)
    from deseq2_amd import _lib
    with pytest.raises(_lib.DsqError):
        native.replaceOutliers(c, nf, mu, 1.0, np.ones(8, bool), trim=0.5)


COLS = ["dispGeneEst", "dispGeneIter", "dispFit", "dispMAP", "dispIter", "dispOutlier", "dispersion", "beta",
        "betaSE", "WaldStatistic", "betaConv", "betaIter", "deviance", "maxCooks", "replace", "baseMean", "baseVar"]


@pytest.mark.parametrize("engine", ["host", "device"])
def test_chain_with_outliers_identical_to_oracle(oracle, engine):
    """~batch + condition, 10 samples per cell of the 6 cells: planted outliers are found by Cook's distance,
    replaced by the trimmed mean and the rows refitted; every column equals the oracle chain's."""
    m = 60
    x = simulate.design_batch_condition(m)
    d = simulate.make_counts(500, x, seed=31)
    c = d["counts"]
    c[::25, 3] = 200000
    c[7::40, 41] += 60000
    E = HostEngine() if engine == "host" else DeviceEngine("cuda:0")
    a = core.DESeq(core.DESeqDataSet(c, x, sizeFactors=d["size_factors"], engine=E))
    b = core.DESeq(core.DESeqDataSet(c, x, sizeFactors=d["size_factors"], engine=HostEngine(oracle)))
    assert b.mcols["replace"].sum() >= 20
    for k in COLS:
        assert_same(a.mcols[k], b.mcols[k], "DESeq(outliers)$" + k)
    assert_same(E.to_numpy(a.assays["cooks"]), b.assays["cooks"], "assays cooks")
    ra = a.assays["replaceCounts"]
    assert_same(ra.view().cpu().numpy() if engine == "device" else ra, b.assays["replaceCounts"], "replaceCounts")


def test_chain_lrt_with_outliers_identical_to_oracle(oracle):
    m = 42
    x = simulate.design_batch_condition(m)
    d = simulate.make_counts(300, x, seed=32)
    c = d["counts"]
    c[::30, 0] = 150000
    red = x[:, :3]
    a = core.DESeq(core.DESeqDataSet(c, x, engine=DeviceEngine("cuda:0")), test="LRT", reduced=red)
    b = core.DESeq(core.DESeqDataSet(c, x, engine=HostEngine(oracle)), test="LRT", reduced=red)
    assert b.mcols["replace"].sum() >= 5
    for k in ("dispersion", "beta", "betaSE", "LRTStatistic", "LRTPvalue", "fullBetaConv", "betaIter", "deviance",
              "maxCooks", "replace"):
        assert_same(a.mcols[k], b.mcols[k], "DESeq(LRT, outliers)$" + k)


@pytest.mark.parametrize("n,m,kind", [(300, 6, "two"), (200, 24, "bc"), (100, 500, "bc"), (50, 130, "f10"),
                                      (61, 900, "f10"), (130, 2000, "f10"), (33, 2100, "bc")])      # long rows: tiles shared through LDS
def test_linear_mu_vs_oracle(oracle, n, m, kind):
    """linearModelMuNormalized (R/core.R:2454-2471) kernel, host ABI and device ABI, bit for bit"""
    x = _design(kind, m)
    c, nf, _, _ = _inputs(n, x, seed=5 * m + n)
    b = oracle.linearMu(c, nf, x, mu_floor=0.5)
    assert_same(native.linearMu(c, nf, x, mu_floor=0.5), b, "linearMu (host ABI)")
    E = DeviceEngine("cuda:0")
    mu = E.clamp_min(E.linear_mu(E.counts(c), E.matrix(nf), E.design(x)), 0.5)
    assert_same(E.to_numpy(mu), b, "linearMu (device ABI)")
