"""Shared input builders for the parity tests (seeded, numpy PCG64)."""
import numpy as np

from deseq2_amd import simulate


def beta_init_qr(counts, nf, x):
    """initial betas of R/fitNbinomGLMs.R:139-145: QR least squares on log(K/s + 0.1)"""
    q, r = np.linalg.qr(x)
    ylog = np.log(counts / nf + 0.1).T
    return np.linalg.solve(r, q.T @ ylog).T.copy()


def rough_alpha(counts, nf, x, min_disp=1e-8):
    """R/core.R:713-728 rough + moments dispersion start (enough for test inputs)"""
    m, p = x.shape
    yn = counts / nf
    q, r = np.linalg.qr(x)
    mu = np.maximum((yn @ q) @ (x @ np.linalg.inv(r)).T, 1.0)
    est = (((yn - mu) ** 2 - mu) / mu ** 2).sum(axis=1) / (m - p)
    rough = np.maximum(est, 0.0)
    bm = yn.mean(axis=1)
    bv = yn.var(axis=1, ddof=1)
    xim = np.mean(1.0 / nf.mean(axis=0))
    mom = (bv - xim * bm) / bm ** 2
    a = np.minimum(rough, mom)
    return np.minimum(np.maximum(min_disp, a), max(10.0, m))


def make_case(n, m, design, seed=1, weights=False, sf_random=False, **kw):
    if design == "two_group":
        x = simulate.design_two_group(m)
    elif design == "batch_condition":
        x = simulate.design_batch_condition(m)
    elif isinstance(design, tuple) and design[0] == "factor":
        x = simulate.design_factor(m, design[1])
    elif isinstance(design, tuple) and design[0] == "factor_cont":
        # a factor and ONE continuous covariate: one design cell per sample -> the general (per-sample) paths
        x = np.column_stack([simulate.design_factor(m, design[1]),
                             np.random.Generator(np.random.PCG64(seed + 77)).normal(0.0, 0.5, m)])
    else:
        raise ValueError(design)
    rng = np.random.Generator(np.random.PCG64(seed + 1000))
    sf = np.exp(rng.normal(0, 0.25, m)) if sf_random else None
    d = simulate.make_counts(n, x, seed=seed, size_factors=sf, **kw)
    if not kw.get("drop_all_zero", True):
        d["counts"][::37] = 0                      # some all-zero genes (allZero flag)
    counts = d["counts"]
    nn = counts.shape[0]
    nf = np.broadcast_to(d["size_factors"][None, :], (nn, m)).copy()
    w = np.ones((nn, m))
    if weights:
        w = rng.uniform(0.05, 1.0, (nn, m))
        w[rng.uniform(size=(nn, m)) < 0.02] = 0.0
        w = w / w.max(axis=1, keepdims=True)      # R/core.R:2702
    d.update(nf=nf, weights=w, x=x)
    with np.errstate(all="ignore"):
        d["beta_init"] = beta_init_qr(counts.astype(float), nf, x)
        d["alpha_init"] = rough_alpha(counts.astype(float), nf, x)
    return d


def assert_same(a, b, what, exact=True, rtol=1e-6):
    """bit-for-bit (NaN == NaN, -0 == +0) when exact, else north_star's 1e-6 relative"""
    a = np.asarray(a); b = np.asarray(b)
    assert a.shape == b.shape, "%s: shape %s vs %s" % (what, a.shape, b.shape)
    if exact:
        ok = (a == b) | (np.isnan(a.astype(float)) & np.isnan(b.astype(float)))
        if not ok.all():
            bad = np.argwhere(~ok)
            i = tuple(bad[0])
            raise AssertionError("%s: %d of %d entries differ; first at %s: %r vs %r" %
                                 (what, bad.shape[0], a.size, i, a[i], b[i]))
    else:
        np.testing.assert_allclose(a, b, rtol=rtol, atol=0, equal_nan=True, err_msg=what)
