"""Synthetic count matrices for tests and bench.py.

Generalises the reference's makeExampleDESeqDataSet (R/core.R:459-471) to designs with
more than two columns, as SURVEY.md section 8(d) specifies: log2 intercept ~ N(4, 2^2),
condition effects ~ N(0, 1), batch effects ~ N(0, 0.5^2), dispersion 4/2^intercept + 0.1,
counts ~ NB(mu = s_j 2^(x_j beta), size = 1/alpha); all-zero rows dropped.
"""
import numpy as np


def design_two_group(m):
    """~condition, two levels split m/2 : m-m/2 (config C1/C2)"""
    cond = (np.arange(m) >= m // 2).astype(np.float64)
    return np.column_stack([np.ones(m), cond])


def design_batch_condition(m, n_batch=3):
    """~batch + condition: batch cyclic with n_batch levels, condition 50/50 (config C3)"""
    batch = np.arange(m) % n_batch
    cond = (np.arange(m) >= m // 2).astype(np.float64)
    cols = [np.ones(m)] + [(batch == b).astype(np.float64) for b in range(1, n_batch)] + [cond]
    return np.column_stack(cols)


def design_factor(m, levels):
    """~group with `levels` levels of equal size (config C4)"""
    grp = (np.arange(m) * levels) // m
    cols = [np.ones(m)] + [(grp == g).astype(np.float64) for g in range(1, levels)]
    return np.column_stack(cols)


def make_counts(n, x, seed=1, intercept_mean=4.0, intercept_sd=2.0, size_factors=None,
                beta_sd=None, drop_all_zero=True):
    """Returns dict(counts int32 (n', m), size_factors (m,), beta (n', p) log2, alpha (n',))"""
    rng = np.random.Generator(np.random.PCG64(seed))
    m, p = x.shape
    if beta_sd is None:
        # last column = condition (sd 1), middle columns = batch-like (sd 0.5)
        beta_sd = np.array([0.5] * (p - 2) + [1.0]) if p >= 2 else np.array([])
    beta = np.empty((n, p))
    beta[:, 0] = rng.normal(intercept_mean, intercept_sd, n)
    for c in range(1, p):
        beta[:, c] = rng.normal(0.0, beta_sd[c - 1], n)
    alpha = 4.0 / 2.0 ** beta[:, 0] + 0.1
    sf = np.ones(m) if size_factors is None else np.asarray(size_factors, float)
    mu = sf[None, :] * 2.0 ** (beta @ x.T)
    size = 1.0 / alpha
    prob = size[:, None] / (size[:, None] + mu)
    counts = rng.negative_binomial(np.broadcast_to(size[:, None], mu.shape), prob).astype(np.int32)
    if drop_all_zero:
        keep = counts.sum(axis=1) > 0
        counts, beta, alpha = counts[keep], beta[keep], alpha[keep]
    return {"counts": counts, "size_factors": sf, "beta": beta, "alpha": alpha, "x": x}


def make_counts_trend_fails(n, x, seed=1):
    """counts whose dispersion GROWS with the mean: disp ~ asymptDisp + extraPois / mean cannot capture it and
    parametricDispersionFit stops with "parametric dispersion fit failed" (R/core.R:2177-2178) -- the case the reference
    answers with fitType = "local" (:885-893)"""
    rng = np.random.default_rng(seed)
    mean = np.exp(rng.uniform(np.log(20), np.log(3000), n))
    alpha = 0.01 + 0.0004 * mean
    lfc = np.column_stack([np.zeros(n)] + [rng.normal(0, .5, n) for _ in range(x.shape[1] - 1)])
    mu = mean[:, None] * 2.0 ** (lfc @ x.T)
    size = 1 / alpha
    return rng.negative_binomial(np.broadcast_to(size[:, None], mu.shape), size[:, None] / (size[:, None] + mu)).astype(np.int32)
