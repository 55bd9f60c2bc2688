import sys, numpy as np
sys.path.insert(0, '.')
import torch
from deseq2_amd import native
from oracle import oracle
from tests.helpers import beta_init_qr, rough_alpha
def run(seed):
    rng = np.random.default_rng(1000 + seed)
    p = int(rng.integers(1, 11)); m = int(rng.integers(p + 1, 260)); n = int(rng.integers(1, 70))
    cols = [np.ones(m)]
    for c in range(p - 1):
        cols.append(rng.normal(size=m) if rng.uniform() < 0.5 else (rng.uniform(size=m) < 0.4).astype(float))
    x = np.column_stack(cols)
    if np.linalg.matrix_rank(x) < p: x[:, 1:] += rng.normal(0, 0.1, (m, p - 1))
    mu = np.exp(rng.normal(3, 1.5, (n, 1))) * np.exp(rng.normal(0, 0.3, (n, m)))
    size = 1.0 / rng.uniform(0.02, 2.0, (n, 1))
    y = rng.negative_binomial(np.broadcast_to(size, mu.shape), size / (size + mu)).astype(np.int32)
    nf = np.exp(rng.normal(0, 0.25, (n, m)))
    useW = bool(rng.uniform() < 0.5)
    w = rng.uniform(0.05, 1.0, (n, m)) if useW else np.ones((n, m))
    if useW: w[rng.uniform(size=w.shape) < 0.03] = 0.0
    alpha = np.nan_to_num(rough_alpha(y.astype(float), nf, x), nan=0.1)
    useQR = bool(rng.uniform() < 0.6); lam = float(10 ** rng.uniform(-6, 0)); prior = bool(rng.uniform() < 0.5)
    print("p", p, "m", m, "n", n, "useW", useW, "useQR", useQR, "prior", prior)
    b0 = np.nan_to_num(beta_init_qr(y.astype(float), nf, x))
    lamv = np.full(p, lam) / np.log(2) ** 2
    ob = oracle.fitBeta(y, x, nf, alpha, np.r_[1.0, np.zeros(p - 1)], b0, lamv, w, useW, 1e-8, 100, useQR, 0.5)
    mu = oracle.fittedMu(x, nf, ob["beta_mat"], 0.5); mu = np.where(np.isfinite(mu), mu, 0.5)
    la = np.log(alpha)
    res=[]
    for useCR in (True, False):
        dargs = (y, x, mu, la, la - 0.1, 0.8, np.log(1e-9), 1.0, 1e-6, 100, prior, np.maximum(w, 1e-6) if useW else w, useW, 1e-2, useCR)
        gd, od = native.fitDisp(*dargs), oracle.fitDisp(*dargs)
        res.append((useCR, int((np.asarray(gd["iter"]) != np.asarray(od["iter"])).sum()), int((np.asarray(gd["initial_dlp"]) != np.asarray(od["initial_dlp"])).sum()), int((np.asarray(gd["last_d2lp"]) != np.asarray(od["last_d2lp"])).sum())))
    print("   ", res, flush=True)
        if res[-1][3]:
            a, b = np.asarray(gd["last_d2lp"]).ravel(), np.asarray(od["last_d2lp"]).ravel()
            bad = np.nonzero(~((a == b) | (np.isnan(a) & np.isnan(b))))[0]
            print("    d2lp", useCR, bad, a[bad], b[bad], np.asarray(gd["log_alpha"]).ravel()[bad], np.asarray(od["log_alpha"]).ravel()[bad], y[bad].sum(axis=1))

for sd in (10,):
    run(sd)
