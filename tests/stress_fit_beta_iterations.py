"""Stress check (run by hand, not collected by pytest): fitBeta iteration counts of the restatement (cell path, closed
deviance split) against the LAPACK restatement (oracle/lapack_oracle.py) on many genes, including the
corners where the deviance split could lose accuracy -- dispersions at the 1e-8 floor, counts of 1e5, size factors
far from 1.  Prints the number of genes whose iteration count differs.

    python tests/stress_fit_beta_iterations.py [genes_per_case]
"""
import sys
import os
import multiprocessing as mp

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.helpers import make_case           # noqa: E402


def one(job):
    n, m, design, seed, imean, alpha_mode, sfr = job
    from oracle import oracle, lapack_oracle as reference
    cont = isinstance(design, tuple) and design[0] == "continuous"
    d = make_case(n, m, design[1] if cont else design, seed=seed, sf_random=sfr, intercept_mean=imean)
    if cont:
        # a continuous covariate: one design cell per sample, i.e. the general (per-sample) path of the restatement
        from tests.helpers import beta_init_qr
        rng0 = np.random.default_rng(seed + 7)
        d["x"] = np.column_stack([d["x"], rng0.normal(size=m)])
        with np.errstate(all="ignore"):
            d["beta_init"] = beta_init_qr(d["counts"].astype(float), d["nf"], d["x"])
    y = d["counts"].astype(float)
    nn = y.shape[0]
    rng = np.random.default_rng(seed)
    if alpha_mode == "rough":
        alpha = np.clip(d["alpha_init"], 1e-8, 10.0)
    elif alpha_mode == "floor":
        alpha = np.full(nn, 1e-8)
    else:
        alpha = 10 ** rng.uniform(-8, 1, nn)
    x = d["x"]
    p = x.shape[1]
    lam = np.full(p, 1e-6)
    contrast = np.zeros(p); contrast[-1] = 1.0
    args = (y, x, d["nf"], alpha, contrast, d["beta_init"], lam, d["weights"], False, 1e-8, 100, True, 0.5)
    a = oracle.fitBeta(*args)
    b = reference.fitBeta(*args)
    ia, ib = np.asarray(a["iter"]).ravel(), np.asarray(b["iter"]).ravel()
    ok = np.isfinite(np.asarray(b["deviance"]).ravel())
    rel = np.abs(np.asarray(a["deviance"]).ravel() - np.asarray(b["deviance"]).ravel())[ok] / (np.abs(np.asarray(b["deviance"]).ravel()[ok]) + 0.1)
    bm = np.abs(np.asarray(a["beta_mat"]) - np.asarray(b["beta_mat"])).max()
    return job, nn, int((ia != ib).sum()), float(rel.max() if rel.size else 0.0), float(bm), int((ib >= 100).sum())


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    jobs = []
    seed = 100
    designs = ((100, "batch_condition"), (500, "batch_condition"), (60, ("factor", 5)), (200, ("factor", 10)), (12, "two_group"))
    if os.environ.get("STRESS_CONTINUOUS"):
        designs = ((100, ("continuous", "batch_condition")), (60, ("continuous", "two_group")), (300, ("continuous", ("factor", 5))))
    for m, design in designs:
        for imean in (4.0, 9.0, 14.0):
            for alpha_mode in ("rough", "floor", "wide"):
                for sfr in (False, True):
                    seed += 1
                    jobs.append((n if m <= 200 else n // 4, m, design, seed, imean, alpha_mode, sfr))
    tot = bad = 0
    with mp.get_context("fork").Pool(8) as pool:
        for job, nn, nbad, rel, bm, nmax in pool.imap_unordered(one, jobs):
            tot += nn; bad += nbad
            print(job, "genes", nn, "iter mismatches", nbad, "max rel dev diff %.2e" % rel, "max |dbeta| %.2e" % bm, "at maxit", nmax, flush=True)
    print("TOTAL genes %d, iteration mismatches %d" % (tot, bad))


if __name__ == "__main__":
    main()
