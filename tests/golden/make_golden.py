#!/usr/bin/env python
"""Generates tests/golden/nmath_golden.json: high-precision (mpmath, 50 digits) reference
values for every scalar primitive the oracle restates (exp, log, log1p, lgamma, digamma,
trigamma, stirlerr, bd0, NB log-density), on fixed inputs.  The reference implementation
(R's nmath) cannot run in this image, so these vectors pin the *mathematical* value; the
tests bound the oracle's error against them in ulps / absolute eps.
Run:  python tests/golden/make_golden.py      (deterministic; commit the json)"""
import json
import os

import mpmath as mp
import numpy as np

mp.mp.dps = 50
HERE = os.path.dirname(os.path.abspath(__file__))


def f(v):
    return float(v)


def main():
    rng = np.random.default_rng(20260925)
    out = {}
    xs = np.concatenate([rng.uniform(-700, 700, 300), rng.uniform(-1, 1, 200), [-745.0, 709.0, 0.0, 1e-300]])
    out["exp"] = {"x": xs.tolist(), "y": [f(mp.exp(mp.mpf(float(v)))) for v in xs]}
    xs = np.concatenate([np.exp(rng.uniform(-700, 700, 300)), rng.uniform(0.5, 2, 300), [1.0, 5e-324, 1e-310]])
    out["log"] = {"x": xs.tolist(), "y": [f(mp.log(mp.mpf(float(v)))) for v in xs]}
    xs = np.concatenate([rng.uniform(-0.999, 5, 300), np.exp(rng.uniform(-40, 40, 200)),
                         -np.exp(rng.uniform(-40, -0.001, 200))])
    out["log1p"] = {"x": xs.tolist(), "y": [f(mp.log1p(mp.mpf(float(v)))) for v in xs]}
    xs = np.concatenate([np.exp(rng.uniform(-12, 30, 500)), rng.uniform(0.5, 12, 300), np.arange(1, 41) / 2.0])
    out["lgamma"] = {"x": xs.tolist(), "y": [f(mp.loggamma(mp.mpf(float(v)))) for v in xs]}
    out["digamma"] = {"x": xs.tolist(), "y": [f(mp.digamma(mp.mpf(float(v)))) for v in xs]}
    out["trigamma"] = {"x": xs.tolist(), "y": [f(mp.polygamma(1, mp.mpf(float(v)))) for v in xs]}
    xs = np.concatenate([rng.uniform(0.01, 15, 200), rng.uniform(15, 2000, 200), np.arange(1, 31) / 2.0])
    st = lambda n: mp.loggamma(n + 1) - (n + mp.mpf(1) / 2) * mp.log(n) + n - mp.log(mp.sqrt(2 * mp.pi))
    out["stirlerr"] = {"x": xs.tolist(), "y": [f(st(mp.mpf(float(v)))) for v in xs]}
    a = np.exp(rng.uniform(-3, 9, 300)); b = a * np.exp(rng.normal(0, 0.3, 300))
    bd0 = lambda x, npp: x * mp.log(x / npp) + npp - x
    out["bd0"] = {"x": a.tolist(), "np": b.tolist(),
                  "y": [f(bd0(mp.mpf(float(u)), mp.mpf(float(v)))) for u, v in zip(a, b)]}
    # NB log pmf: lgamma(x+r) - lgamma(r) - lgamma(x+1) + r log(r/(r+mu)) + x log(mu/(r+mu))
    x = rng.integers(0, 3000, 400).astype(float); x[:80] = 0
    size = np.exp(rng.uniform(np.log(2e-3), np.log(1e8), 400))
    mu = np.exp(rng.uniform(np.log(1e-3), np.log(1e5), 400))

    def nb(x, r, mu):
        x, r, mu = mp.mpf(float(x)), mp.mpf(float(r)), mp.mpf(float(mu))
        return (mp.loggamma(x + r) - mp.loggamma(r) - mp.loggamma(x + 1) + r * mp.log(r / (r + mu))
                + x * mp.log(mu / (r + mu)))
    out["dnbinom_mu_log"] = {"x": x.tolist(), "size": size.tolist(), "mu": mu.tolist(),
                             "y": [f(nb(*t)) for t in zip(x, size, mu)]}
    # literal known answers held by the reference's own tests
    out["kat"] = {
        "test_results_R_9_43_50": {"counts": [100] * 4 + [200] * 4 + [800] * 4,
                                   "beta_log2": [float(np.log2(100)), 0.0, 1.0, 3.0]},
        "test_optim_R_30_39": {"counts": [0, 0, 0, 0, 0, 1000, 1000, 0, 0, 0], "betaIter": 100},
    }
    with open(os.path.join(HERE, "nmath_golden.json"), "w") as fh:
        json.dump(out, fh)
    print("wrote", os.path.join(HERE, "nmath_golden.json"))


if __name__ == "__main__":
    main()
