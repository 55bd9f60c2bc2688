"""GPU fuzz (run by hand on a GPU box, not collected by pytest): the three native routines, HIP library vs the CPU
checker, on random shapes / designs (factor designs with few cells, continuous covariates, mixed; p up to 40) /
weights / ridge / QR / prior settings.  Every output is compared bit for bit, as in tests/test_gpu_edge.py.

    python tests/gpu_fuzz.py [first_seed] [n_seeds]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402
from deseq2_amd import native  # noqa: E402
from oracle import oracle  # noqa: E402
from tests.helpers import assert_same, beta_init_qr, rough_alpha  # noqa: E402

BETA_KEYS = ("beta_mat", "beta_var_mat", "iter", "hat_diagonals", "contrast_num", "contrast_denom", "deviance")
DISP_KEYS = ("log_alpha", "iter", "iter_accept", "last_change", "initial_lp", "initial_dlp", "last_lp", "last_dlp", "last_d2lp")


def design(rng, m):
    kind = rng.integers(3)
    if kind == 0:                                  # factor design(s): few cells
        levels = int(rng.integers(2, 25)) if rng.uniform() < 0.8 else int(rng.integers(25, 41))      # (round 5: the 32- / 48-column builds)
        levels = min(levels, max(2, m // 2))
        f = np.arange(m) % levels
        rng.shuffle(f)
        x = np.column_stack([np.ones(m)] + [(f == l).astype(float) for l in range(1, levels)])
        if rng.uniform() < 0.4 and x.shape[1] <= 46:
            g = (rng.uniform(size=m) < 0.5).astype(float)
            x = np.column_stack([x, g])
    else:
        p = int(rng.integers(1, (25 if rng.uniform() < 0.85 else 34) if kind == 1 else 11))
        p = min(p, m - 1)
        cols = [np.ones(m)]
        for c in range(p - 1):
            cols.append(rng.normal(size=m) if (kind == 1 or rng.uniform() < 0.5) else (rng.uniform(size=m) < 0.4).astype(float))
        x = np.column_stack(cols)
    if np.linalg.matrix_rank(x) < x.shape[1]:
        x[:, 1:] += rng.normal(0, 0.1, (m, x.shape[1] - 1))
    return x


WIDE_BASE = 1000000      # seeds from here on draw WIDE designs (round 6: 11 .. 64 columns; the seeds below keep their meaning)


def design_wide(rng):
    """11 .. 64 columns without (useful) cells: paired designs, factors of many levels, continuous covariates, mixtures"""
    kind = rng.integers(4)
    if kind == 0:                                  # ~ patient + treatment
        patients = int(rng.integers(10, 64))
        reps = int(rng.integers(1, 3))
        m = 2 * patients * reps
        pat = np.repeat(np.arange(patients), 2 * reps)
        trt = np.tile(np.repeat([0.0, 1.0], reps), patients)
        x = np.column_stack([np.ones(m)] + [(pat == k).astype(float) for k in range(1, patients)] + [trt])
    elif kind == 1:                                # a factor of many levels
        levels = int(rng.integers(11, 65))
        m = levels * int(rng.integers(2, 5))
        f = np.arange(m) % levels
        rng.shuffle(f)
        x = np.column_stack([np.ones(m)] + [(f == l).astype(float) for l in range(1, levels)])
    else:                                          # continuous covariates (kind 3: next to a small factor)
        p = int(rng.integers(11, 65))
        m = int(rng.integers(p + 4, max(p + 5, 260)))
        cols = [np.ones(m)]
        if kind == 3:
            f = np.arange(m) % 4
            rng.shuffle(f)
            cols += [(f == l).astype(float) for l in range(1, 4)]
        while len(cols) < p:
            cols.append(rng.normal(size=m))
        x = np.column_stack(cols)
    if np.linalg.matrix_rank(x) < x.shape[1]:
        x[:, 1:] += rng.normal(0, 0.1, (x.shape[0], x.shape[1] - 1))
    return x


def one(seed):
    rng = np.random.default_rng(50000 + seed)
    if seed >= WIDE_BASE:
        x = design_wide(rng)
        m = x.shape[0]
        n = int(rng.integers(1, 24))
    else:
        m = int(rng.integers(4, 300))
        n = int(rng.integers(1, 60))
        x = design(rng, m)
    p = x.shape[1]
    mu = np.exp(rng.normal(3, 1.5, (n, 1))) * np.exp(rng.normal(0, 0.3, (n, m)))
    size = 1.0 / rng.uniform(0.02, 2.0, (n, 1))
    y = rng.negative_binomial(np.broadcast_to(size, mu.shape), size / (size + mu)).astype(np.int32)
    nf = np.exp(rng.normal(0, 0.25, (n, m)))
    useW = bool(rng.uniform() < 0.4)
    w = rng.uniform(0.05, 1.0, (n, m)) if useW else np.ones((n, m))
    if useW:
        w[rng.uniform(size=w.shape) < 0.03] = 0.0
    with np.errstate(all="ignore"):
        alpha = rough_alpha(y.astype(float), nf, x) if m > p else np.full(n, 0.1)
        alpha = np.clip(np.nan_to_num(alpha, nan=0.1), 1e-8, 50.0)
        b0 = beta_init_qr(y.astype(float), nf, x) if np.linalg.matrix_rank(x) == p else np.zeros((n, p))
    b0 = np.nan_to_num(b0)
    useQR = bool(rng.uniform() < 0.6)
    lam = float(10 ** rng.uniform(-6, 0))
    prior = bool(rng.uniform() < 0.5)
    useCR = bool(rng.uniform() < 0.85)
    tag = "seed %d: n=%d m=%d p=%d cells=%d useW=%d useQR=%d prior=%d useCR=%d" % (
        seed, n, m, p, len(np.unique(x, axis=0)), useW, useQR, prior, useCR)
    lamv = np.full(p, lam) / np.log(2) ** 2
    bargs = (y, x, nf, alpha, np.r_[1.0, np.zeros(p - 1)], b0, lamv, w, useW, 1e-8, 100, useQR, 0.5)
    gb, ob = native.fitBeta(*bargs), oracle.fitBeta(*bargs)
    for k in BETA_KEYS:
        assert_same(gb[k], ob[k], tag + " fitBeta$" + k)
    mu_hat = oracle.fittedMu(x, nf, ob["beta_mat"], 0.5)
    mu_hat = np.where(np.isfinite(mu_hat), mu_hat, 0.5)
    la = np.log(alpha)
    dargs = (y, x, mu_hat, la, la - 0.1, 0.8, np.log(1e-9), 1.0, 1e-6, 100, prior, np.maximum(w, 1e-6) if useW else w, useW,
             1e-2, useCR)
    gd, od = native.fitDisp(*dargs), oracle.fitDisp(*dargs)
    for k in DISP_KEYS:
        assert_same(gd[k], od[k], tag + " fitDisp$" + k)
    grid = np.linspace(np.log(1e-8), np.log(max(10, m)), 12)
    gargs = (y, x, mu_hat, grid, la, 1.0, prior, dargs[11], useW, 1e-2, useCR)
    assert_same(native.fitDispGrid(*gargs)["log_alpha"], oracle.fitDispGrid(*gargs)["log_alpha"], tag + " fitDispGrid")
    return tag


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    bad = 0
    for s in range(first, first + count):
        try:
            one(s)
        except AssertionError as e:
            bad += 1
            print("FAIL", str(e)[:300], flush=True)
        except Exception as e:                       # noqa: BLE001
            bad += 1
            print("ERROR seed %d: %r" % (s, e), flush=True)
    print("fuzz: %d seeds, %d failures" % (count, bad))


if __name__ == "__main__":
    main()
