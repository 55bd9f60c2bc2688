"""-m gpu: edge cases and a seeded sweep over shapes for the three native routines (HIP vs oracle, every
output identical): empty and one-gene inputs, the smallest designs, all-zero and huge counts, the
dispersion clamps, rows whose weights are all zero, every compiled design width."""
import numpy as np
import pytest

from deseq2_amd import native, simulate
from tests.helpers import assert_same, beta_init_qr, rough_alpha

pytestmark = pytest.mark.gpu

BETA_KEYS = ["iter", "beta_mat", "beta_var_mat", "deviance", "contrast_num", "contrast_denom", "hat_diagonals"]
DISP_KEYS = ["iter", "iter_accept", "log_alpha", "last_change", "initial_lp", "initial_dlp", "last_lp", "last_dlp",
             "last_d2lp"]


def _both(oracle, y, x, nf, alpha, w, useW, useQR=True, lam=1e-6, prior=False):
    n, m = y.shape
    p = x.shape[1]
    with np.errstate(all="ignore"):
        b0 = beta_init_qr(y.astype(float), nf, x) if np.linalg.matrix_rank(x) == p else np.zeros((n, p))
    b0 = np.nan_to_num(b0)
    lamv = np.full(p, lam) / np.log(2) ** 2
    bargs = (y, x, nf, alpha, np.r_[1.0, np.zeros(p - 1)], b0, lamv, w, useW, 1e-8, 100, useQR, 0.5)
    gb, ob = native.fitBeta(*bargs), oracle.fitBeta(*bargs)
    for k in BETA_KEYS:
        assert_same(gb[k], ob[k], "fitBeta$" + k)
    mu = oracle.fittedMu(x, nf, ob["beta_mat"], 0.5) if n else np.zeros((0, m))
    mu = np.where(np.isfinite(mu), mu, 0.5)
    la = np.log(alpha)
    dargs = (y, x, mu, la, la - 0.1, 0.8, np.log(1e-9), 1.0, 1e-6, 100, prior, np.maximum(w, 1e-6) if useW else w, useW,
             1e-2, True)
    gd, od = native.fitDisp(*dargs), oracle.fitDisp(*dargs)
    for k in DISP_KEYS:
        assert_same(gd[k], od[k], "fitDisp$" + k)
    grid = np.linspace(np.log(1e-8), np.log(max(10, m)), 12)
    gargs = (y, x, mu, grid, la, 1.0, prior, dargs[11], useW, 1e-2, True)
    assert_same(native.fitDispGrid(*gargs)["log_alpha"], oracle.fitDispGrid(*gargs)["log_alpha"], "fitDispGrid")
    return gb, gd


def test_empty_and_single_gene(oracle):
    x = simulate.design_two_group(8)
    for n in (0, 1):
        y = np.full((n, 8), 7, dtype=np.int32)
        _both(oracle, y, x, np.ones((n, 8)), np.full(n, 0.1), np.ones((n, 8)), False)


@pytest.mark.parametrize("m,p", [(2, 1), (3, 1), (3, 2), (4, 3), (11, 10), (64, 1), (65, 2), (128, 4), (129, 5)])
def test_smallest_and_boundary_shapes(oracle, m, p):
    rng = np.random.default_rng(m * 16 + p)
    x = np.column_stack([np.ones(m)] + [rng.normal(size=m) for _ in range(p - 1)]) if p > 1 else np.ones((m, 1))
    y = rng.negative_binomial(2.0, 0.02, size=(37, m)).astype(np.int32)
    _both(oracle, y, x, np.exp(rng.normal(0, 0.2, (37, m))), rng.uniform(0.01, 2.0, 37), np.ones((37, m)), False)


def test_zero_rows_huge_counts_and_clamps(oracle):
    m = 12
    x = simulate.design_two_group(m)
    rng = np.random.default_rng(5)
    y = rng.negative_binomial(1.5, 0.01, size=(40, m)).astype(np.int32)
    y[0] = 0                                           # an all-zero gene: mu pinned at minmu
    y[1] = 0; y[1, 0] = 1                              # a single count
    y[2] = 2 ** 31 - 1                                 # the largest INTSXP count
    y[3] = [2 ** 31 - 1] + [0] * (m - 1)
    y[4, :6] = 0                                       # a whole group at zero: beta runs away, |beta| > 30 abort
    alpha = rng.uniform(0.05, 1.0, 40)
    alpha[5], alpha[6], alpha[7] = 1e-8, 10.0, float(m)   # minDisp clamp, maxDisp clamps of R/core.R:727-728
    _both(oracle, y, x, np.ones((40, m)), alpha, np.ones((40, m)), False)
    _both(oracle, y, x, np.ones((40, m)), alpha, np.ones((40, m)), False, useQR=False, prior=True)


def test_weight_edge_cases(oracle):
    m = 16
    x = simulate.design_batch_condition(m)
    rng = np.random.default_rng(6)
    y = rng.negative_binomial(2.0, 0.05, size=(30, m)).astype(np.int32)
    w = rng.uniform(0.0, 1.0, (30, m))
    w[0] = 0.0                                         # every observation of a gene weighted out
    w[1, 1:] = 0.0                                     # one observation left
    w[2, x[:, 1] == 1] = 0.0                           # a design column loses all its samples (:42 drops it)
    w[3] = 1.0
    _both(oracle, y, x, np.ones((30, m)), rng.uniform(0.05, 1.0, 30), w, True)
    _both(oracle, y, x, np.ones((30, m)), rng.uniform(0.05, 1.0, 30), w, True, useQR=False, lam=0.5, prior=True)


@pytest.mark.parametrize("seed", range(24))
def test_seeded_shape_sweep(oracle, seed):
    rng = np.random.default_rng(1000 + seed)
    p = int(rng.integers(1, 11))
    m = int(rng.integers(p + 1, 260))
    n = int(rng.integers(1, 70))
    cols = [np.ones(m)]
    for c in range(p - 1):
        cols.append(rng.normal(size=m) if rng.uniform() < 0.5 else (rng.uniform(size=m) < 0.4).astype(float))
    x = np.column_stack(cols)
    if np.linalg.matrix_rank(x) < p:
        x[:, 1:] += rng.normal(0, 0.1, (m, p - 1))
    mu = np.exp(rng.normal(3, 1.5, (n, 1))) * np.exp(rng.normal(0, 0.3, (n, m)))
    size = 1.0 / rng.uniform(0.02, 2.0, (n, 1))
    y = rng.negative_binomial(np.broadcast_to(size, mu.shape), size / (size + mu)).astype(np.int32)
    nf = np.exp(rng.normal(0, 0.25, (n, m)))
    useW = bool(rng.uniform() < 0.5)
    w = rng.uniform(0.05, 1.0, (n, m)) if useW else np.ones((n, m))
    if useW:
        w[rng.uniform(size=w.shape) < 0.03] = 0.0
    with np.errstate(all="ignore"):
        alpha = rough_alpha(y.astype(float), nf, x) if m > p else np.full(n, 0.1)
    alpha = np.nan_to_num(alpha, nan=0.1)
    _both(oracle, y, x, nf, alpha, w, useW, useQR=bool(rng.uniform() < 0.6), lam=float(10 ** rng.uniform(-6, 0)),
          prior=bool(rng.uniform() < 0.5))


def test_device_pointers_in_r_layout(oracle):
    """dsq_fit_*_dev also accepts DEVICE pointers in R's column-major layout (an R session holding external
    pointers): same answers as the host-pointer entry."""
    import ctypes as C
    import torch
    from deseq2_amd import _lib as L
    from tests.helpers import make_case
    d = make_case(150, 44, "batch_condition", seed=17, sf_random=True)
    n, m = d["counts"].shape
    p = d["x"].shape[1]
    lam = np.full(p, 1e-6) / np.log(2) ** 2
    contrast = np.r_[1.0, np.zeros(p - 1)]
    want = oracle.fitBeta(d["counts"], d["x"], d["nf"], d["alpha_init"], contrast, d["beta_init"], lam, d["weights"],
                          False, 1e-8, 100, True, 0.5)
    dev = torch.device("cuda:0")

    def col(a, dt=torch.float64):          # column-major n x m == contiguous (m, n) tensor
        return torch.as_tensor(np.ascontiguousarray(np.asarray(a).T), dtype=dt, device=dev)

    def ptr(t):
        return C.c_void_p(t.data_ptr())
    y, x, nf, b0 = col(d["counts"], torch.int32), col(d["x"]), col(d["nf"]), col(d["beta_init"])
    from deseq2_amd import native
    cells = native.cell_index(d["x"])       # the design cells (the device entry point cannot read them off a device x)
    al, ct, lm = (torch.as_tensor(v, dtype=torch.float64, device=dev) for v in (d["alpha_init"], contrast, lam))
    out = {k: torch.zeros(s, dtype=torch.float64, device=dev) for k, s in
           (("beta_mat", (p, n)), ("beta_var_mat", (p, n)), ("iter", (n,)), ("hat", (m, n)), ("cn", (n,)), ("cd", (n,)),
            ("dev", (n,)), ("mu", (m, n)))}
    a = L.DsqFitBetaArgs(n=n, m=m, p=p, layout=L.DSQ_LAYOUT_R, ld=0, y=ptr(y), y_type=L.DSQ_Y_INT32, x=ptr(x),
                         nf=ptr(nf), nf_is_vector=0, alpha_hat=ptr(al), contrast=ptr(ct), beta_mat=ptr(b0),
                         lambda_=ptr(lm), weights=None, useWeights=0, tol=1e-8, maxit=100, useQR=1, minmu=0.5,
                         cell_of=cells.ctypes.data_as(C.c_void_p), ncell=int(cells.max()) + 1)
    o = L.DsqFitBetaOut(beta_mat=ptr(out["beta_mat"]), beta_var_mat=ptr(out["beta_var_mat"]), iter=ptr(out["iter"]),
                        hat_diagonals=ptr(out["hat"]), contrast_num=ptr(out["cn"]), contrast_denom=ptr(out["cd"]),
                        deviance=ptr(out["dev"]), mu=ptr(out["mu"]), mu_floor=0.5)
    L.check(L.lib().dsq_fit_beta_dev(C.byref(a), C.byref(o), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    assert_same(out["beta_mat"].cpu().numpy().T, want["beta_mat"], "beta_mat (R layout, device pointers)")
    assert_same(out["iter"].cpu().numpy(), want["iter"], "iter")
    assert_same(out["hat"].cpu().numpy().T, want["hat_diagonals"], "hat_diagonals")
    assert_same(out["mu"].cpu().numpy().T, oracle.fittedMu(d["x"], d["nf"], want["beta_mat"], 0.5), "mu")
    assert_same(out["dev"].cpu().numpy(), want["deviance"], "deviance")


@pytest.mark.parametrize("shards", [2, 3, 7])
def test_host_entry_points_shard_genes_inside_the_library(oracle, shards, monkeypatch):
    """the host-pointer entry points (what the .Call shim binds) cut [0, n) into the contiguous ranges of
    R/parallel.R:10 and fit them on worker threads with their own streams -- one per visible device; DSQ_HOST_SHARDS
    forces the split on this one GPU.  Sharded == single call == oracle, every output (test_parallel.R:27-37)."""
    from deseq2_amd import native
    from tests.helpers import make_case
    d = make_case(203, 40, "batch_condition", seed=23, weights=True, sf_random=True)
    p = d["x"].shape[1]
    lam = np.full(p, 1e-6) / np.log(2) ** 2
    bargs = (d["counts"], d["x"], d["nf"], d["alpha_init"], np.r_[1.0, np.zeros(p - 1)], d["beta_init"], lam,
             d["weights"], True, 1e-8, 100, True, 0.5)
    one = native.fitBeta(*bargs, want_mu=True, mu_floor=0.5)
    mu = one["mu"]
    la0 = np.log(d["alpha_init"])
    dargs = (d["counts"], d["x"], mu, la0, la0 - 0.1, 0.9, np.log(1e-9), 1.0, 1e-6, 100, True,
             np.maximum(d["weights"], 1e-6), True, 1e-2, True)
    done = native.fitDisp(*dargs)
    grid = np.linspace(np.log(1e-8), np.log(40.0), 20)
    gargs = (d["counts"], d["x"], mu, grid, la0, 1.0, True, d["weights"], True, 1e-2, True)
    gone = native.fitDispGrid(*gargs)
    monkeypatch.setenv("DSQ_HOST_SHARDS", str(shards))
    many = native.fitBeta(*bargs, want_mu=True, mu_floor=0.5)
    dmany = native.fitDisp(*dargs)
    gmany = native.fitDispGrid(*gargs)
    monkeypatch.delenv("DSQ_HOST_SHARDS")
    for k in ("beta_mat", "beta_var_mat", "iter", "hat_diagonals", "contrast_num", "contrast_denom", "deviance", "mu"):
        assert_same(many[k], one[k], "sharded fitBeta$" + k)
    for k in ("log_alpha", "iter", "iter_accept", "last_change", "initial_lp", "initial_dlp", "last_lp", "last_dlp",
              "last_d2lp"):
        assert_same(dmany[k], done[k], "sharded fitDisp$" + k)
    assert_same(gmany["log_alpha"], gone["log_alpha"], "sharded fitDispGrid")
    want = oracle.fitBeta(*bargs)
    assert_same(many["beta_mat"], want["beta_mat"], "sharded fitBeta vs oracle")
    assert_same(dmany["iter"], oracle.fitDisp(*dargs)["iter"], "sharded fitDisp vs oracle")


def test_row_range_entry_point_equals_whole_call():
    """dsq_fit_beta_rows over consecutive ranges (what r_shim.c does between interrupt polls) == one call"""
    from deseq2_amd import native
    from tests.helpers import make_case
    d = make_case(150, 30, "two_group", seed=29, sf_random=True)
    p = d["x"].shape[1]
    args = (d["counts"], d["x"], d["nf"], d["alpha_init"], np.r_[1.0, np.zeros(p - 1)], d["beta_init"],
            np.full(p, 1e-6) / np.log(2) ** 2, d["weights"], False, 1e-8, 100, True, 0.5)
    n = d["counts"].shape[0]
    whole = native.fitBeta(*args, want_mu=True)
    parts = native.fitBeta(*args, want_mu=True, row_ranges=[(0, 64), (64, 1), (65, n - 65)])
    for k in ("beta_mat", "beta_var_mat", "iter", "hat_diagonals", "deviance", "mu"):
        assert_same(parts[k], whole[k], "rows$" + k)
    with pytest.raises(Exception):
        native.fitBeta(*args, row_ranges=[(n - 3, 5)])
