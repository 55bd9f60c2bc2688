"""The product's .Call binding (deseq2_amd/csrc/r_shim.c) EXECUTED on a mock R runtime (tests/r_mock/, tests/rmock.py):
registration table, arities, argument coercions, list names / types / dims, error paths -- what src/RcppExports.cpp:16-94
is to the reference.  The CPU tests stop where the library needs a device (and check that it says so as an R error); the
-m gpu tests run the four entry points and compare with deseq2_amd/native.py (the same library through ctypes)."""
import numpy as np
import pytest

from tests import rmock
from tests.helpers import make_case


def _beta_args(d, y_kind="int", maxit=100.0, **over):
    p = d["x"].shape[1]
    lam = np.full(p, 1e-6) / np.log(2) ** 2
    a = dict(y=rmock.sexp(d["counts"], y_kind), x=rmock.sexp(d["x"], "real"), nf=rmock.sexp(d["nf"], "real"),
             alpha=rmock.sexp(d["alpha_init"], "real"), contrast=rmock.sexp(np.r_[1.0, np.zeros(p - 1)], "real"),
             beta=rmock.sexp(d["beta_init"], "real"), lam=rmock.sexp(lam, "real"), w=rmock.sexp(d["weights"], "real"),
             useW=rmock.sexp([False], "lgl"), tol=rmock.sexp([1e-8], "real"), maxit=rmock.sexp([maxit], "real"),
             useQR=rmock.sexp([True], "lgl"), minmu=rmock.sexp([0.5], "real"))
    a.update(over)
    return list(a.values()), lam


def test_registration_table_is_the_references():
    """src/RcppExports.cpp:84-94: three routines with 15 / 13 / 11 arguments, registered, dynamic lookup off"""
    L = rmock.lib()
    table = {L.rmock_routine_name(i).decode(): L.rmock_routine_arity(i) for i in range(L.rmock_n_routines())}
    assert table["_DESeq2_fitDisp"] == 15 and table["_DESeq2_fitBeta"] == 13 and table["_DESeq2_fitDispGrid"] == 11
    assert table["_DESeq2_mi355x_DESeq"] == 31
    assert L.rmock_dynamic_symbols() == 0                # R_useDynamicSymbols(dll, FALSE)


def test_wrong_argument_count_is_refused_by_the_call_table():
    rmock.lib().rmock_reset()
    with pytest.raises(rmock.RError, match="Incorrect number of arguments"):
        rmock.dotCall("_DESeq2_fitBeta", rmock.sexp([1.0]))
    with pytest.raises(rmock.RError, match="not available"):
        rmock.dotCall("_DESeq2_fitGamma")


def test_dimension_errors_are_r_errors_and_leave_the_runtime_clean():
    """a wrong dimension is an Rf_error (a longjmp out of the shim) BEFORE any pointer reaches the library: the message
    names the argument, nothing stays protected, transient memory is gone, and the next call works as if nothing happened"""
    L = rmock.lib()
    L.rmock_reset()
    d = make_case(8, 6, "two_group", seed=3)
    for key, bad, pat in (("nf", np.ones((8, 5)), "nfSEXP must be a 8 x 6 matrix"),
                          ("x", np.ones((5, 2)), "xSEXP must be a 6 x 2 matrix"),
                          ("alpha", np.ones(7), "alpha_hatSEXP must have length 8"),
                          ("lam", np.ones(3), "lambdaSEXP must have length 2"),
                          ("beta", np.zeros(16), "beta_matSEXP must be a 8 x 2 matrix")):      # a plain vector: no dim
        args, _ = _beta_args(d, **{key: rmock.sexp(bad, "real")})
        with pytest.raises(rmock.RError, match=pat):
            rmock.dotCall("_DESeq2_fitBeta", *args)
        assert L.rmock_protect_depth() == 0 and L.rmock_live_transients() == 0
    # counts of a type Rcpp could not coerce either
    args, _ = _beta_args(d)
    args[0] = L.rmock_new(rmock.STRSXP, 8, 6, None)
    with pytest.raises(rmock.RError, match="integer or numeric matrix"):
        rmock.dotCall("_DESeq2_fitBeta", *args)
    assert L.rmock_protect_depth() == 0
    with pytest.raises(rmock.RError, match="at least 2 grid points"):
        rmock.dotCall("_DESeq2_fitDispGrid", rmock.sexp(d["counts"]), rmock.sexp(d["x"], "real"), rmock.sexp(d["nf"], "real"),
                      rmock.sexp([0.5], "real"), rmock.sexp(np.zeros(8), "real"), rmock.sexp([1.0], "real"), rmock.sexp([False], "lgl"),
                      rmock.sexp(d["weights"], "real"), rmock.sexp([False], "lgl"), rmock.sexp([1e-2], "real"), rmock.sexp([True], "lgl"))
    L.rmock_reset()


def test_library_failures_surface_as_r_errors():
    """whatever the library refuses comes back through chk() as an R error with the library's message: without a device
    that is the very first call (the product has no CPU path); with one, an argument the library itself rejects"""
    import torch
    L = rmock.lib()
    L.rmock_reset()
    d = make_case(8, 6, "two_group", seed=3)
    if not torch.cuda.is_available():
        args, _ = _beta_args(d)
        with pytest.raises(rmock.RError, match="deseq2_mi355x: "):
            rmock.dotCall("_DESeq2_fitBeta", *args)
    else:
        args, _ = _beta_args(d, maxit=-1.0)
        with pytest.raises(rmock.RError, match="deseq2_mi355x: .*maxit"):
            rmock.dotCall("_DESeq2_fitBeta", *args)
    assert L.rmock_protect_depth() == 0 and L.rmock_live_transients() == 0
    L.rmock_reset()


# ---------------------------------------------------------------------------------------------------- on the GPU
def _same(a, b, what):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, "%s: shape %s vs %s" % (what, a.shape, b.shape)
    assert np.array_equal(a, b, equal_nan=True), "%s differs (max |d| %g)" % (what, np.nanmax(np.abs(a.astype(float) - b.astype(float))))


@pytest.mark.gpu
@pytest.mark.parametrize("y_kind", ["int", "real"])
def test_fitBeta_through_the_shim(y_kind):
    """_DESeq2_fitBeta with INTSXP and REALSXP counts, maxit as double 100: the list of src/DESeq2.cpp:458-464 -- names,
    REALSXP `iter` (:317), n x 1 contrast matrices -- and the values native.fitBeta returns"""
    from deseq2_amd import native
    L = rmock.lib()
    L.rmock_reset()
    d = make_case(300, 24, "batch_condition", seed=5)
    args, lam = _beta_args(d, y_kind)
    out = rmock.dotCall("_DESeq2_fitBeta", *args, keep=True)
    got = rmock.value(out)
    assert list(got) == ["beta_mat", "beta_var_mat", "iter", "hat_diagonals", "contrast_num", "contrast_denom", "deviance"]
    assert rmock.rtype(L.rmock_elt(out, 2)) == rmock.REALSXP and got["iter"].shape == (300,)
    assert got["contrast_num"].shape == (300, 1) and got["contrast_denom"].shape == (300, 1)
    assert got["hat_diagonals"].shape == (300, 24) and got["beta_mat"].shape == (300, 4)
    p = d["x"].shape[1]
    ref = native.fitBeta(d["counts"], d["x"], d["nf"], d["alpha_init"], np.r_[1.0, np.zeros(p - 1)], d["beta_init"], lam,
                         d["weights"], False, 1e-8, 100, True, 0.5)
    for k in got:
        _same(got[k], np.asarray(ref[k]).reshape(got[k].shape), "fitBeta$" + k)
    assert L.rmock_interrupt_polls() >= 2          # on entry and after the (single) range
    L.rmock_reset()


@pytest.mark.gpu
def test_fitDisp_and_fitDispGrid_through_the_shim():
    """_DESeq2_fitDisp: INTSXP `iter` / `iter_accept`, the nine names of src/DESeq2.cpp:268-276; _DESeq2_fitDispGrid: the
    one-element list of :512; integer scalars where R would pass them (maxit = 100L, usePrior as 0/1 integer)"""
    from deseq2_amd import native
    L = rmock.lib()
    L.rmock_reset()
    d = make_case(200, 30, "batch_condition", seed=9)
    mu = np.maximum(d["nf"] * np.exp(d["beta_init"] @ d["x"].T * np.log(2)), 0.5)
    la0 = np.log(d["alpha_init"])
    S = rmock.sexp
    out = rmock.dotCall("_DESeq2_fitDisp", S(d["counts"]), S(d["x"], "real"), S(mu, "real"), S(la0, "real"), S(la0, "real"),
                        S([1.0], "real"), S([np.log(1e-9)], "real"), S([1.0], "real"), S([1e-6], "real"), S([100], "int"),
                        S([0], "int"), S(d["weights"], "real"), S([False], "lgl"), S([1e-2], "real"), S([True], "lgl"), keep=True)
    got = rmock.value(out)
    assert list(got) == ["log_alpha", "iter", "iter_accept", "last_change", "initial_lp", "initial_dlp", "last_lp", "last_dlp",
                         "last_d2lp"]
    assert rmock.rtype(L.rmock_elt(out, 1)) == rmock.INTSXP and rmock.rtype(L.rmock_elt(out, 2)) == rmock.INTSXP
    assert rmock.rtype(L.rmock_elt(out, 0)) == rmock.REALSXP
    ref = native.fitDisp(d["counts"], d["x"], mu, la0, la0, 1.0, np.log(1e-9), 1.0, 1e-6, 100, False, d["weights"], False, 1e-2, True)
    for k in got:
        _same(got[k], ref[k], "fitDisp$" + k)
    grid = np.linspace(np.log(1e-8), np.log(30.0), 20)
    gg = rmock.dotCall("_DESeq2_fitDispGrid", S(d["counts"][:40]), S(d["x"], "real"), S(mu[:40], "real"), S(grid, "real"),
                       S(np.zeros(40), "real"), S([1.0], "real"), S([False], "lgl"), S(d["weights"][:40], "real"), S([False], "lgl"),
                       S([1e-2], "real"), S([True], "lgl"))
    assert list(gg) == ["log_alpha"]
    _same(gg["log_alpha"], native.fitDispGrid(d["counts"][:40], d["x"], mu[:40], grid, np.zeros(40), 1.0, False, d["weights"][:40],
                                              False, 1e-2, True)["log_alpha"], "fitDispGrid$log_alpha")
    L.rmock_reset()


@pytest.mark.gpu
def test_ranges_between_interrupt_polls_cover_every_row(monkeypatch):
    """the shim cuts a large call into ranges (about 2.5e7 matrix entries each) and polls R_CheckUserInterrupt between them
    (the reference polls every 100 genes, src/DESeq2.cpp:195): with the range length forced down to 1 000 rows a
    2 600-row call runs as three ranges and equals the one-range result"""
    from deseq2_amd import native
    L = rmock.lib()
    L.rmock_reset()
    d = make_case(2600, 6, "two_group", seed=21)
    n = d["counts"].shape[0]
    assert 2000 < n <= 2600
    args, lam = _beta_args(d)
    monkeypatch.setenv("DSQ_SHIM_ROWS", "1000")
    got = rmock.dotCall("_DESeq2_fitBeta", *args)
    assert L.rmock_interrupt_polls() == 4          # entry + three ranges
    monkeypatch.delenv("DSQ_SHIM_ROWS")
    ref = native.fitBeta(d["counts"], d["x"], d["nf"], d["alpha_init"], np.r_[1.0, 0.0], d["beta_init"], lam, d["weights"], False,
                         1e-8, 100, True, 0.5)
    for k in ("beta_mat", "iter", "deviance", "hat_diagonals"):
        _same(got[k], np.asarray(ref[k]).reshape(got[k].shape), "fitBeta$" + k)
    L.rmock_reset()


@pytest.mark.gpu
def test_an_error_in_the_middle_of_a_call_leaves_library_and_runtime_usable():
    """Rf_error out of the shim AFTER the library has run (a second call with a bad argument between two good ones):
    nothing is freed twice, no stale state -- the third call returns what the first did"""
    L = rmock.lib()
    L.rmock_reset()
    d = make_case(64, 12, "two_group", seed=2)
    args, _ = _beta_args(d)
    first = rmock.dotCall("_DESeq2_fitBeta", *args)
    bad, _ = _beta_args(d, maxit=-3.0)
    with pytest.raises(rmock.RError, match="maxit"):
        rmock.dotCall("_DESeq2_fitBeta", *bad)
    again = rmock.dotCall("_DESeq2_fitBeta", *args)
    for k in first:
        _same(first[k], again[k], "fitBeta$" + k)
    L.rmock_reset()


@pytest.mark.gpu
@pytest.mark.parametrize("test", ["Wald", "LRT"])
def test_DESeq_through_the_shim(test):
    """_DESeq2_mi355x_DESeq (31 arguments) against native.DESeq: NULL for absent arguments, LGLSXP /
    INTSXP columns with NA for the all-zero rows, the n x m assays asked for by the bit mask"""
    from scipy import special as sps
    from scipy.stats import f as fdist
    from deseq2_amd import native
    L = rmock.lib()
    L.rmock_reset()
    d = make_case(600, 16, "two_group", seed=13)
    counts = d["counts"].copy()
    counts[5] = 0                                   # an all-zero row: NA in the integer / logical columns
    n = counts.shape[0]
    x = d["x"]
    m, p = x.shape
    sf = np.exp(np.random.Generator(np.random.PCG64(4)).normal(0, 0.2, m))
    q, a, r = native.design_qr(x)
    S, NIL = rmock.sexp, rmock.sexp(None)
    red = np.ones((m, 1))
    args = [S(counts), S(x, "real"), S(sf, "real"), NIL, NIL, S(q, "real"), S(r, "real"), S([0 if test == "Wald" else 1], "int"),
            NIL if test == "Wald" else S(red, "real"), NIL, NIL, S([7.0], "real"), S([fdist.ppf(.99, p, m - p)], "real"),
            S([sps.polygamma(1, (m - p) / 2.0)], "real"), S([1e-8], "real"), S([100.0], "real"), S([True], "lgl"), S([0.5], "real"),
            S([100], "int"), S([True], "lgl"), S([1 | 4], "int"),
            S([False], "lgl"), NIL, NIL, NIL, NIL, NIL, S([0], "int"), NIL, S([False], "lgl"), NIL]
    out = rmock.dotCall("_DESeq2_mi355x_DESeq", *args, keep=True)
    got = rmock.value(out)
    ref = native.DESeq(counts, x, sf, test=test, reduced=None if test == "Wald" else red, assays=("mu", "cooks"))
    names = [L.rmock_name(out, i).decode() for i in range(L.rmock_length(out))]
    assert names[:4] == ["baseMean", "baseVar", "allZero", "dispGeneEst"] and len(names) == 28
    types = {n: rmock.rtype(L.rmock_elt(out, i)) for i, n in enumerate(names)}
    assert types["allZero"] == rmock.LGLSXP and types["dispIter"] == rmock.INTSXP and types["betaConv"] == rmock.LGLSXP
    assert got["H"] is None and got["replaceCounts"] is None and got["mu"].shape == (n, m)
    na_int = np.iinfo(np.int32).min
    assert got["dispIter"][5] == na_int and got["betaConv"][5] == na_int and got["allZero"][5] == 1
    assert L.rmock_is_na_real(float(got["dispersion"][5])) == 1          # NA_real_, not a plain NaN
    # the shim lets the library derive X R^-1 itself where native.DESeq hands numpy's over: same columns to rounding
    for k in ("baseMean", "baseVar", "dispGeneEst", "dispFit", "dispMAP", "dispersion", "logLike", "maxCooks", "beta", "betaSE"):
        np.testing.assert_allclose(got[k], ref[k], rtol=1e-9, atol=0, equal_nan=True, err_msg=k)
    for k in ("dispGeneIter", "dispIter", "dispOutlier", "betaConv", "replace"):
        g = got[k].astype(np.float64)
        g[got[k] == na_int] = np.nan
        _same(g, ref[k], k)
    if test == "Wald":
        np.testing.assert_allclose(got["stat"], ref["stat"], rtol=1e-8, equal_nan=True)
        assert got["stat"].shape == (n, p) and got["logLikeReduced"].shape == (n,)
    else:
        assert got["stat"].shape == (n, 0)
        np.testing.assert_allclose(got["logLikeReduced"], ref["logLikeReduced"], rtol=1e-9, equal_nan=True)
    np.testing.assert_allclose(got["mu"], ref["mu"], rtol=1e-9)
    assert got["dispersionFunction"].shape == (5,)
    L.rmock_reset()


@pytest.mark.gpu
def test_DESeq_declined_analyses_return_NULL():
    """DSQ_ERR_UNSUPPORTED (here: m - p <= 3 without a caller's dispPriorVar) is not an R error: the shim returns NULL and
    the R caller runs its unchanged code path"""
    from deseq2_amd import native
    L = rmock.lib()
    L.rmock_reset()
    d = make_case(100, 4, "two_group", seed=1)           # m - p = 2
    x = d["x"]
    m, p = x.shape
    q, a, r = native.design_qr(x)
    S, NIL = rmock.sexp, rmock.sexp(None)
    args = [S(d["counts"]), S(x, "real"), S(np.ones(m), "real"), NIL, NIL, S(q, "real"), S(r, "real"), S([0], "int"), NIL, NIL, NIL,
            S([7.0], "real"), S([19.0], "real"), S([1.6], "real"), S([1e-8], "real"), S([100.0], "real"), S([True], "lgl"),
            S([0.5], "real"), S([100], "int"), S([True], "lgl"), S([0], "int"), S([False], "lgl"), NIL, NIL, NIL, NIL, NIL, S([0], "int"),
            NIL, S([False], "lgl"), NIL]
    assert rmock.dotCall("_DESeq2_mi355x_DESeq", *args) is None
    L.rmock_reset()
