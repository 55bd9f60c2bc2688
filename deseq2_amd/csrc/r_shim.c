/*
 * r_shim.c -- the R-side binding of libdeseq2_mi355x.so: three `.Call` entry points with
 * the names, arities and return-list shapes of the reference's src/RcppExports.cpp:16-94
 * (_DESeq2_fitDisp 15 args, _DESeq2_fitBeta 13 args, _DESeq2_fitDispGrid 11 args), so that
 * R/RcppExports.R:4,8,12 and every R caller (R/wrappers.R:35,73,115, R/results.R:797,
 * tests/testthat/test_dispersions.R:67) work unchanged.
 *
 * Uses only Rinternals.h (no Rcpp, no Armadillo).  R is NOT installed in the build image
 * of this repository, so this file is compiled only where R exists (the whole file is inside
 * #ifdef DSQ_HAVE_R so that a build without R headers yields an empty object):
 *     PKG_CPPFLAGS="-DDSQ_HAVE_R -I../../include" PKG_LIBS="-L.. -ldeseq2_mi355x" R CMD SHLIB -o DESeq2.so r_shim.c
 * (see INTEGRATION.md).  It contains no arithmetic: all computation is in the HIP library.
 */
#ifdef DSQ_HAVE_R
#include <R.h>
#include <Rinternals.h>
#include <R_ext/Rdynload.h>
#include <R_ext/Utils.h>
#include <stdlib.h>
#include <string.h>
#include "deseq2_mi355x.h"

static void chk(int rc) {
    /* the reference turns C++ exceptions into R errors (BEGIN_RCPP/END_RCPP) */
    if (rc != DSQ_OK) Rf_error("deseq2_mi355x: %s", dsq_last_error());
}

/* counts arrive as INTSXP from counts(dds) but Rcpp coerced anything numeric
 * (src/DESeq2.cpp:165,285): accept both */
static const void *counts_ptr(SEXP y, int *type) {
    if (TYPEOF(y) == INTSXP) { *type = DSQ_Y_INT32; return INTEGER(y); }
    if (TYPEOF(y) == REALSXP) { *type = DSQ_Y_FLOAT64; return REAL(y); }
    Rf_error("ySEXP must be an integer or numeric matrix");
    return NULL;
}
/* the raw pointers handed to the library are trusted to hold n x m (n x p, n, p) values: check what R passed */
static void need_matrix(SEXP s, int nr, int nc, const char *what) {
    if (!Rf_isMatrix(s) || Rf_nrows(s) != nr || Rf_ncols(s) != nc)
        Rf_error("%s must be a %d x %d matrix", what, nr, nc);
}
static void need_length(SEXP s, int n, const char *what) {
    if (Rf_length(s) != n) Rf_error("%s must have length %d", what, n);
}
/* genes per library call: between two calls R_CheckUserInterrupt() runs (the reference polls every 100 genes,
 * src/DESeq2.cpp:195,320,493; a range here is ~50 ms of GPU work) */
static int rows_per_call(int n, int m) {
    const char *e = getenv("DSQ_SHIM_ROWS");              /* (tests: force several ranges on a small matrix) */
    if (e && atoi(e) > 0) return atoi(e) > n ? n : atoi(e);
    double r = 2.5e7 / (double)(m > 0 ? m : 1);
    if (r < 4096.0) r = 4096.0;
    return r > (double)n ? n : (int)r;
}
static double scalar_d(SEXP s) { return Rf_asReal(s); }
static int scalar_i(SEXP s) { return Rf_asInteger(s); }      /* maxit may arrive as double 100 */
static int scalar_b(SEXP s) { return Rf_asLogical(s) == TRUE; }
static SEXP as_real(SEXP s, int *np) { SEXP r = PROTECT(Rf_coerceVector(s, REALSXP)); (*np)++; return r; }

static SEXP named_list(int k, const char **names, SEXP *vals) {
    SEXP out = PROTECT(Rf_allocVector(VECSXP, k)), nm = PROTECT(Rf_allocVector(STRSXP, k));
    for (int i = 0; i < k; i++) { SET_VECTOR_ELT(out, i, vals[i]); SET_STRING_ELT(nm, i, Rf_mkChar(names[i])); }
    Rf_setAttrib(out, R_NamesSymbol, nm);
    UNPROTECT(2);
    return out;
}

SEXP _DESeq2_fitBeta(SEXP ySEXP, SEXP xSEXP, SEXP nfSEXP, SEXP alpha_hatSEXP, SEXP contrastSEXP,
                     SEXP beta_matSEXP, SEXP lambdaSEXP, SEXP weightsSEXP, SEXP useWeightsSEXP,
                     SEXP tolSEXP, SEXP maxitSEXP, SEXP useQRSEXP, SEXP minmuSEXP) {
    int np = 0;
    R_CheckUserInterrupt();
    int n = Rf_nrows(ySEXP), m = Rf_ncols(ySEXP), p = Rf_ncols(xSEXP);
    need_matrix(xSEXP, m, p, "xSEXP"); need_matrix(nfSEXP, n, m, "nfSEXP"); need_matrix(beta_matSEXP, n, p, "beta_matSEXP");
    need_matrix(weightsSEXP, n, m, "weightsSEXP"); need_length(alpha_hatSEXP, n, "alpha_hatSEXP");
    need_length(contrastSEXP, p, "contrastSEXP"); need_length(lambdaSEXP, p, "lambdaSEXP");
    SEXP x = as_real(xSEXP, &np), nf = as_real(nfSEXP, &np), alpha = as_real(alpha_hatSEXP, &np);
    SEXP con = as_real(contrastSEXP, &np), b0 = as_real(beta_matSEXP, &np), lam = as_real(lambdaSEXP, &np);
    SEXP w = as_real(weightsSEXP, &np);
    DsqFitBetaArgs a = {0};
    a.n = n; a.m = m; a.p = p; a.layout = DSQ_LAYOUT_R;
    a.y = counts_ptr(ySEXP, &a.y_type);
    a.x = REAL(x); a.nf = REAL(nf); a.alpha_hat = REAL(alpha); a.contrast = REAL(con);
    a.beta_mat = REAL(b0); a.lambda = REAL(lam); a.weights = REAL(w);
    a.useWeights = scalar_b(useWeightsSEXP); a.tol = scalar_d(tolSEXP); a.maxit = scalar_i(maxitSEXP);
    a.useQR = scalar_b(useQRSEXP); a.minmu = scalar_d(minmuSEXP);
    SEXP beta = PROTECT(Rf_allocMatrix(REALSXP, n, p)); np++;
    SEXP var = PROTECT(Rf_allocMatrix(REALSXP, n, p)); np++;
    SEXP iter = PROTECT(Rf_allocVector(REALSXP, n)); np++;      /* NumericVector in the reference (:317) */
    SEXP hat = PROTECT(Rf_allocMatrix(REALSXP, n, m)); np++;
    SEXP cn = PROTECT(Rf_allocMatrix(REALSXP, n, 1)); np++;
    SEXP cd = PROTECT(Rf_allocMatrix(REALSXP, n, 1)); np++;
    SEXP dev = PROTECT(Rf_allocVector(REALSXP, n)); np++;
    DsqFitBetaOut o = {0};
    o.beta_mat = REAL(beta); o.beta_var_mat = REAL(var); o.iter = REAL(iter); o.hat_diagonals = REAL(hat);
    o.contrast_num = REAL(cn); o.contrast_denom = REAL(cd); o.deviance = REAL(dev);
    for (int lo = 0, step = rows_per_call(n, m); lo < n; lo += step) {
        chk(dsq_fit_beta_rows(&a, &o, lo, (lo + step < n) ? step : n - lo));
        R_CheckUserInterrupt();
    }
    const char *names[] = {"beta_mat", "beta_var_mat", "iter", "hat_diagonals", "contrast_num",
                           "contrast_denom", "deviance"};                        /* src/DESeq2.cpp:458-464 */
    SEXP vals[] = {beta, var, iter, hat, cn, cd, dev};
    SEXP out = named_list(7, names, vals);
    UNPROTECT(np);
    return out;
}

SEXP _DESeq2_fitDisp(SEXP ySEXP, SEXP xSEXP, SEXP mu_hatSEXP, SEXP log_alphaSEXP,
                     SEXP log_alpha_prior_meanSEXP, SEXP log_alpha_prior_sigmasqSEXP,
                     SEXP min_log_alphaSEXP, SEXP kappa_0SEXP, SEXP tolSEXP, SEXP maxitSEXP,
                     SEXP usePriorSEXP, SEXP weightsSEXP, SEXP useWeightsSEXP, SEXP weightThresholdSEXP,
                     SEXP useCRSEXP) {
    int np = 0;
    R_CheckUserInterrupt();
    int n = Rf_nrows(ySEXP), m = Rf_ncols(ySEXP), p = Rf_ncols(xSEXP);
    need_matrix(xSEXP, m, p, "xSEXP"); need_matrix(mu_hatSEXP, n, m, "mu_hatSEXP"); need_matrix(weightsSEXP, n, m, "weightsSEXP");
    need_length(log_alphaSEXP, n, "log_alphaSEXP"); need_length(log_alpha_prior_meanSEXP, n, "log_alpha_prior_meanSEXP");
    SEXP x = as_real(xSEXP, &np), mu = as_real(mu_hatSEXP, &np), la = as_real(log_alphaSEXP, &np);
    SEXP pm = as_real(log_alpha_prior_meanSEXP, &np), w = as_real(weightsSEXP, &np);
    DsqFitDispArgs a = {0};
    a.n = n; a.m = m; a.p = p; a.layout = DSQ_LAYOUT_R;
    a.y = counts_ptr(ySEXP, &a.y_type);
    a.x = REAL(x); a.mu_hat = REAL(mu); a.log_alpha = REAL(la); a.log_alpha_prior_mean = REAL(pm);
    a.log_alpha_prior_sigmasq = scalar_d(log_alpha_prior_sigmasqSEXP);
    a.min_log_alpha = scalar_d(min_log_alphaSEXP); a.kappa_0 = scalar_d(kappa_0SEXP);
    a.tol = scalar_d(tolSEXP); a.maxit = scalar_i(maxitSEXP); a.usePrior = scalar_b(usePriorSEXP);
    a.weights = REAL(w); a.useWeights = scalar_b(useWeightsSEXP);
    a.weightThreshold = scalar_d(weightThresholdSEXP); a.useCR = scalar_b(useCRSEXP);
    const char *names[] = {"log_alpha", "iter", "iter_accept", "last_change", "initial_lp", "initial_dlp",
                           "last_lp", "last_dlp", "last_d2lp"};                  /* src/DESeq2.cpp:268-276 */
    SEXP vals[9];
    for (int i = 0; i < 9; i++) {
        vals[i] = PROTECT(Rf_allocVector((i == 1 || i == 2) ? INTSXP : REALSXP, n)); np++;
    }
    DsqFitDispOut o = {0};
    o.log_alpha = REAL(vals[0]); o.iter = INTEGER(vals[1]); o.iter_accept = INTEGER(vals[2]);
    o.last_change = REAL(vals[3]); o.initial_lp = REAL(vals[4]); o.initial_dlp = REAL(vals[5]);
    o.last_lp = REAL(vals[6]); o.last_dlp = REAL(vals[7]); o.last_d2lp = REAL(vals[8]);
    for (int lo = 0, step = rows_per_call(n, m); lo < n; lo += step) {
        chk(dsq_fit_disp_rows(&a, &o, lo, (lo + step < n) ? step : n - lo));
        R_CheckUserInterrupt();
    }
    SEXP out = named_list(9, names, vals);
    UNPROTECT(np);
    return out;
}

SEXP _DESeq2_fitDispGrid(SEXP ySEXP, SEXP xSEXP, SEXP mu_hatSEXP, SEXP disp_gridSEXP,
                         SEXP log_alpha_prior_meanSEXP, SEXP log_alpha_prior_sigmasqSEXP, SEXP usePriorSEXP,
                         SEXP weightsSEXP, SEXP useWeightsSEXP, SEXP weightThresholdSEXP, SEXP useCRSEXP) {
    int np = 0;
    R_CheckUserInterrupt();
    int n = Rf_nrows(ySEXP), m = Rf_ncols(ySEXP), p = Rf_ncols(xSEXP);
    need_matrix(xSEXP, m, p, "xSEXP"); need_matrix(mu_hatSEXP, n, m, "mu_hatSEXP"); need_matrix(weightsSEXP, n, m, "weightsSEXP");
    need_length(log_alpha_prior_meanSEXP, n, "log_alpha_prior_meanSEXP");
    if (Rf_length(disp_gridSEXP) < 2) Rf_error("disp_gridSEXP must hold at least 2 grid points");
    SEXP x = as_real(xSEXP, &np), mu = as_real(mu_hatSEXP, &np), grid = as_real(disp_gridSEXP, &np);
    SEXP pm = as_real(log_alpha_prior_meanSEXP, &np), w = as_real(weightsSEXP, &np);
    DsqFitDispGridArgs a = {0};
    a.n = n; a.m = m; a.p = p; a.layout = DSQ_LAYOUT_R;
    a.y = counts_ptr(ySEXP, &a.y_type);
    a.x = REAL(x); a.mu_hat = REAL(mu); a.disp_grid = REAL(grid); a.ngrid = Rf_length(grid);
    a.log_alpha_prior_mean = REAL(pm); a.log_alpha_prior_sigmasq = scalar_d(log_alpha_prior_sigmasqSEXP);
    a.usePrior = scalar_b(usePriorSEXP); a.weights = REAL(w); a.useWeights = scalar_b(useWeightsSEXP);
    a.weightThreshold = scalar_d(weightThresholdSEXP); a.useCR = scalar_b(useCRSEXP);
    SEXP la = PROTECT(Rf_allocVector(REALSXP, n)); np++;
    DsqFitDispGridOut o = {0};
    o.log_alpha = REAL(la);
    for (int lo = 0, step = rows_per_call(n, m); lo < n; lo += step) {
        chk(dsq_fit_disp_grid_rows(&a, &o, lo, (lo + step < n) ? step : n - lo));
        R_CheckUserInterrupt();
    }
    const char *names[] = {"log_alpha"};                                         /* src/DESeq2.cpp:512 */
    SEXP vals[] = {la};
    SEXP out = named_list(1, names, vals);
    UNPROTECT(np);
    return out;
}

/* ---- optional extension entry points: not part of the reference's .Call table; the R-side
 * one-liners that would call them are in INTEGRATION.md section 3 ------------------------------------ */

/* nbinomLogLike(counts, mu, disp, weights, useWeights)   R/core.R:2208-2217 */
SEXP _DESeq2_mi355x_nbinomLogLike(SEXP ySEXP, SEXP muSEXP, SEXP dispSEXP, SEXP weightsSEXP, SEXP useWeightsSEXP) {
    int np = 0;
    int n = Rf_nrows(ySEXP), m = Rf_ncols(ySEXP);
    SEXP mu = as_real(muSEXP, &np), disp = as_real(dispSEXP, &np), w = as_real(weightsSEXP, &np);
    if (Rf_length(disp) != n) Rf_error("disp must have one value per row");
    DsqLogLikeArgs a = {0};
    a.n = n; a.m = m; a.layout = DSQ_LAYOUT_R;
    a.y = counts_ptr(ySEXP, &a.y_type);
    a.mu = REAL(mu); a.disp = REAL(disp); a.weights = REAL(w); a.useWeights = scalar_b(useWeightsSEXP);
    SEXP ll = PROTECT(Rf_allocVector(REALSXP, n)); np++;
    chk(dsq_nbinom_loglike(&a, REAL(ll)));
    UNPROTECT(np);
    return ll;
}

/* calculateCooksDistance + recordMaxCooks   R/core.R:2333-2359; `cells` = 0-based design-cell id per sample
 * (match(rows of the dispersion model matrix, unique rows) - 1L), p = ncol(modelMatrix) */
SEXP _DESeq2_mi355x_cooks(SEXP ySEXP, SEXP nfSEXP, SEXP muSEXP, SEXP hSEXP, SEXP cellsSEXP, SEXP pSEXP) {
    int np = 0;
    int n = Rf_nrows(ySEXP), m = Rf_ncols(ySEXP);
    SEXP nf = as_real(nfSEXP, &np), mu = as_real(muSEXP, &np), h = as_real(hSEXP, &np);
    SEXP cells = PROTECT(Rf_coerceVector(cellsSEXP, INTSXP)); np++;
    if (Rf_length(cells) != m) Rf_error("cells must have one entry per sample");
    int ncell = 0;
    for (int j = 0; j < m; j++) if (INTEGER(cells)[j] + 1 > ncell) ncell = INTEGER(cells)[j] + 1;
    DsqCooksArgs a = {0};
    a.n = n; a.m = m; a.p = scalar_i(pSEXP); a.layout = DSQ_LAYOUT_R;
    a.y = counts_ptr(ySEXP, &a.y_type);
    a.nf = REAL(nf); a.nf_is_vector = (Rf_length(nf) == m && n != 1);
    a.mu = REAL(mu); a.H = REAL(h); a.cell_of = INTEGER(cells); a.ncell = ncell;
    SEXP ck = PROTECT(Rf_allocMatrix(REALSXP, n, m)); np++;
    SEXP mx = PROTECT(Rf_allocVector(REALSXP, n)); np++;
    DsqCooksOut o = {0};
    o.cooks = REAL(ck); o.maxCooks = REAL(mx);
    chk(dsq_cooks_distance(&a, &o));
    for (int i = 0; i < n; i++) if (ISNAN(REAL(mx)[i])) REAL(mx)[i] = NA_REAL;      /* rep(NA, numRow) :2356 */
    const char *names[] = {"cooks", "maxCooks"};
    SEXP vals[] = {ck, mx};
    SEXP out = named_list(2, names, vals);
    UNPROTECT(np);
    return out;
}

/* replaceOutliers' count replacement   R/core.R:2083-2112 */
SEXP _DESeq2_mi355x_replace(SEXP ySEXP, SEXP nfSEXP, SEXP cooksSEXP, SEXP cutoffSEXP, SEXP trimSEXP,
                            SEXP replaceableSEXP) {
    int np = 0;
    int n = Rf_nrows(ySEXP), m = Rf_ncols(ySEXP);
    SEXP nf = as_real(nfSEXP, &np), ck = as_real(cooksSEXP, &np);
    SEXP rep = PROTECT(Rf_coerceVector(replaceableSEXP, LGLSXP)); np++;
    if (Rf_length(rep) != m) Rf_error("replaceable must have one entry per sample");
    int *flags = (int *)R_alloc(m, sizeof(int));
    for (int j = 0; j < m; j++) flags[j] = LOGICAL(rep)[j] == TRUE;
    DsqReplaceArgs a = {0};
    a.n = n; a.m = m; a.layout = DSQ_LAYOUT_R;
    a.y = counts_ptr(ySEXP, &a.y_type);
    a.nf = REAL(nf); a.nf_is_vector = (Rf_length(nf) == m && n != 1);
    a.cooks = REAL(ck); a.cooksCutoff = scalar_d(cutoffSEXP); a.trim = scalar_d(trimSEXP); a.replaceable = flags;
    SEXP newc = PROTECT(Rf_allocMatrix(INTSXP, n, m)); np++;
    SEXP flag = PROTECT(Rf_allocVector(LGLSXP, n)); np++;
    DsqReplaceOut o = {0};
    o.newCounts = INTEGER(newc); o.replace = LOGICAL(flag);
    chk(dsq_replace_outliers(&a, &o));
    const char *names[] = {"counts", "replace"};
    SEXP vals[] = {newc, flag};
    SEXP out = named_list(2, names, vals);
    UNPROTECT(np);
    return out;
}

/* ---- DESeq() behind ONE call (dsq_deseq): what the patch of R/core.R:388-426 in INTEGRATION.md section 4 calls in place
 * of estimateDispersions -> nbinomWaldTest / nbinomLRT -> refitWithoutOutliers.  Arguments: counts(object) (integer
 * matrix), the model matrix, sizeFactors(object), normalizationFactors(object) or NULL, assays(object)[["weights"]] or
 * NULL, qr.Q(qrx), qr.R(qrx) (qrx <- qr(modelMatrix), R/fitNbinomGLMs.R:139-143),
 * test (0 Wald / 1 LRT), the reduced model matrix with its qr.Q / qr.R (NULL, or one column: reduced = ~1),
 * minReplicatesForReplace, qf(.99, p, m - p) (R/core.R:2081),
 * trigamma((m - p) / 2) (:1196), betaTol, maxit, useQR, minmu, the dispersion searches' maxit, useCR, and which n x m
 * assays to bring back (a character-free bit mask: 1 mu, 2 H, 4 cooks, 8 replaceCounts).
 * Returns NULL when the library declines the analysis (DSQ_ERR_UNSUPPORTED: p <= 24 -- 10 < p only without beta prior / weights --,
 * m - p > 3; DSQ_ERR_FIT: the parametric trend failed / no usable gene) -- the R caller then runs its unchanged code
 * path over the three classic routines; any other failure is an R error. ---------------------------------------- */
static SEXP int_col(const int *v, int n, int type, int *np) {      /* -1 -> NA */
    SEXP s = PROTECT(Rf_allocVector(type, n)); (*np)++;
    int *d = (type == LGLSXP) ? LOGICAL(s) : INTEGER(s);
    for (int i = 0; i < n; i++) d[i] = (v[i] < 0) ? R_NaInt : v[i];
    return s;
}
static void nan_to_na(SEXP s) {
    double *d = REAL(s);
    int k = Rf_length(s);
    for (int i = 0; i < k; i++) if (ISNAN(d[i])) d[i] = NA_REAL;
}

SEXP _DESeq2_mi355x_DESeq(SEXP countsSEXP, SEXP xSEXP, SEXP sizeFactorsSEXP, SEXP normFactorsSEXP, SEXP weightsSEXP,
                          SEXP qSEXP, SEXP rSEXP, SEXP testSEXP,
                          SEXP xRedSEXP, SEXP qRedSEXP, SEXP rRedSEXP, SEXP minReplicatesSEXP, SEXP cooksCutoffSEXP, SEXP expVarLogDispSEXP, SEXP betaTolSEXP,
                          SEXP maxitSEXP, SEXP useQRSEXP, SEXP minmuSEXP, SEXP dispMaxitSEXP, SEXP useCRSEXP,
                          SEXP assaysSEXP,
                          /* nbinomWaldTest(betaPrior = TRUE): logical; the expanded model matrix or NULL (standard); integer
                           * codes of the columns of the model matrix / of the expanded one (0 intercept, f level of factor f,
                           * -1 other) and, for the -1 columns of the expanded one, the model-matrix column of the same name
                           * (0-based); the user's betaPriorVar or NULL */
                          SEXP betaPriorSEXP, SEXP xPriorSEXP, SEXP coefFactorSEXP, SEXP priorCoefFactorSEXP,
                          SEXP priorCoefSrcSEXP, SEXP betaPriorVarSEXP,
                          /* estimateDispersionsFit's fitType as DSQ_FIT_*: 0 "parametric" (a trend that does not fit returns
                           * NULL and the caller takes the reference's route to locfit, R/core.R:885-893), 1 "mean" */
                          SEXP fitTypeSEXP,
                          /* a trend R fits itself (fitType = "local", dispersionFunction<-): geneEstOnly TRUE = stop after
                           * estimateDispersionsGeneEst; dispFit = the trend at res$baseMean (or NULL), see dsq_deseq */
                          SEXP dispFitSEXP, SEXP geneEstOnlySEXP,
                          /* estimateDispersionsMAP(dispPriorVar = x) or NULL (estimated; m - p <= 3: required) */
                          SEXP dispPriorVarSEXP) {
    int np = 0;
    R_CheckUserInterrupt();
    int n = Rf_nrows(countsSEXP), m = Rf_ncols(countsSEXP), p = Rf_ncols(xSEXP);
    need_matrix(xSEXP, m, p, "modelMatrix"); need_matrix(qSEXP, m, p, "qr.Q"); need_matrix(rSEXP, p, p, "qr.R");
    SEXP x = as_real(xSEXP, &np), q = as_real(qSEXP, &np), r = as_real(rSEXP, &np);
    const int want = scalar_i(assaysSEXP), wald = scalar_i(testSEXP) == 0;
    DsqDeseqHostArgs a = {0};
    a.n = n; a.m = m; a.p = p;
    a.counts = counts_ptr(countsSEXP, &a.y_type);
    a.x = REAL(x); a.q = REAL(q); a.r = REAL(r); a.xrinv = NULL;
    if (normFactorsSEXP != R_NilValue) {              /* normalizationFactors(object) take precedence (R/core.R:2221-2227) */
        need_matrix(normFactorsSEXP, n, m, "normalizationFactors");
        SEXP nf = as_real(normFactorsSEXP, &np);
        a.normalizationFactors = REAL(nf);
    } else {
        need_length(sizeFactorsSEXP, m, "sizeFactors");
        SEXP sf = as_real(sizeFactorsSEXP, &np);
        a.sizeFactors = REAL(sf);
    }
    if (weightsSEXP != R_NilValue) {
        need_matrix(weightsSEXP, n, m, "weights");
        SEXP w = as_real(weightsSEXP, &np);
        a.weights = REAL(w);
    }
    a.test = wald ? 0 : 1;
    /* the closed form of R/fitNbinomGLMs.R:99-137 is taken only for a reduced model matrix that IS the intercept: one
     * column AND all(modelMatrix == 1) (:99-103).  Any other one-column reduced model (~ 0 + x) is fitted by the IRLS
     * like a wider one. */
    int red_is_intercept = (xRedSEXP == R_NilValue);
    if (!wald && xRedSEXP != R_NilValue && Rf_ncols(xRedSEXP) == 1) {
        need_matrix(xRedSEXP, m, 1, "reduced model matrix");
        SEXP x1 = as_real(xRedSEXP, &np);
        red_is_intercept = 1;
        for (int j = 0; j < m; j++) if (REAL(x1)[j] != 1.0) { red_is_intercept = 0; break; }
    }
    if (!wald && !red_is_intercept) {
        int pr = Rf_ncols(xRedSEXP);
        need_matrix(xRedSEXP, m, pr, "reduced model matrix"); need_matrix(qRedSEXP, m, pr, "qr.Q(reduced)");
        need_matrix(rRedSEXP, pr, pr, "qr.R(reduced)");
        SEXP xr = as_real(xRedSEXP, &np), qr = as_real(qRedSEXP, &np), rr = as_real(rRedSEXP, &np);
        a.x_reduced = REAL(xr); a.q_reduced = REAL(qr); a.r_reduced = REAL(rr); a.p_reduced = pr;
    }
    int pcol = p;
    if (scalar_b(betaPriorSEXP)) {
        a.betaPrior = 1;
        need_length(coefFactorSEXP, p, "coefficient codes of the model matrix");
        SEXP cf = PROTECT(Rf_coerceVector(coefFactorSEXP, INTSXP)); np++;
        a.coef_factor = INTEGER(cf);
        if (xPriorSEXP != R_NilValue) {
            pcol = Rf_ncols(xPriorSEXP);
            need_matrix(xPriorSEXP, m, pcol, "expanded model matrix");
            need_length(priorCoefFactorSEXP, pcol, "coefficient codes of the expanded model matrix");
            SEXP xp = as_real(xPriorSEXP, &np);
            SEXP pcf = PROTECT(Rf_coerceVector(priorCoefFactorSEXP, INTSXP)); np++;
            a.x_prior = REAL(xp); a.p_prior = pcol; a.prior_coef_factor = INTEGER(pcf);
            if (priorCoefSrcSEXP != R_NilValue) {
                need_length(priorCoefSrcSEXP, pcol, "source columns of the expanded model matrix");
                SEXP pcs = PROTECT(Rf_coerceVector(priorCoefSrcSEXP, INTSXP)); np++;
                a.prior_coef_src = INTEGER(pcs);
            }
        }
        if (betaPriorVarSEXP != R_NilValue) {
            need_length(betaPriorVarSEXP, pcol, "betaPriorVar");
            SEXP bv = as_real(betaPriorVarSEXP, &np);
            a.betaPriorVar = REAL(bv);
        }
    }
    a.minReplicatesForReplace = scalar_d(minReplicatesSEXP);
    a.cooksCutoff = scalar_d(cooksCutoffSEXP); a.expVarLogDisp = scalar_d(expVarLogDispSEXP);
    a.betaTol = scalar_d(betaTolSEXP); a.maxit = scalar_i(maxitSEXP); a.useQR = scalar_b(useQRSEXP);
    a.minmu = scalar_d(minmuSEXP); a.disp_maxit = scalar_i(dispMaxitSEXP); a.useCR = scalar_b(useCRSEXP);
    a.fitType = scalar_i(fitTypeSEXP);
    a.geneEstOnly = scalar_b(geneEstOnlySEXP);
    if (dispPriorVarSEXP != R_NilValue) a.dispPriorVar = scalar_d(dispPriorVarSEXP);
    if (dispFitSEXP != R_NilValue) {
        need_length(dispFitSEXP, n, "dispFit");
        SEXP df = as_real(dispFitSEXP, &np);
        a.dispFit = REAL(df);
    }
    /* double columns straight into fresh R vectors; integer columns through scratch (NA_integer_ / NA for -1) */
    enum { BM, BV, DGE, DFIT, DMAP, DISP, BITER, LL, LLR, MAXC, NDBL };
    SEXP dv[NDBL];
    for (int k = 0; k < NDBL; k++) { dv[k] = PROTECT(Rf_allocVector(REALSXP, n)); np++; }
    SEXP beta = PROTECT(Rf_allocMatrix(REALSXP, n, pcol)); np++;
    SEXP se = PROTECT(Rf_allocMatrix(REALSXP, n, pcol)); np++;
    SEXP stat = PROTECT(Rf_allocMatrix(REALSXP, n, wald ? pcol : 0)); np++;
    SEXP pval = PROTECT(Rf_allocMatrix(REALSXP, n, wald ? pcol : 0)); np++;
    SEXP mle = PROTECT(Rf_allocMatrix(REALSXP, n, a.betaPrior ? p : 0)); np++;
    int *iv = (int *)R_alloc((size_t)7 * n, sizeof(int));
    SEXP mu = R_NilValue, H = R_NilValue, ck = R_NilValue, rc = R_NilValue;
    if (want & 1) { mu = PROTECT(Rf_allocMatrix(REALSXP, n, m)); np++; }
    if (want & 2) { H = PROTECT(Rf_allocMatrix(REALSXP, n, m)); np++; }
    if (want & 4) { ck = PROTECT(Rf_allocMatrix(REALSXP, n, m)); np++; }
    if (want & 8) { rc = PROTECT(Rf_allocMatrix(INTSXP, n, m)); np++; }
    DsqDeseqHostOut o;
    memset(&o, 0, sizeof o);
    o.baseMean = REAL(dv[BM]); o.baseVar = REAL(dv[BV]); o.dispGeneEst = REAL(dv[DGE]); o.dispFit = REAL(dv[DFIT]);
    o.dispMAP = REAL(dv[DMAP]); o.dispersion = REAL(dv[DISP]); o.betaIter = REAL(dv[BITER]); o.logLike = REAL(dv[LL]);
    o.logLikeReduced = wald ? NULL : REAL(dv[LLR]); o.maxCooks = REAL(dv[MAXC]);
    o.beta = REAL(beta); o.betaSE = REAL(se); o.stat = wald ? REAL(stat) : NULL; o.pvalue = wald ? REAL(pval) : NULL;
    o.allZero = iv; o.dispGeneIter = iv + n; o.dispIter = iv + 2 * (size_t)n; o.dispOutlier = iv + 3 * (size_t)n;
    o.betaConv = iv + 4 * (size_t)n; o.replace = iv + 5 * (size_t)n; o.weightsFail = iv + 6 * (size_t)n;
    if (want & 1) o.mu = REAL(mu);
    if (want & 2) o.H = REAL(H);
    if (want & 4) o.cooks = REAL(ck);
    if (want & 8) o.replaceCounts = INTEGER(rc);
    if (a.betaPrior) o.mle_beta = REAL(mle);
    int status = dsq_deseq(&a, &o);
    if (status == DSQ_ERR_UNSUPPORTED || status == DSQ_ERR_FIT) { UNPROTECT(np); return R_NilValue; }
    chk(status);
    for (int k = 0; k < NDBL; k++) nan_to_na(dv[k]);
    nan_to_na(beta); nan_to_na(se); nan_to_na(stat); nan_to_na(pval); nan_to_na(mle);
    /* asymptDisp, extraPois (the mean, 0 when fitType "mean" was used: [4] says which), varLogDispEsts, dispPriorVar */
    SEXP fn = PROTECT(Rf_allocVector(REALSXP, 5)); np++;
    for (int k = 0; k < 5; k++) REAL(fn)[k] = o.dispersionFunction[k];                  /* DSQ_SC_COEF0 .. DSQ_SC_FIT_USED */
    SEXP bpv = PROTECT(Rf_allocVector(REALSXP, a.betaPrior ? pcol : 0)); np++;
    for (int k = 0; k < Rf_length(bpv); k++) REAL(bpv)[k] = o.betaPriorVar[k];
    const char *names[] = {"baseMean", "baseVar", "allZero", "dispGeneEst", "dispGeneIter", "dispFit", "dispMAP",
                           "dispersion", "dispIter", "dispOutlier", "beta", "betaSE", "stat", "pvalue", "betaConv",
                           "betaIter", "logLike", "logLikeReduced", "maxCooks", "replace", "weightsFail", "mu", "H", "cooks",
                           "replaceCounts", "dispersionFunction", "MLE_beta", "betaPriorVar"};
    SEXP vals[] = {dv[BM], dv[BV], int_col(o.allZero, n, LGLSXP, &np), dv[DGE], int_col(o.dispGeneIter, n, INTSXP, &np),
                   dv[DFIT], dv[DMAP], dv[DISP], int_col(o.dispIter, n, INTSXP, &np), int_col(o.dispOutlier, n, LGLSXP, &np),
                   beta, se, stat, pval, int_col(o.betaConv, n, LGLSXP, &np), dv[BITER], dv[LL], dv[LLR], dv[MAXC],
                   int_col(o.replace, n, LGLSXP, &np), int_col(o.weightsFail, n, LGLSXP, &np), mu, H, ck, rc, fn, mle, bpv};
    SEXP out = named_list(28, names, vals);
    UNPROTECT(np);
    return out;
}

static const R_CallMethodDef CallEntries[] = {
    {"_DESeq2_fitDisp", (DL_FUNC)&_DESeq2_fitDisp, 15},
    {"_DESeq2_fitBeta", (DL_FUNC)&_DESeq2_fitBeta, 13},
    {"_DESeq2_fitDispGrid", (DL_FUNC)&_DESeq2_fitDispGrid, 11},
    {"_DESeq2_mi355x_nbinomLogLike", (DL_FUNC)&_DESeq2_mi355x_nbinomLogLike, 5},
    {"_DESeq2_mi355x_cooks", (DL_FUNC)&_DESeq2_mi355x_cooks, 6},
    {"_DESeq2_mi355x_replace", (DL_FUNC)&_DESeq2_mi355x_replace, 6},
    {"_DESeq2_mi355x_DESeq", (DL_FUNC)&_DESeq2_mi355x_DESeq, 31},
    {NULL, NULL, 0}};

void R_init_DESeq2(DllInfo *dll) {
    R_registerRoutines(dll, NULL, CallEntries, NULL, NULL);
    R_useDynamicSymbols(dll, FALSE);
}
#endif /* DSQ_HAVE_R */
