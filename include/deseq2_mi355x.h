/*
 * deseq2_mi355x.h -- C ABI of libdeseq2_mi355x.so, the MI355X (gfx950) engine that
 * replaces the three native entry points of thelovelab/DESeq2:
 *
 *   reference routine (src/RcppExports.cpp:84-89)      replaced by
 *   _DESeq2_fitBeta      (src/DESeq2.cpp:283-465)      dsq_fit_beta      / dsq_fit_beta_dev
 *   _DESeq2_fitDisp      (src/DESeq2.cpp:164-277)      dsq_fit_disp      / dsq_fit_disp_dev
 *   _DESeq2_fitDispGrid  (src/DESeq2.cpp:469-513)      dsq_fit_disp_grid / dsq_fit_disp_grid_dev
 *
 * Plain pointers and sizes only.  Two families:
 *   dsq_fit_*      : every array pointer is a HOST pointer in R's layout (what a
 *                    .Call shim gets from REAL()/INTEGER()); the call uploads, runs
 *                    the HIP kernels, downloads and returns synchronously.  This is
 *                    what src/r_shim.c binds.
 *   dsq_fit_*_dev  : every array pointer is a DEVICE pointer; work is enqueued on
 *                    `stream` (a hipStream_t passed as void*, NULL = default stream)
 *                    and the call returns without synchronising.  Used by the
 *                    Python host mirror and bench.py to keep Y / nf / weights / mu
 *                    resident in HBM across the four calls of one DESeq() fit.
 *
 * Matrix layouts (the `layout` field):
 *   DSQ_LAYOUT_R           n x m matrices are column-major as in R: (i,j) at i + n*j
 *   DSQ_LAYOUT_GENE_MAJOR  n x m matrices are row-major with leading dimension `ld`
 *                          (elements): (i,j) at i*ld + j.  This is the engine's
 *                          native layout (one wavefront per gene reads a coalesced
 *                          row); R-layout inputs are transposed on the device first.
 * n x p matrices (beta_mat, beta_var_mat) and n-vectors are ALWAYS in R layout
 * (column-major n x p).  x is m x p column-major, contrast/lambda are p-vectors.
 *
 * Semantics (argument meaning, iteration counters, break conditions, clamps, output
 * list members) follow the reference line by line; see DESIGN.md.  Inputs are
 * borrowed and never written.  No CPU fallback exists: without a usable gfx950
 * device every call fails with DSQ_ERR_DEVICE.
 */
#ifndef DESEQ2_MI355X_H
#define DESEQ2_MI355X_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DSQ_VERSION 100

enum {
    DSQ_OK = 0,
    DSQ_ERR_ARG = 1,         /* bad size / NULL pointer / inconsistent arguments          */
    DSQ_ERR_UNSUPPORTED = 2, /* p or m outside the compiled kernel range                  */
    DSQ_ERR_DEVICE = 3,      /* no gfx950 device, HIP runtime error, kernel launch error  */
    DSQ_ERR_NOMEM = 4,       /* device or host allocation failed                          */
    DSQ_ERR_VALUE = 5,       /* non-integer / negative count in a REALSXP count matrix    */
    DSQ_ERR_FIT = 6          /* dsq_deseq: a condition the reference turns into an R error or a change of method
                                (all rows zero; no gene above 100 * minDisp; the parametric dispersion trend failed,
                                R/core.R:885-893) -- the caller falls back to the call-by-call routines          */
};

enum { DSQ_LAYOUT_R = 0, DSQ_LAYOUT_GENE_MAJOR = 1 };
enum { DSQ_Y_INT32 = 0, DSQ_Y_FLOAT64 = 1 }; /* R INTSXP or REALSXP count matrix */

#define DSQ_MAX_P 64 /* largest number of design columns served: 1..10 by register-resident kernels (one per
                       width), 11..64 by kernels over the design zero-padded to 16, 24, 32, 48 or 64 columns (the reference
                       itself has no limit, src/DESeq2.cpp:283-465).  49..64 columns: fitDisp / fitDispGrid take rows of
                       at most 1024 samples whose working set fits a CU's LDS (else DSQ_ERR_UNSUPPORTED) */

/* ---- fitBeta ------------------------------------------------------------------
 * reference: List fitBeta(ySEXP, xSEXP, nfSEXP, alpha_hatSEXP, contrastSEXP,
 *   beta_matSEXP, lambdaSEXP, weightsSEXP, useWeightsSEXP, tolSEXP, maxitSEXP,
 *   useQRSEXP, minmuSEXP)                                   src/DESeq2.cpp:283        */
typedef struct {
    int32_t n, m, p;
    int32_t layout;         /* DSQ_LAYOUT_* for y / nf / weights / hat_diagonals / mu      */
    int64_t ld;             /* leading dimension for DSQ_LAYOUT_GENE_MAJOR (>= m)          */
    const void *y;          /* n x m counts                                                */
    int32_t y_type;         /* DSQ_Y_INT32 or DSQ_Y_FLOAT64                                */
    const double *x;        /* m x p design, column-major                                  */
    const double *nf;       /* n x m normalization factors, or (nf_is_vector) m size factors */
    int32_t nf_is_vector;   /* extension: 1 = nf points at m size factors shared by all genes */
    const double *alpha_hat;/* n                                                           */
    const double *contrast; /* p                                                           */
    const double *beta_mat; /* n x p initial beta (natural-log scale), column-major        */
    const double *lambda;   /* p ridge values (natural-log scale)                          */
    const double *weights;  /* n x m observation weights; may be NULL when useWeights == 0 */
    int32_t useWeights;
    double tol;
    int32_t maxit;          /* 0 is valid: only the post-loop block runs (R/results.R:797) */
    int32_t useQR;
    double minmu;
    /* extension: design cells.  HOST array of m labels, samples with identical rows of x carrying the same label
     * (any numbering), ncell = number of labels; NULL / 0 = not known.  With at most 32 cells the cell-collapsed
     * kernel runs.  The host-pointer entry point derives the cells from x itself when this is NULL.  */
    const int32_t *cell_of;
    int32_t ncell;
} DsqFitBetaArgs;

typedef struct {
    /* members of the reference's return list (src/DESeq2.cpp:458-464); caller-allocated */
    double *beta_mat;       /* n x p column-major                                          */
    double *beta_var_mat;   /* n x p column-major                                          */
    double *iter;           /* n (double, as the reference's NumericVector)                */
    double *hat_diagonals;  /* n x m in `layout`; may be NULL to skip the 8*m bytes/gene   */
    double *contrast_num;   /* n                                                           */
    double *contrast_denom; /* n                                                           */
    double *deviance;       /* n                                                           */
    /* extension: fitted means, so R/fitNbinomGLMs.R:180 need not redo
     * nf * exp(x beta) on the host.  mu = max(nf*exp(x beta), mu_floor); NULL = skip.   */
    double *mu;             /* n x m in `layout`                                           */
    double mu_floor;        /* 0 = unclamped (fitNbinomGLMs.R:180); minmu = core.R:763     */
} DsqFitBetaOut;

int dsq_fit_beta(const DsqFitBetaArgs *args, const DsqFitBetaOut *out);
int dsq_fit_beta_dev(const DsqFitBetaArgs *args, const DsqFitBetaOut *out, void *stream);

/* ---- fitDisp ------------------------------------------------------------------
 * reference: List fitDisp(ySEXP, xSEXP, mu_hatSEXP, log_alphaSEXP,
 *   log_alpha_prior_meanSEXP, log_alpha_prior_sigmasqSEXP, min_log_alphaSEXP,
 *   kappa_0SEXP, tolSEXP, maxitSEXP, usePriorSEXP, weightsSEXP, useWeightsSEXP,
 *   weightThresholdSEXP, useCRSEXP)                         src/DESeq2.cpp:164        */
typedef struct {
    int32_t n, m, p;
    int32_t layout;
    int64_t ld;
    const void *y;
    int32_t y_type;
    const double *x;                    /* m x p column-major                              */
    const double *mu_hat;               /* n x m                                           */
    const double *log_alpha;            /* n initial values                                */
    const double *log_alpha_prior_mean; /* n                                               */
    double log_alpha_prior_sigmasq;
    double min_log_alpha;
    double kappa_0;
    double tol;
    int32_t maxit;
    int32_t usePrior;
    const double *weights;              /* n x m; may be NULL when useWeights == 0         */
    int32_t useWeights;
    double weightThreshold;
    int32_t useCR;
    const int32_t *cell_of;             /* extension: design cells, as in DsqFitBetaArgs (HOST, may be NULL)   */
    int32_t ncell;
} DsqFitDispArgs;

typedef struct {
    /* members of the reference's return list (src/DESeq2.cpp:268-276); each n long */
    double *log_alpha;
    int32_t *iter;
    int32_t *iter_accept;
    double *last_change;
    double *initial_lp;
    double *initial_dlp;
    double *last_lp;
    double *last_dlp;
    double *last_d2lp;   /* may be NULL (device entry point): the second-derivative launch is skipped */
} DsqFitDispOut;

int dsq_fit_disp(const DsqFitDispArgs *args, const DsqFitDispOut *out);
int dsq_fit_disp_dev(const DsqFitDispArgs *args, const DsqFitDispOut *out, void *stream);

/* ---- fitDispGrid --------------------------------------------------------------
 * reference: List fitDispGrid(ySEXP, xSEXP, mu_hatSEXP, disp_gridSEXP,
 *   log_alpha_prior_meanSEXP, log_alpha_prior_sigmasqSEXP, usePriorSEXP, weightsSEXP,
 *   useWeightsSEXP, weightThresholdSEXP, useCRSEXP)         src/DESeq2.cpp:469        */
typedef struct {
    int32_t n, m, p;
    int32_t layout;
    int64_t ld;
    const void *y;
    int32_t y_type;
    const double *x;
    const double *mu_hat;
    const double *disp_grid;            /* ngrid log-alpha values (R/wrappers.R:70-72)     */
    int32_t ngrid;
    const double *log_alpha_prior_mean; /* n                                               */
    double log_alpha_prior_sigmasq;
    int32_t usePrior;
    const double *weights;
    int32_t useWeights;
    double weightThreshold;
    int32_t useCR;
    const int32_t *cell_of;             /* extension: design cells (HOST, may be NULL)                         */
    int32_t ncell;
} DsqFitDispGridArgs;

typedef struct {
    double *log_alpha;                  /* n (src/DESeq2.cpp:512)                          */
} DsqFitDispGridOut;

int dsq_fit_disp_grid(const DsqFitDispGridArgs *args, const DsqFitDispGridOut *out);
int dsq_fit_disp_grid_dev(const DsqFitDispGridArgs *args, const DsqFitDispGridOut *out, void *stream);

/* The same three routines on the gene rows [row_lo, row_lo + row_cnt) of the SAME full n x . arrays (inputs read, outputs
 * written at those rows only).  Genes are independent (src/DESeq2.cpp:194,319,492), so a caller may walk a large
 * problem range by range -- the R shim does, polling R_CheckUserInterrupt() between ranges as the reference does every
 * 100 genes (:195,320,493).  dsq_fit_*(a, o) == dsq_fit_*_rows(a, o, 0, a->n).                                     */
int dsq_fit_beta_rows(const DsqFitBetaArgs *args, const DsqFitBetaOut *out, int64_t row_lo, int64_t row_cnt);
int dsq_fit_disp_rows(const DsqFitDispArgs *args, const DsqFitDispOut *out, int64_t row_lo, int64_t row_cnt);
int dsq_fit_disp_grid_rows(const DsqFitDispGridArgs *args, const DsqFitDispGridOut *out, int64_t row_lo, int64_t row_cnt);

/* ---- extensions beyond the three .Call routines ----------------------
 * dsq_prefit_moments: what estimateDispersionsGeneEst / fitNbinomGLMs compute in R before the
 * first native call -- baseMean, baseVar, allZero (R/core.R:2138-2146), roughDispEstimate
 * (R/core.R:2422-2437) and the QR least-squares start values (R/fitNbinomGLMs.R:139-145).
 * q (m x p), a = X R^-1 (m x p) and r (p x p) come from the thin QR of the model matrix
 * (stats::qr on the host, as in the reference), all column-major.                          */
typedef struct {
    int32_t n, m, p;
    int32_t layout;
    int64_t ld;
    const void *y;
    int32_t y_type;
    const double *nf;
    int32_t nf_is_vector;
    const double *weights;   /* only enter baseMean / baseVar; may be NULL */
    int32_t useWeights;
    const double *q, *a, *r;
} DsqPrefitArgs;

typedef struct {
    double *baseMean;    /* n */
    double *baseVar;     /* n */
    int32_t *allZero;    /* n */
    double *roughDisp;   /* n */
    double *beta_init;   /* n x p column-major (natural-log scale) */
} DsqPrefitOut;

int dsq_prefit_moments(const DsqPrefitArgs *args, const DsqPrefitOut *out);
int dsq_prefit_moments_dev(const DsqPrefitArgs *args, const DsqPrefitOut *out, void *stream);

/* dsq_linear_mu: linearModelMuNormalized (R/core.R:2454-2471), mu = nf * ((y/nf) Q)(X R^-1)', the closed-form
 * fitted means estimateDispersionsGeneEst uses when the design is a set of groups (R/core.R:735-760); floored at
 * mu_floor when > 0 (:763).  Takes the q / a fields of DsqPrefitArgs (r, weights unused).  mu: n x m.      */
int dsq_linear_mu(const DsqPrefitArgs *args, double mu_floor, double *mu);
int dsq_linear_mu_dev(const DsqPrefitArgs *args, double mu_floor, double *mu, void *stream);

/* dsq_nbinom_loglike: nbinomLogLike (R/core.R:2208-2217), rowSums([w *] dnbinom(y, mu, 1/disp, log)) */
typedef struct {
    int32_t n, m;
    int32_t layout;
    int64_t ld;
    const void *y;
    int32_t y_type;
    const double *mu;        /* n x m */
    const double *disp;      /* n */
    const double *weights;
    int32_t useWeights;
} DsqLogLikeArgs;

int dsq_nbinom_loglike(const DsqLogLikeArgs *args, double *loglike);
int dsq_nbinom_loglike_dev(const DsqLogLikeArgs *args, double *loglike, void *stream);

/* dsq_intercept_fit: the closed form fitNbinomGLMs takes for an intercept-only design with the wide prior
 * (R/fitNbinomGLMs.R:99-137; nbinomLRT's reduced = ~1 reaches it for every gene): betaMatrix = log2 of the
 * [weighted] mean normalized count, mu = nf 2^beta, betaSE and hat from w = [weights] / (1/mu + alpha).
 * Outputs: beta_log2, betaSE (n); mu, hat (n x m, either may be NULL).  mu is floored at mu_floor when > 0.   */
typedef struct {
    int32_t n, m;
    int32_t layout;
    int64_t ld;
    const void *y;
    int32_t y_type;
    const double *nf;
    int32_t nf_is_vector;
    const double *weights;
    int32_t useWeights;
    const double *alpha;     /* n */
    double mu_floor;
} DsqInterceptArgs;

typedef struct {
    double *beta_log2, *betaSE;   /* n */
    double *mu, *hat;             /* n x m or NULL */
} DsqInterceptOut;

int dsq_intercept_fit(const DsqInterceptArgs *args, const DsqInterceptOut *out);
int dsq_intercept_fit_dev(const DsqInterceptArgs *args, const DsqInterceptOut *out, void *stream);

/* dsq_optim_rows: fitNbinomGLMsOptim (R/fitNbinomGLMs.R:340-407) -- the rows fitBeta left unconverged (or with NA /
 * non-positive variance) re-fitted by maximising the penalised NB log posterior over beta in [-30, 30]^p.  The
 * reference runs stats::optim(method = "L-BFGS-B") per row in R; here one wavefront per row runs a damped
 * Fisher-scoring iteration on the same objective over the same box.  The caller passes ONLY those rows
 * (n of them).  lambda: prior precisions on the log2 scale (R's `lambda`); beta_start / beta / betaSE on the log2
 * scale; conv = optim's convergence == 0; mu = nf 2^(x beta) unclamped (:386); logLike at mu clamped to minmu (:398). */
typedef struct {
    int32_t n, m, p;
    int32_t layout;
    int64_t ld;
    const void *y;
    int32_t y_type;
    const double *x;
    const double *nf;
    int32_t nf_is_vector;
    const double *alpha_hat;   /* n */
    const double *lambda;      /* p */
    const double *weights;
    int32_t useWeights;
    const double *beta_start;  /* n x p column-major */
    double minmu;
} DsqOptimArgs;

typedef struct {
    double *beta, *betaSE;     /* n x p column-major */
    int32_t *conv;             /* n */
    double *mu;                /* n x m */
    double *logLike;           /* n */
} DsqOptimOut;

int dsq_optim_rows(const DsqOptimArgs *args, const DsqOptimOut *out);

/* dsq_parametric_dispersion_fit: parametricDispersionFit (R/core.R:2166-2190), the all-gene Gamma-GLM
 * trend disp ~ asymptDisp + extraPois/mean between the two dispersion passes.  means / disps: n values
 * (the genes with dispGeneEst > 100*minDisp, R/core.R:870).  coefs: 2 doubles.  *status: 0 ok,
 * 1 = "parametric dispersion fit failed", 2 = "dispersion fit did not converge" (the caller then falls
 * back as R/core.R:885-893 does).  _dev: device pointers (coefs, status on the device too).        */
int dsq_parametric_dispersion_fit(const double *means, const double *disps, int64_t n, double *coefs, int32_t *status);
int dsq_parametric_dispersion_fit_dev(const double *means, const double *disps, int64_t n, double *coefs,
                                      int32_t *status, void *stream);

/* dsq_cooks_distance: calculateCooksDistance (R/core.R:2333-2340) with its robust method-of-moments
 * dispersion (robustMethodOfMomentsDisp :2277-2299, trimmedCellVariance :2301-2324, trimmedVariance
 * :2326-2331) and recordMaxCooks (:2349-2359), called from nbinomWaldTest (:1457-1460) and nbinomLRT
 * (:1888-1891).  `cell_of` is ALWAYS a host pointer (m design-cell ids 0..ncell-1: samples with identical
 * rows of the dispersion model matrix share a cell, nOrMoreInCell :2366-2371); p = ncol of that matrix.
 * maxCooks is NaN (R: NA) when m <= p or no cell has 3 members.                                      */
typedef struct {
    int32_t n, m, p;
    int32_t layout;
    int64_t ld;
    const void *y;
    int32_t y_type;
    const double *nf;
    int32_t nf_is_vector;
    const double *mu;        /* n x m fitted means (assays "mu") */
    const double *H;         /* n x m hat diagonals (fitBeta's hat_diagonals) */
    const int32_t *cell_of;  /* m, HOST */
    int32_t ncell;
} DsqCooksArgs;
typedef struct {
    double *cooks;       /* n x m, layout of the inputs */
    double *maxCooks;    /* n */
    double *robustDisp;  /* n (may be NULL) */
} DsqCooksOut;
int dsq_cooks_distance(const DsqCooksArgs *args, const DsqCooksOut *out);
int dsq_cooks_distance_dev(const DsqCooksArgs *args, const DsqCooksOut *out, void *stream);

/* dsq_replace_outliers: replaceOutliers (R/core.R:2069-2115).  newCounts = counts, except where
 * cooks > cooksCutoff in a `replaceable` sample: there as.integer(trimmed mean (trim) of the normalized
 * counts * nf).  replace[i] = any(cooks[i,] > cooksCutoff) (:2086).  `replaceable` is a host pointer
 * (m flags, nOrMoreInCell(modelMatrix, minReplicates)).                                              */
typedef struct {
    int32_t n, m;
    int32_t layout;
    int64_t ld;
    const void *y;
    int32_t y_type;
    const double *nf;
    int32_t nf_is_vector;
    const double *cooks;        /* n x m */
    double cooksCutoff;         /* qf(.99, p, m - p) by default (:2081) */
    double trim;                /* .2 */
    const int32_t *replaceable; /* m, HOST */
} DsqReplaceArgs;
typedef struct {
    int32_t *newCounts;  /* n x m int32, layout of the inputs */
    int32_t *replace;    /* n */
} DsqReplaceOut;
int dsq_replace_outliers(const DsqReplaceArgs *args, const DsqReplaceOut *out);
int dsq_replace_outliers_dev(const DsqReplaceArgs *args, const DsqReplaceOut *out, void *stream);

/* getAndCheckWeights (R/core.R:2697-2751) on resident gene-major weights: w_norm = w / rowmax (:2702), w_floor =
 * pmax(w_norm, 1e-6) (:702), weightsFail[i] = 1 when the weights of gene i leave a degenerate design (the two per-gene
 * qr() rank tests of :2711-2722, full-rank model matrices, p <= DSQ_MAX_P), *any_negative |= 1 when a weight is negative
 * (the caller zeroes it first).  All device pointers; x: m x p column-major.                                      */
int dsq_weights_prep_dev(const double *weights_raw, const double *x, int32_t n, int32_t m, int32_t p, int64_t ld,
                         double weightThreshold, double *w_norm, double *w_floor, int32_t *weightsFail,
                         int32_t *any_negative, void *stream);
/* momentsDispEstimate's xim for a resident normalization-factor matrix (R/core.R:2440-2444): mean over the samples of
 * 1 / colMeans(nf), every column summed down the genes in gene order.  scratch_m: m doubles; *out on the device.  */
int dsq_xim_dev(const double *nf, int32_t n, int32_t m, int64_t ld, double *scratch_m, double *out, void *stream);

/* ---- layout helpers (device pointers, async on stream) --------------------------
 * R layout (column-major n x m) <-> gene-major (row-major, leading dimension ld).   */
int dsq_to_gene_major_f64(const double *src_r, double *dst_gm, int32_t n, int32_t m, int64_t ld, void *stream);
int dsq_to_gene_major_i32(const int32_t *src_r, int32_t *dst_gm, int32_t n, int32_t m, int64_t ld, void *stream);
/* REALSXP counts -> int32 gene-major; *bad (device int32) is set non-zero if any value
 * is negative, non-finite or non-integer                                             */
int dsq_counts_f64_to_gene_major_i32(const double *src_r, int32_t *dst_gm, int32_t n, int32_t m,
                                     int64_t ld, int32_t *bad, void *stream);
int dsq_from_gene_major_f64(const double *src_gm, double *dst_r, int32_t n, int32_t m, int64_t ld, void *stream);

/* ---- misc ---------------------------------------------------------------------- */
int dsq_version(void);
const char *dsq_last_error(void);        /* thread-local message of the last failing call  */
int dsq_device_count(void);              /* number of visible HIP devices (0 if none)      */
int dsq_set_device(int device);          /* device used by subsequent calls on this thread */
int dsq_release_workspace(void);         /* free the cached device workspaces              */

/* Kernel timing for bench.py's roofline: when enabled, every dsq_fit_*_dev call brackets its
 * fit kernel launch with HIP events on the launch stream; dsq_profile_last_ms() waits for the
 * most recent one and returns its duration in milliseconds (negative if none was recorded).  */
int dsq_profile_enable(int on);
double dsq_profile_last_ms(void);

/* Parity hook for tests: evaluate one scalar primitive of the device math library on
 * the GPU (op: 0 exp, 1 log, 2 log1p, 3 lgamma, 4 digamma, 5 trigamma, 6 stirlerr,
 * 7 bd0(a,b), 8 dnbinom_mu_log(a=x, b=size, c=mu)).  HOST pointers, n elements.       */
int dsq_test_math(int op, const double *a, const double *b, const double *c, double *out, int64_t n);

/* ---- dsq_deseq_dev: the whole DESeq() chain, device-driven ---------------------------
 * estimateDispersionsGeneEst (R/core.R:657-860) -> estimateDispersionsFit (:864-939, parametric) +
 * estimateDispersionsPriorVar (:1135-1208) -> estimateDispersionsMAP (:943-1131) -> nbinomWaldTest (:1332-1565)
 * or nbinomLRT against ~1 (:1787-2012) -> replaceOutliers / refitWithoutOutliers (:2069-2115, :2484-2563) on
 * matrices resident in HBM in the gene-major layout.  The per-gene decision rules R applies between the native
 * calls (clamps, accept / convergence / refit rules, dispOutlier, betaConv ...) run as small kernels, the rows a
 * rule sends to fitDispGrid or to the outlier refit are compacted on the device and fitted by row-listed launches
 * of the same kernels, so a phase needs no host round trip.  Rows that need R's host-side fallback (the
 * L-BFGS-B rows of fitNbinomGLMsOptim) are only FLAGGED (optim_* outputs): the caller re-does them through the
 * per-call entry points.  Phases (bit mask), each asynchronous on `stream`:
 *   DSQ_PH_GENE_EST   counts -> baseMean .. dispGeneEst, mu-hat
 *   DSQ_PH_TREND      dispersion trend (fitType: parametric / mean) + prior variance over (trend_mean, trend_disp) [n_trend on the device] or,
 *                     when those are NULL, over this call's own genes (multi-GPU: the gathered vectors)
 *   DSQ_PH_MAP_TEST   dispFit, MAP dispersions, final GLM fit [+ the reduced-model fit], Wald statistics / logLik pair
 *   DSQ_PH_OUTLIERS   Cook's distances, replaceOutliers, refit of the replaced rows, maxCooks
 *   DSQ_PH_FINISH     (gene-sharding callers only, see below)
 * status[] (int32, device): see DSQ_ST_*; scalars[] (double, device): see DSQ_SC_*.                            */
#define DSQ_PH_GENE_EST 1
#define DSQ_PH_TREND    2
#define DSQ_PH_MAP_TEST 4
#define DSQ_PH_OUTLIERS 8
/* DSQ_PH_OUTLIERS in two calls, for a caller that brings its own dispersion trend (dispFit_in) AND wants the replaced rows
 * refitted on the chain (refitWithoutOutliers evaluates dispersionFunction(object) at the NEW means, R/core.R:2512):
 *   DSQ_PH_OUTLIERS_DETECT  Cook's distances, replaceOutliers, baseMean / baseVar / allZero of the replaced rows (:2488-2491)
 *   -- the caller reads `replace`, `allZero`, `baseMean`, writes its trend at the rows with replace = 1, allZero = 0 into dispFit_in --
 *   DSQ_PH_OUTLIERS_REFIT   the refit of those rows and the closing steps (:2496-2546)                                   */
#define DSQ_PH_OUTLIERS_DETECT 64
#define DSQ_PH_OUTLIERS_REFIT 128
#define DSQ_PH_PRIOR    32   /* betaPrior = TRUE: the second pass of fitGLMsWithPrior (R/fitNbinomGLMs.R:311-325) with
                                lambda_prior = 1 / betaPriorVar; DSQ_PH_MAP_TEST has run the MLE pass (:256-260) and the
                                caller has turned mle_beta into the prior variance (estimateBetaPriorVar, R/core.R:1601-1689:
                                an all-gene weighted quantile, host code).  Runs before DSQ_PH_OUTLIERS.                */
#define DSQ_PH_FINISH   16   /* the two closing steps of refitWithoutOutliers that depend on whether ANY row of the whole
                                analysis was refitted (R/core.R:2496, 2535-2546: NA results on rows that became all zero,
                                maxCooks): part of DSQ_PH_OUTLIERS unless defer_finish is set -- a caller that shards the
                                genes sets it, adds up N_REFIT over its shards and runs this phase with the total       */

enum { DSQ_ST_N_NONZERO = 0, DSQ_ST_N_GRID_GENEEST, DSQ_ST_N_TREND, DSQ_ST_TREND_STATUS, DSQ_ST_N_ABOVE_MIN,
       DSQ_ST_N_GRID_MAP, DSQ_ST_N_OPTIM_GENEEST, DSQ_ST_N_OPTIM_TEST, DSQ_ST_N_REPLACE, DSQ_ST_N_REFIT,
       DSQ_ST_N_GRID_GENEEST_REFIT, DSQ_ST_N_GRID_MAP_REFIT, DSQ_ST_N_OPTIM_GENEEST_REFIT, DSQ_ST_N_OPTIM_TEST_REFIT,
       DSQ_ST_COUNT = 16 };        /* (14, 15: counters of the chain's own) */
/* estimateDispersionsFit's fitType (R/core.R:864-939).  "local" needs locfit (an R package, not in the reference's
 * tree): a caller that wants the reference's automatic substitution asks for DSQ_FIT_PARAMETRIC, gets DSQ_ERR_FIT when
 * the trend does not fit and takes its own route; DSQ_FIT_PARAMETRIC_OR_MEAN substitutes the mean on the device.   */
#define DSQ_FIT_PARAMETRIC          0
#define DSQ_FIT_MEAN                1    /* mean(dispGeneEst[dispGeneEst > 10 minDisp], trim = 0.001), R/core.R:894-899 */
#define DSQ_FIT_PARAMETRIC_OR_MEAN  2
#define DSQ_FIT_GIVEN               3    /* (reported in DSQ_SC_FIT_USED only) the caller's trend values: dispFit_in        */
enum { DSQ_SC_COEF0 = 0, DSQ_SC_COEF1, DSQ_SC_VAR_LOG_DISP, DSQ_SC_DISP_PRIOR_VAR,
       DSQ_SC_FIT_USED,            /* the trend the analysis ended up with: DSQ_FIT_PARAMETRIC (0.0) or DSQ_FIT_MEAN (1.0:
                                      COEF0 is the mean, COEF1 zero)                                               */
       DSQ_SC_COUNT = 8 };

typedef struct {
    int32_t n, m, p;
    int64_t ld;
    int32_t phases;
    const int32_t *y;              /* n x ld gene-major counts                                                  */
    const double *nf;              /* n x ld normalization factors, or m size factors (nf_is_vector)            */
    int32_t nf_is_vector;
    int32_t useWeights;
    const double *weights_raw;     /* assays[["weights"]] as given: enters baseMean / baseVar (R/core.R:2140)   */
    const double *weights_norm;    /* / row max (R/core.R:2702): GLM fits, MAP, logLik                           */
    const double *weights_floor;   /* pmax(weights_norm, 1e-6) (R/core.R:702): gene-wise dispersion search       */
    const int32_t *force_zero;     /* n flags or NULL: rows treated as all-zero (weightsFail, R/core.R:2737)     */
    const double *x;               /* m x p design, column-major                                                 */
    const double *q, *a, *r;       /* thin QR of the design: Q, X R^-1 (m x p), R (p x p), column-major          */
    double xim;                    /* momentsDispEstimate's mean(1 / sizeFactors) (R/core.R:2440-2444); with a
                                      normalization-factor MATRIX (nf_is_vector = 0) it is not read: the chain takes
                                      mean(1 / colMeans(nf)) over the rows that are not all zero itself              */
    int32_t linearMu;              /* R/core.R:735-742                                                           */
    double minDisp, kappa_0, dispTol, weightThreshold, outlierSD, betaTol, minmu;
    int32_t maxit, useCR, useQR, betaMaxit;
    const double *disp_grid;       /* ngrid log-alpha grid points of fitDispGridWrapper (R/wrappers.R:70-72)     */
    int32_t ngrid;
    double expVarLogDisp;          /* trigamma((m - p) / 2) (R/core.R:1196); ignored when m <= p                  */
    const double *trend_mean, *trend_disp;   /* device vectors of n_trend values, or NULL                        */
    int32_t n_trend;
    const double *lambda;          /* HOST: p ridge values on the natural-log scale (R/fitNbinomGLMs.R:162)      */
    double min_log_alpha;          /* log(minDisp / 10) (R/core.R:775)                                           */
    void *workspace;               /* device, dsq_deseq_workspace_bytes(...): holds the row lists and counters of
                                      the analysis between phases -- the same buffer for all its phases           */
    int64_t workspace_bytes;
    int32_t test;                  /* 0 Wald, 1 LRT (reduced = ~1 unless x_red is given)                         */
    /* outlier phase (all host arrays; R/core.R:2081,2101,2366-2371) */
    const int32_t *cell_of;        /* HOST: design cell of each sample                                           */
    int32_t ncell;
    const int32_t *replaceable;    /* HOST: m flags nOrMoreInCell(x, minReplicatesForReplace)                    */
    double cooksCutoff, trim;
    int32_t do_replace;            /* 0: Cook's distances only                                                   */
    /* nbinomLRT against a reduced model that is not ~1 (R/core.R:1856-1868): its m x p_red model matrix, the thin QR
     * of it (start values, R/fitNbinomGLMs.R:139-145) and its design cells; all NULL / 0 = the intercept-only closed form */
    const double *x_red, *q_red, *a_red, *r_red;   /* device, column-major                                      */
    int32_t p_red;
    const int32_t *cell_of_red;    /* HOST                                                                       */
    int32_t ncell_red;
    int32_t defer_finish;          /* see DSQ_PH_FINISH                                                          */
    const int32_t *n_refit_global; /* device int32 for DSQ_PH_FINISH: refitted rows over ALL shards (NULL: this call's) */
    /* nbinomWaldTest(betaPrior = TRUE) (R/core.R:1416-1432, R/fitNbinomGLMs.R:242-337), Wald only: the model matrix of
     * the prior pass -- the design itself (modelMatrixType "standard") or the expanded one (R/expanded.R:1-18, rank
     * deficient: start values of :146-155) -- with the same design cells as x; p, p_prior <= 10; the workspace is asked
     * for with max(p, p_prior) columns.  beta / betaSE / stat / pvalue are then n x p_prior.                            */
    int32_t betaPrior;
    const double *x_prior;         /* device, m x p_prior column-major                                           */
    int32_t p_prior, prior_expanded, prior_intercept;   /* expanded: rank-deficient start values; first column all ones */
    const double *lambda_prior;    /* HOST, p_prior: 1 / betaPriorVar / log(2)^2; read by DSQ_PH_PRIOR / _OUTLIERS */
    int32_t fitType;               /* DSQ_FIT_*: read by DSQ_PH_TREND                                             */
    /* the caller's dispersion trend -- what R has after fitType = "local" (locfit, R/core.R:889-893) or
     * `dispersionFunction(dds) <- f` (R/methods.R:142-190) --, evaluated by the caller at the baseMean of every gene (the
     * DSQ_PH_GENE_EST phase gives baseMean and dispGeneEst): device, n values.  DSQ_PH_TREND then fits nothing and takes
     * varLogDispEsts / the prior variance from the residuals against it (against trend_fit_in, n_trend values aligned with
     * trend_mean / trend_disp, when those are given); DSQ_PH_MAP_TEST takes dispFit from it.  The refit of replaced rows
     * needs the trend at means the caller has not seen: either do_replace = 0 (the caller refits, R/core.R:2484-2563) or the
     * outlier phase in its two halves, DSQ_PH_OUTLIERS_DETECT / _REFIT above.                                          */
    const double *dispFit_in, *trend_fit_in;
    /* estimateDispersionsMAP(dispPriorVar = x) (R/core.R:970,989-994): > 0 = the caller's prior variance, taken instead of
     * the estimate (varLogDispEsts is still computed: the dispOutlier rule reads it).  The way residual df <= 3 runs on the
     * chain: R's estimate there is a seeded Monte-Carlo match (R/core.R:1155-1190, R's RNG + loess), the caller's job.   */
    double dispPriorVar_in;
} DsqDeseqArgs;

typedef struct {
    /* per-gene results, device, caller-allocated; rows that are all-zero come back NaN / -1 */
    double *baseMean, *baseVar;
    int32_t *allZero;
    double *dispGeneEst;
    int32_t *dispGeneIter;
    double *dispFit, *dispMAP, *dispersion;
    int32_t *dispIter, *dispOutlier;
    double *beta, *betaSE, *stat, *pvalue;      /* n x p column-major, log2 scale; stat / pvalue Wald only        */
    int32_t *betaConv;
    double *betaIter, *logLike, *logLikeReduced, *maxCooks;
    int32_t *replace;
    int32_t *optim_geneest, *optim_test;        /* n flags: rows R hands to the L-BFGS-B fallback                 */
    /* n x ld gene-major */
    double *mu_hat;                             /* clamped fitted means of the gene-wise fit (assays mu of GeneEst) */
    double *mu, *H, *cooks;
    int32_t *replaceCounts;
    int32_t *status;                            /* DSQ_ST_COUNT */
    double *scalars;                            /* DSQ_SC_COUNT */
    double *mle_beta;                           /* betaPrior: n x p, the MLE coefficients (mcols MLE_*), log2 scale */
} DsqDeseqOut;

int dsq_deseq_dev(const DsqDeseqArgs *args, const DsqDeseqOut *out, void *stream);
int64_t dsq_deseq_workspace_bytes(int32_t n, int32_t m, int32_t p, int32_t n_trend);

/* ---- dsq_deseq: DESeq() behind ONE host-pointer call ------------------------------------------------------------
 * What an R session binds as .Call("_DESeq2_mi355x_DESeq", ...) in place of the body of DESeq() between
 * estimateSizeFactors and the final bookkeeping (R/core.R:388-426: estimateDispersions -> nbinomWaldTest / nbinomLRT
 * -> refitWithoutOutliers): every array is a HOST pointer in R's layout (what INTEGER() / REAL() give), the call
 * uploads the count matrix ONCE through pinned staging, runs the device-driven chain of dsq_deseq_dev (all four
 * phases, the optim-fallback rows included, no host decision in between), and downloads the per-gene columns; the
 * n x m assays (mu, H, cooks, replaceCounts) come down only when their output pointer is non-NULL.  Like the three
 * classic routines it cuts the genes into the contiguous ranges of R/parallel.R:10, one per visible device
 * (DSQ_HOST_DEVICES / DSQ_HOST_SHARDS as there); the ranges exchange the two n-vectors of the dispersion trend
 * through host memory, as DESeqParallel does (R/parallel.R:27-40).
 * Covers what the fused chain covers: parametric trend, fitType "mean" or the caller's own trend (geneEstOnly / dispFit), Wald (also with betaPrior = TRUE) or LRT (any nested reduced model), p <= DSQ_MAX_P
 * (wide designs, 10 < p, included: observation weights, reduced models of any width < p and the beta-prior pass since round 5),
 * m - p > 3, size factors or a normalization-factor matrix, observation weights; anything else returns
 * DSQ_ERR_UNSUPPORTED and the caller keeps to the three classic routines.  The design-only quantities R has functions
 * for are passed in: qr.Q / qr.R of the model matrix (R/fitNbinomGLMs.R:139-143), qf(.99, p, m - p) (R/core.R:2081),
 * trigamma((m - p) / 2) (R/core.R:1196).
 * NA convention of the outputs: NaN in double columns, -1 in int32 columns (allZero rows; R/core.R:2534-2536).     */
typedef struct {
    int32_t n, m, p;
    const void *counts;            /* n x m column-major                                                          */
    int32_t y_type;                /* DSQ_Y_INT32 (counts(dds)) or DSQ_Y_FLOAT64                                  */
    const double *x;               /* m x p model matrix, column-major, full rank                                 */
    const double *sizeFactors;     /* m, or NULL when normalizationFactors is given                               */
    const double *normalizationFactors;  /* n x m column-major (normalizationFactors(object)), or NULL.  With a matrix the
                                      genes stay in ONE range (momentsDispEstimate averages the factors over all non-zero
                                      rows -- and, in the outlier refit, over the refitted rows, R/core.R:2440-2444 --
                                      sums a split could only reproduce in another order)                            */
    const double *weights;         /* n x m column-major (assays(object)[["weights"]]), or NULL                   */
    const double *q, *r;           /* qr.Q(qr(x)) (m x p) and qr.R(qr(x)) (p x p), column-major                   */
    const double *xrinv;           /* x %*% solve(R) (m x p), or NULL: computed here by back substitution         */
    int32_t test;                  /* 0 Wald, 1 LRT (reduced = ~1 unless x_reduced is given)                      */
    const double *x_reduced;       /* LRT: the reduced model matrix (m x p_reduced, nested in x, full rank) with its  */
    const double *q_reduced, *r_reduced;   /* qr.Q / qr.R, or all NULL for reduced = ~1 (R/core.R:1856-1868)          */
    int32_t p_reduced;
    double minReplicatesForReplace;/* 7 by default; +Inf switches replaceOutliers / the refit off                 */
    double cooksCutoff;            /* qf(.99, p, m - p)                                                           */
    double expVarLogDisp;          /* trigamma((m - p) / 2)                                                       */
    double betaTol, minmu;         /* nbinomWaldTest / nbinomLRT: 1e-8, 0.5                                       */
    int32_t maxit, useQR;          /* 100, TRUE                                                                   */
    int32_t disp_maxit, useCR;     /* estimateDispersions: 100, TRUE                                              */
    const double *disp_grid;       /* fitDispGridWrapper's seq(log(1e-8), log(max(10, m)), length = 20) (R/wrappers.R:70-72) */
    int32_t ngrid;                 /*   as the caller's own log() gives it; NULL / 0: computed here                */
    /* nbinomWaldTest(betaPrior = TRUE) (R/core.R:1416-1432 -> fitGLMsWithPrior, R/fitNbinomGLMs.R:242-337), Wald only:
     * the MLE pass on x, the all-gene prior variance (estimateBetaPriorVar, R/core.R:1601-1689: dsq_beta_prior_var
     * below, run inside the call on the MLE coefficients of ALL gene ranges), then the pass with lambda = 1 /
     * betaPriorVar on the standard model matrix (x_prior NULL) or on the expanded one (R/expanded.R:1-18).  beta,
     * betaSE, stat and pvalue are then n x p_prior.  The refit of the replaced rows reuses the prior variance
     * (R/core.R:2521-2527).                                                                                        */
    int32_t betaPrior;
    const double *x_prior;         /* expanded model matrix, m x p_prior column-major (same design cells as x), or NULL  */
    int32_t p_prior;               /* columns of x_prior (ignored when x_prior is NULL: p)                         */
    const int32_t *coef_factor;    /* p: what each column of x is -- 0 the intercept, f >= 1 an indicator of a level of
                                      design factor f, -1 anything else (numeric covariate, interaction)            */
    const int32_t *prior_coef_factor;  /* p_prior, the same for the columns of x_prior (expanded only)               */
    const int32_t *prior_coef_src;     /* p_prior: for the -1 columns of x_prior the column of x with the same name   */
    const double *betaPriorVar;    /* optional, p_prior values: the caller's prior variance (nbinomWaldTest's argument);
                                      NULL = estimated                                                              */
    int32_t fitType;               /* DSQ_FIT_PARAMETRIC (0, the default of DESeq()), DSQ_FIT_MEAN, DSQ_FIT_PARAMETRIC_OR_MEAN;
                                      dispersionFunction[DSQ_SC_FIT_USED] says which trend the results carry          */
    /* a trend the library does not fit -- fitType = "local" (locfit: R code, R/core.R:889-893) or `dispersionFunction<-`
     * (R/methods.R:142-190) -- in two calls: geneEstOnly = 1 runs estimateDispersionsGeneEst only (baseMean, baseVar,
     * allZero, dispGeneEst, dispGeneIter come back; every other column NA); the caller fits its trend and calls again with
     * dispFit = its values at baseMean (host, n; rows that are all zero: anything).  Count outliers are then flagged by
     * Cook's distance as always, but NOT replaced: the refit needs the trend at the new means of the replaced rows, so the
     * caller runs refitWithoutOutliers itself on the (few) rows concerned (R/core.R:2484-2563, unchanged code).        */
    const double *dispFit;
    int32_t geneEstOnly;
    double dispPriorVar;           /* > 0: estimateDispersionsMAP(dispPriorVar = x), R/core.R:989-994 -- required when m - p <= 3
                                      (R's estimate there is a seeded Monte-Carlo match over the residuals of the trend,
                                      R/core.R:1155-1190: after the geneEstOnly call the caller has what it needs)       */
} DsqDeseqHostArgs;

typedef struct {
    /* mcols(dds): n each (beta .. pvalue: n x p column-major, log2 scale; stat / pvalue Wald only, else NULL)     */
    double *baseMean, *baseVar;
    int32_t *allZero;
    double *dispGeneEst;
    int32_t *dispGeneIter;
    double *dispFit, *dispMAP, *dispersion;
    int32_t *dispIter, *dispOutlier;
    double *beta, *betaSE, *stat, *pvalue;
    int32_t *betaConv;
    double *betaIter, *logLike, *logLikeReduced /* LRT only, else NULL */, *maxCooks;
    int32_t *replace;              /* NA (-1) on rows that were all zero from the start                           */
    int32_t *weightsFail;          /* optional (NULL): rows whose weights leave a degenerate design, treated as all zero
                                      (getAndCheckWeights, R/core.R:2736-2747)                                    */
    /* assays, n x m column-major, each optional (NULL = stays on the device)                                      */
    double *mu, *H, *cooks;
    int32_t *replaceCounts;
    /* dispersionFunction(dds), indexed by DSQ_SC_*: coefficients asymptDisp / extraPois (fitType "mean": the mean, 0),
     * varLogDispEsts, dispPriorVar, the fit type used                                                             */
    double dispersionFunction[8];
    int32_t status[16];            /* DSQ_ST_*                                                                    */
    /* betaPrior = TRUE: the prior variance used (attr(object, "betaPriorVar")) and, optionally, the n x p MLE
     * coefficients (mcols MLE_*, log2 scale)                                                                      */
    double betaPriorVar[DSQ_MAX_P];
    double *mle_beta;
} DsqDeseqHostOut;

int dsq_deseq(const DsqDeseqHostArgs *args, DsqDeseqHostOut *out);

/* estimateBetaPriorVar (R/core.R:1601-1689, betaPriorMethod = "weighted", upperQuantile = 0.05) on HOST arrays: the
 * n x p MLE coefficients (log2 scale, column-major), baseMean, dispFit, the all-zero flags of the rows; the column
 * coding of DsqDeseqHostArgs.  No device work (an all-gene step on n-vectors like the dispersion trend; a stable radix
 * sort and sequential sums).  betaPriorVar: p values, or p_prior for the expanded model matrix.                    */
typedef struct {
    int32_t n, p;
    const double *mle_beta, *baseMean, *dispFit;
    const int32_t *allZero;
    const int32_t *coef_factor;
    int32_t expanded, p_prior;
    const int32_t *prior_coef_factor, *prior_coef_src;
    double upperQuantile;
} DsqBetaPriorArgs;
int dsq_beta_prior_var(const DsqBetaPriorArgs *args, double *betaPriorVar);

/* kernel timings of the calls since dsq_profile_enable(1): one entry per bracketed launch */
int dsq_profile_count(void);
int dsq_profile_get(int i, char *name, int cap, int32_t *genes, double *ms);

#ifdef __cplusplus
}
#endif
#endif /* DESEQ2_MI355X_H */
