/* oracle/ref_glue.cpp -- TEST INFRASTRUCTURE.  Compiles the reference's own src/DESeq2.cpp (included by
 * path from /root/reference via -DREF_SRC=..., never copied) against the stand-in headers in oracle/shim/
 * and exports its three entry points (fitDisp :164, fitBeta :283, fitDispGrid :469) with plain-pointer
 * signatures for ctypes.  Output: oracle/_ref/libdeseq2_ref.so (git-ignored).  Matrices column-major. */
#include REF_SRC

namespace {
SEXP mk(const double *p, int nrow, int ncol) {
    SEXP s = shim_alloc();
    s->nrow = nrow; s->ncol = ncol;
    s->d.assign(p, p + (size_t)nrow * (ncol ? ncol : 1));
    return s;
}
SEXP sc(double v) { return mk(&v, 1, 0); }
void out(const Rcpp::List &l, const char *name, double *dst) {
    if (!dst) return;
    const SexpRec &r = l.get(name);
    if (r.is_int) for (size_t k = 0; k < r.iv.size(); k++) dst[k] = r.iv[k];
    else std::copy(r.d.begin(), r.d.end(), dst);
}
}  // namespace

extern "C" {

int ref_fit_disp(int n, int m, int p, const double *y, const double *x, const double *mu_hat, const double *log_alpha,
                 const double *prior_mean, double prior_sigmasq, double min_log_alpha, double kappa_0, double tol,
                 int maxit, int usePrior, const double *weights, int useWeights, double weightThreshold, int useCR,
                 double *o_log_alpha, double *o_iter, double *o_iter_accept, double *o_last_change,
                 double *o_initial_lp, double *o_initial_dlp, double *o_last_lp, double *o_last_dlp,
                 double *o_last_d2lp) {
    Rcpp::List r = fitDisp(mk(y, n, m), mk(x, m, p), mk(mu_hat, n, m), mk(log_alpha, n, 0), mk(prior_mean, n, 0),
                           sc(prior_sigmasq), sc(min_log_alpha), sc(kappa_0), sc(tol), sc(maxit), sc(usePrior),
                           mk(weights, n, m), sc(useWeights), sc(weightThreshold), sc(useCR));
    out(r, "log_alpha", o_log_alpha); out(r, "iter", o_iter); out(r, "iter_accept", o_iter_accept);
    out(r, "last_change", o_last_change); out(r, "initial_lp", o_initial_lp); out(r, "initial_dlp", o_initial_dlp);
    out(r, "last_lp", o_last_lp); out(r, "last_dlp", o_last_dlp); out(r, "last_d2lp", o_last_d2lp);
    return 0;
}

int ref_fit_beta(int n, int m, int p, const double *y, const double *x, const double *nf, const double *alpha_hat,
                 const double *contrast, const double *beta_mat, const double *lambda, const double *weights,
                 int useWeights, double tol, int maxit, int useQR, double minmu,
                 double *o_beta_mat, double *o_beta_var_mat, double *o_iter, double *o_hat_diagonals,
                 double *o_contrast_num, double *o_contrast_denom, double *o_deviance) {
    Rcpp::List r = fitBeta(mk(y, n, m), mk(x, m, p), mk(nf, n, m), mk(alpha_hat, n, 0), mk(contrast, p, 0),
                           mk(beta_mat, n, p), mk(lambda, p, 0), mk(weights, n, m), sc(useWeights), sc(tol), sc(maxit),
                           sc(useQR), sc(minmu));
    out(r, "beta_mat", o_beta_mat); out(r, "beta_var_mat", o_beta_var_mat); out(r, "iter", o_iter);
    out(r, "hat_diagonals", o_hat_diagonals); out(r, "contrast_num", o_contrast_num);
    out(r, "contrast_denom", o_contrast_denom); out(r, "deviance", o_deviance);
    return 0;
}

int ref_fit_disp_grid(int n, int m, int p, const double *y, const double *x, const double *mu_hat,
                      const double *disp_grid, int ngrid, const double *prior_mean, double prior_sigmasq, int usePrior,
                      const double *weights, int useWeights, double weightThreshold, int useCR, double *o_log_alpha) {
    Rcpp::List r = fitDispGrid(mk(y, n, m), mk(x, m, p), mk(mu_hat, n, m), mk(disp_grid, ngrid, 0), mk(prior_mean, n, 0),
                               sc(prior_sigmasq), sc(usePrior), mk(weights, n, m), sc(useWeights), sc(weightThreshold),
                               sc(useCR));
    out(r, "log_alpha", o_log_alpha);
    return 0;
}
}
