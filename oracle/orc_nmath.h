/*
 * ORACLE (test infrastructure only -- never linked into, imported by or executed
 * from the product path; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may use it).
 *
 * Scalar math used by the CPU restatement of DESeq2's per-gene NB-GLM kernels.
 *
 * The reference (src/DESeq2.cpp) gets these from two dependencies that are NOT
 * vendored under /root/reference:
 *   - R's nmath (libR; version unpinned by DESCRIPTION): Rf_lgammafn, Rf_digamma,
 *     Rf_trigamma, Rf_dnbinom_mu, R_pow_di       (call sites DESeq2.cpp:50-58,
 *     90-96, 139-145, 369-371, 411-413)
 *   - libm exp/log via Rcpp sugar / Armadillo     (DESeq2.cpp:34,53,55,324,349...)
 * They are restated here from the published algorithms (see each function), with
 * every floating-point operation spelled out (no FMA contraction; explicit fma()
 * where one is intended) so that the HIP kernels can reproduce the SAME bits.
 * Accuracy of every primitive is pinned against mpmath in tests/test_oracle_math.py.
 */
#ifndef ORC_NMATH_H
#define ORC_NMATH_H

#ifdef __cplusplus
extern "C" {
#endif

double orc_exp(double x);
double orc_log(double x);
double orc_log1p(double x);
double orc_lgamma(double x);   /* domain x > 0 (the only one the path uses) */
double orc_digamma(double x);  /* domain x > 0 */
double orc_trigamma(double x); /* domain x > 0 */
double orc_stirlerr(double n);
double orc_bd0(double x, double np);
double orc_dnbinom_mu_log(double x, double size, double mu);
double orc_pnorm_upper2(double z);  /* 2 * pnorm(|z|, lower.tail = FALSE) */

/* vector helpers for the ctypes tests: op selects the function */
void orc_vec_unary(int op, const double *in, double *out, long n);
void orc_vec_dnbinom_mu_log(const double *x, const double *size, const double *mu,
                            double *out, long n);

#ifdef __cplusplus
}
#endif
#endif
