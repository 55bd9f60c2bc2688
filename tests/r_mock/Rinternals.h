/* tests/r_mock: a MOCK of the slice of R's C API that deseq2_amd/csrc/r_shim.c uses -- enough runtime to EXECUTE the
 * .Call shim in an image without R (tests/test_r_shim.py; VERDICT r4 "next" #4).  Not R's headers, not R's behaviour
 * beyond what is stated here, test infrastructure only (never linked into the product):
 *   SEXP          a heap record {type, length, dim (or none), data, names}; every allocation is tracked and released by
 *                 rmock_reset() -- there is no collector, PROTECT / UNPROTECT only keep a depth counter that the
 *                 harness checks for balance after every successful call;
 *   Rf_error      formats the message and longjmps to the harness (rmock_call), like R's error does out of .Call;
 *   R_alloc       tracked transient memory, released when the call returns or errors;
 *   NA            NA_REAL is R's NaN payload 1954, NA_INTEGER / NA_LOGICAL are INT_MIN. */
#ifndef DSQ_RMOCK_RINTERNALS_H
#define DSQ_RMOCK_RINTERNALS_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef struct SEXPREC *SEXP;
typedef int Rboolean;
#ifndef TRUE
#define TRUE 1
#define FALSE 0
#endif
#define NILSXP 0
#define LGLSXP 10
#define INTSXP 13
#define REALSXP 14
#define STRSXP 16
#define VECSXP 19
#define CHARSXP 9
typedef ptrdiff_t R_xlen_t;
extern SEXP R_NamesSymbol, R_NilValue, R_DimSymbol;
extern double R_NaReal;
extern int R_NaInt;
#define NA_REAL R_NaReal
#define NA_INTEGER R_NaInt
#define NA_LOGICAL R_NaInt
#define ISNAN(x) ((x) != (x))
int TYPEOF(SEXP);
int *INTEGER(SEXP);
double *REAL(SEXP);
int *LOGICAL(SEXP);
double Rf_asReal(SEXP);
int Rf_asInteger(SEXP);
int Rf_asLogical(SEXP);
SEXP Rf_coerceVector(SEXP, int);
SEXP Rf_allocVector(int, R_xlen_t);
SEXP Rf_allocMatrix(int, int, int);
SEXP Rf_protect(SEXP);
void Rf_unprotect(int);
SEXP Rf_mkChar(const char *);
SEXP SET_VECTOR_ELT(SEXP, R_xlen_t, SEXP);
void SET_STRING_ELT(SEXP, R_xlen_t, SEXP);
SEXP Rf_setAttrib(SEXP, SEXP, SEXP);
int Rf_nrows(SEXP);
int Rf_ncols(SEXP);
int Rf_length(SEXP);
int Rf_isMatrix(SEXP);
void Rf_error(const char *, ...) __attribute__((noreturn, format(printf, 1, 2)));
char *R_alloc(size_t, int);
#define PROTECT(x) Rf_protect(x)
#define UNPROTECT(n) Rf_unprotect(n)
#ifdef __cplusplus
}
#endif
#endif
