/* tests/r_stub: DECLARATIONS ONLY of the R API entry points deseq2_amd/csrc/r_shim.c uses, so that the shim can be
 * syntax-checked in an image without R (tests/test_capi_cpu.py).  Not R headers, not linked, not shipped. */
#include <stddef.h>
typedef struct SEXPREC *SEXP;
typedef int Rboolean;
#define TRUE 1
#define FALSE 0
#define INTSXP 13
#define REALSXP 14
#define LGLSXP 10
#define VECSXP 19
#define STRSXP 16
extern SEXP R_NamesSymbol; extern SEXP R_NilValue; extern double R_NaReal; extern int R_NaInt;
#define NA_REAL R_NaReal
#define ISNAN(x) ((x)!=(x))
int TYPEOF(SEXP); int *INTEGER(SEXP); double *REAL(SEXP); int *LOGICAL(SEXP);
double Rf_asReal(SEXP); int Rf_asInteger(SEXP); int Rf_asLogical(SEXP);
SEXP Rf_coerceVector(SEXP, int); SEXP Rf_allocVector(int, long); SEXP Rf_allocMatrix(int, int, int);
SEXP Rf_protect(SEXP); void Rf_unprotect(int); SEXP Rf_mkChar(const char*); void SET_VECTOR_ELT(SEXP,long,SEXP); void SET_STRING_ELT(SEXP,long,SEXP);
SEXP Rf_setAttrib(SEXP,SEXP,SEXP); int Rf_nrows(SEXP); int Rf_ncols(SEXP); int Rf_length(SEXP); int Rf_isMatrix(SEXP);
void Rf_error(const char*, ...); char *R_alloc(size_t, int);
#define PROTECT(x) Rf_protect(x)
#define UNPROTECT(n) Rf_unprotect(n)
