/* tests/r_mock/r_mock.c -- the mock R runtime behind tests/r_mock/Rinternals.h (test infrastructure; see that header).
 * Built together with deseq2_amd/csrc/r_shim.c into tests/r_mock/libdsq_rshim_test.so by tests/r_mock/Makefile. */
#include <limits.h>
#include <setjmp.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "R.h"
#include "R_ext/Rdynload.h"
#include "R_ext/Utils.h"

struct SEXPREC {
    int type;
    R_xlen_t length;
    int nrow, ncol;          /* -1: no dim attribute */
    void *data;              /* int / double / SEXP / char payload */
    SEXP names;
    struct SEXPREC *next;    /* allocation list */
};

static struct SEXPREC nil_rec = {NILSXP, 0, -1, -1, NULL, NULL, NULL};
static struct SEXPREC names_sym = {NILSXP, 0, -1, -1, NULL, NULL, NULL}, dim_sym = {NILSXP, 0, -1, -1, NULL, NULL, NULL};
SEXP R_NilValue = &nil_rec, R_NamesSymbol = &names_sym, R_DimSymbol = &dim_sym;
double R_NaReal;
int R_NaInt = INT_MIN;

static SEXP all_objects = NULL;
static long n_objects = 0, n_bytes = 0;
static int protect_depth = 0, protect_max = 0, unprotect_underflow = 0;
static long interrupt_polls = 0;
static jmp_buf *error_jmp = NULL;
static char error_msg[1024];
struct transient { void *p; struct transient *next; };
static struct transient *transients = NULL;

static const R_CallMethodDef *call_table = NULL;
static int dynamic_symbols = -1;

static void init_na(void) {
    union { double d; uint32_t w[2]; } u;
    u.w[1] = 0x7ff00000u; u.w[0] = 1954;          /* R's NA_real_: a NaN whose low word is 1954 */
    R_NaReal = u.d;
}

static size_t elt_size(int type) {
    switch (type) {
    case LGLSXP: case INTSXP: return sizeof(int);
    case REALSXP: return sizeof(double);
    case VECSXP: case STRSXP: return sizeof(SEXP);
    case CHARSXP: return 1;
    default: return 0;
    }
}

static SEXP new_object(int type, R_xlen_t n) {
    SEXP s = (SEXP)calloc(1, sizeof *s);
    size_t bytes = (size_t)(n > 0 ? n : 0) * elt_size(type) + (type == CHARSXP ? 1 : 0);
    s->type = type; s->length = n; s->nrow = s->ncol = -1; s->names = R_NilValue;
    s->data = bytes ? calloc(1, bytes) : NULL;
    if ((type == VECSXP || type == STRSXP) && n > 0)
        for (R_xlen_t i = 0; i < n; i++) ((SEXP *)s->data)[i] = R_NilValue;
    s->next = all_objects; all_objects = s;
    n_objects++; n_bytes += (long)bytes;
    return s;
}

/* ---- the API slice ------------------------------------------------------------------------------------------ */
int TYPEOF(SEXP s) { return s->type; }
int *INTEGER(SEXP s) { if (s->type != INTSXP && s->type != LGLSXP) Rf_error("INTEGER() can only be applied to a 'integer', not a type %d", s->type); return (int *)s->data; }
int *LOGICAL(SEXP s) { if (s->type != LGLSXP) Rf_error("LOGICAL() can only be applied to a 'logical', not a type %d", s->type); return (int *)s->data; }
double *REAL(SEXP s) { if (s->type != REALSXP) Rf_error("REAL() can only be applied to a 'numeric', not a type %d", s->type); return (double *)s->data; }
int Rf_length(SEXP s) { return (int)s->length; }
int Rf_isMatrix(SEXP s) { return s->nrow >= 0; }
int Rf_nrows(SEXP s) {
    if (s->type != LGLSXP && s->type != INTSXP && s->type != REALSXP && s->type != VECSXP && s->type != STRSXP) Rf_error("object is not a matrix");
    return s->nrow >= 0 ? s->nrow : (int)s->length;            /* R: a vector counts as one column */
}
int Rf_ncols(SEXP s) {
    if (s->type != LGLSXP && s->type != INTSXP && s->type != REALSXP && s->type != VECSXP && s->type != STRSXP) Rf_error("object is not a matrix");
    return s->nrow >= 0 ? s->ncol : 1;
}
double Rf_asReal(SEXP s) {
    if (s->length < 1) return R_NaReal;
    switch (s->type) {
    case REALSXP: return ((double *)s->data)[0];
    case INTSXP: case LGLSXP: return ((int *)s->data)[0] == R_NaInt ? R_NaReal : (double)((int *)s->data)[0];
    default: return R_NaReal;
    }
}
int Rf_asInteger(SEXP s) {
    if (s->length < 1) return R_NaInt;
    switch (s->type) {
    case REALSXP: { double d = ((double *)s->data)[0]; return (d != d || d >= 2147483648.0 || d <= -2147483649.0) ? R_NaInt : (int)d; }
    case INTSXP: case LGLSXP: return ((int *)s->data)[0];
    default: return R_NaInt;
    }
}
int Rf_asLogical(SEXP s) {
    if (s->length < 1) return R_NaInt;
    switch (s->type) {
    case LGLSXP: return ((int *)s->data)[0];
    case INTSXP: { int v = ((int *)s->data)[0]; return v == R_NaInt ? R_NaInt : v != 0; }
    case REALSXP: { double d = ((double *)s->data)[0]; return d != d ? R_NaInt : d != 0.0; }
    default: return R_NaInt;
    }
}
SEXP Rf_allocVector(int type, R_xlen_t n) {
    if (n < 0) Rf_error("negative length vectors are not allowed");
    return new_object(type, n);
}
SEXP Rf_allocMatrix(int type, int nr, int nc) {
    if (nr < 0 || nc < 0) Rf_error("negative extents to matrix");
    SEXP s = new_object(type, (R_xlen_t)nr * nc);
    s->nrow = nr; s->ncol = nc;
    return s;
}
SEXP Rf_coerceVector(SEXP s, int type) {
    if (s->type == type) return s;                                 /* R returns the object itself */
    if ((s->type != LGLSXP && s->type != INTSXP && s->type != REALSXP) || (type != LGLSXP && type != INTSXP && type != REALSXP))
        Rf_error("cannot coerce type %d to vector of type %d", s->type, type);
    SEXP r = new_object(type, s->length);
    r->nrow = s->nrow; r->ncol = s->ncol; r->names = s->names;     /* attributes are kept */
    for (R_xlen_t i = 0; i < s->length; i++) {
        if (s->type == REALSXP) {
            double d = ((double *)s->data)[i];
            int v = (d != d) ? R_NaInt : (type == LGLSXP ? d != 0.0 : (int)d);
            ((int *)r->data)[i] = v;
        } else {
            int v = ((int *)s->data)[i];
            if (type == REALSXP) ((double *)r->data)[i] = (v == R_NaInt) ? R_NaReal : (double)v;
            else ((int *)r->data)[i] = (v == R_NaInt) ? R_NaInt : (type == LGLSXP ? v != 0 : v);
        }
    }
    return r;
}
SEXP Rf_protect(SEXP s) { protect_depth++; if (protect_depth > protect_max) protect_max = protect_depth; return s; }
void Rf_unprotect(int n) { protect_depth -= n; if (protect_depth < 0) { unprotect_underflow++; protect_depth = 0; } }
SEXP Rf_mkChar(const char *c) {
    SEXP s = new_object(CHARSXP, (R_xlen_t)strlen(c));
    memcpy(s->data, c, strlen(c) + 1);
    return s;
}
SEXP SET_VECTOR_ELT(SEXP v, R_xlen_t i, SEXP x) {
    if (v->type != VECSXP || i < 0 || i >= v->length) Rf_error("SET_VECTOR_ELT: not a list or index out of range");
    ((SEXP *)v->data)[i] = x;
    return x;
}
void SET_STRING_ELT(SEXP v, R_xlen_t i, SEXP x) {
    if (v->type != STRSXP || x->type != CHARSXP || i < 0 || i >= v->length) Rf_error("SET_STRING_ELT: not a character vector / CHARSXP or index out of range");
    ((SEXP *)v->data)[i] = x;
}
SEXP Rf_setAttrib(SEXP s, SEXP name, SEXP val) {
    if (name == R_NamesSymbol) {
        if (val->type != STRSXP || val->length != s->length) Rf_error("'names' attribute must be the same length as the vector");
        s->names = val;
    } else if (name == R_DimSymbol) {
        if (val->type != INTSXP || val->length != 2) Rf_error("mock: only 2-d dim");
        s->nrow = ((int *)val->data)[0]; s->ncol = ((int *)val->data)[1];
    } else Rf_error("mock: unsupported attribute");
    return val;
}
char *R_alloc(size_t n, int size) {
    struct transient *t = (struct transient *)malloc(sizeof *t);
    t->p = calloc(n ? n : 1, (size_t)size);
    t->next = transients; transients = t;
    return (char *)t->p;
}
static void free_transients(void) {
    while (transients) { struct transient *t = transients; transients = t->next; free(t->p); free(t); }
}
void Rf_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(error_msg, sizeof error_msg, fmt, ap);
    va_end(ap);
    if (!error_jmp) { fprintf(stderr, "r_mock: Rf_error outside rmock_call: %s\n", error_msg); abort(); }
    longjmp(*error_jmp, 1);
}
void R_CheckUserInterrupt(void) { interrupt_polls++; }
int R_registerRoutines(DllInfo *dll, const void *c, const R_CallMethodDef *call, const void *f, const void *e) {
    (void)dll; (void)c; (void)f; (void)e;
    call_table = call;
    return 1;
}
Rboolean R_useDynamicSymbols(DllInfo *dll, Rboolean v) { (void)dll; dynamic_symbols = v; return TRUE; }

/* ---- the harness side (called from Python through ctypes) ------------------------------------------------------ */
void R_init_DESeq2(DllInfo *);

void rmock_load(void) {             /* what R does on library.dynam(): run the init routine */
    init_na();
    call_table = NULL; dynamic_symbols = -1;
    R_init_DESeq2(NULL);
}
int rmock_n_routines(void) { int k = 0; if (call_table) while (call_table[k].name) k++; return k; }
const char *rmock_routine_name(int i) { return call_table[i].name; }
int rmock_routine_arity(int i) { return call_table[i].numArgs; }
int rmock_dynamic_symbols(void) { return dynamic_symbols; }

/* a vector / matrix holding a COPY of `src` (ncol < 0: no dim); LGLSXP / INTSXP from int32, REALSXP from double */
SEXP rmock_new(int type, int nrow, int ncol, const void *src) {
    SEXP s = (ncol < 0) ? new_object(type, nrow) : new_object(type, (R_xlen_t)nrow * ncol);
    if (ncol >= 0) { s->nrow = nrow; s->ncol = ncol; }
    if (src && s->length) memcpy(s->data, src, (size_t)s->length * elt_size(type));
    return s;
}
SEXP rmock_nil(void) { return R_NilValue; }
int rmock_type(SEXP s) { return s->type; }
long rmock_length(SEXP s) { return (long)s->length; }
int rmock_nrow(SEXP s) { return s->nrow; }
int rmock_ncol(SEXP s) { return s->ncol; }
void *rmock_data(SEXP s) { return s->data; }
SEXP rmock_elt(SEXP s, long i) { return (s->type == VECSXP && i >= 0 && i < s->length) ? ((SEXP *)s->data)[i] : NULL; }
const char *rmock_name(SEXP s, long i) {
    if (s->names == R_NilValue || i < 0 || i >= s->names->length) return NULL;
    return (const char *)((SEXP *)s->names->data)[i]->data;
}
const char *rmock_last_error(void) { return error_msg; }
int rmock_protect_depth(void) { return protect_depth; }
int rmock_protect_max(void) { return protect_max; }
int rmock_unprotect_underflow(void) { return unprotect_underflow; }
long rmock_interrupt_polls(void) { return interrupt_polls; }
long rmock_live_objects(void) { return n_objects; }
long rmock_live_transients(void) { long k = 0; for (struct transient *t = transients; t; t = t->next) k++; return k; }
int rmock_is_na_real(double d) { union { double d; uint32_t w[2]; } u; u.d = d; return d != d && u.w[0] == 1954; }

void rmock_reset(void) {            /* release every object: the "garbage collection" between test cases */
    while (all_objects) { SEXP s = all_objects; all_objects = s->next; free(s->data); free(s); }
    n_objects = 0; n_bytes = 0; protect_depth = 0; protect_max = 0; unprotect_underflow = 0; interrupt_polls = 0;
    free_transients();
    error_msg[0] = 0;
}

typedef SEXP (*F5)(SEXP, SEXP, SEXP, SEXP, SEXP);
typedef SEXP (*F6)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP);
typedef SEXP (*F11)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP);
typedef SEXP (*F13)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP);
typedef SEXP (*F15)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP);
typedef SEXP (*F31)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP,
                    SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP);

/* .Call(name, args...): NULL + rmock_last_error() when the routine raised an R error (or is unknown / called with the
 * wrong number of arguments); the protect stack is unwound on error as R's context machinery does */
SEXP rmock_call(const char *name, int nargs, SEXP *a) {
    error_msg[0] = 0;
    const R_CallMethodDef *e = NULL;
    for (int i = 0; call_table && call_table[i].name; i++)
        if (!strcmp(call_table[i].name, name)) e = &call_table[i];
    if (!e) { snprintf(error_msg, sizeof error_msg, "\"%s\" not available for .Call() for package \"DESeq2\"", name); return NULL; }
    if (e->numArgs != nargs) {
        snprintf(error_msg, sizeof error_msg, "Incorrect number of arguments (%d), expecting %d for '%s'", nargs, e->numArgs, name);
        return NULL;
    }
    jmp_buf jb;
    const int depth0 = protect_depth;
    SEXP volatile out = NULL;
    error_jmp = &jb;
    if (setjmp(jb) == 0) {
        switch (nargs) {
        case 5: out = ((F5)e->fun)(a[0], a[1], a[2], a[3], a[4]); break;
        case 6: out = ((F6)e->fun)(a[0], a[1], a[2], a[3], a[4], a[5]); break;
        case 11: out = ((F11)e->fun)(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9], a[10]); break;
        case 13: out = ((F13)e->fun)(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9], a[10], a[11], a[12]); break;
        case 15: out = ((F15)e->fun)(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9], a[10], a[11], a[12], a[13], a[14]); break;
        case 31: out = ((F31)e->fun)(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9], a[10], a[11], a[12], a[13], a[14], a[15],
                                     a[16], a[17], a[18], a[19], a[20], a[21], a[22], a[23], a[24], a[25], a[26], a[27], a[28], a[29], a[30]); break;
        default: snprintf(error_msg, sizeof error_msg, "mock: no dispatcher for %d arguments", nargs); break;
        }
    } else {
        out = NULL;
        protect_depth = depth0;          /* R unwinds the pointer protection stack to the context's depth */
    }
    error_jmp = NULL;
    free_transients();
    return out;
}
