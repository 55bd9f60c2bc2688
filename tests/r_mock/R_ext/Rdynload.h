/* tests/r_mock (see Rinternals.h): R_registerRoutines keeps the .Call table; rmock_call() dispatches through it by name
 * and refuses a call whose argument count differs from the registered arity, as R does */
#ifndef DSQ_RMOCK_RDYNLOAD_H
#define DSQ_RMOCK_RDYNLOAD_H
#include "../Rinternals.h"
typedef void *(*DL_FUNC)(void);
typedef struct { const char *name; DL_FUNC fun; int numArgs; } R_CallMethodDef;
typedef struct _DllInfo DllInfo;
int R_registerRoutines(DllInfo *, const void *, const R_CallMethodDef *, const void *, const void *);
Rboolean R_useDynamicSymbols(DllInfo *, Rboolean);
#endif
