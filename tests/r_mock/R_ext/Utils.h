/* tests/r_mock (see Rinternals.h): counts the polls so that a test can see the shim yield between ranges */
#ifndef DSQ_RMOCK_UTILS_H
#define DSQ_RMOCK_UTILS_H
void R_CheckUserInterrupt(void);
#endif
