/* dsq_arith_spec.h -- the part of the arithmetic specification where the ORDER of a sum depends on the
 * shape of the analysis.  One definition, read by the kernels (csrc/fit_disp.hip) and by the CPU checker of the test suite:
 * a retune of these numbers moves both together, and a tuning build that overrides the kernel's own macros without them
 * fails to compile instead of silently losing bit parity.
 *
 * Cox-Reid Gram sums of fitDisp in GENERAL mode (a design with a continuous covariate: no design cells): from
 * DSQ_SPEC_SERIAL_GRAM_MINP design columns up, on rows of at most DSQ_SPEC_SERIAL_GRAM_MAXM_NARROW samples (at most
 * DSQ_SPEC_SERIAL_GRAM_MAXM from DSQ_SPEC_SERIAL_GRAM_WIDE_P columns up), every matrix entry is the SERIAL sum of its m
 * terms x_ja (x_jb wd_j) in sample order; otherwise the wave-order sum (64 partials + butterfly). */
#ifndef DSQ_ARITH_SPEC_H
#define DSQ_ARITH_SPEC_H
#define DSQ_SPEC_SERIAL_GRAM_MINP 7
#define DSQ_SPEC_SERIAL_GRAM_MAXM_NARROW 256
#define DSQ_SPEC_SERIAL_GRAM_WIDE_P 10
#define DSQ_SPEC_SERIAL_GRAM_MAXM 1024
/* fitBeta's cell-collapsed least squares (one row per design cell + p ridge rows, one row per lane) serves designs of at
 * most this many columns (cells + padded columns <= 64 lanes); wider designs take the general per-sample sums (round 5: the
 * 48-column build; a 40-level factor, a paired design with more than 32 cells) */
#define DSQ_SPEC_BETA_CELL_MAXP 32
#endif
