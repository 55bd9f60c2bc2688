// dsq_rows.hpp -- how a wavefront sees "its" gene.
//
// HBM layout is gene-major: row i of Y (int32), mu / nf / weights (f64) holds the m
// samples of gene i contiguously (leading dimension ld), so lane l's loads of samples
// l, l+64, ... are 256-/512-byte coalesced segments.  Two access modes:
//   RowsLds    the row (as f64) has been staged into this wave's private LDS slab
//              and the design matrix X (column-major m x p) into a block-shared slab;
//              every later pass over the samples is LDS traffic only;
//   RowsGlobal no staging (m*p too large for LDS): every pass re-reads the row and X
//              through L1/L2.
// A lane only ever touches samples j == lane (mod 64), so the per-wave slabs need no
// barriers; only the shared X slab does (one __syncthreads after staging).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dsq {

#define DSQ_DEV __device__ __forceinline__

struct RowsLds {
    const int32_t *y_;   // m (counts stay int32 in LDS: 4 B/sample)
    const double *mu_;   // m
    const double *w_;    // m or nullptr
    const double *x_;    // p x m (column c at x_ + c*m)
    int m;
    DSQ_DEV double y(int j) const { return (double)y_[j]; }
    DSQ_DEV double mu(int j) const { return mu_[j]; }
    DSQ_DEV double w(int j) const { return w_[j]; }
    DSQ_DEV double x(int j, int c) const { return x_[c * m + j]; }
};

struct RowsGlobal {
    const int32_t *y_;
    const double *mu_;
    const double *w_;
    const double *x_;
    int m;
    DSQ_DEV double y(int j) const { return (double)y_[j]; }
    DSQ_DEV double mu(int j) const { return mu_[j]; }
    DSQ_DEV double w(int j) const { return w_[j]; }
    DSQ_DEV double x(int j, int c) const { return x_[c * m + j]; }
};

}  // namespace dsq
