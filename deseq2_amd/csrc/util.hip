// util.hip -- layout kernels and the device-math test hook.
//
// R hands matrices over column-major (gene index fastest); the per-gene kernels want
// gene-major rows.  These are plain HBM-bound tiled transposes: a 64 x 64 tile goes
// through LDS (row padded by one element) so that both the global read (64 consecutive
// genes of one sample) and the global write (64 consecutive samples of one gene) are
// coalesced.  Algorithmic traffic: read + write once = 2 * sizeof(T) bytes per element.
#include "dsq_internal.hpp"
#include "dsq_math.hpp"

namespace dsq {

static constexpr int TILE = 64;

// src: column-major n x m  (i + n*j)  ->  dst: row-major, leading dim ld (i*ld + j)
template <typename TS, typename TD, bool CHECK>
__global__ void __launch_bounds__(256) r_to_gm_kernel(const TS *__restrict__ src, TD *__restrict__ dst, int n,
                                                      int m, long ld, int32_t *bad) {
    __shared__ TD tile[TILE][TILE + 1];
    const int tx = threadIdx.x & 63;   // fast index
    const int ty = threadIdx.x >> 6;   // 0..3
    const long i0 = (long)blockIdx.x * TILE;  // gene tile
    const long j0 = (long)blockIdx.y * TILE;  // sample tile
    int flag = 0;
#pragma unroll 4
    for (int r = ty; r < TILE; r += 4) {
        long i = i0 + tx, j = j0 + r;
        if (i < n && j < m) {
            TS v = src[i + (long)n * j];
            if constexpr (CHECK) {
                double dv = (double)v;
                if (!(dv >= 0.0) || !(dv <= 2147483647.0) || dv != __builtin_rint(dv)) flag = 1;
                tile[r][tx] = (TD)(flag ? 0 : dv);
            } else {
                tile[r][tx] = (TD)v;
            }
        }
    }
    __syncthreads();
#pragma unroll 4
    for (int r = ty; r < TILE; r += 4) {
        long i = i0 + r, j = j0 + tx;
        if (i < n && j < m) dst[i * ld + j] = tile[tx][r];
    }
    if constexpr (CHECK) {
        if (flag) atomicOr(bad, 1);
    }
}

// src: row-major ld  ->  dst: column-major n x m
template <typename T>
__global__ void __launch_bounds__(256) gm_to_r_kernel(const T *__restrict__ src, T *__restrict__ dst,
                                                      int n, int m, long ld) {
    __shared__ T tile[TILE][TILE + 1];
    const int tx = threadIdx.x & 63;
    const int ty = threadIdx.x >> 6;
    const long i0 = (long)blockIdx.x * TILE;
    const long j0 = (long)blockIdx.y * TILE;
#pragma unroll 4
    for (int r = ty; r < TILE; r += 4) {
        long i = i0 + r, j = j0 + tx;
        if (i < n && j < m) tile[r][tx] = src[i * ld + j];
    }
    __syncthreads();
#pragma unroll 4
    for (int r = ty; r < TILE; r += 4) {
        long i = i0 + tx, j = j0 + r;
        if (i < n && j < m) dst[i + (long)n * j] = tile[tx][r];
    }
}

static inline dim3 tgrid(int n, int m) { return dim3((n + TILE - 1) / TILE, (m + TILE - 1) / TILE); }

hipError_t launch_transpose_r_to_gm_f64(const double *src, double *dst, int n, int m, long ld, hipStream_t st) {
    hipLaunchKernelGGL((r_to_gm_kernel<double, double, false>), tgrid(n, m), dim3(256), 0, st, src, dst, n, m, ld,
                       (int32_t *)nullptr);
    return hipGetLastError();
}
hipError_t launch_transpose_r_to_gm_i32(const int32_t *src, int32_t *dst, int n, int m, long ld, hipStream_t st) {
    hipLaunchKernelGGL((r_to_gm_kernel<int32_t, int32_t, false>), tgrid(n, m), dim3(256), 0, st, src, dst, n, m, ld,
                       (int32_t *)nullptr);
    return hipGetLastError();
}
hipError_t launch_counts_f64_to_gm_i32(const double *src, int32_t *dst, int n, int m, long ld, int32_t *bad,
                                       hipStream_t st) {
    hipLaunchKernelGGL((r_to_gm_kernel<double, int32_t, true>), tgrid(n, m), dim3(256), 0, st, src, dst, n, m, ld,
                       bad);
    return hipGetLastError();
}
hipError_t launch_transpose_gm_to_r_f64(const double *src, double *dst, int n, int m, long ld, hipStream_t st) {
    hipLaunchKernelGGL(gm_to_r_kernel<double>, tgrid(n, m), dim3(256), 0, st, src, dst, n, m, ld);
    return hipGetLastError();
}
hipError_t launch_transpose_gm_to_r_i32(const int32_t *src, int32_t *dst, int n, int m, long ld, hipStream_t st) {
    hipLaunchKernelGGL(gm_to_r_kernel<int32_t>, tgrid(n, m), dim3(256), 0, st, src, dst, n, m, ld);
    return hipGetLastError();
}

// ---- device math test hook -------------------------------------------------------
__global__ void test_math_kernel(int op, const double *a, const double *b, const double *c, double *out, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double x = a[i], r;
    switch (op) {
    case 0: r = dexp(x); break;
    case 1: r = dlog(x); break;
    case 2: r = dlog1p(x); break;
    case 3: r = dlgamma(x); break;
    case 4: r = ddigamma(x); break;
    case 5: r = dtrigamma(x); break;
    case 6: r = dstirlerr(x); break;
    case 7: r = dbd0(x, b[i]); break;
    case 8: r = dnbinom_mu_log(x, b[i], c[i]); break;
    case 9: r = dpnorm_upper2(x); break;
    default: r = dnan();
    }
    out[i] = r;
}

hipError_t launch_test_math(int op, const double *a, const double *b, const double *c, double *out, long n,
                            hipStream_t st) {
    int blocks = (int)((n + 255) / 256);
    hipLaunchKernelGGL(test_math_kernel, dim3(blocks), dim3(256), 0, st, op, a, b, c, out, n);
    return hipGetLastError();
}

}  // namespace dsq
