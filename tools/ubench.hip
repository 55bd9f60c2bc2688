// ubench.hip -- instruction-cost microbenchmark for the f64 math the kernels are made of
// (GPU box only).  For each op: K independent dependent-chains per lane, W waves per SIMD;
// prints shader cycles per op per wave (s_memtime), i.e. the issue cost when K*W is large
// and the dependent latency when K = W = 1.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../deseq2_amd/csrc/dsq_math.hpp"
using namespace dsq;

enum { OP_FMA, OP_ADD, OP_MUL, OP_DIV, OP_SQRT, OP_LOG, OP_EXP, OP_LGAMMA, OP_DIGAMMA, OP_BD0, OP_RCP, OP_COUNT };
static const char *names[] = {"fma", "add", "mul", "div", "sqrt", "dlog", "dexp", "dlgamma(x>=10)", "ddigamma(x>=10)", "dbd0(log path)", "v_rcp_f64"};

template <int OP> __device__ __forceinline__ double apply(double x, double c) {
    if constexpr (OP == OP_FMA) return __builtin_fma(x, c, 0.25);
    if constexpr (OP == OP_ADD) return x + c;
    if constexpr (OP == OP_MUL) return x * c;
    if constexpr (OP == OP_DIV) return c / (x + 1.5);
    if constexpr (OP == OP_SQRT) return __builtin_sqrt(x + 2.0);
    if constexpr (OP == OP_LOG) return dlog(x + 3.0);
    if constexpr (OP == OP_EXP) return dexp(x * 0.01);
    if constexpr (OP == OP_LGAMMA) return dlgamma(x * 1e-3 + 12.0) * 0.01;
    if constexpr (OP == OP_DIGAMMA) return ddigamma(x * 1e-3 + 12.0);
    if constexpr (OP == OP_BD0) return dbd0(x + 5.0, 11.0);
    if constexpr (OP == OP_RCP) return __builtin_amdgcn_rcp(x + 1.5);
    return x;
}

template <int OP, int K>
__global__ void __launch_bounds__(256) kern(double *out, long long *cyc, int iters, double c) {
    double x[K];
#pragma unroll
    for (int k = 0; k < K; k++) x[k] = 1.0 + 0.001 * (threadIdx.x + k);
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < K; k++) x[k] = apply<OP>(x[k], c);
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
#pragma unroll
    for (int k = 0; k < K; k++) s += x[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP, int K> void run(int wps, double *out, long long *cyc) {
    int iters = (OP <= OP_MUL) ? 4000 : 600;
    int blocks = 256 * wps;   // 256-thread blocks: one wave per SIMD per block
    hipLaunchKernelGGL((kern<OP, K>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters, 0.999);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((kern<OP, K>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters, 0.999);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks);
    hipMemcpy(h.data(), cyc, blocks * sizeof(long long), hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= blocks;
    // s_memtime ticks at 100 MHz on some parts; report both tick-derived and wall-derived numbers
    double ops_per_wave = (double)iters * K;
    double ns_per_op_wall = ms * 1e6 / ops_per_wave;     // per wave, all waves concurrent
    printf("%-18s K=%d waves/SIMD=%d  ticks/op=%8.2f  wall ns/op/wave=%8.2f  -> ns per op per SIMD=%7.2f\n", names[OP], K, wps,
           avg / ops_per_wave, ns_per_op_wall, ns_per_op_wall / wps);
}

template <int OP> void all(double *out, long long *cyc) {
    run<OP, 1>(1, out, cyc); run<OP, 4>(1, out, cyc); run<OP, 1>(4, out, cyc); run<OP, 4>(4, out, cyc); run<OP, 2>(2, out, cyc);
}

int main() {
    double *out; long long *cyc;
    hipMalloc(&out, 256 * 8 * 256 * sizeof(double)); hipMalloc(&cyc, 256 * 8 * sizeof(long long));
    all<OP_FMA>(out, cyc); all<OP_ADD>(out, cyc); all<OP_DIV>(out, cyc); all<OP_RCP>(out, cyc); all<OP_SQRT>(out, cyc); all<OP_LOG>(out, cyc);
    all<OP_EXP>(out, cyc); all<OP_LGAMMA>(out, cyc); all<OP_DIGAMMA>(out, cyc); all<OP_BD0>(out, cyc);
    return 0;
}
