#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <hip/hip_runtime.h>
__device__ __forceinline__ double xor_partner_sum_dpp(double v) {
    // one butterfly step helper tests
    return v;
}
template <int CTRL, int BANK>
__device__ __forceinline__ int dpp_upd(int old, int src) { return __builtin_amdgcn_update_dpp(old, src, CTRL, 0xF, BANK, false); }

__device__ __forceinline__ double dpp_xor1(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, 0xB1, 0xF, 0xF, true);
    hi = __builtin_amdgcn_mov_dpp(hi, 0xB1, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double dpp_xor2(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, 0x4E, 0xF, 0xF, true);
    hi = __builtin_amdgcn_mov_dpp(hi, 0x4E, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double dpp_xor4(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    int l2 = dpp_upd<0x104, 0x5>(lo, lo);   // row_shl:4 into banks 0,2
    l2 = dpp_upd<0x114, 0xA>(l2, lo);       // row_shr:4 into banks 1,3
    int h2 = dpp_upd<0x104, 0x5>(hi, hi);
    h2 = dpp_upd<0x114, 0xA>(h2, hi);
    return __hiloint2double(h2, l2);
}
__device__ __forceinline__ double dpp_xor8(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, 0x128, 0xF, 0xF, true);
    hi = __builtin_amdgcn_mov_dpp(hi, 0x128, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
typedef unsigned uint2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double sum_xor16(double v) {
    unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    uint2v pl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    uint2v ph = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    double a = __hiloint2double((int)ph[0], (int)pl[0]), b = __hiloint2double((int)ph[1], (int)pl[1]);
    return a + b;
}
__device__ __forceinline__ double sum_xor32(double v) {
    unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    uint2v pl = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    uint2v ph = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    double a = __hiloint2double((int)ph[0], (int)pl[0]), b = __hiloint2double((int)ph[1], (int)pl[1]);
    return a + b;
}
__device__ __forceinline__ double allreduce_dpp(double v) {
    v = v + dpp_xor1(v);
    v = v + dpp_xor2(v);
    v = v + dpp_xor4(v);
    v = v + dpp_xor8(v);
    v = sum_xor16(v);
    v = sum_xor32(v);
    return v;
}
__device__ __forceinline__ double allreduce_shfl(double v) {
    for (int off = 1; off < 64; off <<= 1) v = v + __shfl_xor(v, off, 64);
    return v;
}
__global__ void k(const double *in, double *o1, double *o2, int reps, double *o3) {
    double v = in[threadIdx.x + blockIdx.x * blockDim.x];
    o1[threadIdx.x + blockIdx.x * blockDim.x] = allreduce_dpp(v);
    o2[threadIdx.x + blockIdx.x * blockDim.x] = allreduce_shfl(v);
}
__global__ void bench(const double *in, double *o, int reps, int mode) {
    double v = in[threadIdx.x];
    double acc = 0.0;
    for (int r = 0; r < reps; r++) {
        double t = mode ? allreduce_dpp(v + acc * 1e-300) : allreduce_shfl(v + acc * 1e-300);
        acc += t;
    }
    o[threadIdx.x + blockIdx.x * blockDim.x] = acc;
}
int main() {
    const int N = 256 * 64;
    double *h = (double *)malloc(N * 8), *d, *o1, *o2;
    srand(1);
    for (int i = 0; i < N; i++) h[i] = (rand() / (double)RAND_MAX - 0.5) * exp((rand() % 40) - 20.0);
    hipMalloc(&d, N * 8); hipMalloc(&o1, 1024 * 256 * 8); hipMalloc(&o2, N * 8);
    hipMemcpy(d, h, N * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(N / 256), dim3(256), 0, 0, d, o1, o2, 0, nullptr);
    double *r1 = (double *)malloc(N * 8), *r2 = (double *)malloc(N * 8);
    hipMemcpy(r1, o1, N * 8, hipMemcpyDeviceToHost); hipMemcpy(r2, o2, N * 8, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < N; i++) if (memcmp(&r1[i], &r2[i], 8)) bad++;
    printf("mismatches: %d of %d\n", bad, N);
    for (int mode = 0; mode < 2; mode++) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(bench, dim3(1024), dim3(256), 0, 0, d, o1, 2000, mode);
        hipEventRecord(e0);
        hipLaunchKernelGGL(bench, dim3(1024), dim3(256), 0, 0, d, o1, 2000, mode);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%s: %.3f ms (2000 dependent allreduces per wave, 4096 waves)\n", mode ? "dpp " : "shfl", ms);
    }
    return 0;
}
