// dsq_wave.hpp -- wavefront-level building blocks: one gfx950 wavefront (64 lanes)
// owns one gene.  Lane l holds samples l, l+64, l+128, ...; every sum over samples is
// "lane-serial then xor-butterfly", which is the summation order the arithmetic spec
// fixes (DESIGN.md "Arithmetic").  The p x p matrices are wave-uniform and live in
// registers of every lane (p is a template parameter; MFMA is pointless at p <= 16).
#pragma once
#include <hip/hip_runtime.h>

namespace dsq {

#define DSQ_DEV __device__ __forceinline__

// all-reduce: every lane ends with the same bits (a+b is commutative in IEEE).
DSQ_DEV double wave_allreduce(double v) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) v = v + __shfl_xor(v, off, 64);
    return v;
}

template <int N>
DSQ_DEV void wave_allreduce_n(double (&v)[N]) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        double t[N];
#pragma unroll
        for (int i = 0; i < N; i++) t[i] = __shfl_xor(v[i], off, 64);
#pragma unroll
        for (int i = 0; i < N; i++) v[i] = v[i] + t[i];
    }
}

DSQ_DEV double wave_bcast(double v, int lane) { return __shfl(v, lane, 64); }

// Gene scheduling of the persistent fit kernels.  Every wave starts on gene (block * waves + wave); the
// iteration counts of the fits are data dependent (2..100), so instead of a fixed grid stride the wave
// draws its next gene from a global counter (one relaxed atomic per gene, by lane 0).  Genes are
// independent, so the order does not touch the results.
DSQ_DEV int next_gene(int *counter, int g, int stride, int lane) {
    if (counter == nullptr) return g + stride;
    int nx = 0;
    if (lane == 0) nx = atomicAdd(counter, 1);
    return __builtin_amdgcn_readfirstlane(nx) + stride;
}

// wave-uniform predicate -> scalar branch
DSQ_DEV bool uniform(bool b) { return __builtin_amdgcn_readfirstlane((int)b) != 0; }

// ---- P x P LU with partial pivoting (first maximum wins), reciprocal pivots -----
template <int P>
struct LU {
    double a[P][P];
    double rdiag[P];
    int piv[P];
    int sign;

    DSQ_DEV void factor() {
        sign = 1;
#pragma unroll
        for (int k = 0; k < P; k++) {
            int pr = k;
            double best = __builtin_fabs(a[k][k]);
#pragma unroll
            for (int i = k + 1; i < P; i++) {
                double v = __builtin_fabs(a[i][k]);
                if (v > best) { best = v; pr = i; }
            }
            piv[k] = pr;
            if (pr != k) {
                sign = -sign;
#pragma unroll
                for (int i = k + 1; i < P; i++) {
                    if (i == pr) {
#pragma unroll
                        for (int j = 0; j < P; j++) { double t = a[k][j]; a[k][j] = a[i][j]; a[i][j] = t; }
                    }
                }
            }
            double rinv = 1.0 / a[k][k];
            rdiag[k] = rinv;
#pragma unroll
            for (int i = k + 1; i < P; i++) {
                double l = a[i][k] * rinv;
                a[i][k] = l;
#pragma unroll
                for (int j = k + 1; j < P; j++) a[i][j] = __builtin_fma(-l, a[k][j], a[i][j]);
            }
        }
    }
    DSQ_DEV double det() const {
        double d = a[0][0];
#pragma unroll
        for (int i = 1; i < P; i++) d = d * a[i][i];
        return sign < 0 ? -d : d;
    }
    DSQ_DEV void solve(double (&b)[P]) const {
#pragma unroll
        for (int k = 0; k < P; k++) {
            int pr = piv[k];
            if (pr != k) {
#pragma unroll
                for (int i = k + 1; i < P; i++) {
                    if (i == pr) { double t = b[k]; b[k] = b[i]; b[i] = t; }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < P; i++) {
            double t = b[i];
#pragma unroll
            for (int j = 0; j < i; j++) t = __builtin_fma(-a[i][j], b[j], t);
            b[i] = t;
        }
#pragma unroll
        for (int i = P - 1; i >= 0; i--) {
            double t = b[i];
#pragma unroll
            for (int j = i + 1; j < P; j++) t = __builtin_fma(-a[i][j], b[j], t);
            b[i] = t * rdiag[i];
        }
    }
    DSQ_DEV void inverse(double (&inv)[P][P]) const {
#pragma unroll
        for (int c = 0; c < P; c++) {
            double col[P];
#pragma unroll
            for (int i = 0; i < P; i++) col[i] = (i == c) ? 1.0 : 0.0;
            solve(col);
#pragma unroll
            for (int i = 0; i < P; i++) inv[i][c] = col[i];
        }
    }
};

template <int P>
DSQ_DEV void mat_mul(const double (&a)[P][P], const double (&b)[P][P], double (&c)[P][P]) {
#pragma unroll
    for (int i = 0; i < P; i++)
#pragma unroll
        for (int j = 0; j < P; j++) {
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < P; k++) acc = __builtin_fma(a[i][k], b[k][j], acc);
            c[i][j] = acc;
        }
}

template <int P>
DSQ_DEV double trace_prod(const double (&a)[P][P], const double (&b)[P][P]) {
    double acc = 0.0;
#pragma unroll
    for (int i = 0; i < P; i++)
#pragma unroll
        for (int k = 0; k < P; k++) acc = __builtin_fma(a[i][k], b[k][i], acc);
    return acc;
}

}  // namespace dsq
