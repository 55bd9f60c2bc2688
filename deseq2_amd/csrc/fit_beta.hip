// fit_beta.hip -- gfx950 kernels replacing fitBeta (src/DESeq2.cpp:283-465): ridge-
// penalised IRLS for the NB-GLM coefficients, hat diagonals, sandwich covariance and
// contrast, one wavefront per gene.  Two kernels:
//   fit_beta_cell_kernel  designs with at most 32 distinct rows (every factor design of at most 32 cells, cells + p <= 64): one linear predictor
//                         per design cell, the least squares on the collapsed (C + p) x p system, one sweep over the
//                         samples per iteration -- see the comment above it;
//   fit_beta_kernel       everything else (continuous covariates), described here.
// Both evaluate the deviance as -2 (K + D): K, the mu-independent part of the NB log densities, once per gene; D with one
// logarithm per sample and iteration (irls_constants / cell_dev_closed below).
//
// fit_beta_kernel: per IRLS iteration a wave makes lane-strided passes over its gene's m samples:
//   A  w = [wts] mu/(1+alpha mu), sqrt(w), z = log(mu/nf) + (y-mu)/mu   -> wave slab
//   B  weighted least squares for beta
//        useQR : Householder QR of [sqrt(w) X ; sqrt(ridge)] ((m+p) x p), LAPACK dgeqr2
//                order.  The reflected matrix is never stored: a row's state after k
//                reflections depends only on its initial value and the k wave-uniform
//                reflector parameters, so each of the p stages re-derives its rows on the
//                fly ("replay") and needs one round of p-k+1 wave reductions.
//        else  : normal equations X'WX + ridge by one pass + LU (partial pivoting)
//   C  mu = max(nf exp(X beta), minmu), log(mu / nf) kept for z; deviance = -2 sum [wts] log NB(y; 1/alpha, mu)
// The convergence test and the |beta| > 30 / NaN aborts are wave-uniform scalar flow.
// Slab (sqrt(w); mu, later sqrt(w) z; log(mu / nf) per sample) lives in wave-private LDS, X in a block-shared
// LDS slab; when m*p is too large for that both fall back to L2-resident global memory.
// From DSQ_BETA_WIDE_MIN design columns up the general and the optim kernel are built the "wide" way: the wave-uniform
// p x p work (four p x p matrices in the post-loop block alone: 800 VGPRs at p = 10) lives ONCE per wave in an LDS arena
// instead of in every lane's registers, the loops over p are rolled, the Gram matrices are accumulated two rows per
// pass.  Measured at p = 10: 512 VGPRs + 372 spilled (one wave per SIMD) -> 228, no spill; 20 000 genes, one
// continuous covariate: m = 100 4.9 -> 4.5 ms, m = 200 7.3 -> 7.1 ms, m = 500 12.9 -> 8.5 ms.  At p = 7, 8, 9 the
// register build is still the faster one (p = 9, m = 500: 9.0 vs 11.9 ms), hence the threshold
// (profiles/r03_general_path.txt).  The cell kernel is not affected (its own unrolling, LaneLU from p = 7).
#ifndef DSQ_BETA_WIDE_MIN
#define DSQ_BETA_WIDE_MIN 10
#endif
#define DSQ_WIDE_MIN DSQ_BETA_WIDE_MIN
#include "dsq_internal.hpp"
#include <cstdio>
#include <cstdlib>
#include "dsq_math.hpp"
#include "dsq_wave.hpp"
#include "../../include/dsq_arith_spec.h"
#include "fit_beta_common.hpp"
#include "dsq_prof.hpp"     // nb_offbranch, irls_constants: shared with the rolled wide kernel (fit_beta_wide.hip)

namespace dsq {

template <int P>
struct SymNB { static constexpr int value = P * (P + 1) / 2; };

#define DSQ_BETA_ARENA_MIN DSQ_WIDE_MIN
// WIDE build (DSQ_P >= DSQ_WIDE_MIN, see above and fit_disp.hip): the wave-uniform work arrays live ONCE per wave in an LDS arena instead
// of once per lane in scratch memory, and the Householder-stage loops stay fully unrolled so that a lane's row and
// its partial sums are registers.  In the per-width builds the macros expand to the plain local declarations.
#if DSQ_P >= DSQ_BETA_ARENA_MIN
#define DSQ_BWORK(T, name) T &name = *reinterpret_cast<T *>(arena + arena_off); arena_off += (int)((sizeof(T) + 7) / 8)
#define DSQ_BMARK(name) const int name = arena_off
#define DSQ_BRESET(name) arena_off = name
#define DSQ_UNROLL_Q _Pragma("unroll")
#else
#define DSQ_BWORK(T, name) T name
#define DSQ_BMARK(name)
#define DSQ_BRESET(name)
#define DSQ_UNROLL_Q DSQ_UNROLL_P
#endif
typedef double DsqVecP[DSQ_P];
typedef double DsqMatP[DSQ_P][DSQ_P];
typedef double DsqMatP1[DSQ_P][DSQ_P + 1];

// doubles of per-wave LDS arena the WIDE build needs: lambda, contrast, beta, beta_prev, then the larger of the
// QR stage state (scalS, tS, Rm, gamma) and the post-loop block (G, Gi, T, Sg, LU)
__host__ __device__ inline size_t beta_arena_doubles(int p) { return p >= DSQ_BETA_ARENA_MIN ? (size_t)5 * p * p + 12 * p + 32 : 0; }

// WIDE build: G[a][b] = sum_j x_ja (x_jb w_j) (b >= a, mirrored) and, optionally, rhs[a] = sum_j x_ja zw_j, two
// matrix rows per pass over the samples with the pass and column loops unrolled (register sums); per-sample weights
// are recomputed per pass.  Same terms, same order, same wave reduction as the one-pass form.
template <int P, bool WITH_RHS, class FW>
DSQ_DEV void beta_gram_wide(const double *xs, int m, int lane, FW &&wz, double (&G)[P][P], double *rhs) {
    constexpr int RB = 2;
    static_assert(P % RB == 0, "the wide builds have an even number of columns");
    static_for<P / RB>([&](auto ac) __attribute__((always_inline)) {
        constexpr int a0 = decltype(ac)::value * RB;
        double acc[RB][P], racc[RB];
        _Pragma("unroll")
        for (int i = 0; i < RB; i++) {
            racc[i] = 0.0;
            _Pragma("unroll")
            for (int b = 0; b < P; b++) acc[i][b] = 0.0;
        }
        for (int j = lane; j < m; j += 64) {
            double wv, zw;
            wz(j, wv, zw);
            double xr[P];
            _Pragma("unroll")
            for (int c = a0; c < P; c++) xr[c] = xs[c * m + j];
            _Pragma("unroll")
            for (int i = 0; i < RB; i++) {
                _Pragma("unroll")
                for (int b = a0 + i; b < P; b++) acc[i][b] += xr[a0 + i] * (xr[b] * wv);
                if constexpr (WITH_RHS) racc[i] += xr[a0 + i] * zw;
            }
        }
        // the sums of this pass reduced together (wave_allreduce_many: the bits of one wave_allreduce each)
        constexpr int NR = RB * (P - a0) - RB * (RB - 1) / 2 + (WITH_RHS ? RB : 0);
        double red[NR];
        {
            int q = 0;
            _Pragma("unroll")
            for (int i = 0; i < RB; i++) {
                _Pragma("unroll")
                for (int b = a0 + i; b < P; b++) red[q++] = acc[i][b];
                if constexpr (WITH_RHS) red[q++] = racc[i];
            }
        }
        wave_allreduce_many(red, lane);
        {
            int q = 0;
            _Pragma("unroll")
            for (int i = 0; i < RB; i++) {
                _Pragma("unroll")
                for (int b = a0 + i; b < P; b++) {
                    const double v = red[q++];
                    G[a0 + i][b] = v;
                    G[b][a0 + i] = v;
                }
                if constexpr (WITH_RHS) rhs[a0 + i] = red[q++];
            }
        }
    });
}

// per-sample state a wave keeps across passes: sqrt(w); mu and sqrt(w)*z sharing one slot (mu is
// dead once pass A has turned it into the working response); log(mu / nf) of the current mu (z and the
// closed-form deviance both use it)
static constexpr int kSlabVecs = 3;      // sqrt(w) | mu (or sqrt(w) z) | log(mu / nf)

// min waves per SIMD the register allocator must leave room for (2 => <= 256 unified registers)
#ifndef DSQ_BETA_MINW
#define DSQ_BETA_MINW (DSQ_P <= 6 ? 2 : 1)   /* wide designs already spill at 512 registers */
#endif

// From DSQ_BETA_ROLLED_MIN columns up (the zero-padded wide builds 16, 24, 32, 48) the general kernel below is NOT compiled:
// designs of those widths without design cells run on the rolled kernel of fit_beta_wide.hip (one build for every width; the
// unrolled Householder stages of this file took the optimiser 5-20 minutes per wide build and left one or two waves per CU).
#define DSQ_BETA_ROLLED_MIN 11
#if DSQ_P < DSQ_BETA_ROLLED_MIN
// QRROWS (staged rows, useQR, p >= DSQ_BETA_QRROWS_MIN): the rows of [sqrt(w) X ; sqrt(ridge) | sqrt(w) z] live in
// the wave's LDS and every Householder stage applies the previous reflection to them in place -- the operations of the
// replay on the same values in the same order (hence the same bits), p^2 instead of p^3 work per row, and the
// wave-uniform reflector state shrinks from 2 p^2 registers (512 VGPRs + spills at p = 10) to one reflector.
#ifndef DSQ_BETA_QRROWS_MIN
#define DSQ_BETA_QRROWS_MIN DSQ_BETA_WIDE_MIN
#endif
// QRM = 2 (round 4; m + p <= 64 DSQ_BETA_QRREG_TRIPS, from DSQ_BETA_QRREG_MIN columns): the same rows in REGISTERS -- row
// i of the least squares is trip i / 64 of lane i % 64, (p + 1) doubles each -- so a Householder stage is register
// arithmetic plus its round of wave reductions, no LDS traffic, and the wave's LDS shrinks to the slabs plus R and gamma
// (p = 10, m = 200: 29 KB -> 11 KB; five resident waves per CU -> eight, the register budget of two per SIMD).  The
// per-lane sums run over the trips in the order of the lane-strided loop: the same operations on the same values.
#ifndef DSQ_BETA_QRREG_MIN
#define DSQ_BETA_QRREG_MIN 7
#endif
#ifndef DSQ_BETA_QRREG_TRIPS
#define DSQ_BETA_QRREG_TRIPS (DSQ_P <= 12 ? 4 : 5)     /* two waves per SIMD: 4 (p + 1) doubles of rows; one: 5 (p + 1) */
#endif
#ifndef DSQ_BETA_QRREG_MINW
/* (p = 7, 8, 9 -- the builds whose p x p work sits in registers -- at one wave per SIMD: 1.98 / 2.20 / 3.43 ms against
   2.51 / 2.24 / 3.56 at two, 20 000 x 200, tools/r04u.sh) */
#define DSQ_BETA_QRREG_MINW (DSQ_P <= 9 ? 1 : DSQ_P <= 12 ? 2 : 1)
// QRM = 3: the same with twice the trips (m + p <= 512 at p <= 10) at one wave per SIMD -- the rows of a 500-sample
// analysis with continuous covariates (p = 7..9: instead of the replay; p = 10: instead of the rows in LDS)
#ifndef DSQ_BETA_QRREG2_MAXP
#define DSQ_BETA_QRREG2_MAXP 10
#endif
#endif
// per wave: QRM 1 the (m + p) x (p + 1) rows, R, gamma; QRM 2 R and gamma only
__host__ __device__ static inline size_t beta_qr_doubles(int m, int p, int qrm = 1) {
    return (qrm == 1 ? (size_t)(m + p) * (p + 1) : 0) + (size_t)p * p + p;
}

template <int P, bool USE_W, bool STAGE, int QRM>
__global__ void __launch_bounds__(256, (QRM == 1 ? 2 : QRM == 2 ? DSQ_BETA_QRREG_MINW : QRM == 3 ? 1 : DSQ_BETA_MINW)) fit_beta_kernel(BetaKernelParams kp) {
    constexpr bool QRROWS = QRM == 1;
    constexpr bool QRREG = QRM == 2 || QRM == 3;
    static_assert(QRM == 0 || STAGE, "stored rows need the staged layout");
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int waves = blockDim.x >> 6;
    const int m = kp.m;
    const int M = m + P;
    constexpr int N = SymNB<P>::value;
    const int nwork = DSQ_NWORK(kp);
    if (blockIdx.x * waves >= nwork) return;     // (row-listed launches size the grid without knowing the count)

    const double *xs = kp.x;          // X through L1/L2 unless it also fits in LDS
    double *slab;
    if constexpr (STAGE) {
        if (kp.xlds) {
            for (int t = threadIdx.x; t < P * m; t += blockDim.x) smem[t] = kp.x[t];
            __syncthreads();
            xs = smem;
        }
        slab = smem + (kp.xlds ? (size_t)P * m : 0) + (size_t)wave * m * kSlabVecs;
    } else {
        slab = kp.scratch + ((size_t)blockIdx.x * waves + wave) * (size_t)m * kSlabVecs;
    }
#if DSQ_P >= DSQ_BETA_ARENA_MIN
    double *arena = STAGE ? smem + (kp.xlds ? (size_t)P * m : 0) + (size_t)waves * m * kSlabVecs +
                                (size_t)wave * beta_arena_doubles(P)
                          : smem + (size_t)wave * beta_arena_doubles(P);
    int arena_off = 0;
#endif
    double *sw_s = slab, *mu_s = slab + m, *b_s = mu_s;   // mu and sqrt(w)*z share a slot
    double *lg_s = slab + 2 * (size_t)m;                   // log(mu / nf) of the current mu
    // QRROWS: behind all slabs (and arenas), per wave: the (m + P) x (P + 1) matrix column by column, then R and gamma
    double *qa = nullptr, *qR = nullptr, *qg = nullptr;
    if constexpr (QRROWS) {
        qa = smem + (kp.xlds ? (size_t)P * m : 0) + (size_t)waves * m * kSlabVecs + (size_t)waves * beta_arena_doubles(P) +
             (size_t)wave * beta_qr_doubles(m, P);
        qR = qa + (size_t)M * (P + 1);
        qg = qR + P * P;
    }
    if constexpr (QRREG) {
        qR = smem + (kp.xlds ? (size_t)P * m : 0) + (size_t)waves * m * kSlabVecs + (size_t)waves * beta_arena_doubles(P) +
             (size_t)wave * beta_qr_doubles(m, P, 2);
        qg = qR + P * P;
    }
    (void)qa;

    DSQ_BWORK(DsqVecP, lambda);
    DSQ_BWORK(DsqVecP, contrast);
    DSQ_BMARK(gene_mark);
DSQ_UNROLL_P
    for (int c = 0; c < P; c++) { lambda[c] = kp.lambda[c]; contrast[c] = kp.contrast[c]; }
    const double large = 30.0;

    for (int wi = blockIdx.x * waves + wave; wi < nwork; wi = next_gene(kp.work_counter, wi, gridDim.x * waves, lane)) {
        const int g = DSQ_GENE(kp, wi);
        const int32_t *yg = kp.y + (size_t)g * kp.ld;
        const double *nfg = kp.nf_is_vector ? kp.nf : kp.nf + (size_t)g * kp.ld;
        const double *wg = USE_W ? kp.weights + (size_t)g * kp.ld : nullptr;
        const double alpha = kp.alpha_hat[g];
        const double size = 1.0 / alpha;

        DSQ_BRESET(gene_mark);
        DSQ_BWORK(DsqVecP, beta);
DSQ_UNROLL_P
        for (int c = 0; c < P; c++) beta[c] = kp.beta_init[(size_t)g + (size_t)kp.n * c];

        // mu_hat = nfrow % exp(x * beta_hat), clamped at minmu            (:324-327, :361-364)
        auto update_mu = [&]() {
            for (int j = lane; j < m; j += 64) {
                double eta = xs[j] * beta[0];
DSQ_UNROLL_P
                for (int c = 1; c < P; c++) eta = __builtin_fma(xs[c * m + j], beta[c], eta);
                const double mu = __builtin_fmax(nfg[j] * dexp(eta), kp.minmu);
                mu_s[j] = mu;
                lg_s[j] = dlog(mu / nfg[j]);             // used by z (:349,:397) and by the deviance
            }
        };
        // w_vec / w_sqrt_vec                                      (:336-342, :390-396, :430-436)
        auto wvec = [&](int j, double mu) -> double {
            if constexpr (USE_W) return (wg[j] * mu) / (1.0 + alpha * mu);
            else return mu / (1.0 + alpha * mu);
        };

        update_mu();
        const int abl = kp.ablate;   // profiling only: 0 in production
        // dev = -2 (K + D): the mu-independent part of the NB log densities once per gene, one logarithm per sample and
        // iteration for the rest (the closed split of the cell kernel below)
        const bool fast = (alpha > 0.0) && dfinite(alpha) && dfinite(size) && (size > 0.0);
        double K = 0.0, Kp = 0.0;
        // (staged rows: the sqrt(w) slot of the wave's slab -- m doubles, first written inside the iterations -- lends the pass
        //  its flags and table: 2.5 T doubles, T = 256 from 640 samples, 128 from 320, 64 from 160)
        {
            const int tabT = m >= 640 ? 256 : m >= 320 ? 128 : m >= 160 ? 64 : 0;
            double *scr = (STAGE && tabT > 0) ? sw_s : nullptr;
            if (kp.maxit > 0 && !(abl & 16))
                K = irls_constants<USE_W>(yg, nfg, wg, m, lane, alpha, size, fast, nullptr, kp.kconst_out ? &Kp : nullptr, scr, tabT > 0 ? tabT : kIrlsTab);
        }
        if (kp.kconst_out && lane == 0) kp.kconst_out[g] = Kp;
        double dev = 0.0, dev_old = 0.0;
        double it = 0.0;
        DSQ_BWORK(DsqVecP, beta_prev);   // beta the current mu slot was computed from
        bool mu_lost = false;         // QR mode overwrote mu with sqrt(w)*z and beta then diverged
DSQ_UNROLL_P
        for (int c = 0; c < P; c++) beta_prev[c] = beta[c];
        DSQ_BMARK(iter_mark);
        for (int t = 0; t < kp.maxit; t++) {
            DSQ_BRESET(iter_mark);
            it += 1.0;
DSQ_UNROLL_P
            for (int c = 0; c < P; c++) beta_prev[c] = beta[c];
            if (abl & 2) {
                // (ablated: no least-squares solve)
            } else if (QRREG && kp.useQR) {
                if constexpr (QRREG) {
                constexpr int T = (QRM == 3 ? 2 : 1) * DSQ_BETA_QRREG_TRIPS;
                double ar[T][P + 1];
                // pass A: row i = lane + 64 t of [sqrt(w) X ; sqrt(ridge) | sqrt(w) z] into registers
                _Pragma("unroll")
                for (int t = 0; t < T; t++) {
                    const int i = lane + 64 * t;
                    _Pragma("unroll")
                    for (int c = 0; c <= P; c++) ar[t][c] = 0.0;
                    if (uniform(64 * t < M)) {
                        if (i < m) {
                            const double mu = mu_s[i];
                            const double sw = __builtin_sqrt(wvec(i, mu));
                            const double z = lg_s[i] + ((double)yg[i] - mu) / mu;
                            sw_s[i] = sw;
                            _Pragma("unroll")
                            for (int c = 0; c < P; c++) ar[t][c] = xs[c * m + i] * sw;
                            ar[t][P] = z * sw;
                        } else if (i < M) {
                            _Pragma("unroll")
                            for (int c = 0; c < P; c++) ar[t][c] = (i - m == c) ? __builtin_sqrt(lambda[c]) : 0.0;
                        }
                    }
                }
                // pass B: Householder QR, LAPACK dgeqr2 order; stage k first applies reflection k - 1 to the rows below it
                double tprev[P + 1];
                double scal_prev = 0.0;
                _Pragma("unroll")
                for (int j = 0; j <= P; j++) tprev[j] = 0.0;
                static_for<P>([&](auto kc) __attribute__((always_inline)) {
                    constexpr int k = decltype(kc)::value;
                    double acc[P + 1];
                    _Pragma("unroll")
                    for (int j = 0; j <= P; j++) acc[j] = 0.0;
                    _Pragma("unroll")
                    for (int t = 0; t < T; t++) {
                        const int i = lane + 64 * t;
                        if (uniform(64 * t < M)) {
                            if (i >= k && i < M) {                 // (rows below k are finished rows of R)
                                if (k > 0) {
                                    const double v = ar[t][k - 1] * scal_prev;
                                    _Pragma("unroll")
                                    for (int j = k; j <= P; j++) ar[t][j] = __builtin_fma(v, tprev[j], ar[t][j]);
                                }
                                if (i > k) {
                                    _Pragma("unroll")
                                    for (int j = k; j < P; j++) acc[j] += ar[t][k] * ar[t][j];
                                    acc[P] += ar[t][k] * ar[t][P];
                                }
                            }
                        }
                    }
                    double prow[P + 1];
                    _Pragma("unroll")
                    for (int j = 0; j <= P; j++) prow[j] = 0.0;
                    {
                        double red[P + 1 - k];
                        _Pragma("unroll")
                        for (int j = k; j <= P; j++) red[j - k] = acc[j];
                        wave_allreduce_many(red, lane);
                        _Pragma("unroll")
                        for (int j = k; j <= P; j++) acc[j] = red[j - k];
                    }
                    _Pragma("unroll")
                    for (int j = k; j <= P; j++) prow[j] = lane_read(ar[0][j], k);      // row k: trip 0 of lane k
                    const double alpha_k = prow[k];
                    double tau, scal, bet;
                    if (acc[k] == 0.0) { tau = 0.0; scal = 0.0; bet = alpha_k; }
                    else {
                        bet = -__builtin_copysign(__builtin_sqrt(alpha_k * alpha_k + acc[k]), alpha_k);
                        tau = (bet - alpha_k) / bet;
                        scal = 1.0 / (alpha_k - bet);
                    }
                    scal_prev = scal;
                    _Pragma("unroll")
                    for (int j = k + 1; j <= P; j++) {
                        const double wj = prow[j] + scal * acc[j];
                        tprev[j] = -tau * wj;
                    }
                    if (lane == 0) {
                        qR[k * P + k] = bet;
                        _Pragma("unroll")
                        for (int j = k + 1; j < P; j++) qR[k * P + j] = prow[j] + tprev[j];
                        qg[k] = prow[P] + tprev[P];
                    }
                });
                wave_lds_sync();
DSQ_UNROLL_Q
                for (int i = P - 1; i >= 0; i--) {
                    double tt = qg[i];
DSQ_UNROLL_Q
                    for (int j = i + 1; j < P; j++) tt = __builtin_fma(-qR[i * P + j], beta[j], tt);
                    beta[i] = tt / qR[i * P + i];
                }
                wave_lds_sync();
                }
            } else if (QRROWS && kp.useQR) {
                if constexpr (QRROWS) {
                // pass A: the rows of the least squares into LDS (column c of row i at qa[c M + i]; column P = sqrt(w) z)
                for (int i = lane; i < M; i += 64) {
                    if (i < m) {
                        const double mu = mu_s[i];
                        const double sw = __builtin_sqrt(wvec(i, mu));
                        const double z = lg_s[i] + ((double)yg[i] - mu) / mu;
                        sw_s[i] = sw;
DSQ_UNROLL_Q
                        for (int c = 0; c < P; c++) qa[(size_t)c * M + i] = xs[c * m + i] * sw;
                        qa[(size_t)P * M + i] = z * sw;
                    } else {
DSQ_UNROLL_Q
                        for (int c = 0; c < P; c++) qa[(size_t)c * M + i] = (i - m == c) ? __builtin_sqrt(lambda[c]) : 0.0;
                        qa[(size_t)P * M + i] = 0.0;
                    }
                }
                // pass B: Householder QR, LAPACK dgeqr2 order; stage k first applies reflection k - 1 to the rows below it
                double tprev[P + 1];
                double scal_prev = 0.0;
DSQ_UNROLL_Q
                for (int j = 0; j <= P; j++) tprev[j] = 0.0;
                static_for<P>([&](auto kc) __attribute__((always_inline)) {
                    constexpr int k = decltype(kc)::value;
                    double acc[P + 1], prow[P + 1];
DSQ_UNROLL_Q
                    for (int j = 0; j <= P; j++) { acc[j] = 0.0; prow[j] = 0.0; }
                    for (int i = lane; i < M; i += 64) {
                        if (i < k) continue;                       // finished rows of R (k < 64: only in the first trip)
                        double a[P + 1];
DSQ_UNROLL_Q
                        for (int j = (k > 0 ? k - 1 : 0); j <= P; j++) a[j] = qa[(size_t)j * M + i];
                        if (k > 0) {                               // (i >= k > k - 1: the reflection applies)
                            const double v = a[k - 1] * scal_prev;
DSQ_UNROLL_Q
                            for (int j = k; j <= P; j++) {
                                a[j] = __builtin_fma(v, tprev[j], a[j]);
                                qa[(size_t)j * M + i] = a[j];
                            }
                        }
                        if (i > k) {
DSQ_UNROLL_Q
                            for (int j = k; j < P; j++) acc[j] += a[k] * a[j];
                            acc[P] += a[k] * a[P];
                        } else {
DSQ_UNROLL_Q
                            for (int j = k; j <= P; j++) prow[j] = a[j];
                        }
                    }
                    {
                        double red[P + 1 - k];
                        _Pragma("unroll")
                        for (int j = k; j <= P; j++) red[j - k] = acc[j];
                        wave_allreduce_many(red, lane);
                        _Pragma("unroll")
                        for (int j = k; j <= P; j++) acc[j] = red[j - k];
                    }
DSQ_UNROLL_Q
                    for (int j = k; j <= P; j++) prow[j] = lane_read(prow[j], k);
                    const double alpha_k = prow[k];
                    double tau, scal, bet;
                    if (acc[k] == 0.0) { tau = 0.0; scal = 0.0; bet = alpha_k; }
                    else {
                        bet = -__builtin_copysign(__builtin_sqrt(alpha_k * alpha_k + acc[k]), alpha_k);
                        tau = (bet - alpha_k) / bet;
                        scal = 1.0 / (alpha_k - bet);
                    }
                    scal_prev = scal;
DSQ_UNROLL_Q
                    for (int j = k + 1; j <= P; j++) {
                        const double wj = prow[j] + scal * acc[j];
                        tprev[j] = -tau * wj;
                    }
                    if (lane == 0) {
                        qR[k * P + k] = bet;
DSQ_UNROLL_Q
                        for (int j = k + 1; j < P; j++) qR[k * P + j] = prow[j] + tprev[j];
                        qg[k] = prow[P] + tprev[P];
                    }
                });
                wave_lds_sync();
DSQ_UNROLL_Q
                for (int i = P - 1; i >= 0; i--) {
                    double tt = qg[i];
DSQ_UNROLL_Q
                    for (int j = i + 1; j < P; j++) tt = __builtin_fma(-qR[i * P + j], beta[j], tt);
                    beta[i] = tt / qR[i * P + i];
                }
                wave_lds_sync();
                }
            } else if (QRM == 0 && kp.useQR) {
                // pass A                                                       (:336-353)
                if (!(abl & 1))
                for (int j = lane; j < m; j += 64) {
                    double mu = mu_s[j];
                    double sw = __builtin_sqrt(wvec(j, mu));
                    double z = lg_s[j] + ((double)yg[j] - mu) / mu;
                    sw_s[j] = sw;
                    b_s[j] = z * sw;
                }
                // pass B: Householder QR by replay                              (:344-356)
                DSQ_BWORK(DsqVecP, scalS);
                DSQ_BWORK(DsqMatP1, tS);
                DSQ_BWORK(DsqMatP, Rm);
                DSQ_BWORK(DsqVecP, gamma);
                static_for<P>([&](auto kc) __attribute__((always_inline)) {
                    constexpr int k = decltype(kc)::value;
                    double acc[P + 1];
DSQ_UNROLL_Q
                    for (int j = 0; j <= P; j++) acc[j] = 0.0;
                    double prow[P + 1];
DSQ_UNROLL_Q
                    for (int j = 0; j <= P; j++) prow[j] = 0.0;
                    for (int i = lane; i < M; i += 64) {
                        double a[P], b;
                        if (i < m) {
                            double sw = sw_s[i];
DSQ_UNROLL_Q
                            for (int c = 0; c < P; c++) a[c] = xs[c * m + i] * sw;
                            b = b_s[i];
                        } else {
DSQ_UNROLL_Q
                            for (int c = 0; c < P; c++) a[c] = (i - m == c) ? __builtin_sqrt(lambda[c]) : 0.0;
                            b = 0.0;
                        }
DSQ_UNROLL_Q
                        for (int s = 0; s < k; s++) {
                            if (i > s) {
                                double v = a[s] * scalS[s];
DSQ_UNROLL_Q
                                for (int j = s + 1; j < P; j++) a[j] = __builtin_fma(v, tS[s][j], a[j]);
                                b = __builtin_fma(v, tS[s][P], b);
                            }
                            // rows i <= s are finished rows of R: never revisited (i >= k > s)
                        }
                        if (i > k) {
DSQ_UNROLL_Q
                            for (int j = k; j < P; j++) acc[j] += a[k] * a[j];
                            acc[P] += a[k] * b;
                        } else if (i == k) {
DSQ_UNROLL_Q
                            for (int j = k; j < P; j++) prow[j] = a[j];
                            prow[P] = b;
                        }
                    }
                    // reductions S_kj, j = k..P, and the pivot row from lane k
                    {                                           // reductions S_kj, j = k..P, together
                        double red[P + 1 - k];
                        _Pragma("unroll")
                        for (int j = k; j <= P; j++) red[j - k] = acc[j];
                        wave_allreduce_many(red, lane);
                        _Pragma("unroll")
                        for (int j = k; j <= P; j++) acc[j] = red[j - k];
                    }
DSQ_UNROLL_Q
                    for (int j = k; j <= P; j++) prow[j] = lane_read(prow[j], k);
                    double alpha_k = prow[k];
                    double tau, scal, bet;
                    if (acc[k] == 0.0) { tau = 0.0; scal = 0.0; bet = alpha_k; }
                    else {
                        bet = -__builtin_copysign(__builtin_sqrt(alpha_k * alpha_k + acc[k]), alpha_k);
                        tau = (bet - alpha_k) / bet;
                        scal = 1.0 / (alpha_k - bet);
                    }
                    scalS[k] = scal;
DSQ_UNROLL_Q
                    for (int j = k + 1; j <= P; j++) {
                        double wj = prow[j] + scal * acc[j];
                        tS[k][j] = -tau * wj;
                    }
                    Rm[k][k] = bet;
DSQ_UNROLL_Q
                    for (int j = k + 1; j < P; j++) Rm[k][j] = prow[j] + tS[k][j];
                    gamma[k] = prow[P] + tS[k][P];
                });
DSQ_UNROLL_Q
                for (int i = P - 1; i >= 0; i--) {
                    double tt = gamma[i];
DSQ_UNROLL_Q
                    for (int j = i + 1; j < P; j++) tt = __builtin_fma(-Rm[i][j], beta[j], tt);
                    beta[i] = tt / Rm[i][i];
                }
            } else {
                // solve(beta_hat, x.t() * (x.each_col() % w_vec) + ridge, x.t() * (z % w_vec))  (:398)
                if constexpr (P >= DSQ_WIDE_MIN) {
                    DSQ_BWORK(LU<P>, lu);
                    DSQ_BWORK(DsqVecP, rhs);
                    beta_gram_wide<P, true>(xs, m, lane, [&](int j, double &wv, double &zw) {
                        double mu = mu_s[j];
                        wv = wvec(j, mu);
                        double z = lg_s[j] + ((double)yg[j] - mu) / mu;
                        zw = z * wv;
                    }, lu.a, rhs);
DSQ_UNROLL_P
                    for (int a = 0; a < P; a++) lu.a[a][a] = lu.a[a][a] + lambda[a];
                    lu.factor();
                    lu.solve(rhs);
DSQ_UNROLL_P
                    for (int a = 0; a < P; a++) beta[a] = rhs[a];
                } else {
                    double acc[N + P];
DSQ_UNROLL_P
                    for (int i = 0; i < N + P; i++) acc[i] = 0.0;
                    for (int j = lane; j < m; j += 64) {
                        double mu = mu_s[j];
                        double wv = wvec(j, mu);
                        double z = lg_s[j] + ((double)yg[j] - mu) / mu;
                        double xr[P];
DSQ_UNROLL_P
                        for (int c = 0; c < P; c++) xr[c] = xs[c * m + j];
                        int idx = 0;
DSQ_UNROLL_P
                        for (int a = 0; a < P; a++) {
DSQ_UNROLL_P
                            for (int b = a; b < P; b++) acc[idx++] += xr[a] * (xr[b] * wv);
                            acc[N + a] += xr[a] * (z * wv);
                        }
                    }
                    wave_allreduce_many(acc, lane);
                    LU<P> lu;
                    int idx = 0;
DSQ_UNROLL_P
                    for (int a = 0; a < P; a++)
DSQ_UNROLL_P
                        for (int b = a; b < P; b++) { lu.a[a][b] = acc[idx]; lu.a[b][a] = acc[idx]; idx++; }
DSQ_UNROLL_P
                    for (int a = 0; a < P; a++) lu.a[a][a] = lu.a[a][a] + lambda[a];
                    lu.factor();
                    double rhs[P];
DSQ_UNROLL_P
                    for (int a = 0; a < P; a++) rhs[a] = acc[N + a];
                    lu.solve(rhs);
DSQ_UNROLL_P
                    for (int a = 0; a < P; a++) beta[a] = rhs[a];
                }
            }
            int toolarge = 0;
DSQ_UNROLL_P
            for (int c = 0; c < P; c++) toolarge += (__builtin_fabs(beta[c]) > large) ? 1 : 0;
            if (uniform(toolarge > 0)) { it = (double)kp.maxit; mu_lost = (kp.useQR != 0) && QRM == 0; break; }   // (:357-360)
            if (!(abl & 4)) update_mu();
            double dacc = 0.0;                                                            // (:365-373)
            if (!(abl & 8))
            for (int j = lane; j < m; j += 64) {
                const double y = (double)yg[j], mu = mu_s[j];
                double tj;
                if (cell_dev_closed(y, size, fast)) {
                    const double am = alpha * mu, opm = 1.0 + am, rcp = 1.0 / opm;
                    const double l1p = dlog(opm) + (am - (opm - 1.0)) * rcp;
                    tj = y * lg_s[j] - (y + size) * l1p;
                } else tj = nb_offbranch(y, size, mu);
                if constexpr (USE_W) dacc += wg[j] * tj;
                else dacc += tj;
            }
            dev = -2.0 * (K + wave_allreduce(dacc));
            double conv_test = __builtin_fabs(dev - dev_old) / (__builtin_fabs(dev) + 0.1);
            if (uniform(conv_test != conv_test)) { it = (double)kp.maxit; break; }        // (:375-378)
            if (kp.force_iters > 0) { if (t + 1 >= kp.force_iters) break; }
            else
            if (uniform((t > 0) && (conv_test < kp.tol))) break;                          // (:379-381)
            dev_old = dev;
        }

        // ---- post-loop block (:427-455) ------------------------------------------------
        if (mu_lost) {
            // the reference keeps the mu of the last completed update when beta diverges; here that
            // slot now holds sqrt(w)*z, so rebuild it from the coefficients it was computed from
            for (int j = lane; j < m; j += 64) {
                double eta = xs[j] * beta_prev[0];
DSQ_UNROLL_P
                for (int c = 1; c < P; c++) eta = __builtin_fma(xs[c * m + j], beta_prev[c], eta);
                mu_s[j] = __builtin_fmax(nfg[j] * dexp(eta), kp.minmu);
            }
        }
        DSQ_BRESET(iter_mark);
        DSQ_BWORK(DsqMatP, G);
        DSQ_BWORK(DsqMatP, Gi);
        if constexpr (P >= DSQ_WIDE_MIN) {
            // (a wave-uniform trip count: the exit of a lane-dependent loop is where the toolchain re-materialised the
            //  constants of the LAST sqrt of this block under an empty exec mask -- profiles/r04_exec_remat.md,
            //  tools/exec_lint.py)
            for (int j0 = 0; j0 < m; j0 += 64) {
                const int j = j0 + lane;
                if (j < m) sw_s[j] = __builtin_sqrt(wvec(j, mu_s[j]));
            }
            beta_gram_wide<P, false>(xs, m, lane, [&](int j, double &wv, double &zw) { wv = wvec(j, mu_s[j]); zw = 0.0; },
                                     G, nullptr);
        } else {
            double gacc[N];
DSQ_UNROLL_P
            for (int i = 0; i < N; i++) gacc[i] = 0.0;
            for (int j = lane; j < m; j += 64) {
                double wv = wvec(j, mu_s[j]);
                sw_s[j] = __builtin_sqrt(wv);
                double xr[P];
DSQ_UNROLL_P
                for (int c = 0; c < P; c++) xr[c] = xs[c * m + j];
                int idx = 0;
DSQ_UNROLL_P
                for (int a = 0; a < P; a++)
DSQ_UNROLL_P
                    for (int b = a; b < P; b++) gacc[idx++] += xr[a] * (xr[b] * wv);
            }
            wave_allreduce_many(gacc, lane);
            int idx = 0;
DSQ_UNROLL_P
            for (int a = 0; a < P; a++)
DSQ_UNROLL_P
                for (int b = a; b < P; b++) { G[a][b] = gacc[idx]; G[b][a] = gacc[idx]; idx++; }
        }
        {
            DSQ_BWORK(LU<P>, lu);
DSQ_UNROLL_P
            for (int a = 0; a < P; a++)
DSQ_UNROLL_P
                for (int b = 0; b < P; b++) lu.a[a][b] = G[a][b];
DSQ_UNROLL_P
            for (int a = 0; a < P; a++) lu.a[a][a] = lu.a[a][a] + lambda[a];
            lu.factor();
            lu.inverse(Gi);
        }
        // hat diagonal, loop order of :443-449; fitted means (extension)
        if (kp.hat_diagonals || kp.mu_out) {
            for (int j = lane; j < m; j += 64) {
                if (kp.hat_diagonals) {
                    double sw = sw_s[j];
                    double h = 0.0;
DSQ_UNROLL_P
                    for (int i1 = 0; i1 < P; i1++)
DSQ_UNROLL_P
                        for (int i2 = 0; i2 < P; i2++) {
                            double xw1 = xs[i1 * m + j] * sw;
                            double xw2 = xs[i2 * m + j] * sw;
                            h += xw1 * (xw2 * Gi[i2][i1]);
                        }
                    kp.hat_diagonals[(size_t)g * kp.ld + j] = h;
                }
                if (kp.mu_out) {
                    double eta = xs[j] * beta[0];
DSQ_UNROLL_P
                    for (int c = 1; c < P; c++) eta = __builtin_fma(xs[c * m + j], beta[c], eta);
                    double v = nfg[j] * dexp(eta);
                    if (kp.mu_floor > 0.0) v = __builtin_fmax(v, kp.mu_floor);
                    kp.mu_out[(size_t)g * kp.ld + j] = v;
                }
            }
        }
        // sigma = Gi * G * Gi                                                            (:452)
        DSQ_BWORK(DsqMatP, T);
        DSQ_BWORK(DsqMatP, Sg);
        mat_mul<P>(Gi, G, T);
        mat_mul<P>(T, Gi, Sg);
        double cn = 0.0;
DSQ_UNROLL_P
        for (int c = 0; c < P; c++) cn = __builtin_fma(contrast[c], beta[c], cn);
        double cd = 0.0;
DSQ_UNROLL_P
        for (int b = 0; b < P; b++) {
            double rr = 0.0;
DSQ_UNROLL_P
            for (int a = 0; a < P; a++) rr = __builtin_fma(contrast[a], Sg[a][b], rr);
            cd = __builtin_fma(rr, contrast[b], cd);
        }
        if (lane == 0) {
DSQ_UNROLL_P
            for (int c = 0; c < P; c++) {
                kp.beta_mat[(size_t)g + (size_t)kp.n * c] = beta[c];
                kp.beta_var_mat[(size_t)g + (size_t)kp.n * c] = Sg[c][c];
            }
            kp.iter[g] = it;
            kp.deviance[g] = dev;
            kp.contrast_num[g] = cn;
            kp.contrast_denom[g] = __builtin_sqrt(cd);
        }
    }
}

#endif   // DSQ_P < DSQ_BETA_ROLLED_MIN

// ---- cell-collapsed fitBeta -------------------------------------------------------------------------------------
// Designs with few distinct rows (every factor design: a handful of CELLS of samples sharing a design row x_c).
//   * the linear predictor is one value per cell (lane c owns cell c): mu_j = max(nf_j exp(eta_c), minmu);
//   * an IRLS step needs the samples only through S_c = sum w_j, T_c = sum w_j z_j: the weighted least squares is the
//     Householder QR of the COLLAPSED (C + p) x p matrix [sqrt(S_c) x_c ; sqrt(ridge)] -- one row per LANE, no pass
//     over the samples, no replay (the general kernel re-derives every sample row in every one of the p stages);
//   * one sweep over the samples per iteration computes mu, r = 1 / (1 + alpha mu), the deviance term
//     y lg - (y + size) log1p(alpha mu), w = [wts] mu r and w z = w lg + [wts] r (y - mu), and the cell sums for the
//     NEXT step (one division and one logarithm per sample): mu is never stored, so the kernel keeps only a 1 KB cell
//     slab per wave in LDS and occupancy is set by registers alone;
//   * post-loop: X'WX = sum_c (x_c x_c') S_c, hat diagonal h_j = w_j x_c'(X'WX + ridge)^-1 x_c; wave-uniform p x p
//     matrices in registers up to p = 6, one matrix column per lane (LaneLU, dsq_wave.hpp) from p = 7.
// Sums over samples run cell by cell in wave order over the position in the cell-sorted sample sequence (so a sweep
// is ceil(m / 64) FULL trips whatever the cell sizes).  Arithmetic spec = the CPU checker's
// fit_beta_gene_cells; results are bit-identical to it.
// design widths from DSQ_BETA_LANE_MIN up keep the p x p matrices of the cell kernel one column per lane (LaneLU)
#ifndef DSQ_BETA_LANE_MIN
#define DSQ_BETA_LANE_MIN 7
#endif
// the sweep of the cell kernel issues the loads of this many trips together
#ifndef DSQ_BETA_TRIP_BATCH
#define DSQ_BETA_TRIP_BATCH 4
#endif
#ifndef DSQ_BETA_CELL_MINW
/* (p = 10: three waves per SIMD measured at C4, 60 000 x 2000: 9.53 -> 9.14 ms; four: 11.9) */
#define DSQ_BETA_CELL_MINW (DSQ_P <= 6 ? 3 : DSQ_P == 10 ? 3 : DSQ_P <= 9 ? 2 : 1)
#endif

// LDS carve of the cell kernel, in doubles: ints (cell_start, pc) rounded to 16 bytes; per-wave slab
__host__ __device__ static inline size_t beta_cell_int_doubles(int m) { return (((size_t)DSQ_CMAX + 2 + m + 3) / 4) * 2; }
// per wave: the cell slab (4 doubles per cell) and the parked group sums of the cell closes ([cell][2][8], see sweep)
__host__ __device__ static inline size_t beta_cell_wave_doubles(int m, bool use_w) {
    (void)m; (void)use_w;
    return (4 + 16) * (size_t)DSQ_CMAX;
}
static_assert((4 + 16) * DSQ_CMAX >= kIrlsTab / 2 + 2 * kIrlsTab, "the constants pass borrows the wave's cell slab + park area");

// -DDSQ_WIDE_PROF (make prof): the phases of a gene's fit in shader-clock cycles, per wave, flushed once at the end of the
// kernel and printed by the launch (dsq_prof.hpp)
#ifdef DSQ_WIDE_PROF
__device__ unsigned long long betac_prof[DSQ_PROF_SLOTS];
#define DSQ_CPROF(slot) do { const unsigned long long t1_ = clock64(); pacc_k[slot] += t1_ - pt0_k; pt0_k = t1_; } while (0)
#else
#define DSQ_CPROF(slot)
#endif
template <int P, bool USE_W>
__global__ void __launch_bounds__(256, DSQ_BETA_CELL_MINW) fit_beta_cell_kernel(BetaKernelParams kp) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int waves = blockDim.x >> 6;
    const int m = kp.m;
    const int C = kp.ncell;
    const int Mrows = C + P;
    const int nwork = DSQ_NWORK(kp);
    if (blockIdx.x * waves >= nwork) return;
    // block-shared: x_c (C x P doubles) | cell_start (C + 1) | pc (m: sample | cell << 26, in cell-sorted order);
    // per wave: cell slab (4 doubles per cell: exp(eta), eta, exp(eta) of the reported means, hat factor)
    double *xcs = smem;
    int32_t *starts = reinterpret_cast<int32_t *>(smem + (size_t)DSQ_CMAX * P);
    int32_t *pc = starts + DSQ_CMAX + 2;
    const size_t lnf_doubles = kp.nf_is_vector ? (size_t)m : 0;      // log of the size factors: one table per block
    const size_t shared_doubles = (size_t)DSQ_CMAX * P + beta_cell_int_doubles(m) + lnf_doubles;
    const size_t wave_doubles = beta_cell_wave_doubles(m, USE_W);
    double *lnf_s = kp.nf_is_vector ? smem + (size_t)DSQ_CMAX * P + beta_cell_int_doubles(m) : nullptr;
    if (lnf_s)
        for (int t = threadIdx.x; t < m; t += blockDim.x) lnf_s[t] = dlog(kp.nf[t]);
    double *slab = smem + shared_doubles + (size_t)wave * wave_doubles;
    for (int t = threadIdx.x; t <= C; t += blockDim.x) starts[t] = kp.cell_start[t];
    for (int c = 0; c < C; c++) {
        const int s0 = kp.cell_start[c], s1 = kp.cell_start[c + 1];
        for (int t = s0 + (int)threadIdx.x; t < s1; t += blockDim.x) pc[t] = kp.cell_perm[t] | (c << 26);
    }
    for (int t = threadIdx.x; t < C * P; t += blockDim.x) {
        const int c = t / P, k = t - c * P;
        xcs[t] = kp.x[(size_t)k * m + kp.cell_perm[kp.cell_start[c]]];
    }
    __syncthreads();
    const int last_lane_of_tail = (m - 1) & 63;
    double lambda[P], contrast[P];            // (used by the wave-uniform builds only: P < DSQ_BETA_LANE_MIN)
#pragma unroll
    for (int c = 0; c < P; c++) { lambda[c] = kp.lambda[c]; contrast[c] = kp.contrast[c]; }
    // the ridge rows of the collapsed least squares: lane C + k holds sqrt(lambda_k) in column k
    const double sqrt_lam_lane = (lane >= C && lane < Mrows) ? __builtin_sqrt(kp.lambda[lane - C]) : 0.0;
    const double large = 30.0;
    // (its own SGPR pair: read as a sub-register of the 16-dword kernel-argument tuple, the allocator spilled the TUPLE and
    //  reloaded all sixteen registers in front of every use inside the sweep -- 32 v_readlane per trip for one operand)
    double minmu_ = kp.minmu;
    asm volatile("" : "+v"(minmu_));
    const double minmu = minmu_;

#ifdef DSQ_WIDE_PROF
    unsigned long long pacc_k[DSQ_PROF_SLOTS] = {}, pt0_k = clock64();
#endif
    for (int wi = blockIdx.x * waves + wave; wi < nwork; wi = next_gene(kp.work_counter, wi, gridDim.x * waves, lane)) {
        DSQ_CPROF(7);                                      // (drawing the gene)
        const int g = DSQ_GENE(kp, wi);
        const int32_t *yg = kp.y + (size_t)g * kp.ld;
        const double *nfg = kp.nf_is_vector ? kp.nf : kp.nf + (size_t)g * kp.ld;
        const double *wg = USE_W ? kp.weights + (size_t)g * kp.ld : nullptr;
        const double alpha = kp.alpha_hat[g];
        const double size = 1.0 / alpha;
        double beta[P];
#pragma unroll
        for (int c = 0; c < P; c++) beta[c] = kp.beta_init[(size_t)g + (size_t)kp.n * c];

        double etal = 0.0, expl = 0.0, exp_prev = 0.0;     // lane c: eta_c, exp(eta_c) of cell c
        double Sl = 0.0, Tl = 0.0;                         // lane c: S_c, T_c
        auto cell_eta = [&]() {
            if (lane < C) {
                double eta = xcs[lane * P] * beta[0];
#pragma unroll
                for (int k = 1; k < P; k++) eta = __builtin_fma(xcs[lane * P + k], beta[k], eta);
                etal = eta;
                expl = dexp(eta);
            }
        };
        const bool fast = (alpha > 0.0) && dfinite(alpha) && dfinite(size) && (size > 0.0);
        const bool with_dev_ever = (kp.maxit > 0) && fast;
        double K = 0.0, dev = 0.0;
        // the mu-independent part of the log densities, once per gene (samples in their natural order):
        // K_j = [saddle-point constants] + n log1p(alpha y) - y log y + y log nf_j,  n = y + size;  0 for y = 0
        double Kp = 0.0;
        // (the wave's cell slab and park area -- 128 + 512 doubles, written only from the first sweep on -- lend the constants
        //  pass its presence flags and its table)
        if (with_dev_ever) K = irls_constants<USE_W>(yg, nfg, wg, m, lane, alpha, size, fast, lnf_s, kp.kconst_out ? &Kp : nullptr, slab);
        if (kp.kconst_out && lane == 0) kp.kconst_out[g] = Kp;
        DSQ_CPROF(0);
        // one sweep over the samples at the current beta: positions k = lane, lane + 64, ... of the cell-sorted
        // sequence (full trips); the sums of a cell are closed when the sweep leaves it.  Deviance term of a sample:
        // y lg - (y + size) log1p(alpha mu), lg = log(mu / nf) -- one logarithm (of the rounded 1 + alpha mu, plus
        // its rounding residual) per sample and iteration
        auto sweep = [&](bool with_dev) {
            wave_lds_sync();
            if (lane < C) { slab[4 * lane] = expl; slab[4 * lane + 1] = etal; }
            wave_lds_sync();
            double dacc = 0.0;
            double a1 = 0.0, a2 = 0.0;
            int cur = 0;
            // PARKED CLOSES (round 5; the same scheme as fit_disp.hip's DispGene::pass): when the sweep leaves a cell the
            // butterfly steps inside groups of eight lanes (xor 1, 2, 4; S and T share them as in wave_allreduce_pair) run
            // and the eight group sums of each are parked in the wave's LDS, [cell][2][8]; after the sweep lane 8 g + c
            // takes group g of cell c and the steps xor 8, 16, 32 run once for eight cells together -- lane c ends with
            // S_c, T_c.  The additions of wave_allreduce on the same operands, ~ 16 instead of ~ 50 instructions per close.
            double *park = slab + 4 * (size_t)DSQ_CMAX;
            auto close_cell = [&]() {
                const bool odd = (lane & 1) != 0;
                const double keep = odd ? a2 : a1, send = odd ? a1 : a2;
                double v = keep + lane_xor1(send);
                v = v + lane_xor2(v);
                v = v + lane_xor4(v);
                if ((lane & 6) == 0) park[(cur * 2 + (lane & 1)) * 8 + (lane >> 3)] = v;
                a1 = 0.0; a2 = 0.0;
            };
            // one trip: positions k0 .. k0 + 63 of the cell-sorted sequence; the sample's count, normalization factor (and
            // weight) were loaded by the caller
            auto trip = [&](int k0, bool valid, int cmy, double nf, double y, double wt) {
                double wv = 0.0, wz = 0.0;
                if (valid) {
                    const double e = slab[4 * cmy], eta = slab[4 * cmy + 1];
                    const double raw = nf * e;
                    const double mu = __builtin_fmax(raw, minmu);
                    const double am = alpha * mu, opm = 1.0 + am, rcp = 1.0 / opm;
                    double rw = rcp;
                    if constexpr (USE_W) rw = wt * rcp;
                    wv = mu * rw;
                    // lg = (raw >= minmu) ? eta : log(mu / nf).  Floored means are rare: written as a select the compiler
                    // evaluates the division and the logarithm for every sample (a quarter of the sweep's instructions);
                    // behind a wave-uniform test they run only in a trip that holds a floored (or NaN) mean -- same values
                    const bool floored = !(raw >= minmu);
                    double lg = eta;
                    if (__any(floored)) lg = floored ? dlog(mu / nf) : eta;
                    wz = wv * lg + rw * (y - mu);       // w z, with w / mu = [wts] / (1 + alpha mu): no second division
                    if (with_dev) {
                        double t;
                        if (cell_dev_closed(y, size, fast)) {
                            const double l1p = dlog(opm) + (am - (opm - 1.0)) * rcp;
                            t = y * lg - (y + size) * l1p;
                        } else t = nb_offbranch(y, size, mu);
                        if constexpr (USE_W) dacc += wt * t;
                        else dacc += t;
                    }
                }
                const int c_lo = __builtin_amdgcn_readfirstlane(cmy);
                const int c_hi = __builtin_amdgcn_readlane(cmy, (k0 + 64 <= m) ? 63 : last_lane_of_tail);
                // general form: for c = c_lo .. c_hi { if (c != cur) { close; cur = c; } a += (cmy == c) ? w : 0 }; a trip inside
                // one cell (lanes past the end hold +0.0, what the masked form adds) and a trip across one boundary are
                // written out -- the same additions in the same order, without the loop
                if (c_lo == c_hi) {
                    if (c_lo != cur) { close_cell(); cur = c_lo; }
                    a1 += wv;
                    a2 += wz;
                } else if (c_hi == c_lo + 1) {
                    if (c_lo != cur) { close_cell(); cur = c_lo; }
                    const bool first = cmy == c_lo, second = cmy == c_hi;
                    a1 += first ? wv : 0.0;
                    a2 += first ? wz : 0.0;
                    close_cell(); cur = c_hi;
                    a1 += second ? wv : 0.0;
                    a2 += second ? wz : 0.0;
                } else {
                    for (int c = c_lo; c <= c_hi; c++) {
                        if (c != cur) { close_cell(); cur = c; }
                        const bool mine = cmy == c;
                        a1 += mine ? wv : 0.0;
                        a2 += mine ? wz : 0.0;
                    }
                }
            };
            // the row is read through L2 in every sweep: the loads of NB trips are issued together, so that a wave pays
            // the memory latency once per NB trips (few resident waves per SIMD: they do not hide it)
            constexpr int NB = DSQ_BETA_TRIP_BATCH;
            for (int k0 = 0; k0 < m; k0 += 64 * NB) {
                int cb[NB];
                bool vb[NB];
                double nb_[NB], yb[NB], wb[NB];
                _Pragma("unroll")
                for (int b = 0; b < NB; b++) {
                    const int k = k0 + 64 * b + lane;
                    const bool valid = k < m;
                    const int pk = pc[valid ? k : m - 1];
                    const int j = pk & 0x3ffffff;
                    cb[b] = valid ? (pk >> 26) : -1;
                    vb[b] = valid;
                    nb_[b] = nfg[j];
                    yb[b] = (double)yg[j];
                    wb[b] = 1.0;
                    if constexpr (USE_W) wb[b] = wg[j];
                }
                _Pragma("unroll")
                for (int b = 0; b < NB; b++)
                    if (k0 + 64 * b < m) trip(k0 + 64 * b, vb[b], cb[b], nb_[b], yb[b], wb[b]);
            }
            close_cell();
            wave_lds_sync();
            for (int r8 = 0; r8 < C; r8 += 8) {
                const int c = r8 + (lane & 7);
                const int cc = c < C ? c : C - 1;             // (lanes past the last cell read a valid slot; not used)
                double vs = park[(cc * 2) * 8 + (lane >> 3)], vt = park[(cc * 2 + 1) * 8 + (lane >> 3)];
                vs = vs + lane_xor8(vs);
                vt = vt + lane_xor8(vt);
                double x_, y_;
                lane_pair16(vs, x_, y_); vs = x_ + y_;
                lane_pair16(vt, x_, y_); vt = x_ + y_;
                lane_pair32(vs, x_, y_); vs = x_ + y_;
                lane_pair32(vt, x_, y_); vt = x_ + y_;
                if (lane >= r8 && lane < r8 + 8) { Sl = vs; Tl = vt; }
            }
            if (with_dev) dev = -2.0 * (K + wave_allreduce(dacc));
        };

        cell_eta();
        DSQ_CPROF(3);
        sweep(false);
        DSQ_CPROF(1);
        double dev_old = 0.0, it = 0.0;
        for (int t = 0; t < kp.maxit; t++) {
            it += 1.0;
            exp_prev = expl;
            if (kp.useQR) {
                // rows: lane c < C the cell rows, lanes C .. C+P-1 the ridge rows, the rest zero
                double a[P], b = 0.0;
                const double sS = __builtin_sqrt(Sl);
#pragma unroll
                for (int k = 0; k < P; k++) {
                    double v = 0.0;
                    if (lane < C) v = xcs[lane * P + k] * sS;
                    else if (lane - C == k) v = sqrt_lam_lane;
                    a[k] = v;
                }
                if (lane < C) b = (sS > 0.0) ? Tl / sS : 0.0;
                static_for<P>([&](auto kc) __attribute__((always_inline)) {
                    constexpr int k = decltype(kc)::value;
                    double acc[P + 1];
                    const bool below = (lane > k) && (lane < Mrows);
#pragma unroll
                    for (int j = k; j < P; j++) acc[j] = below ? a[k] * a[j] : 0.0;
                    acc[P] = below ? a[k] * b : 0.0;
#ifdef DSQ_BETA_CELL_SINGLE_REDUCTIONS
#pragma unroll
                    for (int j = k; j <= P; j++) acc[j] = wave_allreduce_low(acc[j], Mrows);   // (only rows < Mrows use them)
#else
                    {                                           // the p + 1 - k sums of the stage together (dsq_wave.hpp)
                        double red[P + 1 - k];
#pragma unroll
                        for (int j = k; j <= P; j++) red[j - k] = acc[j];
                        wave_allreduce_many_low(red, lane, Mrows);
#pragma unroll
                        for (int j = k; j <= P; j++) acc[j] = red[j - k];
                    }
#endif
                    double prow[P + 1];
#pragma unroll
                    for (int j = k; j < P; j++) prow[j] = lane_read(a[j], k);
                    prow[P] = lane_read(b, k);
                    const double alpha_k = prow[k];
                    double tau, scal, bet;
                    if (acc[k] == 0.0) { tau = 0.0; scal = 0.0; bet = alpha_k; }
                    else {
                        bet = -__builtin_copysign(__builtin_sqrt(alpha_k * alpha_k + acc[k]), alpha_k);
                        tau = (bet - alpha_k) / bet;
                        scal = 1.0 / (alpha_k - bet);
                    }
                    double tvec[P + 1];
#pragma unroll
                    for (int j = k + 1; j <= P; j++) tvec[j] = -tau * (prow[j] + scal * acc[j]);
                    if (below) {
                        const double v = a[k] * scal;
#pragma unroll
                        for (int j = k + 1; j < P; j++) a[j] = __builtin_fma(v, tvec[j], a[j]);
                        b = __builtin_fma(v, tvec[P], b);
                    } else if (lane == k) {
#pragma unroll
                        for (int j = k + 1; j < P; j++) a[j] = a[j] + tvec[j];
                        b = b + tvec[P];
                        a[k] = bet;
                    }
                });
#pragma unroll
                for (int i = P - 1; i >= 0; i--) {
                    double tt = lane_read(b, i);
#pragma unroll
                    for (int j = i + 1; j < P; j++) tt = __builtin_fma(-lane_read(a[j], i), beta[j], tt);
                    beta[i] = tt / lane_read(a[i], i);
                }
            } else if constexpr (P >= DSQ_BETA_LANE_MIN) {
                // normal equations with one column of X'WX + ridge per lane (LaneLU): entry (i, b) = sum_c (x_c[i] x_c[b]) S_c
                LaneLU<P> lu;
                const int bl = lane < P ? lane : 0;
#pragma unroll
                for (int i = 0; i < P; i++) lu.a[i] = 0.0;
                double rh = 0.0;
                for (int c = 0; c < C; c++) {
                    const double sc = lane_read(Sl, c), tc = lane_read(Tl, c);
                    const double xb = xcs[c * P + bl];
#pragma unroll
                    for (int i = 0; i < P; i++) lu.a[i] = lu.a[i] + (xcs[c * P + i] * xb) * sc;
                    rh += xb * tc;
                }
#pragma unroll
                for (int i = 0; i < P; i++) lu.a[i] = (lane == i) ? lu.a[i] + kp.lambda[i] : lu.a[i];
                lu.factor(lane);
                double rhs[P];
#pragma unroll
                for (int i = 0; i < P; i++) rhs[i] = lane_read(rh, i);
                lu.solve(rhs);
#pragma unroll
                for (int i = 0; i < P; i++) beta[i] = rhs[i];
            } else {
                LU<P> lu;
                double rhs[P];
#pragma unroll
                for (int a = 0; a < P; a++) {
#pragma unroll
                    for (int b = a; b < P; b++) {
                        double v = 0.0;
                        for (int c = 0; c < C; c++) v += (xcs[c * P + a] * xcs[c * P + b]) * lane_read(Sl, c);
                        lu.a[a][b] = v; lu.a[b][a] = v;
                    }
                    double v = 0.0;
                    for (int c = 0; c < C; c++) v += xcs[c * P + a] * lane_read(Tl, c);
                    rhs[a] = v;
                }
#pragma unroll
                for (int a = 0; a < P; a++) lu.a[a][a] = lu.a[a][a] + lambda[a];
                lu.factor();
                lu.solve(rhs);
#pragma unroll
                for (int a = 0; a < P; a++) beta[a] = rhs[a];
            }
            int toolarge = 0;
#pragma unroll
            for (int c = 0; c < P; c++) toolarge += (__builtin_fabs(beta[c]) > large) ? 1 : 0;
            if (uniform(toolarge > 0)) { it = (double)kp.maxit; expl = exp_prev; break; }          // (:357-360)
            DSQ_CPROF(2);
            cell_eta();
            DSQ_CPROF(3);
            sweep(true);
            DSQ_CPROF(1);
            const double conv_test = __builtin_fabs(dev - dev_old) / (__builtin_fabs(dev) + 0.1);
            if (uniform(conv_test != conv_test)) { it = (double)kp.maxit; break; }                  // (:375-378)
            if (kp.force_iters > 0) { if (t + 1 >= kp.force_iters) break; }
            else if (uniform((t > 0) && (conv_test < kp.tol))) break;                               // (:379-381)
            dev_old = dev;
        }

        DSQ_CPROF(4);
        // ---- post-loop block (:427-455) from the cell sums of the final mu ----------------------------------------
        if constexpr (P >= DSQ_BETA_LANE_MIN) {
            // one matrix column per lane: X'WX, its ridge inverse and sigma = Gi G Gi cost 2 P registers each
            const int bl = lane < P ? lane : 0;
            double Gc[P], Gic[P];
#pragma unroll
            for (int i = 0; i < P; i++) Gc[i] = 0.0;
            for (int c = 0; c < C; c++) {
                const double sc = lane_read(Sl, c);
                const double xb = xcs[c * P + bl];
#pragma unroll
                for (int i = 0; i < P; i++) Gc[i] = Gc[i] + (xcs[c * P + i] * xb) * sc;
            }
            {
                LaneLU<P> lu;
#pragma unroll
                for (int i = 0; i < P; i++) lu.a[i] = (lane == i) ? Gc[i] + kp.lambda[i] : Gc[i];
                lu.factor(lane);
                lu.inverse(Gic, lane);
            }
            DSQ_CPROF(5);
            if (kp.hat_diagonals || kp.mu_out) {
                wave_lds_sync();
                {
                    const int cl = lane < C ? lane : 0;
                    double eta = xcs[cl * P] * beta[0];
#pragma unroll
                    for (int k = 1; k < P; k++) eta = __builtin_fma(xcs[cl * P + k], beta[k], eta);
                    double h = 0.0;
#pragma unroll
                    for (int i1 = 0; i1 < P; i1++) {
                        const double x1 = xcs[cl * P + i1];
#pragma unroll
                        for (int i2 = 0; i2 < P; i2++) h += x1 * (xcs[cl * P + i2] * lane_read(Gic[i2], i1));
                    }
                    if (lane < C) {
                        slab[4 * lane] = expl;
                        slab[4 * lane + 2] = dexp(eta);
                        slab[4 * lane + 3] = h;
                    }
                }
                wave_lds_sync();
                for (int k = lane; k < m; k += 64) {
                    const int pk = pc[k];
                    const int j = pk & 0x3ffffff, c = pk >> 26;
                    const double nf = nfg[j];
                    if (kp.hat_diagonals) {
                        const double mu = __builtin_fmax(nf * slab[4 * c], minmu);
                        const double rcp = 1.0 / (1.0 + alpha * mu);
                        double wv;
                        if constexpr (USE_W) wv = mu * (wg[j] * rcp);
                        else wv = mu * rcp;
                        kp.hat_diagonals[(size_t)g * kp.ld + j] = wv * slab[4 * c + 3];
                    }
                    if (kp.mu_out) {
                        double v = nf * slab[4 * c + 2];
                        if (kp.mu_floor > 0.0) v = __builtin_fmax(v, kp.mu_floor);
                        kp.mu_out[(size_t)g * kp.ld + j] = v;
                    }
                }
            }
            DSQ_CPROF(6);
            double Tc[P], Sgc[P];
            lane_mat_mul<P>(Gic, Gc, Tc);
            lane_mat_mul<P>(Tc, Gic, Sgc);
            double cn = 0.0;
#pragma unroll
            for (int c = 0; c < P; c++) cn = __builtin_fma(kp.contrast[c], beta[c], cn);
            double rr = 0.0;
#pragma unroll
            for (int a = 0; a < P; a++) rr = __builtin_fma(kp.contrast[a], Sgc[a], rr);
            double cd = 0.0;
#pragma unroll
            for (int b = 0; b < P; b++) cd = __builtin_fma(lane_read(rr, b), kp.contrast[b], cd);
            double sdiag = 0.0, bsel = 0.0;
#pragma unroll
            for (int i = 0; i < P; i++) { sdiag = (lane == i) ? Sgc[i] : sdiag; bsel = (lane == i) ? beta[i] : bsel; }
            if (lane < P) {
                kp.beta_mat[(size_t)g + (size_t)kp.n * lane] = bsel;
                kp.beta_var_mat[(size_t)g + (size_t)kp.n * lane] = sdiag;
            }
            if (lane == 0) {
                kp.iter[g] = it;
                kp.deviance[g] = dev;
                kp.contrast_num[g] = cn;
                kp.contrast_denom[g] = __builtin_sqrt(cd);
            }
        } else {
        double G[P][P], Gi[P][P];
#pragma unroll
        for (int a = 0; a < P; a++)
#pragma unroll
            for (int b = a; b < P; b++) {
                double v = 0.0;
                for (int c = 0; c < C; c++) v += (xcs[c * P + a] * xcs[c * P + b]) * lane_read(Sl, c);
                G[a][b] = v; G[b][a] = v;
            }
        {
            LU<P> lu;
#pragma unroll
            for (int a = 0; a < P; a++)
#pragma unroll
                for (int b = 0; b < P; b++) lu.a[a][b] = G[a][b];
#pragma unroll
            for (int a = 0; a < P; a++) lu.a[a][a] = lu.a[a][a] + lambda[a];
            lu.factor();
            lu.inverse(Gi);
        }
        DSQ_CPROF(5);
        if (kp.hat_diagonals || kp.mu_out) {
            // fitted means from the FINAL coefficients (also when they diverged), as the general kernel reports them
            wave_lds_sync();
            if (lane < C) {
                double eta = xcs[lane * P] * beta[0];
#pragma unroll
                for (int k = 1; k < P; k++) eta = __builtin_fma(xcs[lane * P + k], beta[k], eta);
                double h = 0.0;
#pragma unroll
                for (int i1 = 0; i1 < P; i1++)
#pragma unroll
                    for (int i2 = 0; i2 < P; i2++) h += xcs[lane * P + i1] * (xcs[lane * P + i2] * Gi[i2][i1]);
                slab[4 * lane] = expl;
                slab[4 * lane + 2] = dexp(eta);
                slab[4 * lane + 3] = h;
            }
            wave_lds_sync();
            for (int k = lane; k < m; k += 64) {
                const int pk = pc[k];
                const int j = pk & 0x3ffffff, c = pk >> 26;
                const double nf = nfg[j];
                if (kp.hat_diagonals) {
                    const double mu = __builtin_fmax(nf * slab[4 * c], minmu);
                    const double rcp = 1.0 / (1.0 + alpha * mu);
                    double wv;
                    if constexpr (USE_W) wv = mu * (wg[j] * rcp);
                    else wv = mu * rcp;
                    kp.hat_diagonals[(size_t)g * kp.ld + j] = wv * slab[4 * c + 3];
                }
                if (kp.mu_out) {
                    double v = nf * slab[4 * c + 2];
                    if (kp.mu_floor > 0.0) v = __builtin_fmax(v, kp.mu_floor);
                    kp.mu_out[(size_t)g * kp.ld + j] = v;
                }
            }
        }
        DSQ_CPROF(6);
        double T[P][P], Sg[P][P];
        mat_mul<P>(Gi, G, T);
        mat_mul<P>(T, Gi, Sg);
        double cn = 0.0;
#pragma unroll
        for (int c = 0; c < P; c++) cn = __builtin_fma(contrast[c], beta[c], cn);
        double cd = 0.0;
#pragma unroll
        for (int b = 0; b < P; b++) {
            double rr = 0.0;
#pragma unroll
            for (int a = 0; a < P; a++) rr = __builtin_fma(contrast[a], Sg[a][b], rr);
            cd = __builtin_fma(rr, contrast[b], cd);
        }
        if (lane == 0) {
#pragma unroll
            for (int c = 0; c < P; c++) {
                kp.beta_mat[(size_t)g + (size_t)kp.n * c] = beta[c];
                kp.beta_var_mat[(size_t)g + (size_t)kp.n * c] = Sg[c][c];
            }
            kp.iter[g] = it;
            kp.deviance[g] = dev;
            kp.contrast_num[g] = cn;
            kp.contrast_denom[g] = __builtin_sqrt(cd);
        }
        }
        DSQ_CPROF(8);
    }
#ifdef DSQ_WIDE_PROF
    if (lane == 0) for (int q_ = 0; q_ < DSQ_PROF_SLOTS; q_++) atomicAdd(&betac_prof[q_], pacc_k[q_]);
#endif
}

template <int P>
static hipError_t launch_beta_cells(const BetaKernelParams &kp, hipStream_t st) {
    const int waves = 4;
    const size_t lds = ((size_t)DSQ_CMAX * P + beta_cell_int_doubles(kp.m) + (kp.nf_is_vector ? (size_t)kp.m : 0) +
                        waves * beta_cell_wave_doubles(kp.m, kp.useWeights != 0)) * sizeof(double);
    const void *fn = kp.useWeights ? (const void *)fit_beta_cell_kernel<P, true> : (const void *)fit_beta_cell_kernel<P, false>;
    static thread_local int bpc_cache[2];
    static thread_local size_t lds_cache[2];
    DSQ_CACHE_PER_DEVICE(bpc_cache, lds_cache);
    const int wi = kp.useWeights ? 1 : 0;
    if (lds_cache[wi] != lds || bpc_cache[wi] == 0) {
        if (lds > 64 * 1024) (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        int bpc = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, fn, 64 * waves, lds) != hipSuccess || bpc < 1) bpc = 1;
        bpc_cache[wi] = bpc; lds_cache[wi] = lds;
        if (getenv("DSQ_VERBOSE")) fprintf(stderr, "[dsq] fit_beta_cells<P=%d> lds=%zu occupancy-api blocks/CU=%d\n", P, lds, bpc);
    }
    const int cus = device_cu_count();
    int blocks_needed = (kp.n + waves - 1) / waves;
    int grid = blocks_needed < cus * bpc_cache[wi] ? blocks_needed : cus * bpc_cache[wi];
    if (kp.rows_few && grid > cus) grid = cus;
    if (grid < 1) grid = 1;
#ifdef DSQ_WIDE_PROF
    unsigned long long pz[DSQ_PROF_SLOTS] = {}, ph[DSQ_PROF_SLOTS];
    (void)hipMemcpyToSymbol(HIP_SYMBOL(betac_prof), pz, sizeof(pz));
#endif
    if (kp.useWeights) hipLaunchKernelGGL((fit_beta_cell_kernel<P, true>), dim3(grid), dim3(64 * waves), lds, st, kp);
    else hipLaunchKernelGGL((fit_beta_cell_kernel<P, false>), dim3(grid), dim3(64 * waves), lds, st, kp);
#ifdef DSQ_WIDE_PROF
    if (kp.n >= 1000) {
        (void)hipStreamSynchronize(st);
        (void)hipMemcpyFromSymbol(ph, HIP_SYMBOL(betac_prof), sizeof(ph));
        double tot = 0;
        for (int q = 0; q < DSQ_PROF_SLOTS; q++) tot += (double)ph[q];
        static const char *nm[9] = {"start + IRLS constants", "sample sweeps (+ closes, deviance)", "collapsed least squares", "cell eta / exp", "convergence control",
                                    "post-loop: X'WX and its inverse", "post-loop: hat / mu output loop", "next gene", "post-loop: sigma, contrast, results"};
        fprintf(stderr, "[betac_prof] p=%d m=%d n=%d cells=%d maxit=%d:", P, kp.m, kp.n, kp.ncell, kp.maxit);
        for (int q = 0; q < 9; q++) fprintf(stderr, " %s %.1f%%", nm[q], 100.0 * (double)ph[q] / (tot > 0 ? tot : 1));
        fprintf(stderr, "  (%.0f Mcycles over all waves)\n", tot / 1e6);
    }
#endif
    return hipGetLastError();
}

// ---- rows the IRLS did not fit (fitNbinomGLMsOptim, R/fitNbinomGLMs.R:340-407) ---------------------------------------
// Damped Fisher scoring on the penalised NB log posterior over beta in [-30, 30]^p (log2 scale), one wavefront per
// row; the test suite's CPU checker states the same iteration operation for operation.  A handful of rows
// per analysis: no staging, every pass re-reads the row through L2.
// Wide builds (p > 10, zero-padded to 16 / 24 / 32 / 48 columns: a padded coefficient has a zero design column, ridge 1 and start
// value 0, so it stays exactly 0 and adds exact zeros to everything else): the wave-uniform p x p work lives once per
// wave in LDS (DSQ_BWORK) and the Gram matrices are accumulated two rows per pass (beta_gram_wide) -- same terms, same
// order per entry as the one-pass form.
template <int P, bool USE_W>
__global__ void __launch_bounds__(64) optim_rows_kernel(OptimKernelParams kp) {
#if DSQ_P >= DSQ_BETA_ARENA_MIN
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double *arena = smem;
    int arena_off = 0;
#endif
    const int lane = threadIdx.x;
    const int m = kp.m;
    constexpr int N = SymNB<P>::value;
    const double ln2 = 0.6931471805599453, log2e = 1.4426950408889634;
    const double bound = 30.0 * ln2;
    const int nwork = DSQ_NWORK(kp);
    DSQ_BWORK(DsqVecP, lam);
    DSQ_BWORK(DsqVecP, gam);
    DSQ_BWORK(DsqVecP, trial);
    DSQ_BMARK(row_mark);
    for (int wi = blockIdx.x; wi < nwork; wi += gridDim.x) {
        DSQ_BRESET(row_mark);
        const int g = DSQ_GENE(kp, wi);
        const int32_t *yg = kp.y + (size_t)g * kp.ld;
        const double *nfg = kp.nf_is_vector ? kp.nf : kp.nf + (size_t)g * kp.ld;
        const double *wg = USE_W ? kp.weights + (size_t)g * kp.ld : nullptr;
        const double alpha = kp.alpha_hat[g], size = 1.0 / alpha;
DSQ_UNROLL_P
        for (int c = 0; c < P; c++) {
            lam[c] = kp.lamnat[c];
            gam[c] = __builtin_fmin(__builtin_fmax(kp.beta_start[(size_t)g + (size_t)kp.n * c] * ln2, -bound), bound);
        }
        auto eta_of = [&](const double (&b)[P], int j) {
            double eta = kp.x[j] * b[0];
DSQ_UNROLL_P
            for (int c = 1; c < P; c++) eta = __builtin_fma(kp.x[(size_t)c * m + j], b[c], eta);
            return eta;
        };
        auto objective = [&](const double (&b)[P]) {
            double acc = 0.0;
            for (int j = lane; j < m; j += 64) {
                double d = dnbinom_mu_log((double)yg[j], size, nfg[j] * dexp(eta_of(b, j)));
                if constexpr (USE_W) d = wg[j] * d;
                acc += d;
            }
            const double ll = wave_allreduce(acc);
            double pen = 0.0;
DSQ_UNROLL_P
            for (int c = 0; c < P; c++) pen = __builtin_fma(0.5 * lam[c], b[c] * b[c], pen);
            const double f = pen - ll;
            return dfinite(f) ? f : 1e300;
        };
        // the scoring weights of a sample at gam: w = [wts] mu / (1 + alpha mu), r = [wts] (y - mu) / (1 + alpha mu)
        auto score_terms = [&](int j, double &wv, double &rv) {
            const double mu = nfg[j] * dexp(eta_of(gam, j));
            if constexpr (USE_W) { wv = (wg[j] * mu) / (1.0 + alpha * mu); rv = (wg[j] * ((double)yg[j] - mu)) / (1.0 + alpha * mu); }
            else { wv = mu / (1.0 + alpha * mu); rv = ((double)yg[j] - mu) / (1.0 + alpha * mu); }
        };
        double F = objective(gam);
        int converged = 0;
        DSQ_BMARK(iter_mark);
        for (int it = 0; it < 100 && !converged; it++) {
            DSQ_BRESET(iter_mark);
            DSQ_BWORK(LU<P>, lu);
            DSQ_BWORK(DsqVecP, rhs);
            if constexpr (P >= DSQ_WIDE_MIN) {
                beta_gram_wide<P, true>(kp.x, m, lane, score_terms, lu.a, rhs);
DSQ_UNROLL_P
                for (int a = 0; a < P; a++) { lu.a[a][a] = lu.a[a][a] + lam[a]; rhs[a] = rhs[a] - lam[a] * gam[a]; }
            } else {
                double acc[N + P];
DSQ_UNROLL_P
                for (int i = 0; i < N + P; i++) acc[i] = 0.0;
                for (int j = lane; j < m; j += 64) {
                    double wv, rv;
                    score_terms(j, wv, rv);
                    double xr[P];
DSQ_UNROLL_P
                    for (int c = 0; c < P; c++) xr[c] = kp.x[(size_t)c * m + j];
                    int idx = 0;
DSQ_UNROLL_P
                    for (int a = 0; a < P; a++) {
DSQ_UNROLL_P
                        for (int b = a; b < P; b++) acc[idx++] += xr[a] * (xr[b] * wv);
                        acc[N + a] += xr[a] * rv;
                    }
                }
                wave_allreduce_many(acc, lane);
                int idx = 0;
DSQ_UNROLL_P
                for (int a = 0; a < P; a++)
DSQ_UNROLL_P
                    for (int b = a; b < P; b++) { lu.a[a][b] = acc[idx]; lu.a[b][a] = acc[idx]; idx++; }
DSQ_UNROLL_P
                for (int a = 0; a < P; a++) { lu.a[a][a] = lu.a[a][a] + lam[a]; rhs[a] = acc[N + a] - lam[a] * gam[a]; }
            }
            lu.factor();
            lu.solve(rhs);
            double t = 1.0, Ft = 0.0;
            int accepted = 0;
            for (int h = 0; h < 40; h++) {
DSQ_UNROLL_P
                for (int c = 0; c < P; c++) trial[c] = __builtin_fmin(__builtin_fmax(gam[c] + t * rhs[c], -bound), bound);
                Ft = objective(trial);
                if (uniform(Ft < F)) { accepted = 1; break; }
                t = t * 0.5;
            }
            if (!accepted) { converged = 1; break; }
            const double dec = F - Ft;
DSQ_UNROLL_P
            for (int c = 0; c < P; c++) gam[c] = trial[c];
            F = Ft;
            if (uniform(dec <= 1e-10 * (__builtin_fabs(F) + 1e-10))) converged = 1;
        }
        // (:382-400)
        DSQ_BRESET(iter_mark);
        DSQ_BWORK(DsqMatP, G);
        DSQ_BWORK(DsqMatP, Gi);
        DSQ_BWORK(DsqMatP, Sg);
        DSQ_BMARK(lu_mark);                      // (the LU below and T after it share this space: 64 columns fit a CU's LDS)
        // pmax(mu, minmu) enters the covariance and the log likelihood; the reported mean keeps the caller's floor
        auto final_w = [&](int j, double muc) {
            if constexpr (USE_W) return wg[j] / (1.0 / muc + alpha);
            else return 1.0 / (1.0 / muc + alpha);
        };
        double lacc = 0.0;
        if constexpr (P >= DSQ_WIDE_MIN) {
            for (int j = lane; j < m; j += 64) {
                const double mu = nfg[j] * dexp(eta_of(gam, j));
                kp.mu_out[(size_t)g * kp.ld + j] = (kp.mu_floor > 0.0 && mu < kp.mu_floor) ? kp.mu_floor : mu;
                double d = dnbinom_mu_log((double)yg[j], size, __builtin_fmax(mu, kp.minmu));
                if constexpr (USE_W) d = wg[j] * d;
                lacc += d;
            }
            beta_gram_wide<P, false>(kp.x, m, lane, [&](int j, double &wv, double &zw) {
                const double mu = nfg[j] * dexp(eta_of(gam, j));
                wv = final_w(j, __builtin_fmax(mu, kp.minmu));
                zw = 0.0;
            }, G, nullptr);
        } else {
            double gacc[N];
DSQ_UNROLL_P
            for (int i = 0; i < N; i++) gacc[i] = 0.0;
            for (int j = lane; j < m; j += 64) {
                const double mu = nfg[j] * dexp(eta_of(gam, j));
                // pmax(mu, mu_floor) as numpy.maximum / R's `[<-` on a comparison leave a NaN in place
                kp.mu_out[(size_t)g * kp.ld + j] = (kp.mu_floor > 0.0 && mu < kp.mu_floor) ? kp.mu_floor : mu;
                const double muc = __builtin_fmax(mu, kp.minmu);
                const double wv = final_w(j, muc);
                double xr[P];
DSQ_UNROLL_P
                for (int c = 0; c < P; c++) xr[c] = kp.x[(size_t)c * m + j];
                int idx = 0;
DSQ_UNROLL_P
                for (int a = 0; a < P; a++)
DSQ_UNROLL_P
                    for (int b = a; b < P; b++) gacc[idx++] += xr[a] * (xr[b] * wv);
                double d = dnbinom_mu_log((double)yg[j], size, muc);
                if constexpr (USE_W) d = wg[j] * d;
                lacc += d;
            }
            wave_allreduce_many(gacc, lane);
            int idx = 0;
DSQ_UNROLL_P
            for (int a = 0; a < P; a++)
DSQ_UNROLL_P
                for (int b = a; b < P; b++) { G[a][b] = gacc[idx]; G[b][a] = gacc[idx]; idx++; }
        }
        const double ll = wave_allreduce(lacc);
        {
            DSQ_BWORK(LU<P>, lu);
DSQ_UNROLL_P
            for (int a = 0; a < P; a++)
DSQ_UNROLL_P
                for (int b = 0; b < P; b++) lu.a[a][b] = G[a][b];
DSQ_UNROLL_P
            for (int a = 0; a < P; a++) lu.a[a][a] = lu.a[a][a] + lam[a];
            lu.factor();
            lu.inverse(Gi);
        }
        DSQ_BRESET(lu_mark);
        DSQ_BWORK(DsqMatP, T);
        mat_mul<P>(Gi, G, T);
        mat_mul<P>(T, Gi, Sg);
        if (lane == 0) {
DSQ_UNROLL_P
            for (int c = 0; c < P; c++) {
                kp.beta[(size_t)g + (size_t)kp.n * c] = log2e * gam[c];
                kp.betaSE[(size_t)g + (size_t)kp.n * c] = log2e * __builtin_sqrt(__builtin_fmax(Sg[c][c], 0.0));
            }
            kp.conv[g] = converged;
            kp.loglike[g] = ll;
        }
    }
}

// LDS of one single-wave block of the wide build: lam, gam, trial | LU, rhs | G, Gi, Sg, the final LU (then T in its place)
__host__ __device__ static inline size_t optim_arena_doubles(int p) {
    return p >= DSQ_BETA_ARENA_MIN ? (size_t)4 * p * p + 12 * p + 32 : 0;
}

template <>
hipError_t launch_optim_p<DSQ_P>(const OptimKernelParams &kp, hipStream_t st) {
    int grid = kp.n < 4096 ? kp.n : 4096;
    if (kp.rows && grid > 256) grid = 256;       // a row list (its length lives on the device): a handful of rows
    if (grid < 1) grid = 1;
    const size_t lds = optim_arena_doubles(DSQ_P) * sizeof(double);
    if (kp.useWeights) hipLaunchKernelGGL((optim_rows_kernel<DSQ_P, true>), dim3(grid), dim3(64), lds, st, kp);
    else hipLaunchKernelGGL((optim_rows_kernel<DSQ_P, false>), dim3(grid), dim3(64), lds, st, kp);
    return hipGetLastError();
}

#if DSQ_P < DSQ_BETA_ROLLED_MIN
// ---- launch ---------------------------------------------------------------------
// Geometry: W waves (genes) per block share the LDS copy of X; the grid is persistent
// (blocks-per-CU x CUs, grid-stride over genes) so the per-wave scratch slabs stay L2-resident.
static inline size_t beta_lds_doubles(int m, int p, int waves, int xlds, int qrm = 0) {
    return (xlds ? (size_t)p * m : 0) + (size_t)waves * m * kSlabVecs + (size_t)waves * beta_arena_doubles(p) +
           (qrm ? (size_t)waves * beta_qr_doubles(m, p, qrm) : 0);
}

template <int P>
static void beta_geometry(int n, int m, bool useW, int *waves, bool *stage, int *xlds, int *grid, size_t *lds,
                          bool want_qrrows = false, int *qrmode = nullptr) {
    const Tuning &tu = tuning();
    // Pick (waves per block, X in LDS?) maximising resident waves per CU: 160 KiB of LDS per CU, and
    // the register budget of these kernels admits 2 waves per SIMD = 8 per CU.  Ties: bigger blocks
    // (X shared by more genes), then X in LDS.
    const size_t budget = (size_t)tu.beta_lds_kb * 1024, cu_lds = 160 * 1024;
    const int wmax = tu.beta_waves >= 4 ? 4 : tu.beta_waves >= 2 ? 2 : tu.beta_waves == 1 ? 1 : 4;
    int best = -1, best_wpc = 0;
    *stage = false; *waves = wmax; *xlds = 0;
    // stored-row QR: taken when its LDS leaves at least 4 waves on a CU (the replay kernel runs 4 at p >= 7)
    bool qr = false;
    int qrm = 0;
    // rows in registers (QRM 2): whenever the least squares has at most 64 T rows -- the LDS then admits the register budget's
    // resident waves at every width
    static const int qrreg_env = getenv("DSQ_BETA_QRREG") ? atoi(getenv("DSQ_BETA_QRREG")) : 1;
    const int reg_trips = m + P <= 64 * DSQ_BETA_QRREG_TRIPS ? 1 : (P <= DSQ_BETA_QRREG2_MAXP && qrreg_env != 2 && m + P <= 128 * DSQ_BETA_QRREG_TRIPS) ? 2 : 0;
    if (want_qrrows && qrreg_env && P >= DSQ_BETA_QRREG_MIN && reg_trips && tu.beta_stage != 0) {
        const int regw = reg_trips == 2 ? 1 : DSQ_BETA_QRREG_MINW;
        int qbest = -1, qw = 0, qx = 0;
        for (int xl = tu.beta_xlds ? 1 : 0; xl >= 0; xl--)
            for (int w = wmax; w >= 1; w >>= 1) {
                size_t need = beta_lds_doubles(m, P, w, xl, 2) * sizeof(double);
                if (need > budget) continue;
                int wpc = w * (int)(cu_lds / need);
                if (wpc > 4 * regw) wpc = 4 * regw;
                int score = wpc * 100 + w * 2 + xl;
                if (score > qbest) { qbest = score; qw = w; qx = xl; }
            }
        if (qbest >= 0) { qr = true; qrm = 1 + reg_trips; *stage = true; *waves = qw; *xlds = qx; }
    }
    if (!qr && want_qrrows && P >= DSQ_BETA_QRROWS_MIN && tu.beta_stage != 0) {
        int qbest = -1, qw = 0, qx = 0;
        for (int xl = tu.beta_xlds ? 1 : 0; xl >= 0; xl--)
            for (int w = wmax; w >= 1; w >>= 1) {
                size_t need = beta_lds_doubles(m, P, w, xl, 1) * sizeof(double);
                if (need > budget) continue;
                int wpc = w * (int)(cu_lds / need);
                if (wpc > 8) wpc = 8;
                int score = wpc * 100 + w * 2 + xl;
                // (the wide builds: replay costs p^3 per row and IRLS step against p^2 with the rows stored -- at p = 24 even
                //  two resident waves per CU win by an order of magnitude; DSQ_BETA_QRROWS_MINWPC overrides)
                static const int min_wpc_env = getenv("DSQ_BETA_QRROWS_MINWPC") ? atoi(getenv("DSQ_BETA_QRROWS_MINWPC")) : 0;
                const int min_wpc = min_wpc_env > 0 ? min_wpc_env : (P >= 16 ? 1 : 4);
                if (wpc >= min_wpc && score > qbest) { qbest = score; qw = w; qx = xl; }
            }
        if (qbest >= 0) { qr = true; qrm = 1; *stage = true; *waves = qw; *xlds = qx; }
    }
    if (qrmode) *qrmode = qrm;
    if (!qr)
    for (int xl = tu.beta_xlds ? 1 : 0; xl >= 0; xl--)
        for (int w = wmax; w >= 1; w >>= 1) {
            size_t need = beta_lds_doubles(m, P, w, xl) * sizeof(double);
            if (need > budget) continue;
            int blocks = (int)(cu_lds / need);
            int wpc = w * blocks < 8 ? w * blocks : 8;
            int score = wpc * 100 + w * 2 + xl;
            if (score > best) { best = score; best_wpc = wpc; *stage = true; *waves = w; *xlds = xl; }
        }
    // Long rows (m >~ 1000): the LDS slabs leave fewer than 6 waves per CU and staging loses to plain L2-resident
    // rows at full occupancy (measured, p = 4: m = 1250 10.6 vs 9.1 ms, m = 2000 18.5 vs 9.1 ms; m = 800 5.1 vs 5.6).
    if (!qr && *stage && best_wpc < 6 && tu.beta_stage < 0) { *stage = false; *waves = wmax; }
    if (tu.beta_stage == 0) *stage = false;
    if (!*stage) *xlds = 0;
    if (!*stage)
        while (*waves > 1 && (size_t)*waves * beta_arena_doubles(P) * sizeof(double) > budget) *waves >>= 1;
    *lds = *stage ? beta_lds_doubles(m, P, *waves, *xlds, qrm) * sizeof(double)
                  : (size_t)*waves * beta_arena_doubles(P) * sizeof(double);     // unstaged: only the WIDE arena
    static thread_local int bpc_cache[5][2][8];   // [stage + stored rows][useW][waves]: the occupancy query costs ~1 ms, ask once
    static thread_local size_t lds_cache[5][2][8];
    DSQ_CACHE_PER_DEVICE(bpc_cache, lds_cache);
    const int ci = qr ? 1 + qrm : (*stage ? 1 : 0);
    if (lds_cache[ci][useW][*waves] != *lds) { bpc_cache[ci][useW][*waves] = 0; lds_cache[ci][useW][*waves] = *lds; }
    int bpc = bpc_cache[ci][useW][*waves];
    constexpr int kQreg2 = (P >= DSQ_BETA_QRREG_MIN && P <= DSQ_BETA_QRREG2_MAXP) ? 3 : 0;
    const void *fn = qrm == 3 ? (useW ? (const void *)fit_beta_kernel<P, true, true, kQreg2> : (const void *)fit_beta_kernel<P, false, true, kQreg2>)
                  : qrm == 2 ? (useW ? (const void *)fit_beta_kernel<P, true, true, (P >= DSQ_BETA_QRREG_MIN ? 2 : 0)>
                                      : (const void *)fit_beta_kernel<P, false, true, (P >= DSQ_BETA_QRREG_MIN ? 2 : 0)>)
                  : qr ? (useW ? (const void *)fit_beta_kernel<P, true, true, (P >= DSQ_BETA_QRROWS_MIN ? 1 : 0)>
                               : (const void *)fit_beta_kernel<P, false, true, (P >= DSQ_BETA_QRROWS_MIN ? 1 : 0)>)
                  : *stage ? (useW ? (const void *)fit_beta_kernel<P, true, true, 0> : (const void *)fit_beta_kernel<P, false, true, 0>)
                           : (useW ? (const void *)fit_beta_kernel<P, true, false, 0> : (const void *)fit_beta_kernel<P, false, false, 0>);
    if (bpc == 0) {
        if (*lds > 64 * 1024) (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)*lds);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, fn, 64 * *waves, *lds) != hipSuccess || bpc < 1) bpc = 1;
        bpc_cache[ci][useW][*waves] = bpc;
        if (getenv("DSQ_VERBOSE")) fprintf(stderr, "[dsq] fit_beta<P=%d> waves=%d stage=%d stored-rows=%d xlds=%d lds=%zu occupancy-api blocks/CU=%d\n", P, *waves, (int)*stage, qrm, *xlds, *lds, bpc);
    }
    if (tu.beta_bpc > 0) bpc = tu.beta_bpc;
    const int cus = device_cu_count();
    int blocks_needed = (n + *waves - 1) / *waves;
    int cap = cus * bpc;
    *grid = blocks_needed < cap ? blocks_needed : cap;
    if (*grid < 1) *grid = 1;
}

#endif   // DSQ_P < DSQ_BETA_ROLLED_MIN

#ifndef DSQ_P
#error "compile with -DDSQ_P=<number of design columns>"
#endif

// narrowest width whose designs WITHOUT cells take the rolled kernel of fit_beta_wide.hip instead of this file's general
// kernel (DSQ_ROLLED_MINP; the wide builds always do)
static inline int rolled_minp() {
    static const int v = getenv("DSQ_ROLLED_MINP") ? atoi(getenv("DSQ_ROLLED_MINP")) : DSQ_BETA_ROLLED_MIN;
    return v;
}

template <>
void fit_beta_scratch_doubles<DSQ_P>(int n, int m, int useW, size_t *slab, size_t *cscr) {
#if DSQ_P >= DSQ_BETA_ROLLED_MIN
    fit_beta_rolled_scratch_doubles(n, m, DSQ_P, useW, slab, cscr);
#else
    int waves, grid, xlds;
    bool stage;
    size_t lds;
    beta_geometry<DSQ_P>(n, m, useW != 0, &waves, &stage, &xlds, &grid, &lds);
    *slab = stage ? 0 : (size_t)grid * waves * (size_t)m * kSlabVecs;
    *cscr = 0;      // (the hoisted NB-density constants are gone: the deviance needs no per-sample scratch row)
    if (DSQ_P >= rolled_minp()) {        // (the general designs of this width on the rolled kernel: the larger of the two)
        size_t s2 = 0, c2 = 0;
        fit_beta_rolled_scratch_doubles(n, m, DSQ_P, useW, &s2, &c2);
        if (s2 > *slab) *slab = s2;
    }
#endif
}

template <>
hipError_t launch_fit_beta_p<DSQ_P>(const BetaKernelParams &kp0, hipStream_t st) {
    if constexpr (DSQ_P <= DSQ_SPEC_BETA_CELL_MAXP)
        if (kp0.ncell > 0 && kp0.ncell <= DSQ_CMAX && kp0.ncell + DSQ_P <= 64) return launch_beta_cells<DSQ_P>(kp0, st);
#if DSQ_P >= DSQ_BETA_ROLLED_MIN
    return launch_fit_beta_rolled(kp0, st);
#else
    if (DSQ_P >= rolled_minp()) return launch_fit_beta_rolled(kp0, st);
    int waves, grid, xlds;
    bool stage;
    size_t lds;
    int qr = 0;
    beta_geometry<DSQ_P>(kp0.n, kp0.m, kp0.useWeights != 0, &waves, &stage, &xlds, &grid, &lds,
                         kp0.useQR != 0 && kp0.maxit > 0, &qr);
    BetaKernelParams kp = kp0;
    kp.xlds = xlds;
    if (kp.rows_few && grid > device_cu_count()) grid = device_cu_count();   // a row list: its length lives on the device
    constexpr int kQr = (DSQ_P >= DSQ_BETA_QRROWS_MIN) ? 1 : 0, kQreg = (DSQ_P >= DSQ_BETA_QRREG_MIN) ? 2 : 0;
    constexpr int kQreg2 = (DSQ_P >= DSQ_BETA_QRREG_MIN && DSQ_P <= DSQ_BETA_QRREG2_MAXP) ? 3 : 0;
    if (qr == 3) {
        if (kp.useWeights)
            hipLaunchKernelGGL((fit_beta_kernel<DSQ_P, true, true, kQreg2>), dim3(grid), dim3(64 * waves), lds, st, kp);
        else
            hipLaunchKernelGGL((fit_beta_kernel<DSQ_P, false, true, kQreg2>), dim3(grid), dim3(64 * waves), lds, st, kp);
    } else if (qr == 2) {
        if (kp.useWeights)
            hipLaunchKernelGGL((fit_beta_kernel<DSQ_P, true, true, kQreg>), dim3(grid), dim3(64 * waves), lds, st, kp);
        else
            hipLaunchKernelGGL((fit_beta_kernel<DSQ_P, false, true, kQreg>), dim3(grid), dim3(64 * waves), lds, st, kp);
    } else if (qr == 1) {
        if (kp.useWeights)
            hipLaunchKernelGGL((fit_beta_kernel<DSQ_P, true, true, kQr>), dim3(grid), dim3(64 * waves), lds, st, kp);
        else
            hipLaunchKernelGGL((fit_beta_kernel<DSQ_P, false, true, kQr>), dim3(grid), dim3(64 * waves), lds, st, kp);
    } else if (stage) {
        if (kp.useWeights)
            hipLaunchKernelGGL((fit_beta_kernel<DSQ_P, true, true, 0>), dim3(grid), dim3(64 * waves), lds, st, kp);
        else
            hipLaunchKernelGGL((fit_beta_kernel<DSQ_P, false, true, 0>), dim3(grid), dim3(64 * waves), lds, st, kp);
    } else {
        if (kp.useWeights)
            hipLaunchKernelGGL((fit_beta_kernel<DSQ_P, true, false, 0>), dim3(grid), dim3(64 * waves), lds, st, kp);
        else
            hipLaunchKernelGGL((fit_beta_kernel<DSQ_P, false, false, 0>), dim3(grid), dim3(64 * waves), lds, st, kp);
    }
    return hipGetLastError();
#endif
}

}  // namespace dsq
