// fit_beta_wide.hip -- fitBeta (src/DESeq2.cpp:283-465) for designs of 11 .. 64 columns that do not collapse to design
// cells (paired designs such as ~ patient + treatment, many continuous covariates): ONE kernel for every width, all
// loops over the design columns ROLLED, one wavefront per gene.
//
// Why another kernel: the per-width general kernel (fit_beta.hip) is compiled once per padded width (16, 24, 32, 48) with
// its Householder stages fully unrolled; at those widths its wave-uniform p x p state sits in an LDS arena of 5 p^2 + 12 p
// doubles (41 KB at p = 32, 92 KB at p = 48) next to the stored rows -- one or two waves per CU --, the two widest builds
// compile for minutes at -O1, and a paired design of 31 columns cost 259 ms per 20 000 genes (profiles/r05_wide.txt).
// Here the reflected rows [sqrt(w) X ; sqrt(ridge) | sqrt(w) z] and the p x p matrices of the post-loop block live in a
// per-wave slab of GLOBAL memory (coalesced: lane l owns rows l, l + 64, ... of a column-major matrix, or column l of a
// row-major p x p matrix), the LDS holds only a handful of p-vectors, and the register footprint is that of a chunk of
// eight column sums: the occupancy is the register file's, whatever p is.
//
// THE ARITHMETIC IS THAT OF fit_beta_kernel (stored-row Householder QR in LAPACK dgeqr2 order, Gram sums in wave order,
// LU with partial pivoting / first maximum / reciprocal pivots, the closed split of the deviance): the same operations on
// the same values in the same order, so the results keep the oracle's bits (tests/test_gpu_wide.py).  What changes is
// the loop nest: a sum over the samples is still 64 per-lane partials over the trips in order + the xor butterfly, but the
// sums of a Householder stage are taken eight columns at a time (each sum's additions do not depend on its neighbours').
#include "dsq_internal.hpp"
#include <cstdio>
#include <cstdlib>
#include "dsq_math.hpp"
#include "dsq_wave.hpp"
#include "fit_beta_common.hpp"
#include "dsq_prof.hpp"

namespace dsq {

#ifdef DSQ_WIDE_PROF
__device__ unsigned long long betaw_prof[DSQ_PROF_SLOTS];
#endif

// per-wave slab in global memory (doubles): the four per-sample vectors, then the larger of
//   IRLS      the (m + p) x (p + 1) rows (column c of row i at qa[c M + i]) and R (p x p)
//   post-loop G, LU (then T), Gi, Sigma (p x p each, row-major)
__host__ __device__ static inline size_t wide_slab_doubles(int m, int p) {
    const size_t M = (size_t)m + p;
    const size_t irls = M * (p + 1) + (size_t)p * p, post = (size_t)4 * p * p;
    return (size_t)4 * m + (irls > post ? irls : post) + 8;
}
// per-wave LDS (doubles): lambda, contrast, beta, beta_prev, gamma, rdiag, rhs / rr (p each), tprev, accs (p + 1 each), piv (p ints)
__host__ __device__ static inline size_t wide_lds_doubles(int p) { return (size_t)11 * p + 24; }     // (+ the control words)

constexpr int kChunk = 8;     // column sums reduced together (wave_allreduce_many: the bits of one butterfly each)
template <int V> struct IntTag { static constexpr int value = V; };

// ONE WORKGROUP OF NW WAVES PER GENE.  A gene of a wide design is 10^5 .. 10^6 wave instructions, most of them in the
// Householder stages, and its slab leaves room for only a few genes per CU: with a wave per gene the CU ran one wave per
// SIMD (or fewer), every instruction exposed to its own latency.  The NW waves share the gene's slab and split what is
// independent -- the column chunks of a Householder stage (after the first, which updates the multiplier column), the
// (row, column-chunk) sums of the Gram matrices, the rows of an elimination step, the samples of the elementwise passes,
// the rows of the p x p products -- and meet at workgroup barriers.  EVERY SUM OVER THE SAMPLES IS STILL TAKEN BY ONE
// WAVE in wave order (64 per-lane partials over the trips in order, then the butterfly): which wave takes it does not
// enter the result.  Serial recurrences (back substitution, the triangular solves of the inverse, the convergence
// control) stay with wave 0; their results reach the other waves through LDS.
// BIG_LDS: the gene's slab (rows, matrices, per-sample vectors) in LDS instead of global memory -- taken whenever one
// fits a CU (wide_geometry): a Householder stage is a chain of dependent round trips through the slab
// (rows -> column sums -> pivot row -> reflector), ~ 100 ns each in LDS against 1-3 us through L2 / the infinity cache.
// NW = 1, 2, 4 or 8 waves per gene: as many as bring a CU to about eight resident waves (two per SIMD: what the registers of
// this kernel admit) given how many genes' slabs fit its LDS -- small problems run a wave per gene, without barriers.
// (HIP's second launch bound is WAVES PER SIMD, not blocks per CU)
template <bool USE_W, bool BIG_LDS, int NW>
__global__ void __launch_bounds__(64 * NW, BIG_LDS ? 1 : 2) fit_beta_rolled_kernel(BetaKernelParams kp) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    constexpr int NT = 64 * NW;                // threads per gene (= per workgroup)
    // (P: the width the fit runs at -- the design's own, kp.p_true, when the launch found its slab to fit the LDS; kp.p is
    //  the padded width of the n x p arrays, whose padding columns get their outputs, 0, at the end)
    const int m = kp.m, P = kp.p_true > 0 ? kp.p_true : kp.p;
    const int M = m + P;
    const int nwork = DSQ_NWORK(kp);
    if ((int)blockIdx.x >= nwork) return;

    const double *xs = kp.x;
    double *lv = smem;
    double *lambda = lv, *contrast = lv + P, *beta = lv + 2 * P, *beta_prev = lv + 3 * P, *gamma = lv + 4 * P, *rdiag = lv + 5 * P,
           *rhs = lv + 6 * P, *tprev = lv + 7 * P, *accs = lv + 8 * P + 1;
    int *piv = reinterpret_cast<int *>(lv + 9 * P + 2);
    double *ctl = lv + 10 * P + 8;             // [0] loop control of the IRLS, [1] next gene, [2] dev, [3] iterations
    double *tprev2 = lv + 10 * P + 16;         // (the stages of the QR write tprev and tprev2 in turn)
    double *refl = lv + 11 * P + 17;           // reflector constants (bet, tau, scal) of the next stage, two sets in turn
    double *slab;
    if constexpr (BIG_LDS) slab = smem + wide_lds_doubles(P);
    else slab = kp.scratch + (size_t)blockIdx.x * wide_slab_doubles(m, P);
    double *mu_s = slab, *lg_s = slab + m, *sw_s = slab + 2 * (size_t)m, *w_s = slab + 3 * (size_t)m;
    double *big = slab + 4 * (size_t)m;
    double *qa = big, *qR = big + (size_t)M * (P + 1);                         // IRLS
    double *G = big, *LUm = big + (size_t)P * P, *Gi = big + 2 * (size_t)P * P, *Sg = big + 3 * (size_t)P * P;   // post-loop
    double *Tm = LUm;                                                         // (LU is dead once Gi exists)
    // workgroup barrier; in global-slab mode it also orders the slab's stores before the other waves' loads
    auto sync = [] {
        if constexpr (!BIG_LDS) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        else if constexpr (NW == 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        if constexpr (NW == 1) __builtin_amdgcn_wave_barrier();
        else __syncthreads();
        if constexpr (!BIG_LDS) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        else if constexpr (NW == 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };

    for (int c = tid; c < P; c += NT) { lambda[c] = kp.lambda[c]; contrast[c] = kp.contrast[c]; }
    const double large = 30.0;

    DSQ_PROF_DECL;
    int wi = blockIdx.x;
    while (wi < nwork) {
        const int g = DSQ_GENE(kp, wi);
        const int32_t *yg = kp.y + (size_t)g * kp.ld;
        const double *nfg = kp.nf_is_vector ? kp.nf : kp.nf + (size_t)g * kp.ld;
        const double *wg = USE_W ? kp.weights + (size_t)g * kp.ld : nullptr;
        const double alpha = kp.alpha_hat[g];
        const double size = 1.0 / alpha;

        sync();
        for (int c = tid; c < P; c += NT) { const double b = kp.beta_init[(size_t)g + (size_t)kp.n * c]; beta[c] = b; beta_prev[c] = b; }
        sync();

        // mu_hat = nfrow % exp(x * beta_hat), clamped at minmu            (:324-327, :361-364)      [all waves]
        auto update_mu = [&]() {
            for (int j = tid; j < m; j += NT) {
                double eta = xs[j] * beta[0];
                for (int c = 1; c < P; c++) eta = __builtin_fma(xs[(size_t)c * m + j], beta[c], eta);
                const double mu = __builtin_fmax(nfg[j] * dexp(eta), kp.minmu);
                mu_s[j] = mu;
                lg_s[j] = dlog(mu / nfg[j]);
            }
        };
        auto wvec = [&](int j, double mu) -> double {                       // (:336-342, :390-396, :430-436)
            if constexpr (USE_W) return (wg[j] * mu) / (1.0 + alpha * mu);
            else return mu / (1.0 + alpha * mu);
        };
        // G[a][b] = sum_j x_ja (x_jb w_j), b >= a, mirrored; with_rhs: rhs[a] = sum_j x_ja zw_j (zw_j in sw_s).  Each sum by
        // one wave in wave order; the (row, chunk) tasks go round the waves.                                   [all waves]
        auto gram = [&](double *Gm, bool with_rhs) {
            auto gram_chunk = [&](int a, int b0, auto rtag) __attribute__((always_inline)) {
                constexpr int R = decltype(rtag)::value;
                double acc[R];
                _Pragma("unroll")
                for (int u = 0; u < R; u++) acc[u] = 0.0;
                const double *xa_p = xs + (size_t)a * m, *xb_p = xs + (size_t)b0 * m;
                for (int j = lane; j < m; j += 64) {
                    const double wv = w_s[j], xa = xa_p[j];
                    _Pragma("unroll")
                    for (int u = 0; u < R; u++) acc[u] += xa * (xb_p[(size_t)u * m + j] * wv);
                }
                wave_allreduce_many(acc, lane);
                if (lane == 0) {
                    _Pragma("unroll")
                    for (int u = 0; u < R; u++) { Gm[(size_t)a * P + b0 + u] = acc[u]; Gm[(size_t)(b0 + u) * P + a] = acc[u]; }
                }
            };
            int task = 0;
            for (int a = 0; a < P; a++) {
                int b0 = a;
                for (; b0 + kChunk <= P; b0 += kChunk)
                    if ((task++ & (NW - 1)) == wave) gram_chunk(a, b0, IntTag<kChunk>{});
                if (b0 < P && (task++ & (NW - 1)) == wave) {
                    switch (P - b0) {
                        case 1: gram_chunk(a, b0, IntTag<1>{}); break;
                        case 2: gram_chunk(a, b0, IntTag<2>{}); break;
                        case 3: gram_chunk(a, b0, IntTag<3>{}); break;
                        case 4: gram_chunk(a, b0, IntTag<4>{}); break;
                        case 5: gram_chunk(a, b0, IntTag<5>{}); break;
                        case 6: gram_chunk(a, b0, IntTag<6>{}); break;
                        case 7: gram_chunk(a, b0, IntTag<7>{}); break;
                        default: break;
                    }
                }
                if (with_rhs && (task++ & (NW - 1)) == wave) {
                    double racc = 0.0;
                    for (int j = lane; j < m; j += 64) racc += xs[(size_t)a * m + j] * sw_s[j];
                    racc = wave_allreduce(racc);
                    if (lane == 0) rhs[a] = racc;
                }
            }
            sync();
        };
        // LU with partial pivoting of the row-major P x P matrix A in the slab (lane j owns column j), LU<P>::factor's
        // operations: first maximum wins, rows swapped in every column, reciprocal pivots, fma(-l, u, a).  Pivot search
        // and row swap by wave 0, the rows of the elimination step round the waves.                            [all waves]
        auto lu_factor = [&](double *A) {
            for (int k = 0; k < P; k++) {
                if (wave == 0) {
                    // the scan "best = |a_kk|; a later row wins when its |a_ik| > best" with a row per lane: a NaN never wins
                    // from a later row (key -1) and is never beaten in row k (key +inf); the first of equal maxima is the
                    // lowest set bit of the ballot
                    double key = -2.0;
                    if (lane >= k && lane < P) {
                        const double v = __builtin_fabs(A[(size_t)lane * P + k]);
                        key = (v != v) ? (lane == k ? __builtin_inf() : -1.0) : v;
                    }
                    double mx = key, xa, xb;
                    mx = __builtin_fmax(mx, lane_xor1(mx));
                    mx = __builtin_fmax(mx, lane_xor2(mx));
                    mx = __builtin_fmax(mx, lane_xor4(mx));
                    mx = __builtin_fmax(mx, lane_xor8(mx));
                    lane_pair16(mx, xa, xb); mx = __builtin_fmax(xa, xb);
                    lane_pair32(mx, xa, xb); mx = __builtin_fmax(xa, xb);
                    const int pr = (int)__builtin_ctzll(__ballot(key == mx));
                    if (lane == 0) piv[k] = pr;
                    if (pr != k) {
                        for (int j = lane; j < P; j += 64) {
                            const double t = A[(size_t)k * P + j];
                            A[(size_t)k * P + j] = A[(size_t)pr * P + j];
                            A[(size_t)pr * P + j] = t;
                        }
                    }
                }
                sync();
                const double rinv = 1.0 / A[(size_t)k * P + k];
                if (tid == 0) rdiag[k] = rinv;
                const double akj = lane < P ? A[(size_t)k * P + lane] : 0.0;
                const int lc = lane < P ? lane : P - 1;
                for (int i0 = k + 1 + wave; i0 < P; i0 += 4 * NW) {      // four of the wave's rows in flight
                    // (every lane loads A[i][k] BEFORE lane k's store to it: a wave's memory operations issue in program order)
                    double lv4[4], av4[4];
                    _Pragma("unroll")
                    for (int u = 0; u < 4; u++) {
                        const int i = i0 + u * NW;
                        if (i < P) { lv4[u] = A[(size_t)i * P + k]; av4[u] = A[(size_t)i * P + lc]; }
                    }
                    _Pragma("unroll")
                    for (int u = 0; u < 4; u++) {
                        const int i = i0 + u * NW;
                        if (i < P) {
                            const double l = lv4[u] * rinv;
                            if (lane == k) A[(size_t)i * P + k] = l;
                            else if (lane > k && lane < P) A[(size_t)i * P + lane] = __builtin_fma(-l, akj, av4[u]);
                        }
                    }
                }
                sync();
            }
        };
        // LU<P>::solve on ONE right-hand side in LDS                                                        [wave 0 only]
        auto lu_solve_vec = [&](const double *A, double *b) {
            // the right-hand side lives across the lanes (lane j holds b[j]), rows of the factors are read once, an entry per
            // lane: the chains fma(-a[i][j], b[j], t) take their operands from v_readlane, not from LDS round trips
            const int lc = lane < P ? lane : P - 1;
            double bv = b[lc];
            const int pv = piv[lc];
            const double rd = rdiag[lc];
            for (int k = 0; k < P; k++) {
                const int pr = __builtin_amdgcn_readlane(pv, k);
                if (pr != k) {
                    const double t = lane_read(bv, k), u = lane_read(bv, pr);
                    if (lane == k) bv = u;
                    if (lane == pr) bv = t;
                }
            }
            // (the chains walk the lanes: lane j takes the running value of lane j - 1 by one DPP shift and applies its term)
            for (int i = 1; i < P; i++) {
                const double arow = A[(size_t)i * P + lc];
                double tv = __builtin_fma(-arow, bv, lane_read(bv, i));      // (lane 0: the first term on b[i])
                for (int j = 1; j < i; j++) tv = __builtin_fma(-arow, bv, wave_shr1(tv));
                const double t = lane_read(tv, i - 1);
                if (lane == i) bv = t;
            }
            for (int i = P - 1; i >= 0; i--) {
                const double arow = A[(size_t)i * P + lc];
                double tv = bv;                                           // (lane i: b[i])
                for (int j = i + 1; j < P; j++) tv = __builtin_fma(-arow, bv, wave_shr1(tv));
                const double t = lane_read(tv, P - 1) * lane_read(rd, i);
                if (lane == i) bv = t;
            }
            wave_lds_sync();
            if (lane < P) b[lane] = bv;
            wave_lds_sync();
        };

        update_mu();
        sync();
        const bool fast = (alpha > 0.0) && dfinite(alpha) && dfinite(size) && (size > 0.0);
        double K = 0.0, Kp = 0.0;
        if (wave == 0) {
            if (kp.maxit > 0) K = irls_constants<USE_W>(yg, nfg, wg, m, lane, alpha, size, fast, nullptr, kp.kconst_out ? &Kp : nullptr);
            if (kp.kconst_out && lane == 0) kp.kconst_out[g] = Kp;
        }
        double dev = 0.0, dev_old = 0.0;          // (wave 0's)
        double it = 0.0;
        DSQ_PROF(0);
        for (int t = 0; t < kp.maxit; t++) {
            it += 1.0;
            for (int c = tid; c < P; c += NT) beta_prev[c] = beta[c];
            if (kp.useQR) {
                // pass A: the rows of the least squares (column P = sqrt(w) z)                              (:336-353)
                for (int i = tid; i < M; i += NT) {
                    if (i < m) {
                        const double mu = mu_s[i];
                        const double sw = __builtin_sqrt(wvec(i, mu));
                        const double z = lg_s[i] + ((double)yg[i] - mu) / mu;
                        sw_s[i] = sw;
                        for (int c = 0; c < P; c++) qa[(size_t)c * M + i] = xs[(size_t)c * m + i] * sw;
                        qa[(size_t)P * M + i] = z * sw;
                    } else {
                        for (int c = 0; c < P; c++) qa[(size_t)c * M + i] = (i - m == c) ? __builtin_sqrt(lambda[c]) : 0.0;
                        qa[(size_t)P * M + i] = 0.0;
                    }
                }
                sync();
                DSQ_PROF(1);
                // pass B: Householder QR, LAPACK dgeqr2 order, ONE barrier per stage.  Entering stage k, column k already carries
                // reflection k - 1 and accs[k] holds the sum of squares below its diagonal (the look-ahead of stage k - 1; for
                // k = 0 the pass in front of the loop).  Every wave derives reflection k from them; the columns k + 1 .. P then go
                // round the waves in chunks -- a chunk applies reflection k - 1 to its columns on the way, takes their sums with
                // column k, and turns them into the stage's tprev / row k of R itself (its sums never leave the registers).
                // Column k + 1 is a task of its own: its wave goes on to apply reflection k to it and to take ITS sum of squares
                // -- what stage k + 1 starts from -- while the other waves are still in their chunks.  The operations on every
                // element, and the order of every sum, are those of the stage-by-stage statement (one wave's wave-order sums).
                {
                    double a0[1] = {0.0};
                    if (wave == 0) {
                        const double *c0 = qa;
                        for (int i = lane; i < M; i += 64) { const double a = c0[i]; a0[0] += (i > 0) ? a * a : 0.0; }
                        wave_allreduce_many(a0, lane);
                        if (lane == 0) accs[0] = a0[0];
                    }
                }
                sync();
                double scal_prev = 0.0;
                for (int k = 0; k < P; k++) {
                    const double *tp_cur = (k & 1) ? tprev2 : tprev;       // written by stage k - 1
                    double *tp_next = (k & 1) ? tprev : tprev2;
                    // reflection k's constants: stage 0 derives them here; later stages read what the look-ahead wave left
                    auto reflector = [&](double alpha_k, double acck, double &bet, double &tau, double &scal) __attribute__((always_inline)) {
                        if (acck == 0.0) { tau = 0.0; scal = 0.0; bet = alpha_k; }
                        else {
                            bet = -__builtin_copysign(__builtin_sqrt(alpha_k * alpha_k + acck), alpha_k);
                            tau = (bet - alpha_k) / bet;
                            scal = 1.0 / (alpha_k - bet);
                        }
                    };
                    double tau, scal, bet;
                    if (k == 0) reflector(qa[0], accs[0], bet, tau, scal);
                    else { const double *rc = refl + 3 * (k & 1); bet = rc[0]; tau = rc[1]; scal = rc[2]; }
                    if (tid == 0) qR[(size_t)k * P + k] = bet;
                    const int i0 = lane + 64 * (k / 64);                 // (trips whose rows are all finished rows of R: skipped)
                    const double *prevp = qa + (size_t)(k > 0 ? k - 1 : 0) * M, *kcol = qa + (size_t)k * M;
                    // columns j0 .. j0 + R - 1 (j0 > k) of the rows from k down
                    auto chunk_t = [&](int j0, auto rtag, auto ptag, double (&tpn)[decltype(rtag)::value]) __attribute__((always_inline)) {
                        constexpr int R = decltype(rtag)::value;
                        constexpr bool HASPREV = decltype(ptag)::value != 0;     // k > 0: reflection k - 1 is applied on the way
                        double acc[R], tp[R], rowk[R];
                        _Pragma("unroll")
                        for (int u = 0; u < R; u++) { acc[u] = 0.0; rowk[u] = 0.0; tp[u] = HASPREV ? tp_cur[j0 + u] : 0.0; }
                        double *colp = qa + (size_t)j0 * M;
                        auto row = [&](int i, auto ctag) __attribute__((always_inline)) {
                            if (i >= k && i < M) {                        // (rows above k: finished rows of R, first live trip only)
                                const double v = HASPREV ? prevp[i] * scal_prev : 0.0;
                                const double ak = kcol[i];
                                const bool below = i > k;                 // (row k itself is the pivot row: no term)
                                _Pragma("unroll")
                                for (int u = 0; u < R; u++) {
                                    double a = colp[(size_t)u * M + i];
                                    if constexpr (HASPREV) { a = __builtin_fma(v, tp[u], a); colp[(size_t)u * M + i] = a; }
                                    if constexpr (decltype(ctag)::value != 0) rowk[u] = a;      // (first trip: lane k % 64 holds row k)
                                    // (a running sum that starts at +0.0 is never -0.0: adding +0.0 for the pivot row changes no bit)
                                    acc[u] += below ? ak * a : 0.0;
                                }
                            }
                        };
                        row(i0, IntTag<1>{});
                        for (int i = i0 + 64; i < M; i += 64) row(i, IntTag<0>{});
                        wave_allreduce_many(acc, lane);
                        _Pragma("unroll")
                        for (int u = 0; u < R; u++) {
                            const double prow = lane_read(rowk[u], k & 63);
                            const double wj = prow + scal * acc[u];
                            const double tpv = -tau * wj;
                            tpn[u] = tpv;
                            if (lane == 0) {
                                const int j = j0 + u;
                                tp_next[j] = tpv;
                                if (j < P) qR[(size_t)k * P + j] = prow + tpv;
                                else gamma[k] = prow + tpv;
                            }
                        }
                    };
                    auto chunk = [&](int j0, auto rtag) __attribute__((always_inline)) {
                        double tpn[decltype(rtag)::value];
                        if (k > 0) chunk_t(j0, rtag, IntTag<1>{}, tpn);
                        else chunk_t(j0, rtag, IntTag<0>{}, tpn);
                        return tpn[0];
                    };
                    auto chunk_tail = [&](int j0, int r) __attribute__((always_inline)) {
                        switch (r) {
                            case 1: chunk(j0, IntTag<1>{}); break;
                            case 2: chunk(j0, IntTag<2>{}); break;
                            case 3: chunk(j0, IntTag<3>{}); break;
                            case 4: chunk(j0, IntTag<4>{}); break;
                            case 5: chunk(j0, IntTag<5>{}); break;
                            case 6: chunk(j0, IntTag<6>{}); break;
                            case 7: chunk(j0, IntTag<7>{}); break;
                            case 8: chunk(j0, IntTag<8>{}); break;
                            default: break;
                        }
                    };
                    // task 0 (wave 0): column k + 1 alone, then the look-ahead; the chunks of eight from k + 2 go round from wave 1
                    if (wave == 0) {
                        const double tpn = chunk(k + 1, IntTag<1>{});
                        if (k + 1 < P) {
                            const int kk = k + 1;
                            double *col = qa + (size_t)kk * M;
                            double a2[1] = {0.0}, akk = 0.0;
                            for (int i = lane + 64 * (kk / 64); i < M; i += 64) {
                                if (i >= kk) {
                                    const double v = kcol[i] * scal;
                                    double a = col[i];
                                    a = __builtin_fma(v, tpn, a);
                                    col[i] = a;
                                    if (i == kk) akk = a;
                                    a2[0] += (i > kk) ? a * a : 0.0;
                                }
                            }
                            wave_allreduce_many(a2, lane);
                            // ... and reflection k + 1's constants, for every wave to read behind the barrier
                            double nb, nt, ns;
                            reflector(lane_read(akk, kk & 63), a2[0], nb, nt, ns);
                            if (lane == 0) { double *rc = refl + 3 * (kk & 1); rc[0] = nb; rc[1] = nt; rc[2] = ns; }
                        }
                    }
                    {
                        // chunk length of the stage: eight columns, or -- several waves -- as few as give each of the other
                        // NW - 1 waves ONE chunk (late stages have few columns left: a wave with two columns finishes its pass,
                        // its reduction and its tprev sooner than one with eight while the rest idle; which columns share a
                        // chunk does not enter any sum)
                        int rch = kChunk;
                        if constexpr (NW > 1) {
                            const int rest = P - k - 1;                       // columns k + 2 .. P
                            rch = (rest + NW - 2) / (NW - 1);
                            rch = rch < 1 ? 1 : (rch > kChunk ? kChunk : rch);
                        }
                        int task = 1;
                        int j0 = k + 2;
                        if (rch == kChunk) {
                            for (; j0 + kChunk <= P + 1; j0 += kChunk)
                                if ((task++ & (NW - 1)) == wave) chunk(j0, IntTag<kChunk>{});
                        } else {
                            for (; j0 + rch <= P + 1; j0 += rch)
                                if ((task++ & (NW - 1)) == wave) chunk_tail(j0, rch);
                        }
                        if (j0 <= P && (task++ & (NW - 1)) == wave) chunk_tail(j0, P + 1 - j0);
                    }
                    scal_prev = scal;
                    sync();
                }
                DSQ_PROF(2);
                if (wave == 0) {
                    // R beta = gamma from the last row up: the solution lives across the lanes (lane j holds beta[j]), a row of
                    // R is read once, an entry per lane -- the chain's operands come from v_readlane
                    const int lc = lane < P ? lane : P - 1;
                    const double gv = gamma[lc];
                    double bv = 0.0;
                    // the chain tt <- fma(-r[i][j], beta[j], tt), j ascending, walks the lanes: lane j takes the value of lane
                    // j - 1 (one DPP shift) and applies ITS term -- no operand leaves its lane; what the other lanes compute on
                    // the way is never read
                    for (int i = P - 1; i >= 0; i--) {
                        const double rrow = qR[(size_t)i * P + lc];
                        double tv = gv;                                   // (lane i: gamma[i])
                        for (int j = i + 1; j < P; j++) tv = __builtin_fma(-rrow, bv, wave_shr1(tv));
                        const double bi = lane_read(tv, P - 1) / lane_read(rrow, i);
                        if (lane == i) bv = bi;
                    }
                    if (lane < P) beta[lane] = bv;
                }
                sync();
                DSQ_PROF(3);
            } else {
                // solve(beta_hat, x.t() * (x.each_col() % w_vec) + ridge, x.t() * (z % w_vec))            (:398)
                for (int j = tid; j < m; j += NT) {
                    const double mu = mu_s[j];
                    const double wv = wvec(j, mu);
                    const double z = lg_s[j] + ((double)yg[j] - mu) / mu;
                    w_s[j] = wv;
                    sw_s[j] = z * wv;
                }
                sync();
                gram(G, true);
                for (int a = tid; a < P; a += NT) G[(size_t)a * P + a] = G[(size_t)a * P + a] + lambda[a];
                sync();
                DSQ_PROF(1);
                lu_factor(G);
                DSQ_PROF(2);
                if (wave == 0) {
                    lu_solve_vec(G, rhs);
                    wave_lds_sync();
                    for (int a = lane; a < P; a += 64) beta[a] = rhs[a];
                }
                sync();
                DSQ_PROF(3);
            }
            int toolarge = 0;
            for (int c = 0; c < P; c++) toolarge += (__builtin_fabs(beta[c]) > large) ? 1 : 0;
            if (uniform(toolarge > 0)) { it = (double)kp.maxit; break; }                   // (:357-360; the same in every wave)
            update_mu();
            // the deviance terms by all waves, their sum by wave 0 in wave order                                  (:365-373)
            for (int j = tid; j < m; j += NT) {
                const double y = (double)yg[j], mu = mu_s[j];
                double tj;
                if (cell_dev_closed(y, size, fast)) {
                    const double am = alpha * mu, opm = 1.0 + am, rcp = 1.0 / opm;
                    const double l1p = dlog(opm) + (am - (opm - 1.0)) * rcp;
                    tj = y * lg_s[j] - (y + size) * l1p;
                } else tj = nb_offbranch(y, size, mu);
                w_s[j] = tj;
            }
            sync();
            if (wave == 0) {
                double dacc = 0.0;
                for (int j = lane; j < m; j += 64) {
                    if constexpr (USE_W) dacc += wg[j] * w_s[j];
                    else dacc += w_s[j];
                }
                dev = -2.0 * (K + wave_allreduce(dacc));
                const double conv_test = __builtin_fabs(dev - dev_old) / (__builtin_fabs(dev) + 0.1);
                double flag = 0.0;
                if (conv_test != conv_test) flag = 2.0;                                    // (:375-378)
                else if (kp.force_iters > 0) { if (t + 1 >= kp.force_iters) flag = 1.0; }
                else if ((t > 0) && (conv_test < kp.tol)) flag = 1.0;                      // (:379-381)
                dev_old = dev;
                if (lane == 0) ctl[0] = flag;
            }
            sync();
            DSQ_PROF(4);
            const double flag = ctl[0];
            if (uniform(flag == 2.0)) { it = (double)kp.maxit; break; }
            if (uniform(flag == 1.0)) break;
        }

        // ---- post-loop block (:427-455) ------------------------------------------------
        sync();
        for (int j = tid; j < m; j += NT) {
            const double wv = wvec(j, mu_s[j]);
            w_s[j] = wv;
            sw_s[j] = __builtin_sqrt(wv);
        }
        sync();
        gram(G, false);
        DSQ_PROF(5);
        for (int e = tid; e < P * P; e += NT) {
            const int i = e / P, j = e - i * P;
            double v = G[e];
            if (i == j) v = v + lambda[i];
            LUm[e] = v;
        }
        sync();
        lu_factor(LUm);
        DSQ_PROF(6);
        // Gi = inverse: lane c owns right-hand side e_c (LU<P>::inverse = P solves)                              [wave 0]
        // (row i of the factors is read ONCE, an entry per lane, its multipliers come from v_readlane; the solution's entries
        //  are loaded eight ahead of the chain)
        if (wave == 0) {
            const int c = lane < P ? lane : P - 1;       // (the spare lanes shadow the last column and store nothing)
            const bool own = lane < P;
            int pos = c;                                 // e_c under the row swaps: where its 1 ends
            for (int k = 0; k < P; k++) {
                const int pr = piv[k];
                pos = (pos == k) ? pr : ((pos == pr) ? k : pos);
            }
            for (int i = 0; i < P; i++) {
                const double arow = LUm[(size_t)i * P + c];
                double t = (i == pos) ? 1.0 : 0.0;
                int j = 0;
                for (; j + 8 <= i; j += 8) {
                    double xv[8];
                    _Pragma("unroll")
                    for (int u = 0; u < 8; u++) xv[u] = Gi[(size_t)(j + u) * P + c];
                    _Pragma("unroll")
                    for (int u = 0; u < 8; u++) t = __builtin_fma(-lane_read(arow, j + u), xv[u], t);
                }
                for (; j < i; j++) t = __builtin_fma(-lane_read(arow, j), Gi[(size_t)j * P + c], t);
                if (own) Gi[(size_t)i * P + c] = t;
            }
            for (int i = P - 1; i >= 0; i--) {
                const double arow = LUm[(size_t)i * P + c];
                double t = Gi[(size_t)i * P + c];
                int j = i + 1;
                for (; j + 8 <= P; j += 8) {
                    double xv[8];
                    _Pragma("unroll")
                    for (int u = 0; u < 8; u++) xv[u] = Gi[(size_t)(j + u) * P + c];
                    _Pragma("unroll")
                    for (int u = 0; u < 8; u++) t = __builtin_fma(-lane_read(arow, j + u), xv[u], t);
                }
                for (; j < P; j++) t = __builtin_fma(-lane_read(arow, j), Gi[(size_t)j * P + c], t);
                if (own) Gi[(size_t)i * P + c] = t * rdiag[i];
            }
        }
        sync();
        DSQ_PROF(7);
        // hat diagonal, loop order of :443-449; fitted means (extension)                                     [all waves]
        if (kp.hat_diagonals || kp.mu_out) {
            for (int j = tid; j < m; j += NT) {
                if (kp.hat_diagonals) {
                    const double sw = sw_s[j];
                    double h = 0.0;
                    for (int i1 = 0; i1 < P; i1++) {
                        const double xw1 = xs[(size_t)i1 * m + j] * sw;
                        int i2 = 0;
                        for (; i2 + 8 <= P; i2 += 8) {                    // (eight terms' loads ahead of the additions)
                            double x2[8], gv[8];
                            _Pragma("unroll")
                            for (int u = 0; u < 8; u++) { x2[u] = xs[(size_t)(i2 + u) * m + j]; gv[u] = Gi[(size_t)(i2 + u) * P + i1]; }
                            _Pragma("unroll")
                            for (int u = 0; u < 8; u++) h += xw1 * ((x2[u] * sw) * gv[u]);
                        }
                        for (; i2 < P; i2++) {
                            const double xw2 = xs[(size_t)i2 * m + j] * sw;
                            h += xw1 * (xw2 * Gi[(size_t)i2 * P + i1]);
                        }
                    }
                    kp.hat_diagonals[(size_t)g * kp.ld + j] = h;
                }
                if (kp.mu_out) {
                    double eta = xs[j] * beta[0];
                    for (int c = 1; c < P; c++) eta = __builtin_fma(xs[(size_t)c * m + j], beta[c], eta);
                    double v = nfg[j] * dexp(eta);
                    if (kp.mu_floor > 0.0) v = __builtin_fmax(v, kp.mu_floor);
                    kp.mu_out[(size_t)g * kp.ld + j] = v;
                }
            }
        }
        DSQ_PROF(8);
        // sigma = Gi * G * Gi (:452), mat_mul's order: c[i][j] = sum_k fma(a[i][k], b[k][j]), k ascending; lane j owns column
        // j, the rows go round the waves
        if (lane < P) {
            const int j = lane;
            for (int i = wave; i < P; i += NW) {
                double acc = 0.0;
                for (int k = 0; k < P; k++) acc = __builtin_fma(Gi[(size_t)i * P + k], G[(size_t)k * P + j], acc);
                Tm[(size_t)i * P + j] = acc;
            }
        }
        sync();
        if (lane < P) {
            const int j = lane;
            for (int i = wave; i < P; i += NW) {
                double acc = 0.0;
                for (int k = 0; k < P; k++) acc = __builtin_fma(Tm[(size_t)i * P + k], Gi[(size_t)k * P + j], acc);
                Sg[(size_t)i * P + j] = acc;
            }
        }
        sync();
        if (wave == 0) {
            double cn = 0.0;
            for (int c = 0; c < P; c++) cn = __builtin_fma(contrast[c], beta[c], cn);
            if (lane < P) {
                double rr = 0.0;
                for (int a = 0; a < P; a++) rr = __builtin_fma(contrast[a], Sg[(size_t)a * P + lane], rr);
                rhs[lane] = rr;
            }
            wave_lds_sync();
            double cd = 0.0;
            for (int b = 0; b < P; b++) cd = __builtin_fma(rhs[b], contrast[b], cd);
            for (int c = lane; c < kp.p; c += 64) {
                kp.beta_mat[(size_t)g + (size_t)kp.n * c] = c < P ? beta[c] : 0.0;
                kp.beta_var_mat[(size_t)g + (size_t)kp.n * c] = c < P ? Sg[(size_t)c * P + c] : 0.0;
            }
            if (lane == 0) {
                kp.iter[g] = it;
                kp.deviance[g] = dev;
                kp.contrast_num[g] = cn;
                kp.contrast_denom[g] = __builtin_sqrt(cd);
                ctl[1] = (double)(kp.work_counter ? atomicAdd(kp.work_counter, 1) + (int)gridDim.x : wi + (int)gridDim.x);
            }
        }
        sync();
        DSQ_PROF(9);
        wi = (int)ctl[1];
    }
    DSQ_PROF_FLUSH(betaw_prof);
}

// ---- launch ---------------------------------------------------------------------------------------------------------
// One workgroup of four waves per gene, persistent grid.  LDS mode whenever a gene's slab fits the CU's 160 KB next to the
// vectors (as many genes per CU as fit); else the slabs live in global memory: two workgroups per CU by the registers -- but
// no more resident genes than keep their slabs inside DSQ_WIDE_SLAB_MB (default 192): slabs that fit the 256 MB infinity
// cache together are served from there instead of from HBM.
struct WideGeom { int grid, nw; size_t lds; bool big_lds; };
template <bool USE_W, bool BIG_LDS>
static const void *wide_fn(int nw) {
    return nw == 1 ? (const void *)fit_beta_rolled_kernel<USE_W, BIG_LDS, 1> : nw == 2 ? (const void *)fit_beta_rolled_kernel<USE_W, BIG_LDS, 2>
           : nw == 4 ? (const void *)fit_beta_rolled_kernel<USE_W, BIG_LDS, 4> : (const void *)fit_beta_rolled_kernel<USE_W, BIG_LDS, 8>;
}
static const void *wide_fn(bool useW, bool big, int nw) {
    return useW ? (big ? wide_fn<true, true>(nw) : wide_fn<true, false>(nw)) : (big ? wide_fn<false, true>(nw) : wide_fn<false, false>(nw));
}
static WideGeom wide_geometry(int n, int m, int p, bool useW) {
    WideGeom g;
    const size_t cu_lds = 160 * 1024;
    const size_t vec_b = wide_lds_doubles(p) * sizeof(double), slab_b = wide_slab_doubles(m, p) * sizeof(double);
    const int force = getenv("DSQ_WIDE_LDS") ? atoi(getenv("DSQ_WIDE_LDS")) : -1;       // 0: never; k > 0: only with k genes per CU
    const int force_nw = getenv("DSQ_WIDE_NW") ? atoi(getenv("DSQ_WIDE_NW")) : 0;       // waves per gene (1, 2, 4)
    const int fit = (int)(cu_lds / (vec_b + slab_b));              // genes per CU with their slabs in LDS
    g.big_lds = force == 0 ? false : fit >= (force > 0 ? force : 1);
    const int cus = device_cu_count();
    if (g.big_lds) {
        g.lds = vec_b + slab_b;
        g.nw = fit >= 8 ? 1 : fit >= 4 ? 2 : fit >= 2 ? 4 : 8;     // about eight resident waves per CU
        if (force_nw == 1 || force_nw == 2 || force_nw == 4 || force_nw == 8) g.nw = force_nw;
        int bpc = fit;
        if (bpc * g.nw > 8) bpc = 8 / g.nw;                        // (two waves per SIMD by the registers)
        if (bpc < 1) bpc = 1;
        const void *fn = wide_fn(useW, true, g.nw);
        if (g.lds > 64 * 1024) (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)cu_lds);
        const long cap = (long)cus * bpc;
        g.grid = (int)((long)n < cap ? (long)n : cap);
        if (g.grid < 1) g.grid = 1;
        if (getenv("DSQ_VERBOSE")) fprintf(stderr, "[dsq] fit_beta_rolled p=%d m=%d: slab in LDS, lds=%zu, %d waves per gene, %d genes/CU\n", p, m, g.lds, g.nw, bpc);
        return g;
    }
    g.lds = vec_b;
    g.nw = (force_nw == 1 || force_nw == 2 || force_nw == 4 || force_nw == 8) ? force_nw : 4;
    int bpc = 8 / g.nw;
    const int slab_mb = getenv("DSQ_WIDE_SLAB_MB") ? atoi(getenv("DSQ_WIDE_SLAB_MB")) : 192;
    long cap = (long)cus * bpc;
    const long fitb = (long)(((size_t)slab_mb << 20) / slab_b);
    if (cap > fitb) cap = fitb < cus ? cus : fitb;              // (never below one workgroup per CU)
    long gr = (long)n < cap ? (long)n : cap;
    if (gr < 1) gr = 1;
    g.grid = (int)gr;
    if (getenv("DSQ_VERBOSE")) fprintf(stderr, "[dsq] fit_beta_rolled p=%d m=%d: slab in global memory, %d waves per gene, grid %d\n", p, m, g.nw, g.grid);
    return g;
}

void fit_beta_rolled_scratch_doubles(int n, int m, int p, int useW, size_t *slab, size_t *cscr) {
    const WideGeom g = wide_geometry(n, m, p, useW != 0);
    *slab = g.big_lds ? 0 : (size_t)g.grid * wide_slab_doubles(m, p);
    *cscr = 0;
}

hipError_t launch_fit_beta_rolled(const BetaKernelParams &kp0, hipStream_t st) {
    // at the design's own width when its slab then fits the LDS (no scratch involved); else at the padded width the scratch
    // was sized for
    WideGeom g = wide_geometry(kp0.n, kp0.m, kp0.p, kp0.useWeights != 0);
    BetaKernelParams kp = kp0;
    kp.p_true = kp0.p;
    if (kp0.p_true >= 11 && kp0.p_true < kp0.p && !(getenv("DSQ_WIDE_PADDED") && atoi(getenv("DSQ_WIDE_PADDED")))) {
        const WideGeom gt = wide_geometry(kp0.n, kp0.m, kp0.p_true, kp0.useWeights != 0);
        if (gt.big_lds) { g = gt; kp.p_true = kp0.p_true; }
    }
    kp.xlds = 0;
    int grid = g.grid;
    // (a row list: its length lives on the device; the scratch was sized for the full grid, a smaller one uses its head)
    if (kp.rows_few && grid > device_cu_count()) grid = device_cu_count();
    void *args[] = {&kp};
#ifdef DSQ_WIDE_PROF
    {
        unsigned long long z[DSQ_PROF_SLOTS] = {}, h[DSQ_PROF_SLOTS];
        (void)hipMemcpyToSymbol(HIP_SYMBOL(betaw_prof), z, sizeof(z));
        const hipError_t e = hipLaunchKernel(wide_fn(kp.useWeights != 0, g.big_lds, g.nw), dim3(grid), dim3(64 * g.nw), args, g.lds, st);
        (void)hipStreamSynchronize(st);
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(betaw_prof), sizeof(h));
        double tot = 0;
        for (int q = 0; q < DSQ_PROF_SLOTS; q++) tot += (double)h[q];
        static const char *nm[10] = {"start", "rows / gram", "QR stages / LU", "back-subst / solve", "mu + deviance", "post gram", "post LU", "post inverse", "hat", "sigma + out"};
        fprintf(stderr, "[betaw_prof] p=%d m=%d nw=%d lds=%d qr=%d:", kp.p, kp.m, g.nw, (int)g.big_lds, kp.useQR);
        for (int q = 0; q < 10; q++) fprintf(stderr, " %s %.1f%%", nm[q], 100.0 * (double)h[q] / (tot > 0 ? tot : 1));
        fprintf(stderr, "  (%.0f Mcycles of thread 0 over %d workgroups)\n", tot / 1e6, grid);
        return e;
    }
#endif
    return hipLaunchKernel(wide_fn(kp.useWeights != 0, g.big_lds, g.nw), dim3(grid), dim3(64 * g.nw), args, g.lds, st);
}

}  // namespace dsq
