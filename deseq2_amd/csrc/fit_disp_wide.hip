// fit_disp_wide.hip -- fitDisp / fitDispGrid / d2log_posterior (src/DESeq2.cpp:31-277, 469-513) for WIDE designs WITHOUT
// design cells (11 .. 64 columns: paired designs, factors of more than 32 levels, many covariates) on rows of at most
// DSQ_SPEC_SERIAL_GRAM_MAXM samples: one kernel for every width, the loops over the design columns rolled, a workgroup of
// NW waves per gene -- the counterpart of fit_beta_wide.hip.
//
// The per-width kernel (fit_disp.hip, general mode) keeps the K Cox-Reid matrices and their LU / inverse one column per
// lane in REGISTERS -- 2 P .. 5 P doubles per lane, the p x p steps fully unrolled: four-minute builds per padded width, one
// wave per SIMD at 48 columns, 130 - 390 ms per 20 000 genes.  Here the matrices live in the gene's LDS slab (row-major,
// lane j owns column j), the samples' terms are produced by all waves, and what is independent is split between them: the
// entries of the Gram matrices (2 x 2 blocks per thread, read from a SAMPLE-major copy of the design: a wave's loads of one
// sample touch its three or four cache lines, not 64), the rows of an elimination step.  THE ARITHMETIC IS THAT OF DispGene (fit_disp.hip) in
// general mode with the entry-per-lane Gram: every matrix entry is the SERIAL sum of its m terms in sample order (the
// arithmetic spec's order for these shapes, include/dsq_arith_spec.h -- which thread takes an entry does not enter it), the
// likelihood sums are wave-order sums taken by ONE wave over terms the others have left in LDS, LU / inverse / traces are
// LaneLU's operations on the same values, the line search is the per-width kernel's statement by statement: the results keep
// the oracle's bits (tests/test_gpu_wide.py).  The design arrives zero-padded (kp.p columns, kp.padmask): the kernel runs on
// its first p columns, the true width.  Measured (profiles/r06_wide.txt): paired p = 46 132 -> 75 ms, factor p = 48 390 -> 108 ms
// per 20 000 genes; phase shares in profiles/r06_wide_phases.txt (make prof).
#include "dsq_internal.hpp"
#include <cstdio>
#include <cstdlib>
#include "dsq_math.hpp"
#include "dsq_wave.hpp"
#include "../../include/dsq_arith_spec.h"

namespace dsq {

template <int V> struct DTag { static constexpr int value = V; };

#include "dsq_prof.hpp"
#ifdef DSQ_WIDE_PROF
__device__ unsigned long long dispw_prof[DSQ_PROF_SLOTS];
#endif

// LDS per gene (doubles): vectors | mu m | w m | y m int32 | distinct counts 2 m int32 | terms 2 m | wd 3 m | 1, 3 or 4 matrices p p
__host__ __device__ static inline size_t dispw_vec_doubles(int p) { return (size_t)4 * p + 24; }
__host__ __device__ static inline size_t dispw_lds_doubles(int m, int p, bool useW, int mode) {
    const size_t half = ((size_t)m + 1) / 2;
    const int nmat = mode == 2 ? 4 : (mode == 1 ? 1 : 3);       // (mode 2: B0 (-> LU), B1, B2 (-> M = Bi B1, once its trace is taken), Bi)
    return dispw_vec_doubles(p) + (size_t)m * (useW ? 2 : 1) + half + (useW ? 0 : (size_t)m) + (size_t)2 * m + (size_t)3 * m +
           (size_t)nmat * p * p + 8;
}

template <bool USE_W, int MODE, int NW>
// (HIP's second launch bound is WAVES PER SIMD: the one- and two-wave builds at three -- 168 registers, the spills sit in the
//  once-per-gene parts -- so that five or six genes fit a CU where the LDS admits them; the others at two)
__global__ void __launch_bounds__(64 * NW, NW <= 2 ? 3 : 2) fit_disp_rolled_kernel(DispKernelParams kp, int P, const double *xt) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int NT = 64 * NW;
    const int m = kp.m;
    const int nwork = DSQ_NWORK(kp);
    if ((int)blockIdx.x >= nwork) return;
    const double *xs = kp.x;                     // column c of the design at xs + c m (the first P columns of the padded matrix)
    // xt: the same P columns SAMPLE-major (sample j at xt + j P), written by dispw_transpose_kernel before the launch: the
    // threads of a wave hold neighbouring entries (a, b) of a Gram matrix -- with the column-major design their loads of
    // x[j][b] touch 64 cache lines per instruction (a launch at 48 columns spent 4/5 of its time in the L1's address unit),
    // with the sample-major one the three or four lines of row j

    // ---- LDS carve
    double *vec = smem;
    double *tcol = vec, *rdiag = vec + P;        // per-column partial traces; reciprocal pivots
    int *piv = reinterpret_cast<int *>(vec + 2 * P);
    double *ctl = vec + 3 * P + 8;               // results of an evaluation for all waves: [0] lp, [1] dlp, [2] next gene, [3] d2
    double *p0 = vec + dispw_vec_doubles(P);
    double *mu_s = p0; p0 += m;
    double *w_s = nullptr;
    if constexpr (USE_W) { w_s = p0; p0 += m; }
    int32_t *y_s = reinterpret_cast<int32_t *>(p0); p0 += ((size_t)m + 1) / 2;
    int32_t *dist = nullptr;
    if constexpr (!USE_W) { dist = reinterpret_cast<int32_t *>(p0); p0 += m; }
    double *T1 = p0, *T2 = p0 + m; p0 += 2 * (size_t)m;
    double *wdbuf = p0; p0 += 3 * (size_t)m;
    double *Bm = p0;                             // B[k] at Bm + k P P (k < K); MODE 0 / 2: then Bi (and M)
    constexpr int KMAX = MODE == 2 ? 3 : (MODE == 1 ? 1 : 2);
    double *Bi = Bm + (size_t)KMAX * P * P, *Mm = Bm + 2 * (size_t)P * P;     // (M takes the place of B2)
    (void)Mm;

    auto sync = [] {
        if constexpr (NW == 1) wave_lds_sync();
        else __syncthreads();
    };

    DSQ_PROF_DECL;
    int wi = blockIdx.x;
    while (wi < nwork) {
        const int g = DSQ_GENE(kp, wi);
        const int32_t *yg = kp.y + (size_t)g * kp.ld;
        const double *mug = kp.mu_hat + (size_t)g * kp.ld;
        const double *wg = USE_W ? kp.weights + (size_t)g * kp.ld : nullptr;
        sync();
        bool ok_l = true;
        for (int j = tid; j < m; j += NT) {
            const double mu = mug[j];
            mu_s[j] = mu;
            y_s[j] = yg[j];
            if constexpr (USE_W) w_s[j] = wg[j];
            ok_l = ok_l && (mu >= 0.0) && (mu < 1e140);
        }
        // (mu_ok of DispGene: every fitted mean in [0, 1e140) and the log alpha values in [-40, 40] -> the scaling-free reciprocal)
        bool la_ok;
        if constexpr (MODE == 1) la_ok = kp.grid[0] >= -40.0 && kp.grid[kp.ngrid - 1] <= 40.0 && kp.grid[0] <= kp.grid[kp.ngrid - 1];
        else { const double a0 = (MODE == 0) ? kp.log_alpha_in[g] : kp.log_alpha[g]; la_ok = a0 >= -40.0 && a0 <= 40.0; }
        const bool mu_ok = (NW == 1 ? (bool)__all(ok_l) : (bool)__syncthreads_and((int)ok_l)) && la_ok;
        sync();
        const double prior_mean = kp.prior_mean[g];
        const double prior_sigmasq = kp.prior_sigmasq_dev ? *kp.prior_sigmasq_dev : kp.prior_sigmasq;
        const double thr = kp.weightThreshold;
        const bool usePrior = kp.usePrior != 0, useCR = kp.useCR != 0;
        auto rcp1 = [&](double opm) { return mu_ok ? drcp_n(opm) : 1.0 / opm; };
        auto keep_row = [&](int j) -> bool {
            if constexpr (USE_W) return w_s[j] > thr;
            return true;
        };
        // distinct counts (unweighted genes), by wave 0: the ascending values dist[0 .. nv), multiplicities dist[m .. m + nv)
        int nv = 0;
        if constexpr (!USE_W) {
            if (wave == 0) {
                nv = wave_distinct_counts(dist, m, lane, [&](int k) { return (int32_t)y_s[k]; }, m >= 256, nullptr);
                if (lane == 0) ctl[4] = (double)nv;
            }
            sync();
            nv = (int)ctl[4];
        }
        const int32_t *dv = dist, *dc = dist ? dist + m : nullptr;
        // x = x.rows(find(wts > weightThreshold)); x = x.cols(find(sum(abs(x)) > 0.0)): the columns left all-zero   (:41-43)
        unsigned long long dropmask = 0ull;
        if constexpr (USE_W) {
            if (useCR) {
                for (int c = 0; c < P; c++) {
                    bool any = false;
                    for (int j = tid; j < m; j += NT)
                        if (keep_row(j) && __builtin_fabs(xs[(size_t)c * m + j]) > 0.0) any = true;
                    const bool anyb = NW == 1 ? (bool)__any(any) : (bool)__syncthreads_or((int)any);
                    if (!anyb) dropmask |= (1ull << c);
                }
            }
        }

        DSQ_PROF(0);
        // ---- the K Cox-Reid matrices X' diag(wd_k) X from the samples' diagonals in wdbuf: entry (a, b), a <= b, is the SERIAL
        //      sum of its m terms in sample order (the spec's order for these shapes), one entry per thread; a dropped column
        //      contributes exact zeros and carries 1 on its diagonal in the first matrix
        auto gram = [&](auto ktag) __attribute__((always_inline)) {
            constexpr int K = decltype(ktag)::value;
            // a thread takes 2 x 2 blocks of entries -- rows a0, a0 + 1, columns b0, b0 + 1 of the upper triangle --: four design
            // values per sample serve four entries, the products x_b w_k are taken once per column (the same operation, the
            // same bits); SB samples' loads are in flight at a time (the design comes through L1 / L2, sample-major), the
            // additions of an entry stay in sample order
            constexpr int SB = 4;
            const int PB = (P + 1) >> 1, NB = PB * (PB + 1) / 2;
            // the blocks of the last, PARTIAL round (NB mod NT of them) are split into their single entries over all threads
            // when that fills the round better: at 48 columns and 256 threads 300 blocks are one full round and 44 left over
            // -- 176 single entries, a third of a block's work each, instead of a second round for 44 threads
            const int nfull = NB - NB % NT;
            const int nsplit = ((NB - nfull) * 4 <= 2 * NT) ? NB - nfull : 0;       // blocks whose entries are spread
            const int nblk = NB - nsplit;
            for (int q = tid; q < 4 * nsplit; q += NT) {
                const int e = nblk + (q >> 2), ia = (q >> 1) & 1, ib = q & 1;
                int ab = 0, rem = e;
                while (rem >= PB - ab) { rem -= PB - ab; ab++; }
                const int ea = 2 * ab + ia, eb = 2 * (ab + rem) + ib;
                if (ea < P && eb < P && ea <= eb) {
                    const double *xa_p = xt + ea, *xb_p = xt + eb;
                    double acc1[K];
                    _Pragma("unroll")
                    for (int k = 0; k < K; k++) acc1[k] = 0.0;
                    int j = 0;
                    for (; j + 8 <= m; j += 8) {
                        double xa[8], xb[8], wv[K][8];
                        _Pragma("unroll")
                        for (int u = 0; u < 8; u++) {
                            const size_t o = (size_t)(j + u) * P;
                            xa[u] = xa_p[o]; xb[u] = xb_p[o];
                            _Pragma("unroll")
                            for (int k = 0; k < K; k++) wv[k][u] = wdbuf[(size_t)k * m + j + u];
                        }
                        _Pragma("unroll")
                        for (int u = 0; u < 8; u++)
                            _Pragma("unroll")
                            for (int k = 0; k < K; k++) acc1[k] += xa[u] * (xb[u] * wv[k][u]);
                    }
                    for (; j < m; j++) {
                        const size_t o = (size_t)j * P;
                        const double xa = xa_p[o], xb = xb_p[o];
                        _Pragma("unroll")
                        for (int k = 0; k < K; k++) acc1[k] += xa * (xb * wdbuf[(size_t)k * m + j]);
                    }
                    const bool live = !(((dropmask >> ea) | (dropmask >> eb)) & 1ull);
                    _Pragma("unroll")
                    for (int k = 0; k < K; k++) {
                        const double v = live ? acc1[k] : 0.0;
                        Bm[((size_t)k * P + ea) * P + eb] = v;
                        Bm[((size_t)k * P + eb) * P + ea] = v;
                    }
                }
            }
            for (int e = tid; e < nblk; e += NT) {
                int ab = 0, rem = e;
                while (rem >= PB - ab) { rem -= PB - ab; ab++; }
                const int a0 = 2 * ab, b0 = 2 * (ab + rem);
                const bool a1ok = a0 + 1 < P, b1ok = b0 + 1 < P;
                const double *xa0_p = xt + a0, *xa1_p = xt + (a1ok ? a0 + 1 : a0), *xb0_p = xt + b0, *xb1_p = xt + (b1ok ? b0 + 1 : b0);
                double acc[2][2][K];
                _Pragma("unroll")
                for (int q = 0; q < 4 * K; q++) (&acc[0][0][0])[q] = 0.0;
                int j = 0;
                for (; j + SB <= m; j += SB) {
                    double xa0[SB], xa1[SB], xb0[SB], xb1[SB], wv[K][SB];
                    _Pragma("unroll")
                    for (int u = 0; u < SB; u++) {
                        const size_t o = (size_t)(j + u) * P;
                        xa0[u] = xa0_p[o]; xa1[u] = xa1_p[o]; xb0[u] = xb0_p[o]; xb1[u] = xb1_p[o];
                        _Pragma("unroll")
                        for (int k = 0; k < K; k++) wv[k][u] = wdbuf[(size_t)k * m + j + u];
                    }
                    _Pragma("unroll")
                    for (int u = 0; u < SB; u++)
                        _Pragma("unroll")
                        for (int k = 0; k < K; k++) {
                            const double t0 = xb0[u] * wv[k][u], t1 = xb1[u] * wv[k][u];
                            acc[0][0][k] += xa0[u] * t0; acc[0][1][k] += xa0[u] * t1;
                            acc[1][0][k] += xa1[u] * t0; acc[1][1][k] += xa1[u] * t1;
                        }
                }
                for (; j < m; j++) {
                    const size_t o = (size_t)j * P;
                    const double xa0 = xa0_p[o], xa1 = xa1_p[o], xb0 = xb0_p[o], xb1 = xb1_p[o];
                    _Pragma("unroll")
                    for (int k = 0; k < K; k++) {
                        const double w = wdbuf[(size_t)k * m + j];
                        const double t0 = xb0 * w, t1 = xb1 * w;
                        acc[0][0][k] += xa0 * t0; acc[0][1][k] += xa0 * t1;
                        acc[1][0][k] += xa1 * t0; acc[1][1][k] += xa1 * t1;
                    }
                }
                _Pragma("unroll")
                for (int ia = 0; ia < 2; ia++)
                    _Pragma("unroll")
                    for (int ib = 0; ib < 2; ib++) {
                        const int ea = a0 + ia, eb = b0 + ib;
                        if ((ia == 0 || a1ok) && (ib == 0 || b1ok) && ea <= eb) {     // (the block on the diagonal: its lower entry is the mirror)
                            const bool live = !(((dropmask >> ea) | (dropmask >> eb)) & 1ull);
                            _Pragma("unroll")
                            for (int k = 0; k < K; k++) {
                                const double v = live ? acc[ia][ib][k] : 0.0;
                                Bm[((size_t)k * P + ea) * P + eb] = v;
                                Bm[((size_t)k * P + eb) * P + ea] = v;
                            }
                        }
                    }
            }
            sync();
            if constexpr (USE_W) {
                for (int i = tid; i < P; i += NT)
                    if ((dropmask >> i) & 1ull) Bm[(size_t)i * P + i] = 1.0;
                sync();
            }
        };
        // LaneLU<P>::factor on the row-major matrix A (lane j = column j): first maximum wins, rows swapped in every column,
        // reciprocal pivots, fma(-l, u, a); `sign` as LaneLU keeps it.  Pivot search / swap by wave 0, the rows round the waves.
        auto lu_factor = [&](double *A, int &sign) {
            sign = 1;
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
                if (piv[k] != k) sign = -sign;
                const double rinv = 1.0 / A[(size_t)k * P + k];
                if (tid == 0) rdiag[k] = rinv;
                const double akj = lane < P ? A[(size_t)k * P + lane] : 0.0;
                const int lc = lane < P ? lane : P - 1;
                for (int i0 = k + 1 + wave; i0 < P; i0 += 4 * NW) {      // four of the wave's rows in flight
                    double lv[4], av[4];
                    _Pragma("unroll")
                    for (int u = 0; u < 4; u++) {
                        const int i = i0 + u * NW;
                        if (i < P) { lv[u] = A[(size_t)i * P + k]; av[u] = A[(size_t)i * P + lc]; }
                    }
                    _Pragma("unroll")
                    for (int u = 0; u < 4; u++) {
                        const int i = i0 + u * NW;
                        if (i < P) {
                            const double l = lv[u] * rinv;
                            if (lane == k) A[(size_t)i * P + k] = l;
                            else if (lane > k && lane < P) A[(size_t)i * P + lane] = __builtin_fma(-l, akj, av[u]);
                        }
                    }
                }
                sync();
            }
        };
        auto lu_det = [&](const double *A, int sign) -> double {
            double d = A[0];
            for (int i = 1; i < P; i++) d = d * A[(size_t)i * P + i];
            return sign < 0 ? -d : d;
        };
        // Bi = inverse (LaneLU::inverse: lane c solves right-hand side e_c, the fma chains of LaneLU::solve term by term); wave 0.
        // The chains are the critical path of an evaluation (P^2 dependent fma): row i of the factors is read ONCE, one entry
        // per lane, and its multipliers come from v_readlane; the solution's entries are loaded eight ahead of the chain.
        auto lu_inverse = [&](const double *A, double *Inv) {
            if (wave == 0) {
                const int c = lane < P ? lane : P - 1;       // (the spare lanes shadow the last column and store nothing)
                const bool own = lane < P;
                // e_c under the row swaps: where its 1 ends
                int pos = c;
                for (int k = 0; k < P; k++) {
                    const int pr = piv[k];
                    pos = (pos == k) ? pr : ((pos == pr) ? k : pos);
                }
                for (int i = 0; i < P; i++) {
                    const double arow = A[(size_t)i * P + c];
                    double t = (i == pos) ? 1.0 : 0.0;
                    int j = 0;
                    for (; j + 8 <= i; j += 8) {
                        double xv[8];
                        _Pragma("unroll")
                        for (int u = 0; u < 8; u++) xv[u] = Inv[(size_t)(j + u) * P + c];
                        _Pragma("unroll")
                        for (int u = 0; u < 8; u++) t = __builtin_fma(-lane_read(arow, j + u), xv[u], t);
                    }
                    if (j < i) {                      // (the last, partial batch: clamped loads, the fma of the live ones)
                        double xv[8];
                        _Pragma("unroll")
                        for (int u = 0; u < 8; u++) xv[u] = Inv[(size_t)(j + u < i ? j + u : i - 1) * P + c];
                        _Pragma("unroll")
                        for (int u = 0; u < 8; u++)
                            if (j + u < i) t = __builtin_fma(-lane_read(arow, j + u), xv[u], t);
                    }
                    if (own) Inv[(size_t)i * P + c] = t;
                }
                for (int i = P - 1; i >= 0; i--) {
                    const double arow = A[(size_t)i * P + c];
                    double t = Inv[(size_t)i * P + c];
                    int j = i + 1;
                    for (; j + 8 <= P; j += 8) {
                        double xv[8];
                        _Pragma("unroll")
                        for (int u = 0; u < 8; u++) xv[u] = Inv[(size_t)(j + u) * P + c];
                        _Pragma("unroll")
                        for (int u = 0; u < 8; u++) t = __builtin_fma(-lane_read(arow, j + u), xv[u], t);
                    }
                    if (j < P) {
                        double xv[8];
                        _Pragma("unroll")
                        for (int u = 0; u < 8; u++) xv[u] = Inv[(size_t)(j + u < P ? j + u : P - 1) * P + c];
                        _Pragma("unroll")
                        for (int u = 0; u < 8; u++)
                            if (j + u < P) t = __builtin_fma(-lane_read(arow, j + u), xv[u], t);
                    }
                    if (own) Inv[(size_t)i * P + c] = t * rdiag[i];
                }
            }
        };
        // lane_trace_sym: t_c = sum_i fma(A[i][c], B[i][c]) per column, then the columns' t added in ascending order; wave 0
        auto trace_sym = [&](const double *A, const double *B) -> double {
            if (lane < P) {
                double t = 0.0;
                for (int i = 0; i < P; i++) t = __builtin_fma(A[(size_t)i * P + lane], B[(size_t)i * P + lane], t);
                tcol[lane] = t;
            }
            wave_lds_sync();
            double tr = 0.0;
            for (int k = 0; k < P; k++) tr = tr + tcol[k];
            wave_lds_sync();
            return tr;
        };
        // the likelihood sums of an evaluation from the samples' terms: wave-order sums over T1 (and T2), wave 0
        auto wave_sum = [&](const double *T) -> double {
            double acc = 0.0;
            for (int j = lane; j < m; j += 64) acc += T[j];
            return acc;
        };

        // ---- log_posterior (src/DESeq2.cpp:31-64), DispGene::lp                              -> ctl[0] (every wave reads it)
        auto eval_lp = [&](double la) -> double {
            const double alpha = dexp(la);
            const double an1 = 1.0 / alpha;
            double lg_an1 = 0.0;
            if constexpr (USE_W) lg_an1 = dlgamma(an1);
            for (int j = tid; j < m; j += NT) {
                const double y = (double)y_s[j], mu = mu_s[j];
                const double opm = 1.0 + mu * alpha;
                double wd0 = 0.0;
                if (useCR) wd0 = mu * rcp1(opm);
                const double l1 = dlog(opm);
                if constexpr (USE_W) {
                    const double t = dlgamma(y + an1) - lg_an1 - y * (l1 - la) - an1 * l1;
                    T1[j] = w_s[j] * t;
                } else {
                    T1[j] = -(y * (l1 - la)) - an1 * l1;
                }
                wdbuf[j] = keep_row(j) ? wd0 : 0.0;
            }
            sync();
            DSQ_PROF(1);
            if (useCR) gram(DTag<1>{});
            DSQ_PROF(2);
            int sign = 1;
            if (useCR) lu_factor(Bm, sign);
            DSQ_PROF(3);
            if (wave == 0) {
                double cr_term = 0.0;
                if (useCR) cr_term = -0.5 * dlog(lu_det(Bm, sign));
                const double acc = wave_sum(T1);
                double ll_part;
                if constexpr (USE_W) {
                    ll_part = wave_allreduce(acc);
                } else {
                    double accv = 0.0;
                    for (int q0 = 0; q0 <= nv; q0 += 64) {
                        const int q = q0 + lane;
                        const bool live = (q > 0) && (q <= nv);
                        const double lg = dlgamma((live || q == 0) ? (live ? (double)dv[q - 1] : 0.0) + an1 : 20.0);
                        if (q0 == 0) lg_an1 = lane_read(lg, 0);
                        if (live) accv += (double)dc[q - 1] * (lg - lg_an1);
                    }
                    double sv = accv, sa = acc;
                    wave_allreduce_pair(sv, sa, lane);
                    ll_part = sv + sa;
                }
                double prior_part = 0.0;
                if (usePrior) {
                    const double d = la - prior_mean;
                    prior_part = -0.5 * (d * d) / prior_sigmasq;
                }
                if (lane == 0) ctl[0] = ll_part + prior_part + cr_term;
            }
            sync();
            const double r = ctl[0];
            sync();
            DSQ_PROF(5);
            return r;
        };
        // ---- log_posterior AND dlog_posterior at one point, DispGene::lp_dlp                  -> ctl[0], ctl[1]
        auto eval_lp_dlp = [&](double la, bool withPrior, double &dlp_out) -> double {
            const double alpha = dexp(la);
            const double an1 = 1.0 / alpha;
            const double an2 = 1.0 / (alpha * alpha);
            double lg_an1 = 0.0, dg_an1 = 0.0;
            if constexpr (USE_W) dlgamma_digamma(an1, lg_an1, dg_an1);
            for (int j = tid; j < m; j += NT) {
                const double y = (double)y_s[j], mu = mu_s[j];
                const double ma = mu * alpha;
                const double opm = 1.0 + ma;
                const double rr = rcp1(opm);
                const double w0 = mu * rr;
                const double l1 = dlog(opm);
                if constexpr (USE_W) {
                    double lg, dg;
                    dlgamma_digamma(y + an1, lg, dg);
                    const double t = lg - lg_an1 - y * (l1 - la) - an1 * l1;
                    const double t2 = dg_an1 + l1 - ma * rr - dg + y * (alpha * rr);
                    const double w = w_s[j];
                    T1[j] = w * t;
                    T2[j] = w * t2;
                } else {
                    T1[j] = -(y * (l1 - la)) - an1 * l1;
                    T2[j] = l1 - ma * rr + y * (alpha * rr);
                }
                const bool keep = keep_row(j);
                wdbuf[j] = keep ? w0 : 0.0;
                wdbuf[(size_t)m + j] = keep ? -(w0 * w0) : 0.0;
            }
            sync();
            DSQ_PROF(1);
            double cr_lp = 0.0, cr_dlp = 0.0;
            if (useCR) {
                gram(DTag<2>{});
                DSQ_PROF(2);
                int sign = 1;
                lu_factor(Bm, sign);
                DSQ_PROF(3);
                lu_inverse(Bm, Bi);
                DSQ_PROF(4);
                if (wave == 0) {
                    const double detb = lu_det(Bm, sign);
                    wave_lds_sync();
                    const double tr1 = trace_sym(Bi, Bm + (size_t)P * P);
                    cr_lp = -0.5 * dlog(detb);
                    const double ddetb = detb * tr1;
                    cr_dlp = -0.5 * ddetb / detb;
                }
            }
            if (wave == 0) {
                const double acc = wave_sum(T1), acc2 = wave_sum(T2);
                double ll_part, ll_dpart;
                if constexpr (USE_W) {
                    double s1 = acc, s2 = acc2;
                    wave_allreduce_pair(s1, s2, lane);
                    ll_part = s1;
                    ll_dpart = an2 * s2;
                } else {
                    double accv = 0.0, accv2 = 0.0;
                    for (int q0 = 0; q0 <= nv; q0 += 64) {
                        const int q = q0 + lane;
                        const bool live = (q > 0) && (q <= nv);
                        double lg, dg;
                        dlgamma_digamma((live || q == 0) ? (live ? (double)dv[q - 1] : 0.0) + an1 : 20.0, lg, dg);
                        if (q0 == 0) { lg_an1 = lane_read(lg, 0); dg_an1 = lane_read(dg, 0); }
                        if (live) {
                            const double c = (double)dc[q - 1];
                            accv += c * (lg - lg_an1);
                            accv2 += c * (dg_an1 - dg);
                        }
                    }
                    double red[4] = {accv, accv2, acc, acc2};
                    wave_allreduce_many(red, lane);
                    ll_part = red[0] + red[2];
                    ll_dpart = an2 * (red[1] + red[3]);
                }
                double prior_part = 0.0, prior_dpart = 0.0;
                if (usePrior) {
                    const double d = la - prior_mean;
                    prior_part = -0.5 * (d * d) / prior_sigmasq;
                }
                if (withPrior) prior_dpart = -1.0 * (la - prior_mean) / prior_sigmasq;
                if (lane == 0) {
                    ctl[1] = (ll_dpart + cr_dlp) * alpha + prior_dpart;
                    ctl[0] = ll_part + prior_part + cr_lp;
                }
            }
            sync();
            const double r = ctl[0];
            dlp_out = ctl[1];
            sync();
            DSQ_PROF(5);
            return r;
        };

        if constexpr (MODE == 2) {
            // ---- d2log_posterior (src/DESeq2.cpp:111-158), DispGene::d2lp; its dlog_posterior(usePrior = false) is DispGene::dlp
            const double la = kp.log_alpha[g];
            const double alpha = dexp(la);
            const double an1 = 1.0 / alpha;
            const double an2 = 1.0 / (alpha * alpha);
            const double an3 = 1.0 / (alpha * (alpha * alpha));
            const double dg_an1 = ddigamma(an1), tg_an1 = dtrigamma(an1);
            for (int j = tid; j < m; j += NT) {
                const double y = (double)y_s[j], mu = mu_s[j];
                const double ma = mu * alpha, opm = 1.0 + ma;
                const double rr = 1.0 / opm;
                const double w0 = mu * rr;
                const double mpa = mu + an1;
                double t1 = dg_an1 + dlog(opm) - ma * rr - ddigamma(y + an1) + y * (1.0 / mpa);
                double t2 = -1.0 * an2 * tg_an1 + (mu * mu) * alpha * (1.0 / (opm * opm)) + an2 * dtrigamma(y + an1) +
                            an2 * y * (1.0 / (mpa * mpa));
                if constexpr (USE_W) {
                    const double w = w_s[j];
                    t1 = w * t1;
                    t2 = w * t2;
                }
                T1[j] = t1;
                T2[j] = t2;
                const bool keep = keep_row(j);
                wdbuf[j] = keep ? w0 : 0.0;
                wdbuf[(size_t)m + j] = keep ? -(w0 * w0) : 0.0;
                wdbuf[2 * (size_t)m + j] = keep ? 2.0 * (w0 * (w0 * w0)) : 0.0;
            }
            sync();
            double cr_term = 0.0;
            if (useCR) {
                gram(DTag<3>{});
                int sign = 1;
                lu_factor(Bm, sign);
                lu_inverse(Bm, Bi);
                sync();
                const double *B1 = Bm + (size_t)P * P, *B2 = Bm + 2 * (size_t)P * P;
                double detb = 0.0, tr1 = 0.0, tr3 = 0.0;
                if (wave == 0) {                                  // (the traces with B2 first: M is about to take its place)
                    detb = lu_det(Bm, sign);
                    tr1 = trace_sym(Bi, B1);
                    tr3 = trace_sym(Bi, B2);
                }
                sync();
                // M = Bi B1 (lane_mat_mul: c[i] = sum_k fma(Bi[i][k], B1[k][j]), k ascending), the rows round the waves
                if (lane < P)
                    for (int i = wave; i < P; i += NW) {
                        double acc = 0.0;
                        for (int k = 0; k < P; k++) acc = __builtin_fma(Bi[(size_t)i * P + k], B1[(size_t)k * P + lane], acc);
                        Mm[(size_t)i * P + lane] = acc;
                    }
                sync();
                if (wave == 0) {
                    // lane_trace_prod(M, M): the fma chain over (i, k) of M[i][k] M[k][i]
                    double tr2 = 0.0;
                    for (int i = 0; i < P; i++)
                        for (int k = 0; k < P; k++) tr2 = __builtin_fma(Mm[(size_t)i * P + k], Mm[(size_t)k * P + i], tr2);
                    const double ddetb = detb * tr1;
                    const double d2detb = detb * (tr1 * tr1 - tr2 + tr3);
                    const double rr = ddetb / detb;
                    cr_term = 0.5 * (rr * rr) - 0.5 * d2detb / detb;
                }
            }
            double part = 0.0;
            if (wave == 0) {
                const double s1 = wave_allreduce(wave_sum(T1)), s2 = wave_allreduce(wave_sum(T2));
                const double ll_part = -2.0 * an3 * s1 + an2 * s2;
                part = (ll_part + cr_term) * (alpha * alpha);
            }
            sync();
            // dlp(la, false): DispGene::dlp -- the derivative alone
            double dlp0;
            {
                double dg1 = 0.0;
                if constexpr (USE_W) dg1 = ddigamma(an1);
                for (int j = tid; j < m; j += NT) {
                    const double y = (double)y_s[j], mu = mu_s[j];
                    const double ma = mu * alpha;
                    const double rr = rcp1(1.0 + ma);
                    const double w0 = mu * rr;
                    if constexpr (USE_W) {
                        const double t = dg1 + dlog(1.0 + ma) - ma * rr - ddigamma(y + an1) + y * (alpha * rr);
                        T1[j] = w_s[j] * t;
                    } else {
                        T1[j] = dlog(1.0 + ma) - ma * rr + y * (alpha * rr);
                    }
                    const bool keep = keep_row(j);
                    wdbuf[j] = (useCR && keep) ? w0 : 0.0;
                    wdbuf[(size_t)m + j] = (useCR && keep) ? -(w0 * w0) : 0.0;
                }
                sync();
                double cr2 = 0.0;
                if (useCR) {
                    gram(DTag<2>{});
                    int sign = 1;
                    lu_factor(Bm, sign);
                    lu_inverse(Bm, Bi);
                    if (wave == 0) {
                        const double detb = lu_det(Bm, sign);
                        wave_lds_sync();
                        const double tr1 = trace_sym(Bi, Bm + (size_t)P * P);
                        const double ddetb = detb * tr1;
                        cr2 = -0.5 * ddetb / detb;
                    }
                }
                if (wave == 0) {
                    const double acc = wave_sum(T1);
                    double ll_sum;
                    if constexpr (USE_W) {
                        ll_sum = wave_allreduce(acc);
                    } else {
                        double accv = 0.0;
                        for (int q0 = 0; q0 <= nv; q0 += 64) {
                            const int q = q0 + lane;
                            const bool live = (q > 0) && (q <= nv);
                            const double dg = ddigamma((live || q == 0) ? (live ? (double)dv[q - 1] : 0.0) + an1 : 20.0);
                            if (q0 == 0) dg1 = lane_read(dg, 0);
                            if (live) accv += (double)dc[q - 1] * (dg1 - dg);
                        }
                        double sv = accv, sa = acc;
                        wave_allreduce_pair(sv, sa, lane);
                        ll_sum = sv + sa;
                    }
                    const double ll_part = an2 * ll_sum;
                    dlp0 = (ll_part + cr2) * alpha + 0.0;
                    const double prior_part = usePrior ? -1.0 / prior_sigmasq : 0.0;
                    if (lane == 0) kp.last_d2lp[g] = (part + dlp0) + prior_part;
                }
            }
        } else if constexpr (MODE == 1) {
            // fitDispGrid, src/DESeq2.cpp:492-510
            const int ng = kp.ngrid;
            const double delta = kp.grid[1] - kp.grid[0];
            int idx = 0;
            double best = 0.0;
            for (int t = 0; t < ng; t++) {
                const double v = eval_lp(kp.grid[t]);
                if (t == 0 || v > best) { best = v; idx = t; }
            }
            const double a_hat = kp.grid[idx];
            const double start = a_hat - delta, end = a_hat + delta;
            const double step = (end >= start) ? (end - start) / (double)(ng - 1) : -(start - end) / (double)(ng - 1);
            double afine = start;
            for (int t = 0; t < ng; t++) {
                const double a = (t == ng - 1) ? end : start + (double)t * step;
                const double v = eval_lp(a);
                if (t == 0 || v > best) { best = v; afine = a; }
            }
            if (tid == 0) kp.log_alpha[g] = afine;
        } else {
            // fitDisp, src/DESeq2.cpp:194-266 (the per-width kernel's statement of it: one fused evaluation per proposal)
            const double epsilon = 1.0e-4;
            double a = kp.log_alpha_in[g];
            double dlp;
            double lp = eval_lp_dlp(a, usePrior, dlp);
            double kappa = kp.kappa_0;
            const double initial_lp = lp, initial_dlp = dlp;
            double change = -1.0;
            int it = 0, it_acc = 0;
            for (int t = 0; t < kp.maxit; t++) {
                it++;
                const double a_propose = a + kappa * dlp;
                if (a_propose < -30.0) kappa = (-30.0 - a) / dlp;
                if (a_propose > 10.0) kappa = (10.0 - a) / dlp;
                const double a_try = a + kappa * dlp;
                double dlp_try = 0.0;
                const double lp_try = eval_lp_dlp(a_try, usePrior, dlp_try);
                const double theta_kappa = -1.0 * lp_try;
                const double theta_hat_kappa = -1.0 * lp - kappa * epsilon * (dlp * dlp);
                if (kp.force_iters > 0 && t + 1 >= kp.force_iters) break;
                if (theta_kappa <= theta_hat_kappa) {
                    it_acc++;
                    a = a_try;
                    const double lpnew = lp_try;
                    change = lpnew - lp;
                    if (change < kp.tol) { lp = lpnew; break; }
                    if (a < kp.min_log_alpha) break;
                    lp = lpnew;
                    dlp = dlp_try;
                    kappa = __builtin_fmin(kappa * 1.1, kp.kappa_0);
                    if (it_acc % 5 == 0) kappa = kappa / 2.0;
                } else {
                    kappa = kappa / 2.0;
                }
            }
            if (tid == 0) {
                kp.log_alpha[g] = a;
                kp.iter[g] = it;
                kp.iter_accept[g] = it_acc;
                kp.last_change[g] = change;
                kp.initial_lp[g] = initial_lp;
                kp.initial_dlp[g] = initial_dlp;
                kp.last_lp[g] = lp;
                kp.last_dlp[g] = dlp;
            }
        }
        sync();
        DSQ_PROF(6);
        if (tid == 0) ctl[2] = (double)(kp.work_counter ? atomicAdd(kp.work_counter + (MODE == 2 ? 1 : 0), 1) + (int)gridDim.x : wi + (int)gridDim.x);
        sync();
        wi = (int)ctl[2];
    }
    DSQ_PROF_FLUSH(dispw_prof);
}

// ---- launch ---------------------------------------------------------------------------------------------------------
// the rolled kernel serves a launch when the design has no cells, its true width is at least 11 and the rows are short
// enough for the serial Gram sums of the arithmetic spec; NW waves per gene by how many genes' slabs fit a CU
bool fit_disp_rolled_applies(const DispKernelParams &kp, int *p_true) {
    const int pt = kp.p - __builtin_popcountll(kp.padmask);
    if (p_true) *p_true = pt;
    if (getenv("DSQ_DISP_ROLLED") && atoi(getenv("DSQ_DISP_ROLLED")) == 0) return false;
    if (kp.ncell > 0 || pt < 11 || pt > 64) return false;
    if (kp.padmask != 0 && kp.padmask != (dsq_low_bits(kp.p) & ~dsq_low_bits(pt))) return false;   // (padding at the end)
    if (!(kp.m <= DSQ_SPEC_SERIAL_GRAM_MAXM)) return false;
    const size_t need = dispw_lds_doubles(kp.m, pt, kp.useWeights != 0, kp.last_d2lp ? 2 : 0) * sizeof(double);
    return need + 1024 <= 160 * 1024;
}

// the first pt columns of the (padded, column-major) design, sample-major
__global__ void dispw_transpose_kernel(const double *x, int m, int pt, double *xt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m * pt) xt[i] = x[(size_t)(i % pt) * m + i / pt];
}

enum { DSQ_WS_DISP_XT = 35 };         // grow-only workspace slot of the (device, stream), next to fit_disp.hip's DSQ_WS_DISP_DIST

template <bool USE_W, int MODE>
static hipError_t launch_dispw(const DispKernelParams &kp, int pt, const double *xt, hipStream_t st) {
    const size_t lds = dispw_lds_doubles(kp.m, pt, USE_W, MODE) * sizeof(double);
    const int fit = (int)((160 * 1024 - 512) / (lds + 256));
    int nw = fit >= 8 ? 1 : fit >= 4 ? 2 : fit >= 2 ? 4 : 8;
    const int force_nw = getenv("DSQ_WIDE_NW") ? atoi(getenv("DSQ_WIDE_NW")) : 0;
    if (force_nw == 1 || force_nw == 2 || force_nw == 4 || force_nw == 8) nw = force_nw;
    int bpc = fit;
    const int wmax = nw <= 2 ? 12 : 8;           // (the one- and two-wave builds are compiled for three waves per SIMD)
    if (bpc * nw > wmax) bpc = wmax / nw;
    if (bpc < 1) bpc = 1;
    const int cus = device_cu_count();
    long cap = (long)cus * bpc;
    int grid = (int)((long)kp.n < cap ? (long)kp.n : cap);
    if (kp.rows_few && grid > cus) grid = cus;
    if (grid < 1) grid = 1;
    const void *fn = nw == 1 ? (const void *)fit_disp_rolled_kernel<USE_W, MODE, 1> : nw == 2 ? (const void *)fit_disp_rolled_kernel<USE_W, MODE, 2>
                   : nw == 4 ? (const void *)fit_disp_rolled_kernel<USE_W, MODE, 4> : (const void *)fit_disp_rolled_kernel<USE_W, MODE, 8>;
    if (lds > 64 * 1024) {       // (the kernel has 256 B of static LDS -- the workgroup votes --: ask for what the launch needs, not for the CU's 160 KB)
        const hipError_t ea = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (ea != hipSuccess) return ea;
    }
    if (getenv("DSQ_VERBOSE")) fprintf(stderr, "[dsq] fit_disp_rolled p=%d m=%d mode=%d: lds=%zu, %d waves per gene, %d genes/CU\n", pt, kp.m, MODE, lds, nw, bpc);
    DispKernelParams kq = kp;
    int pp = pt;
    const double *xtp = xt;
    void *args[] = {&kq, &pp, &xtp};
#ifdef DSQ_WIDE_PROF
    {
        unsigned long long z[DSQ_PROF_SLOTS] = {}, h[DSQ_PROF_SLOTS];
        (void)hipMemcpyToSymbol(HIP_SYMBOL(dispw_prof), z, sizeof(z));
        const hipError_t e = hipLaunchKernel(fn, dim3(grid), dim3(64 * nw), args, lds, st);
        (void)hipStreamSynchronize(st);
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(dispw_prof), sizeof(h));
        double tot = 0;
        for (int q = 0; q < DSQ_PROF_SLOTS; q++) tot += (double)h[q];
        static const char *nm[8] = {"stage+distinct", "terms", "gram", "lu_factor", "inverse", "wave0 tail", "search logic", "-"};
        fprintf(stderr, "[dispw_prof] p=%d m=%d mode=%d nw=%d:", pt, kp.m, MODE, nw);
        for (int q = 0; q < 7; q++) fprintf(stderr, " %s %.1f%%", nm[q], 100.0 * (double)h[q] / (tot > 0 ? tot : 1));
        fprintf(stderr, "  (%.0f Mcycles of thread 0 over %d workgroups)\n", tot / 1e6, grid);
        return e;
    }
#endif
    return hipLaunchKernel(fn, dim3(grid), dim3(64 * nw), args, lds, st);
}

hipError_t launch_fit_disp_rolled(const DispKernelParams &kp, hipStream_t st, bool grid) {
    int pt = 0;
    (void)fit_disp_rolled_applies(kp, &pt);
    void *v = nullptr;
    if (capi_ws_get(DSQ_WS_DISP_XT, (size_t)kp.m * pt * sizeof(double), &v) != 0) return hipErrorOutOfMemory;
    double *xt = (double *)v;
    hipLaunchKernelGGL(dispw_transpose_kernel, dim3((kp.m * pt + 255) / 256), dim3(256), 0, st, kp.x, kp.m, pt, xt);
    if (grid) return kp.useWeights ? launch_dispw<true, 1>(kp, pt, xt, st) : launch_dispw<false, 1>(kp, pt, xt, st);
    hipError_t e = kp.useWeights ? launch_dispw<true, 0>(kp, pt, xt, st) : launch_dispw<false, 0>(kp, pt, xt, st);
    if (e != hipSuccess || !kp.last_d2lp) return e;
    return kp.useWeights ? launch_dispw<true, 2>(kp, pt, xt, st) : launch_dispw<false, 2>(kp, pt, xt, st);
}

}  // namespace dsq
