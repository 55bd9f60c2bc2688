// fit_beta_common.hpp -- pieces of the IRLS shared by the per-width fitBeta kernels (fit_beta.hip) and the rolled kernel
// of the wide designs (fit_beta_wide.hip): the mu-independent part of the deviance and the off-branch of its sweep.
#pragma once
#include "dsq_internal.hpp"
#include "dsq_math.hpp"
#include "dsq_wave.hpp"

namespace dsq {

// the rarely taken branch of the deviance sweep (a sample whose log density does not follow the closed split, e.g. a
// count below 1e-10 size): the full dnbinom_mu, kept out of line so that its registers do not count against the sweep's
__device__ __noinline__ static double nb_offbranch(double y, double size, double mu) { return dnbinom_mu_log(y, size, mu); }

// (cell_dev_closed / nb_split_const: dsq_math.hpp)
// K = sum_j [wts_j] K_j, the mu-independent part of the IRLS deviance, samples in their natural order:
// K_j = [saddle-point constants of dnbinom_mu, logarithms folded] + n log1p(alpha y) - y log y + y log nf_j  (0 for y = 0);
// kprime (optional): the same without the y log nf_j -- what nbinomLogLike adds to its own sweep (aux.hip)
// scr (round 5; optional, 640 doubles of wave-private LDS): the saddle-point constants depend on a sample only through its
// COUNT -- nb_split_const(y, alpha, size) is two logarithms and two stirlerr evaluations, each with several branches that a
// wave with mixed counts executes one after the other (~ 800 VALU instructions per 64 samples; tools/r05_fb_split.sh: the
// constants pass was 40 % of fit_beta_cell<4> at C3) -- so they are evaluated ONCE PER DISTINCT COUNT below kIrlsTab and
// looked up per sample: presence flags -> ascending list of the counts present -> one evaluation trip per 64 of them ->
// table (base, t) indexed by the count.  Counts from kIrlsTab up take the direct evaluation as before.  The same function
// of the same arguments, the per-sample terms added in the same order: K and K' keep their bits.
static constexpr int kIrlsTab = 256;
template <bool USE_W>
DSQ_DEV double irls_constants(const int32_t *yg, const double *nfg, const double *wg, int m, int lane, double alpha,
                              double size, bool fast, const double *lnf = nullptr, double *kprime = nullptr,
                              double *scr = nullptr, int T = kIrlsTab) {
    // T: counts below it go through the table (64, 128 or 256: scr holds T / 2 + 2 T doubles)
    if (kprime) *kprime = 0.0;
    if (!fast) return 0.0;
    const double st_size = dstirlerr(size);
    double *tab = nullptr;
    if (scr) {
        int32_t *flag = reinterpret_cast<int32_t *>(scr);            // T int32: presence flags, then the list of counts present
        tab = scr + T / 2;                                           // T x (base, t)
        wave_lds_sync();
        for (int t = lane; t < T; t += 64) flag[t] = 0;
        wave_lds_sync();
        for (int j = lane; j < m; j += 64) {
            const int yi = yg[j];
            if (yi > 0 && yi < T) flag[yi] = 1;
        }
        wave_lds_sync();
        int pres[kIrlsTab / 64];
        _Pragma("unroll")
        for (int t = 0; t < kIrlsTab / 64; t++) pres[t] = (64 * t < T) ? flag[64 * t + lane] : 0;
        wave_lds_sync();
        int nv = 0;
        _Pragma("unroll")
        for (int t = 0; t < kIrlsTab / 64; t++) {
            const unsigned long long mask = __ballot(pres[t] != 0);
            const int rank = nv + __popcll(mask & ((1ull << lane) - 1ull));
            if (pres[t] != 0) flag[rank] = 64 * t + lane;
            nv += __popcll(mask);
        }
        wave_lds_sync();
        for (int q0 = 0; q0 < nv; q0 += 64) {
            const int q = q0 + lane;
            const int v = flag[q < nv ? q : nv - 1];                 // (lanes past the end repeat the last count: no divergence)
            double base, t;
            nb_split_const((double)v, alpha, size, st_size, base, t);
            if (q < nv) { tab[2 * v] = base; tab[2 * v + 1] = t; }
        }
        wave_lds_sync();
    }
    double kacc = 0.0, pacc = 0.0;
    for (int j = lane; j < m; j += 64) {
        const int yi = yg[j];
        const double y = (double)yi;
        double kj = 0.0, pj = 0.0;
        if (y != 0.0 && cell_dev_closed(y, size, fast)) {
            double base = 0.0, t = 0.0;
            const bool direct = !(tab && (unsigned)yi < (unsigned)T);      // (a negative count: never a table index)
            if (!direct) { base = tab[2 * yi]; t = tab[2 * yi + 1]; }
            if (__any(direct)) {                                      // (wave-uniform: a trip of small counts skips the evaluation)
                if (direct) nb_split_const(y, alpha, size, st_size, base, t);
            }
            // (lnf: log nf_j from the block's table when the factors are the size-factor vector -- the same function of
            // the same value, evaluated once per block instead of once per gene)
            kj = base + (t + y * (lnf ? lnf[j] : dlog(nfg[j])));
            pj = base + t;
        }
        if constexpr (USE_W) { kacc += wg[j] * kj; pacc += wg[j] * pj; }
        else { kacc += kj; pacc += pj; }
    }
    if (kprime) {
        wave_allreduce_pair(kacc, pacc, lane);         // (the bits of two butterflies)
        *kprime = pacc;
        return kacc;
    }
    return wave_allreduce(kacc);
}

}  // namespace dsq
