// beta_prior.hip -- dsq_beta_prior_var: estimateBetaPriorVar (R/core.R:1601-1689) on the n x p matrix of MLE
// coefficients, for the betaPrior = TRUE branch of nbinomWaldTest inside dsq_deseq (deseq_host.hip).
//
// An all-gene step on n-vectors like the dispersion trend: per model-matrix column (and, for the expanded model matrix,
// per pairwise contrast of a factor's levels, addAllContrasts R/expanded.R:76-98) the prior variance is
//     matchWeightedUpperQuantileForVariance(x, w)  =  (wtd.quantile(|x|, w, 1 - 0.05, normwt = TRUE) / qnorm(1 - 0.05 / 2))^2
// (R/core.R:2416-2419) over the rows with |x| < 10 (:1651), w = 1 / (1 / baseMean + dispFit) (:1641-1642), the intercept
// set to 1e6 (:1669-1671), and for the expanded matrix the mean over a factor's level and contrast variances handed to all
// of its level columns (averagePriorsOverLevels, R/expanded.R:20-73).  Hmisc's wtd.quantile (R/core.R:2762-2800) is a
// stable sort of the values, the weight sums of the distinct values, their running sum and a step interpolation.
//
// Host code: the inputs are what the chain brings down anyway (n (p + 3) doubles), the sort is a stable LSD radix sort on
// the bit patterns (|x| >= 0: the patterns order like the values), every sum is SEQUENTIAL in the order stated below --
// the one definition deseq2_amd/core.py (estimateBetaPriorVar / Hmisc_wtd_quantile) follows as well, so the two give the
// same bits (tests/test_capi_cpu.py::test_beta_prior_var_equals_the_host_mirror).  Columns are independent: a few
// threads take them in turn.
#include "../../include/deseq2_mi355x.h"
#include "dsq_internal.hpp"

#include <atomic>
#include <cmath>
#include <cstring>
#include <thread>
#include <vector>

namespace dsq {
namespace {

const double kQnorm975 = 1.959963984540054;      // qnorm(1 - 0.05 / 2)

// stable ascending order of non-negative doubles: 8 passes of 8 bits over the bit patterns
static void radix_order(const std::vector<double> &v, std::vector<uint32_t> *order) {
    const size_t n = v.size();
    std::vector<uint64_t> key(n), key2(n);
    std::vector<uint32_t> idx(n), idx2(n);
    for (size_t i = 0; i < n; i++) { memcpy(&key[i], &v[i], 8); idx[i] = (uint32_t)i; }
    for (int pass = 0; pass < 8; pass++) {
        const int sh = 8 * pass;
        size_t cnt[257];
        memset(cnt, 0, sizeof cnt);
        for (size_t i = 0; i < n; i++) cnt[((key[i] >> sh) & 0xff) + 1]++;
        bool trivial = false;
        for (int b = 0; b < 256; b++) { if (cnt[b + 1] == n) trivial = true; cnt[b + 1] += cnt[b]; }
        if (trivial) continue;                       // all keys share this digit: the pass is the identity
        for (size_t i = 0; i < n; i++) {
            const size_t d = cnt[(key[i] >> sh) & 0xff]++;
            key2[d] = key[i]; idx2[d] = idx[i];
        }
        key.swap(key2); idx.swap(idx2);
    }
    order->swap(idx);
}

// Hmisc::wtd.quantile(x, weights, probs = prob, normwt = TRUE) for ONE probability (R/core.R:2762-2800)
static double wtd_quantile(std::vector<double> &x, std::vector<double> &w, double prob) {
    // rows with NA or zero weight are dropped (:2771-2775)
    size_t N = 0;
    for (size_t i = 0; i < x.size(); i++)
        if (!(std::isnan(w[i]) || w[i] == 0.0)) { x[N] = x[i]; w[N] = w[i]; N++; }
    x.resize(N); w.resize(N);
    if (N == 0) return NAN;
    double wsum = 0.0;                               // sum(weights), in row order
    for (size_t i = 0; i < N; i++) wsum += w[i];
    for (size_t i = 0; i < N; i++) w[i] = w[i] * (double)N / wsum;      // weights * length(x) / sum(weights)
    std::vector<uint32_t> o;
    radix_order(x, &o);
    // distinct values (ascending) with the sums of their weights, each in sorted order; cs = their running sum
    std::vector<double> ux, cs;
    ux.reserve(N); cs.reserve(N);
    double run = 0.0, tot = 0.0;
    for (size_t k = 0; k < N; k++) {
        const double v = x[o[k]];
        run += w[o[k]];
        if (k + 1 == N || x[o[k + 1]] != v) {
            tot += run;
            ux.push_back(v); cs.push_back(tot);
            run = 0.0;
        }
    }
    const double n = tot;
    const double order = 1.0 + (n - 1.0) * prob;
    const double fl = std::floor(order);
    const double low = fl > 1.0 ? fl : 1.0;
    const double high = (low + 1.0 < n) ? low + 1.0 : n;
    const double frac = std::fmod(order, 1.0);
    auto stepq = [&](double q) {                     // approx(cumsum(wts), x, method = "constant", f = 1, rule = 2)
        size_t lo = 0, hi = cs.size();
        while (lo < hi) { const size_t mid = (lo + hi) / 2; if (cs[mid] < q) lo = mid + 1; else hi = mid; }
        return ux[lo < ux.size() ? lo : ux.size() - 1];
    };
    return (1.0 - frac) * stepq(low) + frac * stepq(high);
}

}  // namespace

int beta_prior_var(const DsqBetaPriorArgs *a, double *out) {
    const int n = a->n, p = a->p;
    // the columns whose variance is matched: the p design columns, then (expanded) the level contrasts factor by factor
    struct Col { int i, j; };                        // value = beta[, i] - (j >= 0 ? beta[, j] : 0)
    std::vector<Col> cols;
    for (int c = 0; c < p; c++) cols.push_back({c, -1});
    int nfac = 0;
    for (int c = 0; c < p; c++) if (a->coef_factor[c] > nfac) nfac = a->coef_factor[c];
    std::vector<int> col_factor(p, 0);
    for (int c = 0; c < p; c++) col_factor[c] = a->coef_factor[c];
    if (a->expanded)
        for (int f = 1; f <= nfac; f++) {
            std::vector<int> idx;
            for (int c = 0; c < p; c++) if (a->coef_factor[c] == f) idx.push_back(c);
            for (size_t j = 0; j + 1 < idx.size(); j++)
                for (size_t i = j + 1; i < idx.size(); i++) { cols.push_back({idx[i], idx[j]}); col_factor.push_back(f); }
        }
    // rows that enter: the ones that are not all zero (objectNZ, R/core.R:1610); weights 1 / (1 / baseMean + dispFit)
    std::vector<int> rows;
    rows.reserve(n);
    for (int g = 0; g < n; g++) if (!a->allZero[g]) rows.push_back(g);
    std::vector<double> wrow(rows.size());
    for (size_t k = 0; k < rows.size(); k++) wrow[k] = 1.0 / (1.0 / a->baseMean[rows[k]] + a->dispFit[rows[k]]);
    std::vector<double> pv(cols.size(), 0.0);
    std::atomic<size_t> next{0};
    auto work = [&]() {
        std::vector<double> x, w;
        for (;;) {
            const size_t c = next.fetch_add(1);
            if (c >= cols.size()) break;
            if (cols[c].j < 0 && a->coef_factor[cols[c].i] == 0) { pv[c] = 1e6; continue; }      // the intercept (:1669-1671)
            x.clear(); w.clear();
            const double *bi = a->mle_beta + (size_t)n * cols[c].i, *bj = cols[c].j >= 0 ? a->mle_beta + (size_t)n * cols[c].j : nullptr;
            if (rows.size() == 1) {                               // a one-gene object: (betaMatrix)^2 (R/core.R:1647,1662)
                const double v = bj ? bi[rows[0]] - bj[rows[0]] : bi[rows[0]];
                pv[c] = v * v;
                continue;
            }
            for (size_t k = 0; k < rows.size(); k++) {
                const double v = bj ? bi[rows[k]] - bj[rows[k]] : bi[rows[k]];
                const double av = std::fabs(v);
                if (av < 10.0) { x.push_back(av); w.push_back(wrow[k]); }                          // useFinite (:1651)
            }
            if (x.empty()) { pv[c] = 1e6; continue; }                                               // :1652-1653
            const double sd = wtd_quantile(x, w, 1.0 - a->upperQuantile) / kQnorm975;
            pv[c] = sd * sd;
        }
    };
    {
        unsigned hw = std::thread::hardware_concurrency();
        size_t T = cols.size() < 8 ? cols.size() : 8;
        if (hw && T > hw) T = hw;
        if ((size_t)n * cols.size() < 400000) T = 1;
        std::vector<std::thread> th;
        for (size_t t = 1; t < T; t++) th.emplace_back(work);
        work();
        for (auto &t : th) t.join();
    }
    if (!a->expanded) {
        for (int c = 0; c < p; c++) out[c] = pv[c];
        return DSQ_OK;
    }
    // averagePriorsOverLevels (R/expanded.R:20-73): a factor's level columns all get the mean of its level and contrast
    // variances (summed in column order, the contrasts after the design columns); other columns keep theirs
    std::vector<double> meanvar(nfac + 1, 0.0);
    for (int f = 1; f <= nfac; f++) {
        double s = 0.0;
        int k = 0;
        for (size_t c = 0; c < cols.size(); c++) if (col_factor[c] == f) { s += pv[c]; k++; }
        meanvar[f] = k ? s / (double)k : 0.0;
    }
    for (int j = 0; j < a->p_prior; j++) {
        const int f = a->prior_coef_factor[j];
        if (f == 0) out[j] = 1e6;
        else if (f > 0) out[j] = (f <= nfac) ? meanvar[f] : 0.0;
        else {
            const int s = a->prior_coef_src ? a->prior_coef_src[j] : -1;
            out[j] = (s >= 0 && s < p) ? pv[s] : 0.0;
        }
        if (!(out[j] > 0.0)) return capi_fail(DSQ_ERR_FIT, "beta prior is not greater than 0 (column %d of the expanded model matrix)", j);
    }
    return DSQ_OK;
}

}  // namespace dsq

extern "C" int dsq_beta_prior_var(const DsqBetaPriorArgs *a, double *betaPriorVar) {
    using namespace dsq;
    if (!a || !betaPriorVar) return capi_fail(DSQ_ERR_ARG, "NULL args/out");
    if (a->n < 1 || a->p < 1 || !a->mle_beta || !a->baseMean || !a->dispFit || !a->allZero || !a->coef_factor)
        return capi_fail(DSQ_ERR_ARG, "dsq_beta_prior_var: NULL input or bad dimensions");
    if (a->expanded && (a->p_prior < 1 || !a->prior_coef_factor)) return capi_fail(DSQ_ERR_ARG, "expanded model matrix: p_prior / prior_coef_factor");
    if (!(a->upperQuantile > 0.0 && a->upperQuantile < 1.0)) return capi_fail(DSQ_ERR_ARG, "upperQuantile");
    if (a->upperQuantile != 0.05) return capi_fail(DSQ_ERR_UNSUPPORTED, "upperQuantile = %g: only the default 0.05 (qnorm(0.975) is a constant here)", a->upperQuantile);
    return beta_prior_var(a, betaPriorVar);
}
