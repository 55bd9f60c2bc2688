// outlier.hip -- the count-outlier machinery around the fits, one
// wavefront per gene:
//   cooks_kernel     robustMethodOfMomentsDisp (R/core.R:2277-2299: trimmedCellVariance :2301-2324 /
//                    trimmedVariance :2326-2331), calculateCooksDistance (:2333-2340) and
//                    recordMaxCooks (:2349-2359) in one pass over the gene's row;
//   replace_kernel   replaceOutliers (:2069-2115): trimmed base mean, replacement counts, flags.
// The trimmed means need order statistics per design cell: the wave sorts the cell's values with
// a bitonic network in its private LDS slice (no workgroup barrier -- waves leave the gene loop
// at different times; a wave-scope fence orders the LDS traffic of successive stages).  A sorted
// multiset is unique, so the result does not depend on the network; the trimmed sums are then
// taken in wave order over the RANK (lane l takes ranks lo+l, lo+l+64, ...).
// Bytes per gene: cooks reads y (4m) + mu, H (16m) [+ nf 8m] and writes cooks (8m): HBM-bound
// in bytes, the sort is LDS/VALU work of O(m log^2 m / 64) per wave.
#include "dsq_internal.hpp"
#include "dsq_math.hpp"
#include "dsq_wave.hpp"
#include <type_traits>

namespace dsq {

// ascending bitonic sort of b[0..n2), n2 a power of two >= 2, by one wavefront
DSQ_DEV void wave_sort(double *b, int n2, int lane) {
    wave_lds_sync();
    for (int k = 2; k <= n2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = lane; t < (n2 >> 1); t += 64) {
                int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                int hi = lo | j;
                bool asc = (lo & k) == 0;
                double a = b[lo], c = b[hi];
                if ((a > c) == asc) { b[lo] = c; b[hi] = a; }
            }
            wave_lds_sync();
        }
    }
}

// ascending sort of a BITONIC sequence b[0..n2) (first falling, then rising -- what squared deviations of an
// ascending array from a value inside its range are): the last merge pass of the network is enough
DSQ_DEV void wave_merge_bitonic(double *b, int n2, int lane) {
    wave_lds_sync();
    for (int j = n2 >> 1; j > 0; j >>= 1) {
        for (int t = lane; t < (n2 >> 1); t += 64) {
            int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));
            int hi = lo | j;
            double a = b[lo], c = b[hi];
            if (a > c) { b[lo] = c; b[hi] = a; }
        }
        wave_lds_sync();
    }
}

DSQ_DEV int pow2_at_least(int n) {
    int v = 2;
    while (v < n) v <<= 1;
    return v;
}

// R's mean(x, trim) on the sorted buffer: order statistics lo..hi, lo = floor(n trim) + 1
DSQ_DEV double trimmed_mean_sorted(const double *sorted, int n, double trim, int lane) {
    int lo = (int)__builtin_floor((double)n * trim) + 1, hi = n + 1 - lo;
    double acc = 0.0;
    for (int r = lo - 1 + lane; r < hi; r += 64) acc += sorted[r];
    acc = wave_allreduce(acc);
    return acc / (double)(hi - lo + 1);
}

DSQ_DEV double wave_max(double v) {
    double o, a, b;
    o = lane_xor1(v); v = (o > v) ? o : v;
    o = lane_xor2(v); v = (o > v) ? o : v;
    o = lane_xor4(v); v = (o > v) ? o : v;
    o = lane_xor8(v); v = (o > v) ? o : v;
    lane_pair16(v, a, b); v = (b > a) ? b : a;
    lane_pair32(v, a, b); v = (b > a) ? b : a;
    return v;
}

static constexpr int kOutlierBatch = 4;      // trips whose loads are issued together (dsq_wave.hpp: sweep_batched)

__global__ void __launch_bounds__(256) cooks_kernel(CooksKernelParams kp) {
    extern __shared__ double smem[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int waves = blockDim.x >> 6;
    const int m = kp.m;
    double *cn = smem + (size_t)wave * (m + kp.sortcap);
    double *buf = cn + m;
    const double inf = __builtin_inf();
    const int nwork = DSQ_NWORK(kp);
    for (int wi = blockIdx.x * waves + wave; wi < nwork; wi += gridDim.x * waves) {
        const int g = DSQ_GENE(kp, wi);
        const int32_t *yg = kp.y + (size_t)g * kp.ld;
        const double *nfg = kp.nf_is_vector ? kp.nf : kp.nf + (size_t)g * kp.ld;
        double acc = 0.0;
        {
            int32_t yb[kOutlierBatch];
            double nb[kOutlierBatch];
            sweep_batched<kOutlierBatch>(m, lane, [&](int j, int b) { yb[b] = yg[j]; nb[b] = nfg[j]; },
                                         [&](int j, int b) {
                                             const double v = (double)yb[b] / nb[b];
                                             cn[j] = v;
                                             acc += v;
                                         });
        }
        const double mean_all = wave_allreduce(acc) / (double)m;
        double v;
        if (kp.any3) {
            v = -inf;
            for (int c = 0; c < kp.ncell; c++) {
                const int s0 = kp.cell_start[c], nc = kp.cell_start[c + 1] - s0;
                if (nc < 3) continue;
                const int tf = nc <= 3 ? 0 : (nc <= 23 ? 1 : 2);
                const double trim = tf == 0 ? 1.0 / 3.0 : (tf == 1 ? 1.0 / 4.0 : 1.0 / 8.0);
                const double scale = tf == 0 ? 2.04 : (tf == 1 ? 1.86 : 1.51);
                const int n2 = pow2_at_least(nc);
                double ve;
                // cells of up to 512 samples: both sorts in registers (more values per lane cost the kernel its occupancy: C3
                // 0.48 -> 0.63 ms with the 16 / 32-value variants compiled in) (element lane * R + r), the sorted values
                // through LDS only for the rank-ordered trimmed sums
                auto in_regs = [&](auto rtag) {
                    constexpr int R = decltype(rtag)::value;
                    double w[R];
                    wave_lds_sync();
                    _Pragma("unroll")
                    for (int r = 0; r < R; r++) {
                        const int e = lane * R + r;
                        w[r] = e < nc ? cn[kp.perm[s0 + e]] : inf;
                    }
                    wave_sort_regs<double, R>(w, lane);
                    _Pragma("unroll")
                    for (int r = 0; r < R; r++) buf[lane * R + r] = w[r];
                    wave_lds_sync();
                    const double cm = trimmed_mean_sorted(buf, nc, trim, lane);
                    // squared deviations of the SORTED values: falling, then rising (+inf padding keeps rising)
                    _Pragma("unroll")
                    for (int r = 0; r < R; r++) {
                        const double d = w[r] - cm;
                        w[r] = (lane * R + r) < nc ? d * d : inf;
                    }
                    wave_merge_regs<double, R>(w, lane);
                    wave_lds_sync();
                    _Pragma("unroll")
                    for (int r = 0; r < R; r++) buf[lane * R + r] = w[r];
                    wave_lds_sync();
                    return scale * trimmed_mean_sorted(buf, nc, trim, lane);
                };
                if (n2 <= 64) ve = in_regs(std::integral_constant<int, 1>());
                else if (n2 == 128) ve = in_regs(std::integral_constant<int, 2>());
                else if (n2 == 256) ve = in_regs(std::integral_constant<int, 4>());
                else if (n2 == 512) ve = in_regs(std::integral_constant<int, 8>());
                else {
                    wave_lds_sync();
                    for (int k = lane; k < n2; k += 64) buf[k] = k < nc ? cn[kp.perm[s0 + k]] : inf;
                    wave_sort(buf, n2, lane);
                    const double cm = trimmed_mean_sorted(buf, nc, trim, lane);
                    wave_lds_sync();
                    // squared deviations of the SORTED values: falling, then rising (+inf padding keeps rising)
                    for (int k = lane; k < nc; k += 64) {
                        double d = buf[k] - cm;
                        buf[k] = d * d;
                    }
                    wave_merge_bitonic(buf, n2, lane);
                    ve = scale * trimmed_mean_sorted(buf, nc, trim, lane);
                }
                if (ve > v) v = ve;
            }
        } else {
            const int n2 = pow2_at_least(m);
            wave_lds_sync();
            for (int k = lane; k < n2; k += 64) buf[k] = k < m ? cn[k] : inf;
            wave_sort(buf, n2, lane);
            const double rm = trimmed_mean_sorted(buf, m, 1.0 / 8.0, lane);
            wave_lds_sync();
            for (int k = lane; k < m; k += 64) {
                double d = buf[k] - rm;
                buf[k] = d * d;
            }
            wave_merge_bitonic(buf, n2, lane);
            v = 1.51 * trimmed_mean_sorted(buf, m, 1.0 / 8.0, lane);
        }
        double alpha = (v - mean_all) / (mean_all * mean_all);
        alpha = __builtin_fmax(alpha, 0.04);
        const double *mug = kp.mu + (size_t)g * kp.ld, *hg = kp.H + (size_t)g * kp.ld;
        double *ckg = kp.cooks + (size_t)g * kp.ld;
        double mx = -inf;
        int isnan_ = 0, anyc = 0;
        {
            double mb[kOutlierBatch], hb[kOutlierBatch];
            int32_t yb[kOutlierBatch], ib[kOutlierBatch];
            sweep_batched<kOutlierBatch>(m, lane,
                                         [&](int j, int b) { mb[b] = mug[j]; hb[b] = hg[j]; yb[b] = yg[j]; ib[b] = kp.in3[j]; },
                                         [&](int j, int b) {
                                             const double mj = mb[b], hj = hb[b], yj = (double)yb[b];
                                             const double V = mj + alpha * (mj * mj);
                                             const double d = yj - mj;
                                             const double pr = (d * d) / V;
                                             const double omh = 1.0 - hj;
                                             const double ck = pr / (double)kp.p * hj / (omh * omh);
                                             ckg[j] = ck;
                                             if (ib[b]) {
                                                 anyc = 1;
                                                 if (ck != ck) isnan_ = 1;
                                                 if (ck > mx) mx = ck;
                                             }
                                         });
        }
        mx = wave_max(mx);
        if (lane == 0) {
            kp.robustDisp[g] = alpha;
            double out = __any(isnan_) ? __builtin_nan("") : mx;
            kp.maxCooks[g] = (m > kp.p && __any(anyc)) ? out : __builtin_nan("");
        }
    }
}

__global__ void __launch_bounds__(256) replace_kernel(ReplaceKernelParams kp) {
    extern __shared__ double smem[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int waves = blockDim.x >> 6;
    const int m = kp.m;
    double *buf = smem + (size_t)wave * kp.sortcap;
    const double inf = __builtin_inf();
    const int n2 = pow2_at_least(m);
    const int nwork = DSQ_NWORK(kp);
    for (int wi = blockIdx.x * waves + wave; wi < nwork; wi += gridDim.x * waves) {
        const int g = DSQ_GENE(kp, wi);
        const int32_t *yg = kp.y + (size_t)g * kp.ld;
        const double *nfg = kp.nf_is_vector ? kp.nf : kp.nf + (size_t)g * kp.ld;
        const double *ckg = kp.cooks + (size_t)g * kp.ld;
        int any = 0;
        {
            double cb[kOutlierBatch];
            sweep_batched<kOutlierBatch>(m, lane, [&](int j, int b) { cb[b] = ckg[j]; },
                                         [&](int, int b) { if (cb[b] > kp.cutoff) any = 1; });
        }
        any = __any(any);
        int32_t *og = kp.newCounts + (size_t)g * kp.ld;
        if (!any) {
            // no distance above the cutoff (almost every gene): the counts pass through, and the trimmed base mean
            // -- the sort -- is never needed
            int32_t yb[kOutlierBatch];
            sweep_batched<kOutlierBatch>(m, lane, [&](int j, int b) { yb[b] = yg[j]; }, [&](int j, int b) { og[j] = yb[b]; });
        } else {
            wave_lds_sync();
            for (int k = lane; k < n2; k += 64) buf[k] = k < m ? (double)yg[k] / nfg[k] : inf;
            wave_sort(buf, n2, lane);
            const double tbm = trimmed_mean_sorted(buf, m, kp.trim, lane);
            for (int j = lane; j < m; j += 64) {
                int rep = (int)(tbm * nfg[j]);
                og[j] = (ckg[j] > kp.cutoff && kp.replaceable[j]) ? rep : yg[j];
            }
        }
        if (lane == 0) kp.replace[g] = any ? 1 : 0;
    }
}

// waves per workgroup so that the wave-private LDS slices fit; 0 = the row does not fit at all
static int outlier_waves(size_t doubles_per_wave) {
    const size_t cap = 160 * 1024;
    int w = 4;
    while (w > 0 && (size_t)w * doubles_per_wave * 8 > cap) w >>= 1;
    return w;
}

template <typename KP, typename F>
static hipError_t launch_outlier(F fn, const KP &kp, size_t doubles_per_wave, hipStream_t st, bool *ok) {
    int waves = outlier_waves(doubles_per_wave);
    *ok = waves > 0;
    if (!*ok) return hipSuccess;
    size_t lds = (size_t)waves * doubles_per_wave * 8;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    int blocks = (kp.n + waves - 1) / waves;
    int per_cu = (int)((160 * 1024) / (lds ? lds : 1));
    if (per_cu > 8) per_cu = 8;
    if (per_cu < 1) per_cu = 1;
    int cap = device_cu_count() * per_cu;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(fn, dim3(blocks), dim3(64 * waves), lds, st, kp);
    return hipGetLastError();
}

hipError_t launch_cooks(const CooksKernelParams &kp0, hipStream_t st, bool *ok) {
    CooksKernelParams kp = kp0;
    if (kp.sortcap < 64) kp.sortcap = 64;      // the register sort hands back 64 R elements (R >= 1), padding included
    return launch_outlier(cooks_kernel, kp, (size_t)kp.m + kp.sortcap, st, ok);
}

hipError_t launch_replace(const ReplaceKernelParams &kp, hipStream_t st, bool *ok) {
    return launch_outlier(replace_kernel, kp, (size_t)kp.sortcap, st, ok);
}

}  // namespace dsq
