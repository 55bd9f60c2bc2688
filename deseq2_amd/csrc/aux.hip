// aux.hip -- the O(n m p) steps the reference does in R around the native fits, as one-wave-
// per-gene kernels:
//   prefit_moments   baseMean / baseVar / allZero (R/core.R:2138-2146), roughDispEstimate
//                    (R/core.R:2422-2437 with linearModelMu :2454-2463) and the QR least-squares
//                    start values of R/fitNbinomGLMs.R:139-145 -- one launch instead of a dozen
//                    dense host passes;
//   nbinom_loglike   nbinomLogLike (R/core.R:2208-2217) for fitNbinomGLMs.R:182 / nbinomLRT.
// Both are HBM-bound in bytes (12..20 m per gene) but still VALU-heavy per byte (one log resp.
// one NB density per sample).  Same wave-order sums as the fit kernels.
#include "dsq_internal.hpp"
#include "dsq_math.hpp"
#include "dsq_wave.hpp"

namespace dsq {

static constexpr int kAuxBatch = 4;          // trips whose loads are issued together (dsq_wave.hpp: sweep_batched)

static inline int aux_grid_fwd(int n) {
    int blocks = (n + 3) / 4;
    int cap = device_cu_count() * 8;
    return blocks < cap ? (blocks < 1 ? 1 : blocks) : cap;
}

// TILED (round 5; long rows): the passes over Q and over A = X R^-1 read p doubles per SAMPLE of design-only data -- 160 KB per
// gene at m = 2000, p = 10, every gene again through L2 (C4: 19 GB per launch, 2.3 ms for a kernel with 0.3 ms of arithmetic).
// The four waves of a block walk the samples in lockstep tiles of kTileS samples whose Q / A columns they stage ONCE in LDS
// for their four genes.  A lane still adds its samples j = lane, lane + 64, ... in increasing order: the same sums.
static constexpr int kTileS = 256;
__host__ __device__ static inline bool aux_tiled(int m, int p) { return (size_t)m * p >= 8192 && p <= 24; }   // (tile <= 48 KB)

template <int P, bool USE_W, bool TILED>
__global__ void __launch_bounds__(256) prefit_kernel(PrefitKernelParams kp) {
    extern __shared__ __attribute__((aligned(16))) double tile[];       // TILED: P x kTileS doubles
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int waves = blockDim.x >> 6;
    const int m = kp.m;
    double rr[P][P];
#pragma unroll
    for (int c = 0; c < P; c++)
#pragma unroll
        for (int k = 0; k < P; k++) rr[c][k] = kp.r[c + P * k];
    const int nwork = DSQ_NWORK(kp);
    // (TILED: every wave of the block runs the same number of rounds -- the block barriers of the tile loops -- and a wave
    //  past the end of the work list goes through them idle)
    for (int wbase = blockIdx.x * waves; wbase < nwork; wbase += gridDim.x * waves) {
        const int wi = wbase + wave;
        const bool active = wi < nwork;
        if (!TILED && !active) break;
        const int g = DSQ_GENE(kp, active ? wi : nwork - 1);
        const int32_t *yg = kp.y + (size_t)g * kp.ld;
        const double *nfg = kp.nf_is_vector ? kp.nf : kp.nf + (size_t)g * kp.ld;
        const double *wg = USE_W ? kp.weights + (size_t)g * kp.ld : nullptr;
        // (every pass re-reads the row through L1 / L2: the loads of kAuxBatch trips are issued together, sweep_batched)
        int32_t yb[kAuxBatch];
        double nb[kAuxBatch], wb[kAuxBatch];
        auto load_row = [&](int j, int b) {
            yb[b] = yg[j]; nb[b] = nfg[j];
            if constexpr (USE_W) wb[b] = wg[j];
        };
        double a2[2] = {0.0, 0.0};
        sweep_batched<kAuxBatch>(m, lane, load_row, [&](int, int b) {
            double yy = (double)yb[b];
            double cn = yy / nb[b];
            if constexpr (USE_W) cn = wb[b] * cn;
            a2[0] += cn;
            a2[1] += yy;
        });
        wave_allreduce_many(a2, lane);
        const double mean = a2[0] / (double)m;
        double av = 0.0;
        sweep_batched<kAuxBatch>(m, lane, load_row, [&](int, int b) {
            double cn = (double)yb[b] / nb[b];
            if constexpr (USE_W) cn = wb[b] * cn;
            double dlt = cn - mean;
            av += dlt * dlt;
        });
        av = wave_allreduce(av);
        double tu[2 * P];
#pragma unroll
        for (int c = 0; c < 2 * P; c++) tu[c] = 0.0;
        double ae = 0.0;
        if constexpr (TILED) {
            static_assert(kTileS == 64 * kAuxBatch, "one tile = one batch of trips");
            auto stage = [&](const double *src, int t0) {
                __syncthreads();                                          // (the previous tile has been consumed)
                const int ts = (m - t0) < kTileS ? (m - t0) : kTileS;
                for (int idx = threadIdx.x; idx < P * kTileS; idx += blockDim.x) {
                    const int c = idx / kTileS, jj = idx - c * kTileS;
                    if (jj < ts) tile[idx] = src[(size_t)c * m + t0 + jj];
                }
                __syncthreads();
            };
            for (int t0 = 0; t0 < m; t0 += kTileS) {
                stage(kp.q, t0);
                _Pragma("unroll")
                for (int b = 0; b < kAuxBatch; b++) { const int j = t0 + 64 * b + lane; if (j < m) { yb[b] = yg[j]; nb[b] = nfg[j]; } }
                _Pragma("unroll")
                for (int b = 0; b < kAuxBatch; b++) {
                    const int j = t0 + 64 * b + lane;
                    if (j < m) {
                        double yn = (double)yb[b] / nb[b];
                        double ly = dlog(yn + 0.1);
#pragma unroll
                        for (int c = 0; c < P; c++) {
                            double qv = tile[c * kTileS + 64 * b + lane];
                            tu[c] += yn * qv;
                            tu[P + c] += ly * qv;
                        }
                    }
                }
            }
            wave_allreduce_many(tu, lane);
            for (int t0 = 0; t0 < m; t0 += kTileS) {
                stage(kp.a, t0);
                _Pragma("unroll")
                for (int b = 0; b < kAuxBatch; b++) { const int j = t0 + 64 * b + lane; if (j < m) { yb[b] = yg[j]; nb[b] = nfg[j]; } }
                _Pragma("unroll")
                for (int b = 0; b < kAuxBatch; b++) {
                    const int j = t0 + 64 * b + lane;
                    if (j < m) {
                        double yn = (double)yb[b] / nb[b];
                        double mu = tu[0] * tile[64 * b + lane];
#pragma unroll
                        for (int c = 1; c < P; c++) mu = __builtin_fma(tu[c], tile[c * kTileS + 64 * b + lane], mu);
                        mu = __builtin_fmax(1.0, mu);
                        double d = yn - mu;
                        ae += (d * d - mu) / (mu * mu);
                    }
                }
            }
        } else {
        sweep_batched<kAuxBatch>(m, lane, [&](int j, int b) { yb[b] = yg[j]; nb[b] = nfg[j]; }, [&](int j, int b) {
            double yn = (double)yb[b] / nb[b];
            double ly = dlog(yn + 0.1);
#pragma unroll
            for (int c = 0; c < P; c++) {
                double qv = kp.q[(size_t)c * m + j];
                tu[c] += yn * qv;
                tu[P + c] += ly * qv;
            }
        });
        wave_allreduce_many(tu, lane);
        sweep_batched<kAuxBatch>(m, lane, [&](int j, int b) { yb[b] = yg[j]; nb[b] = nfg[j]; }, [&](int j, int b) {
            double yn = (double)yb[b] / nb[b];
            double mu = tu[0] * kp.a[j];
#pragma unroll
            for (int c = 1; c < P; c++) mu = __builtin_fma(tu[c], kp.a[(size_t)c * m + j], mu);
            mu = __builtin_fmax(1.0, mu);
            double d = yn - mu;
            ae += (d * d - mu) / (mu * mu);
        });
        }
        ae = wave_allreduce(ae);
        double b[P];
#pragma unroll
        for (int c = P - 1; c >= 0; c--) {
            double v = tu[P + c];
#pragma unroll
            for (int k = c + 1; k < P; k++) v = __builtin_fma(-rr[c][k], b[k], v);
            b[c] = v / rr[c][c];
        }
        if (lane == 0 && active) {
            kp.baseMean[g] = mean;
            kp.baseVar[g] = av / (double)(m - 1);
            kp.allZero[g] = (a2[1] == 0.0) ? 1 : 0;
            kp.roughDisp[g] = __builtin_fmax(ae / (double)(m - P), 0.0);
#pragma unroll
            for (int c = 0; c < P; c++) kp.beta_init[(size_t)g + (size_t)kp.n * c] = b[c];
        }
    }
}

// linearModelMuNormalized (R/core.R:2454-2471): mu = nf * ((yn Q)(X R^-1)'), optionally floored (:763)
template <int P, bool TILED>
__global__ void __launch_bounds__(256) linear_mu_kernel(PrefitKernelParams kp, double mu_floor, double *mu_out) {
    extern __shared__ __attribute__((aligned(16))) double tile[];       // TILED: P x kTileS doubles (see prefit_kernel)
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int waves = blockDim.x >> 6;
    const int m = kp.m;
    const int nwork = DSQ_NWORK(kp);
    for (int wbase = blockIdx.x * waves; wbase < nwork; wbase += gridDim.x * waves) {
        const int wi = wbase + wave;
        const bool active = wi < nwork;
        if (!TILED && !active) break;
        const int g = DSQ_GENE(kp, active ? wi : nwork - 1);
        const int32_t *yg = kp.y + (size_t)g * kp.ld;
        const double *nfg = kp.nf_is_vector ? kp.nf : kp.nf + (size_t)g * kp.ld;
        double tu[P];
#pragma unroll
        for (int c = 0; c < P; c++) tu[c] = 0.0;
        if constexpr (TILED) {
            auto stage = [&](const double *src, int t0) {
                __syncthreads();
                const int ts = (m - t0) < kTileS ? (m - t0) : kTileS;
                for (int idx = threadIdx.x; idx < P * kTileS; idx += blockDim.x) {
                    const int c = idx / kTileS, jj = idx - c * kTileS;
                    if (jj < ts) tile[idx] = src[(size_t)c * m + t0 + jj];
                }
                __syncthreads();
            };
            int32_t yb[kAuxBatch];
            double nb[kAuxBatch];
            for (int t0 = 0; t0 < m; t0 += kTileS) {
                stage(kp.q, t0);
                _Pragma("unroll")
                for (int b = 0; b < kAuxBatch; b++) { const int j = t0 + 64 * b + lane; if (j < m) { yb[b] = yg[j]; nb[b] = nfg[j]; } }
                _Pragma("unroll")
                for (int b = 0; b < kAuxBatch; b++) {
                    const int j = t0 + 64 * b + lane;
                    if (j < m) {
                        double yn = (double)yb[b] / nb[b];
#pragma unroll
                        for (int c = 0; c < P; c++) tu[c] += yn * tile[c * kTileS + 64 * b + lane];
                    }
                }
            }
            wave_allreduce_many(tu, lane);
            double *mg = mu_out + (size_t)g * kp.ld;
            for (int t0 = 0; t0 < m; t0 += kTileS) {
                stage(kp.a, t0);
                _Pragma("unroll")
                for (int b = 0; b < kAuxBatch; b++) {
                    const int j = t0 + 64 * b + lane;
                    if (j < m && active) {
                        double v = tu[0] * tile[64 * b + lane];
#pragma unroll
                        for (int c = 1; c < P; c++) v = __builtin_fma(tu[c], tile[c * kTileS + 64 * b + lane], v);
                        v = v * nfg[j];
                        if (mu_floor > 0.0) v = __builtin_fmax(v, mu_floor);
                        mg[j] = v;
                    }
                }
            }
            continue;
        }
        {
            int32_t yb[kAuxBatch];
            double nb[kAuxBatch];
            sweep_batched<kAuxBatch>(m, lane, [&](int j, int b) { yb[b] = yg[j]; nb[b] = nfg[j]; }, [&](int j, int b) {
                double yn = (double)yb[b] / nb[b];
#pragma unroll
                for (int c = 0; c < P; c++) tu[c] += yn * kp.q[(size_t)c * m + j];
            });
        }
        wave_allreduce_many(tu, lane);
        double *mg = mu_out + (size_t)g * kp.ld;
        for (int j = lane; j < m; j += 64) {
            double v = tu[0] * kp.a[j];
#pragma unroll
            for (int c = 1; c < P; c++) v = __builtin_fma(tu[c], kp.a[(size_t)c * m + j], v);
            v = v * nfg[j];
            if (mu_floor > 0.0) v = __builtin_fmax(v, mu_floor);
            mg[j] = v;
        }
    }
}

// nbinomLogLike (R/core.R:2208-2217) on the closed split of the density (dsq_math.hpp): K' -- the mu-independent part of the
// row's sum, samples in their natural order -- plus one sweep with two logarithms per sample.  The fitBeta launch that
// fitted the row hands K' over (kconst; same counts, dispersions, weights: its constants pass computes it beside its own);
// without it the constants pass runs here.  Round 4: R's dnbinom_mu sample by sample cost 8.2 k instructions per gene for
// this one pass.  Against 40-digit arithmetic the split's terms lose four to five digits to cancellation where bd0 loses
// none (|error| ~ 4e-12 on a log likelihood of -430 against 5e-13); at the dispersion floor (size 1e8) dnbinom_mu's own
// log1p(-x/n) loses nine and the split none -- there the two differ by R's error, ~ 6e-10 per sample, a term of (y, size)
// that cancels in an LRT statistic (tests/test_loglike_cpu.py).
template <bool USE_W>
DSQ_DEV double loglike_constants(const int32_t *yg, const double *wg, int m, int lane, double alpha, double size, bool fast) {
    if (!fast) return 0.0;
    const double st_size = dstirlerr(size);
    double pacc = 0.0;
    for (int j = lane; j < m; j += 64) {
        const double y = (double)yg[j];
        double pj = 0.0;
        if (y != 0.0 && cell_dev_closed(y, size, fast)) {
            double base, t;
            nb_split_const(y, alpha, size, st_size, base, t);
            pj = base + t;
        }
        if constexpr (USE_W) pacc += wg[j] * pj;
        else pacc += pj;
    }
    return wave_allreduce(pacc);
}

template <bool USE_W>
__global__ void __launch_bounds__(256) loglike_kernel(LogLikeKernelParams kp) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int waves = blockDim.x >> 6;
    const int m = kp.m;
    const int nwork = DSQ_NWORK(kp);
    for (int wi = blockIdx.x * waves + wave; wi < nwork; wi += gridDim.x * waves) {
        const int g = DSQ_GENE(kp, wi);
        if (kp.skip && kp.skip[g]) continue;       // (rows another launch is fitting and writing at this moment: pipeline.hip)
        const int32_t *yg = kp.y + (size_t)g * kp.ld;
        const double *mug = kp.mu + (size_t)g * kp.ld;
        const double *wg = USE_W ? kp.weights + (size_t)g * kp.ld : nullptr;
        const double alpha = kp.disp[g];
        const double size = 1.0 / alpha;
        const bool fast = nb_split_fast(alpha, size);
        const double Kp = kp.kconst ? kp.kconst[g] : loglike_constants<USE_W>(yg, wg, m, lane, alpha, size, fast);
        double acc = 0.0;
        for (int j = lane; j < m; j += 64) {
            double d = nb_split_var((double)yg[j], size, alpha, mug[j], fast);
            if constexpr (USE_W) d = wg[j] * d;
            acc += d;
        }
        acc = wave_allreduce(acc);
        if (lane == 0) kp.loglike[g] = Kp + acc;
    }
}

// closed form of the intercept-only model (R/fitNbinomGLMs.R:99-137): b = log(mean of normalized counts)
// [weighted: sum(w K/s) / sum(w)], mu = nf exp(b), w = [weights] / (1/mu + alpha), betaSE = log2(e) sqrt(1/sum w),
// hat = w / sum w.  mu_out is floored at mu_floor when > 0 (R/core.R:763); betaSE / hat use the unfloored mu.
template <bool USE_W>
__global__ void __launch_bounds__(256) intercept_fit_kernel(InterceptKernelParams kp) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int waves = blockDim.x >> 6;
    const int m = kp.m;
    const double log2e = 1.4426950408889634;
    const int nwork = DSQ_NWORK(kp);
    for (int wi = blockIdx.x * waves + wave; wi < nwork; wi += gridDim.x * waves) {
        const int g = DSQ_GENE(kp, wi);
        const int32_t *yg = kp.y + (size_t)g * kp.ld;
        const double *nfg = kp.nf_is_vector ? kp.nf : kp.nf + (size_t)g * kp.ld;
        const double *wg = USE_W ? kp.weights + (size_t)g * kp.ld : nullptr;
        const double alpha = kp.alpha[g];
        double s[2] = {0.0, 0.0};
        for (int j = lane; j < m; j += 64) {
            double cn = (double)yg[j] / nfg[j];
            if constexpr (USE_W) { double w = wg[j]; cn = w * cn; s[1] += w; }
            s[0] += cn;
        }
        wave_allreduce_many(s, lane);
        const double mean = s[0] / (USE_W ? s[1] : (double)m);
        const double b = dlog(mean);
        const double eb = dexp(b);
        double sw = 0.0;
        for (int j = lane; j < m; j += 64) {
            double mu = nfg[j] * eb;
            double wd = 1.0 / (1.0 / mu + alpha);
            if constexpr (USE_W) wd = wg[j] * wd;
            sw += wd;
        }
        const double xtwx = wave_allreduce(sw);
        if (kp.loglike) {               // nbinomLogLike at mu = nf exp(b): the same sums as loglike_kernel on that mu
            const double size = 1.0 / alpha;
            const bool fast = nb_split_fast(alpha, size);
            const double Kp = kp.kconst ? kp.kconst[g] : loglike_constants<USE_W>(yg, wg, m, lane, alpha, size, fast);
            double acc = 0.0;
            for (int j = lane; j < m; j += 64) {
                double d = nb_split_var((double)yg[j], size, alpha, nfg[j] * eb, fast);
                if constexpr (USE_W) d = wg[j] * d;
                acc += d;
            }
            acc = wave_allreduce(acc);
            if (lane == 0) kp.loglike[g] = Kp + acc;
        }
        if (kp.hat || kp.mu_out) {
            for (int j = lane; j < m; j += 64) {
                double mu = nfg[j] * eb;
                if (kp.hat) {
                    double wd = 1.0 / (1.0 / mu + alpha);
                    if constexpr (USE_W) wd = wg[j] * wd;
                    kp.hat[(size_t)g * kp.ld + j] = wd / xtwx;
                }
                if (kp.mu_out) kp.mu_out[(size_t)g * kp.ld + j] = (kp.mu_floor > 0.0) ? __builtin_fmax(mu, kp.mu_floor) : mu;
            }
        }
        if (lane == 0) {
            kp.beta_log2[g] = log2e * b;
            kp.betaSE[g] = log2e * __builtin_sqrt(1.0 / xtwx);
        }
    }
}

// ---- parametricDispersionFit (R/core.R:2166-2190) -----------------------------------------
// The all-gene step between the two dispersion passes: a Gamma-GLM (identity link) IRLS for
// disp ~ a0 + a1/mean inside the reference's outlier-filter loop.  It touches only two n-vectors,
// but every rank of a multi-GPU run fits it over ALL gathered genes (DESeqParallel, R/parallel.R:27),
// so it must not grow with the node: kTrendBlocks workgroups of 16 wavefronts keep the whole nested
// loop on the device (no host round trip per iteration) and meet at a grid barrier per reduction.
// Sums in BLOCK ORDER (the oracle's bsum): partial q = i mod 16384 -> (block, wave, lane); wave
// butterfly; the 16 wave sums of a block added in order; the 16 block sums added in order.  Every
// block reads the same block sums in the same order, so all blocks take identical branches.
static constexpr int kTrendBlocks = 16;

struct TrendWs {
    unsigned int count, gen;
    unsigned int pad[14];
    unsigned long long sums[2][kTrendBlocks][8];   // bit patterns of doubles, double-buffered by parity
};

__device__ __forceinline__ void grid_barrier(TrendWs *ws) {
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned g = __hip_atomic_load(&ws->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        unsigned arrived = atomicAdd(&ws->count, 1u);
        if (arrived == (unsigned)kTrendBlocks - 1u) {
            atomicExch(&ws->count, 0u);
            __threadfence();
            atomicAdd(&ws->gen, 1u);
        } else {
            while (__hip_atomic_load(&ws->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == g) __builtin_amdgcn_s_sleep(2);
        }
        __threadfence();
    }
    __syncthreads();
}

template <int K>
__device__ __forceinline__ void grid_sum(double (&v)[K], double (*red)[8], TrendWs *ws, int &parity) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    wave_allreduce_many(v, lane);         // (the bits of K butterflies, dsq_wave.hpp)
    __syncthreads();                      // previous use of `red` is complete
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < K; k++) red[wave][k] = v[k];
    }
    __syncthreads();
    if (threadIdx.x < K) {
        double tot = red[0][threadIdx.x];
        for (int g = 1; g < 16; g++) tot = tot + red[g][threadIdx.x];
        __hip_atomic_store(&ws->sums[parity][blockIdx.x][threadIdx.x], (unsigned long long)__double_as_longlong(tot),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    grid_barrier(ws);
    // the block sums of every workgroup, added in workgroup order: thread k takes sum k -- its 16 loads in flight together
    // (round 4: every thread used to fetch all K x 16 values itself, two at a time: ~ 60 dependent L2 round trips per pass)
    // -- and hands the total to the block through LDS
    if (threadIdx.x < K) {
        double t[kTrendBlocks];
#pragma unroll
        for (int b = 0; b < kTrendBlocks; b++)
            t[b] = __longlong_as_double((long long)__hip_atomic_load(&ws->sums[parity][b][threadIdx.x], __ATOMIC_RELAXED,
                                                                    __HIP_MEMORY_SCOPE_AGENT));
        double tot = t[0];
#pragma unroll
        for (int b = 1; b < kTrendBlocks; b++) tot = tot + t[b];
        red[0][threadIdx.x] = tot;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; k++) v[k] = red[0][k];
    parity ^= 1;
}

__global__ void __launch_bounds__(1024) trend_fit_kernel(const double *means, const double *disps, long n,
                                                         const int32_t *n_dev, double *coefs_out, int32_t *status_out,
                                                         TrendWs *ws) {
    __shared__ double red[16][8];
    if (n_dev) n = (long)*n_dev;            // fused pipeline: the number of genes in the fit lives on the device
    const long first = (long)blockIdx.x * 1024 + threadIdx.x, stride = 1024L * kTrendBlocks;
    int parity = 0;
    double c0 = 0.1, c1 = 1.0;
    int iter = 0, status = 0;
    for (;;) {
        double b0 = c0, b1 = c1;
        bool converged = false, invalid = false;
        double devold = 0.0;
        // One sweep over the genes and ONE grid reduction per IRLS pass: the deviance sums at the current (b0, b1) and
        // the normal-equation sums the NEXT pass solves (they are taken at the same (b0, b1)) are accumulated together.
        // Per sum the same terms in the same order as two separate sweeps give, so the bits are the separate sweeps';
        // the normal-equation sums of a pass that turns out converged / invalid are simply not used.
        double a[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
        for (int pass = -1; pass < 25 && !invalid; pass++) {
            if (pass >= 0) {
                double det = a[0] * a[2] - a[1] * a[1];
                b0 = (a[2] * a[3] - a[1] * a[4]) / det;
                b1 = (a[0] * a[4] - a[1] * a[3]) / det;
            }
            double v[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};   // a[0..5) for the next pass | sum log r, sum (r - 1), # invalid means
#pragma unroll 1
            for (long i = first; i < n; i += stride) {
                double mean = means[i], y = disps[i];
                double res = y / (c0 + c1 / mean);
                if (!((res > 1e-4) && (res < 15.0))) continue;
                double x = 1.0 / mean;
                double mu = b0 + b1 * x;
                double wgt = 1.0 / (mu * mu);
                double wx = wgt * x;
                v[0] += wgt; v[1] += wx; v[2] += wx * x; v[3] += wgt * y; v[4] += wx * y;
                if (!(mu > 0.0)) { v[7] += 1.0; continue; }
                double r = y / mu;
                v[5] += dlog(r); v[6] += r - 1.0;
            }
            grid_sum<8>(v, red, ws, parity);
            for (int k = 0; k < 5; k++) a[k] = v[k];
            if (v[7] > 0.0) { invalid = true; break; }
            double dev = -2.0 * (v[5] - v[6]);
            if (pass >= 0 && __builtin_fabs(dev - devold) / (__builtin_fabs(dev) + 0.1) < 1e-8) { converged = true; break; }
            devold = dev;
        }
        if (invalid) { status = 1; break; }
        double o0 = c0, o1 = c1;
        c0 = b0; c1 = b1;
        if (!(c0 > 0.0 && c1 > 0.0)) { status = 1; break; }
        double l0 = dlog(c0 / o0), l1 = dlog(c1 / o1);
        if ((l0 * l0 + l1 * l1 < 1e-6) && converged) break;
        iter++;
        if (iter > 10) { status = 2; break; }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { coefs_out[0] = c0; coefs_out[1] = c1; *status_out = status; }
}

size_t trend_fit_workspace_bytes() { return sizeof(TrendWs); }

// ---- getAndCheckWeights (R/core.R:2697-2751) on resident weights -----------------------------------------------------
// One wavefront per gene: w / max(w) (:2702), the same floored at 1e-6 for the gene-wise dispersion search (:702), the
// all(weights >= 0) flag, and the two per-gene rank tests of :2711-2722 on the p x p Gram matrices of w_norm * X and of
// the rows with w_norm > threshold (minus the columns those rows leave all zero) -- rank as qr() counts it: LINPACK
// dqrdc2's column-relative test (tolerance 1e-7 on the norm a column keeps after the accepted columns are projected
// out, relative to its own norm; an exactly zero column never counts), evaluated as a Cholesky factorisation of the
// Gram matrix that skips rejected columns.  A decision, not a value: its sums need no order contract.
struct WeightsPrepParams {
    int n, m, p;
    long ld;
    const double *w_raw, *x;        // n x ld gene-major; m x p column-major
    double thr;
    double *w_norm, *w_floor;       // n x ld
    int32_t *force_zero;            // n: 1 = the weights leave a degenerate design (weightsFail)
    int32_t *neg;                   // 1 int: some weight is negative
};

DSQ_DEV int gram_rank(const double (&G)[DSQ_P_REG][DSQ_P_REG], int p) {
    double Cm[DSQ_P_REG][DSQ_P_REG];
    bool acc[DSQ_P_REG];
    int rank = 0;
    for (int j = 0; j < p; j++) {
        double r = G[j][j];
        for (int k = 0; k < j; k++) if (acc[k]) r -= Cm[k][j] * Cm[k][j];
        const bool ok = (G[j][j] > 0.0) && (r >= 1e-14 * G[j][j]);
        acc[j] = ok;
        if (ok) {
            const double inv = 1.0 / __builtin_sqrt(r);
            for (int c = 0; c < p; c++) {
                double num = G[j][c];
                for (int k = 0; k < j; k++) if (acc[k]) num -= Cm[k][j] * Cm[k][c];
                Cm[j][c] = num * inv;
            }
            rank++;
        }
    }
    return rank;
}

__global__ void __launch_bounds__(256) weights_prep_kernel(WeightsPrepParams kp) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
    const int m = kp.m, p = kp.p;
    for (int g = blockIdx.x * waves + wave; g < kp.n; g += gridDim.x * waves) {
        const double *w = kp.w_raw + (size_t)g * kp.ld;
        double mx = -__builtin_inf();
        int isneg = 0, isnan_ = 0;
        for (int j = lane; j < m; j += 64) {
            const double v = w[j];
            if (v < 0.0) isneg = 1;
            if (v != v) isnan_ = 1;
            if (v > mx) mx = v;
        }
        for (int o = 32; o > 0; o >>= 1) { const double t = __shfl_xor(mx, o, 64); mx = t > mx ? t : mx; }
        if (__any(isnan_)) mx = dnan();                                  // apply(weights, 1, max) is NA then
        if (__any(isneg) && lane == 0) atomicOr(kp.neg, 1);
        double G1[DSQ_P_REG][DSQ_P_REG], G2[DSQ_P_REG][DSQ_P_REG], cs[DSQ_P_REG];
        for (int a = 0; a < p; a++) { cs[a] = 0.0; for (int b = a; b < p; b++) { G1[a][b] = 0.0; G2[a][b] = 0.0; } }
        for (int j = lane; j < m; j += 64) {
            const double wn = w[j] / mx;
            kp.w_norm[(size_t)g * kp.ld + j] = wn;
            kp.w_floor[(size_t)g * kp.ld + j] = (wn != wn) ? wn : (wn > 1e-6 ? wn : 1e-6);      // pmax(weights, 1e-6)
            const double w2 = wn * wn, keep = (wn > kp.thr) ? 1.0 : 0.0;
            for (int a = 0; a < p; a++) {
                const double xa = kp.x[(size_t)a * m + j];
                cs[a] += keep * __builtin_fabs(xa);
                for (int b = a; b < p; b++) {
                    const double xx = xa * kp.x[(size_t)b * m + j];
                    G1[a][b] += w2 * xx;
                    G2[a][b] += keep * xx;
                }
            }
        }
        for (int a = 0; a < p; a++) {
            cs[a] = wave_allreduce(cs[a]);
            for (int b = a; b < p; b++) {
                G1[a][b] = wave_allreduce(G1[a][b]); G1[b][a] = G1[a][b];
                G2[a][b] = wave_allreduce(G2[a][b]); G2[b][a] = G2[a][b];
            }
        }
        if (lane == 0) {
            int ncol = 0;
            for (int a = 0; a < p; a++) ncol += cs[a] > 0.0 ? 1 : 0;
            const bool ok = (gram_rank(G1, p) == p) && (gram_rank(G2, p) == ncol);
            kp.force_zero[g] = ok ? 0 : 1;
        }
    }
}

// ... for designs of more than DSQ_P_REG columns (round 5): one wavefront per gene and workgroup, the design padded to
// PK = 16 / 24 / 32 / 48 columns.  Lane l owns the entries e = l, l + 64, ... of the two PK x PK Gram matrices (registers:
// PK^2 / 64 per matrix); the samples go by in tiles of kRankTile whose design rows, w^2 and keep flags sit in LDS.  The rank
// test is gram_rank's, one matrix column per lane: the same subtractions in the same order for every entry, the Gram
// matrices and the factor in LDS.
static constexpr int kRankTile = 16;
template <int PK>
DSQ_DEV int gram_rank_lds(const double *G, double *Cm, int p, int lane) {
    unsigned long long acc = 0ull;
    int rank = 0;
    for (int j = 0; j < p; j++) {
        const double gjj = G[j * PK + j];
        double r = gjj;
        for (int k = 0; k < j; k++)
            if ((acc >> k) & 1ull) { const double c = Cm[k * PK + j]; r -= c * c; }
        const bool ok = (gjj > 0.0) && (r >= 1e-14 * gjj);
        if (ok) {
            const double inv = 1.0 / __builtin_sqrt(r);
            if (lane < p) {
                double num = G[j * PK + lane];
                for (int k = 0; k < j; k++)
                    if ((acc >> k) & 1ull) num -= Cm[k * PK + j] * Cm[k * PK + lane];
                Cm[j * PK + lane] = num * inv;
            }
            acc |= 1ull << j;
            rank++;
            wave_lds_sync();
        }
    }
    return rank;
}

template <int PK>
__global__ void __launch_bounds__(64) weights_prep_wide_kernel(WeightsPrepParams kp) {
    extern __shared__ double smem[];
    constexpr int T = PK * PK / 64;
    static_assert(T * 64 == PK * PK, "the padded Gram matrix splits evenly over the lanes");
    double *G1 = smem, *G2 = G1 + PK * PK, *Cm = G2 + PK * PK, *xt = Cm + PK * PK, *w2t = xt + kRankTile * PK, *kpt = w2t + kRankTile;
    const int lane = threadIdx.x;
    const int m = kp.m, p = kp.p;
    for (int g = blockIdx.x; g < kp.n; g += gridDim.x) {
        const double *w = kp.w_raw + (size_t)g * kp.ld;
        double mx = -__builtin_inf();
        int isneg = 0, isnan_ = 0;
        for (int j = lane; j < m; j += 64) {
            const double v = w[j];
            if (v < 0.0) isneg = 1;
            if (v != v) isnan_ = 1;
            if (v > mx) mx = v;
        }
        for (int o = 32; o > 0; o >>= 1) { const double t = __shfl_xor(mx, o, 64); mx = t > mx ? t : mx; }
        if (__any(isnan_)) mx = dnan();                                  // apply(weights, 1, max) is NA then
        if (__any(isneg) && lane == 0) atomicOr(kp.neg, 1);
        for (int j = lane; j < m; j += 64) {
            const double wn = w[j] / mx;
            kp.w_norm[(size_t)g * kp.ld + j] = wn;
            kp.w_floor[(size_t)g * kp.ld + j] = (wn != wn) ? wn : (wn > 1e-6 ? wn : 1e-6);      // pmax(weights, 1e-6)
        }
        double g1[T], g2[T], cs = 0.0;
#pragma unroll
        for (int t = 0; t < T; t++) { g1[t] = 0.0; g2[t] = 0.0; }
        for (int j0 = 0; j0 < m; j0 += kRankTile) {
            wave_lds_sync();
            const int nj = (m - j0) < kRankTile ? (m - j0) : kRankTile;
            for (int i = lane; i < kRankTile * PK; i += 64) {
                const int c = i / kRankTile, jj = i - c * kRankTile;
                xt[jj * PK + c] = (c < p && jj < nj) ? kp.x[(size_t)c * m + j0 + jj] : 0.0;
            }
            if (lane < kRankTile) {
                const double wn = lane < nj ? w[j0 + lane] / mx : 0.0;
                w2t[lane] = wn * wn;
                kpt[lane] = (wn > kp.thr) ? 1.0 : 0.0;
            }
            wave_lds_sync();
            for (int jj = 0; jj < nj; jj++) {
                const double w2 = w2t[jj], keep = kpt[jj];
                const double *xr = xt + jj * PK;
                if (lane < PK) cs += keep * __builtin_fabs(xr[lane]);
#pragma unroll
                for (int t = 0; t < T; t++) {
                    const int e = lane + 64 * t;
                    const double xx = xr[e / PK] * xr[e % PK];
                    g1[t] += w2 * xx;
                    g2[t] += keep * xx;
                }
            }
        }
        wave_lds_sync();
#pragma unroll
        for (int t = 0; t < T; t++) { G1[lane + 64 * t] = g1[t]; G2[lane + 64 * t] = g2[t]; }
        wave_lds_sync();
        const int ncol = __popcll(__ballot(lane < p && cs > 0.0));
        const int r1 = gram_rank_lds<PK>(G1, Cm, p, lane);
        wave_lds_sync();
        const int r2 = gram_rank_lds<PK>(G2, Cm, p, lane);
        if (lane == 0) kp.force_zero[g] = (r1 == p && r2 == ncol) ? 0 : 1;
    }
}

hipError_t launch_weights_prep(const double *w_raw, const double *x, int n, int m, int p, long ld, double thr, double *w_norm,
                               double *w_floor, int32_t *force_zero, int32_t *neg, hipStream_t st) {
    WeightsPrepParams kp = {n, m, p, ld, w_raw, x, thr, w_norm, w_floor, force_zero, neg};
    if (p > DSQ_P_REG) {
        const int grid = n < 256 * 8 ? n : 256 * 8;
        switch (dsq_wide_width(p)) {
#define DSQ_X(W)                                                                                                              \
        case W: {                                                                                                             \
            const size_t lds = (size_t)(3 * W * W + kRankTile * W + 2 * kRankTile) * sizeof(double);                           \
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&weights_prep_wide_kernel<W>),                                \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                         \
            hipLaunchKernelGGL((weights_prep_wide_kernel<W>), dim3(grid), dim3(64), lds, st, kp);                              \
            break;                                                                                                            \
        }
        DSQ_WIDE_LIST(DSQ_X)
#undef DSQ_X
        default: return hipErrorInvalidValue;
        }
        return hipGetLastError();
    }
    hipLaunchKernelGGL(weights_prep_kernel, dim3(aux_grid_fwd(n)), dim3(256), 0, st, kp);
    return hipGetLastError();
}

// momentsDispEstimate's xim for a normalization-factor MATRIX (R/core.R:2440-2444): mean over the samples of
// 1 / colMeans(nf).  One thread per sample sums its column down the genes in gene order (the order contract: a plain
// sequential sum, like R's colMeans), one thread adds the m reciprocals in sample order.
__global__ void __launch_bounds__(256) xim_kernel(const double *nf, int n, int m, long ld, double *colmean_recip) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    double s = 0.0;
    for (int g = 0; g < n; g++) s += nf[(size_t)g * ld + j];
    colmean_recip[j] = 1.0 / (s / (double)n);
}
__global__ void xim_final_kernel(const double *colmean_recip, int m, double *out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double s = 0.0;
    for (int j = 0; j < m; j++) s += colmean_recip[j];
    *out = s / (double)m;
}
// the same over the genes rows[0 .. *n_dev) (ascending): estimateDispersionsGeneEst works on the rows that are not all
// zero (objectNZ, R/core.R:693-700), so the column means are taken over those
__global__ void __launch_bounds__(256) xim_rows_kernel(const double *nf, const int32_t *rows, const int32_t *n_dev, int m, long ld,
                                                       double *colmean_recip) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    const int cnt = *n_dev;
    double s = 0.0;
    for (int i = 0; i < cnt; i++) s += nf[(size_t)rows[i] * ld + j];
    colmean_recip[j] = 1.0 / (s / (double)cnt);
}
hipError_t launch_xim_rows(const double *nf, const int32_t *rows, const int32_t *n_dev, int m, long ld, double *scratch_m,
                           double *out, hipStream_t st) {
    hipLaunchKernelGGL(xim_rows_kernel, dim3((m + 255) / 256), dim3(256), 0, st, nf, rows, n_dev, m, ld, scratch_m);
    hipLaunchKernelGGL(xim_final_kernel, dim3(1), dim3(64), 0, st, (const double *)scratch_m, m, out);
    return hipGetLastError();
}
// ... and over the genes with want_a[g] != 0 and want_b[g] == 0, in GENE order whatever order a row list of them would
// have (the refitted rows of refitWithoutOutliers: replace & !allZero -- their list is built with atomics, and the order
// of these sums is part of the result)
__global__ void __launch_bounds__(256) xim_flagged_kernel(const double *nf, int n, int m, long ld, const int32_t *want_a,
                                                          const int32_t *want_b, double *colmean_recip) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    double s = 0.0;
    int cnt = 0;
    for (int g = 0; g < n; g++)
        if (want_a[g] != 0 && want_b[g] == 0) { s += nf[(size_t)g * ld + j]; cnt++; }
    colmean_recip[j] = 1.0 / (s / (double)cnt);
}
hipError_t launch_xim_flagged(const double *nf, int n, int m, long ld, const int32_t *want_a, const int32_t *want_b, double *scratch_m,
                              double *out, hipStream_t st) {
    hipLaunchKernelGGL(xim_flagged_kernel, dim3((m + 255) / 256), dim3(256), 0, st, nf, n, m, ld, want_a, want_b, scratch_m);
    hipLaunchKernelGGL(xim_final_kernel, dim3(1), dim3(64), 0, st, (const double *)scratch_m, m, out);
    return hipGetLastError();
}
hipError_t launch_xim(const double *nf, int n, int m, long ld, double *scratch_m, double *out, hipStream_t st) {
    hipLaunchKernelGGL(xim_kernel, dim3((m + 255) / 256), dim3(256), 0, st, nf, n, m, ld, scratch_m);
    hipLaunchKernelGGL(xim_final_kernel, dim3(1), dim3(64), 0, st, (const double *)scratch_m, m, out);
    return hipGetLastError();
}

hipError_t launch_trend_fit(const double *means, const double *disps, long n, double *coefs, int32_t *status,
                            void *workspace, hipStream_t st) {
    hipError_t e = hipMemsetAsync(workspace, 0, sizeof(TrendWs), st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(trend_fit_kernel, dim3(kTrendBlocks), dim3(1024), 0, st, means, disps, n, (const int32_t *)nullptr,
                       coefs, status, (TrendWs *)workspace);
    return hipGetLastError();
}

hipError_t launch_trend_fit_dev(const double *means, const double *disps, const int32_t *n_dev, double *coefs,
                                int32_t *status, void *workspace, hipStream_t st) {
    hipError_t e = hipMemsetAsync(workspace, 0, sizeof(TrendWs), st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(trend_fit_kernel, dim3(kTrendBlocks), dim3(1024), 0, st, means, disps, 0L, n_dev, coefs, status,
                       (TrendWs *)workspace);
    return hipGetLastError();
}

// ... with the workspace already zeroed by the caller (the chain's one init launch, pipeline.hip)
hipError_t launch_trend_fit_dev_zeroed(const double *means, const double *disps, const int32_t *n_dev, double *coefs,
                                       int32_t *status, void *workspace, hipStream_t st) {
    hipLaunchKernelGGL(trend_fit_kernel, dim3(kTrendBlocks), dim3(1024), 0, st, means, disps, 0L, n_dev, coefs, status,
                       (TrendWs *)workspace);
    return hipGetLastError();
}

static inline int aux_grid(int n) {
    int blocks = (n + 3) / 4;
    int cap = device_cu_count() * 8;
    return blocks < cap ? (blocks < 1 ? 1 : blocks) : cap;
}

template <int P>
static hipError_t launch_prefit_p(const PrefitKernelParams &kp, hipStream_t st) {
    if constexpr (P <= 24) {
        if (aux_tiled(kp.m, P)) {
            const size_t lds = (size_t)P * kTileS * sizeof(double);
            if (kp.useWeights) hipLaunchKernelGGL((prefit_kernel<P, true, true>), dim3(aux_grid(kp.n)), dim3(256), lds, st, kp);
            else hipLaunchKernelGGL((prefit_kernel<P, false, true>), dim3(aux_grid(kp.n)), dim3(256), lds, st, kp);
            return hipGetLastError();
        }
    }
    if (kp.useWeights) hipLaunchKernelGGL((prefit_kernel<P, true, false>), dim3(aux_grid(kp.n)), dim3(256), 0, st, kp);
    else hipLaunchKernelGGL((prefit_kernel<P, false, false>), dim3(aux_grid(kp.n)), dim3(256), 0, st, kp);
    return hipGetLastError();
}

hipError_t launch_prefit(const PrefitKernelParams &kp, hipStream_t st, bool *ok) {
    *ok = true;
    switch (kp.p) {
#define DSQ_X(W) case W: return launch_prefit_p<W>(kp, st);
    DSQ_P_EACH(DSQ_X)
#undef DSQ_X
    default: *ok = false; return hipSuccess;
    }
}

template <int P>
static hipError_t launch_linear_mu_p(const PrefitKernelParams &kp, double mu_floor, double *mu, hipStream_t st) {
    if constexpr (P <= 24) {
        if (aux_tiled(kp.m, P)) {
            hipLaunchKernelGGL((linear_mu_kernel<P, true>), dim3(aux_grid(kp.n)), dim3(256), (size_t)P * kTileS * sizeof(double), st, kp, mu_floor, mu);
            return hipGetLastError();
        }
    }
    hipLaunchKernelGGL((linear_mu_kernel<P, false>), dim3(aux_grid(kp.n)), dim3(256), 0, st, kp, mu_floor, mu);
    return hipGetLastError();
}

hipError_t launch_linear_mu(const PrefitKernelParams &kp, double mu_floor, double *mu, hipStream_t st, bool *ok) {
    *ok = true;
    switch (kp.p) {
#define DSQ_X(W) case W: return launch_linear_mu_p<W>(kp, mu_floor, mu, st);
    DSQ_P_EACH(DSQ_X)
#undef DSQ_X
    default: *ok = false; return hipSuccess;
    }
}

hipError_t launch_intercept_fit(const InterceptKernelParams &kp, hipStream_t st) {
    if (kp.useWeights) hipLaunchKernelGGL((intercept_fit_kernel<true>), dim3(aux_grid(kp.n)), dim3(256), 0, st, kp);
    else hipLaunchKernelGGL((intercept_fit_kernel<false>), dim3(aux_grid(kp.n)), dim3(256), 0, st, kp);
    return hipGetLastError();
}

hipError_t launch_loglike(const LogLikeKernelParams &kp, hipStream_t st) {
    if (kp.useWeights) hipLaunchKernelGGL((loglike_kernel<true>), dim3(aux_grid(kp.n)), dim3(256), 0, st, kp);
    else hipLaunchKernelGGL((loglike_kernel<false>), dim3(aux_grid(kp.n)), dim3(256), 0, st, kp);
    return hipGetLastError();
}
// the same launch BESIDE other work (pipeline.hip: side stream): the grid-stride grid above fills every wave slot of the
// device for the whole launch, so nothing of another stream would get in before it ends; three quarters of it leaves two
// blocks per CU to the (small, latency-bound) launches it runs beside
hipError_t launch_loglike_side(const LogLikeKernelParams &kp, hipStream_t st) {
    int grid = aux_grid(kp.n);
    const int cap = device_cu_count() * 6;
    if (grid > cap) grid = cap;
    if (kp.useWeights) hipLaunchKernelGGL((loglike_kernel<true>), dim3(grid), dim3(256), 0, st, kp);
    else hipLaunchKernelGGL((loglike_kernel<false>), dim3(grid), dim3(256), 0, st, kp);
    return hipGetLastError();
}

}  // namespace dsq
