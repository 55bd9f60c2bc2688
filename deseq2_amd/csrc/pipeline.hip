// pipeline.hip -- dsq_deseq_dev: the whole DESeq() chain driven from the device (include/deseq2_mi355x.h).
//
// The reference's R code between the native calls is per-gene arithmetic on n-vectors (clamps, accept /
// convergence / refit rules: R/core.R:727-728, 785, 826-848, 1019-1024, 1048-1063, 1099-1115; betaConv, log2
// rescaling, Wald statistic: R/fitNbinomGLMs.R:185-198, R/core.R:1471,1507) plus two all-gene steps
// (parametricDispersionFit R/core.R:2166-2190, mad of the log residuals R/methods.R:180).  Here each rule is a
// small elementwise kernel, the rows a rule sends on (fitDispGrid stragglers, replaced-outlier rows) are
// compacted on the device and fitted by ROW-LISTED launches of the same fit kernels (DispKernelParams::rows /
// n_dev), and the all-gene steps are single-workgroup kernels, so a phase is one uninterrupted stream of
// launches: no host decision, no device-to-host copy.  Every formula is evaluated with the operations of the
// host mirror (deseq2_amd/core.py: IEEE + - * / sqrt, the engine's dlog / dexp, numpy's NaN-propagating
// minimum / maximum), so the results equal the call-by-call chain bit for bit (tests/test_gpu_fused.py).
#include "../../include/deseq2_mi355x.h"
#include "dsq_internal.hpp"
#include "dsq_math.hpp"
#include "dsq_wave.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace dsq {

hipError_t launch_trend_fit_dev(const double *means, const double *disps, const int32_t *n_dev, double *coefs,
                                int32_t *status, void *workspace, hipStream_t st);
hipError_t launch_trend_fit_dev_zeroed(const double *means, const double *disps, const int32_t *n_dev, double *coefs,
                                       int32_t *status, void *workspace, hipStream_t st);
size_t trend_fit_workspace_bytes();

// numpy.minimum / numpy.maximum: NaN if either operand is NaN
DSQ_DEV double np_min(double a, double b) { return (a != a || b != b) ? a + b : (a < b ? a : b); }
DSQ_DEV double np_max(double a, double b) { return (a != a || b != b) ? a + b : (a > b ? a : b); }

struct Rows {                 // the genes a rule kernel covers: rows[0 .. *n_dev) or 0 .. n-1
    const int32_t *rows;
    const int32_t *n_dev;
    int n;
};
DSQ_DEV int rows_count(const Rows &r) { return r.n_dev ? *r.n_dev : r.n; }
DSQ_DEV int rows_gene(const Rows &r, int i) { return r.rows ? r.rows[i] : i; }

// ---- ordered compaction by one workgroup (the trend fit sums its input in index order) -------------------------
// mode 0: keep = !(allZero | force_zero) -> rows; also folds force_zero into allZero
// mode 1: keep = disp > thresh -> (means, disps) pairs (useForFit, R/core.R:870)
__global__ void __launch_bounds__(1024) compact_kernel(int mode, int n, int32_t *allZero, const int32_t *force_zero,
                                                       const double *mean_in, const double *disp_in, double thresh,
                                                       int32_t *rows_out, double *mean_out, double *disp_out,
                                                       int32_t *count_out) {
    // tiles of 1024 consecutive elements (coalesced), kept elements ranked by wave ballots + a 16-entry prefix per tile;
    // FOUR tiles per round (round 4: one tile per round was 49 rounds of "load, three barriers" at 50 000 genes -- 66 us of
    // latency, twice per analysis; the loads of a round are now in flight together and the barriers are shared)
    constexpr int NT = 4;
    __shared__ int wcnt[NT][16];
    __shared__ int base_s;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (t == 0) base_s = 0;
    __syncthreads();
    for (int i0 = 0; i0 < n; i0 += NT * 1024) {
        bool keep[NT];
        double mv[NT], dv[NT];
#pragma unroll
        for (int q = 0; q < NT; q++) {
            const int i = i0 + q * 1024 + t;
            keep[q] = false; mv[q] = 0.0; dv[q] = 0.0;
            if (i < n) {
                if (mode == 0) {
                    int z = allZero[i] | (force_zero ? force_zero[i] : 0);
                    if (force_zero) allZero[i] = z ? 1 : 0;
                    keep[q] = !z;
                } else {
                    dv[q] = disp_in[i]; mv[q] = mean_in[i];
                    keep[q] = dv[q] > thresh;
                }
            }
        }
        unsigned long long mask[NT];
#pragma unroll
        for (int q = 0; q < NT; q++) {
            mask[q] = __ballot(keep[q]);
            if (lane == 0) wcnt[q][wave] = __popcll(mask[q]);
        }
        __syncthreads();
        int off = base_s;
#pragma unroll
        for (int q = 0; q < NT; q++) {
            int mine = off;
            for (int w = 0; w < 16; w++) { const int c = wcnt[q][w]; if (w < wave) mine += c; off += c; }
            mine += __popcll(mask[q] & ((1ull << lane) - 1ull));
            if (keep[q]) {
                const int i = i0 + q * 1024 + t;
                if (mode == 0) rows_out[mine] = i;
                else { mean_out[mine] = mv[q]; disp_out[mine] = dv[q]; }
            }
        }
        __syncthreads();
        if (t == 0) base_s = off;               // (every thread has computed the same running total)
        __syncthreads();
    }
    if (t == 0) *count_out = base_s;
}

// ---- stats::mad of the log dispersion residuals + the prior variance (R/methods.R:172-181, R/core.R:1135-1208) ----
// Order statistics by radix selection on the order-preserving 64-bit image of the doubles: 8 passes of 8 bits per
// rank, one workgroup.  A selected order statistic is exact, so the medians equal numpy's.
DSQ_DEV uint64_t key_of(double d) {
    uint64_t u = d2bits(d);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ULL);
}
DSQ_DEV double double_of(uint64_t k) {
    uint64_t u = (k >> 63) ? (k & 0x7fffffffffffffffULL) : ~k;
    return bits2d(u);
}

// the rank-th smallest (0-based) of the <= kSelDirect keys in `keys` (LDS), by counting: thread t takes key t and counts the
// keys that sort before it (smaller, or equal with a smaller index); exactly one thread finds `rank` and publishes its key
static constexpr int kSelDirect = 1024;       // = hist[2048] reinterpreted as 64-bit keys
DSQ_DEV uint64_t rank_direct(const uint64_t *keys, int cnt, long rank, unsigned long long *bc) {
    for (int t = threadIdx.x; t < cnt; t += blockDim.x) {
        const uint64_t mine = keys[t];
        int before = 0;
        for (int j = 0; j < cnt; j++) {
            const uint64_t o = keys[j];
            before += (o < mine || (o == mine && j < t)) ? 1 : 0;
        }
        if (before == (int)rank) bc[0] = mine;
    }
    __syncthreads();
    const uint64_t r = bc[0];
    __syncthreads();
    return r;
}

template <class F>
DSQ_DEV double block_select(int n, long rank, F &&value, unsigned *hist, unsigned long long *bc) {
    // the rank-th smallest (0-based) of value(i), i < n; every thread returns it.  Digits of 11, 11, 11, 11, 11, 9 bits.
    // (r6) Once the candidates left (the keys that share the digits chosen so far) fit the histogram's LDS -- after two
    // passes, usually: the first 22 bits of a double leave a handful of 50 000 residuals -- they are gathered there and the
    // order statistic is taken by direct counting: three passes over the values instead of six, the same (exact) result.
    uint64_t prefix = 0, mask = 0;
    int shift = 64;
    while (shift > 0) {
        const int bits = shift >= 11 + 9 ? 11 : shift;         // 64 = 5 x 11 + 9
        shift -= bits;
        const unsigned nb = 1u << bits;
        for (unsigned b = threadIdx.x; b < nb; b += blockDim.x) hist[b] = 0;
        __syncthreads();
        // (round 4 measured two variants of this pass -- the atomics of lanes that hit the same bin merged by ballot, the
        //  leading digits of log residuals being few; eight loads in flight per thread -- at 0.41 and 0.25 ms for the
        //  kernel against 0.24: neither the LDS atomics nor the L2 round trips are what one workgroup spends its time on)
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            uint64_t k = key_of(value(i));
            if ((k & mask) == prefix) atomicAdd(&hist[(unsigned)(k >> shift) & (nb - 1u)], 1u);
        }
        __syncthreads();
        // the digit whose cumulative count passes `rank`: one wave scans the bins, 64 at a time
        if (threadIdx.x < 64) {
            long r = rank;
            int found = -1;
            unsigned inbin = 0;
            for (unsigned b0 = 0; b0 < nb && found < 0; b0 += 64) {
                const unsigned h = hist[b0 + threadIdx.x];
                unsigned incl = h;                              // inclusive prefix over the 64 lanes
                for (int o = 1; o < 64; o <<= 1) {
                    unsigned v = __shfl_up(incl, o, 64);
                    if ((int)threadIdx.x >= o) incl += v;
                }
                const unsigned tot = __shfl(incl, 63, 64);
                if (r < (long)tot) {
                    const unsigned long long m = __ballot((long)incl > r);
                    const int l = __ffsll((long long)m) - 1;
                    const unsigned before = __shfl(incl, l, 64) - __shfl(h, l, 64);
                    found = (int)b0 + l;
                    inbin = __shfl(h, l, 64);
                    r -= (long)before;
                } else {
                    r -= (long)tot;
                }
            }
            if (threadIdx.x == 0) { bc[0] = (unsigned long long)(found < 0 ? (int)nb - 1 : found); bc[1] = (unsigned long long)r; bc[2] = found < 0 ? 0ull : inbin; }
        }
        __syncthreads();
        prefix |= (uint64_t)bc[0] << shift;
        mask |= (uint64_t)(nb - 1u) << shift;
        rank = (long)bc[1];
        const unsigned left = (unsigned)bc[2];
        __syncthreads();
        if (shift > 0 && left > 0 && left <= (unsigned)kSelDirect) {
            uint64_t *keys = reinterpret_cast<uint64_t *>(hist);
            if (threadIdx.x == 0) bc[2] = 0ull;
            __syncthreads();
            for (int i = threadIdx.x; i < n; i += blockDim.x) {
                const uint64_t k = key_of(value(i));
                if ((k & mask) == prefix) keys[atomicAdd(&bc[2], 1ull)] = k;
            }
            __syncthreads();
            return double_of(rank_direct(keys, (int)left, rank, bc));
        }
    }
    return double_of(prefix);
}

template <class F>
DSQ_DEV double block_median(int n, long k, F &&value, unsigned *hist, unsigned long long *bc) {
    // numpy.median of the k finite values (invalid entries are +inf and sort last): the lower middle order statistic
    // by selection; for an even count the next one is either the same value (a tie) or the smallest value above it
    const double a = block_select(n, (k - 1) / 2, value, hist, bc);
    if (k & 1) return a;
    __syncthreads();
    if (threadIdx.x == 0) { bc[0] = 0ull; bc[1] = key_of(__builtin_inf()); }
    __syncthreads();
    unsigned long long le = 0, mn = key_of(__builtin_inf());
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const double v = value(i);
        if (v <= a) le++;
        else { const unsigned long long kv = key_of(v); if (kv < mn) mn = kv; }
    }
    atomicAdd(&bc[0], le);
    atomicMin(&bc[1], mn);
    __syncthreads();
    const double b = ((long)bc[0] > k / 2) ? a : double_of(bc[1]);
    __syncthreads();
    return (a + b) * 0.5;
}

__global__ void __launch_bounds__(1024) prior_var_kernel(const double *mean, const double *disp, int n, double minDisp,
                                                         double expVarLogDisp, int m_gt_p, double *resbuf,
                                                         double *scalars, int32_t *status, const double *fit_in,
                                                         double pv_in) {
    __shared__ __attribute__((aligned(16))) unsigned hist[2048];
    __shared__ unsigned long long bc[3];
    __shared__ int kshared;
    const double inf = __builtin_inf();
    const double c0 = scalars[DSQ_SC_COEF0], c1 = scalars[DSQ_SC_COEF1];
    if (threadIdx.x == 0) kshared = 0;
    __syncthreads();
    int c = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        double d = disp[i];
        bool above = d >= minDisp * 100.0;                       // aboveMinDisp, R/core.R:897 / :1137
        double r = inf;
        if (above) {
            double fit = fit_in ? fit_in[i] : c0 + c1 / mean[i];       // (fit_in: the caller's trend, DSQ_FIT_GIVEN)
            r = dlog(d) - dlog(fit);
            c++;
        }
        resbuf[i] = r;
    }
    atomicAdd(&kshared, c);
    __syncthreads();
    const long k = kshared;
    if (threadIdx.x == 0) status[DSQ_ST_N_ABOVE_MIN] = (int32_t)k;
    if (k == 0) {
        if (threadIdx.x == 0) { scalars[DSQ_SC_VAR_LOG_DISP] = dnan(); scalars[DSQ_SC_DISP_PRIOR_VAR] = dnan(); }
        return;
    }
    __threadfence_block();
    const double med = block_median(n, k, [&](int i) { return resbuf[i]; }, hist, bc);
    const double med2 = block_median(n, k, [&](int i) {
        double r = resbuf[i];
        return (r == inf) ? inf : __builtin_fabs(r - med);
    }, hist, bc);
    if (threadIdx.x == 0) {
        const double mad = 1.4826 * med2;
        const double v = mad * mad;
        scalars[DSQ_SC_VAR_LOG_DISP] = v;
        double pv = v;
        if (m_gt_p) {
            const double t = v - expVarLogDisp;
            pv = (0.25 > t) ? 0.25 : t;                           // max(varLogDispEsts - expVarLogDisp, 0.25), :1200
        }
        if (pv_in > 0.0) pv = pv_in;                              // estimateDispersionsMAP(dispPriorVar = x), :989-994
        scalars[DSQ_SC_DISP_PRIOR_VAR] = pv;
    }
}

// ---- the same on SIXTEEN workgroups (round 4): the one-workgroup kernel above spends 0.24 ms on 50 000 genes -- its thirteen
// selection passes and the two logarithms per gene all on one CU -- and every rank of a gene-sharded run pays it on the
// gathered vectors.  Here each workgroup histograms its slice, the histograms meet in a global table (one per pass, zeroed
// by the launch), a grid barrier, and every workgroup scans the table itself: the selected order statistics are exact, so
// nothing changes in the results.  The workgroups must be co-resident: 16 x 1024 threads on a 256-CU device.
static constexpr int kSelBlocks = 16;
struct SelWs {
    unsigned int count, gen;
    unsigned int pad[14];
    unsigned long long cnt[8];
    unsigned long long inv_min[8];       // ~key of the smallest value above a median candidate (atomicMax; zero = none)
    unsigned int ghist[16][2048];
    // (r6) the early exit of a selection: the candidates left after a pass, gathered by all workgroups (one list and one fill
    // counter per selection: a launch makes two)
    unsigned long long gfill[2];
    unsigned long long glist[2][1024];
};
size_t prior_var_workspace_bytes() { return sizeof(SelWs); }

DSQ_DEV void sel_barrier(SelWs *ws) {
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned g = __hip_atomic_load(&ws->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        unsigned arrived = atomicAdd(&ws->count, 1u);
        if (arrived == (unsigned)kSelBlocks - 1u) {
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

template <class F>
DSQ_DEV double grid_select(int n, long rank, F &&value, unsigned *hist, unsigned long long *bc, SelWs *ws, int &phase, int slot) {
    uint64_t prefix = 0, mask = 0;
    int shift = 64;
    const long first = (long)blockIdx.x * blockDim.x + threadIdx.x, stride = (long)blockDim.x * kSelBlocks;
    while (shift > 0) {
        const int bits = shift >= 11 + 9 ? 11 : shift;         // 64 = 5 x 11 + 9
        shift -= bits;
        const unsigned nb = 1u << bits;
        for (unsigned b = threadIdx.x; b < nb; b += blockDim.x) hist[b] = 0;
        __syncthreads();
        for (long i = first; i < n; i += stride) {
            uint64_t k = key_of(value((int)i));
            if ((k & mask) == prefix) atomicAdd(&hist[(unsigned)(k >> shift) & (nb - 1u)], 1u);
        }
        __syncthreads();
        unsigned *gh = ws->ghist[phase];
        for (unsigned b = threadIdx.x; b < nb; b += blockDim.x) { const unsigned h = hist[b]; if (h) atomicAdd(&gh[b], h); }
        sel_barrier(ws);
        for (unsigned b = threadIdx.x; b < nb; b += blockDim.x) hist[b] = __hip_atomic_load(&gh[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (threadIdx.x < 64) {
            long r = rank;
            int found = -1;
            unsigned inbin = 0;
            for (unsigned b0 = 0; b0 < nb && found < 0; b0 += 64) {
                const unsigned h = hist[b0 + threadIdx.x];
                unsigned incl = h;
                for (int o = 1; o < 64; o <<= 1) {
                    unsigned v = __shfl_up(incl, o, 64);
                    if ((int)threadIdx.x >= o) incl += v;
                }
                const unsigned tot = __shfl(incl, 63, 64);
                if (r < (long)tot) {
                    const unsigned long long m = __ballot((long)incl > r);
                    const int l = __ffsll((long long)m) - 1;
                    const unsigned before = __shfl(incl, l, 64) - __shfl(h, l, 64);
                    found = (int)b0 + l;
                    inbin = __shfl(h, l, 64);
                    r -= (long)before;
                } else {
                    r -= (long)tot;
                }
            }
            if (threadIdx.x == 0) { bc[0] = (unsigned long long)(found < 0 ? (int)nb - 1 : found); bc[1] = (unsigned long long)r; bc[2] = found < 0 ? 0ull : inbin; }
        }
        __syncthreads();
        prefix |= (uint64_t)bc[0] << shift;
        mask |= (uint64_t)(nb - 1u) << shift;
        rank = (long)bc[1];
        const unsigned left = (unsigned)bc[2];
        __syncthreads();
        phase++;
        // (r6) the candidates left fit one LDS list: every workgroup appends its own to the launch's global list, a grid
        // barrier, then each workgroup ranks the whole list itself (see block_select) -- the same decision in every
        // workgroup: `left` comes from the shared histogram
        if (shift > 0 && left > 0 && left <= (unsigned)kSelDirect) {
            for (long i = first; i < n; i += stride) {
                const uint64_t k = key_of(value((int)i));
                if ((k & mask) == prefix) ws->glist[slot][atomicAdd(&ws->gfill[slot], 1ull)] = k;
            }
            sel_barrier(ws);
            uint64_t *keys = reinterpret_cast<uint64_t *>(hist);
            for (unsigned t = threadIdx.x; t < left; t += blockDim.x)
                keys[t] = __hip_atomic_load(&ws->glist[slot][t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            return double_of(rank_direct(keys, (int)left, rank, bc));
        }
    }
    return double_of(prefix);
}

template <class F>
DSQ_DEV double grid_median(int n, long k, F &&value, unsigned *hist, unsigned long long *bc, SelWs *ws, int &phase, int slot) {
    const double a = grid_select(n, (k - 1) / 2, value, hist, bc, ws, phase, slot);
    if (k & 1) return a;
    const long first = (long)blockIdx.x * blockDim.x + threadIdx.x, stride = (long)blockDim.x * kSelBlocks;
    unsigned long long le = 0, inv = 0;
    for (long i = first; i < n; i += stride) {
        const double v = value((int)i);
        if (v <= a) le++;
        else { const unsigned long long kv = ~key_of(v); if (kv > inv) inv = kv; }
    }
    if (le) atomicAdd(&ws->cnt[slot], le);
    if (inv) atomicMax(&ws->inv_min[slot], inv);
    sel_barrier(ws);
    const unsigned long long tot = __hip_atomic_load(&ws->cnt[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long iv = __hip_atomic_load(&ws->inv_min[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const double b = ((long)tot > k / 2) ? a : double_of(~iv);
    return (a + b) * 0.5;
}

__global__ void __launch_bounds__(1024) prior_var_grid_kernel(const double *mean, const double *disp, int n, double minDisp,
                                                              double expVarLogDisp, int m_gt_p, double *resbuf, double *scalars,
                                                              int32_t *status, const double *fit_in, double pv_in, SelWs *ws) {
    __shared__ __attribute__((aligned(16))) unsigned hist[2048];
    __shared__ unsigned long long bc[3];
    const double inf = __builtin_inf();
    const double c0 = scalars[DSQ_SC_COEF0], c1 = scalars[DSQ_SC_COEF1];
    const long first = (long)blockIdx.x * blockDim.x + threadIdx.x, stride = (long)blockDim.x * kSelBlocks;
    unsigned long long c = 0;
    for (long i = first; i < n; i += stride) {
        const double d = disp[i];
        const bool above = d >= minDisp * 100.0;                 // aboveMinDisp, R/core.R:897 / :1137
        double r = inf;
        if (above) {
            const double fit = fit_in ? fit_in[i] : c0 + c1 / mean[i];
            r = dlog(d) - dlog(fit);
            c++;
        }
        resbuf[i] = r;                                           // (read back by this thread only)
    }
    if (c) atomicAdd(&ws->cnt[7], c);
    sel_barrier(ws);
    const long k = (long)__hip_atomic_load(&ws->cnt[7], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool writer = blockIdx.x == 0 && threadIdx.x == 0;
    if (writer) status[DSQ_ST_N_ABOVE_MIN] = (int32_t)k;
    if (k == 0) {
        if (writer) { scalars[DSQ_SC_VAR_LOG_DISP] = dnan(); scalars[DSQ_SC_DISP_PRIOR_VAR] = dnan(); }
        return;
    }
    int phase = 0;
    const double med = grid_median(n, k, [&](int i) { return resbuf[i]; }, hist, bc, ws, phase, 0);
    const double med2 = grid_median(n, k, [&](int i) {
        const double r = resbuf[i];
        return (r == inf) ? inf : __builtin_fabs(r - med);
    }, hist, bc, ws, phase, 1);
    if (writer) {
        const double mad = 1.4826 * med2;
        const double v = mad * mad;
        scalars[DSQ_SC_VAR_LOG_DISP] = v;
        double pv = v;
        if (m_gt_p) {
            const double t = v - expVarLogDisp;
            pv = (0.25 > t) ? 0.25 : t;                           // max(varLogDispEsts - expVarLogDisp, 0.25), :1200
        }
        if (pv_in > 0.0) pv = pv_in;                              // estimateDispersionsMAP(dispPriorVar = x), :989-994
        scalars[DSQ_SC_DISP_PRIOR_VAR] = pv;
    }
}

// ---- fitType = "mean" (R/core.R:894-899): mean(dispGeneEst[dispGeneEst > 10 minDisp], trim = 0.001) ----------------
// R: the values between the floor(N trim)-th order statistics from either end, then a long-double mean with a
// correction pass -- to double precision the correctly rounded mean.  Here: the two order statistics by radix selection
// (exact), the sum of the kept values as a 192-bit integer in units of 2^-128 (exact for every value >= 2^-75, and
// independent of the order of the additions: a deterministic result without a specified order), the quotient by long
// division, rounded once to nearest-even.  The mirror (core.py, Python integers) and the oracle restate exactly this.
struct U192 { uint64_t w[3]; };
DSQ_DEV void u192_add(U192 &a, const U192 &b) {
    uint64_t c = 0;
    for (int k = 0; k < 3; k++) {
        const uint64_t s = a.w[k] + b.w[k];
        const uint64_t c1 = s < a.w[k];
        const uint64_t t = s + c;
        c = c1 | (uint64_t)(t < s);
        a.w[k] = t;
    }
}
DSQ_DEV U192 u192_fixed(double x) {           // floor(x 2^128), 0 < x < 2^62 finite
    const uint64_t u = d2bits(x);
    int E = (int)((u >> 52) & 0x7ff);
    uint64_t M = u & ((1ull << 52) - 1ull);
    if (E) M |= 1ull << 52; else E = 1;
    const int sh = E - 1075 + 128;
    U192 r = {{0, 0, 0}};
    if (sh >= 0) {
        const int w = sh >> 6, b = sh & 63;
        if (w < 3) { r.w[w] = M << b; if (b && w + 1 < 3) r.w[w + 1] = M >> (64 - b); }
    } else if (-sh < 64) r.w[0] = M >> (-sh);
    return r;
}
DSQ_DEV U192 u192_times(double x, unsigned long c) {      // c copies of x (ties at the two cut points)
    U192 r = {{0, 0, 0}}, v = u192_fixed(x);
    for (; c; c >>= 1) { if (c & 1ul) u192_add(r, v); U192 d = v; u192_add(v, d); }
    return r;
}
DSQ_DEV double u192_mean(const U192 &S, uint64_t cnt) {   // RN-even(S / cnt) 2^-128 (cnt < 2^32)
    uint32_t q[6];
    uint64_t rem = 0;
    for (int k = 5; k >= 0; k--) {
        const uint64_t limb = (S.w[k >> 1] >> ((k & 1) * 32)) & 0xffffffffull;
        const uint64_t cur = (rem << 32) | limb;
        q[k] = (uint32_t)(cur / cnt);
        rem = cur % cnt;
    }
    int h = -1;
    for (int k = 5; k >= 0 && h < 0; k--) if (q[k]) h = k * 32 + 31 - __builtin_clz(q[k]);
    if (h < 0) return 0.0;
    auto bit_range = [&](int lo, int len) {                // bits [lo, lo + len) of the quotient, len <= 53
        uint64_t v = 0;
        for (int b = len - 1; b >= 0; b--) { const int i = lo + b; v = (v << 1) | ((q[i >> 5] >> (i & 31)) & 1u); }
        return v;
    };
    if (h <= 52) return (double)bit_range(0, h + 1) * bits2d((uint64_t)(1023 - 128) << 52);       // (means below 2^-75: truncated)
    const int shift = h - 52;
    uint64_t mant = bit_range(shift, 53);
    const bool half = (q[(shift - 1) >> 5] >> ((shift - 1) & 31)) & 1u;
    bool below = rem != 0;
    for (int i = 0; i < shift - 1 && !below; i++) below = (q[i >> 5] >> (i & 31)) & 1u;
    if (half && (below || (mant & 1ull))) mant++;
    return (double)mant * bits2d((uint64_t)(1023 + shift - 128) << 52);
}

// mode DSQ_FIT_MEAN: always; DSQ_FIT_PARAMETRIC_OR_MEAN: only when the parametric trend did not fit.  The trend then is
// the constant: COEF0 = the mean, COEF1 = 0 (dispFit = COEF0 + COEF1 / baseMean is that constant, exactly).
__global__ void __launch_bounds__(1024) trend_mean_kernel(const double *disp, int n, double minDisp, int mode, double *scalars,
                                                          int32_t *status) {
    __shared__ __attribute__((aligned(16))) unsigned hist[2048];
    __shared__ unsigned long long bc[3];
    __shared__ unsigned long long cnts[5];
    __shared__ U192 part[1024];
    if (mode == DSQ_FIT_PARAMETRIC_OR_MEAN && status[DSQ_ST_TREND_STATUS] == 0) return;
    const double inf = __builtin_inf(), thr = 10.0 * minDisp;
    auto val = [&](int i) { const double d = disp[i]; return (d > thr) ? d : inf; };       // (NaN: not kept, as na.rm)
    if (threadIdx.x < 5) cnts[threadIdx.x] = 0ull;
    __syncthreads();
    unsigned long long c = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) c += val(i) != inf;
    atomicAdd(&cnts[0], c);
    __syncthreads();
    const long N = (long)cnts[0];
    if (N == 0) {                                           // (cannot happen behind N_TREND > 0; kept a failure)
        if (threadIdx.x == 0) status[DSQ_ST_TREND_STATUS] = 3;
        return;
    }
    const long k = (long)__builtin_floor((double)N * 0.001);
    const double a = block_select(n, k, val, hist, bc);
    const double b = block_select(n, N - 1 - k, val, hist, bc);
    U192 acc = {{0, 0, 0}};
    unsigned long long la = 0, ca = 0, lb = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const double v = val(i);
        if (v == inf) continue;
        la += v < a; ca += v == a; lb += v < b;
        if (v > a && v < b) { const U192 f = u192_fixed(v); u192_add(acc, f); }
    }
    atomicAdd(&cnts[1], la); atomicAdd(&cnts[2], ca); atomicAdd(&cnts[3], lb);
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) u192_add(part[threadIdx.x], part[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        double mean = a;
        if (a != b) {                                       // sorted positions k .. N-1-k: the copies of a and of b inside
            U192 S = part[0];
            const U192 ta = u192_times(a, (unsigned long)(cnts[1] + cnts[2] - (unsigned long long)k));
            const U192 tb = u192_times(b, (unsigned long)((unsigned long long)(N - k) - cnts[3]));
            u192_add(S, ta); u192_add(S, tb);
            mean = u192_mean(S, (uint64_t)(N - 2 * k));
        }
        scalars[DSQ_SC_COEF0] = mean;
        scalars[DSQ_SC_COEF1] = 0.0;
        status[DSQ_ST_TREND_STATUS] = 0;
        scalars[DSQ_SC_FIT_USED] = (double)DSQ_FIT_MEAN;
    }
}

__global__ void trend_given_kernel(double *scalars, int32_t *status) {
    scalars[DSQ_SC_COEF0] = dnan(); scalars[DSQ_SC_COEF1] = dnan();
    scalars[DSQ_SC_FIT_USED] = (double)DSQ_FIT_GIVEN;
    status[DSQ_ST_TREND_STATUS] = 0;
}

// ---- per-gene rules ----------------------------------------------------------------------------------------------
struct RuleParams {
    Rows rw;
    int n;                       // capacity / leading dimension of the n x p matrices
    int p;
    double minDisp, maxDisp, xim, outlierSD;
    const double *xim_dev;       // normalization-factor matrix: xim over the non-zero rows, computed by the chain
    int maxit, betaMaxit;
    const double *baseMean, *baseVar, *roughDisp;
    double *alpha_init, *la0;
    // fitDisp outputs
    const double *la_out, *initial_lp, *last_lp;
    const int32_t *iter;
    const double *la_grid;
    double *dge;
    int32_t *dispGeneIter;
    int32_t *grid_flag, *grid_rows, *grid_count;
    const double *scalars;
    double *dispFit, *log_dfit, *la_init, *dispMAP, *dispersion;
    const double *dispFit_in;      // DSQ_FIT_GIVEN: the caller's trend values, per gene
    int32_t *dispIter, *dispOutlier;
    // GLM fit
    const double *beta_nat, *beta_var, *beta_iter, *logLike;
    double *beta, *betaSE, *stat, *pvalue, *betaIter_out;
    int32_t *betaConv, *optim_flag, *optim_count;
    int32_t *optim_rows;           // the flagged rows, listed (any order) for the row-listed optim launch
    int wald;
    // the rows fitNbinomGLMsOptim re-fits (R/fitNbinomGLMs.R:340-407)
    const double *beta_init;       // the IRLS start values (natural-log scale)
    double *opt_start;             // n x p, log2 scale
    const int32_t *opt_conv;
};

// alpha_hat <- pmin(roughDisp, momentsDisp), bounded to [minDisp, maxDisp] (R/core.R:713-728, :2439-2448)
__global__ void alpha_init_kernel(RuleParams q) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows_count(q.rw)) return;
    const int g = rows_gene(q.rw, i);
    const double bm = q.baseMean[g], bv = q.baseVar[g];
    const double xim = q.xim_dev ? *q.xim_dev : q.xim;
    const double mom = (bv - xim * bm) / (bm * bm);
    double a = np_min(q.roughDisp[g], mom);
    a = np_min(np_max(q.minDisp, a), q.maxDisp);
    q.alpha_init[g] = a;
    q.la0[g] = dlog(a);
}

// after the gene-wise fitDisp: accept / convergence / refit rules (R/core.R:785, 826-835)
__global__ void gene_est_post_kernel(RuleParams q) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows_count(q.rw)) return;
    const int g = rows_gene(q.rw, i);
    double d = np_min(dexp(q.la_out[g]), q.maxDisp);
    const double ilp = q.initial_lp[g];
    if (q.last_lp[g] < ilp + __builtin_fabs(ilp) / 1e6) d = q.alpha_init[g];      // noIncrease
    const int it = q.iter[g];
    q.dispGeneIter[g] = it;
    const bool conv = (it < q.maxit) && !(it == 1);
    const bool refit = !conv && (d > q.minDisp * 10.0);
    q.dge[g] = d;
    q.grid_flag[g] = refit ? 1 : 0;
    if (refit) q.grid_rows[atomicAdd(q.grid_count, 1)] = g;
}

// dispGeneEst[refitDisp] <- exp(grid); final clamp (R/core.R:846-848)
__global__ void gene_est_final_kernel(RuleParams q) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows_count(q.rw)) return;
    const int g = rows_gene(q.rw, i);
    double d = q.dge[g];
    if (q.grid_flag[g]) d = dexp(q.la_grid[g]);
    q.dge[g] = np_min(np_max(d, q.minDisp), q.maxDisp);
}

// dispFit from the trend; start value and prior mean of the MAP search (R/core.R:1019-1024)
__global__ void map_init_kernel(RuleParams q) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows_count(q.rw)) return;
    const int g = rows_gene(q.rw, i);
    const double fit = q.dispFit_in ? q.dispFit_in[g] : q.scalars[DSQ_SC_COEF0] + q.scalars[DSQ_SC_COEF1] / q.baseMean[g];
    const double d = q.dge[g];
    double init = (d > 0.1 * fit) ? d : fit;
    if (init != init) init = fit;
    q.dispFit[g] = fit;
    q.log_dfit[g] = dlog(fit);
    q.la_init[g] = dlog(init);
}

// after the MAP fitDisp: dispMAP, convergence, stragglers to the grid (R/core.R:1042-1050)
__global__ void map_post_kernel(RuleParams q) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows_count(q.rw)) return;
    const int g = rows_gene(q.rw, i);
    const int it = q.iter[g];
    q.dispMAP[g] = dexp(q.la_out[g]);
    q.dispIter[g] = it;
    const bool refit = !(it < q.maxit);
    q.grid_flag[g] = refit ? 1 : 0;
    if (refit) q.grid_rows[atomicAdd(q.grid_count, 1)] = g;
}

// dispMAP[refit] <- exp(grid); clamp; dispOutlier; final dispersion (R/core.R:1061-1063, 1099-1115)
__global__ void map_final_kernel(RuleParams q) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows_count(q.rw)) return;
    const int g = rows_gene(q.rw, i);
    double dm = q.dispMAP[g];
    if (q.grid_flag[g]) dm = dexp(q.la_grid[g]);
    dm = np_min(np_max(dm, q.minDisp), q.maxDisp);
    q.dispMAP[g] = dm;
    const double d = q.dge[g];
    const double sd = __builtin_sqrt(q.scalars[DSQ_SC_VAR_LOG_DISP]);
    const bool outlier = dlog(d) > q.log_dfit[g] + q.outlierSD * sd;       // NaN compares false, as the mirror's mask
    q.dispOutlier[g] = outlier ? 1 : 0;
    q.dispersion[g] = outlier ? d : dm;
}

// the host half of fitNbinomGLMs (R/fitNbinomGLMs.R:185-211) + Wald statistic and p-value (R/core.R:1471,1507)
__global__ void beta_post_kernel(RuleParams q) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows_count(q.rw)) return;
    const int g = rows_gene(q.rw, i);
    const double log2e = 1.4426950408889634;
    bool stable = true, varpos = true;
    for (int c = 0; c < q.p; c++) {
        const double b = q.beta_nat[(size_t)g + (size_t)q.n * c], v = q.beta_var[(size_t)g + (size_t)q.n * c];
        if (b != b) stable = false;
        if (v <= 0.0) varpos = false;
        if (q.beta) {
            const double bl = log2e * b;
            const double se = log2e * __builtin_sqrt(np_max(v, 0.0));
            q.beta[(size_t)g + (size_t)q.n * c] = bl;
            q.betaSE[(size_t)g + (size_t)q.n * c] = se;
            if (q.wald) {
                const double z = bl / se;
                q.stat[(size_t)g + (size_t)q.n * c] = z;
                q.pvalue[(size_t)g + (size_t)q.n * c] = dpnorm_upper2(z);
            }
        }
    }
    const double it = q.beta_iter[g];
    const bool conv = it < (double)q.betaMaxit;
    if (q.betaConv) q.betaConv[g] = conv ? 1 : 0;
    if (q.betaIter_out) q.betaIter_out[g] = it;
    const bool optim = !conv || !stable || !varpos;
    q.optim_flag[g] = optim ? 1 : 0;
    if (optim) {
        const int k = atomicAdd(q.optim_count, 1);
        if (q.optim_rows) q.optim_rows[k] = g;
        // start values of the fallback (:350-355): the IRLS estimate (log2 scale) when it is finite and inside the box,
        // else the IRLS's own start values -- on the natural-log scale, as the reference passes them
        bool usable = stable;
        for (int c = 0; c < q.p && usable; c++)
            usable = __builtin_fabs(log2e * q.beta_nat[(size_t)g + (size_t)q.n * c]) < 30.0;
        for (int c = 0; c < q.p; c++)
            q.opt_start[(size_t)g + (size_t)q.n * c] = usable ? log2e * q.beta_nat[(size_t)g + (size_t)q.n * c]
                                                              : q.beta_init[(size_t)g + (size_t)q.n * c];
    }
}

// after the fallback: betaConv[row] <- TRUE where optim converged (:378-380); Wald statistic and p-value from its
// coefficients (the optim kernel has written beta / betaSE / logLike / mu of these rows in place)
__global__ void optim_post_kernel(RuleParams q) {
    const int cnt = rows_count(q.rw);                       // (a handful: the grid is small and strides)
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += gridDim.x * blockDim.x) {
        const int g = rows_gene(q.rw, i);
        if (q.betaConv && q.opt_conv[g]) q.betaConv[g] = 1;
        if (q.wald && q.stat) {
            for (int c = 0; c < q.p; c++) {
                const double z = q.beta[(size_t)g + (size_t)q.n * c] / q.betaSE[(size_t)g + (size_t)q.n * c];
                q.stat[(size_t)g + (size_t)q.n * c] = z;
                q.pvalue[(size_t)g + (size_t)q.n * c] = dpnorm_upper2(z);
            }
        }
    }
}

// rows flagged -> list (order irrelevant: the listed launches write results at the gene's own position)
__global__ void list_kernel(Rows rw, const int32_t *flag, int want, int32_t *rows_out, int32_t *count) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows_count(rw)) return;
    const int g = rows_gene(rw, i);
    if ((flag[g] != 0) == (want != 0)) rows_out[atomicAdd(count, 1)] = g;
}

// refitWithoutOutliers: result columns of rows that became all-zero are NA (R/core.R:2535), only when some row
// is actually refitted (:2496)
struct NaRowsParams {
    Rows rw;
    int n, p;
    const int32_t *allZero, *n_refit;
    double *beta, *betaSE, *stat, *pvalue, *betaIter, *logLike, *logLikeReduced, *maxCooks;
    int32_t *betaConv;
};
__global__ void na_rows_kernel(NaRowsParams q) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows_count(q.rw) || *q.n_refit <= 0) return;
    const int g = rows_gene(q.rw, i);
    if (!q.allZero[g]) return;
    const double nan = dnan();
    for (int c = 0; c < q.p; c++) {
        q.beta[(size_t)g + (size_t)q.n * c] = nan;
        q.betaSE[(size_t)g + (size_t)q.n * c] = nan;
        if (q.stat) { q.stat[(size_t)g + (size_t)q.n * c] = nan; q.pvalue[(size_t)g + (size_t)q.n * c] = nan; }
    }
    q.betaIter[g] = nan; q.logLike[g] = nan; q.maxCooks[g] = nan;
    if (q.logLikeReduced) q.logLikeReduced[g] = nan;
    q.betaConv[g] = -1;
}

// the n x m assays of the rows that were never fitted (all-zero counts, or weights that leave a degenerate design): NA, as
// buildMatrixWithNARows leaves them in R -- the kernels skip those rows, so without this they keep whatever the buffer
// held (found by the 8-range host-entry test of round 5: two calls returned different garbage there)
__global__ void __launch_bounds__(256) na_assay_rows_kernel(int n, int m, long ld, const int32_t *allZero, double *a0, double *a1,
                                                            int32_t *zero_p = nullptr, int zero_n = 0) {
    // (zero_p: the row-list counters of the outlier phase that follows in the same call -- a memset of 8 int32 at an odd
    //  offset is three fill commands of the runtime)
    if (blockIdx.x == 0 && (int)threadIdx.x < zero_n) zero_p[threadIdx.x] = 0;
    const int lane = threadIdx.x & 63;
    const int g = blockIdx.x * blockDim.x + threadIdx.x;            // one gene per lane for the flag ...
    unsigned long long todo = __ballot(g < n && allZero[g] != 0);
    const int base = g - lane;
    const double nan = dnan();
    while (todo) {                                                  // ... the (few) flagged rows of the wave written by all lanes
        const int l = __builtin_ctzll(todo);
        todo &= todo - 1;
        const size_t row = (size_t)(base + l) * ld;
        for (int j = lane; j < m; j += 64) {
            if (a0) a0[row + j] = nan;
            if (a1) a1[row + j] = nan;
        }
    }
}

// maxCooks after the refit (R/core.R:2538-2546): NA everywhere when every sample is replaceable, else the row
// maximum of the ORIGINAL Cook's distances over the samples in cells of >= 3, replaceable samples zeroed
__global__ void __launch_bounds__(256) masked_max_kernel(Rows rw, int m, long ld, const double *cooks, const int32_t *use,
                                                         const int32_t *zero, int all_replaceable, int valid,
                                                         const int32_t *n_refit, double *maxCooks) {
    if (*n_refit <= 0) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
    const int nw = rows_count(rw);
    for (int wi = blockIdx.x * waves + wave; wi < nw; wi += gridDim.x * waves) {
        const int g = rows_gene(rw, wi);
        if (all_replaceable || !valid) {
            if (lane == 0) maxCooks[g] = dnan();
            continue;
        }
        const double *ck = cooks + (size_t)g * ld;
        double mx = -__builtin_inf();
        int isnan_ = 0;
        for (int j = lane; j < m; j += 64) {
            if (!use[j]) continue;
            double v = zero[j] ? 0.0 : ck[j];
            if (v != v) isnan_ = 1;
            if (v > mx) mx = v;
        }
        double o, a, b;
        o = lane_xor1(mx); mx = (o > mx) ? o : mx;
        o = lane_xor2(mx); mx = (o > mx) ? o : mx;
        o = lane_xor4(mx); mx = (o > mx) ? o : mx;
        o = lane_xor8(mx); mx = (o > mx) ? o : mx;
        lane_pair16(mx, a, b); mx = (b > a) ? b : a;
        lane_pair32(mx, a, b); mx = (b > a) ? b : a;
        if (lane == 0) maxCooks[g] = __any(isnan_) ? dnan() : mx;
    }
}

// ---- longest-expected-first order of a full-size fit launch ---------------------------------------------------------
// The persistent fit kernels hand out genes in list order through a counter.  With many genes per wave slot the order does
// not matter (measured at 50 000 genes: 2 %); at one rank's share of an 8-GPU run (6 250 genes on 3 072 wave slots, two genes
// per slot) a launch ends when the last slot has worked off its two genes, and two slow genes that meet in one slot set the
// time: fit_beta 0.30 ms against 0.16 ms for an eighth of the 50 000-gene launch.  The second fit of a gene costs about what
// its first one did (the test's IRLS after the gene-wise IRLS: same counts, nearly the same dispersion), so the rows are
// listed by DESCENDING iteration count of the first fit: the slow genes start first, the quick ones fill the gaps
// (longest-processing-time-first).  A counting sort by one workgroup; the order inside a bin is whatever the atomics give --
// every gene's results are written at its own position and no kernel couples two genes, so the order changes no bit.
// What an order can and cannot do at 6 250 genes on 3 072 slots (a simulation on the iteration counts of that workload,
// cost = iterations): fit_beta in list order ends at 2.5x the ideal, longest-first at 1.8x = the slowest single gene (19
// iterations) -- the floor of any order; the dispersion searches end at 1.4x either way: 6 250 = 2.03 x 3 072, the last
// hundred genes are a third round whatever the order.
// small_first: the key is a MEAN COUNT and the slow genes are the ones with few counts (the gene-wise IRLS, which has no
// earlier fit to go by: its iteration count falls with log(baseMean), correlation -0.84 at C3) -- the bins are sixths of
// an octave of 1 + mean, taken in ascending order.
__global__ void __launch_bounds__(1024) lpt_order_kernel(Rows rw, const int32_t *key_i, const double *key_d, int small_first,
                                                         int32_t *out) {
    __shared__ int hist[128], cursor[128];
    const int cnt = rows_count(rw);
    if (threadIdx.x < 128) hist[threadIdx.x] = 0;
    __syncthreads();
    auto key_of = [&](int g) {
        double kd = key_i ? (double)key_i[g] : key_d[g];
        if (small_first) {
            kd = 127.0 - 8.656170245333781 * dlog(1.0 + kd);      // 6 / ln 2: a mean of 2^21 reaches bin 0
            if (kd < 0.0) kd = 0.0;
        }
        if (!(kd >= 0.0)) kd = 127.0;             // (NaN: an aborted fit -- treat as long)
        return kd > 127.0 ? 127 : (int)kd;
    };
    for (int i = threadIdx.x; i < cnt; i += 1024) atomicAdd(&hist[key_of(rows_gene(rw, i))], 1);
    __syncthreads();
    if (threadIdx.x == 0) {
        int acc = 0;
        for (int k = 127; k >= 0; k--) { cursor[k] = acc; acc += hist[k]; }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < cnt; i += 1024) {
        const int g = rows_gene(rw, i);
        out[atomicAdd(&cursor[key_of(g)], 1)] = g;
    }
}

// ---- the fills in front of a call's first kernel, as ONE launch ----------------------------------------------------
// A chain used to open with a dozen memset / copy commands (the NA patterns of the result columns, the status block, the
// dynamic-scheduling counters, the ridge / contrast block): ~ 5 us of dependent dispatch each -- 0.11 ms of a 2.7 ms step
// at one rank's share of C3.  The segments (4-byte aligned, word patterns) and the small block (by value) ride in the
// kernel's arguments: no host buffer is read after the launch returns.
struct InitSeg { uint32_t *p; uint32_t words; uint32_t val; };
struct InitParams {
    int nseg;
    InitSeg seg[24];
    double *blk_dst;
    int nblk;
    double blk[3 * DSQ_P_WIDE + 8];
};
__global__ void __launch_bounds__(256) chain_init_kernel(InitParams q) {
    const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x, nth = gridDim.x * blockDim.x;
    for (int s = 0; s < q.nseg; s++) {
        uint32_t *p = q.seg[s].p;
        const uint32_t v = q.seg[s].val, w = q.seg[s].words;
        for (uint32_t i = tid; i < w; i += nth) p[i] = v;
    }
    if (tid < (unsigned)q.nblk) q.blk_dst[tid] = q.blk[tid];
}

// ---- orchestration -----------------------------------------------------------------------------------------------
#define PIPE_HIP(expr)                                                                                   \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess)                                                                            \
            return capi_fail(e_ == hipErrorOutOfMemory ? DSQ_ERR_NOMEM : DSQ_ERR_DEVICE, "%s: %s", #expr, \
                             hipGetErrorString(e_));                                                     \
    } while (0)

static inline int kern_width(int p) { return p > DSQ_P_REG ? dsq_wide_width(p) : p; }

struct Pipe {
    const DsqDeseqArgs *a;
    const DsqDeseqOut *o;
    hipStream_t st;
    int n, m, p;
    long ld;
    double maxDisp, min_log_alpha;
    // workspace (device)
    double *roughDisp, *beta_init, *alpha_init, *la0, *la_out, *last_change, *initial_lp, *initial_dlp, *last_lp, *last_dlp;
    double *la_grid, *log_dfit, *la_init, *beta_nat, *beta_var, *beta_iter, *cnum, *cden, *dev, *lam, *contrast, *resbuf;
    double *trend_mean_c, *trend_disp_c, *robustDisp, *scratch, *cscratch;
    double *kconst;              // n: K' of each row from its last fitBeta launch (-> LogLikeKernelParams.kconst)
    double *opt_start, *opt_beta, *opt_se, *opt_ll;
    int32_t *iter, *iter_accept, *grid_flag, *rows_nz, *rows_grid, *rows_rep, *rows_refit, *counters, *work_counters;
    int32_t *rows_opt, *opt_conv;
    int32_t *rows_lpt;             // the non-zero rows in longest-expected-first order (lpt_order_kernel)
    double *lam_prior;             // betaPrior: 1 / betaPriorVar on the natural-log scale (device copy of a->lambda_prior)
    // WIDE designs (10 < p <= 48): the fit kernels run at the padded width pk = 16 / 24 / 32 / 48 on the design zero-padded to pk
    // columns (ridge 1, start value 0, contrast 0 on the padding: the real coefficients keep their bits, csrc/capi.hip
    // "wide designs"); the n x . work matrices have pk columns, the rule kernels and the results keep the true p
    int pk;
    const double *x_k;             // the design at the kernels' width (the caller's, or the padded copy)
    unsigned long long padmask;
    double *xim_cur;               // ... the one the rule kernels read now: over the non-zero rows, or (refit) over the refitted rows
    double *xim_dev;               // normalization-factor matrix: mean(1 / colMeans(nf)) over the non-zero rows (one double
                                   // behind the lambda block of the caller's workspace: it persists between the phases)
    // nbinomLRT against a reduced model that is not ~1 / the beta-prior refit (never both: the prior is Wald only)
    double *red_binit, *red_beta, *red_se, *red_mu;
    // ... a reduced model of more than DSQ_P_REG columns runs at ITS padded width (round 5): the reduced design zero-padded to
    // red_pk columns, ridge 1 on the padding (kept in the lambda block's third part, which only the beta prior uses otherwise)
    int red_pk;
    const double *red_x_k;
    double *red_lam;
    // ... and so does the (expanded) design of the beta-prior pass: pri_pk columns, the padded copy pri_x_k
    int pri_pk;
    const double *pri_x_k;
    const int32_t *red_cell_perm, *red_cell_start;
    int red_ncell;
    int32_t *cells_dev;            // perm | in3 | cell_start | use3 | replaceable
    void *trend_ws;
    int next_counter;
    const int32_t *cell_perm, *cell_start;   // design cells for the cell-collapsed fitBeta kernel (ncell = 0: general)
    int ncell;
    const char *tag;               // appended to the profile names of the refit chain's launches
    // settings of the test's GLM fits and the floor of the gene-wise estimate's fitted means.  The main chain takes the
    // caller's (DESeq() hands betaTol / maxit / useQR / minmu to nbinomWaldTest / nbinomLRT and minmu to
    // estimateDispersions -> estimateDispersionsGeneEst, R/core.R:393-405, R/methods.R:552, where it is the floor of
    // :763 -- the IRLS inside that fitNbinomGLMs call keeps its own default minmu = 0.5, :755-757); the refit of the
    // replaced rows runs every step on its DEFAULTS (refitWithoutOutliers passes none of them on, R/core.R:2509-2531)
    double t_tol, t_minmu, ge_floor;
    int t_maxit, t_useQR;
    // the full-row nbinomLogLike of the test's fit on a SIDE stream (overlap below): set while it is in flight
    hipStream_t side;
    hipEvent_t ev_fork, ev_join;
    bool overlap, forked, ll_pending;
    LogLikeKernelParams ll;        // the deferred full-row launch (test_fit -> run_chain)
    // host-side facts of the design cells
    int any3, maxcell, all_replaceable;
};

// the row-list counters of the chain ARE the caller's status block (no copies at the end of a call): a counter sits at
// the index of the DSQ_ST_* entry that reports it; the two spare entries count the optim rows of the reduced / MLE fits
enum { CNT_NZ = DSQ_ST_N_NONZERO, CNT_GRID1 = DSQ_ST_N_GRID_GENEEST, CNT_TREND = DSQ_ST_N_TREND, CNT_GRID2 = DSQ_ST_N_GRID_MAP,
       CNT_OPT1 = DSQ_ST_N_OPTIM_GENEEST, CNT_OPT2 = DSQ_ST_N_OPTIM_TEST, CNT_REP = DSQ_ST_N_REPLACE, CNT_REFIT = DSQ_ST_N_REFIT,
       CNT_GRID1R = DSQ_ST_N_GRID_GENEEST_REFIT, CNT_GRID2R = DSQ_ST_N_GRID_MAP_REFIT, CNT_OPT1R = DSQ_ST_N_OPTIM_GENEEST_REFIT,
       CNT_OPT2R = DSQ_ST_N_OPTIM_TEST_REFIT, CNT_OPT3 = 14, CNT_OPT3R = 15, CNT_N = DSQ_ST_COUNT };
static_assert(DSQ_ST_N_OPTIM_TEST_REFIT < 14 && DSQ_ST_COUNT == 16, "status block layout");

static int *next_work_counter(Pipe &P) {
    int *c = P.work_counters + (P.next_counter % 60);
    P.next_counter++;
    return c;
}

static inline dim3 ew_grid(int n) { return dim3((unsigned)((n + 255) / 256 > 0 ? (n + 255) / 256 : 1)); }

static RuleParams rule_params(const Pipe &P, const Rows &rw) {
    RuleParams q;
    memset(&q, 0, sizeof q);
    const DsqDeseqArgs *a = P.a;
    const DsqDeseqOut *o = P.o;
    q.rw = rw; q.n = P.n; q.p = P.p;
    q.minDisp = a->minDisp; q.maxDisp = P.maxDisp; q.xim = a->xim; q.outlierSD = a->outlierSD;
    q.xim_dev = a->nf_is_vector ? nullptr : P.xim_cur;
    q.maxit = a->maxit; q.betaMaxit = P.t_maxit;
    q.baseMean = o->baseMean; q.baseVar = o->baseVar; q.roughDisp = P.roughDisp;
    q.alpha_init = P.alpha_init; q.la0 = P.la0;
    q.la_out = P.la_out; q.initial_lp = P.initial_lp; q.last_lp = P.last_lp; q.iter = P.iter; q.la_grid = P.la_grid;
    q.dge = o->dispGeneEst; q.dispGeneIter = o->dispGeneIter;
    q.grid_flag = P.grid_flag; q.grid_rows = P.rows_grid;
    q.scalars = o->scalars;
    q.dispFit_in = a->dispFit_in;
    q.dispFit = o->dispFit; q.log_dfit = P.log_dfit; q.la_init = P.la_init; q.dispMAP = o->dispMAP;
    q.dispersion = o->dispersion; q.dispIter = o->dispIter; q.dispOutlier = o->dispOutlier;
    q.beta_nat = P.beta_nat; q.beta_var = P.beta_var; q.beta_iter = P.beta_iter;
    q.optim_rows = P.rows_opt; q.beta_init = P.beta_init; q.opt_start = P.opt_start; q.opt_conv = P.opt_conv;
    return q;
}

// which model matrix a GLM fit of the chain runs on: the design itself, nbinomLRT's reduced model, or the (expanded)
// design of the beta-prior refit with its ridge 1 / betaPriorVar (R/fitNbinomGLMs.R:311-325)
enum { DES_FULL = 0, DES_REDUCED = 1, DES_PRIOR = 2 };
struct DesignSel {
    const double *x, *beta_init, *lam;
    int p;                         // the width the kernels run at (the padded one for a wide design)
    const int32_t *cperm, *cstart;
    int ncell;
    int p_true;                    // the design's own number of columns
};
static DesignSel design_of(const Pipe &P, int which) {
    const DsqDeseqArgs *a = P.a;
    DesignSel d;
    if (which == DES_REDUCED) d = {P.red_x_k, P.red_binit, P.red_lam, P.red_pk, P.red_cell_perm, P.red_cell_start, P.red_ncell, a->p_red};
    else if (which == DES_PRIOR) d = {P.pri_x_k, a->prior_expanded ? P.red_binit : P.beta_init, P.lam_prior, P.pri_pk,
                                      P.cell_perm, P.cell_start, P.ncell, a->p_prior};
    else d = {P.x_k, P.beta_init, P.lam, P.pk, P.cell_perm, P.cell_start, P.ncell, P.p};
    return d;
}

static int launch_fit_beta(Pipe &P, const Rows &rw, const int32_t *y, const double *alpha, const double *weights,
                           double *mu_out, double mu_floor, double *hat, double tol, int maxit, int useQR, double minmu,
                           const char *name, int which = DES_FULL) {
    const DsqDeseqArgs *a = P.a;
    const DesignSel ds = design_of(P, which);
    BetaKernelParams kp;
    memset(&kp, 0, sizeof kp);
    kp.n = P.n; kp.m = P.m; kp.p = ds.p; kp.ld = P.ld;
    kp.y = y; kp.nf = a->nf; kp.nf_is_vector = a->nf_is_vector;
    kp.weights = a->useWeights ? weights : nullptr; kp.useWeights = a->useWeights ? 1 : 0;
    kp.x = ds.x; kp.alpha_hat = alpha; kp.contrast = P.contrast;
    kp.beta_init = ds.beta_init; kp.lambda = ds.lam;
    kp.tol = tol; kp.minmu = minmu; kp.mu_floor = mu_floor; kp.maxit = maxit; kp.useQR = useQR ? 1 : 0;
    kp.beta_mat = P.beta_nat; kp.beta_var_mat = P.beta_var; kp.iter = P.beta_iter;
    kp.contrast_num = P.cnum; kp.contrast_denom = P.cden; kp.deviance = P.dev;
    kp.hat_diagonals = hat; kp.mu_out = mu_out;
    kp.kconst_out = P.kconst;            // K' of the rows this launch fits: the nbinomLogLike launch behind it reads it
    kp.scratch = P.scratch; kp.cscratch = P.cscratch;
    kp.work_counter = next_work_counter(P);
    kp.rows = rw.rows; kp.n_dev = rw.n_dev; kp.rows_few = (rw.rows && rw.rows != P.rows_nz && rw.rows != P.rows_lpt) ? 1 : 0;
    kp.cell_perm = ds.cperm; kp.cell_start = ds.cstart; kp.ncell = ds.ncell;
    kp.p_true = ds.p_true;
    bool ok = false;
    char nm[32];
    snprintf(nm, sizeof nm, "%s%s", name, P.tag);
    capi_prof_begin(nm, P.n, P.st);
    PIPE_HIP(dispatch_fit_beta(kp.p, kp, P.st, &ok));
    capi_prof_end(P.st);
    if (!ok) return capi_fail(DSQ_ERR_UNSUPPORTED, "dsq_deseq_dev: no register kernel for p=%d", P.p);
    return DSQ_OK;
}

static int launch_fit_disp(Pipe &P, const Rows &rw, const int32_t *y, const double *mu, const double *la_in,
                           const double *prior_mean, bool usePrior, const double *weights, bool useCR, bool grid,
                           const char *name) {
    const DsqDeseqArgs *a = P.a;
    DispKernelParams kp;
    memset(&kp, 0, sizeof kp);
    kp.n = P.n; kp.m = P.m; kp.p = P.pk; kp.ld = P.ld;
    kp.y = y; kp.mu_hat = mu; kp.weights = a->useWeights ? weights : nullptr; kp.useWeights = a->useWeights ? 1 : 0;
    kp.x = P.x_k; kp.padmask = P.padmask;
    kp.log_alpha_in = la_in; kp.prior_mean = prior_mean;
    kp.prior_sigmasq = 1.0;
    kp.prior_sigmasq_dev = usePrior ? P.o->scalars + DSQ_SC_DISP_PRIOR_VAR : nullptr;
    kp.min_log_alpha = P.min_log_alpha; kp.kappa_0 = a->kappa_0; kp.tol = a->dispTol;
    kp.weightThreshold = a->weightThreshold; kp.maxit = a->maxit;
    kp.usePrior = usePrior ? 1 : 0; kp.useCR = useCR ? 1 : 0;
    kp.work_counter = next_work_counter(P);
    kp.rows = rw.rows; kp.n_dev = rw.n_dev; kp.rows_few = (rw.rows && rw.rows != P.rows_nz && rw.rows != P.rows_lpt) ? 1 : 0;
    if (P.p >= tuning().disp_cell_minp) { kp.cell_perm = P.cell_perm; kp.cell_start = P.cell_start; kp.ncell = P.ncell; }
    if (grid) {
        kp.grid = a->disp_grid; kp.ngrid = a->ngrid; kp.log_alpha = P.la_grid;
    } else {
        kp.log_alpha = P.la_out; kp.iter = P.iter; kp.iter_accept = P.iter_accept; kp.last_change = P.last_change;
        kp.initial_lp = P.initial_lp; kp.initial_dlp = P.initial_dlp; kp.last_lp = P.last_lp; kp.last_dlp = P.last_dlp;
        kp.last_d2lp = nullptr;          // never read by estimateDispersions* (R/core.R:784-787, 1042)
    }
    bool ok = false;
    char nm[32];
    snprintf(nm, sizeof nm, "%s%s", name, P.tag);
    capi_prof_begin(nm, P.n, P.st);
    PIPE_HIP(dispatch_fit_disp(P.pk, kp, P.st, grid, &ok));
    capi_prof_end(P.st);
    if (!ok) return capi_fail(DSQ_ERR_UNSUPPORTED, "dsq_deseq_dev: no register kernel for p=%d", P.p);
    return DSQ_OK;
}

static int launch_prefit_rows(Pipe &P, const Rows &rw, const int32_t *y) {
    const DsqDeseqArgs *a = P.a;
    PrefitKernelParams kp;
    memset(&kp, 0, sizeof kp);
    kp.n = P.n; kp.m = P.m; kp.p = P.p; kp.ld = P.ld;
    kp.y = y; kp.nf = a->nf; kp.nf_is_vector = a->nf_is_vector;
    kp.weights = a->useWeights ? a->weights_raw : nullptr; kp.useWeights = a->useWeights ? 1 : 0;
    kp.q = a->q; kp.a = a->a; kp.r = a->r;
    kp.baseMean = P.o->baseMean; kp.baseVar = P.o->baseVar; kp.roughDisp = P.roughDisp; kp.beta_init = P.beta_init;
    kp.allZero = P.o->allZero;
    kp.rows = rw.rows; kp.n_dev = rw.n_dev;
    bool ok = false;
    capi_prof_begin(rw.rows ? "prefit_moments:refit" : "prefit_moments", P.n, P.st);
    PIPE_HIP(launch_prefit(kp, P.st, &ok));
    capi_prof_end(P.st);
    if (!ok) return capi_fail(DSQ_ERR_UNSUPPORTED, "dsq_deseq_dev: p=%d", P.p);
    return DSQ_OK;
}

// the first p columns of two n x . work matrices -> the caller's n x p matrices, for the listed rows
__global__ void copy_rows_cols_kernel(Rows rw, int n, int p, const double *src_a, const double *src_b, double *dst_a, double *dst_b) {
    const int cnt = rows_count(rw);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += gridDim.x * blockDim.x) {
        const int g = rows_gene(rw, i);
        for (int c = 0; c < p; c++) {
            dst_a[(size_t)g + (size_t)n * c] = src_a[(size_t)g + (size_t)n * c];
            dst_b[(size_t)g + (size_t)n * c] = src_b[(size_t)g + (size_t)n * c];
        }
    }
}

// fitNbinomGLMsOptim (R/fitNbinomGLMs.R:340-407) on the rows beta_post_kernel listed (their number lives on the device:
// usually zero, the launch then finds nothing to do): start values from P.opt_start, coefficients / standard errors /
// logLike / fitted means written at the rows' own positions
static int launch_optim(Pipe &P, int cnt_optim, const int32_t *y, const double *alpha, const double *weights, double minmu,
                        double mu_floor, double *beta, double *betaSE, double *loglike, double *mu_out, int which = DES_FULL) {
    const DsqDeseqArgs *a = P.a;
    const DesignSel ds = design_of(P, which);
    OptimKernelParams kp;
    memset(&kp, 0, sizeof kp);
    kp.n = P.n; kp.m = P.m; kp.p = ds.p; kp.ld = P.ld;
    kp.y = y; kp.nf = a->nf; kp.nf_is_vector = a->nf_is_vector;
    kp.weights = a->useWeights ? weights : nullptr; kp.useWeights = a->useWeights ? 1 : 0;
    kp.x = ds.x; kp.alpha_hat = alpha; kp.lamnat = ds.lam; kp.beta_start = P.opt_start;
    kp.minmu = minmu; kp.mu_floor = mu_floor;
    // (a padded design: the kernel writes ds.p columns -- into the work matrices; the listed rows' true columns are copied
    //  to the caller's n x p matrices behind the launch)
    const bool via_work = ds.p != ds.p_true && which != DES_REDUCED && beta != P.opt_beta;     // (the reduced fit's coefficients are work matrices already)
    kp.beta = via_work ? P.opt_beta : beta; kp.betaSE = via_work ? P.opt_se : betaSE;
    kp.conv = P.opt_conv; kp.mu_out = mu_out; kp.loglike = loglike;
    kp.rows = P.rows_opt; kp.n_dev = P.counters + cnt_optim;
    bool ok = false;
    char nm[32];
    snprintf(nm, sizeof nm, "optim_rows%s", P.tag);
    capi_prof_begin(nm, P.n, P.st);
    PIPE_HIP(dispatch_optim_rows(kp.p, kp, P.st, &ok));
    capi_prof_end(P.st);
    if (!ok) return capi_fail(DSQ_ERR_UNSUPPORTED, "dsq_deseq_dev: no optim kernel for p=%d", P.p);
    if (via_work) {
        const Rows orw = {P.rows_opt, P.counters + cnt_optim, P.n};
        hipLaunchKernelGGL(copy_rows_cols_kernel, dim3(16), dim3(256), 0, P.st, orw, P.n, ds.p_true, (const double *)P.opt_beta,
                           (const double *)P.opt_se, beta, betaSE);
        PIPE_HIP(hipGetLastError());
    }
    return DSQ_OK;
}

// the rows of a full-size launch in longest-expected-first order (see lpt_order_kernel); DSQ_LPT=0 switches it off
static Rows lpt_rows(Pipe &P, const Rows &rw, const int32_t *key_i, const double *key_d, int small_first = 0) {
    // only where it pays: below ~ 5 genes per resident wave slot (measured, C3 shapes: 6 250 genes fit_beta 0.305 -> 0.256 ms;
    // 50 000 genes: the launch gains 0.03 ms and the one-workgroup sort in front of it costs 0.1)
    static const bool on = !(getenv("DSQ_LPT") && atoi(getenv("DSQ_LPT")) == 0);
    static const int maxn = getenv("DSQ_LPT_MAXN") ? atoi(getenv("DSQ_LPT_MAXN")) : 16384;
    if (!on || rw.rows != P.rows_nz || P.n > maxn) return rw;
    hipLaunchKernelGGL(lpt_order_kernel, dim3(1), dim3(1024), 0, P.st, rw, key_i, key_d, small_first, P.rows_lpt);
    return Rows{P.rows_lpt, rw.n_dev, rw.n};
}

// estimateDispersionsGeneEst on the rows `rw` of the count matrix y (R/core.R:657-860, niter = 1); mu-hat -> mu_hat
static int gene_est(Pipe &P, const Rows &rw, const int32_t *y, double *mu_hat, int cnt_grid, int cnt_optim,
                    int32_t *optim_flag) {
    const DsqDeseqArgs *a = P.a;
    RuleParams q = rule_params(P, rw);
    hipLaunchKernelGGL(alpha_init_kernel, ew_grid(P.n), dim3(256), 0, P.st, q);
    int rc;
    if (a->linearMu) {
        PrefitKernelParams kp;
        memset(&kp, 0, sizeof kp);
        kp.n = P.n; kp.m = P.m; kp.p = P.p; kp.ld = P.ld; kp.y = y; kp.nf = a->nf; kp.nf_is_vector = a->nf_is_vector;
        kp.q = a->q; kp.a = a->a;
        kp.rows = rw.rows; kp.n_dev = rw.n_dev;
        bool ok = false;
        capi_prof_begin("linear_mu", P.n, P.st);
        PIPE_HIP(launch_linear_mu(kp, P.ge_floor, mu_hat, P.st, &ok));     // minmu of estimateDispersionsGeneEst (:763)
        capi_prof_end(P.st);
        if (!ok) return capi_fail(DSQ_ERR_UNSUPPORTED, "dsq_deseq_dev: p=%d", P.p);
    } else {
        // fitNbinomGLMs(alpha_hat = alpha_hat) with mu floored at minmu (R/core.R:755-763); rows the IRLS leaves
        // to the optim fallback are flagged for the caller
        // (the arguments of THIS fitNbinomGLMs call are its defaults -- betaTol 1e-8, maxit 100, QR, the IRLS's own minmu
        // 0.5, R/core.R:755-757; the caller's minmu is the FLOOR of the fitted means it hands to the search, :763)
        // (small launches: the rows with the fewest counts first -- they take the most iterations; see lpt_order_kernel)
        rc = launch_fit_beta(P, lpt_rows(P, rw, nullptr, P.o->baseMean, 1), y, P.alpha_init, a->weights_norm, mu_hat, P.ge_floor, nullptr,
                             1e-8, 100, 1, 0.5, "fit_beta");
        if (rc) return rc;
        RuleParams b = rule_params(P, rw);
        b.betaMaxit = 100;
        b.optim_flag = optim_flag; b.optim_count = P.counters + cnt_optim;
        hipLaunchKernelGGL(beta_post_kernel, ew_grid(P.n), dim3(256), 0, P.st, b);
        // rows the IRLS left: the fallback's fitted means (floored at minmu, :763) replace theirs before the search
        rc = launch_optim(P, cnt_optim, y, P.alpha_init, a->weights_norm, 0.5, P.ge_floor, P.opt_beta, P.opt_se, P.opt_ll, mu_hat);
        if (rc) return rc;
    }
    rc = launch_fit_disp(P, rw, y, mu_hat, P.la0, P.la0, false, a->weights_floor, a->useCR != 0, false, "fit_disp");
    if (rc) return rc;
    q.grid_count = P.counters + cnt_grid;
    hipLaunchKernelGGL(gene_est_post_kernel, ew_grid(P.n), dim3(256), 0, P.st, q);
    Rows gr = {P.rows_grid, P.counters + cnt_grid, P.n};
    rc = launch_fit_disp(P, gr, y, mu_hat, P.la0, P.la0, false, a->weights_floor, a->useCR != 0, true, "fit_disp_grid");
    if (rc) return rc;
    hipLaunchKernelGGL(gene_est_final_kernel, ew_grid(P.n), dim3(256), 0, P.st, q);
    PIPE_HIP(hipGetLastError());
    return DSQ_OK;
}

// estimateDispersionsMAP (R/core.R:943-1131) on the rows `rw`
static int map_est(Pipe &P, const Rows &rw, const int32_t *y, const double *mu_hat, int cnt_grid) {
    const DsqDeseqArgs *a = P.a;
    RuleParams q = rule_params(P, rw);
    hipLaunchKernelGGL(map_init_kernel, ew_grid(P.n), dim3(256), 0, P.st, q);
    // (no longest-first order here: the gene-wise search's iteration count does not predict the MAP search's -- other start
    //  value, the prior --; measured at 6 250 genes: 0.435 ms either way)
    int rc = launch_fit_disp(P, rw, y, mu_hat, P.la_init, P.log_dfit, true, a->weights_norm, a->useCR != 0, false, "fit_disp");
    if (rc) return rc;
    q.grid_count = P.counters + cnt_grid;
    hipLaunchKernelGGL(map_post_kernel, ew_grid(P.n), dim3(256), 0, P.st, q);
    Rows gr = {P.rows_grid, P.counters + cnt_grid, P.n};
    rc = launch_fit_disp(P, gr, y, mu_hat, P.la_init, P.log_dfit, true, a->weights_norm, true, true, "fit_disp_grid");   // useCR = TRUE, :1061
    if (rc) return rc;
    hipLaunchKernelGGL(map_final_kernel, ew_grid(P.n), dim3(256), 0, P.st, q);
    PIPE_HIP(hipGetLastError());
    return DSQ_OK;
}

// the start values of a GLM fit on a rank-deficient (expanded) model matrix (R/fitNbinomGLMs.R:146-155): the intercept
// column starts at the log of the UNWEIGHTED mean normalized count, every other coefficient at 0 (or all at 1 when the
// first column is not an intercept)
__global__ void prior_start_kernel(Rows rw, int n, int p, int intercept, const double *bm, double *binit) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows_count(rw)) return;
    const int g = rows_gene(rw, i);
    for (int c = 0; c < p; c++) binit[(size_t)g + (size_t)n * c] = intercept ? (c == 0 ? dlog(bm[g]) : 0.0) : 1.0;
}

// fitGLMsWithPrior's first pass (R/fitNbinomGLMs.R:256-260): the MLE fit on the design with the wide prior.  Its fitted
// means and hat diagonals are the ones the object keeps (R/core.R:1429-1431); its coefficients feed the prior variance.
static int mle_fit(Pipe &P, const Rows &rw, const int32_t *y, double *mu_out, double *hat) {
    const DsqDeseqArgs *a = P.a;
    const DsqDeseqOut *o = P.o;
    int rc = launch_fit_beta(P, rw, y, o->dispersion, a->weights_norm, mu_out, 0.0, hat, P.t_tol, P.t_maxit, P.t_useQR,
                             P.t_minmu, "fit_beta_mle");
    if (rc) return rc;
    const int cnt3 = P.tag[0] ? CNT_OPT3R : CNT_OPT3;
    RuleParams b = rule_params(P, rw);
    b.beta = o->mle_beta; b.betaSE = P.red_se; b.wald = 0;
    b.optim_flag = P.grid_flag; b.optim_count = P.counters + cnt3;           // (the grid flags are free between the searches)
    hipLaunchKernelGGL(beta_post_kernel, ew_grid(P.n), dim3(256), 0, P.st, b);
    rc = launch_optim(P, cnt3, y, o->dispersion, a->weights_norm, P.t_minmu, 0.0, o->mle_beta, P.red_se, P.opt_ll, mu_out);
    if (rc) return rc;
    PIPE_HIP(hipGetLastError());
    return DSQ_OK;
}

// ... and its second pass (:311-325): the fit with lambda = 1 / betaPriorVar on the standard or the expanded model
// matrix; coefficients, standard errors, Wald statistics, betaConv, betaIter and logLike are this fit's
static int prior_fit(Pipe &P, const Rows &rw, const int32_t *y, int cnt_optim) {
    const DsqDeseqArgs *a = P.a;
    const DsqDeseqOut *o = P.o;
    int rc;
    if (a->prior_expanded) {
        PrefitKernelParams pk;            // getBaseMeansAndVariances without weights on the intercept-only design
        memset(&pk, 0, sizeof pk);
        pk.n = P.n; pk.m = P.m; pk.p = 1; pk.ld = P.ld; pk.y = y; pk.nf = a->nf; pk.nf_is_vector = a->nf_is_vector;
        pk.q = a->x_prior; pk.a = a->x_prior; pk.r = P.lam;                      // (only baseMean is read)
        pk.baseMean = P.cnum; pk.baseVar = P.cden; pk.roughDisp = P.dev; pk.allZero = P.opt_conv; pk.beta_init = P.opt_ll;
        pk.rows = rw.rows; pk.n_dev = rw.n_dev;
        bool ok = false;
        PIPE_HIP(launch_prefit(pk, P.st, &ok));
        if (!ok) return capi_fail(DSQ_ERR_UNSUPPORTED, "dsq_deseq_dev: prefit p=1");
        hipLaunchKernelGGL(prior_start_kernel, ew_grid(P.n), dim3(256), 0, P.st, rw, P.n, a->p_prior, a->prior_intercept,
                           (const double *)P.cnum, P.red_binit);
        if (P.pri_pk > a->p_prior)        // (start values 0 on the padding of a wide expanded design)
            PIPE_HIP(hipMemsetAsync(P.red_binit + (size_t)P.n * a->p_prior, 0, (size_t)P.n * (P.pri_pk - a->p_prior) * sizeof(double), P.st));
    }
    rc = launch_fit_beta(P, rw, y, o->dispersion, a->weights_norm, P.red_mu, 0.0, nullptr, P.t_tol, P.t_maxit, P.t_useQR,
                         P.t_minmu, "fit_beta_prior", DES_PRIOR);
    if (rc) return rc;
    LogLikeKernelParams lk;
    memset(&lk, 0, sizeof lk);
    lk.n = P.n; lk.m = P.m; lk.ld = P.ld; lk.y = y; lk.mu = P.red_mu; lk.disp = o->dispersion;
    lk.weights = a->useWeights ? a->weights_norm : nullptr; lk.useWeights = a->useWeights ? 1 : 0;
    lk.loglike = o->logLike; lk.rows = rw.rows; lk.n_dev = rw.n_dev; lk.kconst = P.kconst;
    capi_prof_begin(P.tag[0] ? "nbinom_loglike:refit" : "nbinom_loglike", P.n, P.st);
    PIPE_HIP(launch_loglike(lk, P.st));
    capi_prof_end(P.st);
    const DesignSel ds = design_of(P, DES_PRIOR);
    RuleParams b = rule_params(P, rw);
    b.p = ds.p_true; b.beta_init = ds.beta_init;
    b.beta = o->beta; b.betaSE = o->betaSE; b.stat = o->stat; b.pvalue = o->pvalue; b.wald = 1;
    b.betaConv = o->betaConv; b.betaIter_out = o->betaIter;
    b.optim_flag = o->optim_test; b.optim_count = P.counters + cnt_optim;
    hipLaunchKernelGGL(beta_post_kernel, ew_grid(P.n), dim3(256), 0, P.st, b);
    if (P.pri_pk > ds.p_true)             // (columns p_prior .. of the optim start values may hold another fit's: zero on the padding)
        PIPE_HIP(hipMemsetAsync(P.opt_start + (size_t)P.n * ds.p_true, 0, (size_t)P.n * (P.pri_pk - ds.p_true) * sizeof(double), P.st));
    rc = launch_optim(P, cnt_optim, y, o->dispersion, a->weights_norm, P.t_minmu, 0.0, o->beta, o->betaSE, o->logLike, P.red_mu,
                      DES_PRIOR);
    if (rc) return rc;
    const Rows orw = {P.rows_opt, P.counters + cnt_optim, P.n};
    RuleParams ob = rule_params(P, orw);
    ob.p = ds.p_true;
    ob.beta = o->beta; ob.betaSE = o->betaSE; ob.stat = o->stat; ob.pvalue = o->pvalue; ob.wald = 1; ob.betaConv = o->betaConv;
    hipLaunchKernelGGL(optim_post_kernel, dim3(16), dim3(256), 0, P.st, ob);
    PIPE_HIP(hipGetLastError());
    return DSQ_OK;
}

hipError_t launch_loglike_side(const LogLikeKernelParams &kp, hipStream_t st);       // aux.hip
hipError_t launch_xim_flagged(const double *nf, int n, int m, long ld, const int32_t *want_a, const int32_t *want_b, double *scratch_m,
                              double *out, hipStream_t st);
// the side stream's work has to be finished before anything that rewrites what it reads or reads what it writes
static int join_side(Pipe &P) {
    if (!P.forked) return DSQ_OK;
    P.forked = false;
    PIPE_HIP(hipEventRecord(P.ev_join, P.side));
    PIPE_HIP(hipStreamWaitEvent(P.st, P.ev_join, 0));
    return DSQ_OK;
}

// the deferred full-row nbinomLogLike: on the side stream (forked here) or on the chain's own stream
static int launch_pending_ll(Pipe &P, bool beside) {
    if (!P.ll_pending) return DSQ_OK;
    P.ll_pending = false;
    if (beside) {
        PIPE_HIP(hipEventRecord(P.ev_fork, P.st));
        PIPE_HIP(hipStreamWaitEvent(P.side, P.ev_fork, 0));
        PIPE_HIP(launch_loglike_side(P.ll, P.side));
        P.forked = true;
    } else {
        PIPE_HIP(launch_loglike(P.ll, P.st));
    }
    return DSQ_OK;
}

// nbinomWaldTest / nbinomLRT(reduced = ~1) on the rows `rw` (R/core.R:1403-1408, 1471, 1507; 1850-1878)
static int test_fit(Pipe &P, const Rows &rw, const int32_t *y, double *mu_out, double *hat, int cnt_optim) {
    const DsqDeseqArgs *a = P.a;
    const DsqDeseqOut *o = P.o;
    if (a->betaPrior) return mle_fit(P, rw, y, mu_out, hat);         // (the prior fit follows once lambda is known)
    // (P.beta_iter: the iteration counts of the gene-wise estimate's IRLS on the same rows, when that fit ran)
    static const bool by_mean = getenv("DSQ_LPT_KEY2") && atoi(getenv("DSQ_LPT_KEY2")) == 1;
    int rc = launch_fit_beta(P, by_mean ? lpt_rows(P, rw, nullptr, o->baseMean, 1) : (a->linearMu ? rw : lpt_rows(P, rw, nullptr, P.beta_iter)),
                             y, o->dispersion, a->weights_norm, mu_out, 0.0, hat,
                             P.t_tol, P.t_maxit, P.t_useQR, P.t_minmu, "fit_beta");
    if (rc) return rc;
    LogLikeKernelParams lk;
    memset(&lk, 0, sizeof lk);
    lk.n = P.n; lk.m = P.m; lk.ld = P.ld; lk.y = y; lk.mu = mu_out; lk.disp = o->dispersion;
    lk.weights = a->useWeights ? a->weights_norm : nullptr; lk.useWeights = a->useWeights ? 1 : 0;
    lk.loglike = o->logLike; lk.rows = rw.rows; lk.n_dev = rw.n_dev; lk.kconst = P.kconst;
    RuleParams b = rule_params(P, rw);
    b.beta = o->beta; b.betaSE = o->betaSE; b.stat = o->stat; b.pvalue = o->pvalue; b.wald = (a->test == 0) ? 1 : 0;
    b.betaConv = o->betaConv; b.betaIter_out = o->betaIter;
    b.optim_flag = o->optim_test; b.optim_count = P.counters + cnt_optim;
    // OVERLAP (the main chain, when this call also runs the outlier phase): nothing on the way to the refit of the replaced
    // rows reads the log likelihoods -- beta_post / the optim fallback / Cook's distances / replaceOutliers / the refit's
    // own dispersion searches -- and the refit is latency, not throughput: a handful of rows, each one gene's serial search
    // (~0.45 ms of a 12.8 ms step at C3 on an otherwise idle device).  So the full-row nbinomLogLike is DEFERRED: run_chain
    // launches it on a side stream when the refit starts (beside Cook's distances, another full-size launch, it would only
    // share the device: measured).  It leaves the rows flagged for the optim fallback alone (`skip`): the fallback
    // writes their logLike (and rewrites their fitted means) itself, R/fitNbinomGLMs.R:386,398-399.  What it may read
    // half-updated -- the dispersion of a row the refit is re-estimating -- only feeds that row's logLike, which the
    // refit's own test fit writes after the join (run(): join_side before the refit's test_fit).
    const bool overlap = P.overlap && !P.tag[0];
    if (!overlap) {
        capi_prof_begin(P.tag[0] ? "nbinom_loglike:refit" : "nbinom_loglike", P.n, P.st);
        PIPE_HIP(launch_loglike(lk, P.st));
        capi_prof_end(P.st);
    }
    hipLaunchKernelGGL(beta_post_kernel, ew_grid(P.n), dim3(256), 0, P.st, b);
    if (overlap) {
        // (launched by run_chain: beside the refit of the replaced rows when there is one, else right behind this fit)
        P.ll = lk;
        P.ll.skip = o->optim_test;
        P.ll_pending = true;
    }
    // rows for the optim fallback (R/fitNbinomGLMs.R:203-227): coefficients, standard errors, logLike (:398-399) and
    // fitted means (:386) of those rows in place, then betaConv and the Wald columns from them
    rc = launch_optim(P, cnt_optim, y, o->dispersion, a->weights_norm, P.t_minmu, 0.0, o->beta, o->betaSE, o->logLike, mu_out);
    if (rc) return rc;
    {
        const Rows orw = {P.rows_opt, P.counters + cnt_optim, P.n};
        RuleParams ob = rule_params(P, orw);
        ob.beta = o->beta; ob.betaSE = o->betaSE; ob.stat = o->stat; ob.pvalue = o->pvalue; ob.wald = b.wald;
        ob.betaConv = o->betaConv;
        hipLaunchKernelGGL(optim_post_kernel, dim3(16), dim3(256), 0, P.st, ob);
    }
    if (a->test == 1 && a->x_red) {
        // nbinomLRT's reduced fit (R/core.R:1856-1868): fitNbinomGLMs on the reduced model matrix at the same
        // dispersions -- QR start values, IRLS, logLik at its fitted means, its own optim-fallback rows
        PrefitKernelParams pk;
        memset(&pk, 0, sizeof pk);
        pk.n = P.n; pk.m = P.m; pk.p = a->p_red; pk.ld = P.ld; pk.y = y; pk.nf = a->nf; pk.nf_is_vector = a->nf_is_vector;
        pk.q = a->q_red; pk.a = a->a_red; pk.r = a->r_red;
        pk.baseMean = P.cnum; pk.baseVar = P.cden; pk.roughDisp = P.dev; pk.allZero = P.opt_conv;      // (not read)
        pk.beta_init = P.red_binit;
        pk.rows = rw.rows; pk.n_dev = rw.n_dev;
        bool ok = false;
        capi_prof_begin(P.tag[0] ? "prefit_reduced:refit" : "prefit_reduced", P.n, P.st);
        PIPE_HIP(launch_prefit(pk, P.st, &ok));
        capi_prof_end(P.st);
        if (!ok) return capi_fail(DSQ_ERR_UNSUPPORTED, "dsq_deseq_dev: reduced design with p=%d", a->p_red);
        rc = launch_fit_beta(P, rw, y, o->dispersion, a->weights_norm, P.red_mu, 0.0, nullptr, P.t_tol, P.t_maxit, P.t_useQR,
                             P.t_minmu, "fit_beta_reduced", DES_REDUCED);
        if (rc) return rc;
        lk.mu = P.red_mu; lk.loglike = o->logLikeReduced;
        capi_prof_begin(P.tag[0] ? "nbinom_loglike_red:refit" : "nbinom_loglike_red", P.n, P.st);
        PIPE_HIP(launch_loglike(lk, P.st));
        capi_prof_end(P.st);
        const int cnt3 = P.tag[0] ? CNT_OPT3R : CNT_OPT3;
        RuleParams rb = rule_params(P, rw);
        rb.p = a->p_red; rb.beta_init = P.red_binit;
        rb.optim_flag = P.grid_flag; rb.optim_count = P.counters + cnt3;      // (the grid flags are free between the searches)
        hipLaunchKernelGGL(beta_post_kernel, ew_grid(P.n), dim3(256), 0, P.st, rb);
        if (P.red_pk > a->p_red)         // (columns p_red .. of the start values may hold the full fit's: zero on the padding)
            PIPE_HIP(hipMemsetAsync(P.opt_start + (size_t)P.n * a->p_red, 0, (size_t)P.n * (P.red_pk - a->p_red) * sizeof(double), P.st));
        rc = launch_optim(P, cnt3, y, o->dispersion, a->weights_norm, P.t_minmu, 0.0, P.red_beta, P.red_se, o->logLikeReduced,
                          P.red_mu, DES_REDUCED);
        if (rc) return rc;
    } else if (a->test == 1) {
        InterceptKernelParams ik;
        memset(&ik, 0, sizeof ik);
        ik.n = P.n; ik.m = P.m; ik.ld = P.ld; ik.y = y; ik.nf = a->nf; ik.nf_is_vector = a->nf_is_vector;
        ik.weights = a->useWeights ? a->weights_norm : nullptr; ik.useWeights = a->useWeights ? 1 : 0;
        ik.alpha = o->dispersion; ik.beta_log2 = P.cnum; ik.betaSE = P.cden;      // not read by nbinomLRT
        ik.loglike = o->logLikeReduced; ik.rows = rw.rows; ik.n_dev = rw.n_dev;
        ik.kconst = P.kconst;            // (the full model's fit of the same rows: same counts, dispersions, weights)
        capi_prof_begin("intercept_fit", P.n, P.st);
        PIPE_HIP(launch_intercept_fit(ik, P.st));
        capi_prof_end(P.st);
    }
    PIPE_HIP(hipGetLastError());
    return DSQ_OK;
}

// design cells -> sample permutation grouped by cell, offsets, ">= 3 in cell" flags (nOrMoreInCell, R/core.R:2366), the
// replaceable flags: small host arrays, uploaded for the outlier kernels
struct OutlierMeta {
    int32_t *dperm, *din3, *drepl, *duse3, *dstart;
    int maxcell, any3, all_rep;
};
static int outlier_meta(const DsqDeseqArgs *a, int m, hipStream_t st, OutlierMeta *M) {
    static thread_local std::vector<int32_t> meta;   // (capi_upload_table takes its own copy)
    meta.assign((size_t)4 * m + a->ncell + 1, 0);
    int32_t *perm = meta.data(), *in3 = perm + m, *repl = in3 + m, *use3 = repl + m, *start = use3 + m;
    for (int j = 0; j < m; j++) {
        if (a->cell_of[j] < 0 || a->cell_of[j] >= a->ncell) return capi_fail(DSQ_ERR_VALUE, "cell_of[%d] out of range", j);
        start[a->cell_of[j] + 1]++;
    }
    int maxcell = 0, any3 = 0;
    for (int c = 0; c < a->ncell; c++) {
        int sz = start[c + 1];
        if (sz > maxcell) maxcell = sz;
        if (sz >= 3) any3 = 1;
        start[c + 1] += start[c];
    }
    {
        std::vector<int32_t> fill(start, start + a->ncell);
        for (int j = 0; j < m; j++) perm[fill[a->cell_of[j]]++] = j;
    }
    int all_rep = 1;
    for (int j = 0; j < m; j++) {
        in3[j] = (start[a->cell_of[j] + 1] - start[a->cell_of[j]]) >= 3;
        use3[j] = in3[j];
        repl[j] = a->replaceable[j] ? 1 : 0;
        if (!repl[j]) all_rep = 0;
    }
    void *mv;
    int rc = capi_upload_table(DSQ_WS_PIPE_META + 1, meta.data(), meta.size() * sizeof(int32_t), st, &mv);
    if (rc) return rc;
    M->dperm = (int32_t *)mv; M->din3 = M->dperm + m; M->drepl = M->din3 + m; M->duse3 = M->drepl + m; M->dstart = M->duse3 + m;
    M->maxcell = maxcell; M->any3 = any3; M->all_rep = all_rep;
    return DSQ_OK;
}

// the closing steps of refitWithoutOutliers that ask whether ANY row of the analysis was refitted (R/core.R:2496):
// result columns of the rows that became all zero are NA (:2535), maxCooks is recomputed (:2538-2546)
static int outlier_finish(Pipe &P, const Rows &nz, const Rows &rep, const OutlierMeta &M, const int32_t *n_refit) {
    const DsqDeseqOut *o = P.o;
    const int n = P.n, m = P.m, p = P.p;
    NaRowsParams nr;
    memset(&nr, 0, sizeof nr);
    nr.rw = rep; nr.n = n; nr.p = P.a->betaPrior ? P.a->p_prior : p; nr.allZero = o->allZero; nr.n_refit = n_refit;
    nr.beta = o->beta; nr.betaSE = o->betaSE; nr.stat = o->stat; nr.pvalue = o->pvalue; nr.betaIter = o->betaIter;
    nr.logLike = o->logLike; nr.logLikeReduced = o->logLikeReduced; nr.maxCooks = o->maxCooks; nr.betaConv = o->betaConv;
    hipLaunchKernelGGL(na_rows_kernel, ew_grid(n), dim3(256), 0, P.st, nr);
    int grid = (n + 3) / 4, capg = device_cu_count() * 8;
    if (grid > capg) grid = capg;
    hipLaunchKernelGGL(masked_max_kernel, dim3(grid), dim3(256), 0, P.st, nz, m, P.ld, (const double *)o->cooks,
                       (const int32_t *)M.duse3, (const int32_t *)M.drepl, M.all_rep, (m > p && M.any3) ? 1 : 0, n_refit,
                       o->maxCooks);
    PIPE_HIP(hipGetLastError());
    return DSQ_OK;
}

static size_t align8(size_t v) { return (v + 7) & ~(size_t)7; }

struct Carve {
    size_t o_rough, o_binit, o_ainit, o_la0, o_laout, o_lchg, o_ilp, o_idlp, o_llp, o_ldlp, o_lagrid, o_ldfit, o_lainit,
        o_bnat, o_bvar, o_biter, o_cnum, o_cden, o_dev, o_lam, o_res, o_tm, o_td, o_robust, o_ostart, o_obeta, o_ose, o_oll, o_rbinit, o_rbeta, o_rse, o_kc, dbl;
    size_t i_iter, i_itacc, i_gflag, i_nz, i_grid, i_rep, i_refit, i_cnt, i_wc, i_opt, i_oconv, i_lpt, ints;
    size_t bytes;
};

static Carve carve(int n, int p, int nt) {
    Carve c;
    const size_t nd = align8((size_t)n), np_ = align8((size_t)n * p), ntd = align8((size_t)nt);
    size_t d = 0;
    auto takeD = [&](size_t k) { size_t off = d; d += align8(k); return off; };
    c.o_rough = takeD(nd); c.o_binit = takeD(np_); c.o_ainit = takeD(nd); c.o_la0 = takeD(nd); c.o_laout = takeD(nd);
    c.o_lchg = takeD(nd); c.o_ilp = takeD(nd); c.o_idlp = takeD(nd); c.o_llp = takeD(nd); c.o_ldlp = takeD(nd);
    c.o_lagrid = takeD(nd); c.o_ldfit = takeD(nd); c.o_lainit = takeD(nd); c.o_bnat = takeD(np_); c.o_bvar = takeD(np_);
    c.o_biter = takeD(nd); c.o_cnum = takeD(nd); c.o_cden = takeD(nd); c.o_dev = takeD(nd); c.o_lam = takeD(3 * (size_t)p + 8);
    c.o_res = takeD(ntd); c.o_tm = takeD(ntd); c.o_td = takeD(ntd); c.o_robust = takeD(nd);
    c.o_ostart = takeD(np_); c.o_obeta = takeD(np_); c.o_ose = takeD(np_); c.o_oll = takeD(nd);
    c.o_rbinit = takeD(np_); c.o_rbeta = takeD(np_); c.o_rse = takeD(np_);
    c.o_kc = takeD(nd);
    c.dbl = d;
    size_t i = 0;
    auto takeI = [&](size_t k) { size_t off = i; i += align8(k); return off; };
    c.i_iter = takeI(nd); c.i_itacc = takeI(nd); c.i_gflag = takeI(nd); c.i_nz = takeI(nd); c.i_grid = takeI(nd);
    c.i_rep = takeI(nd); c.i_refit = takeI(nd); c.i_cnt = takeI(16); c.i_wc = takeI(64);
    c.i_opt = takeI(nd); c.i_oconv = takeI(nd);
    c.i_lpt = takeI(nd);
    c.ints = i;
    c.bytes = d * sizeof(double) + i * sizeof(int32_t) + 256;
    return c;
}

static int run_chain(const DsqDeseqArgs *a, const DsqDeseqOut *o, hipStream_t st, Pipe &P);
static int run(const DsqDeseqArgs *a, const DsqDeseqOut *o, hipStream_t st) {
    Pipe P;
    memset(&P, 0, sizeof P);
    const int rc = run_chain(a, o, st, P);
    const int rl = rc ? DSQ_OK : launch_pending_ll(P, false);      // (no refit in this analysis: behind everything else)
    const int rj = join_side(P);             // (whatever path the chain left by: nothing stays in flight beside `st`)
    return rc ? rc : (rl ? rl : rj);
}
static int run_chain(const DsqDeseqArgs *a, const DsqDeseqOut *o, hipStream_t st, Pipe &P) {
    if (!a || !o) return capi_fail(DSQ_ERR_ARG, "NULL args/out");
    if (a->n < 1 || a->m < 2 || a->p < 1 || a->m <= a->p) return capi_fail(DSQ_ERR_ARG, "bad dimensions n=%d m=%d p=%d", a->n, a->m, a->p);
    if (a->p > DSQ_P_WIDE) return capi_fail(DSQ_ERR_UNSUPPORTED, "dsq_deseq_dev: p=%d > %d design columns", a->p, DSQ_P_WIDE);
    if (a->betaPrior) {
        if (a->test != 0) return capi_fail(DSQ_ERR_ARG, "betaPrior: Wald test only (R/core.R:1787)");
        if (!a->x_prior || a->p_prior < 1 || !o->mle_beta) return capi_fail(DSQ_ERR_ARG, "betaPrior needs x_prior / p_prior / mle_beta");
        if (a->p_prior > DSQ_P_WIDE) return capi_fail(DSQ_ERR_UNSUPPORTED, "dsq_deseq_dev: %d columns in the prior pass > %d", a->p_prior, DSQ_P_WIDE);
        if ((a->phases & DSQ_PH_PRIOR) && !a->lambda_prior) return capi_fail(DSQ_ERR_ARG, "DSQ_PH_PRIOR needs lambda_prior");
        if ((a->phases & (DSQ_PH_OUTLIERS | DSQ_PH_OUTLIERS_REFIT)) && a->do_replace && !a->lambda_prior) return capi_fail(DSQ_ERR_ARG, "betaPrior: the outlier refit needs lambda_prior");
    }
    if (a->ld < a->m) return capi_fail(DSQ_ERR_ARG, "ld < m");
    // estimateDispersionsPriorVar's branch for 1..3 residual degrees of freedom matches a seeded Monte-Carlo sample
    // (R/core.R:1155-1190, R's RNG + loess): not reproducible here, so the prior variance is not computed at all
    if ((a->phases & DSQ_PH_TREND) && a->m - a->p <= 3 && !(a->dispPriorVar_in > 0.0))
        return capi_fail(DSQ_ERR_UNSUPPORTED, "dsq_deseq_dev: %d residual degrees of freedom: the prior variance of R/core.R:1155-1190 (seeded Monte-Carlo matching) is the caller's (dispPriorVar_in)", a->m - a->p);
    if (!a->y || !a->nf || !a->x || !a->q || !a->a || !a->r || !a->disp_grid || a->ngrid < 2 || !a->lambda) return capi_fail(DSQ_ERR_ARG, "NULL input");
    if (a->trend_mean && (!a->trend_disp || a->n_trend < 1)) return capi_fail(DSQ_ERR_ARG, "trend vectors");
    if (a->useWeights && (!a->weights_raw || !a->weights_norm || !a->weights_floor)) return capi_fail(DSQ_ERR_ARG, "useWeights without weights");
    if (a->test != 0 && a->test != 1) return capi_fail(DSQ_ERR_ARG, "test must be 0 (Wald) or 1 (LRT)");
    // the chain's nbinomLogLike reads K' (the mu-independent part of the row's log density) from the fitBeta launch in
    // front of it, and that launch forms K' inside its IRLS set-up: a chain without IRLS iterations has no K'
    if (a->betaMaxit < 1) return capi_fail(DSQ_ERR_ARG, "dsq_deseq_dev: betaMaxit must be >= 1 (DESeq() itself runs maxit = 100)");
    if (a->fitType < DSQ_FIT_PARAMETRIC || a->fitType > DSQ_FIT_PARAMETRIC_OR_MEAN) return capi_fail(DSQ_ERR_ARG, "fitType must be one of DSQ_FIT_*");
    if (a->dispFit_in && a->trend_mean && !a->trend_fit_in) return capi_fail(DSQ_ERR_ARG, "dispFit_in with gathered trend vectors needs trend_fit_in");
    if (a->dispFit_in && (a->phases & DSQ_PH_OUTLIERS) && a->do_replace)
        return capi_fail(DSQ_ERR_ARG, "dispFit_in: the refit of replaced rows needs the trend at their NEW means -- split the phase (DSQ_PH_OUTLIERS_DETECT, update dispFit_in at the replaced rows, DSQ_PH_OUTLIERS_REFIT) or run with do_replace = 0");
    if ((a->phases & DSQ_PH_OUTLIERS) && (a->phases & (DSQ_PH_OUTLIERS_DETECT | DSQ_PH_OUTLIERS_REFIT)))
        return capi_fail(DSQ_ERR_ARG, "DSQ_PH_OUTLIERS is both halves: do not combine it with DSQ_PH_OUTLIERS_DETECT / _REFIT");
    if (a->x_red && (a->test != 1 || !a->q_red || !a->a_red || !a->r_red || a->p_red < 1 || a->p_red >= a->p))
        return capi_fail(DSQ_ERR_ARG, "reduced model: LRT only, with its QR factors and 1 <= p_red < p");
    if (!o->baseMean || !o->baseVar || !o->allZero || !o->dispGeneEst || !o->dispGeneIter || !o->dispFit || !o->dispMAP ||
        !o->dispersion || !o->dispIter || !o->dispOutlier || !o->beta || !o->betaSE || !o->betaConv || !o->betaIter ||
        !o->logLike || !o->maxCooks || !o->replace || !o->optim_geneest || !o->optim_test || !o->mu_hat || !o->mu || !o->H ||
        !o->cooks || !o->replaceCounts || !o->status || !o->scalars)
        return capi_fail(DSQ_ERR_ARG, "NULL output");
    if (a->test == 0 && (!o->stat || !o->pvalue)) return capi_fail(DSQ_ERR_ARG, "Wald test needs stat / pvalue outputs");
    if (a->test == 1 && !o->logLikeReduced) return capi_fail(DSQ_ERR_ARG, "LRT needs logLikeReduced");
    if ((a->phases & (DSQ_PH_OUTLIERS | DSQ_PH_OUTLIERS_DETECT | DSQ_PH_OUTLIERS_REFIT | DSQ_PH_FINISH)) && (!a->cell_of || !a->replaceable || a->ncell < 1)) return capi_fail(DSQ_ERR_ARG, "outlier phase needs cell_of / replaceable");
    int rc = capi_check_device();
    if (rc) return rc;

    P.a = a; P.o = o; P.st = st; P.tag = "";
    P.t_tol = a->betaTol; P.t_maxit = a->betaMaxit; P.t_useQR = a->useQR; P.t_minmu = a->minmu; P.ge_floor = a->minmu;
    {
        // (see test_fit) only when the test's fit and the outlier phase are enqueued by this one call; the profiling passes
        // time every launch on one stream; DSQ_OVERLAP=0 switches it off
        static const bool env_on = !(getenv("DSQ_OVERLAP") && atoi(getenv("DSQ_OVERLAP")) == 0);
        P.overlap = env_on && !capi_prof_on() && !a->betaPrior && (a->phases & DSQ_PH_MAP_TEST) && (a->phases & DSQ_PH_OUTLIERS);
        if (P.overlap && capi_side_stream(st, &P.side, &P.ev_fork, &P.ev_join) != DSQ_OK) P.overlap = false;
    }
    const int n = P.n = a->n, m = P.m = a->m, p = P.p = a->p;
    P.ld = a->ld;
    P.maxDisp = m > 10 ? (double)m : 10.0;
    P.min_log_alpha = a->min_log_alpha;
    // ---- workspace carve (caller-owned: the row lists and counters persist between the phases of an analysis)
    const int nt_cap = a->n_trend > n ? a->n_trend : n;      // (n_trend is the capacity even in the phases without a trend)
    const int pk = P.pk = kern_width(p);
    const int pkp = a->betaPrior ? kern_width(a->p_prior) : 0;
    const int pmax = pkp > pk ? pkp : pk;                                     // columns of the n x . work matrices
    Carve cv = carve(n, pmax, nt_cap);
    if (!a->workspace || a->workspace_bytes < (int64_t)cv.bytes)
        return capi_fail(DSQ_ERR_ARG, "workspace of %lld bytes, dsq_deseq_workspace_bytes() asks for %zu",
                         (long long)a->workspace_bytes, cv.bytes);
    double *D = (double *)a->workspace;
    int32_t *I = (int32_t *)(D + cv.dbl);
    P.roughDisp = D + cv.o_rough; P.beta_init = D + cv.o_binit; P.alpha_init = D + cv.o_ainit; P.la0 = D + cv.o_la0;
    P.la_out = D + cv.o_laout; P.last_change = D + cv.o_lchg; P.initial_lp = D + cv.o_ilp; P.initial_dlp = D + cv.o_idlp;
    P.last_lp = D + cv.o_llp; P.last_dlp = D + cv.o_ldlp; P.la_grid = D + cv.o_lagrid; P.log_dfit = D + cv.o_ldfit;
    P.la_init = D + cv.o_lainit; P.beta_nat = D + cv.o_bnat; P.beta_var = D + cv.o_bvar; P.beta_iter = D + cv.o_biter;
    P.cnum = D + cv.o_cnum; P.cden = D + cv.o_cden; P.dev = D + cv.o_dev;
    P.lam = D + cv.o_lam; P.contrast = P.lam + pmax; P.lam_prior = P.contrast + pmax; P.xim_dev = P.lam + 3 * (size_t)pmax; P.xim_cur = P.xim_dev;
    P.resbuf = D + cv.o_res; P.trend_mean_c = D + cv.o_tm; P.trend_disp_c = D + cv.o_td; P.robustDisp = D + cv.o_robust;
    P.iter = I + cv.i_iter; P.iter_accept = I + cv.i_itacc; P.grid_flag = I + cv.i_gflag; P.rows_nz = I + cv.i_nz;
    P.rows_grid = I + cv.i_grid; P.rows_rep = I + cv.i_rep; P.rows_refit = I + cv.i_refit; P.counters = o->status;
    P.work_counters = I + cv.i_wc;
    P.opt_start = D + cv.o_ostart; P.opt_beta = D + cv.o_obeta; P.opt_se = D + cv.o_ose; P.opt_ll = D + cv.o_oll;
    P.rows_opt = I + cv.i_opt; P.opt_conv = I + cv.i_oconv;
    P.rows_lpt = I + cv.i_lpt;
    P.red_binit = D + cv.o_rbinit; P.red_beta = D + cv.o_rbeta; P.red_se = D + cv.o_rse;
    P.kconst = D + cv.o_kc;
    {
        size_t slab_d = 0, cscr_d = 0;
        dispatch_beta_scratch(pk, n, m, a->useWeights, &slab_d, &cscr_d);
        if (a->x_red || a->betaPrior) {
            size_t s2 = 0, c2 = 0;
            dispatch_beta_scratch(a->betaPrior ? pkp : kern_width(a->p_red), n, m, a->useWeights, &s2, &c2);
            if (s2 > slab_d) slab_d = s2;
            if (c2 > cscr_d) cscr_d = c2;
        }
        void *b;
        rc = capi_ws_get(DSQ_WS_PIPE_SCRATCH, (slab_d + cscr_d) * sizeof(double) + 64, &b);
        if (rc) return rc;
        P.scratch = (double *)b; P.cscratch = (double *)b + slab_d;
    }
    // ONE launch for every fill this call needs in front of its first kernel (chain_init_kernel): the dynamic-scheduling
    // counters of the fit launches of THIS call, the ridge (R/fitNbinomGLMs.R:73,162) / default contrast (R/wrappers.R:105-108)
    // / prior block, and -- gene-wise phase -- the NA patterns of the result columns and the status block
    static thread_local InitParams ip;
    ip.nseg = 0; ip.nblk = 0;
    auto fill_words = [&](void *p_, size_t bytes, uint32_t val) {
        if (!p_ || !bytes) return;
        // neighbouring regions with the same pattern are one segment
        if (ip.nseg && ip.seg[ip.nseg - 1].val == val && (char *)ip.seg[ip.nseg - 1].p + 4 * (size_t)ip.seg[ip.nseg - 1].words == (char *)p_ &&
            (size_t)ip.seg[ip.nseg - 1].words + bytes / 4 < 0xFFFFFFFFull) { ip.seg[ip.nseg - 1].words += (uint32_t)(bytes / 4); return; }
        ip.seg[ip.nseg++] = {(uint32_t *)p_, (uint32_t)(bytes / 4), val};
    };
    fill_words(P.work_counters, 64 * sizeof(int32_t), 0u);
    for (int c = 0; c < pmax; c++) {
        ip.blk[c] = c < p ? a->lambda[c] : (c < pk ? 1.0 : 0.0);          // (ridge 1 on the padding of a wide design)
        ip.blk[pmax + c] = (c == 0) ? 1.0 : 0.0;
        ip.blk[2 * pmax + c] = (a->betaPrior && a->lambda_prior) ? (c < a->p_prior ? a->lambda_prior[c] : (c < pkp ? 1.0 : 0.0)) : 0.0;
        if (a->x_red) ip.blk[2 * pmax + c] = c < a->p_red ? a->lambda[c] : (c < kern_width(a->p_red) ? 1.0 : 0.0);
    }
    ip.blk_dst = P.lam; ip.nblk = 3 * pmax;
    const bool with_gene_est = (a->phases & DSQ_PH_GENE_EST) != 0;
    if (with_gene_est) {
        // results of rows that turn out all-zero stay NA: 0xFF bytes are a NaN / -1.  The outputs of a caller usually sit
        // side by side (packed blocks): sorted by address, neighbours with the same pattern merge
        struct Fill { char *p; size_t bytes; uint32_t val; };
        std::vector<Fill> fills;
        auto fill = [&](void *p_, size_t bytes, uint32_t val) { if (p_ && bytes) fills.push_back({(char *)p_, bytes, val}); };
        fill(o->status, DSQ_ST_COUNT * sizeof(int32_t), 0u);
        for (double *v : {o->dispGeneEst, o->dispFit, o->dispMAP, o->dispersion, o->betaIter, o->logLike, o->maxCooks, o->logLikeReduced})
            fill(v, (size_t)n * sizeof(double), 0xFFFFFFFFu);
        for (double *v : {o->beta, o->betaSE, o->stat, o->pvalue})
            fill(v, (size_t)n * (a->betaPrior ? a->p_prior : p) * sizeof(double), 0xFFFFFFFFu);
        if (a->betaPrior) fill(o->mle_beta, (size_t)n * p * sizeof(double), 0xFFFFFFFFu);
        for (int32_t *v : {o->dispGeneIter, o->dispIter, o->dispOutlier, o->betaConv}) fill(v, (size_t)n * sizeof(int32_t), 0xFFFFFFFFu);
        for (int32_t *v : {o->replace, o->optim_geneest, o->optim_test, P.grid_flag}) fill(v, (size_t)n * sizeof(int32_t), 0u);
        std::sort(fills.begin(), fills.end(), [](const Fill &x, const Fill &y) { return x.p < y.p; });
        for (const Fill &f : fills) {
            if (ip.nseg >= 23) return capi_fail(DSQ_ERR_ARG, "dsq_deseq_dev: the result columns lie in more than 23 separate regions");
            fill_words(f.p, f.bytes, f.val);
        }
    }
    if (a->phases & DSQ_PH_TREND) fill_words(o->scalars + DSQ_SC_FIT_USED, sizeof(double), 0u);     // 0.0 = DSQ_FIT_PARAMETRIC
    void *tws_zeroed = nullptr;          // the trend fit's barrier / partial-sum block, zeroed by the same launch
    if ((a->phases & DSQ_PH_TREND) && !a->dispFit_in && a->fitType != DSQ_FIT_MEAN) {
        rc = capi_ws_get(DSQ_WS_PIPE_META, trend_fit_workspace_bytes() + 64, &tws_zeroed);
        if (rc) return rc;
        fill_words(tws_zeroed, (trend_fit_workspace_bytes() + 3) / 4 * 4, 0u);
    }
    void *sws_zeroed = nullptr;          // ... and the selection workspace of the sixteen-workgroup prior variance
    if (a->phases & DSQ_PH_TREND) {
        static const int one_block = getenv("DSQ_PRIOR_VAR_ONE_BLOCK") ? atoi(getenv("DSQ_PRIOR_VAR_ONE_BLOCK")) : 0;
        const int nt_ = a->trend_mean ? a->n_trend : n;
        if (!(one_block || nt_ < 16384)) {     // (6 250 genes: 0.116 ms on one workgroup, 0.146 on sixteen; 50 000: 0.243 / 0.124)
            rc = capi_ws_get(DSQ_WS_PIPE_SEL, prior_var_workspace_bytes() + 64, &sws_zeroed);
            if (rc) return rc;
            fill_words(sws_zeroed, (prior_var_workspace_bytes() + 3) / 4 * 4, 0u);
        }
    }
    {
        size_t words = 0;
        for (int sgi = 0; sgi < ip.nseg; sgi++) words += ip.seg[sgi].words;
        unsigned blocks = (unsigned)((words + 256 * 8 - 1) / (256 * 8));
        if (blocks < 1) blocks = 1;
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(chain_init_kernel, dim3(blocks), dim3(256), 0, st, ip);
        PIPE_HIP(hipGetLastError());
    }
    P.x_k = a->x; P.padmask = 0;
    if (pk > p) {
        // the design zero-padded to the kernels' width; the padded columns of the start values (the moments kernel writes
        // the true p columns) and of the optim start values are zero
        void *b;
        rc = capi_ws_get(DSQ_WS_PIPE_PADX, (size_t)m * pk * sizeof(double), &b);
        if (rc) return rc;
        PIPE_HIP(hipMemsetAsync(b, 0, (size_t)m * pk * sizeof(double), st));
        PIPE_HIP(hipMemcpyAsync(b, a->x, (size_t)m * p * sizeof(double), hipMemcpyDeviceToDevice, st));
        P.x_k = (const double *)b;
        P.padmask = dsq_low_bits(pk) & ~dsq_low_bits(p);
        PIPE_HIP(hipMemsetAsync(P.beta_init + (size_t)n * p, 0, (size_t)n * (pk - p) * sizeof(double), st));
        PIPE_HIP(hipMemsetAsync(P.opt_start + (size_t)n * p, 0, (size_t)n * (pk - p) * sizeof(double), st));
    }
    P.red_x_k = a->x_red; P.red_pk = a->x_red ? a->p_red : 0; P.red_lam = a->x_red ? P.lam_prior : P.lam;
    if (a->x_red && kern_width(a->p_red) > a->p_red) {
        // the reduced design at its own padded width; the padded columns of its start values are zero (the moments kernel
        // writes the true p_red columns; those of the optim start values are cleared in front of the reduced fit)
        const int pkr = kern_width(a->p_red);
        void *b;
        rc = capi_ws_get(DSQ_WS_PIPE_PADXR, (size_t)m * pkr * sizeof(double), &b);
        if (rc) return rc;
        PIPE_HIP(hipMemsetAsync(b, 0, (size_t)m * pkr * sizeof(double), st));
        PIPE_HIP(hipMemcpyAsync(b, a->x_red, (size_t)m * a->p_red * sizeof(double), hipMemcpyDeviceToDevice, st));
        P.red_x_k = (const double *)b; P.red_pk = pkr;
        PIPE_HIP(hipMemsetAsync(P.red_binit + (size_t)n * a->p_red, 0, (size_t)n * (pkr - a->p_red) * sizeof(double), st));
    }
    P.pri_x_k = a->x_prior; P.pri_pk = a->betaPrior ? a->p_prior : 0;
    if (a->betaPrior && pkp > a->p_prior) {
        void *b;
        rc = capi_ws_get(DSQ_WS_PIPE_PADXR, (size_t)m * pkp * sizeof(double), &b);      // (never beside a reduced model: Wald only)
        if (rc) return rc;
        PIPE_HIP(hipMemsetAsync(b, 0, (size_t)m * pkp * sizeof(double), st));
        PIPE_HIP(hipMemcpyAsync(b, a->x_prior, (size_t)m * a->p_prior * sizeof(double), hipMemcpyDeviceToDevice, st));
        P.pri_x_k = (const double *)b; P.pri_pk = pkp;
    }
    const Rows nz = {P.rows_nz, P.counters + CNT_NZ, n};
    if (a->cell_of && a->ncell > 0)
        P.ncell = capi_upload_cells(a->cell_of, m, DSQ_WS_PIPE_META + 2, st, &P.cell_perm, &P.cell_start);
    if (a->x_red || a->betaPrior) {
        if (a->x_red && a->cell_of_red && a->ncell_red > 0)
            P.red_ncell = capi_upload_cells(a->cell_of_red, m, DSQ_WS_PIPE_META + 3, st, &P.red_cell_perm, &P.red_cell_start);
        void *b;        // the reduced / prior fit's fitted means: read once by its logLik
        rc = capi_ws_get(DSQ_WS_PIPE_META + 4, (size_t)n * P.ld * sizeof(double), &b);
        if (rc) return rc;
        P.red_mu = (double *)b;
    }

    // ================================================================ gene-wise estimates
    if (a->phases & DSQ_PH_GENE_EST) {
        const Rows all = {nullptr, nullptr, n};
        rc = launch_prefit_rows(P, all, a->y);                                   // getBaseMeansAndVariances + moments
        if (rc) return rc;
        hipLaunchKernelGGL(compact_kernel, dim3(1), dim3(1024), 0, st, 0, n, o->allZero, a->force_zero,
                           (const double *)nullptr, (const double *)nullptr, 0.0, P.rows_nz, (double *)nullptr,
                           (double *)nullptr, P.counters + CNT_NZ);
        if (!a->nf_is_vector) {
            // momentsDispEstimate's mean(1 / colMeans(normalizationFactors)) over the rows that are not all zero
            // (R/core.R:2440-2444 on objectNZ): columns summed down the listed genes in gene order
            void *b;
            rc = capi_ws_get(DSQ_WS_PIPE_META + 5, ((size_t)m + 8) * sizeof(double), &b);
            if (rc) return rc;
            PIPE_HIP(launch_xim_rows(a->nf, P.rows_nz, P.counters + CNT_NZ, m, P.ld, (double *)b, P.xim_dev, st));
        }
        rc = gene_est(P, nz, a->y, o->mu_hat, CNT_GRID1, CNT_OPT1, o->optim_geneest);
        if (rc) return rc;
    }
    // ================================================================ dispersion trend + prior variance
    if (a->phases & DSQ_PH_TREND) {
        const double *tm = a->trend_mean ? a->trend_mean : o->baseMean;
        const double *td = a->trend_mean ? a->trend_disp : o->dispGeneEst;
        const int nt = a->trend_mean ? a->n_trend : n;
        if (!with_gene_est) PIPE_HIP(hipMemsetAsync(P.counters + CNT_TREND, 0, sizeof(int32_t), st));      // (else: the status fill)
        hipLaunchKernelGGL(compact_kernel, dim3(1), dim3(1024), 0, st, 1, nt, (int32_t *)nullptr, (const int32_t *)nullptr,
                           tm, td, 100.0 * a->minDisp, (int32_t *)nullptr, P.trend_mean_c, P.trend_disp_c,
                           P.counters + CNT_TREND);
        void *tws;
        rc = capi_ws_get(DSQ_WS_PIPE_META, trend_fit_workspace_bytes() + 64, &tws);
        if (rc) return rc;
        capi_prof_begin("trend_fit", nt, st);
        if (a->dispFit_in) {
            // the caller's trend (fitType "local" evaluated by R, dispersionFunction<-): nothing to fit; the coefficients are NA
            hipLaunchKernelGGL(trend_given_kernel, dim3(1), dim3(1), 0, st, o->scalars, o->status);
        } else {
            if (a->fitType != DSQ_FIT_MEAN)
                PIPE_HIP(launch_trend_fit_dev_zeroed(P.trend_mean_c, P.trend_disp_c, P.counters + CNT_TREND, o->scalars + DSQ_SC_COEF0,
                                                     o->status + DSQ_ST_TREND_STATUS, tws_zeroed, st));
            if (a->fitType != DSQ_FIT_PARAMETRIC)            // R/core.R:894-899 over the same vector, uncompacted
                hipLaunchKernelGGL(trend_mean_kernel, dim3(1), dim3(1024), 0, st, td, nt, a->minDisp, (int)a->fitType, o->scalars, o->status);
        }
        capi_prof_end(st);
        capi_prof_begin("prior_var", nt, st);
        {
            const double *fin = a->dispFit_in ? (a->trend_mean ? a->trend_fit_in : a->dispFit_in) : (const double *)nullptr;
            if (!sws_zeroed)
                hipLaunchKernelGGL(prior_var_kernel, dim3(1), dim3(1024), 0, st, tm, td, nt, a->minDisp, a->expVarLogDisp,
                                   (m > p) ? 1 : 0, P.resbuf, o->scalars, o->status, fin, a->dispPriorVar_in);
            else
                hipLaunchKernelGGL(prior_var_grid_kernel, dim3(kSelBlocks), dim3(1024), 0, st, tm, td, nt, a->minDisp, a->expVarLogDisp,
                                   (m > p) ? 1 : 0, P.resbuf, o->scalars, o->status, fin, a->dispPriorVar_in, (SelWs *)sws_zeroed);
        }
        capi_prof_end(st);
        PIPE_HIP(hipGetLastError());
    }
    // ================================================================ MAP dispersions + test
    bool counters_zeroed = false;
    if (a->phases & DSQ_PH_MAP_TEST) {
        if (!with_gene_est) {            // (else still zero from the status fill: nothing in between counts into them)
            PIPE_HIP(hipMemsetAsync(P.counters + CNT_GRID2, 0, sizeof(int32_t), st));
            PIPE_HIP(hipMemsetAsync(P.counters + CNT_OPT2, 0, sizeof(int32_t), st));
            PIPE_HIP(hipMemsetAsync(P.counters + CNT_OPT3, 0, sizeof(int32_t), st));
        }
        rc = map_est(P, nz, a->y, o->mu_hat, CNT_GRID2);
        if (rc) return rc;
        rc = test_fit(P, nz, a->y, o->mu, o->H, CNT_OPT2);
        if (rc) return rc;
        // (the counters of the outlier phase -- REP .. OPT3R -- are zeroed here when that phase follows in this call and no
        //  beta-prior pass, which counts into OPT3, comes in between)
        counters_zeroed = (a->phases & (DSQ_PH_OUTLIERS | DSQ_PH_OUTLIERS_DETECT)) && !((a->phases & DSQ_PH_PRIOR) && a->betaPrior);
        hipLaunchKernelGGL(na_assay_rows_kernel, ew_grid(n), dim3(256), 0, st, n, m, P.ld, (const int32_t *)o->allZero, o->mu, o->H,
                           counters_zeroed ? P.counters + CNT_REP : (int32_t *)nullptr, counters_zeroed ? (int)(CNT_N - CNT_REP) : 0);
    }
    // ================================================================ betaPrior: the pass with lambda = 1 / betaPriorVar
    if ((a->phases & DSQ_PH_PRIOR) && a->betaPrior) {
        PIPE_HIP(hipMemsetAsync(P.counters + CNT_OPT2, 0, sizeof(int32_t), st));
        rc = prior_fit(P, nz, a->y, CNT_OPT2);
        if (rc) return rc;
    }
    // ================================================================ count outliers
    // (one call: DSQ_PH_OUTLIERS; a caller with its own dispersion trend splits it -- DSQ_PH_OUTLIERS_DETECT up to the moments
    //  of the replaced rows, then, with dispFit_in updated at those rows' new means, DSQ_PH_OUTLIERS_REFIT)
    const bool ph_detect = (a->phases & (DSQ_PH_OUTLIERS | DSQ_PH_OUTLIERS_DETECT)) != 0;
    const bool ph_refit = (a->phases & (DSQ_PH_OUTLIERS | DSQ_PH_OUTLIERS_REFIT)) != 0;
    if (ph_detect || ph_refit) {
        OutlierMeta M;
        rc = outlier_meta(a, m, st, &M);
        if (rc) return rc;
        int32_t *dperm = M.dperm, *din3 = M.din3, *drepl = M.drepl, *dstart = M.dstart;
        const int any3 = M.any3, maxcell = M.maxcell;
      if (ph_detect) {
        if (!counters_zeroed) PIPE_HIP(hipMemsetAsync(P.counters + CNT_REP, 0, (CNT_N - CNT_REP) * sizeof(int32_t), st));      // REP .. OPT3R

        CooksKernelParams ck;
        memset(&ck, 0, sizeof ck);
        ck.n = n; ck.m = m; ck.p = p; ck.ld = P.ld; ck.y = a->y; ck.nf = a->nf; ck.nf_is_vector = a->nf_is_vector;
        ck.mu = o->mu; ck.H = o->H; ck.perm = dperm; ck.cell_start = dstart; ck.in3 = din3; ck.ncell = a->ncell; ck.any3 = any3;
        int cap = 2; while (cap < (any3 ? maxcell : m)) cap <<= 1;
        ck.sortcap = cap;
        ck.cooks = o->cooks; ck.maxCooks = o->maxCooks; ck.robustDisp = P.robustDisp;
        ck.rows = nz.rows; ck.n_dev = nz.n_dev;
        bool ok = true;
        capi_prof_begin("cooks_distance", n, st);
        PIPE_HIP(launch_cooks(ck, st, &ok));
        capi_prof_end(st);
        if (!ok) return capi_fail(DSQ_ERR_UNSUPPORTED, "m=%d samples: a gene row plus its sort buffer exceeds the 160 KiB LDS", m);
        // (before the replacement: allZero still says which rows had no fit -- a row that only BECOMES all zero keeps its assays)
        hipLaunchKernelGGL(na_assay_rows_kernel, ew_grid(n), dim3(256), 0, st, n, m, P.ld, (const int32_t *)o->allZero, o->cooks,
                           (double *)nullptr);
        if (a->do_replace) {
            ReplaceKernelParams rk;
            memset(&rk, 0, sizeof rk);
            rk.n = n; rk.m = m; rk.ld = P.ld; rk.y = a->y; rk.nf = a->nf; rk.nf_is_vector = a->nf_is_vector;
            rk.cooks = o->cooks; rk.cutoff = a->cooksCutoff; rk.trim = a->trim; rk.replaceable = drepl;
            int cap2 = 2; while (cap2 < m) cap2 <<= 1;
            rk.sortcap = cap2;
            rk.newCounts = o->replaceCounts; rk.replace = o->replace;
            rk.rows = nz.rows; rk.n_dev = nz.n_dev;
            capi_prof_begin("replace_outliers", n, st);
            PIPE_HIP(launch_replace(rk, st, &ok));
            capi_prof_end(st);
            if (!ok) return capi_fail(DSQ_ERR_UNSUPPORTED, "m=%d samples: the sort buffer exceeds the 160 KiB LDS", m);
            // rows with a replacement (R/core.R:2488-2490) -> their moments on the new counts (:2491) -> the ones
            // that are still non-zero are refitted (:2496-2500)
            hipLaunchKernelGGL(list_kernel, ew_grid(n), dim3(256), 0, st, nz, (const int32_t *)o->replace, 1, P.rows_rep,
                               P.counters + CNT_REP);
            const Rows rep = {P.rows_rep, P.counters + CNT_REP, n};
            rc = launch_prefit_rows(P, rep, o->replaceCounts);
            if (rc) return rc;
            hipLaunchKernelGGL(list_kernel, ew_grid(n), dim3(256), 0, st, rep, (const int32_t *)o->allZero, 0, P.rows_refit,
                               P.counters + CNT_REFIT);
        }
      }
        if (ph_refit && a->do_replace) {
            const Rows rep = {P.rows_rep, P.counters + CNT_REP, n};
            const Rows rf = {P.rows_refit, P.counters + CNT_REFIT, n};
            // the same chain on the replaced rows; their mu-hat and fitted means go to the (now dead) mu_hat matrix,
            // assays mu / H keep the original fit as in R (the refit runs on a subset object, :2500-2531)
            if ((rc = launch_pending_ll(P, true))) return rc;      // the full-row log likelihoods, beside the refit
            P.tag = ":refit";
            P.t_tol = 1e-8; P.t_maxit = 100; P.t_useQR = 1; P.t_minmu = 0.5; P.ge_floor = 0.5;
            if (!a->nf_is_vector) {
                // momentsDispEstimate of the refitted subset averages the normalization factors over ITS rows
                // (R/core.R:2440-2444 on objectSub, :2500-2509): the second scalar behind the lambda block
                void *b;
                rc = capi_ws_get(DSQ_WS_PIPE_META + 5, ((size_t)m + 8) * sizeof(double), &b);
                if (rc) return rc;
                PIPE_HIP(launch_xim_flagged(a->nf, n, m, P.ld, o->replace, o->allZero, (double *)b, P.xim_dev + 1, st));
                P.xim_cur = P.xim_dev + 1;
            }
            rc = gene_est(P, rf, o->replaceCounts, o->mu_hat, CNT_GRID1R, CNT_OPT1R, o->optim_geneest);
            if (rc) return rc;
            rc = map_est(P, rf, o->replaceCounts, o->mu_hat, CNT_GRID2R);
            if (rc) return rc;
            if ((rc = join_side(P))) return rc;          // the full-row log likelihoods are down before the refit writes its rows'
            rc = test_fit(P, rf, o->replaceCounts, o->mu_hat, nullptr, CNT_OPT2R);
            if (rc) return rc;
            if (a->betaPrior) {
                rc = prior_fit(P, rf, o->replaceCounts, CNT_OPT2R);
                if (rc) return rc;
            }
            if (!a->defer_finish) {
                rc = outlier_finish(P, nz, rep, M, P.counters + CNT_REFIT);
                if (rc) return rc;
            }
        }
    }
    // ================================================================ (sharding callers) the closing steps, with the
    // number of refitted rows over all shards
    if ((a->phases & DSQ_PH_FINISH) && a->do_replace) {
        OutlierMeta M;
        rc = outlier_meta(a, m, st, &M);
        if (rc) return rc;
        const Rows rep = {P.rows_rep, P.counters + CNT_REP, n};
        rc = outlier_finish(P, nz, rep, M, a->n_refit_global ? a->n_refit_global : P.counters + CNT_REFIT);
        if (rc) return rc;
    }
    return DSQ_OK;
}

// (deseq_host.hip: the host-pointer entry drives the same chain from its per-device worker threads, under the call lock)
int pipeline_run(const DsqDeseqArgs *a, const DsqDeseqOut *o, hipStream_t st) { return run(a, o, st); }

}  // namespace dsq

extern "C" int64_t dsq_deseq_workspace_bytes(int32_t n, int32_t m, int32_t p, int32_t n_trend) {
    (void)m;
    if (n < 1 || p < 1) return 0;
    return (int64_t)dsq::carve(n, dsq::kern_width(p), n_trend > n ? n_trend : n).bytes;
}

extern "C" int dsq_deseq_dev(const DsqDeseqArgs *args, const DsqDeseqOut *out, void *stream) {
    std::lock_guard<std::mutex> lk(dsq::capi_mutex());
    dsq::capi_latch_stream((hipStream_t)stream);
    return dsq::run(args, out, (hipStream_t)stream);
}
