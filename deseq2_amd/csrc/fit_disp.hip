// fit_disp.hip -- gfx950 kernels replacing fitDisp (src/DESeq2.cpp:164-277) and
// fitDispGrid (:469-513): Cox-Reid adjusted profile-likelihood dispersion fit.
//
// One wavefront per gene.  Lane l owns samples l, l+64, ...; the gene's row (counts as int32, mu_hat,
// weights) sits in a wave-private LDS slab and the design matrix X in a block-shared slab, so the ~13
// log_posterior + dlog_posterior evaluations per gene re-read LDS, not HBM.  Each evaluation is
//   * one pass over the DISTINCT counts of the gene (lgamma / digamma of v + 1/alpha, weighted by multiplicity;
//     position 0 evaluates lgamma(1/alpha) itself),
//   * one FUSED pass over the samples: r = 1 / (1 + mu alpha) and log(1 + mu alpha) give the likelihood terms AND the
//     Cox-Reid diagonals w = mu r, -(w w), 2 w^3 (DispGene::pass): per-cell sums for factor designs (p >= 4), p(p+1)/2
//     running sums otherwise (one matrix row per sweep through an LDS arena from p = 7),
//   * the p x p algebra (LU with partial pivoting, determinant, inverse, traces): wave-uniform registers up to p = 3,
//     one matrix column per lane (LaneLU, dsq_wave.hpp) from p = 4.
// The Armijo line search itself is wave-uniform scalar control flow.
// design widths from DSQ_DISP_WIDE_MIN up are served by the zero-padded builds of DSQ_WIDE_LIST (p = 16, 24, 32, 48)
#ifndef DSQ_DISP_WIDE_MIN
#define DSQ_DISP_WIDE_MIN 11
#endif
#define DSQ_WIDE_MIN DSQ_DISP_WIDE_MIN
#include "dsq_internal.hpp"
#include <cstdio>
#include <cstdlib>
#include "dsq_math.hpp"
#include "dsq_rows.hpp"
#include "dsq_wave.hpp"

namespace dsq {

// design widths from DSQ_DISP_LANE_MIN up keep the p x p Cox-Reid matrices ONE COLUMN PER LANE (LaneLU, dsq_wave.hpp)
// instead of wave-uniform in every lane's registers: 2 p instead of 2 p^2 VGPRs per matrix, p^2 instead of p^3 work
#ifndef DSQ_DISP_LANE_MIN
#define DSQ_DISP_LANE_MIN 4
#endif
// rows read through L2 (not staged in LDS): the cell sweep issues the loads of this many trips together
#ifndef DSQ_DISP_TRIP_BATCH
#define DSQ_DISP_TRIP_BATCH 4
#endif
// ... and from DSQ_DISP_ROWPASS_MIN up the general (non-cell) pass accumulates one matrix row per sweep over the samples
// (the K p(p+1)/2 running sums of a single sweep no longer fit in registers)
#include "../../include/dsq_arith_spec.h"
#ifndef DSQ_DISP_ROWPASS_MIN
#define DSQ_DISP_ROWPASS_MIN DSQ_SPEC_SERIAL_GRAM_MINP
#endif
typedef double DsqMat1[1][DSQ_P][DSQ_P];
typedef double DsqMat2[2][DSQ_P][DSQ_P];
typedef double DsqMat3[3][DSQ_P][DSQ_P];
typedef double DsqMat[DSQ_P][DSQ_P];

// -DDSQ_WIDE_PROF (make prof): the phases of a gene's fit in shader-clock cycles, per wave, summed over the launch and
// printed by it (dsq_prof.hpp) -- the dynamic attribution PC sampling would give, were it available on this pool
#include "dsq_prof.hpp"
#ifdef DSQ_WIDE_PROF
__device__ unsigned long long disp_prof[DSQ_PROF_SLOTS];
#define DSQ_GPROF(slot) do { const unsigned long long t1_ = clock64(); pacc[slot] += t1_ - pt0; pt0 = t1_; } while (0)
#else
#define DSQ_GPROF(slot)
#endif

template <int P>
struct SymN { static constexpr int value = P * (P + 1) / 2; };

template <int P, bool USE_W, class Rows>
struct DispGene {
    static constexpr bool LANE = (P >= DSQ_DISP_LANE_MIN);
    // K Cox-Reid matrices: wave-uniform B[k][a][b], or lane columns B[k][i] = entry (i, lane)
    template <int K>
    using Bmat = typename std::conditional<LANE, double[K][P], double[K][P][P]>::type;
    Rows r;
    int m, lane;
    double prior_mean, prior_sigmasq, thr;
    bool usePrior, useCR;
    unsigned long long dropmask;  // bit c: design column c is all-zero over the kept rows (:41-43)
    unsigned long long padmask;   // bit c: column c is zero padding of a wide design (WIDE translation unit only)
    int ablate;         // profiling only (DSQ_ABLATE), 0 in production
#ifdef DSQ_WIDE_PROF
    mutable unsigned long long pt0, pacc[DSQ_PROF_SLOTS];
#endif
    // every fitted mean of the gene lies in [0, 1e140): with alpha in [e^-30, e^10] (the search's clamp, :215-224) and
    // on the dispersion grid, 1 + mu alpha is then a normal number far from the ends of the exponent range and its
    // reciprocal takes the scaling-free division (drcp_n: same quotient, 4 instructions less per sample)
    bool mu_ok;
    DSQ_DEV double rcp1(double opm) const { return mu_ok ? drcp_n(opm) : 1.0 / opm; }
    double *arena;      // lane-column builds: K P P doubles of wave-private LDS (general-mode Cox-Reid rows)
    double *wdbuf;      // general mode, entry-per-lane Gram: K m doubles of wave-private LDS (the samples' diagonals)
    // unweighted genes: the distinct count values (ascending) and their multiplicities, in wave-private LDS -- the
    // lgamma / digamma terms of the likelihood depend on a sample only through its count, so they are evaluated once
    // per DISTINCT count (often a handful) instead of once per sample
    const int32_t *dv, *dc;
    int nv;
    bool hist_ok;       // the histogram form of wave_distinct_counts may run: the buffer sits in LDS, or hist_lds is given
    int32_t *hist_lds;  // long rows (the buffer in global memory): m int32 of the wave's LDS for the histogram
    // design cells (block-shared LDS): cperm[k] = sample | cell << 26 at position k of the cell-sorted sequence, cell
    // offsets; C = 0 -> general per-sample Gram
    const int32_t *cperm, *cstart;
    int C;
    // SORTED staging (unweighted staged rows of a factor design): the row sits in LDS in the cell-sorted order itself, so
    // position k of the sweep reads slot k -- no sample indirection; cid[k] = cell of position k, crep[c] = a sample of cell c
    bool sorted;
    const uint8_t *cid;
    const int32_t *crep;
    // lane-column builds: the products x_c[i] x_c[b] of the cell rows (a property of the design, not of the gene), laid
    // out [cell][i][b] in block-shared LDS once per block; nullptr when the table would be too large -- then xcs, the
    // cell rows themselves ([cell][i], ncell p doubles), and the product is formed on the fly
    const double *xxs, *xcs;
    // sort the gene's counts (bitonic network in the wave's LDS slice), keep the first of every run and its length.
    // buf: 2 m int32 -- sorted values in [0, n2), n2 = pow2 >= m (<= 2m), then dv = buf[0..nv), dc = buf[m..m+nv)
    DSQ_DEV void build_distinct(int32_t *buf) {
        if constexpr (USE_W) { dv = dc = nullptr; nv = 0; return; }
#ifdef DSQ_ABLATE_BUILD
        if (ablate & 16) { dv = buf; dc = buf + m; nv = 0; return; }
#endif
        // (the histogram form only in the builds of p >= 4: compiled into the p = 2, 3 kernels it cost them 5 % at m = 100
        //  even when not taken -- registers)
        nv = wave_distinct_counts(buf, m, lane, [&](int k) { return (int32_t)r.y(k); }, (P >= 4) && hist_ok, hist_lds);
        dv = buf; dc = buf + m;
    }
    DSQ_DEV static void lds_sync() { wave_lds_sync(); }

    DSQ_DEV bool keep_row(int j) const {
        if constexpr (USE_W) return r.w(j) > thr;
        return true;
    }

    // x = x.rows(find(wts > weightThreshold)); x = x.cols(find(sum(abs(x)) > 0.0))
    DSQ_DEV void setup_cr() {
        dropmask = padmask;
        if constexpr (USE_W) {
            if (useCR) {
DSQ_UNROLL_P
                for (int c = 0; c < P; c++) {
                    bool any = false;
                    for (int j = lane; j < m; j += 64)
                        if (keep_row(j) && __builtin_fabs(r.x(j, c)) > 0.0) any = true;
                    if (!__any(any)) dropmask |= (1ull << c);
                }
            }
        }
    }

    // ONE pass over the samples of the gene.  f(j, wd, lik): the caller's per-sample work -- when lik, add the sample's
    // likelihood terms to the caller's own running sums (f captures them); when useCR, also return in wd[0..K) the
    // sample's diagonals of the K Cox-Reid matrices X' diag(wd_k) X, which the pass accumulates over the kept rows.
    // The diagonals come from the reciprocal the likelihood needs anyway (w = mu r, r = 1 / (1 + mu alpha)), so a fused
    // pass costs one division and one logarithm per sample.  Order of the samples: their natural order, or -- CELL MODE --
    // the cell-sorted sequence (position k, partial k mod 64, for the caller's sums as well).
    // A dropped column contributes exact zeros; putting 1 on its diagonal in the first matrix leaves det / inverse /
    // traces equal to those of the compacted matrix.
    template <int K, class F>
    DSQ_DEV void pass(F &&f, Bmat<K> &B) const {
        if (C > 0) {
            // CELL MODE: X' diag(wd) X = sum_c S_c x_c x_c', S_c = sum of wd over the kept samples of cell c (lane c keeps
            // S_c; a cell's sums are closed when the sweep leaves it), outer products added serially in cell order.
            // Per sample: K additions instead of K p(p+1)/2 multiply-adds; K C wave reductions instead of K p(p+1)/2.
            // (round 4 tried DEFERRED closes -- the butterfly steps xor 1, 2, 4 when the sweep leaves a cell, the eight group
            //  sums parked in one lane per group, the steps xor 8, 16, 32 once per evaluation for all cells together: the same
            //  bits with ~ 200 of ~ 1 100 VALU instructions per evaluation gone on paper -- and measured it SLOWER: C3 2.95 ->
            //  3.01 ms, C4 12.8 -> 13.8 ms; nine more spilled VGPRs and the scalar branches around the parked registers cost
            //  more than the shuffles saved.  Not in the tree; profiles/r04_c4_experiments.md.)
            double Sl[K], acc[K];
            _Pragma("unroll")
            for (int k = 0; k < K; k++) { Sl[k] = 0.0; acc[k] = 0.0; }
            int cur = 0;
            // PARKED CLOSES (round 5).  A cell's sum is the butterfly over the 64 per-lane partials (steps xor 1, 2, 4, 8,
            // 16, 32).  When the sweep leaves a cell only the steps inside a group of eight lanes run (xor 1, 2, 4; two
            // sums share them as in wave_allreduce_pair) and the eight group sums are parked in the wave's LDS arena,
            // [cell][k][group]; after the sweep lane 8 g + c picks up group g of cell c and the steps xor 8, 16, 32 run ONCE
            // for eight cells at a time -- lane c ends with S_c, where the Gram build reads it.  The same additions on the
            // same operands in the same order per cell (the bits of wave_allreduce), 16 VALU instructions per close
            // instead of ~ 50; the parked state lives in 24 doubles of LDS per cell, not in registers (round 4's
            // register-parked variant lost to its spills, profiles/r04_c4_experiments.md).
            double *park = arena;
            const int pgrp = lane >> 3;
            auto close_cell = [&]() {
                if constexpr (K >= 2) {
                    const bool odd = (lane & 1) != 0;
                    const double keep = odd ? acc[1] : acc[0], send = odd ? acc[0] : acc[1];
                    double v = keep + lane_xor1(send);
                    v = v + lane_xor2(v);
                    v = v + lane_xor4(v);
                    if ((lane & 6) == 0) park[(cur * 3 + (lane & 1)) * 8 + pgrp] = v;      // lanes 8 g (k = 0) and 8 g + 1 (k = 1)
                    acc[0] = 0.0; acc[1] = 0.0;
                }
                if constexpr (K == 1 || K == 3) {
                    double v = acc[K - 1];
                    v = v + lane_xor1(v);
                    v = v + lane_xor2(v);
                    v = v + lane_xor4(v);
                    if ((lane & 7) == 0) park[(cur * 3 + (K - 1)) * 8 + pgrp] = v;
                    acc[K - 1] = 0.0;
                }
            };
            const int tail_lane = (m - 1) & 63;
            // one trip of the sweep: positions k0 .. k0 + 63 (lane l takes k0 + l), count and mean already loaded
            auto trip = [&](int k0, bool valid, int j, int cmy, double yv, double mv) {
                double wd[K];
                _Pragma("unroll")
                for (int k = 0; k < K; k++) wd[k] = 0.0;
                if (valid) {
                    f(j, yv, mv, wd, true);
                    if (!keep_row(j)) {
                        _Pragma("unroll")
                        for (int k = 0; k < K; k++) wd[k] = 0.0;
                    }
                }
                if (useCR) {
                    const int c_lo = __builtin_amdgcn_readfirstlane(cmy);
                    const int c_hi = __builtin_amdgcn_readlane(cmy, (k0 + 64 <= m) ? 63 : tail_lane);
                    // the general form is: for c = c_lo .. c_hi { if (c != cur) { close; cur = c; } acc += (cmy == c) ? wd : 0 }.
                    // A trip inside ONE cell (lanes past the end hold wd = +0.0, what the masked form adds for them) and
                    // a trip across ONE boundary are written out: the same additions in the same order, without the loop
                    if (c_lo == c_hi) {
                        if (c_lo != cur) { close_cell(); cur = c_lo; }
                        _Pragma("unroll")
                        for (int k = 0; k < K; k++) acc[k] += wd[k];
                    } else if (c_hi == c_lo + 1) {
                        if (c_lo != cur) { close_cell(); cur = c_lo; }
                        const bool first = cmy == c_lo, second = cmy == c_hi;
                        _Pragma("unroll")
                        for (int k = 0; k < K; k++) acc[k] += first ? wd[k] : 0.0;
                        close_cell(); cur = c_hi;
                        _Pragma("unroll")
                        for (int k = 0; k < K; k++) acc[k] += second ? wd[k] : 0.0;
                    } else {
                        for (int c = c_lo; c <= c_hi; c++) {
                            if (c != cur) { close_cell(); cur = c; }
                            const bool mine = cmy == c;
                            _Pragma("unroll")
                            for (int k = 0; k < K; k++) acc[k] += mine ? wd[k] : 0.0;
                        }
                    }
                }
            };
            // rows read through L2 (long rows, not staged): the loads of NB trips are issued together, so that a wave pays
            // the memory latency once per NB trips instead of once per trip (two resident waves per SIMD do not hide it)
            constexpr int NB = std::is_same<Rows, RowsGlobal>::value ? DSQ_DISP_TRIP_BATCH : 1;
            for (int k0 = 0; k0 < m; k0 += 64 * NB) {
                int jb[NB], cb[NB];
                bool vb[NB];
                double yb[NB], mb[NB];
                _Pragma("unroll")
                for (int b = 0; b < NB; b++) {
                    const int kk = k0 + 64 * b + lane;
                    const bool valid = kk < m;
                    int j, cmy;
                    if (sorted) { j = valid ? kk : m - 1; cmy = valid ? (int)cid[j] : -1; }
                    else { const int pk = cperm[valid ? kk : m - 1]; j = pk & 0x3ffffff; cmy = valid ? (pk >> 26) : -1; }
                    jb[b] = j; cb[b] = cmy; vb[b] = valid;
                    yb[b] = r.y(j); mb[b] = r.mu(j);
                }
                _Pragma("unroll")
                for (int b = 0; b < NB; b++)
                    if (k0 + 64 * b < m) trip(k0 + 64 * b, vb[b], jb[b], cb[b], yb[b], mb[b]);
            }
            DSQ_GPROF(4);
            if (!useCR) return;
            close_cell();
            wave_lds_sync();
            for (int r8 = 0; r8 < C; r8 += 8) {               // eight cells per round: lane 8 g + j <- group g of cell r8 + j
                const int c = r8 + (lane & 7);
                const int cc = c < C ? c : C - 1;             // (lanes past the last cell: a valid slot, their result is not read)
                _Pragma("unroll")
                for (int k = 0; k < K; k++) {
                    double v = park[(cc * 3 + k) * 8 + pgrp];
                    v = v + lane_xor8(v);
                    double x_, y_;
                    lane_pair16(v, x_, y_); v = x_ + y_;
                    lane_pair32(v, x_, y_); v = x_ + y_;
                    Sl[k] = (lane >= r8 && lane < r8 + 8) ? v : Sl[k];
                }
            }
            wave_lds_sync();                                  // (the next evaluation parks into the same slots)
            DSQ_GPROF(5);
            if constexpr (LANE) {
                // lane b builds column b: entry (i, b) = sum_c (x_c[i] x_c[b]) S_c, cells in order
                const int bl = lane < P ? lane : 0;
                _Pragma("unroll")
                for (int k = 0; k < K; k++)
                    _Pragma("unroll")
                    for (int i = 0; i < P; i++) B[k][i] = 0.0;
                if constexpr (P == 4) {
                    if (xxs) {
                        // ENTRY PER LANE (round 5): lane e = 4 i + b accumulates entry (i, b) of all K matrices over the cells
                        // -- one table read and K multiply-adds per cell instead of P of each -- then lane b picks up its
                        // column: entry (i, b) sits four i lanes to the right in the same row of sixteen (DPP row_shl).  The
                        // same products added in the same cell order.
                        double E[K];
                        _Pragma("unroll")
                        for (int k = 0; k < K; k++) E[k] = 0.0;
                        const int el = lane & 15;
                        for (int c = 0; c < C; c++) {
                            const double xx = xxs[c * 16 + el];
                            _Pragma("unroll")
                            for (int k = 0; k < K; k++) E[k] = E[k] + xx * lane_read(Sl[k], c);
                        }
                        _Pragma("unroll")
                        for (int k = 0; k < K; k++) {
                            B[k][0] = E[k];
                            B[k][1] = dpp_row_shl<4>(E[k]);
                            B[k][2] = dpp_row_shl<8>(E[k]);
                            B[k][3] = dpp_row_shl<12>(E[k]);
                        }
                        if constexpr (USE_W) {
                            _Pragma("unroll")
                            for (int i = 0; i < P; i++)
                                if (lane == i && ((dropmask >> i) & 1ull)) B[0][i] = 1.0;
                        }
                        return;
                    }
                }
                for (int c = 0; c < C; c++) {
                    double sc[K];
                    _Pragma("unroll")
                    for (int k = 0; k < K; k++) sc[k] = lane_read(Sl[k], c);
                    if (xxs) {
                        _Pragma("unroll")
                        for (int i = 0; i < P; i++) {
                            const double xx = xxs[(c * P + i) * P + bl];
                            _Pragma("unroll")
                            for (int k = 0; k < K; k++) B[k][i] = B[k][i] + xx * sc[k];
                        }
                    } else {
                        const double xb = xcs[c * P + bl];
                        _Pragma("unroll")
                        for (int i = 0; i < P; i++) {
                            const double xx = xcs[c * P + i] * xb;
                            _Pragma("unroll")
                            for (int k = 0; k < K; k++) B[k][i] = B[k][i] + xx * sc[k];
                        }
                    }
                }
                if constexpr (USE_W || (P >= DSQ_WIDE_MIN)) {
                    _Pragma("unroll")
                    for (int i = 0; i < P; i++)
                        if (lane == i && ((dropmask >> i) & 1ull)) B[0][i] = 1.0;
                }
            } else {
DSQ_UNROLL_P
                for (int a = 0; a < P; a++)
DSQ_UNROLL_P
                    for (int b = a; b < P; b++)
                        _Pragma("unroll")
                        for (int k = 0; k < K; k++) B[k][a][b] = 0.0;
                for (int c = 0; c < C; c++) {
                    const int j0 = sorted ? crep[c] : (cperm[cstart[c]] & 0x3ffffff);
                    double sc[K];
                    _Pragma("unroll")
                    for (int k = 0; k < K; k++) sc[k] = lane_read(Sl[k], c);
DSQ_UNROLL_P
                    for (int a = 0; a < P; a++) {
                        const double xa = r.x(j0, a);
DSQ_UNROLL_P
                        for (int b = a; b < P; b++) {
                            const double xx = xa * r.x(j0, b);
                            _Pragma("unroll")
                            for (int k = 0; k < K; k++) B[k][a][b] = B[k][a][b] + xx * sc[k];
                        }
                    }
                }
DSQ_UNROLL_P
                for (int a = 0; a < P; a++)
DSQ_UNROLL_P
                    for (int b = a; b < P; b++)
                        _Pragma("unroll")
                        for (int k = 0; k < K; k++) B[k][b][a] = B[k][a][b];
                if constexpr (USE_W || (P >= DSQ_WIDE_MIN)) {
DSQ_UNROLL_P
                    for (int c = 0; c < P; c++)
                        if (dropmask & (1ull << c)) B[0][c][c] = 1.0;
                }
            }
            return;
        }
        if (!useCR) {
            for (int j = lane; j < m; j += 64) {
                double wd[K];
                f(j, r.y(j), r.mu(j), wd, true);
            }
            return;
        }
        if constexpr (P >= DSQ_DISP_ROWPASS_MIN) {
            if (wdbuf) {
                // ENTRY PER LANE: one sweep over the samples leaves the K diagonals of every sample in LDS (zero for a
                // row below the weight threshold), then lane e owns entry e = (a, b), a <= b, of the upper triangle and adds
                // its m terms x_ja (x_jb wd_j) SERIALLY in sample order -- no cross-lane reduction at all (the one-row-per-
                // sweep form below needs K p (p + 1) / 2 butterflies and p sweeps).  The order of these sums is part of
                // the arithmetic spec: the CPU checker (oracle/deseq2_oracle.c: cr_gram) takes them serially under the same
                // condition (disp_serial_gram).
                for (int j = lane; j < m; j += 64) {
                    double wd[K];
                    f(j, r.y(j), r.mu(j), wd, true);
                    const bool keep = keep_row(j);
                    _Pragma("unroll")
                    for (int k = 0; k < K; k++) wdbuf[(size_t)k * m + j] = keep ? wd[k] : 0.0;
                }
                wave_lds_sync();
                constexpr int NE = SymN<P>::value;
                for (int e0 = 0; e0 < NE; e0 += 64) {
                    const int e = e0 + lane;
                    int a = 0, rem = e < NE ? e : 0;
                    while (rem >= P - a) { rem -= P - a; a++; }
                    const int b = a + rem;
                    const bool live = e < NE && !(((dropmask >> a) | (dropmask >> b)) & 1ull);
                    double acc[K];
                    _Pragma("unroll")
                    for (int k = 0; k < K; k++) acc[k] = 0.0;
                    if (live) {
                        for (int j = 0; j < m; j++) {
                            const double xa = r.x(j, a), xb = r.x(j, b);
                            _Pragma("unroll")
                            for (int k = 0; k < K; k++) acc[k] += xa * (xb * wdbuf[(size_t)k * m + j]);
                        }
                    }
                    if (e < NE) {
                        _Pragma("unroll")
                        for (int k = 0; k < K; k++) {
                            arena[(k * P + a) * P + b] = acc[k];
                            arena[(k * P + b) * P + a] = acc[k];
                        }
                    }
                }
                wave_lds_sync();
                const int bl = lane < P ? lane : 0;
                _Pragma("unroll")
                for (int k = 0; k < K; k++)
                    _Pragma("unroll")
                    for (int i = 0; i < P; i++) B[k][i] = arena[(k * P + i) * P + bl];
                wave_lds_sync();
                if constexpr (USE_W || (P >= DSQ_WIDE_MIN)) {
                    _Pragma("unroll")
                    for (int i = 0; i < P; i++)
                        if (lane == i && ((dropmask >> i) & 1ull)) B[0][i] = 1.0;
                }
                return;
            }
            // (long rows: one matrix row per sweep, wave-order sums)
            // K * P(P+1)/2 per-lane running sums do not fit in registers.  One matrix row per pass over the samples
            // instead (a rolled loop over the rows; the diagonals are recomputed per pass, a division per sample, the
            // caller's likelihood terms added in the first pass only).  Same terms, same order per entry, same wave
            // reduction.  The reduced rows go through the wave's LDS arena (K P P doubles), from which every lane then
            // takes its column.
            bool first = true;
            for (int a0 = 0; a0 < P; a0++) {
                if ((dropmask >> a0) & 1ull) continue;              // a dropped / padding column: exact zeros, left out
                double acc[K][P];
                _Pragma("unroll")
                for (int k = 0; k < K; k++)
                    _Pragma("unroll")
                    for (int b = 0; b < P; b++) acc[k][b] = 0.0;
                for (int j = lane; j < m; j += 64) {
                    double wd[K];
                    f(j, r.y(j), r.mu(j), wd, first);
                    if (keep_row(j)) {
                        const double xa = r.x(j, a0);
                        _Pragma("unroll")
                        for (int b = 0; b < P; b++) {
                            const double xb = r.x(j, b);
                            _Pragma("unroll")
                            for (int k = 0; k < K; k++) acc[k][b] += xa * (xb * wd[k]);
                        }
                    }
                }
                {
                    // the K p sums of the row reduced together (wave_allreduce_many: the bits of one butterfly each, a
                    // quarter of the instructions; the entries left of the diagonal come along and are not stored)
                    double red[K * P];
                    _Pragma("unroll")
                    for (int b = 0; b < P; b++)
                        _Pragma("unroll")
                        for (int k = 0; k < K; k++) red[b * K + k] = acc[k][b];
                    wave_allreduce_many(red, lane);
                    _Pragma("unroll")
                    for (int b = 0; b < P; b++) {
                        if (b < a0) continue;                     // (wave-uniform) the lower triangle is the mirror
                        _Pragma("unroll")
                        for (int k = 0; k < K; k++) {
                            const double v = red[b * K + k];
                            if (lane == 0) {
                                arena[(k * P + a0) * P + b] = v;
                                arena[(k * P + b) * P + a0] = v;
                            }
                        }
                    }
                }
                first = false;
            }
            if (first) {                                          // every column dropped: the caller's sums still run
                for (int j = lane; j < m; j += 64) {
                    double wd[K];
                    f(j, r.y(j), r.mu(j), wd, true);
                }
            }
            wave_lds_sync();
            const int bl = lane < P ? lane : 0;
            _Pragma("unroll")
            for (int k = 0; k < K; k++)
                _Pragma("unroll")
                for (int i = 0; i < P; i++) {
                    const bool dropped = (((dropmask >> i) | (dropmask >> bl)) & 1ull) != 0;
                    B[k][i] = dropped ? 0.0 : arena[(k * P + i) * P + bl];
                }
            wave_lds_sync();
            if constexpr (USE_W || (P >= DSQ_WIDE_MIN)) {
                _Pragma("unroll")
                for (int i = 0; i < P; i++)
                    if (lane == i && ((dropmask >> i) & 1ull)) B[0][i] = 1.0;
            }
            return;
        } else {
            constexpr int N = SymN<P>::value;
            double acc[K * N];
DSQ_UNROLL_P
            for (int i = 0; i < K * N; i++) acc[i] = 0.0;
            for (int j = lane; j < m; j += 64) {
                double wd[K];
                f(j, r.y(j), r.mu(j), wd, true);
                if (keep_row(j)) {
                    double xr[P];
DSQ_UNROLL_P
                    for (int c = 0; c < P; c++) xr[c] = r.x(j, c);
                    int idx = 0;
DSQ_UNROLL_P
                    for (int a = 0; a < P; a++)
DSQ_UNROLL_P
                        for (int b = a; b < P; b++) {
DSQ_UNROLL_P
                            for (int k = 0; k < K; k++) acc[k * N + idx] += xr[a] * (xr[b] * wd[k]);
                            idx++;
                        }
                }
            }
            wave_allreduce_many(acc, lane);
            if constexpr (LANE) {
                // the reduced entries are wave-uniform: lane b picks column b
                _Pragma("unroll")
                for (int k = 0; k < K; k++)
                    _Pragma("unroll")
                    for (int i = 0; i < P; i++) {
                        double v = 0.0;
                        _Pragma("unroll")
                        for (int b = 0; b < P; b++) {
                            const int lo = i < b ? i : b, hi = i < b ? b : i;
                            const int idx = lo * P - (lo * (lo - 1)) / 2 + (hi - lo);
                            v = (lane == b) ? acc[k * N + idx] : v;
                        }
                        B[k][i] = v;
                    }
                if constexpr (USE_W) {
                    _Pragma("unroll")
                    for (int i = 0; i < P; i++)
                        if (lane == i && ((dropmask >> i) & 1ull)) B[0][i] = 1.0;
                }
            } else {
DSQ_UNROLL_P
                for (int k = 0; k < K; k++) {
                    int idx = 0;
DSQ_UNROLL_P
                    for (int a = 0; a < P; a++)
DSQ_UNROLL_P
                        for (int b = a; b < P; b++) {
                            B[k][a][b] = acc[k * N + idx];
                            B[k][b][a] = acc[k * N + idx];
                            idx++;
                        }
                }
            }
        }
        if constexpr (!LANE && USE_W) {
DSQ_UNROLL_P
            for (int c = 0; c < P; c++)
                if (dropmask & (1ull << c)) B[0][c][c] = 1.0;
        }
    }

    // det(B0) and trace(B0^-1 B1) of the Cox-Reid term (:46, :85)
    DSQ_DEV void cr_algebra(const Bmat<2> &B, double &detb, double &tr1) const {
#ifdef DSQ_ABLATE_BUILD
        if (ablate & 8) { detb = 1.0; tr1 = 0.0; return; }
#endif
        if constexpr (LANE) {
            LaneLU<P> lu;
            _Pragma("unroll")
            for (int i = 0; i < P; i++) lu.a[i] = B[0][i];
            lu.factor(lane);
            detb = lu.det();
            double Bi[P];
            lu.inverse(Bi, lane);
            tr1 = lane_trace_sym<P>(Bi, B[1]);
        } else {
            LU<P> lu;
DSQ_UNROLL_P
            for (int a = 0; a < P; a++)
DSQ_UNROLL_P
                for (int b = 0; b < P; b++) lu.a[a][b] = B[0][a][b];
            lu.factor();
            detb = lu.det();
            DsqMat Bi;
            lu.inverse(Bi);
            tr1 = trace_sym<P>(Bi, B[1]);
        }
    }

    // log_posterior, src/DESeq2.cpp:31-64.  log(mu + 1/alpha) = log(1 + mu alpha) - log alpha: one logarithm per sample;
    // without weights the lgamma terms run over the distinct counts (:53,55 regrouped; the CPU checker states the same)
    DSQ_DEV double lp(double la) const {
        const double alpha = dexp(la);
        const double an1 = 1.0 / alpha;
        double lg_an1 = 0.0;
        if constexpr (USE_W) lg_an1 = dlgamma(an1);
        double acc = 0.0;
        double cr_term = 0.0;
        {
            Bmat<1> B;
            pass<1>(
                [&](int j, double y, double mu, double(&wd)[1], bool lik) {
                    const double opm = 1.0 + mu * alpha;
                    if (useCR) wd[0] = mu * rcp1(opm);
                    if (lik) {
                        const double l1 = dlog(opm);
                        if constexpr (USE_W) {
                            const double t = dlgamma(y + an1) - lg_an1 - y * (l1 - la) - an1 * l1;
                            acc += r.w(j) * t;
                        } else {
                            acc += -(y * (l1 - la)) - an1 * l1;
                        }
                    }
                },
                B);
            if (useCR) {
                if constexpr (LANE) {
                    LaneLU<P> lu;
                    _Pragma("unroll")
                    for (int i = 0; i < P; i++) lu.a[i] = B[0][i];
                    lu.factor(lane);
                    cr_term = -0.5 * dlog(lu.det());
                } else {
                    LU<P> lu;
DSQ_UNROLL_P
                    for (int a = 0; a < P; a++)
DSQ_UNROLL_P
                        for (int b = 0; b < P; b++) lu.a[a][b] = B[0][a][b];
                    lu.factor();
                    cr_term = -0.5 * dlog(lu.det());
                }
            }
        }
        double ll_part;
        if constexpr (USE_W) {
            ll_part = wave_allreduce(acc);
        } else {
            // distinct counts at positions 1 .. nv; position 0 (lane 0 of the first trip) evaluates lgamma(1/alpha) itself
            double accv = 0.0;
            for (int q0 = 0; q0 <= nv; q0 += 64) {
                const int q = q0 + lane;
                const bool live = (q > 0) && (q <= nv);
                const double lg = dlgamma((live || q == 0) ? (live ? (double)dv[q - 1] : 0.0) + an1 : 20.0);
                if (q0 == 0) lg_an1 = lane_read(lg, 0);
                if (live) accv += (double)dc[q - 1] * (lg - lg_an1);
            }
            double sv = accv, sa = acc;
            wave_allreduce_pair(sv, sa, lane);              // (the bits of two butterflies)
            ll_part = sv + sa;
        }
        double prior_part = 0.0;
        if (usePrior) {
            double d = la - prior_mean;
            prior_part = -0.5 * (d * d) / prior_sigmasq;
        }
        return ll_part + prior_part + cr_term;
    }

    // log_posterior AND dlog_posterior at the same point (src/DESeq2.cpp:31-64, 68-107).  The line
    // search needs both at every accepted point (:233/:246 and :205/:206); they share exp(la), the
    // Cox-Reid Gram matrix and its LU, log(1 + mu alpha), its reciprocal, and -- inside lgamma and
    // digamma of the same argument -- the shift, log(xs) and 1/xs.  Each shared value is produced
    // by the very expression the separate functions use, so lp and dlp keep their bits.
    DSQ_DEV double lp_dlp(double la, bool withPrior, double &dlp_out) const {
        DSQ_GPROF(10);                                     // (the search's own statements since the last evaluation)
        const double alpha = dexp(la);
        const double an1 = 1.0 / alpha;
        const double an2 = 1.0 / (alpha * alpha);
        double lg_an1 = 0.0, dg_an1 = 0.0;
        if constexpr (USE_W) dlgamma_digamma(an1, lg_an1, dg_an1);
        double acc = 0.0, acc2 = 0.0;
        DSQ_GPROF(3);
        double cr_lp = 0.0, cr_dlp = 0.0;
        {
            Bmat<2> B;
            pass<2>(
                [&](int j, double y, double mu, double(&wd)[2], bool lik) {
                    const double ma = mu * alpha;
                    const double opm = 1.0 + ma;
#ifdef DSQ_ABLATE_BUILD      // tuning build only (make ablate): skip one component to price it (tools/kbench.py, DSQ_ABLATE)
                    const double rr = (ablate & 2) ? opm : rcp1(opm);
#else
                    const double rr = rcp1(opm);
#endif
                    {   // (also without the Cox-Reid term, where nothing reads them: three multiplications instead of four selects)
                        const double w0 = mu * rr;
                        wd[0] = w0;
                        wd[1] = -(w0 * w0);
                    }
                    if (lik) {
#ifdef DSQ_ABLATE_BUILD
                        const double l1 = (ablate & 1) ? opm : dlog(opm);
#else
                        const double l1 = dlog(opm);
#endif
                        if constexpr (USE_W) {
                            double lg, dg;
                            dlgamma_digamma(y + an1, lg, dg);
                            const double t = lg - lg_an1 - y * (l1 - la) - an1 * l1;
                            const double t2 = dg_an1 + l1 - ma * rr - dg + y * (alpha * rr);
                            const double w = r.w(j);
                            acc += w * t;
                            acc2 += w * t2;
                        } else {
                            acc += -(y * (l1 - la)) - an1 * l1;
                            acc2 += l1 - ma * rr + y * (alpha * rr);
                        }
                    }
                },
                B);
            DSQ_GPROF(6);
            if (useCR) {
                double detb, tr1;
                cr_algebra(B, detb, tr1);
                cr_lp = -0.5 * dlog(detb);
                double ddetb = detb * tr1;
                cr_dlp = -0.5 * ddetb / detb;
            }
            DSQ_GPROF(7);
        }
        double ll_part, ll_dpart;
        if constexpr (USE_W) {
            double s1 = acc, s2 = acc2;
            wave_allreduce_pair(s1, s2, lane);
            ll_part = s1;
            ll_dpart = an2 * s2;
        } else {
            double accv = 0.0, accv2 = 0.0;
#ifdef DSQ_ABLATE_BUILD
            if (!(ablate & 4))
#endif
            for (int q0 = 0; q0 <= nv; q0 += 64) {        // position 0: lgamma / digamma of 1/alpha itself (see lp)
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
            DSQ_GPROF(8);
#ifdef DSQ_DISP_SINGLE_REDUCTIONS
            double sv = wave_allreduce(accv), sv2 = wave_allreduce(accv2);
            ll_part = sv + wave_allreduce(acc);
            ll_dpart = an2 * (sv2 + wave_allreduce(acc2));
#else
            double red[4] = {accv, accv2, acc, acc2};      // the four sums of an evaluation reduced together: the bits of four
            wave_allreduce_many(red, lane);                 // butterflies (dsq_wave.hpp), less than half their instructions
            ll_part = red[0] + red[2];
            ll_dpart = an2 * (red[1] + red[3]);
#endif
        }
        double prior_part = 0.0, prior_dpart = 0.0;
        if (usePrior) {
            double d = la - prior_mean;
            prior_part = -0.5 * (d * d) / prior_sigmasq;
        }
        if (withPrior) prior_dpart = -1.0 * (la - prior_mean) / prior_sigmasq;
        dlp_out = (ll_dpart + cr_dlp) * alpha + prior_dpart;
        DSQ_GPROF(9);
        return ll_part + prior_part + cr_lp;
    }

    // dlog_posterior, src/DESeq2.cpp:68-107
    DSQ_DEV double dlp(double la, bool withPrior) const {
        const double alpha = dexp(la);
        const double an1 = 1.0 / alpha;
        const double an2 = 1.0 / (alpha * alpha);
        double dg_an1 = 0.0;
        if constexpr (USE_W) dg_an1 = ddigamma(an1);
        double acc = 0.0;
        double cr_term = 0.0;
        {
            Bmat<2> B;
            pass<2>(
                [&](int j, double y, double mu, double(&wd)[2], bool lik) {
                    const double ma = mu * alpha;
                    const double rr = rcp1(1.0 + ma);
                    if (useCR) {
                        const double w0 = mu * rr;
                        wd[0] = w0;
                        wd[1] = -(w0 * w0);
                    }
                    if (lik) {
                        if constexpr (USE_W) {
                            const double t = dg_an1 + dlog(1.0 + ma) - ma * rr - ddigamma(y + an1) + y * (alpha * rr);
                            acc += r.w(j) * t;
                        } else {
                            acc += dlog(1.0 + ma) - ma * rr + y * (alpha * rr);
                        }
                    }
                },
                B);
            if (useCR) {
                double detb, tr1;
                cr_algebra(B, detb, tr1);
                double ddetb = detb * tr1;
                cr_term = -0.5 * ddetb / detb;
            }
        }
        double ll_sum;
        if constexpr (USE_W) {
            ll_sum = wave_allreduce(acc);
        } else {
            double accv = 0.0;
            for (int q0 = 0; q0 <= nv; q0 += 64) {        // position 0: digamma(1/alpha) itself (see lp)
                const int q = q0 + lane;
                const bool live = (q > 0) && (q <= nv);
                const double dg = ddigamma((live || q == 0) ? (live ? (double)dv[q - 1] : 0.0) + an1 : 20.0);
                if (q0 == 0) dg_an1 = lane_read(dg, 0);
                if (live) accv += (double)dc[q - 1] * (dg_an1 - dg);
            }
            double sv = accv, sa = acc;
            wave_allreduce_pair(sv, sa, lane);
            ll_sum = sv + sa;
        }
        double ll_part = an2 * ll_sum;
        double prior_part = 0.0;
        if (withPrior) prior_part = -1.0 * (la - prior_mean) / prior_sigmasq;
        return (ll_part + cr_term) * alpha + prior_part;
    }

    // d2log_posterior, src/DESeq2.cpp:111-158
    DSQ_DEV double d2lp(double la) const {
        const double alpha = dexp(la);
        const double an1 = 1.0 / alpha;
        const double an2 = 1.0 / (alpha * alpha);
        const double an3 = 1.0 / (alpha * (alpha * alpha));
        const double dg_an1 = ddigamma(an1), tg_an1 = dtrigamma(an1);
        double acc1 = 0.0, acc2 = 0.0;
        double cr_term = 0.0;
        {
            Bmat<3> B;
            pass<3>(
                [&](int j, double y, double mu, double(&wd)[3], bool lik) {
                    const double ma = mu * alpha, opm = 1.0 + ma;
                    const double rr = 1.0 / opm;
                    if (useCR) {
                        const double w0 = mu * rr;
                        wd[0] = w0;
                        wd[1] = -(w0 * w0);
                        wd[2] = 2.0 * (w0 * (w0 * w0));
                    }
                    if (lik) {
                        const double mpa = mu + an1;
                        double t1 = dg_an1 + dlog(opm) - ma * rr - ddigamma(y + an1) + y * (1.0 / mpa);
                        double t2 = -1.0 * an2 * tg_an1 + (mu * mu) * alpha * (1.0 / (opm * opm)) +
                                    an2 * dtrigamma(y + an1) + an2 * y * (1.0 / (mpa * mpa));
                        if constexpr (USE_W) {
                            const double w = r.w(j);
                            t1 = w * t1;
                            t2 = w * t2;
                        }
                        acc1 += t1;
                        acc2 += t2;
                    }
                },
                B);
            if (useCR) {
                double detb, tr1, tr2, tr3;
                if constexpr (LANE) {
                    LaneLU<P> lu;
                    _Pragma("unroll")
                    for (int i = 0; i < P; i++) lu.a[i] = B[0][i];
                    lu.factor(lane);
                    detb = lu.det();
                    double Bi[P], M[P];
                    lu.inverse(Bi, lane);
                    tr1 = lane_trace_sym<P>(Bi, B[1]);
                    lane_mat_mul<P>(Bi, B[1], M);
                    tr2 = lane_trace_prod<P>(M, M);
                    tr3 = lane_trace_sym<P>(Bi, B[2]);
                } else {
                    LU<P> lu;
DSQ_UNROLL_P
                    for (int a = 0; a < P; a++)
DSQ_UNROLL_P
                        for (int b = 0; b < P; b++) lu.a[a][b] = B[0][a][b];
                    lu.factor();
                    detb = lu.det();
                    DsqMat Bi, M;
                    lu.inverse(Bi);
                    tr1 = trace_sym<P>(Bi, B[1]);
                    mat_mul<P>(Bi, B[1], M);
                    tr2 = trace_prod<P>(M, M);
                    tr3 = trace_sym<P>(Bi, B[2]);
                }
                double ddetb = detb * tr1;
                double d2detb = detb * (tr1 * tr1 - tr2 + tr3);
                double rr = ddetb / detb;
                cr_term = 0.5 * (rr * rr) - 0.5 * d2detb / detb;
            }
        }
        double s1 = wave_allreduce(acc1), s2 = wave_allreduce(acc2);
        double ll_part = -2.0 * an3 * s1 + an2 * s2;
        double prior_part = usePrior ? -1.0 / prior_sigmasq : 0.0;
        double dlp0 = dlp(la, false);
        return ((ll_part + cr_term) * (alpha * alpha) + dlp0) + prior_part;
    }
};

// ---- staging --------------------------------------------------------------------
// LDS carve (doubles): [ X: p*m ][ per wave slab ]
//   staged slab    : mu m | (w m) | y int32 m | (distinct counts: 2 m int32, unweighted only)
//   unstaged "slab": the distinct-count buffer only (2 m int32, unweighted only); the row itself is re-read through L2
// lane-column builds: per-wave LDS arena through which the general-mode pass hands the reduced Cox-Reid rows to the
// lanes (3 p p doubles: the second-derivative kernel has three matrices)
// (with design cells the sums are per cell and the arena holds only the parked group sums of the cell closes, 192 bytes
// per cell -- at p = 10, m = 2000 the 9.6 KiB of a general-mode arena per block were what kept a second block off the CU)
__host__ __device__ inline size_t disp_arena_doubles(int p, int ncell) {
    if (ncell > 0) return (size_t)24 * ncell;          // cell mode: the parked group sums, [cell][3][8] (DispGene::pass)
    return (p >= DSQ_DISP_ROWPASS_MIN) ? (size_t)3 * p * p : 0;
}

// general mode (no design cells) from DSQ_DISP_ROWPASS_MIN columns up, rows of at most DSQ_DISP_SERIAL_MAXM samples:
// the Cox-Reid Gram sums are taken ONE MATRIX ENTRY PER LANE, serially over the samples (see DispGene::pass); the
// per-sample diagonals go through 3 m doubles of the wave's LDS
// (measured, 20 000 genes: p = 10, m = 200: 6.1 -> 3.6 ms; p = 16: 35.9 -> 20.2; p = 24: 90.8 -> 47.3; p = 10, m = 500:
// 17.6 -> 15.7; but p = 7, m = 500: 5.6 -> 12.9 -- m serial steps on 28 of 64 lanes and 12 KB more LDS per wave lose to
// seven sweeps: narrow designs take it for short rows only)
#ifndef DSQ_DISP_SERIAL_MAXM
#define DSQ_DISP_SERIAL_MAXM DSQ_SPEC_SERIAL_GRAM_MAXM
#endif
#ifndef DSQ_DISP_SERIAL_MAXM_NARROW
#define DSQ_DISP_SERIAL_MAXM_NARROW DSQ_SPEC_SERIAL_GRAM_MAXM_NARROW
#endif
#ifndef DSQ_TUNING_BUILD
// these thresholds choose a summation ORDER: the test suite's CPU checker reads the same header (include/dsq_arith_spec.h)
static_assert(DSQ_DISP_ROWPASS_MIN == DSQ_SPEC_SERIAL_GRAM_MINP && DSQ_DISP_SERIAL_MAXM == DSQ_SPEC_SERIAL_GRAM_MAXM &&
              DSQ_DISP_SERIAL_MAXM_NARROW == DSQ_SPEC_SERIAL_GRAM_MAXM_NARROW,
              "a -D override of the serial-Gram thresholds changes the arithmetic spec: edit include/dsq_arith_spec.h (or build with -DDSQ_TUNING_BUILD and expect the parity tests to fail)");
#endif
__host__ __device__ inline bool disp_serial_gram(int p, int ncell, int m) {
    return p >= DSQ_DISP_ROWPASS_MIN && ncell <= 0 &&
           m <= (p >= DSQ_SPEC_SERIAL_GRAM_WIDE_P ? DSQ_DISP_SERIAL_MAXM : DSQ_DISP_SERIAL_MAXM_NARROW);
}
template <bool USE_W>
__host__ __device__ inline size_t disp_slab_doubles(int m, bool stage, bool serial) {
    const size_t half = ((size_t)m + 1) / 2;                 // m int32
    const size_t dist = USE_W ? 0 : (size_t)m;               // 2 m int32
    return (stage ? (size_t)m * (USE_W ? 2 : 1) + half + dist : dist) + (serial ? (size_t)3 * m : 0);
}

// block-shared design-cell lists (int32: cell_start[DSQ_CMAX + 2] | m entries sample | cell << 26 in cell-sorted order)
// behind everything else
__host__ __device__ inline size_t disp_cell_doubles(int m, int ncell, bool sorted = false) {
    if (ncell <= 0) return 0;
    // sorted staging: cell_start[DSQ_CMAX + 2] | crep[DSQ_CMAX] int32 | m cell bytes
    if (sorted) return ((size_t)(2 * DSQ_CMAX + 2) * 4 + (size_t)m + 7) / 8;
    return ((size_t)m + DSQ_CMAX + 3) / 2;
}
// block-shared table of the cell-row products x_c[i] x_c[b] (lane-column builds, when it is small: 2 KiB -- a large
// table costs a resident block on long-row shapes, measured at C4: 18.5 -> 26.9 ms with an 8 KiB table)
__host__ __device__ inline size_t disp_xx_doubles(int p, int ncell) {
    const size_t k = (size_t)ncell * p * p;
    return (p >= DSQ_DISP_LANE_MIN && ncell > 0 && k <= 256) ? k : 0;
}
// ... else the cell rows themselves (ncell p doubles)
__host__ __device__ inline size_t disp_xc_doubles(int p, int ncell) {
    return (p >= DSQ_DISP_LANE_MIN && ncell > 0 && disp_xx_doubles(p, ncell) == 0) ? (size_t)ncell * p : 0;
}
// sorted staging applies to staged, unweighted rows of a design with cells
template <bool USE_W>
__host__ __device__ inline bool disp_sorted(bool stage, int ncell) { return stage && !USE_W && ncell > 0; }

template <bool USE_W>
__host__ __device__ inline size_t disp_lds_doubles(int m, int p, int waves, int xlds, int ncell) {
    return (xlds ? (size_t)p * m : 0) + (size_t)waves * disp_slab_doubles<USE_W>(m, true, disp_serial_gram(p, ncell, m)) +
           (size_t)waves * disp_arena_doubles(p, ncell);
}

// 1: the line search evaluates the likelihood alone at a proposal and the derivative only after the Armijo test accepts
// it (what src/DESeq2.cpp:225-246 does); 0: both in one fused evaluation per proposal.  More than half of the proposals
// are rejected (C3: 52 % in the gene-wise fit, 61 % in the MAP fit), yet the fused form wins -- the likelihood alone and
// the derivative alone each cost ~0.8 of the fused evaluation, they share the logarithm, the reciprocal and the Cox-Reid
// sweep.  Measured (same bits either way): C3 3.07 ms fused vs 3.44 ms split; C4 18.8 vs 23.3 ms (profiles/r03_ablation.md)
#ifndef DSQ_DISP_SPLIT
#define DSQ_DISP_SPLIT 0
#endif
#ifndef DSQ_DISP_MINW
/* p <= 4: the search kernel needs 177 registers, so 3 waves per SIMD cost ten spills and pay (6.42 -> 6.15 ms
 * at C3); p = 5, 6: 2 waves; wider designs already spill at 512 registers */
#define DSQ_DISP_MINW (DSQ_P <= 4 ? 3 : DSQ_P <= 10 ? 2 : 1)
#endif

// MODE 0: fitDisp line search (all outputs but last_d2lp); MODE 1: fitDispGrid; MODE 2: last_d2lp only,
// evaluated at the log_alpha the MODE-0 launch stored (same stream).  The second derivative needs
// three p x p matrices at once; keeping it out of the search kernel saves that kernel's registers, and
// callers that never read last_d2lp (DESeq() itself does not) skip the launch.
// Long rows (the unstaged search / grid instantiation without weights) keep their distinct-count buffer in global memory
// (disp_global_dv below), so LDS no longer caps their resident waves, and get a register budget for three waves per SIMD
// where the staged instantiations of the same width take two: measured together at C4 (60 000 x 2000, p = 10), fit_disp
// 12.73 -> 12.27 ms (four waves: 12.20; the buffer in global memory at two waves: 13.1 -- either change alone is a loss or
// nothing, profiles/r04_c4_experiments.md).  Same arithmetic, same bits.
#ifndef DSQ_DISP_MINW_LONG
#define DSQ_DISP_MINW_LONG (DSQ_P <= 10 ? 3 : 1)
#endif
template <bool USE_W, bool STAGE, int MODE>
__host__ __device__ constexpr bool disp_global_dv() { return !STAGE && !USE_W && MODE != 2; }
// ... whose histogram of the counts (wave_distinct_counts, round 5) still fits in LDS: m int32 per wave, up to 2 560 samples
// (C4: 8 KB per wave, three blocks of four waves per CU as before); the sort of a 2 000-sample row in 32 registers per lane
// was ~ 10 % of fit_disp<10>
__host__ __device__ inline size_t disp_hist_doubles(int p, int m) { return (p >= 4 && m >= 256 && m <= 2560) ? ((size_t)m + 1) / 2 : 0; }

template <int P, bool USE_W, bool STAGE, int MODE>
__global__ void __launch_bounds__(256, (disp_global_dv<USE_W, STAGE, MODE>() ? DSQ_DISP_MINW_LONG : DSQ_DISP_MINW)) fit_disp_kernel(DispKernelParams kp) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int waves = blockDim.x >> 6;
    const int m = kp.m;
    const int nwork = DSQ_NWORK(kp);
    if (blockIdx.x * waves >= nwork) return;     // (row-listed launches size the grid without knowing the count)

    const double *xs = smem;
    const bool serial_gram = disp_serial_gram(P, kp.ncell, m);
    const size_t slab_d = disp_global_dv<USE_W, STAGE, MODE>() ? disp_hist_doubles(P, m) + (serial_gram ? (size_t)3 * m : 0)
                                                               : disp_slab_doubles<USE_W>(m, STAGE, serial_gram);
    const size_t xoff = (STAGE && kp.xlds) ? (size_t)P * m : 0;
    double *slab = smem + xoff + (size_t)wave * slab_d;
    double *arena = smem + xoff + (size_t)waves * slab_d + (size_t)wave * disp_arena_doubles(P, kp.ncell);
    // design cells: lists in block-shared LDS behind the slabs and arenas
    int32_t *cstart_s = reinterpret_cast<int32_t *>(smem + xoff + (size_t)waves * (slab_d + disp_arena_doubles(P, kp.ncell)));
    int32_t *cperm_s = cstart_s + DSQ_CMAX + 2;
    int32_t *crep_s = cperm_s;                                           // sorted staging: crep[DSQ_CMAX] | cell bytes
    uint8_t *cid_s = reinterpret_cast<uint8_t *>(crep_s + DSQ_CMAX);
    const int C = kp.ncell;      // (the host passes cells only for the design widths that take the cell mode)
    const bool sorted = disp_sorted<USE_W>(STAGE, C);
    if (C > 0) {
        for (int t = threadIdx.x; t <= C; t += blockDim.x) cstart_s[t] = kp.cell_start[t];
        for (int c = 0; c < C; c++) {
            const int s0 = kp.cell_start[c], s1 = kp.cell_start[c + 1];
            if (sorted) {
                if (threadIdx.x == 0) crep_s[c] = kp.cell_perm[s0];
                for (int t = s0 + (int)threadIdx.x; t < s1; t += blockDim.x) cid_s[t] = (uint8_t)c;
            } else {
                for (int t = s0 + (int)threadIdx.x; t < s1; t += blockDim.x) cperm_s[t] = kp.cell_perm[t] | (c << 26);
            }
        }
    }
    if constexpr (STAGE) {
        if (kp.xlds) {
            for (int t = threadIdx.x; t < P * m; t += blockDim.x) smem[t] = kp.x[t];
        } else {
            xs = kp.x;
        }
    }
    const double *xx_s = nullptr;
    if (disp_xx_doubles(P, C) > 0) {
        double *t_ = reinterpret_cast<double *>(cstart_s) + disp_cell_doubles(m, C, sorted);
        for (int t = threadIdx.x; t < C * P * P; t += blockDim.x) {
            const int c = t / (P * P), i = (t / P) % P, b = t % P;
            const int j0 = kp.cell_perm[kp.cell_start[c]];
            t_[t] = kp.x[(size_t)i * m + j0] * kp.x[(size_t)b * m + j0];
        }
        xx_s = t_;
    }
    const double *xc_s = nullptr;
    if (disp_xc_doubles(P, C) > 0) {
        double *t_ = reinterpret_cast<double *>(cstart_s) + disp_cell_doubles(m, C, sorted);
        for (int t = threadIdx.x; t < C * P; t += blockDim.x) {
            const int c = t / P, i = t % P;
            t_[t] = kp.x[(size_t)i * m + kp.cell_perm[kp.cell_start[c]]];
        }
        xc_s = t_;
    }
    if (C > 0 || (STAGE && kp.xlds)) __syncthreads();

#ifdef DSQ_WIDE_PROF
    unsigned long long pacc_k[DSQ_PROF_SLOTS] = {}, pt_gene = clock64();
#endif
    for (int wi = blockIdx.x * waves + wave; wi < nwork; wi = next_gene(kp.work_counter, wi, gridDim.x * waves, lane)) {
        const int g = DSQ_GENE(kp, wi);
        const int32_t *yg = kp.y + (size_t)g * kp.ld;
        const double *mug = kp.mu_hat + (size_t)g * kp.ld;
        const double *wg = USE_W ? kp.weights + (size_t)g * kp.ld : nullptr;
#ifdef DSQ_WIDE_PROF
        pacc_k[11] += clock64() - pt_gene;              // (drawing the next gene)
#endif

        using Rows = typename std::conditional<STAGE, RowsLds, RowsGlobal>::type;
        DispGene<P, USE_W, Rows> G;
#ifdef DSQ_WIDE_PROF
        for (int q_ = 0; q_ < DSQ_PROF_SLOTS; q_++) G.pacc[q_] = pacc_k[q_];       // (the wave's running totals: flushed once, at the end)
        G.pt0 = clock64();
#endif
        int32_t *dist;
        if constexpr (STAGE) {
            double *ms = slab, *ws = slab + (size_t)m;
            int32_t *ys = reinterpret_cast<int32_t *>(slab + (size_t)m * (USE_W ? 2 : 1));
            dist = ys + 2 * (((size_t)m + 1) / 2);
            if (sorted) {
                for (int k = lane; k < m; k += 64) {              // slot k = position k of the cell-sorted sequence
                    const int j = kp.cell_perm[k];
                    ys[k] = yg[j];
                    ms[k] = mug[j];
                }
            } else {
                for (int j = lane; j < m; j += 64) {
                    double mu = mug[j];
                    ys[j] = yg[j];
                    ms[j] = mu;
                    if constexpr (USE_W) ws[j] = wg[j];
                }
            }
            G.r.y_ = ys; G.r.mu_ = ms; G.r.w_ = USE_W ? ws : nullptr; G.r.x_ = xs; G.r.m = m;
        } else {
            if constexpr (disp_global_dv<USE_W, STAGE, MODE>())
                dist = kp.dist_global + (size_t)(blockIdx.x * waves + wave) * 2 * (size_t)m;      // this wave's slot
            else dist = reinterpret_cast<int32_t *>(slab);
            G.r.y_ = yg; G.r.mu_ = mug; G.r.w_ = wg; G.r.x_ = kp.x; G.r.m = m;
        }
        G.m = m; G.lane = lane;
        {
            bool ok = true;
            for (int j = lane; j < m; j += 64) { const double mu = G.r.mu(j); ok = ok && (mu >= 0.0) && (mu < 1e140); }
            // log alpha: the search keeps its proposals in [-30, 10]; its start value and the grid come from the caller
            bool la_ok;
            if constexpr (MODE == 1) la_ok = kp.grid[0] >= -40.0 && kp.grid[kp.ngrid - 1] <= 40.0 && kp.grid[0] <= kp.grid[kp.ngrid - 1];
            else { const double a0 = (MODE == 0) ? kp.log_alpha_in[g] : kp.log_alpha[g]; la_ok = a0 >= -40.0 && a0 <= 40.0; }
            G.mu_ok = __all(ok) && la_ok;
        }
        G.prior_mean = kp.prior_mean[g];
        G.prior_sigmasq = kp.prior_sigmasq_dev ? *kp.prior_sigmasq_dev : kp.prior_sigmasq;
        G.thr = kp.weightThreshold;
        G.usePrior = kp.usePrior != 0;
        G.useCR = kp.useCR != 0;
        G.ablate = kp.ablate;
        G.padmask = kp.padmask;
        G.arena = arena;
        G.wdbuf = serial_gram ? slab + slab_d - (size_t)3 * m : nullptr;
        G.C = C; G.cperm = cperm_s; G.cstart = cstart_s;
        G.sorted = sorted; G.cid = cid_s; G.crep = crep_s; G.xxs = xx_s; G.xcs = xc_s;
        // (the histogram pays from ~ 256 samples: below, the register sort is a few hundred instructions and a row with one
        //  large count would pay both -- measured at m = 100: fit_disp 0.387 -> 0.406 ms with it)
        G.hist_lds = (disp_global_dv<USE_W, STAGE, MODE>() && disp_hist_doubles(P, m) > 0) ? reinterpret_cast<int32_t *>(slab) : nullptr;
        G.hist_ok = (!disp_global_dv<USE_W, STAGE, MODE>() || G.hist_lds != nullptr) && m >= 256;
#ifdef DSQ_WIDE_PROF
        { const unsigned long long t1_ = clock64(); G.pacc[0] += t1_ - G.pt0; G.pt0 = t1_; }
#endif
        G.build_distinct(dist);
#ifdef DSQ_WIDE_PROF
        { const unsigned long long t1_ = clock64(); G.pacc[1] += t1_ - G.pt0; G.pt0 = t1_; }
#endif
        G.setup_cr();
#ifdef DSQ_WIDE_PROF
        { const unsigned long long t1_ = clock64(); G.pacc[2] += t1_ - G.pt0; G.pt0 = t1_; }
#endif

        if constexpr (MODE == 2) {
            double d2 = G.d2lp(kp.log_alpha[g]);
            if (lane == 0) kp.last_d2lp[g] = d2;
        } else if constexpr (MODE == 1) {
            // fitDispGrid, src/DESeq2.cpp:492-510
            const int ng = kp.ngrid;
            const double delta = kp.grid[1] - kp.grid[0];
            int idx = 0;
            double best = 0.0;
            for (int t = 0; t < ng; t++) {
                double v = G.lp(kp.grid[t]);
                if (t == 0 || v > best) { best = v; idx = t; }
            }
            double a_hat = kp.grid[idx];
            double start = a_hat - delta, end = a_hat + delta;
            double step = (end >= start) ? (end - start) / (double)(ng - 1)
                                         : -(start - end) / (double)(ng - 1);
            double afine = start;
            for (int t = 0; t < ng; t++) {
                double a = (t == ng - 1) ? end : start + (double)t * step;
                double v = G.lp(a);
                if (t == 0 || v > best) { best = v; afine = a; }
            }
            if (lane == 0) kp.log_alpha[g] = afine;
        } else {
            // fitDisp, src/DESeq2.cpp:194-266
            const double epsilon = 1.0e-4;
            double a = kp.log_alpha_in[g];
            double dlp;
#if DSQ_DISP_SPLIT
            double lp = G.lp(a);
            dlp = G.dlp(a, G.usePrior);
#else
            double lp = G.lp_dlp(a, G.usePrior, dlp);
#endif
            double kappa = kp.kappa_0;
            const double initial_lp = lp, initial_dlp = dlp;
            double change = -1.0;
            int it = 0, it_acc = 0;
            for (int t = 0; t < kp.maxit; t++) {
                it++;
                double a_propose = a + kappa * dlp;
                if (a_propose < -30.0) kappa = (-30.0 - a) / dlp;
                if (a_propose > 10.0) kappa = (10.0 - a) / dlp;
                const double a_try = a + kappa * dlp;
                // the reference evaluates log_posterior(a + kappa*dlp) for the Armijo test and
                // again after accepting (:225,:233): same argument, same value -> evaluated once;
                // the derivative it asks for after accepting (:246) is at that same point too
                // ... evaluated together with it, or (DSQ_DISP_SPLIT) only after the test accepts and the search goes on;
                // lp() / dlp() / lp_dlp() return the same bits for the same argument
                double dlp_try = 0.0;
#if DSQ_DISP_SPLIT
                const double lp_try = G.lp(a_try);
#else
                const double lp_try = G.lp_dlp(a_try, G.usePrior, dlp_try);
#endif
                double theta_kappa = -1.0 * lp_try;
                double theta_hat_kappa = -1.0 * lp - kappa * epsilon * (dlp * dlp);
                if (kp.force_iters > 0 && t + 1 >= kp.force_iters) break;   // profiling only
#ifdef DSQ_ABLATE_BUILD
                if (kp.force_iters > 0) {       // fixed work per gene: force_iters evaluations at one nearby point, no search logic
                    dlp = dlp_try * 0.0 + 1.0e-3;
                    lp = lp_try * 0.0 + lp;
                    if (!(dlp == dlp)) dlp = 1.0e-3;
                    if (!(lp == lp)) lp = 0.0;
                    continue;
                }
#endif
                if (uniform(theta_kappa <= theta_hat_kappa)) {
                    it_acc++;
                    a = a_try;
                    double lpnew = lp_try;
                    change = lpnew - lp;
                    if (uniform(change < kp.tol)) { lp = lpnew; break; }
                    if (uniform(a < kp.min_log_alpha)) break;
                    lp = lpnew;
#if DSQ_DISP_SPLIT
                    dlp = G.dlp(a, G.usePrior);
#else
                    dlp = dlp_try;
#endif
                    kappa = __builtin_fmin(kappa * 1.1, kp.kappa_0);
                    if (it_acc % 5 == 0) kappa = kappa / 2.0;
                } else {
                    kappa = kappa / 2.0;
                }
            }
            if (lane == 0) {
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
#ifdef DSQ_WIDE_PROF
        { const unsigned long long t1_ = clock64(); G.pacc[10] += t1_ - G.pt0; G.pt0 = t1_; }
        for (int q_ = 0; q_ < DSQ_PROF_SLOTS; q_++) pacc_k[q_] = G.pacc[q_];
        pt_gene = clock64();
#endif
    }
#ifdef DSQ_WIDE_PROF
    if (lane == 0) for (int q_ = 0; q_ < DSQ_PROF_SLOTS; q_++) atomicAdd(&disp_prof[q_], pacc_k[q_]);
#endif
}

// ---- launch ---------------------------------------------------------------------
enum { DSQ_WS_DISP_DIST = 36 };       // a grow-only workspace slot between the call slots and the chain's
template <int P, bool USE_W, int MODE>
static hipError_t launch_disp_p(const DispKernelParams &kp, hipStream_t st) {
    const Tuning &tu = tuning();
    // same choice as fit_beta: maximise resident waves per CU (LDS 160 KiB, registers allow 8 waves)
    const size_t budget = (size_t)tu.disp_lds_kb * 1024, cu_lds = 160 * 1024;
    const int wmax = tu.disp_waves >= 4 ? 4 : tu.disp_waves >= 2 ? 2 : tu.disp_waves == 1 ? 1 : 4;
    int best = -1, best_wpc = 0, waves = wmax, xlds = 0;
    bool stage = false;
    for (int xl = tu.disp_xlds ? 1 : 0; xl >= 0; xl--)
        for (int w = wmax; w >= 1; w >>= 1) {
            size_t need = (disp_lds_doubles<USE_W>(kp.m, P, w, xl, kp.ncell) + disp_cell_doubles(kp.m, kp.ncell, disp_sorted<USE_W>(true, kp.ncell)) +
                           disp_xx_doubles(P, kp.ncell) + disp_xc_doubles(P, kp.ncell)) * sizeof(double);
            if (need > budget) continue;
            int blocks = (int)(cu_lds / need);
            const int wcap = 4 * (DSQ_DISP_MINW);     // waves per CU the register budget of this build admits
            int wpc = w * blocks < wcap ? w * blocks : wcap;
            int score = wpc * 100 + w * 2 + xl;
            if (score > best) { best = score; best_wpc = wpc; stage = true; waves = w; xlds = xl; }
        }
    // long rows: below 6 resident waves per CU the staged kernel loses to L2-resident rows at full occupancy
    // (measured, p = 4: m = 1250 8.4 vs 7.6 ms, m = 2000 12.1 vs 7.7 ms; m = 800 4.4 vs 4.8 ms)
    if (stage && best_wpc < 6 && tu.disp_stage < 0) { stage = false; waves = wmax; }
    if (tu.disp_stage == 0) stage = false;
    const size_t unstaged_wave = (disp_slab_doubles<USE_W>(kp.m, false, disp_serial_gram(P, kp.ncell, kp.m)) +
                                  disp_arena_doubles(P, kp.ncell)) * sizeof(double);
    const size_t cell_bytes = (disp_cell_doubles(kp.m, kp.ncell, disp_sorted<USE_W>(stage, kp.ncell)) + disp_xx_doubles(P, kp.ncell) +
                               disp_xc_doubles(P, kp.ncell)) * sizeof(double);
    const bool gdv = !stage && disp_global_dv<USE_W, false, MODE>();
    const size_t unstaged_lds = gdv ? (disp_hist_doubles(P, kp.m) + (disp_serial_gram(P, kp.ncell, kp.m) ? (size_t)3 * kp.m : 0) + disp_arena_doubles(P, kp.ncell)) * sizeof(double)
                                    : unstaged_wave;
    if (!stage)
        while (waves > 1 && (size_t)waves * unstaged_lds + cell_bytes > budget) waves >>= 1;
    size_t lds = (stage ? disp_lds_doubles<USE_W>(kp.m, P, waves, xlds, kp.ncell) * sizeof(double)
                        : (size_t)waves * unstaged_lds) + cell_bytes;   // unstaged: [distinct-count buffer +] WIDE arena
    DispKernelParams kq = kp;
    kq.xlds = xlds;
    kq.dist_global = nullptr;
    if (kq.work_counter && MODE == 2) kq.work_counter += 1;   // the d2 pass has its own counter
    const void *fn = stage ? (const void *)fit_disp_kernel<P, USE_W, true, MODE> : (const void *)fit_disp_kernel<P, USE_W, false, MODE>;
    static thread_local int bpc_cache[2][8];      // [stage][waves]: the occupancy query costs ~1 ms, ask once
    static thread_local size_t lds_cache[2][8];
    DSQ_CACHE_PER_DEVICE(bpc_cache, lds_cache);
    if (lds_cache[stage][waves] != lds) { bpc_cache[stage][waves] = 0; lds_cache[stage][waves] = lds; }
    int bpc = bpc_cache[stage][waves];
    if (bpc == 0) {
        if (lds > 64 * 1024) (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, fn, 64 * waves, lds) != hipSuccess || bpc < 1) bpc = 1;
        bpc_cache[stage][waves] = bpc;
        if (getenv("DSQ_VERBOSE")) fprintf(stderr, "[dsq] fit_disp<P=%d,mode=%d> waves=%d stage=%d lds=%zu occupancy-api blocks/CU=%d\n", P, MODE, waves, (int)stage, lds, bpc);
    }
    if (tu.disp_bpc > 0) bpc = tu.disp_bpc;
    const int cus = device_cu_count();
    int blocks_needed = (kp.n + waves - 1) / waves;
    int grid = blocks_needed < cus * bpc ? blocks_needed : cus * bpc;
    if (kp.rows_few && grid > cus) grid = cus;        // a row list (stragglers, refits): its length lives on the device
    if (grid < 1) grid = 1;
    if (gdv) {
        void *v = nullptr;       // one slot of 2 m int32 per resident wave, grow-only workspace of the (device, stream)
        if (capi_ws_get(DSQ_WS_DISP_DIST, (size_t)grid * waves * 2 * (size_t)kp.m * sizeof(int32_t), &v) != 0) return hipErrorOutOfMemory;
        kq.dist_global = (int32_t *)v;
    }
#ifdef DSQ_WIDE_PROF
    unsigned long long pz[DSQ_PROF_SLOTS] = {}, ph[DSQ_PROF_SLOTS];
    (void)hipMemcpyToSymbol(HIP_SYMBOL(disp_prof), pz, sizeof(pz));
#endif
    if (stage)
        hipLaunchKernelGGL((fit_disp_kernel<P, USE_W, true, MODE>), dim3(grid), dim3(64 * waves), lds, st, kq);
    else
        hipLaunchKernelGGL((fit_disp_kernel<P, USE_W, false, MODE>), dim3(grid), dim3(64 * waves), lds, st, kq);
#ifdef DSQ_WIDE_PROF
    if (MODE == 0 && kp.n >= 1000) {
        (void)hipStreamSynchronize(st);
        (void)hipMemcpyFromSymbol(ph, HIP_SYMBOL(disp_prof), sizeof(ph));
        double tot = 0;
        for (int q = 0; q < DSQ_PROF_SLOTS; q++) tot += (double)ph[q];
        static const char *nm[12] = {"stage row", "distinct counts", "setup", "eval head (exp, 1/alpha)", "sample sweep (+ closes in it)", "parked closes", "cell sums -> matrices",
                                     "LU / inverse / trace", "lgamma over distinct counts", "reductions + tail", "search statements + results", "next gene"};
        fprintf(stderr, "[disp_prof] p=%d m=%d n=%d cells=%d waves/block=%d:", P, kp.m, kp.n, kp.ncell, waves);
        for (int q = 0; q < 12; q++) fprintf(stderr, " %s %.1f%%", nm[q], 100.0 * (double)ph[q] / (tot > 0 ? tot : 1));
        fprintf(stderr, "  (%.0f Mcycles over all waves)\n", tot / 1e6);
    }
#endif
    return hipGetLastError();
}

// One translation unit per design width: compiled with -DDSQ_P=<p> (csrc/Makefile), so the
// fully unrolled p x p register math of each width builds in parallel.
#ifndef DSQ_P
#error "compile with -DDSQ_P=<number of design columns>"
#endif

// wide designs without cells on short rows: the rolled kernel of fit_disp_wide.hip (one build for every width)
bool fit_disp_rolled_applies(const DispKernelParams &kp, int *p_true);
hipError_t launch_fit_disp_rolled(const DispKernelParams &kp, hipStream_t st, bool grid);

template <>
hipError_t launch_fit_disp_p<DSQ_P>(const DispKernelParams &kp, hipStream_t st, bool grid) {
#if DSQ_P >= 16
    if (fit_disp_rolled_applies(kp, nullptr)) return launch_fit_disp_rolled(kp, st, grid);
#endif
#if DSQ_P > DSQ_DISP_PERWIDTH_MAX
    // (beyond DSQ_DISP_PERWIDTH_MAX columns there is no per-width kernel: rows of more than DSQ_SPEC_SERIAL_GRAM_MAXM samples, or
    //  whose slab does not fit the LDS, are refused -- capi.hip says so before the launch)
    return hipErrorNotSupported;
#else
    if (grid)
        return kp.useWeights ? launch_disp_p<DSQ_P, true, 1>(kp, st) : launch_disp_p<DSQ_P, false, 1>(kp, st);
    hipError_t e = kp.useWeights ? launch_disp_p<DSQ_P, true, 0>(kp, st) : launch_disp_p<DSQ_P, false, 0>(kp, st);
    if (e != hipSuccess || !kp.last_d2lp) return e;
    return kp.useWeights ? launch_disp_p<DSQ_P, true, 2>(kp, st) : launch_disp_p<DSQ_P, false, 2>(kp, st);
#endif
}

}  // namespace dsq
