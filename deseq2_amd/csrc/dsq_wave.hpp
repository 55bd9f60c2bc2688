// dsq_wave.hpp -- wavefront-level building blocks: one gfx950 wavefront (64 lanes)
// owns one gene.  Lane l holds samples l, l+64, l+128, ...; every sum over samples is
// "lane-serial then xor-butterfly", which is the summation order the arithmetic spec
// fixes (DESIGN.md "Arithmetic").  The p x p matrices are either wave-uniform in the
// registers of every lane (LU<P>, narrow designs) or held one column per lane (LaneLU<P>);
// p is a template parameter, and MFMA is pointless at p <= 24.
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>

// The per-width kernels (one translation unit per design width, -DDSQ_P=p, p <= 10) want every p-loop of the
// wave-uniform algebra fully unrolled so that the p x p state stays in registers.  The WIDE translation units
// (-DDSQ_P=16, 24, 32, 48, serving 11 <= p <= 48 on zero-padded designs) keep the loops of the general fitBeta kernel rolled,
// with its work arrays in an LDS arena; everything built on LaneLU is unrolled at every width (2 p registers per matrix).
#ifndef DSQ_WIDE_MIN
#define DSQ_WIDE_MIN 11
#endif
#if defined(DSQ_P) && DSQ_P >= DSQ_WIDE_MIN
#define DSQ_UNROLL_P _Pragma("nounroll")
#else
#define DSQ_UNROLL_P _Pragma("unroll")
#endif

#include <utility>

namespace dsq {

#define DSQ_DEV __device__ __forceinline__
// the wave-uniform p x p algebra below uses nothing of the device: it also compiles for the host, where the probe of
// tools/lu_probe.hip runs the SAME template as the reference for what the device computed
#define DSQ_HD __host__ __device__ __forceinline__

// ---- cross-lane exchange without LDS ------------------------------------------------------------
// __shfl_xor compiles to ds_bpermute_b32 (an LDS-crossbar round trip per 32-bit half).  The butterfly
// partners lane ^ 1, 2, 4, 8 are reachable with DPP modifiers inside a row of 16 lanes and lane ^ 16, 32
// with gfx950's v_permlane16_swap / v_permlane32_swap, all plain VALU moves.  Same partners, same
// additions (a + b is commutative in IEEE), so the sums keep their bits; a dependent chain of
// all-reduces runs 2.2x faster (tools/dppbench.hip) -- it matters in fitBeta's Householder replay,
// which serialises ~14 reductions per IRLS iteration.
template <int CTRL>
DSQ_DEV double dpp_mov(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
DSQ_DEV double lane_xor1(double v) { return dpp_mov<0xB1>(v); }    // quad_perm [1,0,3,2]
DSQ_DEV double lane_xor2(double v) { return dpp_mov<0x4E>(v); }    // quad_perm [2,3,0,1]
DSQ_DEV double lane_xor8(double v) { return dpp_mov<0x128>(v); }   // row_ror:8
DSQ_DEV double lane_xor4(double v) {                               // row_shl:4 into banks 0,2; row_shr:4 into 1,3
    int lo = __double2loint(v), hi = __double2hiint(v);
    int l2 = __builtin_amdgcn_update_dpp(lo, lo, 0x104, 0xF, 0x5, false);
    l2 = __builtin_amdgcn_update_dpp(l2, lo, 0x114, 0xF, 0xA, false);
    int h2 = __builtin_amdgcn_update_dpp(hi, hi, 0x104, 0xF, 0x5, false);
    h2 = __builtin_amdgcn_update_dpp(h2, hi, 0x114, 0xF, 0xA, false);
    return __hiloint2double(h2, l2);
}
// lane l <- lane l + N of the same row of sixteen (N = 1 .. 15; lanes whose source lies outside the row keep their own value)
template <int N>
DSQ_DEV double dpp_row_shl(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x100 + N, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x100 + N, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
typedef unsigned dsq_u2 __attribute__((ext_vector_type(2)));
// (self, partner) across rows: swap(v, v) leaves [row0,row0,row2,row2] and [row1,row1,row3,row3]
DSQ_DEV void lane_pair16(double v, double &a, double &b) {
    unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    dsq_u2 pl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    dsq_u2 ph = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    a = __hiloint2double((int)ph[0], (int)pl[0]);
    b = __hiloint2double((int)ph[1], (int)pl[1]);
}
DSQ_DEV void lane_pair32(double v, double &a, double &b) {
    unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    dsq_u2 pl = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    dsq_u2 ph = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    a = __hiloint2double((int)ph[0], (int)pl[0]);
    b = __hiloint2double((int)ph[1], (int)pl[1]);
}

// lane l <- lane l - 1 across the whole wave (DPP wave_shr:1, a gfx9 control); lane 0 keeps its own value
DSQ_DEV double wave_shr1(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x138, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x138, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}

// all-reduce: every lane ends with the same bits (a+b is commutative in IEEE).
DSQ_DEV double wave_allreduce(double v) {
    v = v + lane_xor1(v);
    v = v + lane_xor2(v);
    v = v + lane_xor4(v);
    v = v + lane_xor8(v);
    double a, b;
    lane_pair16(v, a, b); v = a + b;
    lane_pair32(v, a, b); v = a + b;
    return v;
}

// value of lane `src` (a compile-time-known or wave-uniform lane) in every lane: v_readlane, no LDS
DSQ_DEV double lane_read(double v, int src) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}

// all-reduce of TWO values at the price of little more than one: the first butterfly step hands the odd lanes the b sums
// and the even lanes the a sums (each lane adds its own half to the partner's matching half -- the additions
// a[l] + a[l^1] and b[l] + b[l^1] of the plain butterfly, each done once instead of twice), the remaining five steps
// run on the one combined register (the partners l ^ 2 .. l ^ 32 have the parity of l).  Every even lane ends with the
// bits wave_allreduce(a) gives, every odd lane with those of wave_allreduce(b); both are then read back wave-uniformly.
DSQ_DEV void wave_allreduce_pair(double &a, double &b, int lane) {
    const bool odd = (lane & 1) != 0;
    const double keep = odd ? b : a, send = odd ? a : b;
    double v = keep + lane_xor1(send);
    v = v + lane_xor2(v);
    v = v + lane_xor4(v);
    v = v + lane_xor8(v);
    double x, y;
    lane_pair16(v, x, y); v = x + y;
    lane_pair32(v, x, y); v = x + y;
    a = lane_read(v, 0);
    b = lane_read(v, 1);
}

// all-reduce of a value that is +0.0 outside lanes [0, nlive) (nlive wave-uniform): when the live lanes sit in the first
// row(s) of 16 the cross-row steps only ever add +0.0, which is done here without the permlane swaps.  Lanes < nlive end
// with the same bits as wave_allreduce gives (x + 0.0 keeps the -0.0 -> +0.0 behaviour of the skipped additions); the
// other lanes do NOT hold the total.
DSQ_DEV double wave_allreduce_low(double v, int nlive) {
    v = v + lane_xor1(v);
    v = v + lane_xor2(v);
    v = v + lane_xor4(v);
    v = v + lane_xor8(v);
    double a, b;
    if (nlive > 16) { lane_pair16(v, a, b); v = a + b; } else v = v + 0.0;
    if (nlive > 32) { lane_pair32(v, a, b); v = a + b; } else v = v + 0.0;
    return v;
}

// compile-time loop index (a stage whose reductions have a width that depends on the stage)
template <class F, int... K>
DSQ_DEV void static_for_impl(F &&f, std::integer_sequence<int, K...>) { (f(std::integral_constant<int, K>{}), ...); }
template <int N, class F>
DSQ_DEV void static_for(F &&f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// All-reduce of N values for little more than N additions (round 4).  The butterfly of wave_allreduce adds, at the step
// with partner l ^ b, own + partner in BOTH lanes of a pair -- the same sum twice.  Here the lane with bit b clear keeps
// the sum of value 2i and its partner the sum of value 2i + 1 (each sends the half it does not keep), so every step
// halves the number of live registers; after the four DPP steps value j of a block of 16 sits in the lanes whose low
// four bits are j, the two cross-row steps run on what is left (one register per 16 values) and the totals are read
// back wave-uniformly.  Same partners, same operands of every addition (a + b is commutative in IEEE): the bits of
// wave_allreduce, ~ 8 N + 20 instructions instead of ~ 35 N.
template <int BIT>
DSQ_DEV double lane_xor_bit(double v) {
    if constexpr (BIT == 1) return lane_xor1(v);
    else if constexpr (BIT == 2) return lane_xor2(v);
    else if constexpr (BIT == 4) return lane_xor4(v);
    else return lane_xor8(v);
}
template <int N, int BIT>
DSQ_DEV void wave_merge_level(const double (&in)[N], double (&out)[(N + 1) / 2], int lane) {
    const bool hi = (lane & BIT) != 0;
    _Pragma("unroll")
    for (int i = 0; i < N / 2; i++) {
        const double keep = hi ? in[2 * i + 1] : in[2 * i], send = hi ? in[2 * i] : in[2 * i + 1];
        out[i] = keep + lane_xor_bit<BIT>(send);
    }
    if constexpr (N & 1) out[N / 2] = in[N - 1] + lane_xor_bit<BIT>(in[N - 1]);
}
template <int N>
DSQ_DEV void wave_allreduce_many(double (&v)[N], int lane) {
    if constexpr (N == 1) { v[0] = wave_allreduce(v[0]); }
    else {
        constexpr int N1 = (N + 1) / 2, N2 = (N1 + 1) / 2, N3 = (N2 + 1) / 2, N4 = (N3 + 1) / 2;
        double r1[N1], r2[N2], r3[N3], r4[N4];
        wave_merge_level<N, 1>(v, r1, lane);
        wave_merge_level<N1, 2>(r1, r2, lane);
        wave_merge_level<N2, 4>(r2, r3, lane);
        wave_merge_level<N3, 8>(r3, r4, lane);
        _Pragma("unroll")
        for (int i = 0; i < N4; i++) {
            double a, b;
            lane_pair16(r4[i], a, b); r4[i] = a + b;
            lane_pair32(r4[i], a, b); r4[i] = a + b;
        }
        _Pragma("unroll")
        for (int j = 0; j < N; j++) v[j] = lane_read(r4[j >> 4], j & 15);
    }
}

// ... of N values that are +0.0 outside lanes [0, nlive) (wave_allreduce_low's case): the four DPP steps merged as above,
// the cross-row steps done only when live lanes sit beyond the first row(s); the totals are read from the first row,
// whose 16 lanes hold them whatever nlive is.
template <int N>
DSQ_DEV void wave_allreduce_many_low(double (&v)[N], int lane, int nlive) {
    if constexpr (N == 1) { v[0] = lane_read(wave_allreduce_low(v[0], nlive), 0); }
    else {
        constexpr int N1 = (N + 1) / 2, N2 = (N1 + 1) / 2, N3 = (N2 + 1) / 2, N4 = (N3 + 1) / 2;
        double r1[N1], r2[N2], r3[N3], r4[N4];
        wave_merge_level<N, 1>(v, r1, lane);
        wave_merge_level<N1, 2>(r1, r2, lane);
        wave_merge_level<N2, 4>(r2, r3, lane);
        wave_merge_level<N3, 8>(r3, r4, lane);
        _Pragma("unroll")
        for (int i = 0; i < N4; i++) {
            double a, b;
            if (nlive > 16) { lane_pair16(r4[i], a, b); r4[i] = a + b; } else r4[i] = r4[i] + 0.0;
            if (nlive > 32) { lane_pair32(r4[i], a, b); r4[i] = a + b; } else r4[i] = r4[i] + 0.0;
        }
        _Pragma("unroll")
        for (int j = 0; j < N; j++) v[j] = lane_read(r4[j >> 4], j & 15);
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

// widest LU<P> whose solve() swaps the right-hand side with select chains (LU_SELECT below; see solve).  The three forms
// are equal -- tools/lu_probe.hip: bit for bit against the host build at P = 4, 5, 6, 10, on the device and under the
// sanitizers.  Round 2 saw the select forms give wrong results inside ONE kernel of that time, fit_disp<6> with weights at
// the 256-VGPR cap; round 4 reproduced it on that commit and traced it to the toolchain's SGPR spilling into VGPR lanes
// (the same source is right with -mllvm -amdgpu-spill-sgpr-to-vgpr=false; profiles/r04_lu_solve.md).  So this is a
// register-pressure choice, not a fence: the select chain where it removes scratch traffic (the narrow widths), the
// conditional swap where the wave-uniform matrices already fill the register file.
#ifndef DSQ_LU_SELECT_MAXP
#define DSQ_LU_SELECT_MAXP 4
#endif
enum { LU_SWAP = 0, LU_SELECT = 1, LU_PAIRSEL = 2 };
// ---- P x P LU with partial pivoting (first maximum wins), reciprocal pivots -----
template <int P, int FORM = (P <= DSQ_LU_SELECT_MAXP ? LU_SELECT : LU_SWAP)>
struct LU {
    double a[P][P];
    double rdiag[P];
    int piv[P];
    int sign;

    DSQ_HD void factor() {
        sign = 1;
DSQ_UNROLL_P
        for (int k = 0; k < P; k++) {
            int pr = k;
            double best = __builtin_fabs(a[k][k]);
DSQ_UNROLL_P
            for (int i = k + 1; i < P; i++) {
                double v = __builtin_fabs(a[i][k]);
                if (v > best) { best = v; pr = i; }
            }
            piv[k] = pr;
            if (pr != k) {
                sign = -sign;
DSQ_UNROLL_P
                for (int i = k + 1; i < P; i++) {
                    if (i == pr) {
DSQ_UNROLL_P
                        for (int j = 0; j < P; j++) { double t = a[k][j]; a[k][j] = a[i][j]; a[i][j] = t; }
                    }
                }
            }
            double rinv = 1.0 / a[k][k];
            rdiag[k] = rinv;
DSQ_UNROLL_P
            for (int i = k + 1; i < P; i++) {
                double l = a[i][k] * rinv;
                a[i][k] = l;
DSQ_UNROLL_P
                for (int j = k + 1; j < P; j++) a[i][j] = __builtin_fma(-l, a[k][j], a[i][j]);
            }
        }
    }
    DSQ_HD double det() const {
        double d = a[0][0];
DSQ_UNROLL_P
        for (int i = 1; i < P; i++) d = d * a[i][i];
        return sign < 0 ? -d : d;
    }
    DSQ_HD void solve(double (&b)[P]) const {
DSQ_UNROLL_P
        for (int k = 0; k < P; k++) {
            int pr = piv[k];
            if (pr != k) {
                if constexpr (FORM == LU_SELECT) {
                    // b[k] <-> b[pr] as select chains: written as a conditional swap the compiler turns it into a
                    // dynamically indexed access, which moves b[] (a register array otherwise) into scratch memory
                    // (fit_disp<4>: 48 B/lane of scratch and ~90 scratch accesses per evaluation, 209 -> 177 VGPRs
                    // without it).  Only for the narrow widths (see DSQ_LU_SELECT_MAXP above).
                    const double bk = b[k];
                    double picked = bk;
DSQ_UNROLL_P
                    for (int i = k + 1; i < P; i++) picked = (i == pr) ? b[i] : picked;
DSQ_UNROLL_P
                    for (int i = k + 1; i < P; i++) b[i] = (i == pr) ? bk : b[i];
                    b[k] = picked;
                } else if constexpr (FORM == LU_PAIRSEL) {
                    // (the form LaneLU::solve uses at every width)
DSQ_UNROLL_P
                    for (int i = k + 1; i < P; i++) {
                        const bool sw = (i == pr);
                        const double bi = b[i], bk = b[k];
                        b[i] = sw ? bk : bi;
                        b[k] = sw ? bi : bk;
                    }
                } else {
DSQ_UNROLL_P
                    for (int i = k + 1; i < P; i++) {
                        if (i == pr) { double t = b[k]; b[k] = b[i]; b[i] = t; }
                    }
                }
            }
        }
DSQ_UNROLL_P
        for (int i = 0; i < P; i++) {
            double t = b[i];
DSQ_UNROLL_P
            for (int j = 0; j < i; j++) t = __builtin_fma(-a[i][j], b[j], t);
            b[i] = t;
        }
DSQ_UNROLL_P
        for (int i = P - 1; i >= 0; i--) {
            double t = b[i];
DSQ_UNROLL_P
            for (int j = i + 1; j < P; j++) t = __builtin_fma(-a[i][j], b[j], t);
            b[i] = t * rdiag[i];
        }
    }
    DSQ_HD void inverse(double (&inv)[P][P]) const {
DSQ_UNROLL_P
        for (int c = 0; c < P; c++) {
            double col[P];
DSQ_UNROLL_P
            for (int i = 0; i < P; i++) col[i] = (i == c) ? 1.0 : 0.0;
            solve(col);
DSQ_UNROLL_P
            for (int i = 0; i < P; i++) inv[i][c] = col[i];
        }
    }
};

// ---- P x P algebra with ONE COLUMN PER LANE -------------------------------------------------------------------------
// Lane j (< P) holds column j of a matrix in P registers (a[i] = A[i][j]); lanes >= P idle.  A wave-uniform P x P matrix
// costs 2 P^2 VGPRs in every lane (512-register kernels that spill from P = 7 up); a lane-column matrix costs 2 P.  The
// factorisation broadcasts one multiplier at a time (v_readlane), every lane updates its own column: the SAME operations
// on the same values as LU<P> above (first maximum wins, reciprocal pivots, fma(-l, u, a)), so the results keep their
// bits; a solve runs P right-hand sides at once (lane c owns right-hand side c), which turns the P^3 inverse into P^2.
template <int P>
struct LaneLU {
    double a[P];
    double rdiag[P];   // wave-uniform
    int piv[P];        // wave-uniform
    int sign;

    DSQ_DEV void factor(int lane) {
        sign = 1;
#pragma unroll
        for (int k = 0; k < P; k++) {
            // pivot search: lane k scans its own column
            int pr = k;
            double best = __builtin_fabs(a[k]);
#pragma unroll
            for (int i = k + 1; i < P; i++) {
                double v = __builtin_fabs(a[i]);
                if (v > best) { best = v; pr = i; }
            }
            pr = __builtin_amdgcn_readlane(pr, k);
            piv[k] = pr;
            if (pr != k) {
                sign = -sign;
                // rows k <-> pr in every column, as selects on the wave-uniform pr (a conditional swap would make
                // a[] a dynamically indexed array, i.e. scratch memory)
#pragma unroll
                for (int i = k + 1; i < P; i++) {
                    const bool sw = (i == pr);
                    const double ai = a[i], ak = a[k];
                    a[i] = sw ? ak : ai;
                    a[k] = sw ? ai : ak;
                }
            }
            const double rinv = 1.0 / lane_read(a[k], k);
            rdiag[k] = rinv;
#pragma unroll
            for (int i = k + 1; i < P; i++) {
                const double l = lane_read(a[i], k) * rinv;
                const double upd = __builtin_fma(-l, a[k], a[i]);
                a[i] = (lane == k) ? l : (lane > k ? upd : a[i]);
            }
        }
    }
    DSQ_DEV double det() const {
        double d = lane_read(a[0], 0);
#pragma unroll
        for (int i = 1; i < P; i++) d = d * lane_read(a[i], i);
        return sign < 0 ? -d : d;
    }
    // lane c: b = right-hand side c on entry, solution c on return
    DSQ_DEV void solve(double (&b)[P]) const {
#pragma unroll
        for (int k = 0; k < P; k++) {
            const int pr = piv[k];
            if (pr != k) {
#pragma unroll
                for (int i = k + 1; i < P; i++) {
                    const bool sw = (i == pr);
                    const double bi = b[i], bk = b[k];
                    b[i] = sw ? bk : bi;
                    b[k] = sw ? bi : bk;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < P; i++) {
            double t = b[i];
#pragma unroll
            for (int j = 0; j < i; j++) t = __builtin_fma(-lane_read(a[i], j), b[j], t);
            b[i] = t;
        }
#pragma unroll
        for (int i = P - 1; i >= 0; i--) {
            double t = b[i];
#pragma unroll
            for (int j = i + 1; j < P; j++) t = __builtin_fma(-lane_read(a[i], j), b[j], t);
            b[i] = t * rdiag[i];
        }
    }
    // lane c: column c of the inverse
    DSQ_DEV void inverse(double (&inv)[P], int lane) const {
#pragma unroll
        for (int i = 0; i < P; i++) inv[i] = (i == lane) ? 1.0 : 0.0;
        solve(inv);
    }
};

// C = A B, all three in lane columns (lane j: c[i] = sum_k fma(A[i][k], B[k][j]), k ascending as in mat_mul)
template <int P>
DSQ_DEV void lane_mat_mul(const double (&a)[P], const double (&b)[P], double (&c)[P]) {
#pragma unroll
    for (int i = 0; i < P; i++) {
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < P; k++) acc = __builtin_fma(lane_read(a[i], k), b[k], acc);
        c[i] = acc;
    }
}

// trace(A B) for a SYMMETRIC B: sum over the columns k (ascending) of t_k = sum_i fma(A[i][k], B[i][k]) (i ascending)
template <int P>
DSQ_DEV double lane_trace_sym(const double (&a)[P], const double (&b)[P]) {
    double t = 0.0;
#pragma unroll
    for (int i = 0; i < P; i++) t = __builtin_fma(a[i], b[i], t);
    double tr = 0.0;
#pragma unroll
    for (int k = 0; k < P; k++) tr = tr + lane_read(t, k);
    return tr;
}

// trace(A B), general: the fma chain over (i, k) of A[i][k] B[k][i] of trace_prod
template <int P>
DSQ_DEV double lane_trace_prod(const double (&a)[P], const double (&b)[P]) {
    double acc = 0.0;
#pragma unroll
    for (int i = 0; i < P; i++)
#pragma unroll
        for (int k = 0; k < P; k++) acc = __builtin_fma(lane_read(a[i], k), lane_read(b[k], i), acc);
    return acc;
}

template <int P>
DSQ_HD void mat_mul(const double (&a)[P][P], const double (&b)[P][P], double (&c)[P][P]) {
DSQ_UNROLL_P
    for (int i = 0; i < P; i++)
DSQ_UNROLL_P
        for (int j = 0; j < P; j++) {
            double acc = 0.0;
DSQ_UNROLL_P
            for (int k = 0; k < P; k++) acc = __builtin_fma(a[i][k], b[k][j], acc);
            c[i][j] = acc;
        }
}

template <int P>
DSQ_HD double trace_prod(const double (&a)[P][P], const double (&b)[P][P]) {
    double acc = 0.0;
DSQ_UNROLL_P
    for (int i = 0; i < P; i++)
DSQ_UNROLL_P
        for (int k = 0; k < P; k++) acc = __builtin_fma(a[i][k], b[k][i], acc);
    return acc;
}

// trace(A B) for a SYMMETRIC B (wave-uniform matrices): the same sums as lane_trace_sym
template <int P>
DSQ_HD double trace_sym(const double (&a)[P][P], const double (&b)[P][P]) {
    double tr = 0.0;
DSQ_UNROLL_P
    for (int k = 0; k < P; k++) {
        double t = 0.0;
DSQ_UNROLL_P
        for (int i = 0; i < P; i++) t = __builtin_fma(a[i][k], b[i][k], t);
        tr = tr + t;
    }
    return tr;
}

// ---- distinct counts of a gene ---------------------------------------------------------------------------------------
// wave-private LDS ordering point (one wave owns the slab: no workgroup barrier needed)
DSQ_DEV void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- bitonic networks on REGISTERS -------------------------------------------------------------------------------
// Element e = lane * R + r sits in register r of lane `lane` (64 R elements).  Compare-exchanges at a distance below R
// pair two registers of one lane, the others pair the same register of two lanes (xor-shuffle over DPP / permlane
// swaps): the comparators of the LDS networks (outlier.hip: wave_sort; wave_distinct_counts below) on the same element
// indices, so even the intermediate states are the same -- without an LDS round trip and a wave fence per stage.
DSQ_DEV double lane_xor_any(double v, int d, int lane) {
    double a, b;
    switch (d) {
    case 1: return lane_xor1(v);
    case 2: return lane_xor2(v);
    case 4: return lane_xor4(v);
    case 8: return lane_xor8(v);
    case 16: lane_pair16(v, a, b); return (lane & 16) ? a : b;
    default: lane_pair32(v, a, b); return (lane & 32) ? a : b;
    }
}
DSQ_DEV int32_t lane_xor_any(int32_t v, int d, int lane) {
    switch (d) {
    case 1: return __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true);
    case 2: return __builtin_amdgcn_mov_dpp(v, 0x4E, 0xF, 0xF, true);
    case 4: {
        int l2 = __builtin_amdgcn_update_dpp(v, v, 0x104, 0xF, 0x5, false);
        return __builtin_amdgcn_update_dpp(l2, v, 0x114, 0xF, 0xA, false);
    }
    case 8: return __builtin_amdgcn_mov_dpp(v, 0x128, 0xF, 0xF, true);
    case 16: {
        dsq_u2 p = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
        return (int32_t)((lane & 16) ? p[0] : p[1]);
    }
    default: {
        dsq_u2 p = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
        return (int32_t)((lane & 32) ? p[0] : p[1]);
    }
    }
}

// stages j = keff/2 .. 1 of the level with block size keff (keff = 64 R: the final merge level, every block ascending).
// Fully unrolled: levels, stages and registers are compile-time constants after inlining (a rolled form with the stage
// as a run-time value was measured 3x slower: the shuffle distance has to be a constant).
template <class T, int R>
DSQ_DEV void sort_level_regs(T (&v)[R], int lane, int keff) {
    _Pragma("unroll")
    for (int j = 64 * R / 2; j > 0; j >>= 1) {
        if (j >= keff) continue;
        if (j < R) {
            _Pragma("unroll")
            for (int r = 0; r < R; r++) {
                if ((r & j) != 0) continue;
                const int r2 = r | j;
                const bool asc = (keff < R) ? ((r & keff) == 0) : ((lane & (keff / R)) == 0);
                const T a = v[r], c = v[r2];
                const bool sw = (a > c) == asc;
                v[r] = sw ? c : a;
                v[r2] = sw ? a : c;
            }
        } else {
            const int d = j / R;
            const bool lower = (lane & d) == 0;
            const bool asc = (lane & (keff / R)) == 0;
            _Pragma("unroll")
            for (int r = 0; r < R; r++) {
                const T own = v[r];
                const T oth = lane_xor_any(own, d, lane);
                const T a = lower ? own : oth, c = lower ? oth : own;
                const bool sw = (a > c) == asc;
                v[r] = sw ? oth : own;
            }
        }
    }
}
template <class T, int R>
DSQ_DEV void wave_sort_regs(T (&v)[R], int lane) {
    _Pragma("unroll")
    for (int k = 2; k <= 64 * R; k <<= 1) sort_level_regs<T, R>(v, lane, k);
}
// ascending sort of a BITONIC sequence (first falling, then rising): the last level is enough
template <class T, int R>
DSQ_DEV void wave_merge_regs(T (&v)[R], int lane) { sort_level_regs<T, R>(v, lane, 64 * R); }

// Sort the gene's m counts (bitonic network in the wave's LDS slice buf), keep the first of every run and its length.
// buf: 2 m int32 -- sorted values in [0, n2), n2 = pow2 >= m (<= 2m); on return the distinct values (ascending) are
// buf[0..nv), their multiplicities buf[m..m+nv); returns nv.  yfun(k) = count of sample k as int32.
// hist (round 5; buf in LDS): when every count of the row is below m, a HISTOGRAM replaces the sort -- buf[m + y] counts the
// samples with count y (LDS atomic adds), then one ordered compaction over y = 0 .. m-1 leaves the same ascending distinct
// values and multiplicities the sort + run-length pass produce (~ 160 instead of ~ 1 300 VALU instructions at m = 500; the
// compaction writes slot rank <= y behind its own reads).  A row with a count >= m takes the sort as before.
// hist_lds: m int32 of wave-private LDS for the histogram when buf itself is not in LDS (long rows: their distinct-count
// buffer lives in global memory); nullptr: the histogram sits in buf[m .. 2 m).
template <class F>
DSQ_DEV int wave_distinct_counts(int32_t *buf, int m, int lane, F &&yfun, bool hist = false, int32_t *hist_lds = nullptr) {
    if (hist) {
        int32_t *h = hist_lds ? hist_lds : buf + m;
        wave_lds_sync();
        for (int t = lane; t < m; t += 64) h[t] = 0;
        wave_lds_sync();
        bool big = false;
        for (int k = lane; k < m; k += 64) {
            const int32_t y = yfun(k);
            if (y >= 0 && y < m) atomicAdd(&h[y], 1);
            else big = true;
        }
        wave_lds_sync();
        if (!__any(big)) {
            int base = 0;
            for (int v0 = 0; v0 < m; v0 += 64) {
                const int v = v0 + lane;
                const int32_t c = v < m ? h[v] : 0;
                const bool head = c > 0;
                const unsigned long long mask = __ballot(head);
                const int rank = base + __popcll(mask & ((1ull << lane) - 1ull));
                wave_lds_sync();                              // every lane has read before any lane writes
                if (head) { buf[rank] = v; buf[m + rank] = c; }
                base += __popcll(mask);
                wave_lds_sync();
            }
            return base;
        }
    }
    int n2 = 2;
    while (n2 < m) n2 <<= 1;
    wave_lds_sync();
    // rows of up to 2048 samples: the sort in registers (element lane * R + r), the sorted values then go to buf
    auto in_regs = [&](auto rtag) {
        constexpr int R = decltype(rtag)::value;
        int32_t w[R];
        _Pragma("unroll")
        for (int r = 0; r < R; r++) {
            const int e = lane * R + r;
            w[r] = e < m ? yfun(e) : 0x7fffffff;
        }
        wave_sort_regs<int32_t, R>(w, lane);
        _Pragma("unroll")
        for (int r = 0; r < R; r++)
            if (lane * R + r < n2) buf[lane * R + r] = w[r];
        wave_lds_sync();
    };
    bool sorted_in_regs = true;
    if (n2 <= 64) in_regs(std::integral_constant<int, 1>());
    else if (n2 == 128) in_regs(std::integral_constant<int, 2>());
    else if (n2 == 256) in_regs(std::integral_constant<int, 4>());
    else if (n2 == 512) in_regs(std::integral_constant<int, 8>());
    else if (n2 == 1024) in_regs(std::integral_constant<int, 16>());
    else if (n2 == 2048) in_regs(std::integral_constant<int, 32>());
    else sorted_in_regs = false;                 // (longer rows: the LDS network below)
    if (!sorted_in_regs) {
    for (int k = lane; k < n2; k += 64) buf[k] = k < m ? yfun(k) : 0x7fffffff;
    wave_lds_sync();
    for (int kk = 2; kk <= n2; kk <<= 1)
        for (int jj = kk >> 1; jj > 0; jj >>= 1) {
            for (int t = lane; t < (n2 >> 1); t += 64) {
                int lo = ((t & ~(jj - 1)) << 1) | (t & (jj - 1));
                int hi = lo | jj;
                bool asc = (lo & kk) == 0;
                int32_t a = buf[lo], c = buf[hi];
                if ((a > c) == asc) { buf[lo] = c; buf[hi] = a; }
            }
            wave_lds_sync();
        }
    }
    int base = 0;
    for (int k0 = 0; k0 < m; k0 += 64) {
        const int k = k0 + lane;
        const bool valid = k < m;
        const int32_t v = valid ? buf[k] : 0;
        const int32_t prev = (valid && k > 0) ? buf[k - 1] : -1;
        const bool head = valid && (k == 0 || v != prev);
        const unsigned long long mask = __ballot(head);
        const int rank = base + __popcll(mask & ((1ull << lane) - 1ull));
        wave_lds_sync();                                  // every lane has read before any lane writes
        if (head) { buf[rank] = v; buf[m + rank] = k; }
        base += __popcll(mask);
        wave_lds_sync();
    }
    const int nv = base;
    for (int i0 = 0; i0 < nv; i0 += 64) {
        const int i = i0 + lane;
        const bool valid = i < nv;
        const int s0 = valid ? buf[m + i] : 0;
        const int s1 = valid ? ((i + 1 < nv) ? buf[m + i + 1] : m) : 0;
        wave_lds_sync();
        if (valid) buf[m + i] = s1 - s0;
        wave_lds_sync();
    }
    return nv;
}

// Lane-strided sweep over the samples j = lane, lane + 64, ... < m of a row that is read through L1 / L2: `load(j, b)`
// issues the loads of sample j into register slot b, `use(j, b)` consumes them -- in ascending j per lane, so every
// per-lane running sum sees its terms in the usual order -- with the loads of NB trips in flight together.  A wave that
// issues one trip's loads and waits pays the L2 / MALL latency once per trip; the streaming kernels around the fits
// (moments, Cook's distances, replacement, the intercept fit) ran at 30-40 % issue at m = 2000 for that reason.
template <int NB, class LoadF, class UseF>
DSQ_DEV void sweep_batched(int m, int lane, LoadF &&load, UseF &&use) {
    for (int j0 = lane; j0 < m; j0 += 64 * NB) {
        _Pragma("unroll")
        for (int b = 0; b < NB; b++) {
            const int j = j0 + 64 * b;
            if (j < m) load(j, b);
        }
        _Pragma("unroll")
        for (int b = 0; b < NB; b++) {
            const int j = j0 + 64 * b;
            if (j < m) use(j, b);
        }
    }
}

}  // namespace dsq
