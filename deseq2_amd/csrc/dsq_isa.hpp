// dsq_isa.hpp -- the polynomial cores of the f64 math (dsq_math.hpp) written in gfx950 ISA.
//
// Why (round 5; VERDICT r4 "next" #1, profiles/r05_isa_cores.md): left to the compiler, the ~40 polynomial coefficients
// of log / exp / lgamma / digamma are loop invariants of a fit kernel.  It hoists them, runs out of SGPRs (fit_disp<4>:
// 236 SGPR spill slots) and parks them in VGPRs -- 48 of the search kernel's 168 registers held constants -- and then
// every Horner step `p = fma(p, r, C)` becomes `v_mov_b64 tmp, vC ; v_fmac_f64 tmp, p, r`, because the two-address
// v_fmac destroys its addend.  Here each core is ONE asm block:
//   * the coefficients are fetched by the block itself with s_load_dwordx2 from a constant table (scalar cache, issued at
//     the head of the block, in flight while the division / range reduction runs) into TRANSIENT SGPR pairs that the
//     register allocator picks ("=&s") and that are dead again at the end of the block;
//   * every Horner step is the three-address VOP3 `v_fma_f64 vP, vP, vR, s[C]` (one SGPR operand per instruction: the
//     gfx9 constant-bus limit; a chain starts with one v_mov_b64 from its leading coefficient);
//   * no coefficient lives in a register outside the block.
// The operations, their operands and their order are exactly those of the C++ expressions they replace (dsq_math.hpp keeps
// them under DSQ_ISA_CORES == 0, and the CPU checker of the test suite states the same sequence): same fmas on the same
// values, so the same bits -- tests/test_gpu_math.py and bench.py's result_digest do not move.
//
// Hazards handled by hand (the compiler's hazard recognizer does not look inside asm): gfx950 forwards a transcendental
// result (v_rcp_f64) to the next VALU instruction only after one wait state -- an independent instruction follows each;
// SMEM results are used only behind `s_waitcnt lgkmcnt(0)` (which also drains the compiler's own LDS reads: harmless;
// an extra operation in flight only makes the compiler's own counted waits more conservative).  The blocks are not
// `volatile` and clobber nothing: pure functions of their inputs (the table is immutable), free to be CSE'd or dropped.
#pragma once
#include <hip/hip_runtime.h>

#ifndef DSQ_ISA_CORES
#define DSQ_ISA_CORES 1
#endif

namespace dsq {

#define DSQ_ISA_DEV __device__ __forceinline__

// ---- the coefficient table (byte offsets below are 8 * index) ----------------------------------------------------
// Written with the very constant expressions dsq_math.hpp uses, so the compiler folds them to the same doubles.
enum {
    ISA_LG1 = 0, ISA_LG2, ISA_LG3, ISA_LG4, ISA_LG5, ISA_LG6, ISA_LG7, ISA_LN2LO, ISA_LN2HI,      // log_core
    ISA_EXP_C13, ISA_EXP_C12, ISA_EXP_C11, ISA_EXP_C10, ISA_EXP_C9, ISA_EXP_C8, ISA_EXP_C7,       // dexp: 1/13! ... 1/3!
    ISA_EXP_C6, ISA_EXP_C5, ISA_EXP_C4, ISA_EXP_C3, ISA_INVLN2,
    ISA_LGC0, ISA_LGC1, ISA_LGC2, ISA_LGC3, ISA_LGC4, ISA_LGC5, ISA_LGC6, ISA_LGC7,              // lgamma's Stirling sum
    ISA_DGC0, ISA_DGC1, ISA_DGC2, ISA_DGC3, ISA_DGC4, ISA_DGC5, ISA_DGC6, ISA_DGC7,              // digamma's
    ISA_TGC0, ISA_TGC1, ISA_TGC2, ISA_TGC3, ISA_TGC4, ISA_TGC5, ISA_TGC6, ISA_TGC7,              // trigamma's
    ISA_TAB_N
};
__constant__ const double kIsaTab[ISA_TAB_N] = {
    6.666666666666735130e-01, 3.999999999940941908e-01, 2.857142874366239149e-01, 2.222219843214978396e-01,
    1.818357216161805012e-01, 1.531383769920937332e-01, 1.479819860511658591e-01,
    1.90821492927058770002e-10, 6.93147180369123816490e-01,
    1.0 / 6227020800.0, 1.0 / 479001600.0, 1.0 / 39916800.0, 1.0 / 3628800.0, 1.0 / 362880.0, 1.0 / 40320.0, 1.0 / 5040.0,
    1.0 / 720.0, 1.0 / 120.0, 1.0 / 24.0, 1.0 / 6.0, 1.44269504088896338700e+00,
    -3617.0 / 122400.0, 1.0 / 156.0, -691.0 / 360360.0, 1.0 / 1188.0, -1.0 / 1680.0, 1.0 / 1260.0, -1.0 / 360.0, 1.0 / 12.0,
    -3617.0 / 8160.0, 1.0 / 12.0, -691.0 / 32760.0, 1.0 / 132.0, -1.0 / 240.0, 1.0 / 252.0, -1.0 / 120.0, 1.0 / 12.0,
    -3617.0 / 510.0, 7.0 / 6.0, -691.0 / 2730.0, 5.0 / 66.0, -1.0 / 30.0, 1.0 / 42.0, -1.0 / 30.0, 1.0 / 6.0,
};

#define DSQ_ISA_STR2(x) #x
#define DSQ_ISA_STR(x) DSQ_ISA_STR2(x)
#define DSQ_ISA_LD(reg, idx) "s_load_dwordx2 %[" #reg "], %[tab], " DSQ_ISA_STR(idx) "*8\n\t"

// ---- log_core(f, dk, c) of dsq_math.hpp ---------------------------------------------------------------------------
//   hfsq = 0.5 f f;  s = f / (2 + f) [scaling-free division: ddiv_n];  z = s s;  w = z z
//   t1 = w fma(w, fma(w, Lg6, Lg4), Lg2);  t2 = z fma(w, fma(w, fma(w, Lg7, Lg5), Lg3), Lg1);  R = t2 + t1
//   u = fma(s, hfsq + R, fma(dk, ln2lo, c));  return fma(dk, ln2hi, (u - hfsq) + f)
// 30 VALU instructions (the compiler's form: 34 with its five constant copies), 6 VGPR pairs + 9 SGPR pairs inside.
template <bool WITH_C>
DSQ_ISA_DEV double isa_log_core(double f, double dk, double c) {
    double out, r, e, q, h, s;
    double L1, L2, L3, L4, L5, L6, L7, NLO, NHI;
    if constexpr (WITH_C) {
        asm(DSQ_ISA_LD(L6, 5) DSQ_ISA_LD(L7, 6) DSQ_ISA_LD(L4, 3) DSQ_ISA_LD(L5, 4) DSQ_ISA_LD(L2, 1) DSQ_ISA_LD(L3, 2)
            DSQ_ISA_LD(L1, 0) DSQ_ISA_LD(NLO, 7) DSQ_ISA_LD(NHI, 8)
            "v_add_f64 %[out], %[f], 2.0\n\t"                    /* d = 2 + f */
            "v_mul_f64 %[h], %[f], 0.5\n\t"                      /* 0.5 f */
            "v_rcp_f64 %[r], %[out]\n\t"
            "v_mul_f64 %[h], %[h], %[f]\n\t"                     /* hfsq (also the wait state behind v_rcp) */
            "v_fma_f64 %[e], -%[out], %[r], 1.0\n\t"
            "v_fma_f64 %[r], %[r], %[e], %[r]\n\t"
            "v_fma_f64 %[e], -%[out], %[r], 1.0\n\t"
            "v_fma_f64 %[r], %[r], %[e], %[r]\n\t"
            "v_mul_f64 %[q], %[f], %[r]\n\t"                     /* q0 = f r */
            "v_fma_f64 %[e], -%[out], %[q], %[f]\n\t"            /* residual */
            "v_fma_f64 %[s], %[e], %[r], %[q]\n\t"               /* s */
            "v_mul_f64 %[e], %[s], %[s]\n\t"                     /* z */
            "v_mul_f64 %[q], %[e], %[e]\n\t"                     /* w */
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_mov_b64 %[out], %[L6]\n\t"
            "v_mov_b64 %[r], %[L7]\n\t"
            "v_fma_f64 %[out], %[q], %[out], %[L4]\n\t"
            "v_fma_f64 %[r], %[q], %[r], %[L5]\n\t"
            "v_fma_f64 %[out], %[q], %[out], %[L2]\n\t"
            "v_fma_f64 %[r], %[q], %[r], %[L3]\n\t"
            "v_mul_f64 %[out], %[q], %[out]\n\t"                 /* t1 */
            "v_fma_f64 %[r], %[q], %[r], %[L1]\n\t"
            "v_mul_f64 %[r], %[e], %[r]\n\t"                     /* t2 */
            "v_add_f64 %[out], %[r], %[out]\n\t"                 /* R = t2 + t1 */
            "v_add_f64 %[out], %[h], %[out]\n\t"                 /* hfsq + R */
            "v_fma_f64 %[r], %[dk], %[NLO], %[c]\n\t"
            "v_fma_f64 %[out], %[s], %[out], %[r]\n\t"           /* u */
            "v_add_f64 %[out], %[out], -%[h]\n\t"
            "v_add_f64 %[out], %[out], %[f]\n\t"
            "v_fma_f64 %[out], %[dk], %[NHI], %[out]\n\t"
            : [out] "=&v"(out), [r] "=&v"(r), [e] "=&v"(e), [q] "=&v"(q), [h] "=&v"(h), [s] "=&v"(s),
              [L1] "=&s"(L1), [L2] "=&s"(L2), [L3] "=&s"(L3), [L4] "=&s"(L4), [L5] "=&s"(L5), [L6] "=&s"(L6), [L7] "=&s"(L7),
              [NLO] "=&s"(NLO), [NHI] "=&s"(NHI)
            : [f] "v"(f), [dk] "v"(dk), [c] "v"(c), [tab] "s"(kIsaTab));
    } else {
        asm(DSQ_ISA_LD(L6, 5) DSQ_ISA_LD(L7, 6) DSQ_ISA_LD(L4, 3) DSQ_ISA_LD(L5, 4) DSQ_ISA_LD(L2, 1) DSQ_ISA_LD(L3, 2)
            DSQ_ISA_LD(L1, 0) DSQ_ISA_LD(NLO, 7) DSQ_ISA_LD(NHI, 8)
            "v_add_f64 %[out], %[f], 2.0\n\t"
            "v_mul_f64 %[h], %[f], 0.5\n\t"
            "v_rcp_f64 %[r], %[out]\n\t"
            "v_mul_f64 %[h], %[h], %[f]\n\t"
            "v_fma_f64 %[e], -%[out], %[r], 1.0\n\t"
            "v_fma_f64 %[r], %[r], %[e], %[r]\n\t"
            "v_fma_f64 %[e], -%[out], %[r], 1.0\n\t"
            "v_fma_f64 %[r], %[r], %[e], %[r]\n\t"
            "v_mul_f64 %[q], %[f], %[r]\n\t"
            "v_fma_f64 %[e], -%[out], %[q], %[f]\n\t"
            "v_fma_f64 %[s], %[e], %[r], %[q]\n\t"
            "v_mul_f64 %[e], %[s], %[s]\n\t"
            "v_mul_f64 %[q], %[e], %[e]\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_mov_b64 %[out], %[L6]\n\t"
            "v_mov_b64 %[r], %[L7]\n\t"
            "v_fma_f64 %[out], %[q], %[out], %[L4]\n\t"
            "v_fma_f64 %[r], %[q], %[r], %[L5]\n\t"
            "v_fma_f64 %[out], %[q], %[out], %[L2]\n\t"
            "v_fma_f64 %[r], %[q], %[r], %[L3]\n\t"
            "v_mul_f64 %[out], %[q], %[out]\n\t"
            "v_fma_f64 %[r], %[q], %[r], %[L1]\n\t"
            "v_mul_f64 %[r], %[e], %[r]\n\t"
            "v_add_f64 %[out], %[r], %[out]\n\t"
            "v_add_f64 %[out], %[h], %[out]\n\t"
            "v_fma_f64 %[r], %[dk], %[NLO], 0\n\t"               /* fma(dk, ln2lo, +0.0) */
            "v_fma_f64 %[out], %[s], %[out], %[r]\n\t"
            "v_add_f64 %[out], %[out], -%[h]\n\t"
            "v_add_f64 %[out], %[out], %[f]\n\t"
            "v_fma_f64 %[out], %[dk], %[NHI], %[out]\n\t"
            : [out] "=&v"(out), [r] "=&v"(r), [e] "=&v"(e), [q] "=&v"(q), [h] "=&v"(h), [s] "=&v"(s),
              [L1] "=&s"(L1), [L2] "=&s"(L2), [L3] "=&s"(L3), [L4] "=&s"(L4), [L5] "=&s"(L5), [L6] "=&s"(L6), [L7] "=&s"(L7),
              [NLO] "=&s"(NLO), [NHI] "=&s"(NHI)
            : [f] "v"(f), [dk] "v"(dk), [tab] "s"(kIsaTab));
    }
    return out;
}

// ---- the reduced-argument part of dexp(x) ---------------------------------------------------------------------------
//   kf = rint(x / ln2);  hi = fma(-kf, ln2hi, x);  lo = kf ln2lo;  r = hi - lo;  rerr = (hi - r) - lo
//   p = 1/13!;  p = fma(p, r, 1/12!) ... fma(p, r, 1/3!);  p = fma(p, r, 0.5);  t = fma(r r, p, r) + rerr;  y = 1 + t
// returns y, and kf through kf_out (the caller scales by 2^kf in two steps, as before)
DSQ_ISA_DEV double isa_exp_core(double x, double &kf_out) {
    double y, kf, r, t, u;
    double C13, C12, C11, C10, C9, C8, C7, C6, C5, C4, C3, ILN2, NLO, NHI;
    asm(DSQ_ISA_LD(ILN2, 20) DSQ_ISA_LD(NHI, 8) DSQ_ISA_LD(NLO, 7)
        "s_waitcnt lgkmcnt(0)\n\t"
        DSQ_ISA_LD(C13, 9) DSQ_ISA_LD(C12, 10) DSQ_ISA_LD(C11, 11) DSQ_ISA_LD(C10, 12) DSQ_ISA_LD(C9, 13) DSQ_ISA_LD(C8, 14)
        DSQ_ISA_LD(C7, 15) DSQ_ISA_LD(C6, 16) DSQ_ISA_LD(C5, 17) DSQ_ISA_LD(C4, 18) DSQ_ISA_LD(C3, 19)
        "v_mul_f64 %[kf], %[x], %[ILN2]\n\t"
        "v_rndne_f64 %[kf], %[kf]\n\t"
        "v_fma_f64 %[t], -%[kf], %[NHI], %[x]\n\t"               /* hi */
        "v_mul_f64 %[u], %[kf], %[NLO]\n\t"                      /* lo */
        "v_add_f64 %[r], %[t], -%[u]\n\t"                        /* r = hi - lo */
        "v_add_f64 %[t], %[t], -%[r]\n\t"                        /* hi - r */
        "v_add_f64 %[t], %[t], -%[u]\n\t"                        /* rerr */
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_mov_b64 %[y], %[C13]\n\t"
        "v_fma_f64 %[y], %[y], %[r], %[C12]\n\t"
        "v_fma_f64 %[y], %[y], %[r], %[C11]\n\t"
        "v_fma_f64 %[y], %[y], %[r], %[C10]\n\t"
        "v_fma_f64 %[y], %[y], %[r], %[C9]\n\t"
        "v_fma_f64 %[y], %[y], %[r], %[C8]\n\t"
        "v_fma_f64 %[y], %[y], %[r], %[C7]\n\t"
        "v_fma_f64 %[y], %[y], %[r], %[C6]\n\t"
        "v_fma_f64 %[y], %[y], %[r], %[C5]\n\t"
        "v_fma_f64 %[y], %[y], %[r], %[C4]\n\t"
        "v_fma_f64 %[y], %[y], %[r], %[C3]\n\t"
        "v_fma_f64 %[y], %[y], %[r], 0.5\n\t"
        "v_mul_f64 %[u], %[r], %[r]\n\t"                         /* r2 */
        "v_fma_f64 %[y], %[u], %[y], %[r]\n\t"                   /* t = fma(r2, p, r) */
        "v_add_f64 %[y], %[y], %[t]\n\t"                         /* + rerr */
        "v_add_f64 %[y], 1.0, %[y]\n\t"                          /* 1 + t */
        : [y] "=&v"(y), [kf] "=&v"(kf), [r] "=&v"(r), [t] "=&v"(t), [u] "=&v"(u),
          [C13] "=&s"(C13), [C12] "=&s"(C12), [C11] "=&s"(C11), [C10] "=&s"(C10), [C9] "=&s"(C9), [C8] "=&s"(C8), [C7] "=&s"(C7),
          [C6] "=&s"(C6), [C5] "=&s"(C5), [C4] "=&s"(C4), [C3] "=&s"(C3), [ILN2] "=&s"(ILN2), [NLO] "=&s"(NLO), [NHI] "=&s"(NHI)
        : [x] "v"(x), [tab] "s"(kIsaTab));
    kf_out = kf;
    return y;
}

// ---- the Stirling sums of lgamma and digamma at the same r2 = 1 / xs^2 ----------------------------------------------
//   c = fma(... fma(fma(C0, r2, C1), r2, C2) ..., r2, C7)   (lgamma: -3617/122400, 1/156, ..., 1/12)
//   d = the same with digamma's coefficients                 (-3617/8160, 1/12, ..., 1/12)
#define DSQ_ISA_HORNER8(acc, K0, K1, K2, K3, K4, K5, K6, K7)                                                             \
    "v_mov_b64 %[" #acc "], %[" #K0 "]\n\t"                                                                              \
    "v_fma_f64 %[" #acc "], %[" #acc "], %[r2], %[" #K1 "]\n\t"                                                          \
    "v_fma_f64 %[" #acc "], %[" #acc "], %[r2], %[" #K2 "]\n\t"                                                          \
    "v_fma_f64 %[" #acc "], %[" #acc "], %[r2], %[" #K3 "]\n\t"                                                          \
    "v_fma_f64 %[" #acc "], %[" #acc "], %[r2], %[" #K4 "]\n\t"                                                          \
    "v_fma_f64 %[" #acc "], %[" #acc "], %[r2], %[" #K5 "]\n\t"                                                          \
    "v_fma_f64 %[" #acc "], %[" #acc "], %[r2], %[" #K6 "]\n\t"                                                          \
    "v_fma_f64 %[" #acc "], %[" #acc "], %[r2], %[" #K7 "]\n\t"

template <int BASE>
DSQ_ISA_DEV double isa_stirling8(double r2) {
    double c;
    double K0, K1, K2, K3, K4, K5, K6, K7;
    asm("s_load_dwordx2 %[K0], %[tab], %[o0]\n\t"
        "s_load_dwordx2 %[K1], %[tab], %[o1]\n\t"
        "s_load_dwordx2 %[K2], %[tab], %[o2]\n\t"
        "s_load_dwordx2 %[K3], %[tab], %[o3]\n\t"
        "s_load_dwordx2 %[K4], %[tab], %[o4]\n\t"
        "s_load_dwordx2 %[K5], %[tab], %[o5]\n\t"
        "s_load_dwordx2 %[K6], %[tab], %[o6]\n\t"
        "s_load_dwordx2 %[K7], %[tab], %[o7]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        DSQ_ISA_HORNER8(c, K0, K1, K2, K3, K4, K5, K6, K7)
        : [c] "=&v"(c), [K0] "=&s"(K0), [K1] "=&s"(K1), [K2] "=&s"(K2), [K3] "=&s"(K3), [K4] "=&s"(K4), [K5] "=&s"(K5),
          [K6] "=&s"(K6), [K7] "=&s"(K7)
        : [r2] "v"(r2), [tab] "s"(kIsaTab), [o0] "n"(8 * BASE), [o1] "n"(8 * BASE + 8), [o2] "n"(8 * BASE + 16),
          [o3] "n"(8 * BASE + 24), [o4] "n"(8 * BASE + 32), [o5] "n"(8 * BASE + 40), [o6] "n"(8 * BASE + 48), [o7] "n"(8 * BASE + 56));
    return c;
}
DSQ_ISA_DEV double isa_stirling_lgamma(double r2) { return isa_stirling8<ISA_LGC0>(r2); }
DSQ_ISA_DEV double isa_stirling_digamma(double r2) { return isa_stirling8<ISA_DGC0>(r2); }
DSQ_ISA_DEV double isa_stirling_trigamma(double r2) { return isa_stirling8<ISA_TGC0>(r2); }

// both sums in one block: the sixteen loads in flight together, the two chains interleaved (each step of one chain
// covers the result latency of the other)
DSQ_ISA_DEV void isa_stirling_pair(double r2, double &c_out, double &d_out) {
    double c, d;
    double K0, K1, K2, K3, K4, K5, K6, K7, D0, D1, D2, D3, D4, D5, D6, D7;
    asm(DSQ_ISA_LD(K0, 21) DSQ_ISA_LD(D0, 29) DSQ_ISA_LD(K1, 22) DSQ_ISA_LD(D1, 30) DSQ_ISA_LD(K2, 23) DSQ_ISA_LD(D2, 31)
        DSQ_ISA_LD(K3, 24) DSQ_ISA_LD(D3, 32) DSQ_ISA_LD(K4, 25) DSQ_ISA_LD(D4, 33) DSQ_ISA_LD(K5, 26) DSQ_ISA_LD(D5, 34)
        DSQ_ISA_LD(K6, 27) DSQ_ISA_LD(D6, 35) DSQ_ISA_LD(K7, 28) DSQ_ISA_LD(D7, 36)
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_mov_b64 %[c], %[K0]\n\t"
        "v_mov_b64 %[d], %[D0]\n\t"
        "v_fma_f64 %[c], %[c], %[r2], %[K1]\n\t"
        "v_fma_f64 %[d], %[d], %[r2], %[D1]\n\t"
        "v_fma_f64 %[c], %[c], %[r2], %[K2]\n\t"
        "v_fma_f64 %[d], %[d], %[r2], %[D2]\n\t"
        "v_fma_f64 %[c], %[c], %[r2], %[K3]\n\t"
        "v_fma_f64 %[d], %[d], %[r2], %[D3]\n\t"
        "v_fma_f64 %[c], %[c], %[r2], %[K4]\n\t"
        "v_fma_f64 %[d], %[d], %[r2], %[D4]\n\t"
        "v_fma_f64 %[c], %[c], %[r2], %[K5]\n\t"
        "v_fma_f64 %[d], %[d], %[r2], %[D5]\n\t"
        "v_fma_f64 %[c], %[c], %[r2], %[K6]\n\t"
        "v_fma_f64 %[d], %[d], %[r2], %[D6]\n\t"
        "v_fma_f64 %[c], %[c], %[r2], %[K7]\n\t"
        "v_fma_f64 %[d], %[d], %[r2], %[D7]\n\t"
        : [c] "=&v"(c), [d] "=&v"(d),
          [K0] "=&s"(K0), [K1] "=&s"(K1), [K2] "=&s"(K2), [K3] "=&s"(K3), [K4] "=&s"(K4), [K5] "=&s"(K5), [K6] "=&s"(K6), [K7] "=&s"(K7),
          [D0] "=&s"(D0), [D1] "=&s"(D1), [D2] "=&s"(D2), [D3] "=&s"(D3), [D4] "=&s"(D4), [D5] "=&s"(D5), [D6] "=&s"(D6), [D7] "=&s"(D7)
        : [r2] "v"(r2), [tab] "s"(kIsaTab));
    c_out = c;
    d_out = d;
}

static_assert(ISA_LG6 == 5 && ISA_LG7 == 6 && ISA_LN2LO == 7 && ISA_LN2HI == 8 && ISA_EXP_C13 == 9 && ISA_EXP_C3 == 19 &&
              ISA_INVLN2 == 20 && ISA_LGC0 == 21 && ISA_LGC7 == 28 && ISA_DGC0 == 29 && ISA_DGC7 == 36 && ISA_TGC0 == 37,
              "the asm blocks address kIsaTab by literal index");

}  // namespace dsq
