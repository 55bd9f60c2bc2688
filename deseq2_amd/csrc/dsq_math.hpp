// dsq_math.hpp -- f64 scalar math for the gfx950 NB-GLM kernels.
//
// DESeq2's native code (src/DESeq2.cpp) leans on R's nmath (lgammafn, digamma,
// trigamma, dnbinom_mu) and libm exp/log.  None of that exists on the device, and
// the ROCm device libm cannot be reproduced on a CPU, so the engine carries its own
// f64 routines with a fully pinned operation sequence: Cody-Waite + Taylor exp,
// atanh-series log/log1p, Stirling + upward-shift lgamma/digamma/trigamma, and the
// Loader saddle-point NB density (stirlerr / bd0) that R's dnbinom_mu uses.
// Every rounding is explicit: compile with -ffp-contract=off; fma only where written.
// tests/test_gpu_math.py compares these bit-for-bit with the CPU oracle.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "dsq_isa.hpp"

namespace dsq {

#define DSQ_DEV __device__ __forceinline__

DSQ_DEV double bits2d(uint64_t u) { return __builtin_bit_cast(double, u); }
DSQ_DEV uint64_t d2bits(double d) { return __builtin_bit_cast(uint64_t, d); }

constexpr double kLn2Hi = 6.93147180369123816490e-01;
constexpr double kLn2Lo = 1.90821492927058770002e-10;
constexpr double kInvLn2 = 1.44269504088896338700e+00;
constexpr double kLnSqrt2Pi = 0.918938533204672741780329736406;
constexpr double kLn2Pi = 1.837877066409345483560659472811;
constexpr double kTwoPi = 6.283185307179586476925286766559;
constexpr double kDblMin = 2.2250738585072014e-308;
constexpr double kInf = __builtin_huge_val();

DSQ_DEV double dnan() { return __builtin_nan(""); }
DSQ_DEV bool dfinite(double x) { return __builtin_fabs(x) < kInf; }

// ---------------------------------------------------------------------- exp
DSQ_DEV double dexp(double x) {
    if (x != x) return x;
    if (x > 709.782712893384) return kInf;
    if (x < -745.1332191019412) return 0.0;
#if DSQ_ISA_CORES
    double kf;
    const double y = isa_exp_core(x, kf);        // dsq_isa.hpp: the same operations, coefficients from the scalar cache
#else
    double kf = __builtin_rint(x * kInvLn2);
    double hi = __builtin_fma(-kf, kLn2Hi, x);
    double lo = kf * kLn2Lo;
    double r = hi - lo;
    double rerr = (hi - r) - lo;
    double p = 1.0 / 6227020800.0;
    p = __builtin_fma(p, r, 1.0 / 479001600.0);
    p = __builtin_fma(p, r, 1.0 / 39916800.0);
    p = __builtin_fma(p, r, 1.0 / 3628800.0);
    p = __builtin_fma(p, r, 1.0 / 362880.0);
    p = __builtin_fma(p, r, 1.0 / 40320.0);
    p = __builtin_fma(p, r, 1.0 / 5040.0);
    p = __builtin_fma(p, r, 1.0 / 720.0);
    p = __builtin_fma(p, r, 1.0 / 120.0);
    p = __builtin_fma(p, r, 1.0 / 24.0);
    p = __builtin_fma(p, r, 1.0 / 6.0);
    p = __builtin_fma(p, r, 0.5);
    double r2 = r * r;
    double t = __builtin_fma(r2, p, r);
    t = t + rerr;
    double y = 1.0 + t;
#endif
    int k = (int)kf;
    int k1 = k >> 1;
    int k2 = k - k1;
    double s1 = bits2d((uint64_t)(uint32_t)(k1 + 1023) << 52);
    double s2 = bits2d((uint64_t)(uint32_t)(k2 + 1023) << 52);
    return (y * s1) * s2;
}

// ------------------------------------------------------------- division fast path
// 1 / x and a / x for a divisor KNOWN to be a normal number far from the ends of the exponent range (and a numerator
// that is zero or not tiny): the sequence the compiler's IEEE division runs -- v_rcp_f64, two Newton steps, q0 = a r,
// the residual fma(-x, q0, a), the correction fma(res, r, q0) -- WITHOUT the range scaling around it (v_div_scale x 2,
// v_div_fmas, v_div_fixup: no-ops when nothing needs scaling).  Same operations on the same values, so the same
// correctly rounded quotient as `a / x` (and as the CPU checker's division), at 7 instead of 11 instructions.
DSQ_DEV double drcp_n(double x) {
    double r = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-x, r, 1.0);
    r = __builtin_fma(r, e, r);
    const double res = __builtin_fma(-x, r, 1.0);
    return __builtin_fma(res, r, r);
}
DSQ_DEV double ddiv_n(double a, double x) {
    double r = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-x, r, 1.0);
    r = __builtin_fma(r, e, r);
    const double q0 = a * r;
    const double res = __builtin_fma(-x, q0, a);
    return __builtin_fma(res, r, q0);
}

// ---------------------------------------------------------------------- log
constexpr double kLg1 = 6.666666666666735130e-01;
constexpr double kLg2 = 3.999999999940941908e-01;
constexpr double kLg3 = 2.857142874366239149e-01;
constexpr double kLg4 = 2.222219843214978396e-01;
constexpr double kLg5 = 1.818357216161805012e-01;
constexpr double kLg6 = 1.531383769920937332e-01;
constexpr double kLg7 = 1.479819860511658591e-01;

// fdlibm's e_log.c evaluation (as musl restates it) with the polynomial and the closing sum written as fused multiply-adds
// (round 4: ~10 of its ~45 instructions; every fma drops one rounding of the mul + add pair it replaces, so the error
// bound of the original form -- below 1 ulp -- still holds; the CPU checker of the test suite runs the same sequence)
DSQ_DEV double log_core(double f, double dk, double c) {
#if DSQ_ISA_CORES
    return isa_log_core<true>(f, dk, c);         // dsq_isa.hpp: the same operations, coefficients from the scalar cache
#else
    double hfsq = 0.5 * f * f;
    double s = ddiv_n(f, 2.0 + f);      // 2 + f in [1.7, 2.42], f = 0 or |f| >= 2^-53: nothing to scale
    double z = s * s;
    double w = z * z;
    double t1 = w * __builtin_fma(w, __builtin_fma(w, kLg6, kLg4), kLg2);
    double t2 = z * __builtin_fma(w, __builtin_fma(w, __builtin_fma(w, kLg7, kLg5), kLg3), kLg1);
    double R = t2 + t1;
    double u = __builtin_fma(s, hfsq + R, __builtin_fma(dk, kLn2Lo, c));
    return __builtin_fma(dk, kLn2Hi, (u - hfsq) + f);
#endif
}
// c = +0.0 (the plain logarithm)
DSQ_DEV double log_core0(double f, double dk) {
#if DSQ_ISA_CORES
    return isa_log_core<false>(f, dk, 0.0);
#else
    return log_core(f, dk, 0.0);
#endif
}

// general version: NaN / negative / zero / subnormal / +inf handled
DSQ_DEV double dlog_full(double x) {
    if (x != x) return x;
    if (x < 0.0) return dnan();
    if (x == 0.0) return -kInf;
    if (x == kInf) return x;
    int k = 0;
    if (x < kDblMin) { x *= 18014398509481984.0; k = -54; }
    uint64_t ix = d2bits(x);
    ix += (uint64_t)(0x3ff00000u - 0x3fe6a09eu) << 32;
    k += (int)(ix >> 52) - 0x3ff;
    ix = (ix & 0x000fffffffffffffULL) + ((uint64_t)0x3fe6a09eu << 32);
    double f = bits2d(ix) - 1.0;
    return log_core0(f, (double)k);
}

// positive normal finite argument: same bits as dlog_full, no special-case tests
DSQ_DEV double dlog_pn(double x) {
    uint64_t ix = d2bits(x);
    ix += (uint64_t)(0x3ff00000u - 0x3fe6a09eu) << 32;
    int k = (int)(ix >> 52) - 0x3ff;
    ix = (ix & 0x000fffffffffffffULL) + ((uint64_t)0x3fe6a09eu << 32);
    double f = bits2d(ix) - 1.0;
    return log_core0(f, (double)k);
}

// The special cases almost never occur in the kernels; test them once per wave (a scalar
// branch) instead of per lane.  Both branches return identical bits for ordinary arguments.
DSQ_DEV double dlog(double x) {
    bool special = !(x >= kDblMin && x < kInf);
    if (__any(special)) return dlog_full(x);
    return dlog_pn(x);
}

DSQ_DEV double dlog1p(double x) {
    if (x != x) return x;
    if (x < -1.0) return dnan();
    if (x == -1.0) return -kInf;
    if (x == kInf) return x;
    if (__builtin_fabs(x) < 1.1102230246251565e-16) return x;
    if (x > -0.2928932188134524 && x < 0.41421356237309503) return log_core0(x, 0.0);
    double u = 1.0 + x;
    uint64_t iu = d2bits(u);
    iu += (uint64_t)(0x3ff00000u - 0x3fe6a09eu) << 32;
    int k = (int)(iu >> 52) - 0x3ff;
    double c = 0.0;
    if (k < 54) {
        c = (k >= 2) ? 1.0 - (u - x) : x - (u - 1.0);
        c = c / u;
    }
    iu = (iu & 0x000fffffffffffffULL) + ((uint64_t)0x3fe6a09eu << 32);
    double f = bits2d(iu) - 1.0;
    return log_core(f, (double)k, c);
}

// ------------------------------------------------------------------- lgamma
// Stirling with the lgammacor sum for x >= 10, upward shift below (domain x > 0).
DSQ_DEV double dlgamma(double x) {
    if (x != x) return x;
    if (x <= 0.0) return (x == 0.0) ? kInf : dnan();
    if (x == kInf) return x;
    double prod = 1.0, xs = x;
    bool shifted = false;
    for (int i = 0; i < 10; i++) {
        if (!__any(xs < 10.0)) break;   // wave-uniform exit; lanes already >= 10 are untouched
        if (xs < 10.0) { prod = prod * xs; xs = xs + 1.0; shifted = true; }
    }
    double lx = dlog(xs);
    double rx = 1.0 / xs;
    double r2 = rx * rx;
#if DSQ_ISA_CORES
    double c = isa_stirling_lgamma(r2);
#else
    double c = -3617.0 / 122400.0;
    c = __builtin_fma(c, r2, 1.0 / 156.0);
    c = __builtin_fma(c, r2, -691.0 / 360360.0);
    c = __builtin_fma(c, r2, 1.0 / 1188.0);
    c = __builtin_fma(c, r2, -1.0 / 1680.0);
    c = __builtin_fma(c, r2, 1.0 / 1260.0);
    c = __builtin_fma(c, r2, -1.0 / 360.0);
    c = __builtin_fma(c, r2, 1.0 / 12.0);
#endif
    double cor = c * rx;
    double res = kLnSqrt2Pi + (xs - 0.5) * lx - xs + cor;
    if (__any(shifted)) {
        double lp = dlog(prod);
        if (shifted) res = res - lp;
    }
    return res;
}

DSQ_DEV double ddigamma(double x) {
    if (x != x) return x;
    if (x <= 0.0) return dnan();
    if (x == kInf) return x;
    double num = 0.0, den = 1.0, xs = x;
    bool shifted = false;
    for (int i = 0; i < 10; i++) {
        if (!__any(xs < 10.0)) break;
        if (xs < 10.0) {
            num = __builtin_fma(num, xs, den); den = den * xs; xs = xs + 1.0; shifted = true;
        }
    }
    double lx = dlog(xs);
    double rx = 1.0 / xs;
    double r2 = rx * rx;
#if DSQ_ISA_CORES
    double c = isa_stirling_digamma(r2);
#else
    double c = -3617.0 / 8160.0;
    c = __builtin_fma(c, r2, 1.0 / 12.0);
    c = __builtin_fma(c, r2, -691.0 / 32760.0);
    c = __builtin_fma(c, r2, 1.0 / 132.0);
    c = __builtin_fma(c, r2, -1.0 / 240.0);
    c = __builtin_fma(c, r2, 1.0 / 252.0);
    c = __builtin_fma(c, r2, -1.0 / 120.0);
    c = __builtin_fma(c, r2, 1.0 / 12.0);
#endif
    double res = (lx - 0.5 * rx) - c * r2;
    if (shifted) res = res - num / den;
    return res;
}

// lgamma(x) and digamma(x) of the SAME argument in one go.  Both routines shift x up the same
// way and both need log(xs) and 1/xs; evaluating them together shares that work.  Every value is
// produced by the same operations as in dlgamma / ddigamma, so the results are bit-identical.
DSQ_DEV void dlgamma_digamma(double x, double &lg, double &dg) {
    if (x != x) { lg = x; dg = x; return; }
    if (x <= 0.0) { lg = (x == 0.0) ? kInf : dnan(); dg = dnan(); return; }
    if (x == kInf) { lg = x; dg = x; return; }
    double prod = 1.0, num = 0.0, den = 1.0, xs = x;
    bool shifted = false;
    for (int i = 0; i < 10; i++) {
        if (!__any(xs < 10.0)) break;
        if (xs < 10.0) {
            prod = prod * xs;
            num = __builtin_fma(num, xs, den); den = den * xs;
            xs = xs + 1.0; shifted = true;
        }
    }
    double lx = dlog(xs);
    double rx = 1.0 / xs;
    double r2 = rx * rx;
#if DSQ_ISA_CORES
    double c, d;
    isa_stirling_pair(r2, c, d);                 // dsq_isa.hpp: both sums, the same operations
    double cor = c * rx;
    double res = kLnSqrt2Pi + (xs - 0.5) * lx - xs + cor;
#else
    double c = -3617.0 / 122400.0;
    c = __builtin_fma(c, r2, 1.0 / 156.0);
    c = __builtin_fma(c, r2, -691.0 / 360360.0);
    c = __builtin_fma(c, r2, 1.0 / 1188.0);
    c = __builtin_fma(c, r2, -1.0 / 1680.0);
    c = __builtin_fma(c, r2, 1.0 / 1260.0);
    c = __builtin_fma(c, r2, -1.0 / 360.0);
    c = __builtin_fma(c, r2, 1.0 / 12.0);
    double cor = c * rx;
    double res = kLnSqrt2Pi + (xs - 0.5) * lx - xs + cor;
    double d = -3617.0 / 8160.0;
    d = __builtin_fma(d, r2, 1.0 / 12.0);
    d = __builtin_fma(d, r2, -691.0 / 32760.0);
    d = __builtin_fma(d, r2, 1.0 / 132.0);
    d = __builtin_fma(d, r2, -1.0 / 240.0);
    d = __builtin_fma(d, r2, 1.0 / 252.0);
    d = __builtin_fma(d, r2, -1.0 / 120.0);
    d = __builtin_fma(d, r2, 1.0 / 12.0);
#endif
    double dres = (lx - 0.5 * rx) - d * r2;
    if (__any(shifted)) {
        double lp = dlog(prod);
        double q = num / den;
        if (shifted) { res = res - lp; dres = dres - q; }
    }
    lg = res;
    dg = dres;
}

DSQ_DEV double dtrigamma(double x) {
    if (x != x) return x;
    if (x <= 0.0) return dnan();
    if (x == kInf) return 0.0;
    double num = 0.0, den = 1.0, xs = x;
    bool shifted = false;
    for (int i = 0; i < 10; i++) {
        if (!__any(xs < 10.0)) break;
        if (xs < 10.0) {
            double d2 = xs * xs;
            num = __builtin_fma(num, d2, den); den = den * d2; xs = xs + 1.0; shifted = true;
        }
    }
    double rx = 1.0 / xs;
    double r2 = rx * rx;
#if DSQ_ISA_CORES
    double c = isa_stirling_trigamma(r2);
#else
    double c = -3617.0 / 510.0;
    c = __builtin_fma(c, r2, 7.0 / 6.0);
    c = __builtin_fma(c, r2, -691.0 / 2730.0);
    c = __builtin_fma(c, r2, 5.0 / 66.0);
    c = __builtin_fma(c, r2, -1.0 / 30.0);
    c = __builtin_fma(c, r2, 1.0 / 42.0);
    c = __builtin_fma(c, r2, -1.0 / 30.0);
    c = __builtin_fma(c, r2, 1.0 / 6.0);
#endif
    double res = rx + r2 * (0.5 + rx * c);
    if (shifted) res = res + num / den;
    return res;
}

// -------------------------------------------------------------------- pnorm
// 2 * pnorm(|z|, lower.tail = FALSE), the Wald p-value of R/core.R:1507.  R's pnorm (nmath/pnorm.c, not in the
// reference tree) is W. J. Cody's rational Chebyshev approximation (Math. Comp. 1969, Algorithm 715): three
// ranges, split exp(-y^2/2) for the tails.  Restated for the upper tail of y = |z| with the engine's exp.
DSQ_DEV double dpnorm_upper2(double z) {
    if (z != z) return z;
    const double y = __builtin_fabs(z);
    double upper;
    if (y <= 0.67448975) {
        double xnum = 0.0, xden = 0.0;
        if (y > 5.5511151231257827e-17) {
            const double xsq = y * y;
            xnum = 0.065682337918207449113 * xsq;
            xden = xsq;
            xnum = (xnum + 2.2352520354606839287) * xsq;  xden = (xden + 47.20258190468824187) * xsq;
            xnum = (xnum + 161.02823106855587881) * xsq;  xden = (xden + 976.09855173777669322) * xsq;
            xnum = (xnum + 1067.6894854603709582) * xsq;  xden = (xden + 10260.932208618978205) * xsq;
        }
        const double temp = y * (xnum + 18154.981253343561249) / (xden + 45507.789335026729956);
        upper = 0.5 - temp;
    } else if (y <= 5.656854249492380195206754896838) {
        double xnum = 1.0765576773720192317e-8 * y, xden = y;
        xnum = (xnum + 0.39894151208813466764) * y;  xden = (xden + 22.266688044328115691) * y;
        xnum = (xnum + 8.8831497943883759412) * y;   xden = (xden + 235.38790178262499861) * y;
        xnum = (xnum + 93.506656132177855979) * y;   xden = (xden + 1519.377599407554805) * y;
        xnum = (xnum + 597.27027639480026226) * y;   xden = (xden + 6485.558298266760755) * y;
        xnum = (xnum + 2494.5375852903726711) * y;   xden = (xden + 18615.571640885098091) * y;
        xnum = (xnum + 6848.1904505362823326) * y;   xden = (xden + 34900.952721145977266) * y;
        xnum = (xnum + 11602.651437647350124) * y;   xden = (xden + 38912.003286093271411) * y;
        const double temp = (xnum + 9842.7148383839780218) / (xden + 19685.429676859990727);
        const double xsq = __builtin_trunc(y * 16.0) / 16.0;
        const double del = (y - xsq) * (y + xsq);
        upper = dexp(-xsq * xsq * 0.5) * dexp(-del * 0.5) * temp;
    } else if (y < 38.5) {
        const double xsq = 1.0 / (y * y);
        double xnum = 0.02307344176494017303 * xsq, xden = xsq;
        xnum = (xnum + 0.21589853405795699) * xsq;       xden = (xden + 1.28426009614491121) * xsq;
        xnum = (xnum + 0.1274011611602473639) * xsq;     xden = (xden + 0.468238212480865118) * xsq;
        xnum = (xnum + 0.022235277870649807) * xsq;      xden = (xden + 0.0659881378689285515) * xsq;
        xnum = (xnum + 0.001421619193227893466) * xsq;   xden = (xden + 0.00378239633202758244) * xsq;
        double temp = xsq * (xnum + 2.9112874951168792e-5) / (xden + 7.29751555083966205e-5);
        temp = (0.398942280401432677939946059934 - temp) / y;
        const double ysq = __builtin_trunc(y * 16.0) / 16.0;
        const double del = (y - ysq) * (y + ysq);
        upper = dexp(-ysq * ysq * 0.5) * dexp(-del * 0.5) * temp;
    } else {
        upper = 0.0;
    }
    return 2.0 * upper;
}

// ----------------------------------------------------------------- stirlerr
// log(n!) - log(sqrt(2 pi n) (n/e)^n); table at half-integers <= 15.
__device__ const double kSferrHalves[31] = {
    0.0,
    0.1534264097200273452913839393,   0.08106146679532725821967026359,
    0.05481412105191765389613870235,  0.04134069595540929409382208141,
    0.03316287351993628748511050974,  0.02767792568499833914878929275,
    0.02374616365629749597133027909,  0.02079067210376509311152277177,
    0.01848845053267318523077935748,  0.01664469118982119216319486537,
    0.01513497322191737887351383688,  0.01387612882307074799874572702,
    0.01281046524292022692425065528,  0.01189670994589177009505572412,
    0.0111045597582069173266307552,   0.01041126526197209649747856713,
    0.009799416126158803298390373402, 0.009255462182712732917728636633,
    0.008768700134139385462955047269, 0.00833056343336287125646931866,
    0.007934114564314020547249562491, 0.007573675487951840794972024212,
    0.007244554301320383179546196602, 0.006942840107209529865664152663,
    0.006665247032707682442356180895, 0.006408994188004207068439631083,
    0.006171712263039457647534604798, 0.005951370112758847735624416046,
    0.005746216513010115682026102477, 0.00555473355196280137103868996};

constexpr double kS0 = 1.0 / 12.0;
constexpr double kS1 = 1.0 / 360.0;
constexpr double kS2 = 1.0 / 1260.0;
constexpr double kS3 = 1.0 / 1680.0;
constexpr double kS4 = 1.0 / 1188.0;

DSQ_DEV double dstirlerr(double n) {
    if (n <= 15.0) {
        double nn = n + n;
        int inn = (int)nn;
        if (nn == (double)inn) return kSferrHalves[inn];
        return dlgamma(n + 1.0) - (n + 0.5) * dlog(n) + n - kLnSqrt2Pi;
    }
    double nn = n * n;
    if (n > 500.0) return (kS0 - kS1 / nn) / n;
    if (n > 80.0) return (kS0 - (kS1 - kS2 / nn) / nn) / n;
    if (n > 35.0) return (kS0 - (kS1 - (kS2 - kS3 / nn) / nn) / nn) / n;
    return (kS0 - (kS1 - (kS2 - (kS3 - kS4 / nn) / nn) / nn) / nn) / n;
}
// the same for an argument the CALLER knows to be below 1e100 in every active lane (dnbinom_mu_log's range guard):
// above 15 the divisors and quotients are normal numbers far from the ends of the exponent range and the scaling-free
// division gives the same (correctly rounded) quotients -- 2 to 5 divisions per call, and a wave whose lanes fall into
// several ranges runs all of them.  (Used by the density only: in fitBeta's once-per-gene constants the plain form
// measured faster -- the compiler shares the refinement of 1 / nn between its divisions.)
DSQ_DEV double dstirlerr_n(double n) {
    if (n <= 15.0) return dstirlerr(n);
    double nn = n * n;
    if (n > 500.0) return ddiv_n(kS0 - ddiv_n(kS1, nn), n);
    if (n > 80.0) return ddiv_n(kS0 - ddiv_n(kS1 - ddiv_n(kS2, nn), nn), n);
    if (n > 35.0) return ddiv_n(kS0 - ddiv_n(kS1 - ddiv_n(kS2 - ddiv_n(kS3, nn), nn), nn), n);
    return ddiv_n(kS0 - ddiv_n(kS1 - ddiv_n(kS2 - ddiv_n(kS3 - ddiv_n(kS4, nn), nn), nn), nn), n);
}

// ---------------------------------------------------------------------- bd0
// ej / (2j+1): IEEE division by a small odd constant d done as q0 = x*rc, r = fma(-q0,d,x),
// q = fma(r,rc,q0) with rc = RN(1/d).  With the exact residual this returns the correctly
// rounded quotient (Markstein's final-rounding theorem; checked against x/d on 1.3e9 random
// x for every odd d <= 131), i.e. the SAME bits as the division in the oracle, at 3 instead
// of ~11 instructions.  Outside a safe exponent range (or past the table) the true division runs.
// fastdiv: the caller vouches (dnbinom_mu_log's guard) that x, np and x / np are normal numbers well inside the exponent
// range in every active lane: the two divisions then take the scaling-free form (same quotients)
DSQ_DEV double dbd0(double x, double np, bool fastdiv = false) {
    if (!dfinite(x) || !dfinite(np) || np == 0.0) return dnan();
    if (__builtin_fabs(x - np) < 0.1 * (x + np)) {
        double v = fastdiv ? ddiv_n(x - np, x + np) : (x - np) / (x + np);
        double s = (x - np) * v;
        if (__builtin_fabs(s) < kDblMin) return s;
        double ej = 2.0 * x * v;
        v = v * v;
        // Safe range for the fast quotient, tested once: |ej| only shrinks, by the factor v < 0.01
        // per step, so |ej| > 1e-150 and v > 1e-10 keep all 12 unrolled steps above 1e-270.
        const bool ok = (__builtin_fabs(ej) < 1e280) && (__builtin_fabs(ej) > 1e-150) && (v > 1e-10);
        int j = 1;
        if (!__any(!ok)) {
#pragma unroll
            for (int u = 1; u <= 12; u++) {      // fully unrolled: d and rc are literals
                ej = ej * v;
                const double d = (double)(2 * u + 1);
                const double rc = 1.0 / d;        // folded at compile time = RN(1/d)
                double q0 = ej * rc;
                double rem = __builtin_fma(-q0, d, ej);
                double quo = __builtin_fma(rem, rc, q0);
                double s1 = s + quo;
                if (s1 == s) return s1;
                s = s1;
            }
            j = 13;
        }
        for (; j < 1000; j++) {
            ej = ej * v;
            double s1 = s + ej / (double)((j << 1) + 1);
            if (s1 == s) return s1;
            s = s1;
        }
    }
    return x * dlog(fastdiv ? ddiv_n(x, np) : x / np) + np - x;
}

// st_x / lg_x: dstirlerr(x) and dlog(x) when the caller already holds them (x = size is one value per gene in
// nbinomLogLike: evaluated once per gene instead of once per sample -- the same function on the same argument, hence
// the same bits); NaN = not given.
DSQ_DEV double dbinom_raw_log(double x, double n, double p, double q, double st_x, double lg_x, bool fastdiv = false) {
    if (p == 0.0) return (x == 0.0) ? 0.0 : -kInf;
    if (q == 0.0) return (x == n) ? 0.0 : -kInf;
    if (x == 0.0) {
        if (n == 0.0) return 0.0;
        return (p < 0.1) ? -dbd0(n, n * q) - n * p : n * dlog(q);
    }
    if (x == n) {
        return (q < 0.1) ? -dbd0(n, n * p) - n * q : n * dlog(p);
    }
    if (x < 0.0 || x > n) return -kInf;
    double lc = (fastdiv ? dstirlerr_n(n) : dstirlerr(n)) - st_x - (fastdiv ? dstirlerr_n(n - x) : dstirlerr(n - x)) -
                dbd0(x, n * p, fastdiv) - dbd0(n - x, n * q, fastdiv);
    double lf = kLn2Pi + lg_x + dlog1p(fastdiv ? ddiv_n(-x, n) : -x / n);
    return lc - 0.5 * lf;
}
DSQ_DEV double dbinom_raw_log(double x, double n, double p, double q) {
    return dbinom_raw_log(x, n, p, q, dstirlerr(x), dlog(x));
}

DSQ_DEV double dpois_raw_log(double x, double lambda) {
    if (lambda == 0.0) return (x == 0.0) ? 0.0 : -kInf;
    if (!dfinite(lambda)) return -kInf;
    if (x < 0.0) return -kInf;
    if (x <= lambda * kDblMin) return -lambda;
    if (lambda < x * kDblMin) {
        if (!dfinite(x)) return -kInf;
        return -lambda + x * dlog(lambda) - dlgamma(x + 1.0);
    }
    return -0.5 * dlog(kTwoPi * x) + (-dstirlerr(x) - dbd0(x, lambda));
}

// log NB(x; size, mu) for a non-negative integer-valued x (saddle-point form).  st_size / lg_size: see dbinom_raw_log.
DSQ_DEV double dnbinom_mu_log(double x, double size, double mu, double st_size, double lg_size) {
    if (x != x || size != size || mu != mu) return x + size + mu;
    if (mu < 0.0 || size < 0.0) return dnan();
    if (x < 0.0 || !dfinite(x)) return -kInf;
    if (x == 0.0 && size == 0.0) return 0.0;
    if (!dfinite(size)) return dpois_raw_log(x, mu);
    if (x == 0.0)
        return size * (size < mu ? dlog(size / (size + mu)) : dlog1p(-mu / (size + mu)));
    if (x < 1e-10 * size) {
        double p = (size < mu ? dlog(size / (1.0 + size / mu)) : dlog(mu / (1.0 + mu / size)));
        return x * p - mu - dlgamma(x + 1.0) + dlog1p(x * (x - 1.0) / (2.0 * size));
    }
    // the common regime -- size, count and mean positive and between 1e-60 and 1e60 in every active lane: every
    // divisor and quotient below (down to n p ~ 1e-180, up to x / (n p) ~ 1e240) is a normal number inside the
    // exponent range, and the divisions take the
    // scaling-free form (dsq_math.hpp: ddiv_n -- the same correctly rounded quotients, 7 instead of 11 instructions;
    // a sample costs about a dozen of them)
    const bool ok = size > 1e-60 && size < 1e60 && mu > 1e-60 && mu < 1e60 && x < 1e60;
    if (!__any(!ok)) {
        const double spm = size + mu;
        const double p = ddiv_n(size, size + x);
        const double ans = dbinom_raw_log(size, x + size, ddiv_n(size, spm), ddiv_n(mu, spm), st_size, lg_size, true);
        return dlog(p) + ans;
    }
    double p = size / (size + x);
    double ans = dbinom_raw_log(size, x + size, size / (size + mu), mu / (size + mu), st_size, lg_size);
    return dlog(p) + ans;
}
// the per-gene constants of nbinomLogLike's density: valid inputs for the 5-argument form whatever `size` is (for a
// size the general branch is never reached with -- NaN, infinite, zero, negative -- they are not read)
DSQ_DEV void dnbinom_size_terms(double size, double &st_size, double &lg_size) {
    const bool ok = size > 0.0 && dfinite(size);
    st_size = dstirlerr(ok ? size : 1.0);
    lg_size = dlog(ok ? size : 1.0);
}

DSQ_DEV double dnbinom_mu_log(double x, double size, double mu) {
    double st_size, lg_size;
    dnbinom_size_terms(size, st_size, lg_size);
    return dnbinom_mu_log(x, size, mu, st_size, lg_size);
}

// ---- the closed split of log NB(y; size = 1/alpha, mu) ---------------------------------------
// On the saddle-point form of dnbinom_mu the two bd0 terms are -n [log1p(alpha mu) - log1p(alpha y)] - y [log y - log mu]
// (n = y + size), so   log f = K' + y log mu - n log1p(alpha mu),   K' = [saddle-point constants] + n log1p(alpha y) - y log y
// independent of mu (0 for y = 0).  fitBeta's deviance has used it since round 2 (with lg = log(mu / nf), the y log nf
// in its constants); nbinomLogLike takes it since round 4 (aux.hip) -- the sum over a row is K' once per gene (fitBeta's
// constants pass hands it over) plus one sweep with two logarithms per sample, instead of R's dnbinom_mu sample by sample.
// `fast`: alpha and size positive and finite; a sample outside the split (a count below 1e-10 size ...) keeps the full form.
DSQ_DEV bool cell_dev_closed(double y, double size, bool fast) {
    const double n = y + size;
    const bool gen = (y > 0.0) && dfinite(y) && !(y < 1e-10 * size) && (n != size) && dfinite(n);
    return fast && (y == 0.0 || gen);
}
DSQ_DEV bool nb_split_fast(double alpha, double size) { return (alpha > 0.0) && dfinite(alpha) && dfinite(size) && (size > 0.0); }
// y != 0 inside the split: K' = base + t  (fitBeta: K = base + (t + y log nf))
DSQ_DEV void nb_split_const(double y, double alpha, double size, double st_size, double &base, double &t) {
    // log(size/(size+y)) = -L, log1p(-size/n) = log y - log size - L, L = log1p(alpha y)
    const double n = y + size;
    const double L = dlog1p(alpha * y), ly = dlog(y);
    const double c0 = dstirlerr(n) - st_size - dstirlerr(n - size);
    base = -L + (c0 - 0.5 * (kLn2Pi + ly - L));
    t = n * L - y * ly;
}
// the mu-dependent part (the whole log density for a sample outside the split)
DSQ_DEV double nb_split_var(double y, double size, double alpha, double mu, bool fast) {
    if (cell_dev_closed(y, size, fast)) {
        const double am = alpha * mu, opm = 1.0 + am, rcp = 1.0 / opm;
        const double l1p = dlog(opm) + (am - (opm - 1.0)) * rcp;
        return (y == 0.0) ? -(size * l1p) : y * dlog(mu) - (y + size) * l1p;
    }
    return dnbinom_mu_log(y, size, mu);
}

}  // namespace dsq
