/*
 * ORACLE (test infrastructure only -- see orc_nmath.h).
 *
 * Build with -ffp-contract=off: every rounding below is intentional.  The HIP
 * device math (deseq2_amd/csrc/dsq_math.hpp) is an independent implementation
 * of the same operation sequences; tests/test_gpu_math.py checks bit equality.
 */
#include "orc_nmath.h"
#include <math.h>
#include <stdint.h>
#include <string.h>

static inline uint64_t d2u(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }
static inline double u2d(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }

#define ORC_LN2_HI   6.93147180369123816490e-01 /* 0x3fe62e42fee00000 */
#define ORC_LN2_LO   1.90821492927058770002e-10 /* 0x3dea39ef35793c76 */
#define ORC_INV_LN2  1.44269504088896338700e+00
#define ORC_LN_SQRT_2PI 0.918938533204672741780329736406
#define ORC_LN_2PI      1.837877066409345483560659472811

/* ------------------------------------------------------------------ exp ----
 * Cody-Waite reduction x = k ln2 + r, |r| <= ln2/2; degree-13 Taylor polynomial
 * (truncation 4e-18); result 1 + (r + r^2 p(r) + rounding error of r); scaled by
 * 2^k in two exact-or-once-rounded steps.  Max error measured < 1 ulp.          */
double orc_exp(double x) {
    if (x != x) return x;
    if (x > 709.782712893384) return INFINITY;
    if (x < -745.1332191019412) return 0.0;
    double kf = rint(x * ORC_INV_LN2);
    double hi = fma(-kf, ORC_LN2_HI, x);
    double lo = kf * ORC_LN2_LO;
    double r = hi - lo;
    double rerr = (hi - r) - lo;
    double p = 1.0 / 6227020800.0;          /* 1/13! */
    p = fma(p, r, 1.0 / 479001600.0);       /* 1/12! */
    p = fma(p, r, 1.0 / 39916800.0);
    p = fma(p, r, 1.0 / 3628800.0);
    p = fma(p, r, 1.0 / 362880.0);
    p = fma(p, r, 1.0 / 40320.0);
    p = fma(p, r, 1.0 / 5040.0);
    p = fma(p, r, 1.0 / 720.0);
    p = fma(p, r, 1.0 / 120.0);
    p = fma(p, r, 1.0 / 24.0);
    p = fma(p, r, 1.0 / 6.0);
    p = fma(p, r, 0.5);
    double r2 = r * r;
    double t = fma(r2, p, r);
    t = t + rerr;
    double y = 1.0 + t;
    int k = (int)kf;
    int k1 = k >> 1;              /* arithmetic shift: floor(k/2) */
    int k2 = k - k1;
    double s1 = u2d((uint64_t)(k1 + 1023) << 52);
    double s2 = u2d((uint64_t)(k2 + 1023) << 52);
    return (y * s1) * s2;
}

/* ------------------------------------------------------------------ log ----
 * fdlibm/musl-style: x = 2^k m, m in [sqrt(2)/2, sqrt(2)); f = m-1; s = f/(2+f);
 * log(1+f) = f - f^2/2 + s (f^2/2 + R(s^2)); error < 1 ulp.                      */
#define LG1 6.666666666666735130e-01
#define LG2 3.999999999940941908e-01
#define LG3 2.857142874366239149e-01
#define LG4 2.222219843214978396e-01
#define LG5 1.818357216161805012e-01
#define LG6 1.531383769920937332e-01
#define LG7 1.479819860511658591e-01

/* (round 4: polynomial and closing sum as fused multiply-adds -- the sequence of csrc/dsq_math.hpp: log_core) */
static inline double log_core(double f, double dk, double c) {
    double hfsq = 0.5 * f * f;
    double s = f / (2.0 + f);
    double z = s * s;
    double w = z * z;
    double t1 = w * fma(w, fma(w, LG6, LG4), LG2);
    double t2 = z * fma(w, fma(w, fma(w, LG7, LG5), LG3), LG1);
    double R = t2 + t1;
    double u = fma(s, hfsq + R, fma(dk, ORC_LN2_LO, c));
    return fma(dk, ORC_LN2_HI, (u - hfsq) + f);
}

double orc_log(double x) {
    if (x != x) return x;
    if (x < 0.0) return NAN;
    if (x == 0.0) return -INFINITY;
    if (x == INFINITY) return x;
    int k = 0;
    if (x < 2.2250738585072014e-308) { x *= 18014398509481984.0; k = -54; }
    uint64_t ix = d2u(x);
    ix += (uint64_t)(0x3ff00000u - 0x3fe6a09eu) << 32;
    k += (int)(ix >> 52) - 0x3ff;
    ix = (ix & 0x000fffffffffffffULL) + ((uint64_t)0x3fe6a09eu << 32);
    double f = u2d(ix) - 1.0;
    return log_core(f, (double)k, 0.0);
}

/* ---------------------------------------------------------------- log1p ---- */
double orc_log1p(double x) {
    if (x != x) return x;
    if (x < -1.0) return NAN;
    if (x == -1.0) return -INFINITY;
    if (x == INFINITY) return x;
    if (fabs(x) < 1.1102230246251565e-16) return x;       /* |x| < 2^-53 */
    if (x > -0.2928932188134524 && x < 0.41421356237309503) {
        return log_core(x, 0.0, 0.0);
    }
    double u = 1.0 + x;
    uint64_t iu = d2u(u);
    iu += (uint64_t)(0x3ff00000u - 0x3fe6a09eu) << 32;
    int k = (int)(iu >> 52) - 0x3ff;
    double c = 0.0;
    if (k < 54) {
        c = (k >= 2) ? 1.0 - (u - x) : x - (u - 1.0);
        c = c / u;
    }
    iu = (iu & 0x000fffffffffffffULL) + ((uint64_t)0x3fe6a09eu << 32);
    double f = u2d(iu) - 1.0;
    return log_core(f, (double)k, c);
}

/* --------------------------------------------------------------- lgamma ----
 * R's lgammafn (nmath/lgamma.c) for x > 10 is
 *     M_LN_SQRT_2PI + (x - 0.5) * log(x) - x + lgammacor(x)
 * with lgammacor = the Stirling correction sum B_2k / (2k(2k-1) x^(2k-1)).  That
 * structure is kept (same operation order, correction evaluated as a polynomial in
 * 1/x^2 instead of R's Chebyshev fit).  For 0 < x < 10 R takes log(gamma(x)); here
 * the argument is shifted up by the recurrence lgamma(x) = lgamma(x+n) - log(x(x+1)..)
 * Absolute error a few 1e-16 * max(1,|result|), as R.  Domain: x > 0.            */
double orc_lgamma(double x) {
    if (x != x) return x;
    if (x <= 0.0) return (x == 0.0) ? INFINITY : NAN;
    if (x == INFINITY) return x;
    double prod = 1.0, xs = x;
    int shifted = 0;
    for (int i = 0; i < 10; i++) {
        if (xs < 10.0) { prod = prod * xs; xs = xs + 1.0; shifted = 1; }
    }
    double lx = orc_log(xs);
    double rx = 1.0 / xs;
    double r2 = rx * rx;
    double c = -3617.0 / 122400.0;
    c = fma(c, r2, 1.0 / 156.0);
    c = fma(c, r2, -691.0 / 360360.0);
    c = fma(c, r2, 1.0 / 1188.0);
    c = fma(c, r2, -1.0 / 1680.0);
    c = fma(c, r2, 1.0 / 1260.0);
    c = fma(c, r2, -1.0 / 360.0);
    c = fma(c, r2, 1.0 / 12.0);
    double cor = c * rx;
    double res = ORC_LN_SQRT_2PI + (xs - 0.5) * lx - xs + cor;
    if (shifted) res = res - orc_log(prod);
    return res;
}

/* -------------------------------------------------------------- digamma ----
 * R's digamma/trigamma (nmath/polygamma.c, Amos' dpsifn) = asymptotic series
 * above a threshold + downward recurrence below it.  Restated with threshold 10
 * and the recurrence sum kept as one fraction num/den (a single division).     */
double orc_digamma(double x) {
    if (x != x) return x;
    if (x <= 0.0) return NAN;
    if (x == INFINITY) return x;
    double num = 0.0, den = 1.0, xs = x;
    int shifted = 0;
    for (int i = 0; i < 10; i++) {
        if (xs < 10.0) { num = fma(num, xs, den); den = den * xs; xs = xs + 1.0; shifted = 1; }
    }
    double lx = orc_log(xs);
    double rx = 1.0 / xs;
    double r2 = rx * rx;
    double c = -3617.0 / 8160.0;
    c = fma(c, r2, 1.0 / 12.0);
    c = fma(c, r2, -691.0 / 32760.0);
    c = fma(c, r2, 1.0 / 132.0);
    c = fma(c, r2, -1.0 / 240.0);
    c = fma(c, r2, 1.0 / 252.0);
    c = fma(c, r2, -1.0 / 120.0);
    c = fma(c, r2, 1.0 / 12.0);
    double res = (lx - 0.5 * rx) - c * r2;
    if (shifted) res = res - num / den;
    return res;
}

double orc_trigamma(double x) {
    if (x != x) return x;
    if (x <= 0.0) return NAN;
    if (x == INFINITY) return 0.0;
    double num = 0.0, den = 1.0, xs = x;
    int shifted = 0;
    for (int i = 0; i < 10; i++) {
        if (xs < 10.0) {
            double d2 = xs * xs;
            num = fma(num, d2, den); den = den * d2; xs = xs + 1.0; shifted = 1;
        }
    }
    double rx = 1.0 / xs;
    double r2 = rx * rx;
    double c = -3617.0 / 510.0;
    c = fma(c, r2, 7.0 / 6.0);
    c = fma(c, r2, -691.0 / 2730.0);
    c = fma(c, r2, 5.0 / 66.0);
    c = fma(c, r2, -1.0 / 30.0);
    c = fma(c, r2, 1.0 / 42.0);
    c = fma(c, r2, -1.0 / 30.0);
    c = fma(c, r2, 1.0 / 6.0);
    /* 1/x + 1/(2x^2) + (1/x^3) c */
    double res = rx + r2 * (0.5 + rx * c);
    if (shifted) res = res + num / den;
    return res;
}

/* ------------------------------------------------------------- stirlerr ----
 * R nmath/stirlerr.c (Loader's saddle-point error term), pre-4.4 form:
 *   stirlerr(n) = log(n!) - log( sqrt(2 pi n) (n/e)^n )
 * table for 2n integer <= 30, lgamma form for other n <= 15, series above.     */
static const double orc_sferr_halves[31] = {
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
    0.005746216513010115682026102477, 0.00555473355196280137103868996
};
#define ORC_S0 (1.0 / 12.0)
#define ORC_S1 (1.0 / 360.0)
#define ORC_S2 (1.0 / 1260.0)
#define ORC_S3 (1.0 / 1680.0)
#define ORC_S4 (1.0 / 1188.0)

double orc_stirlerr(double n) {
    double nn;
    if (n <= 15.0) {
        nn = n + n;
        if (nn == (double)(int)nn) return orc_sferr_halves[(int)nn];
        return orc_lgamma(n + 1.0) - (n + 0.5) * orc_log(n) + n - ORC_LN_SQRT_2PI;
    }
    nn = n * n;
    if (n > 500.0) return (ORC_S0 - ORC_S1 / nn) / n;
    if (n > 80.0)  return (ORC_S0 - (ORC_S1 - ORC_S2 / nn) / nn) / n;
    if (n > 35.0)  return (ORC_S0 - (ORC_S1 - (ORC_S2 - ORC_S3 / nn) / nn) / nn) / n;
    return (ORC_S0 - (ORC_S1 - (ORC_S2 - (ORC_S3 - ORC_S4 / nn) / nn) / nn) / nn) / n;
}

/* ------------------------------------------------------------------ bd0 ----
 * R nmath/bd0.c: x log(x/np) + np - x, by Taylor series when x ~ np.           */
double orc_bd0(double x, double np) {
    if (!isfinite(x) || !isfinite(np) || np == 0.0) return NAN;
    if (fabs(x - np) < 0.1 * (x + np)) {
        double v = (x - np) / (x + np);
        double s = (x - np) * v;
        if (fabs(s) < 2.2250738585072014e-308) return s;
        double ej = 2.0 * x * v;
        v = v * v;
        for (int j = 1; j < 1000; j++) {
            ej = ej * v;
            double s1 = s + ej / (double)((j << 1) + 1);
            if (s1 == s) return s1;
            s = s1;
        }
    }
    return x * orc_log(x / np) + np - x;
}

/* R nmath/dbinom.c dbinom_raw(x, n, p, q, give_log = TRUE) */
static double orc_dbinom_raw_log(double x, double n, double p, double q) {
    double lf, lc;
    if (p == 0.0) return (x == 0.0) ? 0.0 : -INFINITY;
    if (q == 0.0) return (x == n) ? 0.0 : -INFINITY;
    if (x == 0.0) {
        if (n == 0.0) return 0.0;
        lc = (p < 0.1) ? -orc_bd0(n, n * q) - n * p : n * orc_log(q);
        return lc;
    }
    if (x == n) {
        lc = (q < 0.1) ? -orc_bd0(n, n * p) - n * q : n * orc_log(p);
        return lc;
    }
    if (x < 0.0 || x > n) return -INFINITY;
    lc = orc_stirlerr(n) - orc_stirlerr(x) - orc_stirlerr(n - x)
         - orc_bd0(x, n * p) - orc_bd0(n - x, n * q);
    lf = ORC_LN_2PI + orc_log(x) + orc_log1p(-x / n);
    return lc - 0.5 * lf;
}

/* R nmath/dpois.c dpois_raw(x, lambda, give_log = TRUE) (size = Inf limit) */
static double orc_dpois_raw_log(double x, double lambda) {
    if (lambda == 0.0) return (x == 0.0) ? 0.0 : -INFINITY;
    if (!isfinite(lambda)) return -INFINITY;
    if (x < 0.0) return -INFINITY;
    if (x <= lambda * 2.2250738585072014e-308) return -lambda;
    if (lambda < x * 2.2250738585072014e-308) {
        if (!isfinite(x)) return -INFINITY;
        return -lambda + x * orc_log(lambda) - orc_lgamma(x + 1.0);
    }
    return -0.5 * orc_log(6.283185307179586476925286766559 * x)
           + (-orc_stirlerr(x) - orc_bd0(x, lambda));
}

/* R nmath/dnbinom.c dnbinom_mu(x, size, mu, give_log = TRUE), R >= 3.x/4.x form.
 * x must be a non-negative integer value (counts); callers guarantee that.     */
double orc_dnbinom_mu_log(double x, double size, double mu) {
    if (x != x || size != size || mu != mu) return x + size + mu;
    if (mu < 0.0 || size < 0.0) return NAN;
    if (x < 0.0 || !isfinite(x)) return -INFINITY;
    if (x == 0.0 && size == 0.0) return 0.0;
    if (!isfinite(size)) return orc_dpois_raw_log(x, mu);
    if (x == 0.0)
        return size * (size < mu ? orc_log(size / (size + mu))
                                 : orc_log1p(-mu / (size + mu)));
    if (x < 1e-10 * size) {
        double p = (size < mu ? orc_log(size / (1.0 + size / mu))
                              : orc_log(mu / (1.0 + mu / size)));
        return x * p - mu - orc_lgamma(x + 1.0) + orc_log1p(x * (x - 1.0) / (2.0 * size));
    } else {
        double p = size / (size + x);
        double ans = orc_dbinom_raw_log(size, x + size, size / (size + mu), mu / (size + mu));
        return orc_log(p) + ans;
    }
}

/* 2 * pnorm(|z|, lower.tail = FALSE): the Wald p-value (R/core.R:1507).  R's pnorm (nmath/pnorm.c) is Cody's
 * rational Chebyshev approximation of the normal distribution function (W. J. Cody, Math. Comp. 23 (1969); ACM
 * Algorithm 715): |z| <= 0.67448975 by a (4,4) rational in z^2, <= sqrt(32) by an (8,8) rational times
 * exp(-z^2/2) with the argument split at 1/16, beyond by an asymptotic (5,5) rational in 1/z^2.  Upper tail
 * only, restated with orc_exp.                                                                                 */
double orc_pnorm_upper2(double z) {
    if (z != z) return z;
    const double y = fabs(z);
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
        const double xsq = trunc(y * 16.0) / 16.0;
        const double del = (y - xsq) * (y + xsq);
        upper = orc_exp(-xsq * xsq * 0.5) * orc_exp(-del * 0.5) * temp;
    } else if (y < 38.5) {
        const double xsq = 1.0 / (y * y);
        double xnum = 0.02307344176494017303 * xsq, xden = xsq;
        xnum = (xnum + 0.21589853405795699) * xsq;       xden = (xden + 1.28426009614491121) * xsq;
        xnum = (xnum + 0.1274011611602473639) * xsq;     xden = (xden + 0.468238212480865118) * xsq;
        xnum = (xnum + 0.022235277870649807) * xsq;      xden = (xden + 0.0659881378689285515) * xsq;
        xnum = (xnum + 0.001421619193227893466) * xsq;   xden = (xden + 0.00378239633202758244) * xsq;
        double temp = xsq * (xnum + 2.9112874951168792e-5) / (xden + 7.29751555083966205e-5);
        temp = (0.398942280401432677939946059934 - temp) / y;
        const double ysq = trunc(y * 16.0) / 16.0;
        const double del = (y - ysq) * (y + ysq);
        upper = orc_exp(-ysq * ysq * 0.5) * orc_exp(-del * 0.5) * temp;
    } else {
        upper = 0.0;
    }
    return 2.0 * upper;
}

void orc_vec_unary(int op, const double *in, double *out, long n) {
    for (long i = 0; i < n; i++) {
        double x = in[i], r;
        switch (op) {
        case 0: r = orc_exp(x); break;
        case 1: r = orc_log(x); break;
        case 2: r = orc_log1p(x); break;
        case 3: r = orc_lgamma(x); break;
        case 4: r = orc_digamma(x); break;
        case 5: r = orc_trigamma(x); break;
        case 6: r = orc_stirlerr(x); break;
        case 9: r = orc_pnorm_upper2(x); break;
        default: r = NAN;
        }
        out[i] = r;
    }
}

void orc_vec_dnbinom_mu_log(const double *x, const double *size, const double *mu,
                            double *out, long n) {
    for (long i = 0; i < n; i++) out[i] = orc_dnbinom_mu_log(x[i], size[i], mu[i]);
}
