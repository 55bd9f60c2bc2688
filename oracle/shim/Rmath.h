/* oracle/shim/Rmath.h -- TEST INFRASTRUCTURE: the five Rmath entry points src/DESeq2.cpp calls, each
 * evaluated from its textbook definition in binary128 (libquadmath) and rounded once to double -- an
 * evaluation that shares nothing with oracle/orc_nmath.c's double-precision algorithms, so agreement
 * between the two is evidence, not tautology.  Slow; used only to build oracle/_ref.              */
#pragma once
#include <cmath>
#include <quadmath.h>

inline double R_pow_di(double x, int n) {       /* R's arithmetic.c: repeated squaring */
    double xn = 1.0;
    if (std::isnan(x)) return x;
    if (n != 0) {
        if (!std::isfinite(x)) return std::pow(x, (double)n);
        bool is_neg = (n < 0);
        if (is_neg) n = -n;
        for (;;) {
            if (n & 01) xn *= x;
            if (n >>= 1) x *= x; else break;
        }
        if (is_neg) xn = 1. / xn;
    }
    return xn;
}

#ifdef SHIM_DOUBLE_MATH
/* timing build (oracle/_ref/libdeseq2_ref_fast.so, bench.py's cpu_baseline): the same five entry points in
 * plain double via libm, so the CPU baseline is not slowed down by binary128 emulation.  Not used for parity. */
inline double Rf_lgammafn(double x) { return std::lgamma(x); }
inline double Rf_digamma(double x) {
    double acc = 0.0;
    while (x < 10.0) { acc -= 1.0 / x; x += 1.0; }
    double i = 1.0 / x, i2 = i * i;
    return acc + std::log(x) - 0.5 * i - i2 * (1.0 / 12 - i2 * (1.0 / 120 - i2 * (1.0 / 252 - i2 * (1.0 / 240 - i2 * (1.0 / 132)))));
}
inline double Rf_trigamma(double x) {
    double acc = 0.0;
    while (x < 10.0) { acc += 1.0 / (x * x); x += 1.0; }
    double i = 1.0 / x, i2 = i * i;
    return acc + i + 0.5 * i2 + i * i2 * (1.0 / 6 - i2 * (1.0 / 30 - i2 * (1.0 / 42 - i2 * (1.0 / 30 - i2 * (5.0 / 66)))));
}
inline double Rf_dnbinom_mu(double x, double size, double mu, int give_log) {
    double lp = std::lgamma(x + size) - std::lgamma(size) - std::lgamma(x + 1.0) - size * std::log1p(mu / size);
    if (x > 0) lp += x * (std::log(mu) - std::log(size + mu));
    return give_log ? lp : std::exp(lp);
}
#else
inline double Rf_lgammafn(double x) { return (double)lgammaq((__float128)x); }

/* digamma / trigamma for x > 0: shift up to x >= 40, then the asymptotic series */
inline __float128 shim_digammaq(__float128 x) {
    __float128 acc = 0;
    while (x < 40) { acc -= 1 / x; x += 1; }
    __float128 i = 1 / x, i2 = i * i;
    /* B_2k / (2k): 1/12, -1/120, 1/252, -1/240, 1/132, -691/32760, 1/12, -3617/8160, 43867/14364 */
    __float128 s = i2 * (1 / (__float128)12 - i2 * (1 / (__float128)120 - i2 * (1 / (__float128)252 - i2 * (1 / (__float128)240 -
                   i2 * (1 / (__float128)132 - i2 * ((__float128)691 / 32760 - i2 * (1 / (__float128)12 - i2 * ((__float128)3617 / 8160 -
                   i2 * ((__float128)43867 / 14364)))))))));
    return acc + logq(x) - i / 2 - s;
}
inline __float128 shim_trigammaq(__float128 x) {
    __float128 acc = 0;
    while (x < 40) { acc += 1 / (x * x); x += 1; }
    __float128 i = 1 / x, i2 = i * i;
    /* B_2k: 1/6, -1/30, 1/42, -1/30, 5/66, -691/2730, 7/6, -3617/510, 43867/798 */
    __float128 s = i * i2 * (1 / (__float128)6 - i2 * (1 / (__float128)30 - i2 * (1 / (__float128)42 - i2 * (1 / (__float128)30 -
                   i2 * ((__float128)5 / 66 - i2 * ((__float128)691 / 2730 - i2 * ((__float128)7 / 6 - i2 * ((__float128)3617 / 510 -
                   i2 * ((__float128)43867 / 798)))))))));
    return acc + i + i2 / 2 + s;
}
inline double Rf_digamma(double x) { return (double)shim_digammaq((__float128)x); }
inline double Rf_trigamma(double x) { return (double)shim_trigammaq((__float128)x); }

/* dnbinom(x, size, mu = mu, log): log Gamma(x+size) - log Gamma(size) - log x! + size log(size/(size+mu))
 * + x log(mu/(size+mu)) */
inline double Rf_dnbinom_mu(double x, double size, double mu, int give_log) {
    __float128 X = x, S = size, M = mu;
    __float128 lp = lgammaq(X + S) - lgammaq(S) - lgammaq(X + 1) + S * (logq(S) - logq(S + M));
    if (x > 0) lp += X * (logq(M) - logq(S + M));
    double r = (double)lp;
    return give_log ? r : std::exp(r);
}
#endif
