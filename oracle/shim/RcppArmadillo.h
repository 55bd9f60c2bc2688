/* oracle/shim/RcppArmadillo.h -- TEST INFRASTRUCTURE.  A minimal stand-in for the parts of Rcpp
 * (sugar on NumericMatrix rows, NumericVector/IntegerVector, List::create, as<>) and Armadillo
 * (dense mat/vec expressions, det/inv/solve/qr_econ) that the reference's src/DESeq2.cpp uses, so that
 * THAT FILE compiles unchanged, from where it lies under /root/reference, into oracle/_ref/ (see
 * oracle/Makefile target `ref`).  The control flow, formulas, clamps and convergence rules executed are
 * then the reference's own; what sits underneath them is this file's plain dense linear algebra
 * (LU with partial pivoting, Householder QR) and oracle/shim/Rmath.h's special functions (evaluated in
 * binary128), i.e. not bit-identical to LAPACK / R's nmath but accurate to a few ulp.
 * Everything is eager and unoptimised on purpose.  Nothing here is shipped or measured as product. */
#pragma once
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "Rmath.h"

/* ------------------------------------------------------------------ SEXP stand-in */
struct SexpRec {
    std::vector<double> d;   /* REALSXP / LGLSXP payload */
    std::vector<int> iv;     /* INTSXP payload */
    int nrow = 0, ncol = 0;  /* ncol == 0: plain vector */
    bool is_int = false;
};
typedef SexpRec *SEXP;

inline SEXP shim_alloc() {
    static std::vector<std::unique_ptr<SexpRec>> arena;
    arena.emplace_back(new SexpRec());
    return arena.back().get();
}

/* ------------------------------------------------------------------ Armadillo stand-in */
namespace arma {
typedef unsigned long long uword;
typedef std::vector<uword> uvec;

struct span {
    uword a, b;
    span(uword a_, uword b_) : a(a_), b(b_) {}
};

class Mat;
class Col;
struct UMat { uword n_rows, n_cols; std::vector<unsigned char> v; };
struct UCol { std::vector<unsigned char> v; };

class Mat {
  public:
    uword n_rows = 0, n_cols = 0, n_elem = 0;
    std::vector<double> mem;   /* column-major */
    Mat() {}
    Mat(uword r, uword c) : n_rows(r), n_cols(c), n_elem(r * c), mem(r * c, 0.0) {}
    double &operator()(uword i) { return mem[i]; }
    double operator()(uword i) const { return mem[i]; }
    double &operator()(uword i, uword j) { return mem[i + n_rows * j]; }
    double operator()(uword i, uword j) const { return mem[i + n_rows * j]; }

    Mat t() const {
        Mat o(n_cols, n_rows);
        for (uword j = 0; j < n_cols; j++)
            for (uword i = 0; i < n_rows; i++) o(j, i) = (*this)(i, j);
        return o;
    }
    Mat i() const;

    struct RowProxy {
        Mat &m; uword r;
        operator Mat() const {
            Mat o(1, m.n_cols);
            for (uword j = 0; j < m.n_cols; j++) o(0, j) = m(r, j);
            return o;
        }
        Mat t() const { return Mat(*this).t(); }
        RowProxy &operator=(const Mat &v) {
            if (v.n_elem != m.n_cols) { std::fprintf(stderr, "shim: row assignment size mismatch\n"); std::abort(); }
            for (uword j = 0; j < m.n_cols; j++) m(r, j) = v.mem[j];
            return *this;
        }
    };
    RowProxy row(uword r) { return RowProxy{*this, r}; }

    struct SpanProxy {
        Mat &m; span s;
        SpanProxy &operator=(const Mat &v) {
            if (v.n_elem != s.b - s.a + 1) { std::fprintf(stderr, "shim: span assignment size mismatch\n"); std::abort(); }
            for (uword k = 0; k < v.n_elem; k++) m.mem[s.a + k] = v.mem[k];
            return *this;
        }
    };
    SpanProxy operator()(const span &s) { return SpanProxy{*this, s}; }

    Mat rows(const uvec &idx) const {
        Mat o(idx.size(), n_cols);
        for (uword j = 0; j < n_cols; j++)
            for (uword k = 0; k < idx.size(); k++) o(k, j) = (*this)(idx[k], j);
        return o;
    }
    Mat cols(const uvec &idx) const {
        Mat o(n_rows, idx.size());
        for (uword k = 0; k < idx.size(); k++)
            for (uword i = 0; i < n_rows; i++) o(i, k) = (*this)(i, idx[k]);
        return o;
    }

    struct EachCol { const Mat &m; };
    EachCol each_col() const { return EachCol{*this}; }

    double max(uword &idx) const {      /* first maximum, like arma::Mat::max(uword&) */
        idx = 0;
        double best = mem[0];
        for (uword k = 1; k < n_elem; k++)
            if (mem[k] > best) { best = mem[k]; idx = k; }
        return best;
    }
};

class Col : public Mat {
  public:
    Col() {}
    explicit Col(uword n) : Mat(n, 1) {}
    Col(const Mat &m) : Mat(m) { n_rows = m.n_elem; n_cols = m.n_elem ? 1 : 0; }
    Col(const std::vector<double> &v) : Mat(v.size(), 1) { mem = v; }
    Col &operator=(const Mat &m) {
        mem = m.mem; n_elem = m.n_elem; n_rows = m.n_elem; n_cols = m.n_elem ? 1 : 0;
        return *this;
    }
    using Mat::operator();
    Col operator()(const uvec &idx) const {
        Col o(idx.size());
        for (uword k = 0; k < idx.size(); k++) o.mem[k] = mem[idx[k]];
        return o;
    }
};
typedef Mat mat;
typedef Col vec;
typedef Col colvec;

inline void shim_same(const Mat &a, const Mat &b, const char *op) {
    if (a.n_rows != b.n_rows || a.n_cols != b.n_cols) {
        std::fprintf(stderr, "shim: %s on %llux%llu and %llux%llu\n", op, a.n_rows, a.n_cols, b.n_rows, b.n_cols);
        std::abort();
    }
}
#define SHIM_EW(OP, NAME)                                                                         \
    inline Mat operator OP(const Mat &a, const Mat &b) {                                          \
        shim_same(a, b, NAME);                                                                    \
        Mat o(a.n_rows, a.n_cols);                                                                \
        for (uword k = 0; k < a.n_elem; k++) o.mem[k] = a.mem[k] OP b.mem[k];                     \
        return o;                                                                                 \
    }                                                                                             \
    inline Mat operator OP(const Mat &a, double s) {                                              \
        Mat o(a.n_rows, a.n_cols);                                                                \
        for (uword k = 0; k < a.n_elem; k++) o.mem[k] = a.mem[k] OP s;                            \
        return o;                                                                                 \
    }                                                                                             \
    inline Mat operator OP(double s, const Mat &a) {                                              \
        Mat o(a.n_rows, a.n_cols);                                                                \
        for (uword k = 0; k < a.n_elem; k++) o.mem[k] = s OP a.mem[k];                            \
        return o;                                                                                 \
    }
SHIM_EW(+, "+")
SHIM_EW(-, "-")
SHIM_EW(/, "/")
#undef SHIM_EW
/* Schur product */
inline Mat operator%(const Mat &a, const Mat &b) {
    shim_same(a, b, "%");
    Mat o(a.n_rows, a.n_cols);
    for (uword k = 0; k < a.n_elem; k++) o.mem[k] = a.mem[k] * b.mem[k];
    return o;
}
inline Mat operator%(const Mat::EachCol &e, const Mat &v) {
    const Mat &m = e.m;
    if (v.n_elem != m.n_rows) { std::fprintf(stderr, "shim: each_col() %% size mismatch\n"); std::abort(); }
    Mat o(m.n_rows, m.n_cols);
    for (uword j = 0; j < m.n_cols; j++)
        for (uword i = 0; i < m.n_rows; i++) o(i, j) = m(i, j) * v.mem[i];
    return o;
}
inline Mat operator*(const Mat &a, double s) { Mat o = a; for (auto &x : o.mem) x *= s; return o; }
inline Mat operator*(double s, const Mat &a) { Mat o = a; for (auto &x : o.mem) x = s * x; return o; }
inline Mat operator*(const Mat &a, const Mat &b) {
    if (a.n_cols != b.n_rows) { std::fprintf(stderr, "shim: matmul %llux%llu * %llux%llu\n", a.n_rows, a.n_cols, b.n_rows, b.n_cols); std::abort(); }
    Mat o(a.n_rows, b.n_cols);
    for (uword j = 0; j < b.n_cols; j++)
        for (uword i = 0; i < a.n_rows; i++) {
            double acc = 0.0;
            for (uword k = 0; k < a.n_cols; k++) acc += a(i, k) * b(k, j);
            o(i, j) = acc;
        }
    return o;
}

template <typename F> inline Mat shim_map(const Mat &a, F f) { Mat o = a; for (auto &x : o.mem) x = f(x); return o; }
inline Mat sqrt(const Mat &a) { return shim_map(a, [](double x) { return std::sqrt(x); }); }
inline Mat exp(const Mat &a) { return shim_map(a, [](double x) { return std::exp(x); }); }
inline Mat log(const Mat &a) { return shim_map(a, [](double x) { return std::log(x); }); }
inline Mat abs(const Mat &a) { return shim_map(a, [](double x) { return std::fabs(x); }); }
inline Col abs(const Col &a) { return Col(shim_map(a, [](double x) { return std::fabs(x); })); }

inline UMat operator>(const Mat &a, double s) {
    UMat u{a.n_rows, a.n_cols, std::vector<unsigned char>(a.n_elem)};
    for (uword k = 0; k < a.n_elem; k++) u.v[k] = a.mem[k] > s;
    return u;
}
inline UCol operator>(const Col &a, double s) {
    UCol u{std::vector<unsigned char>(a.n_elem)};
    for (uword k = 0; k < a.n_elem; k++) u.v[k] = a.mem[k] > s;
    return u;
}
inline uword sum(const UCol &u) { uword c = 0; for (auto b : u.v) c += b; return c; }
inline uvec find(const UCol &u) { uvec o; for (uword k = 0; k < u.v.size(); k++) if (u.v[k]) o.push_back(k); return o; }
inline uvec find(const UMat &u) { uvec o; for (uword k = 0; k < u.v.size(); k++) if (u.v[k]) o.push_back(k); return o; }
/* sum(matrix): column sums as a row vector */
inline Mat sum(const Mat &a) {
    Mat o(1, a.n_cols);
    for (uword j = 0; j < a.n_cols; j++) {
        double acc = 0.0;
        for (uword i = 0; i < a.n_rows; i++) acc += a(i, j);
        o(0, j) = acc;
    }
    return o;
}

inline Mat zeros(uword r, uword c) { return Mat(r, c); }
inline Col zeros(uword n) { return Col(n); }
inline Col ones(uword n) { Col o(n); for (auto &x : o.mem) x = 1.0; return o; }
template <typename T> inline T linspace(double a, double b, uword n) {
    T o(n);
    if (n == 1) { o.mem[0] = b; return o; }
    double delta = (b - a) / double(n - 1);
    for (uword k = 0; k + 1 < n; k++) o.mem[k] = a + double(k) * delta;
    o.mem[n - 1] = b;
    return o;
}
inline Mat diagmat(const Mat &v) {
    Mat o(v.n_elem, v.n_elem);
    for (uword k = 0; k < v.n_elem; k++) o(k, k) = v.mem[k];
    return o;
}
inline Col diagvec(const Mat &a) {
    Col o(std::min(a.n_rows, a.n_cols));
    for (uword k = 0; k < o.n_elem; k++) o.mem[k] = a(k, k);
    return o;
}
inline Mat join_cols(const Mat &a, const Mat &b) {
    if (a.n_cols != b.n_cols) { std::fprintf(stderr, "shim: join_cols mismatch\n"); std::abort(); }
    Mat o(a.n_rows + b.n_rows, a.n_cols);
    for (uword j = 0; j < a.n_cols; j++) {
        for (uword i = 0; i < a.n_rows; i++) o(i, j) = a(i, j);
        for (uword i = 0; i < b.n_rows; i++) o(a.n_rows + i, j) = b(i, j);
    }
    return o;
}
inline double trace(const Mat &a) {
    double acc = 0.0;
    for (uword k = 0; k < std::min(a.n_rows, a.n_cols); k++) acc += a(k, k);
    return acc;
}

/* LU with partial pivoting (row interchanges), as LAPACK's getrf does */
struct ShimLU {
    Mat a; std::vector<uword> piv; int sign = 1; bool singular = false;
    explicit ShimLU(const Mat &m) : a(m), piv(m.n_rows) {
        const uword n = a.n_rows;
        for (uword k = 0; k < n; k++) {
            uword pr = k; double best = std::fabs(a(k, k));
            for (uword i = k + 1; i < n; i++) if (std::fabs(a(i, k)) > best) { best = std::fabs(a(i, k)); pr = i; }
            piv[k] = pr;
            if (pr != k) { sign = -sign; for (uword j = 0; j < n; j++) std::swap(a(k, j), a(pr, j)); }
            if (a(k, k) == 0.0) { singular = true; continue; }
            for (uword i = k + 1; i < n; i++) {
                a(i, k) /= a(k, k);
                for (uword j = k + 1; j < n; j++) a(i, j) -= a(i, k) * a(k, j);
            }
        }
    }
    void solve_inplace(Mat &b) const {
        const uword n = a.n_rows;
        for (uword c = 0; c < b.n_cols; c++) {
            for (uword k = 0; k < n; k++) if (piv[k] != k) std::swap(b(k, c), b(piv[k], c));
            for (uword i = 1; i < n; i++) for (uword k = 0; k < i; k++) b(i, c) -= a(i, k) * b(k, c);
            for (uword ii = n; ii-- > 0;) {
                for (uword k = ii + 1; k < n; k++) b(ii, c) -= a(ii, k) * b(k, c);
                b(ii, c) /= a(ii, ii);
            }
        }
    }
};
inline double det(const Mat &m) {
    if (m.n_rows != m.n_cols) { std::fprintf(stderr, "shim: det of non-square\n"); std::abort(); }
    ShimLU lu(m);
    double d = lu.sign;
    for (uword k = 0; k < m.n_rows; k++) d *= lu.a(k, k);
    return d;
}
inline Mat Mat::i() const {
    if (n_rows != n_cols) { std::fprintf(stderr, "shim: inverse of non-square\n"); std::abort(); }
    ShimLU lu(*this);
    Mat o(n_rows, n_rows);
    for (uword k = 0; k < n_rows; k++) o(k, k) = 1.0;
    lu.solve_inplace(o);
    return o;
}
inline bool solve(Mat &out, const Mat &a, const Mat &b) {
    ShimLU lu(a);
    Mat x = b;
    lu.solve_inplace(x);
    out = x;
    return !lu.singular;
}
inline bool solve(Col &out, const Mat &a, const Mat &b) { Mat x; bool ok = solve(x, a, b); out = x; return ok; }

/* economical QR by Householder reflections (the dgeqrf / dorgqr construction) */
inline bool qr_econ(Mat &q, Mat &r, const Mat &A) {
    const uword m = A.n_rows, n = A.n_cols;
    Mat a = A;
    std::vector<double> tau(n, 0.0);
    for (uword k = 0; k < n && k < m; k++) {
        double xnorm = 0.0;
        for (uword i = k + 1; i < m; i++) xnorm = std::hypot(xnorm, a(i, k));
        double alpha = a(k, k);
        if (xnorm == 0.0) { tau[k] = 0.0; continue; }
        double beta = -std::copysign(std::hypot(alpha, xnorm), alpha);
        tau[k] = (beta - alpha) / beta;
        double scal = 1.0 / (alpha - beta);
        for (uword i = k + 1; i < m; i++) a(i, k) *= scal;
        a(k, k) = beta;
        for (uword j = k + 1; j < n; j++) {
            double w = a(k, j);
            for (uword i = k + 1; i < m; i++) w += a(i, k) * a(i, j);
            w *= tau[k];
            a(k, j) -= w;
            for (uword i = k + 1; i < m; i++) a(i, j) -= a(i, k) * w;
        }
    }
    r = Mat(n, n);
    for (uword j = 0; j < n; j++) for (uword i = 0; i <= j && i < m; i++) r(i, j) = a(i, j);
    q = Mat(m, n);
    for (uword j = 0; j < n; j++) q(j, j) = 1.0;
    for (uword kk = n; kk-- > 0;) {
        if (kk >= m) continue;
        for (uword j = 0; j < n; j++) {
            double w = q(kk, j);
            for (uword i = kk + 1; i < m; i++) w += a(i, kk) * q(i, j);
            w *= tau[kk];
            q(kk, j) -= w;
            for (uword i = kk + 1; i < m; i++) q(i, j) -= a(i, kk) * w;
        }
    }
    return true;
}
}  // namespace arma

/* ------------------------------------------------------------------ Rcpp stand-in */
namespace Rcpp {

/* a NumericMatrix row / the value of a sugar expression: evaluated eagerly */
class RVec {
  public:
    std::vector<double> v;
    RVec() {}
    explicit RVec(std::size_t n) : v(n) {}
    std::size_t size() const { return v.size(); }
    operator arma::Col() const { return arma::Col(v); }
};
template <typename F> inline RVec rmap(const RVec &a, F f) { RVec o(a.size()); for (std::size_t k = 0; k < a.size(); k++) o.v[k] = f(a.v[k]); return o; }
template <typename F> inline RVec rzip(const RVec &a, const RVec &b, F f) {
    if (a.size() != b.size()) { std::fprintf(stderr, "shim: sugar size mismatch\n"); std::abort(); }
    RVec o(a.size());
    for (std::size_t k = 0; k < a.size(); k++) o.v[k] = f(a.v[k], b.v[k]);
    return o;
}
#define SHIM_ARITH(T) typename std::enable_if<std::is_arithmetic<T>::value, int>::type = 0
#define SHIM_SUGAR(OP)                                                                                         \
    inline RVec operator OP(const RVec &a, const RVec &b) { return rzip(a, b, [](double x, double y) { return x OP y; }); } \
    template <typename T, SHIM_ARITH(T)> inline RVec operator OP(const RVec &a, T s) {                          \
        double d = (double)s; return rmap(a, [d](double x) { return x OP d; }); }                              \
    template <typename T, SHIM_ARITH(T)> inline RVec operator OP(T s, const RVec &a) {                          \
        double d = (double)s; return rmap(a, [d](double x) { return d OP x; }); }
SHIM_SUGAR(+)
SHIM_SUGAR(-)
SHIM_SUGAR(*)
SHIM_SUGAR(/)
#undef SHIM_SUGAR
template <typename T, SHIM_ARITH(T)> inline RVec pow(const RVec &a, T e) { double d = (double)e; return rmap(a, [d](double x) { return std::pow(x, d); }); }
inline RVec log(const RVec &a) { return rmap(a, [](double x) { return std::log(x); }); }
inline RVec lgamma(const RVec &a) { return rmap(a, [](double x) { return Rf_lgammafn(x); }); }
inline RVec digamma(const RVec &a) { return rmap(a, [](double x) { return Rf_digamma(x); }); }
inline RVec trigamma(const RVec &a) { return rmap(a, [](double x) { return Rf_trigamma(x); }); }
inline double sum(const RVec &a) { double acc = 0.0; for (double x : a.v) acc += x; return acc; }

struct Placeholder {};
static const Placeholder _ = Placeholder();

class NumericVector {
  public:
    std::shared_ptr<SexpRec> rec;
    NumericVector(SEXP s) : rec(s, [](SexpRec *) {}) {}
    NumericVector(int n) : rec(new SexpRec()) { rec->d.assign(n, 0.0); rec->nrow = n; }
    NumericVector(const RVec &r) : rec(new SexpRec()) { rec->d = r.v; rec->nrow = (int)r.size(); }
    double &operator()(int i) { return rec->d[i]; }
    double operator()(int i) const { return rec->d[i]; }
    operator SEXP() const { return rec.get(); }
};
class IntegerVector {
  public:
    std::shared_ptr<SexpRec> rec;
    IntegerVector(int n) : rec(new SexpRec()) { rec->iv.assign(n, 0); rec->nrow = n; rec->is_int = true; }
    int &operator()(int i) { return rec->iv[i]; }
    operator SEXP() const { return rec.get(); }
};
class NumericMatrix {
  public:
    SEXP s;
    typedef RVec Row;
    NumericMatrix(SEXP s_) : s(s_) {}
    int nrow() const { return s->nrow; }
    int ncol() const { return s->ncol; }
    Row row(int i) const {
        Row r((std::size_t)s->ncol);
        for (int j = 0; j < s->ncol; j++) r.v[j] = s->d[(std::size_t)i + (std::size_t)s->nrow * j];
        return r;
    }
    Row operator()(int i, const Placeholder &) const { return row(i); }
};
inline SEXP clone(SEXP s) { SEXP c = shim_alloc(); *c = *s; return c; }

template <typename T> inline T as(SEXP s);
template <> inline double as<double>(SEXP s) { return s->is_int ? (double)s->iv[0] : s->d[0]; }
template <> inline int as<int>(SEXP s) { return s->is_int ? s->iv[0] : (int)s->d[0]; }
template <> inline bool as<bool>(SEXP s) { return (s->is_int ? (double)s->iv[0] : s->d[0]) != 0.0; }
template <> inline arma::Mat as<arma::Mat>(SEXP s) {
    arma::Mat m(s->nrow, s->ncol ? s->ncol : 1);
    m.mem = s->d;
    return m;
}
template <> inline arma::Col as<arma::Col>(SEXP s) { return arma::Col(s->d); }

struct NamedArg { std::string name; SexpRec value; };
inline SexpRec shim_wrap(const NumericVector &v) { return *v.rec; }
inline SexpRec shim_wrap(const IntegerVector &v) { return *v.rec; }
inline SexpRec shim_wrap(const arma::Mat &m) { SexpRec r; r.d = m.mem; r.nrow = (int)m.n_rows; r.ncol = (int)m.n_cols; return r; }
template <typename T> inline NamedArg Named(const char *name, const T &v) { return NamedArg{name, shim_wrap(v)}; }

class List {
  public:
    std::vector<NamedArg> items;
    template <typename... A> static List create(const A &...a) { List l; l.items = {a...}; return l; }
    const SexpRec &get(const char *name) const {
        for (auto &it : items) if (it.name == name) return it.value;
        std::fprintf(stderr, "shim: no list element '%s'\n", name);
        std::abort();
    }
};
inline void checkUserInterrupt() {}
}  // namespace Rcpp
