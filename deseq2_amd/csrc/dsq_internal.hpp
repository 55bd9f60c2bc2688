// dsq_internal.hpp -- launch-level parameter blocks shared by the kernels and the C ABI.
// All pointers are device pointers; n x m matrices are gene-major with leading dim ld.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <functional>
#include <mutex>
#include <string.h>

namespace dsq {

struct DispKernelParams {
    int n, m, p;
    long ld;
    const int32_t *y;
    const double *mu_hat;
    const double *weights;  // nullptr unless useWeights
    const double *x;        // m x p column-major
    const double *log_alpha_in;
    const double *prior_mean;
    double prior_sigmasq, min_log_alpha, kappa_0, tol, weightThreshold;
    int maxit, usePrior, useWeights, useCR;
    // fitDisp outputs
    double *log_alpha;
    int32_t *iter, *iter_accept;
    double *last_change, *initial_lp, *initial_dlp, *last_lp, *last_dlp, *last_d2lp;
    // fitDispGrid
    const double *grid;
    int ngrid;
    int ablate, force_iters; // profiling only
    int xlds;                // 1: X staged in LDS, 0: read through L1/L2
    int *work_counter;       // zeroed int: waves draw their next gene from it (nullptr: static grid-stride)
    unsigned long long padmask;   // WIDE kernels: bit c set = design column c is zero padding
    const double *prior_sigmasq_dev;   // non-null: the prior variance is read from the device (fused pipeline)
    // design cells (see BetaKernelParams): ncell > 0 -> the Cox-Reid matrices are assembled from per-cell sums
    const int32_t *cell_perm, *cell_start;
    int ncell;
    // fused pipeline (pipeline.hip): the launch covers the genes rows[0 .. *n_dev) of full-size arrays (rows == nullptr:
    // genes 0 .. n-1); n stays the capacity / leading dimension of the n-vectors and n x p matrices
    const int32_t *rows;
    const int32_t *n_dev;
    int rows_few;            // the list is expected to be short (stragglers, refits): a one-block-per-CU grid is enough
    // unstaged rows (long rows read through L2), no weights: the distinct-count buffers of the resident waves (2 m int32
    // each) in GLOBAL memory -- set by the launch (fit_disp.hip); the kernel instantiation decides at compile time
    int32_t *dist_global;
};

struct BetaKernelParams {
    int n, m, p;
    long ld;
    const int32_t *y;
    const double *nf;       // gene-major matrix, or m-vector when nf_is_vector
    int nf_is_vector;
    const double *weights;  // nullptr unless useWeights
    const double *x;        // m x p column-major
    const double *alpha_hat;
    const double *contrast;
    const double *beta_init;  // n x p column-major
    const double *lambda;
    double tol, minmu, mu_floor;
    int maxit, useQR, useWeights;
    double *beta_mat, *beta_var_mat, *iter, *hat_diagonals, *contrast_num, *contrast_denom, *deviance;
    double *mu_out;
    double *scratch;        // global per-wave-slot scratch when rows are not staged in LDS
    double *cscratch;       // (unused since the closed-form deviance; kept so that the launch plumbing stays put)
    int ablate, force_iters; // profiling only (env DSQ_ABLATE / DSQ_FORCE_ITERS): skip phases / fixed trip count
    int xlds;                // 1: X staged in LDS, 0: read through L1/L2
    int *work_counter;       // zeroed int: waves draw their next gene from it (nullptr: static grid-stride)
    // fused pipeline (pipeline.hip): the launch covers the genes rows[0 .. *n_dev) of full-size arrays (rows == nullptr:
    // genes 0 .. n-1); n stays the capacity / leading dimension of the n-vectors and n x p matrices
    const int32_t *rows;
    const int32_t *n_dev;
    int rows_few;            // the list is expected to be short (stragglers, refits): a one-block-per-CU grid is enough
    // design cells (samples with identical design rows, numbered by first appearance): ncell > 0 selects the
    // cell-collapsed kernel; cell_perm = samples grouped by cell (ascending inside a cell), cell_start = ncell + 1 offsets
    const int32_t *cell_perm, *cell_start;
    int ncell;
    double *kconst_out;     // n values or NULL: K' of the row -- the mu-independent part of sum [wts] log NB(y; 1/alpha, mu) on
                            // the closed split (dsq_math.hpp) -- for the nbinomLogLike launch that follows this fit
    int p_true;             // WIDE designs: the design's own number of columns (the zero padding follows them); 0 = p.  The
                            // rolled kernel runs at this width -- the padded coefficients' ridge rows and zero columns only
                            // ever add exact zeros to the real ones' sums -- and writes the padding's outputs (0) itself
};

struct PrefitKernelParams {
    int n, m, p;
    long ld;
    const int32_t *y;
    const double *nf;
    int nf_is_vector;
    const double *weights;
    int useWeights;
    const double *q, *a, *r;   // Q (m x p), X R^-1 (m x p), R (p x p), all column-major
    double *baseMean, *baseVar, *roughDisp, *beta_init;
    int32_t *allZero;
    // fused pipeline (pipeline.hip): the launch covers the genes rows[0 .. *n_dev) of full-size arrays (rows == nullptr:
    // genes 0 .. n-1); n stays the capacity / leading dimension of the n-vectors and n x p matrices
    const int32_t *rows;
    const int32_t *n_dev;
};

struct LogLikeKernelParams {
    int n, m;
    long ld;
    const int32_t *y;
    const double *mu;
    const double *disp;
    const double *weights;
    int useWeights;
    double *loglike;
    // fused pipeline (pipeline.hip): the launch covers the genes rows[0 .. *n_dev) of full-size arrays (rows == nullptr:
    // genes 0 .. n-1); n stays the capacity / leading dimension of the n-vectors and n x p matrices
    const int32_t *rows;
    const int32_t *n_dev;
    const int32_t *skip;     // n flags or NULL: genes with a non-zero flag are left alone (neither read nor written)
    const double *kconst;    // n values or NULL: the mu-independent part of each row's sum, from the fitBeta launch that
                             // fitted the row with the SAME dispersions / weights (BetaKernelParams.kconst_out); NULL: computed here
};

struct InterceptKernelParams {
    int n, m;
    long ld;
    const int32_t *y;
    const double *nf;
    int nf_is_vector;
    const double *weights;
    int useWeights;
    const double *alpha;
    double mu_floor;
    double *beta_log2, *betaSE, *mu_out, *hat;
    double *loglike;         // optional: nbinomLogLike at the (unfloored) fitted means, as loglike_kernel computes it
    // fused pipeline (pipeline.hip): the launch covers the genes rows[0 .. *n_dev) of full-size arrays (rows == nullptr:
    // genes 0 .. n-1); n stays the capacity / leading dimension of the n-vectors and n x p matrices
    const int32_t *rows;
    const int32_t *n_dev;
    const double *kconst;    // as LogLikeKernelParams.kconst (the full model's fit has the same counts, dispersions, weights)
};

struct OptimKernelParams {
    int n, m, p;
    long ld;
    const int32_t *y;
    const double *nf;
    int nf_is_vector;
    const double *weights;
    int useWeights;
    const double *x;           // m x p column-major
    const double *alpha_hat;   // n
    const double *lamnat;      // p: prior precisions on the natural-log scale
    const double *beta_start;  // n x p column-major, log2 scale
    double minmu;
    double *beta, *betaSE;     // n x p column-major, log2 scale
    int32_t *conv;             // n
    double *mu_out;            // n x ld
    double *loglike;           // n
    double mu_floor;           // > 0: the fitted means are stored floored at it (fitMu[fitMu < minmu] <- minmu, R/core.R:763)
    // fused pipeline (pipeline.hip): the launch covers the rows rows[0 .. *n_dev) of full-size arrays (n = their
    // capacity / leading dimension); rows == nullptr: rows 0 .. n-1
    const int32_t *rows;
    const int32_t *n_dev;
};

struct CooksKernelParams {
    int n, m, p;
    long ld;
    const int32_t *y;
    const double *nf;
    int nf_is_vector;
    const double *mu, *H;        // gene-major
    const int32_t *perm;         // sample indices grouped by design cell
    const int32_t *cell_start;   // ncell + 1 offsets into perm
    const int32_t *in3;          // m flags: sample sits in a cell with >= 3 members
    int ncell, any3;
    int sortcap;                 // doubles of sort buffer per wave (power of two)
    double *cooks, *maxCooks, *robustDisp;
    const int32_t *rows;         // fused pipeline: see DispKernelParams
    const int32_t *n_dev;
};

struct ReplaceKernelParams {
    int n, m;
    long ld;
    const int32_t *y;
    const double *nf;
    int nf_is_vector;
    const double *cooks;
    double cutoff, trim;
    const int32_t *replaceable;  // m flags
    int sortcap;
    int32_t *newCounts, *replace;
    const int32_t *rows;
    const int32_t *n_dev;
};

// gene index of work item i, and the number of work items, of a (possibly row-listed) launch
#define DSQ_NWORK(kp) ((kp).n_dev ? *(kp).n_dev : (kp).n)
#define DSQ_GENE(kp, i) ((kp).rows ? (kp).rows[i] : (i))

hipError_t launch_cooks(const CooksKernelParams &kp, hipStream_t st, bool *ok);
hipError_t launch_replace(const ReplaceKernelParams &kp, hipStream_t st, bool *ok);
hipError_t launch_prefit(const PrefitKernelParams &kp, hipStream_t st, bool *ok);
hipError_t launch_linear_mu(const PrefitKernelParams &kp, double mu_floor, double *mu, hipStream_t st, bool *ok);
hipError_t launch_loglike(const LogLikeKernelParams &kp, hipStream_t st);
hipError_t launch_intercept_fit(const InterceptKernelParams &kp, hipStream_t st);
hipError_t launch_trend_fit(const double *means, const double *disps, long n, double *coefs, int32_t *status,
                            void *workspace, hipStream_t st);
size_t trend_fit_workspace_bytes();
// getAndCheckWeights on resident weights: w / rowmax, pmax(., 1e-6), the weightsFail flags, the negative-weight flag
hipError_t launch_weights_prep(const double *w_raw, const double *x, int n, int m, int p, long ld, double thr, double *w_norm,
                               double *w_floor, int32_t *force_zero, int32_t *neg, hipStream_t st);
// xim of a normalization-factor matrix -> *out (device); scratch_m: m doubles
hipError_t launch_xim(const double *nf, int n, int m, long ld, double *scratch_m, double *out, hipStream_t st);
hipError_t launch_xim_rows(const double *nf, const int32_t *rows, const int32_t *n_dev, int m, long ld, double *scratch_m,
                           double *out, hipStream_t st);

// Register-resident kernels exist for 1 <= p <= DSQ_P_REG (one translation unit per p,
// explicit specialisations in fit_disp.hip / fit_beta.hip compiled with -DDSQ_P=p).
#define DSQ_P_REG 10
// 11 <= p <= DSQ_P_WIDE run on the translation units of DSQ_WIDE_LIST (-DDSQ_P=16, 24, ...: loops not unrolled, p x p state in
// scratch memory) over designs zero-padded to the next listed width: a padded column gets ridge 1 and a unit diagonal in
// the Cox-Reid matrix, which leaves every quantity of the real coefficients unchanged (capi.hip, "wide designs").
// (round 5: 32 and 48 added -- `~ patient + treatment` with up to 47 patients, factors of up to 48 levels.  Round 6: 64 --
//  its fits are the ROLLED kernels alone (fit_beta_wide.hip, fit_disp_wide.hip: one build for every width); the per-width
//  fitDisp kernel, whose 64-column build compiles for half an hour, stops at DSQ_DISP_PERWIDTH_MAX)
#define DSQ_P_WIDE0 16
#define DSQ_WIDE_LIST(X) X(16) X(24) X(32) X(48) X(64)
#define DSQ_P_WIDE 64
#define DSQ_DISP_PERWIDTH_MAX 48
static inline unsigned long long dsq_low_bits(int k) { return k >= 64 ? ~0ull : ((1ull << k) - 1ull); }   // bits 0 .. k - 1
static inline int dsq_wide_width(int p) { return p <= 16 ? 16 : p <= 24 ? 24 : p <= 32 ? 32 : p <= 48 ? 48 : 64; }   // padded width of a wide p
// every design width 1 .. DSQ_P_WIDE (the small per-width kernels of aux.hip: pre-fit moments, linear mu)
#define DSQ_P_EACH(X) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) X(17) X(18) X(19) X(20) \
    X(21) X(22) X(23) X(24) X(25) X(26) X(27) X(28) X(29) X(30) X(31) X(32) X(33) X(34) X(35) X(36) X(37) X(38) X(39) X(40) \
    X(41) X(42) X(43) X(44) X(45) X(46) X(47) X(48) X(49) X(50) X(51) X(52) X(53) X(54) X(55) X(56) X(57) X(58) X(59) X(60) \
    X(61) X(62) X(63) X(64)
#define DSQ_CMAX 32       // most design cells the cell-collapsed paths take
#define DSQ_DISP_CELL_MINP 4   // fitDisp assembles the Cox-Reid matrices from cell sums from this design width up
template <int P> hipError_t launch_fit_disp_p(const DispKernelParams &kp, hipStream_t st, bool grid);
template <int P> hipError_t launch_fit_beta_p(const BetaKernelParams &kp, hipStream_t st);
template <int P> hipError_t launch_optim_p(const OptimKernelParams &kp, hipStream_t st);
// doubles of global scratch one fitBeta launch needs: `slab` (per-wave mu/sqrt(w)/sqrt(w)z when they
// do not fit in LDS, else 0) and `cscr` (hoisted NB-density constants)
template <int P> void fit_beta_scratch_doubles(int n, int m, int use_weights, size_t *slab, size_t *cscr);

// launch-geometry knobs (environment variables DSQ_*; read once) -- used for tuning sweeps
struct Tuning {
    int beta_waves, beta_stage, beta_bpc, beta_lds_kb;
    int disp_waves, disp_stage, disp_bpc, disp_lds_kb;
    int ablate, force_iters;
    int disp_xlds, beta_xlds;
    int dynamic;             // DSQ_DYNAMIC (default 1): dynamic gene scheduling in the fit kernels
    int beta_cells;          // DSQ_BETA_CELLS (default 1): cell-collapsed fitBeta for designs with <= DSQ_CMAX cells
    int disp_cell_minp;      // DSQ_DISP_CELL_MINP (profiling; default = the macro): fitDisp cell mode from this width up
};
const Tuning &tuning();

hipError_t launch_transpose_r_to_gm_f64(const double *src, double *dst, int n, int m, long ld, hipStream_t st);
hipError_t launch_transpose_r_to_gm_i32(const int32_t *src, int32_t *dst, int n, int m, long ld, hipStream_t st);
hipError_t launch_counts_f64_to_gm_i32(const double *src, int32_t *dst, int n, int m, long ld, int32_t *bad, hipStream_t st);
hipError_t launch_transpose_gm_to_r_f64(const double *src, double *dst, int n, int m, long ld, hipStream_t st);
hipError_t launch_transpose_gm_to_r_i32(const int32_t *src, int32_t *dst, int n, int m, long ld, hipStream_t st);
hipError_t launch_test_math(int op, const double *a, const double *b, const double *c, double *out, long n, hipStream_t st);

int device_cu_count();
// launch-geometry caches are per host thread (the multi-device host entry points drive one device per worker thread)
// and are dropped when that thread moves to another device: the MaxDynamicSharedMemorySize attribute and the occupancy
// figure belong to a device
#define DSQ_CACHE_PER_DEVICE(a, b)                                                        \
    do {                                                                                  \
        static thread_local int cache_dev_ = -1;                                          \
        int dev_now_ = 0;                                                                 \
        (void)hipGetDevice(&dev_now_);                                                    \
        if (dev_now_ != cache_dev_) { memset(a, 0, sizeof a); memset(b, 0, sizeof b); cache_dev_ = dev_now_; } \
    } while (0)

// ---- shared by capi.hip and pipeline.hip (the fused DESeq() chain) ------------------------------------------
int capi_fail(int code, const char *fmt, ...);
int capi_upload_table(int slot, const void *src, size_t bytes, hipStream_t st, void **dev_out);   // small host table -> slot (pinned ring, skipped when unchanged)
int capi_ws_get(int slot, size_t bytes, void **out);             // grow-only workspace of the current (device, stream)
int capi_check_device();
int capi_upload_cells(const int32_t *labels, int m, int slot, hipStream_t st, const int32_t **perm_dev,
                      const int32_t **start_dev);
std::mutex &capi_mutex();                                        // the library's call lock
void capi_latch_stream(hipStream_t s);                           // workspace key of the current call (under the lock)
// the rolled kernel of the wide designs without design cells (fit_beta_wide.hip): any kp.p in 11 .. 64
hipError_t launch_fit_beta_rolled(const BetaKernelParams &kp, hipStream_t st);
void fit_beta_rolled_scratch_doubles(int n, int m, int p, int useW, size_t *slab, size_t *cscr);
hipError_t dispatch_fit_beta(int p, const BetaKernelParams &kp, hipStream_t st, bool *ok);
void dispatch_beta_scratch(int p, int n, int m, int useW, size_t *slab, size_t *cscr);
hipError_t dispatch_fit_disp(int p, const DispKernelParams &kp, hipStream_t st, bool grid, bool *ok);
bool fit_disp_rolled_applies(const DispKernelParams &kp, int *p_true);       // fit_disp_wide.hip
hipError_t dispatch_optim_rows(int p, const OptimKernelParams &kp, hipStream_t st, bool *ok);
void capi_prof_begin(const char *name, int n, hipStream_t st);   // no-ops unless dsq_profile_enable(1)
void capi_prof_end(hipStream_t st);
bool capi_prof_on();
// a second stream (with two events) next to `main` on the current device: the chain runs work that nothing downstream
// waits for beside its serial tail (pipeline.hip); created once per (device, main stream), under the call lock
int capi_side_stream(hipStream_t main, hipStream_t *side, hipEvent_t *fork_ev, hipEvent_t *join_ev);
// stage.hip: pageable host memory <-> device through pinned chunks packed by a small thread pool; rows [lo, lo + cnt) of
// a column-major n_total x cols host matrix <-> a contiguous column-major cnt x cols device matrix.  stage_d2h returns
// when the host rows are complete.
int stage_h2d(void *dev, const void *host, size_t e, size_t n_total, size_t lo, size_t cnt, size_t cols, hipStream_t st);
int stage_d2h(void *host, const void *dev, size_t e, size_t n_total, size_t lo, size_t cnt, size_t cols, hipStream_t st);
// first touch of a large result matrix in the caller's (fresh, pageable) memory on helper threads, ahead of the copy into
// it -- announce the outputs when an entry point starts, wait before it returns (stage.hip)
void stage_prefault(void *host, size_t bytes);
void stage_prefault_finish();
struct PrefaultScope { ~PrefaultScope() { stage_prefault_finish(); } };
// (three slots between the call slots of capi.hip and the chain's: the padded reduced / prior design, the padded design, the
//  prior-variance selection workspace)
enum { DSQ_WS_PIPE_PADXR = 37, DSQ_WS_PIPE_PADX = 38, DSQ_WS_PIPE_SEL = 39 };
enum { DSQ_WS_PIPE = 40, DSQ_WS_PIPE_SCRATCH = 41, DSQ_WS_PIPE_META = 42, DSQ_WS_HOSTDESEQ = 48, DSQ_WS_COUNT = 72 };

}  // namespace dsq
