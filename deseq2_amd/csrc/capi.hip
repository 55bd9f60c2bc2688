// capi.hip -- the C ABI of libdeseq2_mi355x.so (include/deseq2_mi355x.h).
//
// Host side of the engine: argument validation, device workspaces, layout conversion
// (R column-major <-> gene-major), kernel dispatch on the design width p, and the
// host-pointer convenience entry points the R .Call shim binds.  No CPU fallback: if
// HIP cannot give us a device, every entry point fails with DSQ_ERR_DEVICE.
#include "../../include/deseq2_mi355x.h"
#include "dsq_internal.hpp"

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <condition_variable>
#include <functional>
#include <map>
#include <mutex>
#include <thread>
#include <vector>

namespace dsq {

static thread_local char g_err[512] = "";
static std::mutex g_mu;

static int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}
int capi_fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

#define DSQ_HIP(expr)                                                                            \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return fail(e_ == hipErrorOutOfMemory ? DSQ_ERR_NOMEM : DSQ_ERR_DEVICE, "%s: %s", #expr, \
                        hipGetErrorString(e_));                                                  \
    } while (0)

static int env_int(const char *name, int dflt) {
    const char *v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}

const Tuning &tuning() {
    static Tuning t = {env_int("DSQ_BETA_WAVES", 4), env_int("DSQ_BETA_STAGE", -1), env_int("DSQ_BETA_BPC", 0),
                       env_int("DSQ_BETA_LDS_KB", 160),
                       env_int("DSQ_DISP_WAVES", 4), env_int("DSQ_DISP_STAGE", -1), env_int("DSQ_DISP_BPC", 0),
                       env_int("DSQ_DISP_LDS_KB", 160), env_int("DSQ_ABLATE", 0), env_int("DSQ_FORCE_ITERS", 0),
                       env_int("DSQ_DISP_XLDS", 1), env_int("DSQ_BETA_XLDS", 1), env_int("DSQ_DYNAMIC", 1),
                       env_int("DSQ_BETA_CELLS", 1), env_int("DSQ_DISP_CELL_MINP", DSQ_DISP_CELL_MINP)};
    return t;
}

int device_cu_count() {
    static int cached[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (cached[dev] == 0) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        cached[dev] = v;
    }
    return cached[dev];
}

// ---- optional kernel timing (HIP events on the launch stream) ----------------------
// dsq_profile_enable(1) starts a list of (name, genes, event pair) -- one entry per bracketed launch (a fit call has
// one, the fused pipeline one per kernel); dsq_profile_count / dsq_profile_get read the durations back.
struct ProfEntry { char name[32]; int n; hipEvent_t e0, e1; };
static bool g_prof = false;
static std::vector<ProfEntry> g_prof_list;
static std::vector<hipEvent_t> g_prof_free;
static hipEvent_t prof_event() {
    if (!g_prof_free.empty()) { hipEvent_t e = g_prof_free.back(); g_prof_free.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}
static void prof_clear() {
    for (auto &p : g_prof_list) { g_prof_free.push_back(p.e0); g_prof_free.push_back(p.e1); }
    g_prof_list.clear();
}
static std::mutex g_prof_mu;
void capi_prof_begin(const char *name, int n, hipStream_t st) {
    if (!g_prof) return;
    std::lock_guard<std::mutex> plk(g_prof_mu);
    ProfEntry p;
    snprintf(p.name, sizeof p.name, "%s", name);
    p.n = n; p.e0 = prof_event(); p.e1 = prof_event();
    (void)hipEventRecord(p.e0, st);
    g_prof_list.push_back(p);
}
void capi_prof_end(hipStream_t st) {
    if (!g_prof) return;
    std::lock_guard<std::mutex> plk(g_prof_mu);
    if (g_prof_list.empty()) return;
    (void)hipEventRecord(g_prof_list.back().e1, st);
}
bool capi_prof_on() { return g_prof; }

struct SideStream { hipStream_t s = nullptr; hipEvent_t f = nullptr, j = nullptr; };
static std::map<std::pair<int, hipStream_t>, SideStream> g_side;      // (device, main stream) -> its side stream; under g_mu / the
static std::mutex g_side_mu;                                          // worker threads of a sharded host call: own mutex
int capi_side_stream(hipStream_t main, hipStream_t *side, hipEvent_t *fork_ev, hipEvent_t *join_ev) {
    int dev = 0;
    DSQ_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_side_mu);
    SideStream &e = g_side[{dev, main}];
    if (!e.s) {
        DSQ_HIP(hipStreamCreateWithFlags(&e.s, hipStreamNonBlocking));
        DSQ_HIP(hipEventCreateWithFlags(&e.f, hipEventDisableTiming));
        DSQ_HIP(hipEventCreateWithFlags(&e.j, hipEventDisableTiming));
    }
    *side = e.s; *fork_ev = e.f; *join_ev = e.j;
    return DSQ_OK;
}

static void prof_begin(hipStream_t st) { capi_prof_begin("call", 0, st); }
static void prof_end(hipStream_t st) { capi_prof_end(st); }

// ---- workspace pool: grow-only device buffers, one per (device, slot) -------------
// Keyed by (device, stream): calls issued on different streams (e.g. the chunks of a pipelined DESeq(),
// deseq2_amd/parallel.py) must not share scratch, counters or staging buffers while both are in flight.
// The stream of the current API call is latched at entry (WsScope, under g_mu).
struct Slot {
    void *p = nullptr; size_t bytes = 0;
    std::vector<unsigned char> table;      // capi_upload_table: the bytes the slot holds (uploaded to `table_p`)
    void *table_p = nullptr;
};
static constexpr int WS_SLOTS_MAX = DSQ_WS_COUNT;
static std::map<hipStream_t, std::vector<Slot>> g_pool[64];
static std::mutex g_pool_mu;      // the maps themselves (worker threads of a multi-device call look their slots up concurrently)
static thread_local hipStream_t g_ws_stream = nullptr;
// (its destructor also waits for the first-touch threads of stage.hip: no entry point returns while they are at work)
struct WsScope { explicit WsScope(hipStream_t s) { g_ws_stream = s; } ~WsScope() { stage_prefault_finish(); } };
std::mutex &capi_mutex() { return g_mu; }
void capi_latch_stream(hipStream_t s) { g_ws_stream = s; }

static int ws_get(int slot, size_t bytes, void **out) {
    int dev = 0;
    DSQ_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return fail(DSQ_ERR_DEVICE, "device index %d out of range", dev);
    std::unique_lock<std::mutex> plk(g_pool_mu);
    auto &pool = g_pool[dev][g_ws_stream];
    if ((int)pool.size() < WS_SLOTS_MAX) pool.resize(WS_SLOTS_MAX);     // never reallocated afterwards: `s` stays valid
    Slot &s = pool[slot];
    plk.unlock();
    if (s.bytes < bytes) {
        if (s.p) { DSQ_HIP(hipDeviceSynchronize()); DSQ_HIP(hipFree(s.p)); s.p = nullptr; s.bytes = 0; }
        s.table_p = nullptr;               // (a new allocation may land on the old address: its bytes are not the table's)
        size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipMalloc(&s.p, want);
        if (e != hipSuccess) { s.p = nullptr; return fail(DSQ_ERR_NOMEM, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e)); }
        s.bytes = want;
    }
    *out = s.p;
    return DSQ_OK;
}

enum {  // workspace slots
    WS_Y = 0, WS_NF, WS_W, WS_MU, WS_HAT, WS_MUOUT, WS_SCRATCH, WS_BAD, WS_CELLS, WS_COOKS_IN, WS_COUNTER, WS_TREND, WS_PAD_X, WS_PAD_VEC, WS_PAD_BETA,
    WS_CELLS_BETA,
    // host-entry staging
    WS_H_Y, WS_H_X, WS_H_NF, WS_H_W, WS_H_MU, WS_H_VEC, WS_H_OUTMAT, WS_H_OUTMAT2, WS_H_OUTVEC,
    WS_COUNT
};
static_assert(WS_COUNT <= DSQ_WS_PIPE_PADXR, "pipeline workspace slots follow the call slots");
int capi_ws_get(int slot, size_t bytes, void **out) { return ws_get(slot, bytes, out); }

// Small host tables of an ASYNCHRONOUS call (the chain: design cells, outlier metadata).  Two things the plain
// hipMemcpyAsync from a thread_local pageable buffer did not give: (1) the source may be rewritten as soon as this
// returns -- the bytes travel through a ring of PINNED buffers, each fenced by an event recorded behind its copy (a
// buffer is reused only when its copy has run), so nothing depends on how the runtime stages pageable copies; (2) the
// tables of a design are the same analysis after analysis: a slot that already holds these bytes is not uploaded again
// (one memcmp of a few KiB instead of a copy command in front of every chain).
namespace {
struct PinBuf { void *h = nullptr; size_t cap = 0; hipEvent_t done = nullptr; bool used = false; };
constexpr int kPinRing = 8;
PinBuf g_pin[kPinRing];
int g_pin_next = 0;
std::mutex g_pin_mu;
}
int capi_upload_table(int slot, const void *src, size_t bytes, hipStream_t st, void **dev_out) {
    void *v;
    int rc = ws_get(slot, bytes, &v);
    if (rc) return rc;
    *dev_out = v;
    int dev = 0;
    DSQ_HIP(hipGetDevice(&dev));
    Slot *sl;
    {
        std::unique_lock<std::mutex> plk(g_pool_mu);
        sl = &g_pool[dev][g_ws_stream][slot];
    }
    if (sl->table_p == v && sl->table.size() == bytes && memcmp(sl->table.data(), src, bytes) == 0) return DSQ_OK;
    std::lock_guard<std::mutex> lk(g_pin_mu);
    PinBuf &b = g_pin[g_pin_next];
    g_pin_next = (g_pin_next + 1) % kPinRing;
    if (b.used) DSQ_HIP(hipEventSynchronize(b.done));              // (eight uploads ago: long done)
    if (b.cap < bytes) {
        if (b.h) DSQ_HIP(hipHostFree(b.h));
        b.h = nullptr; b.cap = 0;
        DSQ_HIP(hipHostMalloc(&b.h, bytes + bytes / 2 + 256, hipHostMallocDefault));
        b.cap = bytes + bytes / 2 + 256;
    }
    if (!b.done) DSQ_HIP(hipEventCreateWithFlags(&b.done, hipEventDisableTiming));
    memcpy(b.h, src, bytes);
    DSQ_HIP(hipMemcpyAsync(v, b.h, bytes, hipMemcpyHostToDevice, st));
    DSQ_HIP(hipEventRecord(b.done, st));
    b.used = true;
    sl->table.assign((const unsigned char *)src, (const unsigned char *)src + bytes);
    sl->table_p = v;
    return DSQ_OK;
}

static inline long round_ld(int m) { return ((long)m + 7) & ~7L; }

static int check_device() {
    int cnt = 0;
    hipError_t e = hipGetDeviceCount(&cnt);
    if (e != hipSuccess || cnt <= 0)
        return fail(DSQ_ERR_DEVICE, "no HIP device available (%s); libdeseq2_mi355x has no CPU path",
                    e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    return DSQ_OK;
}

int capi_check_device() { return check_device(); }

// counts -> int32 gene-major.  Returns the pointer to use and its ld.
static int prep_counts(const void *y, int y_type, int layout, long ld_in, int n, int m, hipStream_t st,
                       const int32_t **out, long *ld_out, bool *checked_async) {
    *checked_async = false;
    if (layout == DSQ_LAYOUT_GENE_MAJOR) {
        if (y_type != DSQ_Y_INT32)
            return fail(DSQ_ERR_UNSUPPORTED, "gene-major counts must be int32 (y_type = DSQ_Y_INT32)");
        *out = (const int32_t *)y;
        *ld_out = ld_in;
        return DSQ_OK;
    }
    long ld = round_ld(m);
    void *buf;
    int rc = ws_get(WS_Y, (size_t)n * ld * sizeof(int32_t), &buf);
    if (rc) return rc;
    if (y_type == DSQ_Y_INT32) {
        DSQ_HIP(launch_transpose_r_to_gm_i32((const int32_t *)y, (int32_t *)buf, n, m, ld, st));
    } else if (y_type == DSQ_Y_FLOAT64) {
        void *bad;
        rc = ws_get(WS_BAD, sizeof(int32_t), &bad);
        if (rc) return rc;
        DSQ_HIP(hipMemsetAsync(bad, 0, sizeof(int32_t), st));
        DSQ_HIP(launch_counts_f64_to_gm_i32((const double *)y, (int32_t *)buf, n, m, ld, (int32_t *)bad, st));
        *checked_async = true;
    } else {
        return fail(DSQ_ERR_ARG, "unknown y_type %d", y_type);
    }
    *out = (const int32_t *)buf;
    *ld_out = ld;
    return DSQ_OK;
}

static int prep_matrix(const double *src, int layout, long ld_in, int n, int m, int slot, hipStream_t st,
                       const double **out, long ld_expected) {
    if (layout == DSQ_LAYOUT_GENE_MAJOR) {
        (void)ld_in;
        *out = src;
        return DSQ_OK;
    }
    void *buf;
    int rc = ws_get(slot, (size_t)n * ld_expected * sizeof(double), &buf);
    if (rc) return rc;
    DSQ_HIP(launch_transpose_r_to_gm_f64(src, (double *)buf, n, m, ld_expected, st));
    *out = (const double *)buf;
    return DSQ_OK;
}

// zeroed counters for the dynamic gene scheduling of the persistent fit kernels (dsq_wave.hpp next_gene)
static int work_counter(hipStream_t st, int **out) {
    *out = nullptr;
    if (!tuning().dynamic) return DSQ_OK;
    void *v;
    int rc = ws_get(WS_COUNTER, 4 * sizeof(int), &v);
    if (rc) return rc;
    DSQ_HIP(hipMemsetAsync(v, 0, 4 * sizeof(int), st));
    *out = (int *)v;
    return DSQ_OK;
}

template <int P>
struct DispatchP {
    static hipError_t beta(int p, const BetaKernelParams &kp, hipStream_t st, bool *ok) {
        if (p == P) { *ok = true; return launch_fit_beta_p<P>(kp, st); }
        return DispatchP<P - 1>::beta(p, kp, st, ok);
    }
    static void beta_scratch(int p, int n, int m, int useW, size_t *slab, size_t *cscr) {
        if (p == P) { fit_beta_scratch_doubles<P>(n, m, useW, slab, cscr); return; }
        DispatchP<P - 1>::beta_scratch(p, n, m, useW, slab, cscr);
    }
    static hipError_t disp(int p, const DispKernelParams &kp, hipStream_t st, bool grid, bool *ok) {
        if (p == P) { *ok = true; return launch_fit_disp_p<P>(kp, st, grid); }
        return DispatchP<P - 1>::disp(p, kp, st, grid, ok);
    }
    static hipError_t optim(int p, const OptimKernelParams &kp, hipStream_t st, bool *ok) {
        if (p == P) { *ok = true; return launch_optim_p<P>(kp, st); }
        return DispatchP<P - 1>::optim(p, kp, st, ok);
    }
};
template <>
struct DispatchP<0> {
    static hipError_t beta(int, const BetaKernelParams &, hipStream_t, bool *ok) { *ok = false; return hipSuccess; }
    static void beta_scratch(int, int, int, int, size_t *slab, size_t *cscr) { *slab = 0; *cscr = 0; }
    static hipError_t disp(int, const DispKernelParams &, hipStream_t, bool, bool *ok) { *ok = false; return hipSuccess; }
    static hipError_t optim(int, const OptimKernelParams &, hipStream_t, bool *ok) { *ok = false; return hipSuccess; }
};

// (wide designs: the caller passes the PADDED width, one of DSQ_WIDE_LIST -- the chain of pipeline.hip, dsq_optim_rows)
hipError_t dispatch_fit_beta(int p, const BetaKernelParams &kp, hipStream_t st, bool *ok) {
#define DSQ_X(W) if (p == W) { *ok = true; return launch_fit_beta_p<W>(kp, st); }
    DSQ_WIDE_LIST(DSQ_X)
#undef DSQ_X
    return DispatchP<DSQ_P_REG>::beta(p, kp, st, ok);
}
void dispatch_beta_scratch(int p, int n, int m, int useW, size_t *slab, size_t *cscr) {
#define DSQ_X(W) if (p == W) { fit_beta_scratch_doubles<W>(n, m, useW, slab, cscr); return; }
    DSQ_WIDE_LIST(DSQ_X)
#undef DSQ_X
    DispatchP<DSQ_P_REG>::beta_scratch(p, n, m, useW, slab, cscr);
}
hipError_t dispatch_fit_disp(int p, const DispKernelParams &kp, hipStream_t st, bool grid, bool *ok) {
    // (beyond DSQ_DISP_PERWIDTH_MAX columns the rolled kernel of fit_disp_wide.hip is the only one: what it does not take --
    //  rows of more than 1024 samples, a working set beyond the LDS -- is refused)
    if (p > DSQ_DISP_PERWIDTH_MAX && !fit_disp_rolled_applies(kp, nullptr)) { *ok = false; return hipSuccess; }
#define DSQ_X(W) if (p == W) { *ok = true; return launch_fit_disp_p<W>(kp, st, grid); }
    DSQ_WIDE_LIST(DSQ_X)
#undef DSQ_X
    return DispatchP<DSQ_P_REG>::disp(p, kp, st, grid, ok);
}
hipError_t dispatch_optim_rows(int p, const OptimKernelParams &kp, hipStream_t st, bool *ok) {
    // (wide designs: the caller passes the padded width, dsq_optim_rows)
#define DSQ_X(W) if (p == W) { *ok = true; return launch_optim_p<W>(kp, st); }
    DSQ_WIDE_LIST(DSQ_X)
#undef DSQ_X
    return DispatchP<DSQ_P_REG>::optim(p, kp, st, ok);
}

// ---- wide designs (DSQ_P_REG < p <= DSQ_P_WIDE): zero-padded to DSQ_P_WIDE columns --------------------------
// A padded coefficient has an all-zero design column, ridge 1 and start value 0: its estimate is exactly 0 and the
// Gram / QR / LU arithmetic of the real coefficients sees only extra exact zeros (x + 0 = x, 0 * y = 0), so their
// results keep their bits (tests/test_gpu_wide.py compares with the oracle run at the true p).
static inline bool is_wide(int p) { return p > DSQ_P_REG && p <= DSQ_P_WIDE; }
static inline int wide_width(int p) { return dsq_wide_width(p); }   // padded width for a wide p

static int wide_pad_matrix(int slot, const double *src, size_t rows, int p, hipStream_t st, double **out) {
    // column-major rows x p  ->  rows x wide_width(p), new columns zero
    const size_t pw = wide_width(p);
    void *b;
    int rc = ws_get(slot, rows * pw * sizeof(double), &b);
    if (rc) return rc;
    DSQ_HIP(hipMemsetAsync(b, 0, rows * pw * sizeof(double), st));
    DSQ_HIP(hipMemcpyAsync(b, src, rows * (size_t)p * sizeof(double), hipMemcpyDeviceToDevice, st));
    *out = (double *)b;
    return DSQ_OK;
}

static int wide_pad_x(int m, int p, const double *x, hipStream_t st, const double **xout, unsigned long long *padmask) {
    double *b;
    int rc = wide_pad_matrix(WS_PAD_X, x, (size_t)m, p, st, &b);
    if (rc) return rc;
    *xout = b;
    *padmask = dsq_low_bits(wide_width(p)) & ~dsq_low_bits(p);
    return DSQ_OK;
}

// design cells -> device arrays for the cell-collapsed fitBeta kernel: cells renumbered in order of first appearance,
// samples grouped by cell (ascending inside a cell).  labels: m host ints (any numbering).  Returns the number of
// cells (0: more than DSQ_CMAX, or cells switched off) and the device pointers.
int capi_upload_cells(const int32_t *labels, int m, int slot, hipStream_t st, const int32_t **perm_dev,
                      const int32_t **start_dev) {
    *perm_dev = *start_dev = nullptr;
    if (!labels || !tuning().beta_cells) return 0;
    if (m >= (1 << 26)) return 0;      // the kernels pack (sample | cell << 26) into one int32
    static thread_local std::vector<int32_t> buf;
    std::map<int32_t, int> id;
    std::vector<int> cell(m);
    int C = 0;
    for (int j = 0; j < m; j++) {
        auto it = id.find(labels[j]);
        if (it == id.end()) {
            if (C == DSQ_CMAX) return 0;
            it = id.emplace(labels[j], C++).first;
        }
        cell[j] = it->second;
    }
    buf.assign((size_t)m + C + 1, 0);
    int32_t *start = buf.data(), *perm = buf.data() + C + 1;
    for (int j = 0; j < m; j++) start[cell[j] + 1]++;
    for (int c = 0; c < C; c++) start[c + 1] += start[c];
    std::vector<int> fill(start, start + C);
    for (int j = 0; j < m; j++) perm[fill[cell[j]]++] = j;
    void *v;
    if (capi_upload_table(slot, buf.data(), buf.size() * sizeof(int32_t), st, &v)) return 0;
    *start_dev = (const int32_t *)v;
    *perm_dev = (const int32_t *)v + C + 1;
    return C;
}

// cell labels of a HOST design matrix (m x p column-major): rows compared exactly
static void cells_of_host_design(const double *x, int m, int p, std::vector<int32_t> *labels) {
    labels->assign(m, 0);
    std::vector<int> reps;
    for (int j = 0; j < m; j++) {
        int found = -1;
        for (size_t c = 0; c < reps.size() && found < 0; c++) {
            bool same = true;
            for (int k = 0; k < p && same; k++) same = x[j + (size_t)m * k] == x[reps[c] + (size_t)m * k];
            if (same) found = (int)c;
        }
        if (found < 0) { found = (int)reps.size(); reps.push_back(j); }
        (*labels)[j] = found;
        if ((int)reps.size() > DSQ_CMAX) { labels->clear(); return; }
    }
}

// =============================================================== fitBeta (device)
static int fit_beta_dev_locked(const DsqFitBetaArgs *a, const DsqFitBetaOut *o, hipStream_t st) {
    if (!a || !o) return fail(DSQ_ERR_ARG, "NULL args/out");
    if (a->n < 0 || a->m < 1 || a->p < 1) return fail(DSQ_ERR_ARG, "bad dimensions n=%d m=%d p=%d", a->n, a->m, a->p);
    if (a->p > DSQ_P_WIDE)
        return fail(DSQ_ERR_UNSUPPORTED, "p=%d design columns: kernels are compiled for 1..%d", a->p, DSQ_P_WIDE);
    if (!a->y || !a->x || !a->nf || !a->alpha_hat || !a->contrast || !a->beta_mat || !a->lambda)
        return fail(DSQ_ERR_ARG, "NULL input array");
    if (a->useWeights && !a->weights) return fail(DSQ_ERR_ARG, "useWeights set but weights is NULL");
    if (!o->beta_mat || !o->beta_var_mat || !o->iter || !o->contrast_num || !o->contrast_denom || !o->deviance)
        return fail(DSQ_ERR_ARG, "NULL output array");
    if (a->maxit < 0) return fail(DSQ_ERR_ARG, "maxit < 0");
    if (a->layout == DSQ_LAYOUT_GENE_MAJOR && a->ld < a->m) return fail(DSQ_ERR_ARG, "ld < m");
    if (a->layout != DSQ_LAYOUT_R && a->layout != DSQ_LAYOUT_GENE_MAJOR) return fail(DSQ_ERR_ARG, "bad layout");
    int rc = check_device();
    if (rc) return rc;
    if (a->n == 0) return DSQ_OK;

    BetaKernelParams kp;
    memset(&kp, 0, sizeof kp);
    kp.n = a->n; kp.m = a->m; kp.p = a->p;
    bool ycheck = false;
    long ld = 0;
    rc = prep_counts(a->y, a->y_type, a->layout, a->ld, a->n, a->m, st, &kp.y, &ld, &ycheck);
    if (rc) return rc;
    kp.ld = ld;
    if (a->nf_is_vector) { kp.nf = a->nf; kp.nf_is_vector = 1; }
    else {
        rc = prep_matrix(a->nf, a->layout, a->ld, a->n, a->m, WS_NF, st, &kp.nf, ld);
        if (rc) return rc;
    }
    if (a->useWeights) {
        rc = prep_matrix(a->weights, a->layout, a->ld, a->n, a->m, WS_W, st, &kp.weights, ld);
        if (rc) return rc;
    }
    kp.x = a->x; kp.alpha_hat = a->alpha_hat; kp.contrast = a->contrast; kp.beta_init = a->beta_mat;
    kp.lambda = a->lambda;
    kp.tol = a->tol; kp.minmu = a->minmu; kp.mu_floor = o->mu_floor;
    kp.maxit = a->maxit; kp.useQR = a->useQR ? 1 : 0; kp.useWeights = a->useWeights ? 1 : 0;
    kp.ablate = tuning().ablate; kp.force_iters = tuning().force_iters;
    rc = work_counter(st, &kp.work_counter); if (rc) return rc;
    if (a->cell_of && a->ncell > 0)
        kp.ncell = capi_upload_cells(a->cell_of, a->m, WS_CELLS_BETA, st, &kp.cell_perm, &kp.cell_start);
    kp.beta_mat = o->beta_mat; kp.beta_var_mat = o->beta_var_mat; kp.iter = o->iter;
    kp.contrast_num = o->contrast_num; kp.contrast_denom = o->contrast_denom; kp.deviance = o->deviance;
    const bool wide = is_wide(a->p);
    const int pk = wide ? wide_width(a->p) : a->p;       // the kernel's design width
    double *wide_out = nullptr;
    if (wide) {
        unsigned long long padmask;
        rc = wide_pad_x(a->m, a->p, a->x, st, &kp.x, &padmask); if (rc) return rc;
        static thread_local double ones[DSQ_P_WIDE];
        for (int c = 0; c < DSQ_P_WIDE; c++) ones[c] = 1.0;
        void *v;
        rc = ws_get(WS_PAD_VEC, 2 * DSQ_P_WIDE * sizeof(double), &v); if (rc) return rc;
        double *vec = (double *)v;
        DSQ_HIP(hipMemcpyAsync(vec, ones, sizeof ones, hipMemcpyHostToDevice, st));                 // ridge 1 on padding
        DSQ_HIP(hipMemcpyAsync(vec, a->lambda, a->p * sizeof(double), hipMemcpyDeviceToDevice, st));
        DSQ_HIP(hipMemsetAsync(vec + DSQ_P_WIDE, 0, DSQ_P_WIDE * sizeof(double), st));
        DSQ_HIP(hipMemcpyAsync(vec + DSQ_P_WIDE, a->contrast, a->p * sizeof(double), hipMemcpyDeviceToDevice, st));
        kp.lambda = vec; kp.contrast = vec + DSQ_P_WIDE;
        const size_t npw = (size_t)a->n * pk;
        rc = ws_get(WS_PAD_BETA, 3 * npw * sizeof(double), &v); if (rc) return rc;
        double *bb = (double *)v;
        DSQ_HIP(hipMemsetAsync(bb, 0, npw * sizeof(double), st));
        DSQ_HIP(hipMemcpyAsync(bb, a->beta_mat, (size_t)a->n * a->p * sizeof(double), hipMemcpyDeviceToDevice, st));
        kp.beta_init = bb; kp.beta_mat = bb + npw; kp.beta_var_mat = bb + 2 * npw;
        kp.p_true = a->p;
        wide_out = bb + npw;
        kp.p = pk;
    }
    // n x m outputs: directly when gene-major, through a workspace when R layout
    double *hat_ws = nullptr, *mu_ws = nullptr;
    if (o->hat_diagonals) {
        if (a->layout == DSQ_LAYOUT_GENE_MAJOR) kp.hat_diagonals = o->hat_diagonals;
        else {
            void *b; rc = ws_get(WS_HAT, (size_t)a->n * ld * sizeof(double), &b); if (rc) return rc;
            hat_ws = (double *)b; kp.hat_diagonals = hat_ws;
        }
    }
    if (o->mu) {
        if (a->layout == DSQ_LAYOUT_GENE_MAJOR) kp.mu_out = o->mu;
        else {
            void *b; rc = ws_get(WS_MUOUT, (size_t)a->n * ld * sizeof(double), &b); if (rc) return rc;
            mu_ws = (double *)b; kp.mu_out = mu_ws;
        }
    }
    size_t slab_d = 0, cscr_d = 0;
    dispatch_beta_scratch(pk, a->n, a->m, a->useWeights, &slab_d, &cscr_d);
    {
        void *b; rc = ws_get(WS_SCRATCH, (slab_d + cscr_d) * sizeof(double) + 64, &b); if (rc) return rc;
        kp.scratch = (double *)b;
        kp.cscratch = (double *)b + slab_d;
    }
    bool ok = false;
    prof_begin(st);
    DSQ_HIP(dispatch_fit_beta(pk, kp, st, &ok));
    prof_end(st);
    if (!ok) return fail(DSQ_ERR_UNSUPPORTED, "no kernel for p=%d", a->p);
    if (wide) {     // the real coefficients are the leading columns of the padded n x 16 results
        const size_t npw = (size_t)a->n * pk, npp = (size_t)a->n * a->p * sizeof(double);
        DSQ_HIP(hipMemcpyAsync(o->beta_mat, wide_out, npp, hipMemcpyDeviceToDevice, st));
        DSQ_HIP(hipMemcpyAsync(o->beta_var_mat, wide_out + npw, npp, hipMemcpyDeviceToDevice, st));
    }
    if (hat_ws) DSQ_HIP(launch_transpose_gm_to_r_f64(hat_ws, o->hat_diagonals, a->n, a->m, ld, st));
    if (mu_ws) DSQ_HIP(launch_transpose_gm_to_r_f64(mu_ws, o->mu, a->n, a->m, ld, st));
    if (ycheck) {
        int32_t bad = 0;
        void *badp; rc = ws_get(WS_BAD, sizeof(int32_t), &badp); if (rc) return rc;
        DSQ_HIP(hipMemcpyAsync(&bad, badp, sizeof bad, hipMemcpyDeviceToHost, st));
        DSQ_HIP(hipStreamSynchronize(st));
        if (bad) return fail(DSQ_ERR_VALUE, "count matrix holds negative, non-finite or non-integer values");
    }
    return DSQ_OK;
}

// =============================================================== fitDisp (device)
static int disp_common(int n, int m, int p, int layout, long ld_in, const void *y, int y_type, const double *x,
                       const double *mu_hat, const double *weights, int useWeights, hipStream_t st,
                       DispKernelParams *kp, bool *ycheck, const int32_t *cell_of = nullptr, int ncell = 0) {
    if (n < 0 || m < 1 || p < 1) return fail(DSQ_ERR_ARG, "bad dimensions n=%d m=%d p=%d", n, m, p);
    if (p > DSQ_P_WIDE)
        return fail(DSQ_ERR_UNSUPPORTED, "p=%d design columns: kernels are compiled for 1..%d", p, DSQ_P_WIDE);
    if (!y || !x || !mu_hat) return fail(DSQ_ERR_ARG, "NULL input array");
    if (useWeights && !weights) return fail(DSQ_ERR_ARG, "useWeights set but weights is NULL");
    if (layout == DSQ_LAYOUT_GENE_MAJOR && ld_in < m) return fail(DSQ_ERR_ARG, "ld < m");
    if (layout != DSQ_LAYOUT_R && layout != DSQ_LAYOUT_GENE_MAJOR) return fail(DSQ_ERR_ARG, "bad layout");
    int rc = check_device();
    if (rc) return rc;
    memset(kp, 0, sizeof *kp);
    kp->n = n; kp->m = m; kp->p = p;
    if (n == 0) return DSQ_OK;
    long ld = 0;
    rc = prep_counts(y, y_type, layout, ld_in, n, m, st, &kp->y, &ld, ycheck);
    if (rc) return rc;
    kp->ld = ld;
    rc = prep_matrix(mu_hat, layout, ld_in, n, m, WS_MU, st, &kp->mu_hat, ld);
    if (rc) return rc;
    if (useWeights) {
        rc = prep_matrix(weights, layout, ld_in, n, m, WS_W, st, &kp->weights, ld);
        if (rc) return rc;
    }
    kp->x = x;
    kp->useWeights = useWeights ? 1 : 0;
    if (cell_of && ncell > 0 && p >= tuning().disp_cell_minp)
        kp->ncell = capi_upload_cells(cell_of, m, WS_CELLS_BETA, st, &kp->cell_perm, &kp->cell_start);
    if (is_wide(p)) {           // zero-padded design, unit diagonal on the padding in the Cox-Reid matrix
        rc = wide_pad_x(m, p, x, st, &kp->x, &kp->padmask);
        if (rc) return rc;
        kp->p = wide_width(p);
    }
    return DSQ_OK;
}

static int finish_ycheck(bool ycheck, hipStream_t st) {
    if (!ycheck) return DSQ_OK;
    int32_t bad = 0;
    void *badp;
    int rc = ws_get(WS_BAD, sizeof(int32_t), &badp);
    if (rc) return rc;
    DSQ_HIP(hipMemcpyAsync(&bad, badp, sizeof bad, hipMemcpyDeviceToHost, st));
    DSQ_HIP(hipStreamSynchronize(st));
    if (bad) return fail(DSQ_ERR_VALUE, "count matrix holds negative, non-finite or non-integer values");
    return DSQ_OK;
}

static int fit_disp_dev_locked(const DsqFitDispArgs *a, const DsqFitDispOut *o, hipStream_t st) {
    if (!a || !o) return fail(DSQ_ERR_ARG, "NULL args/out");
    if (!a->log_alpha || !a->log_alpha_prior_mean) return fail(DSQ_ERR_ARG, "NULL input vector");
    if (!o->log_alpha || !o->iter || !o->iter_accept || !o->last_change || !o->initial_lp || !o->initial_dlp ||
        !o->last_lp || !o->last_dlp)
        return fail(DSQ_ERR_ARG, "NULL output array");   // last_d2lp may be NULL: its kernel is then skipped
    if (a->maxit < 0) return fail(DSQ_ERR_ARG, "maxit < 0");
    DispKernelParams kp;
    bool ycheck = false;
    int rc = disp_common(a->n, a->m, a->p, a->layout, a->ld, a->y, a->y_type, a->x, a->mu_hat, a->weights,
                         a->useWeights, st, &kp, &ycheck, a->cell_of, a->ncell);
    if (rc) return rc;
    if (a->n == 0) return DSQ_OK;
    kp.log_alpha_in = a->log_alpha; kp.prior_mean = a->log_alpha_prior_mean;
    kp.prior_sigmasq = a->log_alpha_prior_sigmasq; kp.min_log_alpha = a->min_log_alpha;
    kp.kappa_0 = a->kappa_0; kp.tol = a->tol; kp.weightThreshold = a->weightThreshold;
    kp.maxit = a->maxit; kp.usePrior = a->usePrior ? 1 : 0; kp.useCR = a->useCR ? 1 : 0;
    kp.ablate = tuning().ablate; kp.force_iters = tuning().force_iters;
    rc = work_counter(st, &kp.work_counter); if (rc) return rc;
    kp.log_alpha = o->log_alpha; kp.iter = o->iter; kp.iter_accept = o->iter_accept;
    kp.last_change = o->last_change; kp.initial_lp = o->initial_lp; kp.initial_dlp = o->initial_dlp;
    kp.last_lp = o->last_lp; kp.last_dlp = o->last_dlp; kp.last_d2lp = o->last_d2lp;
    bool ok = false;
    prof_begin(st);
    DSQ_HIP(dispatch_fit_disp(kp.p, kp, st, false, &ok));       // (kp.p: the padded width for a wide design)
    prof_end(st);
    if (!ok) return fail(DSQ_ERR_UNSUPPORTED, "no kernel for p=%d, m=%d (49..%d columns: rows of at most 1024 samples whose working set fits the LDS)", a->p, a->m, DSQ_P_WIDE);
    return finish_ycheck(ycheck, st);
}

static int fit_disp_grid_dev_locked(const DsqFitDispGridArgs *a, const DsqFitDispGridOut *o, hipStream_t st) {
    if (!a || !o) return fail(DSQ_ERR_ARG, "NULL args/out");
    if (!a->disp_grid || !a->log_alpha_prior_mean || !o->log_alpha) return fail(DSQ_ERR_ARG, "NULL array");
    if (a->ngrid < 2) return fail(DSQ_ERR_ARG, "disp_grid needs at least 2 points");
    DispKernelParams kp;
    bool ycheck = false;
    int rc = disp_common(a->n, a->m, a->p, a->layout, a->ld, a->y, a->y_type, a->x, a->mu_hat, a->weights,
                         a->useWeights, st, &kp, &ycheck, a->cell_of, a->ncell);
    if (rc) return rc;
    if (a->n == 0) return DSQ_OK;
    kp.prior_mean = a->log_alpha_prior_mean; kp.prior_sigmasq = a->log_alpha_prior_sigmasq;
    kp.weightThreshold = a->weightThreshold;
    kp.usePrior = a->usePrior ? 1 : 0; kp.useCR = a->useCR ? 1 : 0;
    kp.grid = a->disp_grid; kp.ngrid = a->ngrid; kp.log_alpha = o->log_alpha;
    rc = work_counter(st, &kp.work_counter); if (rc) return rc;
    bool ok = false;
    prof_begin(st);
    DSQ_HIP(dispatch_fit_disp(kp.p, kp, st, true, &ok));       // (kp.p: the padded width for a wide design)
    prof_end(st);
    if (!ok) return fail(DSQ_ERR_UNSUPPORTED, "no kernel for p=%d", a->p);
    return finish_ycheck(ycheck, st);
}

// =============================================================== extensions (device)
static int prefit_dev_locked(const DsqPrefitArgs *a, const DsqPrefitOut *o, hipStream_t st) {
    if (!a || !o) return fail(DSQ_ERR_ARG, "NULL args/out");
    if (a->n < 0 || a->m < 2 || a->p < 1 || a->m <= a->p) return fail(DSQ_ERR_ARG, "bad dimensions n=%d m=%d p=%d", a->n, a->m, a->p);
    if (!a->y || !a->nf || !a->q || !a->a || !a->r) return fail(DSQ_ERR_ARG, "NULL input array");
    if (a->useWeights && !a->weights) return fail(DSQ_ERR_ARG, "useWeights set but weights is NULL");
    if (!o->baseMean || !o->baseVar || !o->allZero || !o->roughDisp || !o->beta_init) return fail(DSQ_ERR_ARG, "NULL output array");
    if (a->layout == DSQ_LAYOUT_GENE_MAJOR && a->ld < a->m) return fail(DSQ_ERR_ARG, "ld < m");
    int rc = check_device();
    if (rc) return rc;
    if (a->n == 0) return DSQ_OK;
    PrefitKernelParams kp;
    memset(&kp, 0, sizeof kp);
    kp.n = a->n; kp.m = a->m; kp.p = a->p;
    bool ycheck = false;
    long ld = 0;
    rc = prep_counts(a->y, a->y_type, a->layout, a->ld, a->n, a->m, st, &kp.y, &ld, &ycheck);
    if (rc) return rc;
    kp.ld = ld;
    if (a->nf_is_vector) { kp.nf = a->nf; kp.nf_is_vector = 1; }
    else { rc = prep_matrix(a->nf, a->layout, a->ld, a->n, a->m, WS_NF, st, &kp.nf, ld); if (rc) return rc; }
    if (a->useWeights) { rc = prep_matrix(a->weights, a->layout, a->ld, a->n, a->m, WS_W, st, &kp.weights, ld); if (rc) return rc; }
    kp.useWeights = a->useWeights ? 1 : 0;
    kp.q = a->q; kp.a = a->a; kp.r = a->r;
    kp.baseMean = o->baseMean; kp.baseVar = o->baseVar; kp.allZero = o->allZero; kp.roughDisp = o->roughDisp;
    kp.beta_init = o->beta_init;
    bool ok = false;
    prof_begin(st);
    DSQ_HIP(launch_prefit(kp, st, &ok));
    prof_end(st);
    if (!ok) return fail(DSQ_ERR_UNSUPPORTED, "p=%d design columns: kernels are compiled for 1..%d", a->p, DSQ_P_WIDE);
    return finish_ycheck(ycheck, st);
}

static int linear_mu_dev_locked(const DsqPrefitArgs *a, double mu_floor, double *mu, hipStream_t st) {
    if (!a || !mu) return fail(DSQ_ERR_ARG, "NULL args/out");
    if (a->n < 0 || a->m < 1 || a->p < 1) return fail(DSQ_ERR_ARG, "bad dimensions");
    if (!a->y || !a->nf || !a->q || !a->a) return fail(DSQ_ERR_ARG, "NULL input array");
    if (a->layout == DSQ_LAYOUT_GENE_MAJOR && a->ld < a->m) return fail(DSQ_ERR_ARG, "ld < m");
    int rc = check_device();
    if (rc) return rc;
    if (a->n == 0) return DSQ_OK;
    PrefitKernelParams kp;
    memset(&kp, 0, sizeof kp);
    kp.n = a->n; kp.m = a->m; kp.p = a->p;
    bool ycheck = false;
    long ld = 0;
    rc = prep_counts(a->y, a->y_type, a->layout, a->ld, a->n, a->m, st, &kp.y, &ld, &ycheck);
    if (rc) return rc;
    kp.ld = ld;
    if (a->nf_is_vector) { kp.nf = a->nf; kp.nf_is_vector = 1; }
    else { rc = prep_matrix(a->nf, a->layout, a->ld, a->n, a->m, WS_NF, st, &kp.nf, ld); if (rc) return rc; }
    kp.q = a->q; kp.a = a->a;
    double *dst = mu;
    if (a->layout != DSQ_LAYOUT_GENE_MAJOR) {
        void *b; rc = ws_get(WS_MUOUT, (size_t)a->n * ld * sizeof(double), &b); if (rc) return rc;
        dst = (double *)b;
    }
    bool ok = false;
    prof_begin(st);
    DSQ_HIP(launch_linear_mu(kp, mu_floor, dst, st, &ok));
    prof_end(st);
    if (!ok) return fail(DSQ_ERR_UNSUPPORTED, "p=%d design columns: kernels are compiled for 1..%d", a->p, DSQ_P_WIDE);
    if (a->layout != DSQ_LAYOUT_GENE_MAJOR) DSQ_HIP(launch_transpose_gm_to_r_f64(dst, mu, a->n, a->m, ld, st));
    return finish_ycheck(ycheck, st);
}

static int loglike_dev_locked(const DsqLogLikeArgs *a, double *out, hipStream_t st) {
    if (!a || !out) return fail(DSQ_ERR_ARG, "NULL args/out");
    if (a->n < 0 || a->m < 1) return fail(DSQ_ERR_ARG, "bad dimensions");
    if (!a->y || !a->mu || !a->disp) return fail(DSQ_ERR_ARG, "NULL input array");
    if (a->useWeights && !a->weights) return fail(DSQ_ERR_ARG, "useWeights set but weights is NULL");
    if (a->layout == DSQ_LAYOUT_GENE_MAJOR && a->ld < a->m) return fail(DSQ_ERR_ARG, "ld < m");
    int rc = check_device();
    if (rc) return rc;
    if (a->n == 0) return DSQ_OK;
    LogLikeKernelParams kp;
    memset(&kp, 0, sizeof kp);
    kp.n = a->n; kp.m = a->m;
    bool ycheck = false;
    long ld = 0;
    rc = prep_counts(a->y, a->y_type, a->layout, a->ld, a->n, a->m, st, &kp.y, &ld, &ycheck);
    if (rc) return rc;
    kp.ld = ld;
    rc = prep_matrix(a->mu, a->layout, a->ld, a->n, a->m, WS_MU, st, &kp.mu, ld);
    if (rc) return rc;
    if (a->useWeights) { rc = prep_matrix(a->weights, a->layout, a->ld, a->n, a->m, WS_W, st, &kp.weights, ld); if (rc) return rc; }
    kp.useWeights = a->useWeights ? 1 : 0;
    kp.disp = a->disp; kp.loglike = out;
    prof_begin(st);
    DSQ_HIP(launch_loglike(kp, st));
    prof_end(st);
    return finish_ycheck(ycheck, st);
}

static int intercept_dev_locked(const DsqInterceptArgs *a, const DsqInterceptOut *o, hipStream_t st) {
    if (!a || !o) return fail(DSQ_ERR_ARG, "NULL args/out");
    if (a->n < 0 || a->m < 1) return fail(DSQ_ERR_ARG, "bad dimensions");
    if (!a->y || !a->nf || !a->alpha) return fail(DSQ_ERR_ARG, "NULL input array");
    if (a->useWeights && !a->weights) return fail(DSQ_ERR_ARG, "useWeights set but weights is NULL");
    if (!o->beta_log2 || !o->betaSE) return fail(DSQ_ERR_ARG, "NULL output array");
    if (a->layout == DSQ_LAYOUT_GENE_MAJOR && a->ld < a->m) return fail(DSQ_ERR_ARG, "ld < m");
    int rc = check_device();
    if (rc) return rc;
    if (a->n == 0) return DSQ_OK;
    InterceptKernelParams kp;
    memset(&kp, 0, sizeof kp);
    kp.n = a->n; kp.m = a->m;
    bool ycheck = false;
    long ld = 0;
    rc = prep_counts(a->y, a->y_type, a->layout, a->ld, a->n, a->m, st, &kp.y, &ld, &ycheck);
    if (rc) return rc;
    kp.ld = ld;
    if (a->nf_is_vector) { kp.nf = a->nf; kp.nf_is_vector = 1; }
    else { rc = prep_matrix(a->nf, a->layout, a->ld, a->n, a->m, WS_NF, st, &kp.nf, ld); if (rc) return rc; }
    if (a->useWeights) { rc = prep_matrix(a->weights, a->layout, a->ld, a->n, a->m, WS_W, st, &kp.weights, ld); if (rc) return rc; }
    kp.useWeights = a->useWeights ? 1 : 0;
    kp.alpha = a->alpha; kp.mu_floor = a->mu_floor;
    kp.beta_log2 = o->beta_log2; kp.betaSE = o->betaSE;
    double *mu_ws = nullptr, *hat_ws = nullptr;
    if (o->mu) {
        if (a->layout == DSQ_LAYOUT_GENE_MAJOR) kp.mu_out = o->mu;
        else { void *b; rc = ws_get(WS_MUOUT, (size_t)a->n * ld * sizeof(double), &b); if (rc) return rc; mu_ws = (double *)b; kp.mu_out = mu_ws; }
    }
    if (o->hat) {
        if (a->layout == DSQ_LAYOUT_GENE_MAJOR) kp.hat = o->hat;
        else { void *b; rc = ws_get(WS_HAT, (size_t)a->n * ld * sizeof(double), &b); if (rc) return rc; hat_ws = (double *)b; kp.hat = hat_ws; }
    }
    prof_begin(st);
    DSQ_HIP(launch_intercept_fit(kp, st));
    prof_end(st);
    if (mu_ws) DSQ_HIP(launch_transpose_gm_to_r_f64(mu_ws, o->mu, a->n, a->m, ld, st));
    if (hat_ws) DSQ_HIP(launch_transpose_gm_to_r_f64(hat_ws, o->hat, a->n, a->m, ld, st));
    return finish_ycheck(ycheck, st);
}

// ---- host-pointer staging helpers --------------------------------------------------
// =============================================================== Cook's distances / replaceOutliers
static int next_pow2(int n) { int v = 2; while (v < n) v <<= 1; return v; }

static int cooks_dev_locked(const DsqCooksArgs *a, const DsqCooksOut *o, hipStream_t st) {
    if (!a || !o) return fail(DSQ_ERR_ARG, "NULL args/out");
    if (a->n < 0 || a->m < 1 || a->p < 1 || a->ncell < 1) return fail(DSQ_ERR_ARG, "bad dimensions");
    if (!a->y || !a->nf || !a->mu || !a->H || !a->cell_of) return fail(DSQ_ERR_ARG, "NULL input array");
    if (!o->cooks || !o->maxCooks) return fail(DSQ_ERR_ARG, "NULL output array");
    if (a->layout == DSQ_LAYOUT_GENE_MAJOR && a->ld < a->m) return fail(DSQ_ERR_ARG, "ld < m");
    const int m = a->m;
    // design cells -> sample permutation grouped by cell, offsets, ">= 3 in cell" flags
    static thread_local std::vector<int32_t> meta;
    meta.assign((size_t)2 * m + a->ncell + 1, 0);
    int32_t *perm = meta.data(), *in3 = perm + m, *start = in3 + m;
    for (int j = 0; j < m; j++) {
        if (a->cell_of[j] < 0 || a->cell_of[j] >= a->ncell) return fail(DSQ_ERR_VALUE, "cell_of[%d] out of range", j);
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
    for (int j = 0; j < m; j++) in3[j] = (start[a->cell_of[j] + 1] - start[a->cell_of[j]]) >= 3;
    int rc = check_device();
    if (rc) return rc;
    if (a->n == 0) return DSQ_OK;
    CooksKernelParams kp;
    memset(&kp, 0, sizeof kp);
    kp.n = a->n; kp.m = m; kp.p = a->p; kp.ncell = a->ncell; kp.any3 = any3;
    kp.sortcap = next_pow2(any3 ? maxcell : m);
    bool ycheck = false;
    long ld = 0;
    rc = prep_counts(a->y, a->y_type, a->layout, a->ld, a->n, m, st, &kp.y, &ld, &ycheck);
    if (rc) return rc;
    kp.ld = ld;
    kp.nf_is_vector = a->nf_is_vector ? 1 : 0;
    if (kp.nf_is_vector) kp.nf = a->nf;
    else { rc = prep_matrix(a->nf, a->layout, a->ld, a->n, m, WS_NF, st, &kp.nf, ld); if (rc) return rc; }
    rc = prep_matrix(a->mu, a->layout, a->ld, a->n, m, WS_MU, st, &kp.mu, ld); if (rc) return rc;
    rc = prep_matrix(a->H, a->layout, a->ld, a->n, m, WS_W, st, &kp.H, ld); if (rc) return rc;
    void *v;
    rc = ws_get(WS_CELLS, meta.size() * sizeof(int32_t) + (size_t)a->n * 8, &v); if (rc) return rc;
    DSQ_HIP(hipMemcpyAsync(v, meta.data(), meta.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
    kp.perm = (int32_t *)v; kp.in3 = kp.perm + m; kp.cell_start = kp.in3 + m;
    kp.maxCooks = o->maxCooks;
    if (o->robustDisp) kp.robustDisp = o->robustDisp;
    else kp.robustDisp = (double *)((char *)v + ((meta.size() * sizeof(int32_t) + 7) & ~(size_t)7));
    if (a->layout == DSQ_LAYOUT_GENE_MAJOR) kp.cooks = o->cooks;
    else { void *b; rc = ws_get(WS_HAT, (size_t)a->n * ld * sizeof(double), &b); if (rc) return rc; kp.cooks = (double *)b; }
    bool ok = true;
    prof_begin(st);
    DSQ_HIP(launch_cooks(kp, st, &ok));
    prof_end(st);
    if (!ok) return fail(DSQ_ERR_UNSUPPORTED, "m=%d samples: a gene row plus its sort buffer exceeds the 160 KiB LDS", m);
    if (a->layout != DSQ_LAYOUT_GENE_MAJOR) DSQ_HIP(launch_transpose_gm_to_r_f64(kp.cooks, o->cooks, a->n, m, ld, st));
    return finish_ycheck(ycheck, st);
}

static int replace_dev_locked(const DsqReplaceArgs *a, const DsqReplaceOut *o, hipStream_t st) {
    if (!a || !o) return fail(DSQ_ERR_ARG, "NULL args/out");
    if (a->n < 0 || a->m < 1) return fail(DSQ_ERR_ARG, "bad dimensions");
    if (!a->y || !a->nf || !a->cooks || !a->replaceable) return fail(DSQ_ERR_ARG, "NULL input array");
    if (!o->newCounts || !o->replace) return fail(DSQ_ERR_ARG, "NULL output array");
    if (!(a->trim >= 0.0 && a->trim < 0.5)) return fail(DSQ_ERR_ARG, "trim must be in [0, 0.5)");
    if (a->layout == DSQ_LAYOUT_GENE_MAJOR && a->ld < a->m) return fail(DSQ_ERR_ARG, "ld < m");
    int rc = check_device();
    if (rc) return rc;
    if (a->n == 0) return DSQ_OK;
    const int m = a->m;
    ReplaceKernelParams kp;
    memset(&kp, 0, sizeof kp);
    kp.n = a->n; kp.m = m; kp.cutoff = a->cooksCutoff; kp.trim = a->trim; kp.sortcap = next_pow2(m);
    bool ycheck = false;
    long ld = 0;
    rc = prep_counts(a->y, a->y_type, a->layout, a->ld, a->n, m, st, &kp.y, &ld, &ycheck);
    if (rc) return rc;
    kp.ld = ld;
    kp.nf_is_vector = a->nf_is_vector ? 1 : 0;
    if (kp.nf_is_vector) kp.nf = a->nf;
    else { rc = prep_matrix(a->nf, a->layout, a->ld, a->n, m, WS_NF, st, &kp.nf, ld); if (rc) return rc; }
    rc = prep_matrix(a->cooks, a->layout, a->ld, a->n, m, WS_COOKS_IN, st, &kp.cooks, ld); if (rc) return rc;
    static thread_local std::vector<int32_t> flags;
    flags.assign(a->replaceable, a->replaceable + m);
    void *v;
    rc = ws_get(WS_CELLS, (size_t)m * sizeof(int32_t), &v); if (rc) return rc;
    DSQ_HIP(hipMemcpyAsync(v, flags.data(), (size_t)m * sizeof(int32_t), hipMemcpyHostToDevice, st));
    kp.replaceable = (int32_t *)v;
    kp.replace = o->replace;
    if (a->layout == DSQ_LAYOUT_GENE_MAJOR) kp.newCounts = o->newCounts;
    else { void *b; rc = ws_get(WS_HAT, (size_t)a->n * ld * sizeof(int32_t), &b); if (rc) return rc; kp.newCounts = (int32_t *)b; }
    bool ok = true;
    prof_begin(st);
    DSQ_HIP(launch_replace(kp, st, &ok));
    prof_end(st);
    if (!ok) return fail(DSQ_ERR_UNSUPPORTED, "m=%d samples: the sort buffer exceeds the 160 KiB LDS", m);
    if (a->layout != DSQ_LAYOUT_GENE_MAJOR) DSQ_HIP(launch_transpose_gm_to_r_i32(kp.newCounts, o->newCounts, a->n, m, ld, st));
    return finish_ycheck(ycheck, st);
}

static int up(int slot, const void *host, size_t bytes, hipStream_t st, void **dev) {
    int rc = ws_get(slot, bytes ? bytes : 8, dev);
    if (rc) return rc;
    if (bytes) return stage_h2d(*dev, host, 1, bytes, 0, bytes, 1, st);
    return DSQ_OK;
}
// device -> pageable host memory, complete on return
static int down(void *host, const void *dev, size_t bytes, hipStream_t st) {
    return stage_d2h(host, dev, 1, bytes, 0, bytes, 1, st);
}

}  // namespace dsq

using namespace dsq;

extern "C" {

int dsq_version(void) { return DSQ_VERSION; }
const char *dsq_last_error(void) { return g_err; }

int dsq_device_count(void) {
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess) return 0;
    return cnt;
}

int dsq_set_device(int device) {
    if (hipSetDevice(device) != hipSuccess) return fail(DSQ_ERR_DEVICE, "hipSetDevice(%d) failed", device);
    return DSQ_OK;
}

int dsq_profile_enable(int on) {
    std::lock_guard<std::mutex> lk(g_mu);
    WsScope ws(nullptr);
    g_prof = on != 0;
    prof_clear();
    return DSQ_OK;
}

static double prof_ms(const ProfEntry &p) {
    float ms = 0.f;
    if (hipEventSynchronize(p.e1) != hipSuccess || hipEventElapsedTime(&ms, p.e0, p.e1) != hipSuccess) return -1.0;
    return (double)ms;
}

double dsq_profile_last_ms(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    WsScope ws(nullptr);
    if (g_prof_list.empty()) return -1.0;
    return prof_ms(g_prof_list.back());
}

int dsq_profile_count(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    return (int)g_prof_list.size();
}

int dsq_profile_get(int i, char *name, int cap, int32_t *genes, double *ms) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (i < 0 || i >= (int)g_prof_list.size()) return fail(DSQ_ERR_ARG, "profile entry %d out of range", i);
    const ProfEntry &p = g_prof_list[i];
    if (name && cap > 0) snprintf(name, (size_t)cap, "%s", p.name);
    if (genes) *genes = p.n;
    if (ms) *ms = prof_ms(p);
    return DSQ_OK;
}

int dsq_release_workspace(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    WsScope ws(nullptr);
    int cur = 0;
    (void)hipGetDevice(&cur);
    for (int d = 0; d < 64; d++) {
        if (g_pool[d].empty()) continue;
        (void)hipSetDevice(d);
        (void)hipDeviceSynchronize();
        for (auto &kv : g_pool[d]) for (auto &s : kv.second) if (s.p) { (void)hipFree(s.p); s.p = nullptr; s.bytes = 0; }
        g_pool[d].clear();
    }
    (void)hipSetDevice(cur);
    return DSQ_OK;
}

int dsq_fit_beta_dev(const DsqFitBetaArgs *args, const DsqFitBetaOut *out, void *stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    WsScope ws((hipStream_t)stream);
    return fit_beta_dev_locked(args, out, (hipStream_t)stream);
}
int dsq_fit_disp_dev(const DsqFitDispArgs *args, const DsqFitDispOut *out, void *stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    WsScope ws((hipStream_t)stream);
    return fit_disp_dev_locked(args, out, (hipStream_t)stream);
}
int dsq_fit_disp_grid_dev(const DsqFitDispGridArgs *args, const DsqFitDispGridOut *out, void *stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    WsScope ws((hipStream_t)stream);
    return fit_disp_grid_dev_locked(args, out, (hipStream_t)stream);
}

int dsq_weights_prep_dev(const double *weights_raw, const double *x, int32_t n, int32_t m, int32_t p, int64_t ld,
                         double weightThreshold, double *w_norm, double *w_floor, int32_t *weightsFail,
                         int32_t *any_negative, void *stream) {
    if (!weights_raw || !x || !w_norm || !w_floor || !weightsFail || !any_negative || n < 0 || m < 1 || ld < m)
        return fail(DSQ_ERR_ARG, "bad arguments");
    if (p < 1 || p > DSQ_P_WIDE) return fail(DSQ_ERR_UNSUPPORTED, "dsq_weights_prep_dev: p=%d design columns (1..%d)", p, DSQ_P_WIDE);
    if (int rc = check_device()) return rc;
    if (n == 0) return DSQ_OK;
    DSQ_HIP(launch_weights_prep(weights_raw, x, n, m, p, ld, weightThreshold, w_norm, w_floor, weightsFail, any_negative,
                                (hipStream_t)stream));
    return DSQ_OK;
}
int dsq_xim_dev(const double *nf, int32_t n, int32_t m, int64_t ld, double *scratch_m, double *out, void *stream) {
    if (!nf || !scratch_m || !out || n < 1 || m < 1 || ld < m) return fail(DSQ_ERR_ARG, "bad arguments");
    if (int rc = check_device()) return rc;
    DSQ_HIP(launch_xim(nf, n, m, ld, scratch_m, out, (hipStream_t)stream));
    return DSQ_OK;
}

int dsq_to_gene_major_f64(const double *src_r, double *dst_gm, int32_t n, int32_t m, int64_t ld, void *stream) {
    if (!src_r || !dst_gm || n < 0 || m < 1 || ld < m) return fail(DSQ_ERR_ARG, "bad arguments");
    if (int rc = check_device()) return rc;
    if (n == 0) return DSQ_OK;
    DSQ_HIP(launch_transpose_r_to_gm_f64(src_r, dst_gm, n, m, ld, (hipStream_t)stream));
    return DSQ_OK;
}
int dsq_to_gene_major_i32(const int32_t *src_r, int32_t *dst_gm, int32_t n, int32_t m, int64_t ld, void *stream) {
    if (!src_r || !dst_gm || n < 0 || m < 1 || ld < m) return fail(DSQ_ERR_ARG, "bad arguments");
    if (int rc = check_device()) return rc;
    if (n == 0) return DSQ_OK;
    DSQ_HIP(launch_transpose_r_to_gm_i32(src_r, dst_gm, n, m, ld, (hipStream_t)stream));
    return DSQ_OK;
}
int dsq_counts_f64_to_gene_major_i32(const double *src_r, int32_t *dst_gm, int32_t n, int32_t m, int64_t ld,
                                     int32_t *bad, void *stream) {
    if (!src_r || !dst_gm || !bad || n < 0 || m < 1 || ld < m) return fail(DSQ_ERR_ARG, "bad arguments");
    if (int rc = check_device()) return rc;
    if (n == 0) return DSQ_OK;
    DSQ_HIP(launch_counts_f64_to_gm_i32(src_r, dst_gm, n, m, ld, bad, (hipStream_t)stream));
    return DSQ_OK;
}
int dsq_from_gene_major_f64(const double *src_gm, double *dst_r, int32_t n, int32_t m, int64_t ld, void *stream) {
    if (!src_gm || !dst_r || n < 0 || m < 1 || ld < m) return fail(DSQ_ERR_ARG, "bad arguments");
    if (int rc = check_device()) return rc;
    if (n == 0) return DSQ_OK;
    DSQ_HIP(launch_transpose_gm_to_r_f64(src_gm, dst_r, n, m, ld, (hipStream_t)stream));
    return DSQ_OK;
}

// ------------------------------------------------------------ host-pointer entries
// (what src/r_shim.c binds: R memory in, R memory out, synchronous)
//
// Genes are independent inside every native routine (src/DESeq2.cpp:194,319,492) and the reference's only parallelism
// splits them into contiguous ranges (R/parallel.R:10).  The host-pointer entry points do the same INSIDE the library:
// [0, n) is cut into one range per visible device (idx <- sort(rep(seq_len(G), length.out = n))), each range is
// uploaded / fitted / downloaded by a persistent worker thread bound to its device and its own stream, so an
// unchanged R session calling .Call("fitBeta", ...) uses every GPU of the node.  DSQ_HOST_DEVICES caps the number
// of devices, DSQ_HOST_SHARDS forces a number of ranges (ranges beyond the device count share devices: used by the
// tests to exercise the split on one GPU).
} // extern "C"

namespace dsq {

// rows [lo, lo + cnt) of a column-major n x cols host matrix <-> a column-major cnt x cols device matrix
static int up_rows(int slot, const void *host, size_t elem, size_t n, size_t lo, size_t cnt, size_t cols, hipStream_t st,
                   void **dev) {
    int rc = ws_get(slot, cnt * cols * elem ? cnt * cols * elem : 8, dev);
    if (rc) return rc;
    return stage_h2d(*dev, host, elem, n, lo, cnt, cols, st);
}
static int down_rows(void *host, const void *dev, size_t elem, size_t n, size_t lo, size_t cnt, size_t cols, hipStream_t st) {
    return stage_d2h(host, dev, elem, n, lo, cnt, cols, st);
}

static int fit_beta_host_range(const DsqFitBetaArgs *a, const DsqFitBetaOut *o, size_t lo, size_t cnt, hipStream_t st,
                               const int32_t *cells, int ncell) {
    const size_t n = a->n, m = a->m, p = a->p;
    const size_t ye = a->y_type == DSQ_Y_INT32 ? 4 : 8;
    DsqFitBetaArgs d = *a;
    DsqFitBetaOut od = *o;
    d.n = (int32_t)cnt;
    d.cell_of = cells; d.ncell = ncell;
    void *v;
    int rc;
    if ((rc = up_rows(WS_H_Y, a->y, ye, n, lo, cnt, m, st, &v))) return rc; d.y = v;
    // x, alpha_hat, contrast, beta_mat, lambda share one staging buffer
    size_t off_x = 0, off_alpha = off_x + m * p, off_con = off_alpha + cnt, off_beta = off_con + p,
           off_lam = off_beta + cnt * p, tot = off_lam + p;
    if ((rc = ws_get(WS_H_VEC, tot * 8, &v))) return rc;
    double *vec = (double *)v;
    DSQ_HIP(hipMemcpyAsync(vec + off_x, a->x, m * p * 8, hipMemcpyHostToDevice, st));
    DSQ_HIP(hipMemcpyAsync(vec + off_alpha, a->alpha_hat + lo, cnt * 8, hipMemcpyHostToDevice, st));
    DSQ_HIP(hipMemcpyAsync(vec + off_con, a->contrast, p * 8, hipMemcpyHostToDevice, st));
    if (cnt == n) DSQ_HIP(hipMemcpyAsync(vec + off_beta, a->beta_mat, n * p * 8, hipMemcpyHostToDevice, st));
    else DSQ_HIP(hipMemcpy2DAsync(vec + off_beta, cnt * 8, a->beta_mat + lo, n * 8, cnt * 8, p, hipMemcpyHostToDevice, st));
    DSQ_HIP(hipMemcpyAsync(vec + off_lam, a->lambda, p * 8, hipMemcpyHostToDevice, st));
    d.x = vec + off_x; d.alpha_hat = vec + off_alpha; d.contrast = vec + off_con; d.beta_mat = vec + off_beta;
    d.lambda = vec + off_lam;
    if (a->nf_is_vector) { if ((rc = up_rows(WS_H_NF, a->nf, 8, m, 0, m, 1, st, &v))) return rc; }
    else if ((rc = up_rows(WS_H_NF, a->nf, 8, n, lo, cnt, m, st, &v))) return rc;
    d.nf = (double *)v;
    if (a->useWeights) { if ((rc = up_rows(WS_H_W, a->weights, 8, n, lo, cnt, m, st, &v))) return rc; d.weights = (double *)v; }
    else d.weights = nullptr;
    // outputs
    size_t o_beta = 0, o_var = o_beta + cnt * p, o_iter = o_var + cnt * p, o_cn = o_iter + cnt, o_cd = o_cn + cnt,
           o_dev = o_cd + cnt, o_tot = o_dev + cnt;
    if ((rc = ws_get(WS_H_OUTVEC, o_tot * 8, &v))) return rc;
    double *ov = (double *)v;
    od.beta_mat = ov + o_beta; od.beta_var_mat = ov + o_var; od.iter = ov + o_iter; od.contrast_num = ov + o_cn;
    od.contrast_denom = ov + o_cd; od.deviance = ov + o_dev;
    double *hat_d = nullptr, *mu_d = nullptr;
    if (o->hat_diagonals) { if ((rc = ws_get(WS_H_OUTMAT, cnt * m * 8, &v))) return rc; hat_d = (double *)v; }
    if (o->mu) { if ((rc = ws_get(WS_H_OUTMAT2, cnt * m * 8, &v))) return rc; mu_d = (double *)v; }
    od.hat_diagonals = hat_d; od.mu = mu_d;
    rc = fit_beta_dev_locked(&d, &od, st);
    if (rc) return rc;
    if ((rc = down_rows(o->beta_mat, od.beta_mat, 8, n, lo, cnt, p, st))) return rc;
    if ((rc = down_rows(o->beta_var_mat, od.beta_var_mat, 8, n, lo, cnt, p, st))) return rc;
    DSQ_HIP(hipMemcpyAsync(o->iter + lo, od.iter, cnt * 8, hipMemcpyDeviceToHost, st));
    DSQ_HIP(hipMemcpyAsync(o->contrast_num + lo, od.contrast_num, cnt * 8, hipMemcpyDeviceToHost, st));
    DSQ_HIP(hipMemcpyAsync(o->contrast_denom + lo, od.contrast_denom, cnt * 8, hipMemcpyDeviceToHost, st));
    DSQ_HIP(hipMemcpyAsync(o->deviance + lo, od.deviance, cnt * 8, hipMemcpyDeviceToHost, st));
    if (hat_d && (rc = down_rows(o->hat_diagonals, hat_d, 8, n, lo, cnt, m, st))) return rc;
    if (mu_d && (rc = down_rows(o->mu, mu_d, 8, n, lo, cnt, m, st))) return rc;
    DSQ_HIP(hipStreamSynchronize(st));
    return DSQ_OK;
}

static int disp_host_stage(size_t n, size_t lo, size_t cnt, int m_, int p_, const void *y, int y_type, const double *x,
                           const double *mu_hat, const double *weights, int useWeights, hipStream_t st, const void **yd,
                           const double **xd, const double **mud, const double **wd) {
    const size_t m = m_, p = p_;
    void *v;
    int rc;
    if ((rc = up_rows(WS_H_Y, y, y_type == DSQ_Y_INT32 ? 4 : 8, n, lo, cnt, m, st, &v))) return rc; *yd = v;
    if ((rc = up_rows(WS_H_X, x, 8, m, 0, m, p, st, &v))) return rc; *xd = (double *)v;
    if ((rc = up_rows(WS_H_MU, mu_hat, 8, n, lo, cnt, m, st, &v))) return rc; *mud = (double *)v;
    if (useWeights) { if ((rc = up_rows(WS_H_W, weights, 8, n, lo, cnt, m, st, &v))) return rc; *wd = (double *)v; }
    else *wd = nullptr;
    return DSQ_OK;
}

static int fit_disp_host_range(const DsqFitDispArgs *a, const DsqFitDispOut *o, size_t lo, size_t cnt, hipStream_t st,
                               const int32_t *cells, int ncell) {
    const size_t n = a->n;
    DsqFitDispArgs d = *a;
    DsqFitDispOut od = *o;
    d.n = (int32_t)cnt;
    d.cell_of = cells; d.ncell = ncell;
    int rc = disp_host_stage(n, lo, cnt, a->m, a->p, a->y, a->y_type, a->x, a->mu_hat, a->weights, a->useWeights, st,
                             &d.y, &d.x, &d.mu_hat, &d.weights);
    if (rc) return rc;
    void *v;
    if ((rc = ws_get(WS_H_VEC, 2 * cnt * 8 + 8, &v))) return rc;
    double *vec = (double *)v;
    DSQ_HIP(hipMemcpyAsync(vec, a->log_alpha + lo, cnt * 8, hipMemcpyHostToDevice, st));
    DSQ_HIP(hipMemcpyAsync(vec + cnt, a->log_alpha_prior_mean + lo, cnt * 8, hipMemcpyHostToDevice, st));
    d.log_alpha = vec; d.log_alpha_prior_mean = vec + cnt;
    if ((rc = ws_get(WS_H_OUTVEC, 8 * cnt * 8 + 8, &v))) return rc;
    double *ov = (double *)v;
    od.log_alpha = ov; od.last_change = ov + cnt; od.initial_lp = ov + 2 * cnt; od.initial_dlp = ov + 3 * cnt;
    od.last_lp = ov + 4 * cnt; od.last_dlp = ov + 5 * cnt; od.last_d2lp = ov + 6 * cnt;
    od.iter = (int32_t *)(ov + 7 * cnt); od.iter_accept = od.iter + cnt;
    rc = fit_disp_dev_locked(&d, &od, st);
    if (rc) return rc;
    double *const dst[7] = {o->log_alpha, o->last_change, o->initial_lp, o->initial_dlp, o->last_lp, o->last_dlp, o->last_d2lp};
    for (int k = 0; k < 7; k++) DSQ_HIP(hipMemcpyAsync(dst[k] + lo, ov + k * cnt, cnt * 8, hipMemcpyDeviceToHost, st));
    DSQ_HIP(hipMemcpyAsync(o->iter + lo, od.iter, cnt * 4, hipMemcpyDeviceToHost, st));
    DSQ_HIP(hipMemcpyAsync(o->iter_accept + lo, od.iter_accept, cnt * 4, hipMemcpyDeviceToHost, st));
    DSQ_HIP(hipStreamSynchronize(st));
    return DSQ_OK;
}

static int fit_disp_grid_host_range(const DsqFitDispGridArgs *a, const DsqFitDispGridOut *o, size_t lo, size_t cnt,
                                    hipStream_t st, const int32_t *cells, int ncell) {
    const size_t n = a->n, ng = a->ngrid;
    DsqFitDispGridArgs d = *a;
    DsqFitDispGridOut od = *o;
    d.n = (int32_t)cnt;
    d.cell_of = cells; d.ncell = ncell;
    int rc = disp_host_stage(n, lo, cnt, a->m, a->p, a->y, a->y_type, a->x, a->mu_hat, a->weights, a->useWeights, st,
                             &d.y, &d.x, &d.mu_hat, &d.weights);
    if (rc) return rc;
    void *v;
    if ((rc = ws_get(WS_H_VEC, (cnt + ng) * 8, &v))) return rc;
    double *vec = (double *)v;
    DSQ_HIP(hipMemcpyAsync(vec, a->log_alpha_prior_mean + lo, cnt * 8, hipMemcpyHostToDevice, st));
    DSQ_HIP(hipMemcpyAsync(vec + cnt, a->disp_grid, ng * 8, hipMemcpyHostToDevice, st));
    d.log_alpha_prior_mean = vec; d.disp_grid = vec + cnt;
    if ((rc = ws_get(WS_H_OUTVEC, cnt * 8 + 8, &v))) return rc;
    od.log_alpha = (double *)v;
    rc = fit_disp_grid_dev_locked(&d, &od, st);
    if (rc) return rc;
    DSQ_HIP(hipMemcpyAsync(o->log_alpha + lo, od.log_alpha, cnt * 8, hipMemcpyDeviceToHost, st));
    DSQ_HIP(hipStreamSynchronize(st));
    return DSQ_OK;
}

// ---- worker threads: one per (device, lane); each owns a stream and (through thread_local state) its workspaces ----
struct HostWorker {
    int dev = 0;
    hipStream_t st = nullptr;
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    std::function<int()> job;
    bool has_job = false, done = false;
    int rc = 0;
    char err[512] = "";
    void loop() {
        (void)hipSetDevice(dev);
        (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
        for (;;) {
            std::unique_lock<std::mutex> lk(m);
            cv.wait(lk, [&] { return has_job; });
            std::function<int()> j = std::move(job);
            has_job = false;
            lk.unlock();
            g_ws_stream = st;
            int r = j();
            lk.lock();
            rc = r;
            snprintf(err, sizeof err, "%s", g_err);
            done = true;
            cv.notify_all();
        }
    }
};
static std::vector<HostWorker *> g_workers;      // grown under g_mu; worker k serves device k % ndev

static HostWorker *host_worker(int k, int ndev) {
    while ((int)g_workers.size() <= k) {
        HostWorker *w = new HostWorker();
        w->dev = (int)g_workers.size() % ndev;
        w->th = std::thread([w] { w->loop(); });
        w->th.detach();
        g_workers.push_back(w);
    }
    return g_workers[k];
}

// number of gene ranges of a host-pointer call over n genes, and the devices they go to
static void host_plan(size_t n, int *nshards, int *ndev) {
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || cnt < 1) cnt = 1;
    const int cap = env_int("DSQ_HOST_DEVICES", 0);
    if (cap > 0 && cap < cnt) cnt = cap;
    int s = env_int("DSQ_HOST_SHARDS", 0);
    if (s <= 0) s = cnt;
    if ((size_t)s > n) s = (int)n;
    if (s < 1) s = 1;
    *nshards = s; *ndev = cnt;
}

// run f(lo, cnt, stream, range index, number of ranges) over the ranges of R/parallel.R:10; one range: on the caller's
// thread, device and null stream
template <class F>
static int host_sharded_ix(size_t row_lo, size_t n, F &&f0, int max_shards = 0) {
    int S, ndev;
    host_plan(n, &S, &ndev);
    if (max_shards > 0 && S > max_shards) S = max_shards;
    auto f = [&](size_t lo, size_t cnt, hipStream_t st, int k) { return f0(row_lo + lo, cnt, st, k, S < 1 ? 1 : S); };
    if (S <= 1) return f((size_t)0, n, (hipStream_t) nullptr, 0);
    std::vector<HostWorker *> ws(S);
    const size_t big = n / S + 1, nbig = n % S, small = n / S;      // the first n %% S ranges hold one gene more
    size_t lo = 0;
    for (int k = 0; k < S; k++) {
        const size_t cnt = (size_t)k < nbig ? big : small;
        HostWorker *w = ws[k] = host_worker(k, ndev);
        {
            std::lock_guard<std::mutex> lk(w->m);
            w->job = [&f, lo, cnt, w, k] { return f(lo, cnt, w->st, k); };
            w->has_job = true; w->done = false;
        }
        w->cv.notify_all();
        lo += cnt;
    }
    int rc = DSQ_OK;
    for (int k = 0; k < S; k++) {
        HostWorker *w = ws[k];
        std::unique_lock<std::mutex> lk(w->m);
        w->cv.wait(lk, [&] { return w->done; });
        if (w->rc && !rc) { rc = w->rc; snprintf(g_err, sizeof g_err, "%s", w->err); }
    }
    return rc;
}

template <class F>
static int host_sharded(size_t row_lo, size_t n, F &&f0) {
    return host_sharded_ix(row_lo, n, [&](size_t lo, size_t cnt, hipStream_t st, int, int) { return f0(lo, cnt, st); });
}
// (deseq_host.hip) the caller holds the library's call lock
int capi_host_sharded(size_t n, const std::function<int(size_t, size_t, hipStream_t, int, int)> &f, int max_shards) {
    return host_sharded_ix((size_t)0, n, f, max_shards);
}
int capi_host_shards(size_t n) {
    int S, ndev;
    host_plan(n, &S, &ndev);
    return S < 1 ? 1 : S;
}

static void host_cells(const double *x, int m, int p, const int32_t *given, int ngiven, std::vector<int32_t> *labels,
                       const int32_t **cells, int *ncell) {
    *cells = given; *ncell = ngiven;
    if (given) return;
    cells_of_host_design(x, m, p, labels);          // R hands over the design matrix itself: find its cells here
    if (!labels->empty()) { *cells = labels->data(); *ncell = 1 + *std::max_element(labels->begin(), labels->end()); }
}

}  // namespace dsq

extern "C" {

int dsq_fit_beta(const DsqFitBetaArgs *a, const DsqFitBetaOut *o) { return dsq_fit_beta_rows(a, o, 0, a ? a->n : 0); }

int dsq_fit_beta_rows(const DsqFitBetaArgs *a, const DsqFitBetaOut *o, int64_t row_lo, int64_t row_cnt) {
    std::lock_guard<std::mutex> lk(g_mu);
    WsScope ws(nullptr);
    if (!a || !o) return fail(DSQ_ERR_ARG, "NULL args/out");
    if (a->layout != DSQ_LAYOUT_R) return fail(DSQ_ERR_ARG, "host entry points take R layout only");
    if (a->n < 0 || a->m < 1 || a->p < 1) return fail(DSQ_ERR_ARG, "bad dimensions");
    if (!a->y || !a->x || !a->nf || !a->alpha_hat || !a->contrast || !a->beta_mat || !a->lambda)
        return fail(DSQ_ERR_ARG, "NULL input array");
    if (a->useWeights && !a->weights) return fail(DSQ_ERR_ARG, "useWeights set but weights is NULL");
    if (!o->beta_mat || !o->beta_var_mat || !o->iter || !o->contrast_num || !o->contrast_denom || !o->deviance)
        return fail(DSQ_ERR_ARG, "NULL output array");
    if (int rc = check_device()) return rc;
    if (row_lo < 0 || row_cnt < 0 || row_lo + row_cnt > a->n) return fail(DSQ_ERR_ARG, "row range outside [0, n)");
    if (row_cnt == 0) return DSQ_OK;
    std::vector<int32_t> labels;
    const int32_t *cells; int ncell;
    host_cells(a->x, a->m, a->p, a->cell_of, a->ncell, &labels, &cells, &ncell);
    if (row_lo == 0) {          // the n x m results land in fresh pages: take the faults while the inputs go up (stage.hip)
        stage_prefault(o->hat_diagonals, (size_t)a->n * a->m * 8);
        stage_prefault(o->mu, (size_t)a->n * a->m * 8);
    }
    return host_sharded((size_t)row_lo, (size_t)row_cnt, [&](size_t lo, size_t cnt, hipStream_t st) {
        return fit_beta_host_range(a, o, lo, cnt, st, cells, ncell);
    });
}

int dsq_fit_disp(const DsqFitDispArgs *a, const DsqFitDispOut *o) { return dsq_fit_disp_rows(a, o, 0, a ? a->n : 0); }

int dsq_fit_disp_rows(const DsqFitDispArgs *a, const DsqFitDispOut *o, int64_t row_lo, int64_t row_cnt) {
    std::lock_guard<std::mutex> lk(g_mu);
    WsScope ws(nullptr);
    if (!a || !o) return fail(DSQ_ERR_ARG, "NULL args/out");
    if (a->layout != DSQ_LAYOUT_R) return fail(DSQ_ERR_ARG, "host entry points take R layout only");
    if (a->n < 0 || a->m < 1 || a->p < 1) return fail(DSQ_ERR_ARG, "bad dimensions");
    if (!a->y || !a->x || !a->mu_hat || !a->log_alpha || !a->log_alpha_prior_mean)
        return fail(DSQ_ERR_ARG, "NULL input array");
    if (a->useWeights && !a->weights) return fail(DSQ_ERR_ARG, "useWeights set but weights is NULL");
    if (!o->log_alpha || !o->iter || !o->iter_accept || !o->last_change || !o->initial_lp || !o->initial_dlp ||
        !o->last_lp || !o->last_dlp || !o->last_d2lp)
        return fail(DSQ_ERR_ARG, "NULL output array");
    if (int rc = check_device()) return rc;
    if (row_lo < 0 || row_cnt < 0 || row_lo + row_cnt > a->n) return fail(DSQ_ERR_ARG, "row range outside [0, n)");
    if (row_cnt == 0) return DSQ_OK;
    std::vector<int32_t> labels;
    const int32_t *cells; int ncell;
    host_cells(a->x, a->m, a->p, a->cell_of, a->ncell, &labels, &cells, &ncell);
    return host_sharded((size_t)row_lo, (size_t)row_cnt, [&](size_t lo, size_t cnt, hipStream_t st) {
        return fit_disp_host_range(a, o, lo, cnt, st, cells, ncell);
    });
}

int dsq_fit_disp_grid(const DsqFitDispGridArgs *a, const DsqFitDispGridOut *o) { return dsq_fit_disp_grid_rows(a, o, 0, a ? a->n : 0); }

int dsq_fit_disp_grid_rows(const DsqFitDispGridArgs *a, const DsqFitDispGridOut *o, int64_t row_lo, int64_t row_cnt) {
    std::lock_guard<std::mutex> lk(g_mu);
    WsScope ws(nullptr);
    if (!a || !o) return fail(DSQ_ERR_ARG, "NULL args/out");
    if (a->layout != DSQ_LAYOUT_R) return fail(DSQ_ERR_ARG, "host entry points take R layout only");
    if (a->n < 0 || a->m < 1 || a->p < 1 || a->ngrid < 2) return fail(DSQ_ERR_ARG, "bad dimensions");
    if (!a->y || !a->x || !a->mu_hat || !a->disp_grid || !a->log_alpha_prior_mean || !o->log_alpha)
        return fail(DSQ_ERR_ARG, "NULL array");
    if (a->useWeights && !a->weights) return fail(DSQ_ERR_ARG, "useWeights set but weights is NULL");
    if (int rc = check_device()) return rc;
    if (row_lo < 0 || row_cnt < 0 || row_lo + row_cnt > a->n) return fail(DSQ_ERR_ARG, "row range outside [0, n)");
    if (row_cnt == 0) return DSQ_OK;
    std::vector<int32_t> labels;
    const int32_t *cells; int ncell;
    host_cells(a->x, a->m, a->p, a->cell_of, a->ncell, &labels, &cells, &ncell);
    return host_sharded((size_t)row_lo, (size_t)row_cnt, [&](size_t lo, size_t cnt, hipStream_t st) {
        return fit_disp_grid_host_range(a, o, lo, cnt, st, cells, ncell);
    });
}

int dsq_parametric_dispersion_fit_dev(const double *means, const double *disps, int64_t n, double *coefs,
                                      int32_t *status, void *stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    WsScope ws((hipStream_t)stream);
    if (!means || !disps || !coefs || !status || n < 1) return fail(DSQ_ERR_ARG, "bad arguments");
    if (int rc = check_device()) return rc;
    prof_begin((hipStream_t)stream);
    void *tws;
    if (int rc = ws_get(WS_TREND, trend_fit_workspace_bytes(), &tws)) return rc;
    DSQ_HIP(launch_trend_fit(means, disps, (long)n, coefs, status, tws, (hipStream_t)stream));
    prof_end((hipStream_t)stream);
    return DSQ_OK;
}

int dsq_parametric_dispersion_fit(const double *means, const double *disps, int64_t n, double *coefs, int32_t *status) {
    std::lock_guard<std::mutex> lk(g_mu);
    WsScope ws(nullptr);
    if (!means || !disps || !coefs || !status || n < 1) return fail(DSQ_ERR_ARG, "bad arguments");
    if (int rc = check_device()) return rc;
    hipStream_t st = nullptr;
    void *v;
    int rc;
    if ((rc = ws_get(WS_H_VEC, (2 * (size_t)n + 4) * 8, &v))) return rc;
    double *d = (double *)v;
    DSQ_HIP(hipMemcpyAsync(d, means, n * 8, hipMemcpyHostToDevice, st));
    DSQ_HIP(hipMemcpyAsync(d + n, disps, n * 8, hipMemcpyHostToDevice, st));
    void *tws;
    if ((rc = ws_get(WS_TREND, trend_fit_workspace_bytes(), &tws))) return rc;
    DSQ_HIP(launch_trend_fit(d, d + n, (long)n, d + 2 * n, (int32_t *)(d + 2 * n + 2), tws, st));
    DSQ_HIP(hipMemcpyAsync(coefs, d + 2 * n, 16, hipMemcpyDeviceToHost, st));
    DSQ_HIP(hipMemcpyAsync(status, d + 2 * n + 2, 4, hipMemcpyDeviceToHost, st));
    DSQ_HIP(hipStreamSynchronize(st));
    return DSQ_OK;
}

int dsq_prefit_moments_dev(const DsqPrefitArgs *args, const DsqPrefitOut *out, void *stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    WsScope ws((hipStream_t)stream);
    return prefit_dev_locked(args, out, (hipStream_t)stream);
}
int dsq_linear_mu_dev(const DsqPrefitArgs *args, double mu_floor, double *mu, void *stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    WsScope ws((hipStream_t)stream);
    return linear_mu_dev_locked(args, mu_floor, mu, (hipStream_t)stream);
}
int dsq_nbinom_loglike_dev(const DsqLogLikeArgs *args, double *loglike, void *stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    WsScope ws((hipStream_t)stream);
    return loglike_dev_locked(args, loglike, (hipStream_t)stream);
}

int dsq_prefit_moments(const DsqPrefitArgs *a, const DsqPrefitOut *o) {
    std::lock_guard<std::mutex> lk(g_mu);
    WsScope ws(nullptr);
    if (!a || !o) return fail(DSQ_ERR_ARG, "NULL args/out");
    if (a->layout != DSQ_LAYOUT_R) return fail(DSQ_ERR_ARG, "host entry points take R layout only");
    if (a->n < 0 || a->m < 2 || a->p < 1) return fail(DSQ_ERR_ARG, "bad dimensions");
    if (!a->y || !a->nf || !a->q || !a->a || !a->r) return fail(DSQ_ERR_ARG, "NULL input array");
    if (a->useWeights && !a->weights) return fail(DSQ_ERR_ARG, "useWeights set but weights is NULL");
    if (!o->baseMean || !o->baseVar || !o->allZero || !o->roughDisp || !o->beta_init) return fail(DSQ_ERR_ARG, "NULL output array");
    if (int rc = check_device()) return rc;
    if (a->n == 0) return DSQ_OK;
    hipStream_t st = nullptr;
    const size_t n = a->n, m = a->m, p = a->p;
    DsqPrefitArgs d = *a;
    DsqPrefitOut od = *o;
    void *v;
    int rc;
    if ((rc = up(WS_H_Y, a->y, n * m * (a->y_type == DSQ_Y_INT32 ? 4 : 8), st, &v))) return rc; d.y = v;
    if ((rc = up(WS_H_NF, a->nf, (a->nf_is_vector ? m : n * m) * 8, st, &v))) return rc; d.nf = (double *)v;
    if (a->useWeights) { if ((rc = up(WS_H_W, a->weights, n * m * 8, st, &v))) return rc; d.weights = (double *)v; }
    else d.weights = nullptr;
    size_t tot = 2 * m * p + p * p;
    if ((rc = ws_get(WS_H_VEC, tot * 8, &v))) return rc;
    double *vec = (double *)v;
    DSQ_HIP(hipMemcpyAsync(vec, a->q, m * p * 8, hipMemcpyHostToDevice, st));
    DSQ_HIP(hipMemcpyAsync(vec + m * p, a->a, m * p * 8, hipMemcpyHostToDevice, st));
    DSQ_HIP(hipMemcpyAsync(vec + 2 * m * p, a->r, p * p * 8, hipMemcpyHostToDevice, st));
    d.q = vec; d.a = vec + m * p; d.r = vec + 2 * m * p;
    if ((rc = ws_get(WS_H_OUTVEC, (4 * n + n * p) * 8, &v))) return rc;
    double *ov = (double *)v;
    od.baseMean = ov; od.baseVar = ov + n; od.roughDisp = ov + 2 * n; od.allZero = (int32_t *)(ov + 3 * n);
    od.beta_init = ov + 4 * n;
    rc = prefit_dev_locked(&d, &od, st);
    if (rc) return rc;
    DSQ_HIP(hipMemcpyAsync(o->baseMean, od.baseMean, n * 8, hipMemcpyDeviceToHost, st));
    DSQ_HIP(hipMemcpyAsync(o->baseVar, od.baseVar, n * 8, hipMemcpyDeviceToHost, st));
    DSQ_HIP(hipMemcpyAsync(o->roughDisp, od.roughDisp, n * 8, hipMemcpyDeviceToHost, st));
    DSQ_HIP(hipMemcpyAsync(o->allZero, od.allZero, n * 4, hipMemcpyDeviceToHost, st));
    DSQ_HIP(hipMemcpyAsync(o->beta_init, od.beta_init, n * p * 8, hipMemcpyDeviceToHost, st));
    DSQ_HIP(hipStreamSynchronize(st));
    return DSQ_OK;
}

int dsq_linear_mu(const DsqPrefitArgs *a, double mu_floor, double *mu) {
    std::lock_guard<std::mutex> lk(g_mu);
    WsScope ws(nullptr);
    if (!a || !mu) return fail(DSQ_ERR_ARG, "NULL args/out");
    if (a->layout != DSQ_LAYOUT_R) return fail(DSQ_ERR_ARG, "host entry points take R layout only");
    if (a->n < 0 || a->m < 1 || a->p < 1) return fail(DSQ_ERR_ARG, "bad dimensions");
    if (!a->y || !a->nf || !a->q || !a->a) return fail(DSQ_ERR_ARG, "NULL input array");
    if (int rc = check_device()) return rc;
    if (a->n == 0) return DSQ_OK;
    hipStream_t st = nullptr;
    const size_t n = a->n, m = a->m, p = a->p;
    DsqPrefitArgs d = *a;
    void *v;
    int rc;
    stage_prefault(mu, n * m * 8);
    if ((rc = up(WS_H_Y, a->y, n * m * (a->y_type == DSQ_Y_INT32 ? 4 : 8), st, &v))) return rc; d.y = v;
    if ((rc = up(WS_H_NF, a->nf, (a->nf_is_vector ? m : n * m) * 8, st, &v))) return rc; d.nf = (double *)v;
    if ((rc = ws_get(WS_H_VEC, 2 * m * p * 8, &v))) return rc;
    double *vec = (double *)v;
    DSQ_HIP(hipMemcpyAsync(vec, a->q, m * p * 8, hipMemcpyHostToDevice, st));
    DSQ_HIP(hipMemcpyAsync(vec + m * p, a->a, m * p * 8, hipMemcpyHostToDevice, st));
    d.q = vec; d.a = vec + m * p;
    if ((rc = ws_get(WS_H_OUTMAT, n * m * 8, &v))) return rc;
    rc = linear_mu_dev_locked(&d, mu_floor, (double *)v, st);
    if (rc) return rc;
    return down(mu, v, n * m * 8, st);
}

int dsq_nbinom_loglike(const DsqLogLikeArgs *a, double *loglike) {
    std::lock_guard<std::mutex> lk(g_mu);
    WsScope ws(nullptr);
    if (!a || !loglike) return fail(DSQ_ERR_ARG, "NULL args/out");
    if (a->layout != DSQ_LAYOUT_R) return fail(DSQ_ERR_ARG, "host entry points take R layout only");
    if (a->n < 0 || a->m < 1) return fail(DSQ_ERR_ARG, "bad dimensions");
    if (!a->y || !a->mu || !a->disp) return fail(DSQ_ERR_ARG, "NULL input array");
    if (a->useWeights && !a->weights) return fail(DSQ_ERR_ARG, "useWeights set but weights is NULL");
    if (int rc = check_device()) return rc;
    if (a->n == 0) return DSQ_OK;
    hipStream_t st = nullptr;
    const size_t n = a->n, m = a->m;
    DsqLogLikeArgs d = *a;
    void *v;
    int rc;
    if ((rc = up(WS_H_Y, a->y, n * m * (a->y_type == DSQ_Y_INT32 ? 4 : 8), st, &v))) return rc; d.y = v;
    if ((rc = up(WS_H_MU, a->mu, n * m * 8, st, &v))) return rc; d.mu = (double *)v;
    if (a->useWeights) { if ((rc = up(WS_H_W, a->weights, n * m * 8, st, &v))) return rc; d.weights = (double *)v; }
    else d.weights = nullptr;
    if ((rc = up(WS_H_VEC, a->disp, n * 8, st, &v))) return rc; d.disp = (double *)v;
    if ((rc = ws_get(WS_H_OUTVEC, n * 8, &v))) return rc;
    rc = loglike_dev_locked(&d, (double *)v, st);
    if (rc) return rc;
    DSQ_HIP(hipMemcpyAsync(loglike, v, n * 8, hipMemcpyDeviceToHost, st));
    DSQ_HIP(hipStreamSynchronize(st));
    return DSQ_OK;
}

int dsq_intercept_fit_dev(const DsqInterceptArgs *args, const DsqInterceptOut *out, void *stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    WsScope ws((hipStream_t)stream);
    return intercept_dev_locked(args, out, (hipStream_t)stream);
}

int dsq_intercept_fit(const DsqInterceptArgs *a, const DsqInterceptOut *o) {
    std::lock_guard<std::mutex> lk(g_mu);
    WsScope ws(nullptr);
    if (!a || !o) return fail(DSQ_ERR_ARG, "NULL args/out");
    if (a->layout != DSQ_LAYOUT_R) return fail(DSQ_ERR_ARG, "host entry points take R layout only");
    if (a->n < 0 || a->m < 1) return fail(DSQ_ERR_ARG, "bad dimensions");
    if (!a->y || !a->nf || !a->alpha) return fail(DSQ_ERR_ARG, "NULL input array");
    if (a->useWeights && !a->weights) return fail(DSQ_ERR_ARG, "useWeights set but weights is NULL");
    if (!o->beta_log2 || !o->betaSE) return fail(DSQ_ERR_ARG, "NULL output array");
    if (int rc = check_device()) return rc;
    if (a->n == 0) return DSQ_OK;
    hipStream_t st = nullptr;
    const size_t n = a->n, m = a->m;
    DsqInterceptArgs d = *a;
    DsqInterceptOut od = *o;
    void *v;
    int rc;
    stage_prefault(o->mu, n * m * 8);
    stage_prefault(o->hat, n * m * 8);
    if ((rc = up(WS_H_Y, a->y, n * m * (a->y_type == DSQ_Y_INT32 ? 4 : 8), st, &v))) return rc; d.y = v;
    if ((rc = up(WS_H_NF, a->nf, (a->nf_is_vector ? m : n * m) * 8, st, &v))) return rc; d.nf = (double *)v;
    if (a->useWeights) { if ((rc = up(WS_H_W, a->weights, n * m * 8, st, &v))) return rc; d.weights = (double *)v; }
    else d.weights = nullptr;
    if ((rc = up(WS_H_VEC, a->alpha, n * 8, st, &v))) return rc; d.alpha = (double *)v;
    if ((rc = ws_get(WS_H_OUTVEC, 2 * n * 8, &v))) return rc;
    od.beta_log2 = (double *)v; od.betaSE = (double *)v + n;
    od.mu = od.hat = nullptr;
    if (o->mu) { if ((rc = ws_get(WS_H_OUTMAT, n * m * 8, &v))) return rc; od.mu = (double *)v; }
    if (o->hat) { if ((rc = ws_get(WS_H_OUTMAT2, n * m * 8, &v))) return rc; od.hat = (double *)v; }
    rc = intercept_dev_locked(&d, &od, st);
    if (rc) return rc;
    DSQ_HIP(hipMemcpyAsync(o->beta_log2, od.beta_log2, n * 8, hipMemcpyDeviceToHost, st));
    DSQ_HIP(hipMemcpyAsync(o->betaSE, od.betaSE, n * 8, hipMemcpyDeviceToHost, st));
    if (o->mu && (rc = down(o->mu, od.mu, n * m * 8, st))) return rc;
    if (o->hat && (rc = down(o->hat, od.hat, n * m * 8, st))) return rc;
    DSQ_HIP(hipStreamSynchronize(st));
    return DSQ_OK;
}

int dsq_optim_rows(const DsqOptimArgs *a, const DsqOptimOut *o) {
    std::lock_guard<std::mutex> lk(g_mu);
    WsScope ws(nullptr);
    if (!a || !o) return fail(DSQ_ERR_ARG, "NULL args/out");
    if (a->layout != DSQ_LAYOUT_R) return fail(DSQ_ERR_ARG, "host entry points take R layout only");
    if (a->n < 0 || a->m < 1 || a->p < 1) return fail(DSQ_ERR_ARG, "bad dimensions");
    if (a->p > DSQ_P_WIDE) return fail(DSQ_ERR_UNSUPPORTED, "dsq_optim_rows: p=%d > %d design columns", a->p, DSQ_P_WIDE);
    if (!a->y || !a->x || !a->nf || !a->alpha_hat || !a->lambda || !a->beta_start) return fail(DSQ_ERR_ARG, "NULL input array");
    if (a->useWeights && !a->weights) return fail(DSQ_ERR_ARG, "useWeights set but weights is NULL");
    if (!o->beta || !o->betaSE || !o->conv || !o->mu || !o->logLike) return fail(DSQ_ERR_ARG, "NULL output array");
    if (int rc = check_device()) return rc;
    if (a->n == 0) return DSQ_OK;
    hipStream_t st = nullptr;
    // wide designs (see above): the kernel runs at the padded width pk -- zero design columns, ridge 1, start value 0
    const size_t n = a->n, m = a->m, p = a->p, pk = is_wide(a->p) ? wide_width(a->p) : a->p;
    void *v;
    int rc;
    OptimKernelParams kp;
    memset(&kp, 0, sizeof kp);
    kp.n = a->n; kp.m = a->m; kp.p = (int)pk; kp.minmu = a->minmu;
    if ((rc = up(WS_H_Y, a->y, n * m * (a->y_type == DSQ_Y_INT32 ? 4 : 8), st, &v))) return rc;
    bool ycheck = false;
    long ld = 0;
    rc = prep_counts(v, a->y_type, DSQ_LAYOUT_R, 0, a->n, a->m, st, &kp.y, &ld, &ycheck);
    if (rc) return rc;
    kp.ld = ld;
    if (a->nf_is_vector) { if ((rc = up(WS_H_NF, a->nf, m * 8, st, &v))) return rc; kp.nf = (double *)v; kp.nf_is_vector = 1; }
    else {
        if ((rc = up(WS_H_NF, a->nf, n * m * 8, st, &v))) return rc;
        if ((rc = prep_matrix((double *)v, DSQ_LAYOUT_R, 0, a->n, a->m, WS_NF, st, &kp.nf, ld))) return rc;
    }
    if (a->useWeights) {
        if ((rc = up(WS_H_W, a->weights, n * m * 8, st, &v))) return rc;
        if ((rc = prep_matrix((double *)v, DSQ_LAYOUT_R, 0, a->n, a->m, WS_W, st, &kp.weights, ld))) return rc;
        kp.useWeights = 1;
    }
    // x | alpha | lambda (natural-log scale) | beta_start
    const size_t off_x = 0, off_al = m * pk, off_lam = off_al + n, off_b = off_lam + pk, tot = off_b + n * pk;
    if ((rc = ws_get(WS_H_VEC, tot * 8, &v))) return rc;
    double *vec = (double *)v;
    static thread_local double lamnat[DSQ_P_WIDE];
    const double ln2 = 0.6931471805599453;
    for (size_t c = 0; c < pk; c++) lamnat[c] = c < p ? a->lambda[c] / (ln2 * ln2) : 1.0;
    if (pk != p) DSQ_HIP(hipMemsetAsync(vec, 0, tot * 8, st));
    DSQ_HIP(hipMemcpyAsync(vec + off_x, a->x, m * p * 8, hipMemcpyHostToDevice, st));
    DSQ_HIP(hipMemcpyAsync(vec + off_al, a->alpha_hat, n * 8, hipMemcpyHostToDevice, st));
    DSQ_HIP(hipMemcpyAsync(vec + off_lam, lamnat, pk * 8, hipMemcpyHostToDevice, st));
    DSQ_HIP(hipMemcpyAsync(vec + off_b, a->beta_start, n * p * 8, hipMemcpyHostToDevice, st));
    kp.x = vec + off_x; kp.alpha_hat = vec + off_al; kp.lamnat = vec + off_lam; kp.beta_start = vec + off_b;
    // outputs: beta | betaSE | loglike | conv ; mu (gene-major, then R layout)
    if ((rc = ws_get(WS_H_OUTVEC, (2 * n * pk + 2 * n) * 8, &v))) return rc;
    double *ov = (double *)v;
    kp.beta = ov; kp.betaSE = ov + n * pk; kp.loglike = ov + 2 * n * pk; kp.conv = (int32_t *)(ov + 2 * n * pk + n);
    void *mu_gm, *mu_r;
    if ((rc = ws_get(WS_MUOUT, n * (size_t)ld * 8, &mu_gm))) return rc;
    if ((rc = ws_get(WS_H_OUTMAT, n * m * 8, &mu_r))) return rc;
    kp.mu_out = (double *)mu_gm;
    bool ok = false;
    prof_begin(st);
    DSQ_HIP(dispatch_optim_rows((int)pk, kp, st, &ok));
    prof_end(st);
    if (!ok) return fail(DSQ_ERR_UNSUPPORTED, "no kernel for p=%d", a->p);
    DSQ_HIP(launch_transpose_gm_to_r_f64(kp.mu_out, (double *)mu_r, a->n, a->m, ld, st));
    DSQ_HIP(hipMemcpyAsync(o->beta, kp.beta, n * p * 8, hipMemcpyDeviceToHost, st));
    DSQ_HIP(hipMemcpyAsync(o->betaSE, kp.betaSE, n * p * 8, hipMemcpyDeviceToHost, st));
    DSQ_HIP(hipMemcpyAsync(o->logLike, kp.loglike, n * 8, hipMemcpyDeviceToHost, st));
    DSQ_HIP(hipMemcpyAsync(o->conv, kp.conv, n * 4, hipMemcpyDeviceToHost, st));
    DSQ_HIP(hipMemcpyAsync(o->mu, mu_r, n * m * 8, hipMemcpyDeviceToHost, st));
    DSQ_HIP(hipStreamSynchronize(st));
    return finish_ycheck(ycheck, st);
}

int dsq_cooks_distance_dev(const DsqCooksArgs *args, const DsqCooksOut *out, void *stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    WsScope ws((hipStream_t)stream);
    return cooks_dev_locked(args, out, (hipStream_t)stream);
}
int dsq_replace_outliers_dev(const DsqReplaceArgs *args, const DsqReplaceOut *out, void *stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    WsScope ws((hipStream_t)stream);
    return replace_dev_locked(args, out, (hipStream_t)stream);
}

int dsq_cooks_distance(const DsqCooksArgs *a, const DsqCooksOut *o) {
    std::lock_guard<std::mutex> lk(g_mu);
    WsScope ws(nullptr);
    if (!a || !o) return fail(DSQ_ERR_ARG, "NULL args/out");
    if (a->layout != DSQ_LAYOUT_R) return fail(DSQ_ERR_ARG, "host entry points take R layout only");
    if (a->n < 0 || a->m < 1) return fail(DSQ_ERR_ARG, "bad dimensions");
    if (!a->y || !a->nf || !a->mu || !a->H || !a->cell_of) return fail(DSQ_ERR_ARG, "NULL input array");
    if (!o->cooks || !o->maxCooks) return fail(DSQ_ERR_ARG, "NULL output array");
    if (int rc = check_device()) return rc;
    if (a->n == 0) return DSQ_OK;
    hipStream_t st = nullptr;
    const size_t n = a->n, m = a->m;
    DsqCooksArgs d = *a;
    DsqCooksOut od = *o;
    void *v;
    int rc;
    stage_prefault(o->cooks, n * m * 8);
    if ((rc = up(WS_H_Y, a->y, n * m * (a->y_type == DSQ_Y_INT32 ? 4 : 8), st, &v))) return rc; d.y = v;
    if ((rc = up(WS_H_NF, a->nf, (a->nf_is_vector ? m : n * m) * 8, st, &v))) return rc; d.nf = (double *)v;
    if ((rc = up(WS_H_MU, a->mu, n * m * 8, st, &v))) return rc; d.mu = (double *)v;
    if ((rc = up(WS_H_W, a->H, n * m * 8, st, &v))) return rc; d.H = (double *)v;
    if ((rc = ws_get(WS_H_OUTMAT, n * m * 8, &v))) return rc; od.cooks = (double *)v;
    if ((rc = ws_get(WS_H_OUTVEC, 2 * n * 8, &v))) return rc;
    od.maxCooks = (double *)v; od.robustDisp = (double *)v + n;
    rc = cooks_dev_locked(&d, &od, st);
    if (rc) return rc;
    if ((rc = down(o->cooks, od.cooks, n * m * 8, st))) return rc;
    DSQ_HIP(hipMemcpyAsync(o->maxCooks, od.maxCooks, n * 8, hipMemcpyDeviceToHost, st));
    if (o->robustDisp) DSQ_HIP(hipMemcpyAsync(o->robustDisp, od.robustDisp, n * 8, hipMemcpyDeviceToHost, st));
    DSQ_HIP(hipStreamSynchronize(st));
    return DSQ_OK;
}

int dsq_replace_outliers(const DsqReplaceArgs *a, const DsqReplaceOut *o) {
    std::lock_guard<std::mutex> lk(g_mu);
    WsScope ws(nullptr);
    if (!a || !o) return fail(DSQ_ERR_ARG, "NULL args/out");
    if (a->layout != DSQ_LAYOUT_R) return fail(DSQ_ERR_ARG, "host entry points take R layout only");
    if (a->n < 0 || a->m < 1) return fail(DSQ_ERR_ARG, "bad dimensions");
    if (!a->y || !a->nf || !a->cooks || !a->replaceable) return fail(DSQ_ERR_ARG, "NULL input array");
    if (!o->newCounts || !o->replace) return fail(DSQ_ERR_ARG, "NULL output array");
    if (int rc = check_device()) return rc;
    if (a->n == 0) return DSQ_OK;
    hipStream_t st = nullptr;
    const size_t n = a->n, m = a->m;
    DsqReplaceArgs d = *a;
    DsqReplaceOut od = *o;
    void *v;
    int rc;
    stage_prefault(o->newCounts, n * m * 4);
    if ((rc = up(WS_H_Y, a->y, n * m * (a->y_type == DSQ_Y_INT32 ? 4 : 8), st, &v))) return rc; d.y = v;
    if ((rc = up(WS_H_NF, a->nf, (a->nf_is_vector ? m : n * m) * 8, st, &v))) return rc; d.nf = (double *)v;
    if ((rc = up(WS_H_MU, a->cooks, n * m * 8, st, &v))) return rc; d.cooks = (double *)v;
    if ((rc = ws_get(WS_H_OUTMAT, n * m * 4, &v))) return rc; od.newCounts = (int32_t *)v;
    if ((rc = ws_get(WS_H_OUTVEC, n * 4, &v))) return rc; od.replace = (int32_t *)v;
    rc = replace_dev_locked(&d, &od, st);
    if (rc) return rc;
    if ((rc = down(o->newCounts, od.newCounts, n * m * 4, st))) return rc;
    DSQ_HIP(hipMemcpyAsync(o->replace, od.replace, n * 4, hipMemcpyDeviceToHost, st));
    DSQ_HIP(hipStreamSynchronize(st));
    return DSQ_OK;
}

int dsq_test_math(int op, const double *a, const double *b, const double *c, double *out, int64_t n) {
    std::lock_guard<std::mutex> lk(g_mu);
    WsScope ws(nullptr);
    if (!a || !out || n < 0 || ((op == 7 || op == 8) && !b) || (op == 8 && !c)) return fail(DSQ_ERR_ARG, "bad arguments");
    if (int rc = check_device()) return rc;
    if (n == 0) return DSQ_OK;
    hipStream_t st = nullptr;
    void *v;
    int rc;
    if ((rc = ws_get(WS_H_VEC, 4 * (size_t)n * 8, &v))) return rc;
    double *d = (double *)v;
    DSQ_HIP(hipMemcpyAsync(d, a, n * 8, hipMemcpyHostToDevice, st));
    if (b) DSQ_HIP(hipMemcpyAsync(d + n, b, n * 8, hipMemcpyHostToDevice, st));
    if (c) DSQ_HIP(hipMemcpyAsync(d + 2 * n, c, n * 8, hipMemcpyHostToDevice, st));
    DSQ_HIP(launch_test_math(op, d, d + n, d + 2 * n, d + 3 * n, n, st));
    DSQ_HIP(hipMemcpyAsync(out, d + 3 * n, n * 8, hipMemcpyDeviceToHost, st));
    DSQ_HIP(hipStreamSynchronize(st));
    return DSQ_OK;
}

}  // extern "C"
