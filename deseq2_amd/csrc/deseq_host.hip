// deseq_host.hip -- dsq_deseq: the whole DESeq() chain behind ONE host-pointer call (include/deseq2_mi355x.h).
//
// The R-side caller (the patch of R/core.R:388-426) hands over counts(dds), the model matrix, the size
// factors and three design-only quantities R has functions for (qr.Q / qr.R, qf, trigamma).  Here:
//   * the design facts the chain needs (design cells as nOrMoreInCell sees them R/core.R:2366-2371, replaceable samples
//     :2101, "one group per column" :735-742, mean(1 / sizeFactors) :2440-2444, the dispersion grid R/wrappers.R:70-72)
//     are derived on the host -- O(m p) work;
//   * the genes are cut into the contiguous ranges of R/parallel.R:10, one per visible device, each driven by a
//     persistent worker thread with its own stream (capi.hip: host_sharded);
//   * a range uploads its rows of the count matrix ONCE (pinned staging, stage.hip), converts them to the gene-major
//     layout on the device, enqueues the device-driven chain of pipeline.hip (every phase, no host decision), and
//     downloads the per-gene columns; n x m assays only on request;
//   * with more than one range, the two n-vectors of the dispersion trend (baseMean, dispGeneEst) are exchanged
//     through host memory between the gene-wise phase and the trend, exactly the exchange of DESeqParallel
//     (R/parallel.R:27-40): every range then fits the same trend over the same gathered vectors.
// No arithmetic on the data path here: everything per-gene is in the kernels.
#include "../../include/deseq2_mi355x.h"
#include "dsq_internal.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

namespace dsq {

int pipeline_run(const DsqDeseqArgs *a, const DsqDeseqOut *o, hipStream_t st);      // pipeline.hip
int capi_host_sharded(size_t n, const std::function<int(size_t, size_t, hipStream_t, int, int)> &f, int max_shards);   // capi.hip
int capi_host_shards(size_t n);
int beta_prior_var(const DsqBetaPriorArgs *a, double *out);                        // beta_prior.hip

// DSQ_TIMING=1: wall-clock marks of one dsq_deseq call on stderr (where does a PCIe-inclusive call spend its time)
static bool timing_on() { static int on = getenv("DSQ_TIMING") ? atoi(getenv("DSQ_TIMING")) : 0; return on != 0; }
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

#define HD_HIP(expr)                                                                                     \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess)                                                                            \
            return capi_fail(e_ == hipErrorOutOfMemory ? DSQ_ERR_NOMEM : DSQ_ERR_DEVICE, "%s: %s", #expr, \
                             hipGetErrorString(e_));                                                     \
    } while (0)

namespace {

enum {   // device workspace slots of a range (grow-only, cached per (device, stream): a second call allocates nothing)
    HD_YR = DSQ_WS_HOSTDESEQ, HD_Y, HD_VEC, HD_MAT, HD_IVEC, HD_MUHAT, HD_MU, HD_H, HD_COOKS, HD_REPC, HD_STAT, HD_WORK,
    HD_DESIGN, HD_TREND, HD_OUTR, HD_NFR, HD_NF, HD_WRAW, HD_WNORM, HD_WFLOOR, HD_MLE, HD_END
};
static_assert(HD_END <= DSQ_WS_COUNT, "workspace slots");

enum { V_BASEMEAN = 0, V_BASEVAR, V_DGE, V_DFIT, V_DMAP, V_DISP, V_BITER, V_LL, V_LLR, V_MAXCOOKS, V_COUNT };
enum { I_ALLZERO = 0, I_DGITER, I_DITER, I_DOUTLIER, I_BCONV, I_REPLACE, I_OPT1, I_OPT2, I_FORCEZERO, I_COUNT };

// what the chain needs to know about the design alone
struct Facts {
    std::vector<int32_t> cells, replaceable, cells_red;
    int ncell = 0, do_replace = 0, linearMu = 0, ncell_red = 0;
    std::vector<double> a, grid, lam, a_red;
    double xim = 0.0;
    int pcol = 0;                      // columns of beta / betaSE / stat / pvalue: p, or p_prior with the beta prior
    int prior_intercept = 0;
};

// A = X R^-1 by back substitution over the columns (linearModelMuNormalized, R/core.R:2455-2457)
static void x_rinv(const double *x, const double *r, int m, int p, std::vector<double> *out) {
    out->assign((size_t)m * p, 0.0);
    for (int j = 0; j < p; j++)
        for (int i = 0; i < m; i++) {
            double v = x[i + (size_t)m * j];
            for (int k = 0; k < j; k++) v -= (*out)[i + (size_t)m * k] * r[k + (size_t)p * j];
            (*out)[i + (size_t)m * j] = v / r[j + (size_t)p * j];
        }
}

// samples with identical model-matrix rows share a cell; cells numbered in the lexicographic order of their rows
static void design_cells(const double *x, int m, int p, std::vector<int32_t> *cell, int *ncell) {
    std::vector<int> rep;                         // first sample of every distinct row
    std::vector<int> of(m);
    auto same = [&](int i, int j) {
        for (int k = 0; k < p; k++) if (x[i + (size_t)m * k] != x[j + (size_t)m * k]) return false;
        return true;
    };
    for (int j = 0; j < m; j++) {
        int f = -1;
        for (size_t c = 0; c < rep.size() && f < 0; c++) if (same(j, rep[c])) f = (int)c;
        if (f < 0) { f = (int)rep.size(); rep.push_back(j); }
        of[j] = f;
    }
    std::vector<int> order(rep.size());
    for (size_t c = 0; c < rep.size(); c++) order[c] = (int)c;
    std::sort(order.begin(), order.end(), [&](int a, int b) {
        for (int k = 0; k < p; k++) {
            const double va = x[rep[a] + (size_t)m * k], vb = x[rep[b] + (size_t)m * k];
            if (va != vb) return va < vb;
        }
        return false;
    });
    std::vector<int> rank(rep.size());
    for (size_t r = 0; r < order.size(); r++) rank[order[r]] = (int)r;
    cell->resize(m);
    for (int j = 0; j < m; j++) (*cell)[j] = rank[of[j]];
    *ncell = (int)rep.size();
}

static void design_facts(const DsqDeseqHostArgs *a, Facts *f) {
    const int m = a->m, p = a->p;
    design_cells(a->x, m, p, &f->cells, &f->ncell);
    std::vector<int> size(f->ncell, 0);
    for (int j = 0; j < m; j++) size[f->cells[j]]++;
    f->replaceable.assign(m, 0);
    if (std::isfinite(a->minReplicatesForReplace))
        for (int j = 0; j < m; j++)
            if (!a->dispFit && (double)size[f->cells[j]] >= a->minReplicatesForReplace) { f->replaceable[j] = 1; f->do_replace = 1; }
    f->linearMu = (f->ncell == p && !a->weights) ? 1 : 0;                   // R/core.R:735-742
    if (a->xrinv) f->a.assign(a->xrinv, a->xrinv + (size_t)m * p);
    else x_rinv(a->x, a->r, m, p, &f->a);
    if (a->x_reduced) {
        design_cells(a->x_reduced, m, a->p_reduced, &f->cells_red, &f->ncell_red);
        x_rinv(a->x_reduced, a->r_reduced, m, a->p_reduced, &f->a_red);      // (feeds only outputs nobody reads)
    }
    // the 20-point grid of fitDispGridWrapper (R/wrappers.R:70-72): seq(log(1e-8), log(max(10, m)), length = 20)
    if (a->disp_grid && a->ngrid >= 2) f->grid.assign(a->disp_grid, a->disp_grid + a->ngrid);
    else {
        const double lo = std::log(1e-8), hi = std::log(m > 10 ? (double)m : 10.0);
        f->grid.resize(20);
        const double step = (hi - lo) / 19.0;
        for (int k = 0; k < 20; k++) f->grid[k] = (k == 19) ? hi : lo + (double)k * step;
    }
    const double ln2 = 0.6931471805599453;
    f->lam.assign(p, 1e-6 / (ln2 * ln2));                                    // R/fitNbinomGLMs.R:73,162
    f->pcol = (a->betaPrior && a->x_prior) ? a->p_prior : p;
    if (a->betaPrior) {
        const double *xp = a->x_prior ? a->x_prior : a->x;
        f->prior_intercept = 1;
        for (int j = 0; j < m; j++) if (xp[j] != 1.0) { f->prior_intercept = 0; break; }
    }
    if (a->normalizationFactors) {
        f->xim = 0.0;       // (the chain takes mean(1 / colMeans(nf)) over the rows that are not all zero itself)
    } else {
        double s = 0.0;                                                      // mean(1 / sizeFactors), in sample order
        for (int j = 0; j < m; j++) s += 1.0 / a->sizeFactors[j];
        f->xim = s / (double)m;
    }
}

// the gene ranges meet here between the gene-wise phase and the trend
struct Exchange {
    std::mutex mu;
    std::condition_variable cv;
    int arrived[3] = {0, 0, 0}, target = 1;
    // betaPrior: the MLE coefficients / baseMean / dispFit / all-zero flags of ALL ranges (stage 2), the prior variance
    // the first range through computes from them, its ridge on the natural-log scale
    std::vector<double> mle, dfit;
    std::vector<int32_t> allzero;
    std::mutex prior_mu;
    int prior_state = 0;               // 0 not yet computed, 1 done, < 0 failed (the DSQ_ERR_* code)
    bool trend_ok = true;
    std::vector<double> bpv, lam_prior;
    bool failed = false;
    std::vector<double> bm, dge;
    long n_refit = 0;                  // refitted rows over all ranges (stage 1)
    std::vector<int32_t> status;       // target x DSQ_ST_COUNT
    std::vector<double> scalars;       // target x DSQ_SC_COUNT
    bool wait(int stage, long add = 0) {
        std::unique_lock<std::mutex> lk(mu);
        n_refit += add;
        if (++arrived[stage] >= target) cv.notify_all();
        else cv.wait(lk, [&] { return arrived[stage] >= target || failed; });
        return !failed;
    }
    void fail() {
        std::lock_guard<std::mutex> lk(mu);
        failed = true;
        cv.notify_all();
    }
};

static int check_args(const DsqDeseqHostArgs *a, const DsqDeseqHostOut *o) {
    if (!a || !o) return capi_fail(DSQ_ERR_ARG, "NULL args/out");
    if (a->n < 1 || a->m < 2 || a->p < 1) return capi_fail(DSQ_ERR_ARG, "bad dimensions n=%d m=%d p=%d", a->n, a->m, a->p);
    if (!a->counts || !a->x || !a->q || !a->r) return capi_fail(DSQ_ERR_ARG, "NULL input array");
    if (!a->sizeFactors == !a->normalizationFactors) return capi_fail(DSQ_ERR_ARG, "exactly one of sizeFactors / normalizationFactors must be given");
    if (a->y_type != DSQ_Y_INT32 && a->y_type != DSQ_Y_FLOAT64) return capi_fail(DSQ_ERR_ARG, "unknown y_type %d", a->y_type);
    if (a->test != 0 && a->test != 1) return capi_fail(DSQ_ERR_ARG, "test must be 0 (Wald) or 1 (LRT)");
    if (a->fitType < DSQ_FIT_PARAMETRIC || a->fitType > DSQ_FIT_PARAMETRIC_OR_MEAN) return capi_fail(DSQ_ERR_ARG, "fitType must be one of DSQ_FIT_*");
    if (a->geneEstOnly && a->dispFit) return capi_fail(DSQ_ERR_ARG, "geneEstOnly and dispFit are the two calls of one analysis");
    if (a->x_reduced && (a->test != 1 || !a->q_reduced || !a->r_reduced || a->p_reduced < 1 || a->p_reduced >= a->p))
        return capi_fail(DSQ_ERR_ARG, "reduced model: LRT only, with qr.Q / qr.R of its model matrix and 1 <= p_reduced < p");
    if (a->betaPrior) {
        if (a->test != 0) return capi_fail(DSQ_ERR_ARG, "betaPrior: the Wald test only (R/core.R:1791: nbinomLRT has no beta prior)");
        if (!a->coef_factor) return capi_fail(DSQ_ERR_ARG, "betaPrior: coef_factor (what each model-matrix column is) must be given");
        if (a->x_prior && (a->p_prior < 1 || !a->prior_coef_factor)) return capi_fail(DSQ_ERR_ARG, "betaPrior on the expanded model matrix: p_prior / prior_coef_factor");
        if (a->x_prior && a->p_prior > DSQ_P_WIDE) return capi_fail(DSQ_ERR_UNSUPPORTED, "dsq_deseq: expanded model matrix with %d > %d columns", a->p_prior, DSQ_P_WIDE);
    }
    if (a->m <= a->p) return capi_fail(DSQ_ERR_ARG, "the number of samples and the number of model coefficients are equal");
    if (a->p > DSQ_P_WIDE) return capi_fail(DSQ_ERR_UNSUPPORTED, "dsq_deseq: p=%d > %d design columns", a->p, DSQ_P_WIDE);
    if (a->m - a->p <= 3 && !a->geneEstOnly && !(a->dispPriorVar > 0.0))
        return capi_fail(DSQ_ERR_UNSUPPORTED, "dsq_deseq: %d residual degrees of freedom: the prior variance of R/core.R:1155-1190 (seeded Monte-Carlo matching) is the caller's -- geneEstOnly, then dispPriorVar", a->m - a->p);
    if (!(a->cooksCutoff > 0.0) || !(a->expVarLogDisp > 0.0)) return capi_fail(DSQ_ERR_ARG, "cooksCutoff = qf(.99, p, m - p) and expVarLogDisp = trigamma((m - p) / 2) must be given");
    if (a->maxit < 1 || a->disp_maxit < 1 || !(a->betaTol > 0.0) || !(a->minmu > 0.0)) return capi_fail(DSQ_ERR_ARG, "betaTol / maxit / minmu / disp_maxit");
    if (!(a->minReplicatesForReplace >= 3.0)) return capi_fail(DSQ_ERR_ARG, "at least 3 replicates are necessary in order to indentify a sample as a count outlier");
    for (int j = 0; a->sizeFactors && j < a->m; j++)
        if (!(a->sizeFactors[j] > 0.0) || !std::isfinite(a->sizeFactors[j])) return capi_fail(DSQ_ERR_VALUE, "sizeFactors[%d] is not a positive finite number", j);
    if (!o->baseMean || !o->baseVar || !o->allZero || !o->dispGeneEst || !o->dispGeneIter || !o->dispFit || !o->dispMAP ||
        !o->dispersion || !o->dispIter || !o->dispOutlier || !o->beta || !o->betaSE || !o->betaConv || !o->betaIter ||
        !o->logLike || !o->maxCooks || !o->replace)
        return capi_fail(DSQ_ERR_ARG, "NULL output column");
    if (a->test == 0 && (!o->stat || !o->pvalue)) return capi_fail(DSQ_ERR_ARG, "the Wald test needs the stat / pvalue outputs");
    if (a->test == 1 && !o->logLikeReduced) return capi_fail(DSQ_ERR_ARG, "the LRT needs logLikeReduced");
    return DSQ_OK;
}

// one gene range: rows [lo, lo + cnt) of the analysis
static int deseq_range(const DsqDeseqHostArgs *a, DsqDeseqHostOut *o, const Facts &F, Exchange &X, size_t lo, size_t cnt,
                       hipStream_t st, int shard, int nshards) {
    const size_t n = a->n, m = a->m, p = a->p;
    const long ld = ((long)m + 7) & ~7L;
    int rc;
    void *v;
    const double t_begin = now_ms();
    // ---- counts: R layout rows -> device -> gene-major int32
    const size_t ye = a->y_type == DSQ_Y_INT32 ? 4 : 8;
    if ((rc = capi_ws_get(HD_YR, cnt * m * ye, &v))) return rc;
    void *y_r = v;
    if ((rc = stage_h2d(y_r, a->counts, ye, n, lo, cnt, m, st))) return rc;
    if ((rc = capi_ws_get(HD_Y, cnt * (size_t)ld * 4, &v))) return rc;
    int32_t *y = (int32_t *)v;
    // ---- status block first (the REALSXP conversion flags bad counts in it)
    if ((rc = capi_ws_get(HD_STAT, (DSQ_ST_COUNT + 4) * 4 + DSQ_SC_COUNT * 8 + 64, &v))) return rc;
    double *scalars = (double *)v;
    int32_t *status = (int32_t *)(scalars + DSQ_SC_COUNT), *bad = status + DSQ_ST_COUNT, *neg = bad + 1, *refit_total = bad + 2;
    HD_HIP(hipMemsetAsync(v, 0, (DSQ_ST_COUNT + 4) * 4 + DSQ_SC_COUNT * 8, st));
    if (a->y_type == DSQ_Y_INT32) HD_HIP(launch_transpose_r_to_gm_i32((const int32_t *)y_r, y, (int)cnt, (int)m, ld, st));
    else HD_HIP(launch_counts_f64_to_gm_i32((const double *)y_r, y, (int)cnt, (int)m, ld, bad, st));
    // ---- normalization-factor matrix / observation weights: rows up, gene-major on the device
    const double *nf_gm = nullptr, *w_raw = nullptr;
    double *w_norm = nullptr, *w_floor = nullptr;
    if (a->normalizationFactors) {
        void *r_, *g_;
        if ((rc = capi_ws_get(HD_NFR, cnt * m * 8, &r_))) return rc;
        if ((rc = stage_h2d(r_, a->normalizationFactors, 8, n, lo, cnt, m, st))) return rc;
        if ((rc = capi_ws_get(HD_NF, cnt * (size_t)ld * 8, &g_))) return rc;
        HD_HIP(launch_transpose_r_to_gm_f64((const double *)r_, (double *)g_, (int)cnt, (int)m, ld, st));
        nf_gm = (const double *)g_;
    }
    if (a->weights) {
        void *r_, *g_;
        if ((rc = capi_ws_get(HD_NFR, cnt * m * 8, &r_))) return rc;      // (the staging slot is free again: same stream)
        if ((rc = stage_h2d(r_, a->weights, 8, n, lo, cnt, m, st))) return rc;
        if ((rc = capi_ws_get(HD_WRAW, cnt * (size_t)ld * 8, &g_))) return rc;
        HD_HIP(launch_transpose_r_to_gm_f64((const double *)r_, (double *)g_, (int)cnt, (int)m, ld, st));
        w_raw = (const double *)g_;
        if ((rc = capi_ws_get(HD_WNORM, cnt * (size_t)ld * 8, &g_))) return rc;
        w_norm = (double *)g_;
        if ((rc = capi_ws_get(HD_WFLOOR, cnt * (size_t)ld * 8, &g_))) return rc;
        w_floor = (double *)g_;
    }
    // ---- design: x | q | a | r | grid | size factors, one small staging vector
    const size_t pr = a->x_reduced ? (size_t)a->p_reduced : 0;
    const size_t pcol = (size_t)F.pcol, pxp = (a->betaPrior && a->x_prior) ? (size_t)a->p_prior : 0;
    const size_t off_x = 0, off_q = off_x + m * p, off_a = off_q + m * p, off_r = off_a + m * p, off_g = off_r + p * p,
                 off_sf = off_g + F.grid.size(), off_xr = off_sf + m, off_qr = off_xr + m * pr, off_ar = off_qr + m * pr,
                 off_rr = off_ar + m * pr, off_xp = off_rr + pr * pr, dtot = off_xp + m * pxp;
    if ((rc = capi_ws_get(HD_DESIGN, dtot * 8, &v))) return rc;
    double *dd = (double *)v;
    {
        static thread_local std::vector<double> hb;
        hb.resize(dtot);
        memcpy(hb.data() + off_x, a->x, m * p * 8); memcpy(hb.data() + off_q, a->q, m * p * 8);
        memcpy(hb.data() + off_a, F.a.data(), m * p * 8); memcpy(hb.data() + off_r, a->r, p * p * 8);
        memcpy(hb.data() + off_g, F.grid.data(), F.grid.size() * 8);
        if (a->sizeFactors) memcpy(hb.data() + off_sf, a->sizeFactors, m * 8);
        if (pr) {
            memcpy(hb.data() + off_xr, a->x_reduced, m * pr * 8); memcpy(hb.data() + off_qr, a->q_reduced, m * pr * 8);
            memcpy(hb.data() + off_ar, F.a_red.data(), m * pr * 8); memcpy(hb.data() + off_rr, a->r_reduced, pr * pr * 8);
        }
        if (pxp) memcpy(hb.data() + off_xp, a->x_prior, m * pxp * 8);
        HD_HIP(hipMemcpyAsync(dd, hb.data(), dtot * 8, hipMemcpyHostToDevice, st));      // (pageable: staged at once)
    }
    // ---- outputs
    if ((rc = capi_ws_get(HD_VEC, V_COUNT * cnt * 8, &v))) return rc;
    double *vec = (double *)v;
    if ((rc = capi_ws_get(HD_MAT, 4 * pcol * cnt * 8, &v))) return rc;
    double *mat = (double *)v;
    double *mle = nullptr;
    if (a->betaPrior) { if ((rc = capi_ws_get(HD_MLE, p * cnt * 8, &v))) return rc; mle = (double *)v; }
    if ((rc = capi_ws_get(HD_IVEC, I_COUNT * cnt * 4, &v))) return rc;
    int32_t *ivec = (int32_t *)v;
    double *mats[4];
    const int slots[4] = {HD_MUHAT, HD_MU, HD_H, HD_COOKS};
    for (int k = 0; k < 4; k++) { if ((rc = capi_ws_get(slots[k], cnt * (size_t)ld * 8, &v))) return rc; mats[k] = (double *)v; }
    if ((rc = capi_ws_get(HD_REPC, cnt * (size_t)ld * 4, &v))) return rc;
    int32_t *repc = (int32_t *)v;
    const int nt = nshards > 1 ? (int)n : 0;
    const int64_t wsb = dsq_deseq_workspace_bytes((int32_t)cnt, (int32_t)m, (int32_t)(pcol > p ? pcol : p), nt);
    if ((rc = capi_ws_get(HD_WORK, (size_t)wsb, &v))) return rc;
    void *work = v;

    DsqDeseqArgs d;
    memset(&d, 0, sizeof d);
    d.n = (int32_t)cnt; d.m = (int32_t)m; d.p = (int32_t)p; d.ld = ld;
    d.y = y;
    if (nf_gm) { d.nf = nf_gm; d.nf_is_vector = 0; } else { d.nf = dd + off_sf; d.nf_is_vector = 1; }
    d.useWeights = a->weights ? 1 : 0;
    d.x = dd + off_x; d.q = dd + off_q; d.a = dd + off_a; d.r = dd + off_r;
    d.xim = F.xim; d.linearMu = F.linearMu;
    d.minDisp = 1e-8; d.kappa_0 = 1.0; d.dispTol = 1e-6; d.weightThreshold = 1e-2; d.outlierSD = 2.0;
    d.betaTol = a->betaTol; d.minmu = a->minmu; d.maxit = a->disp_maxit; d.useCR = a->useCR ? 1 : 0; d.useQR = a->useQR ? 1 : 0;
    d.betaMaxit = a->maxit;
    d.disp_grid = dd + off_g; d.ngrid = (int32_t)F.grid.size(); d.expVarLogDisp = a->expVarLogDisp;
    d.n_trend = nt; d.lambda = F.lam.data(); d.min_log_alpha = std::log(1e-8 / 10.0);
    d.workspace = work; d.workspace_bytes = wsb; d.test = a->test; d.fitType = a->fitType;
    d.dispPriorVar_in = a->dispPriorVar > 0.0 ? a->dispPriorVar : 0.0;
    d.cell_of = F.cells.data(); d.ncell = F.ncell; d.replaceable = F.replaceable.data();
    d.cooksCutoff = a->cooksCutoff; d.trim = 0.2; d.do_replace = F.do_replace;      // (dispFit given: Facts leaves it 0)
    if (pr) {
        d.x_red = dd + off_xr; d.q_red = dd + off_qr; d.a_red = dd + off_ar; d.r_red = dd + off_rr; d.p_red = (int32_t)pr;
        d.cell_of_red = F.cells_red.data(); d.ncell_red = F.ncell_red;
    }
    DsqDeseqOut od;
    memset(&od, 0, sizeof od);
    od.baseMean = vec + V_BASEMEAN * cnt; od.baseVar = vec + V_BASEVAR * cnt; od.dispGeneEst = vec + V_DGE * cnt;
    od.dispFit = vec + V_DFIT * cnt; od.dispMAP = vec + V_DMAP * cnt; od.dispersion = vec + V_DISP * cnt;
    od.betaIter = vec + V_BITER * cnt; od.logLike = vec + V_LL * cnt; od.logLikeReduced = vec + V_LLR * cnt;
    od.maxCooks = vec + V_MAXCOOKS * cnt;
    od.beta = mat; od.betaSE = mat + pcol * cnt;
    od.stat = a->test == 0 ? mat + 2 * pcol * cnt : nullptr; od.pvalue = a->test == 0 ? mat + 3 * pcol * cnt : nullptr;
    od.mle_beta = mle;
    if (a->betaPrior) {
        d.betaPrior = 1;
        d.x_prior = pxp ? dd + off_xp : dd + off_x; d.p_prior = (int32_t)pcol;
        d.prior_expanded = pxp ? 1 : 0;              // (the expanded matrix is rank deficient: start values of R/fitNbinomGLMs.R:146-155)
        d.prior_intercept = F.prior_intercept;
    }
    od.allZero = ivec + I_ALLZERO * cnt; od.dispGeneIter = ivec + I_DGITER * cnt; od.dispIter = ivec + I_DITER * cnt;
    od.dispOutlier = ivec + I_DOUTLIER * cnt; od.betaConv = ivec + I_BCONV * cnt; od.replace = ivec + I_REPLACE * cnt;
    od.optim_geneest = ivec + I_OPT1 * cnt; od.optim_test = ivec + I_OPT2 * cnt;
    if (a->weights) {
        // getAndCheckWeights (R/core.R:2697-2751): w / rowmax, its 1e-6 floor for the gene-wise search, the weightsFail rows
        HD_HIP(launch_weights_prep(w_raw, d.x, (int)cnt, (int)m, (int)p, ld, 1e-2, w_norm, w_floor, ivec + I_FORCEZERO * cnt, neg, st));
        d.weights_raw = w_raw; d.weights_norm = w_norm; d.weights_floor = w_floor; d.force_zero = ivec + I_FORCEZERO * cnt;
    }
    od.mu_hat = mats[0]; od.mu = mats[1]; od.H = mats[2]; od.cooks = mats[3]; od.replaceCounts = repc;
    od.status = status; od.scalars = scalars;

    // ---- the chain
    // betaPrior: between the MLE pass and the pass with the ridge every range hands its MLE coefficients (+ baseMean, dispFit,
    // the all-zero flags) to the host, the first range through computes the prior variance over ALL rows (R/core.R:1601-1689,
    // beta_prior.hip) -- DESeqParallel's exchange for this branch, R/parallel.R:34-48
    auto prior_exchange = [&]() -> int {
        HD_HIP(hipMemcpy2DAsync(X.mle.data() + lo, n * 8, mle, cnt * 8, cnt * 8, p, hipMemcpyDeviceToHost, st));
        HD_HIP(hipMemcpyAsync(X.bm.data() + lo, od.baseMean, cnt * 8, hipMemcpyDeviceToHost, st));
        HD_HIP(hipMemcpyAsync(X.dfit.data() + lo, od.dispFit, cnt * 8, hipMemcpyDeviceToHost, st));
        HD_HIP(hipMemcpyAsync(X.allzero.data() + lo, od.allZero, cnt * 4, hipMemcpyDeviceToHost, st));
        int32_t hst[DSQ_ST_COUNT];
        HD_HIP(hipMemcpyAsync(hst, status, sizeof hst, hipMemcpyDeviceToHost, st));
        HD_HIP(hipStreamSynchronize(st));
        if (shard == 0 && (hst[DSQ_ST_N_TREND] == 0 || hst[DSQ_ST_TREND_STATUS] != 0 || hst[DSQ_ST_N_ABOVE_MIN] == 0)) X.trend_ok = false;
        if (!X.wait(2)) return capi_fail(DSQ_ERR_DEVICE, "another gene range of this call failed");
        std::lock_guard<std::mutex> lk(X.prior_mu);
        if (X.prior_state == 0) {
            X.bpv.assign(pcol, 0.0);
            int r = DSQ_OK;
            if (!X.trend_ok) r = DSQ_ERR_FIT;                // (no dispersion trend: dsq_deseq reports it at the end)
            else if (a->betaPriorVar) for (size_t c = 0; c < pcol; c++) X.bpv[c] = a->betaPriorVar[c];
            else {
                DsqBetaPriorArgs b;
                memset(&b, 0, sizeof b);
                b.n = (int32_t)n; b.p = (int32_t)p; b.mle_beta = X.mle.data(); b.baseMean = X.bm.data(); b.dispFit = X.dfit.data();
                b.allZero = X.allzero.data(); b.coef_factor = a->coef_factor; b.expanded = pxp ? 1 : 0; b.p_prior = (int32_t)pcol;
                b.prior_coef_factor = a->prior_coef_factor; b.prior_coef_src = a->prior_coef_src; b.upperQuantile = 0.05;
                r = beta_prior_var(&b, X.bpv.data());
            }
            if (r == DSQ_OK)
                for (size_t c = 0; c < pcol; c++)
                    if (X.bpv[c] == 0.0) r = capi_fail(DSQ_ERR_FIT, "beta prior variances are equal to zero for some variables");
            const double ln2 = 0.6931471805599453;
            X.lam_prior.assign(pcol, 0.0);
            for (size_t c = 0; c < pcol; c++) X.lam_prior[c] = (1.0 / X.bpv[c]) / (ln2 * ln2);      // R/fitNbinomGLMs.R:311,162
            X.prior_state = (r == DSQ_OK) ? 1 : (r < 0 ? r : -r);
        }
        return X.prior_state == 1 ? DSQ_OK : -1;
    };
    bool prior_skipped = false;
    if (a->dispFit) {
        // the caller's trend: this range's values beside the gathered vectors (behind them: all n, for the residuals)
        if ((rc = capi_ws_get(HD_TREND, 3 * n * 8, &v))) return rc;
        double *tv = (double *)v;
        HD_HIP(hipMemcpyAsync(tv + 2 * n, a->dispFit, n * 8, hipMemcpyHostToDevice, st));
        d.dispFit_in = tv + 2 * n + lo;
        d.trend_fit_in = tv + 2 * n;
    }
    if (a->geneEstOnly) {
        // estimateDispersionsGeneEst alone (the caller fits its own trend next): every other column NA
        HD_HIP(hipMemsetAsync(vec, 0xFF, V_COUNT * cnt * 8, st));
        HD_HIP(hipMemsetAsync(mat, 0xFF, 4 * pcol * cnt * 8, st));
        HD_HIP(hipMemsetAsync(ivec, 0xFF, I_COUNT * cnt * 4, st));
        HD_HIP(hipMemsetAsync(scalars, 0xFF, DSQ_SC_COUNT * 8, st));
        HD_HIP(hipMemsetAsync(status, 0, DSQ_ST_COUNT * 4, st));
        d.phases = DSQ_PH_GENE_EST;
        if ((rc = pipeline_run(&d, &od, st))) return rc;
    } else if (nshards == 1 && a->betaPrior) {
        d.phases = DSQ_PH_GENE_EST | DSQ_PH_TREND | DSQ_PH_MAP_TEST;
        if ((rc = pipeline_run(&d, &od, st))) return rc;
        if (prior_exchange() == DSQ_OK) {
            d.lambda_prior = X.lam_prior.data();
            d.phases = DSQ_PH_PRIOR | DSQ_PH_OUTLIERS;
            if ((rc = pipeline_run(&d, &od, st))) return rc;
        } else prior_skipped = true;
    } else if (nshards == 1) {
        d.phases = DSQ_PH_GENE_EST | DSQ_PH_TREND | DSQ_PH_MAP_TEST | DSQ_PH_OUTLIERS;
        if ((rc = pipeline_run(&d, &od, st))) return rc;
    } else {
        d.phases = DSQ_PH_GENE_EST;
        if ((rc = pipeline_run(&d, &od, st))) return rc;
        // the trend's input vectors of all ranges, through host memory (R/parallel.R:27-28)
        HD_HIP(hipMemcpyAsync(X.bm.data() + lo, od.baseMean, cnt * 8, hipMemcpyDeviceToHost, st));
        HD_HIP(hipMemcpyAsync(X.dge.data() + lo, od.dispGeneEst, cnt * 8, hipMemcpyDeviceToHost, st));
        HD_HIP(hipStreamSynchronize(st));
        if (!X.wait(0)) return capi_fail(DSQ_ERR_DEVICE, "another gene range of this call failed");
        if ((rc = capi_ws_get(HD_TREND, 3 * n * 8, &v))) return rc;
        double *tv = (double *)v;
        HD_HIP(hipMemcpyAsync(tv, X.bm.data(), n * 8, hipMemcpyHostToDevice, st));
        HD_HIP(hipMemcpyAsync(tv + n, X.dge.data(), n * 8, hipMemcpyHostToDevice, st));
        d.trend_mean = tv; d.trend_disp = tv + n;
        d.defer_finish = 1;
        if (a->betaPrior) {
            d.phases = DSQ_PH_TREND | DSQ_PH_MAP_TEST;
            if ((rc = pipeline_run(&d, &od, st))) return rc;
            if (prior_exchange() == DSQ_OK) {
                d.lambda_prior = X.lam_prior.data();
                d.phases = DSQ_PH_PRIOR | DSQ_PH_OUTLIERS;
                if ((rc = pipeline_run(&d, &od, st))) return rc;
            } else prior_skipped = true;
        } else {
            d.phases = DSQ_PH_TREND | DSQ_PH_MAP_TEST | DSQ_PH_OUTLIERS;
            if ((rc = pipeline_run(&d, &od, st))) return rc;
        }
        if (F.do_replace && !prior_skipped) {
            // refitWithoutOutliers' closing steps ask whether ANY row of the whole object was refitted (R/core.R:2496):
            // the ranges add up their counts, then each finishes its own rows
            int32_t mine = 0;
            HD_HIP(hipMemcpyAsync(&mine, status + DSQ_ST_N_REFIT, 4, hipMemcpyDeviceToHost, st));
            HD_HIP(hipStreamSynchronize(st));
            if (!X.wait(1, mine)) return capi_fail(DSQ_ERR_DEVICE, "another gene range of this call failed");
            const int32_t total = (int32_t)(X.n_refit > 0x7fffffffL ? 0x7fffffffL : X.n_refit);
            static thread_local int32_t total_h;
            total_h = total;
            HD_HIP(hipMemcpyAsync(refit_total, &total_h, 4, hipMemcpyHostToDevice, st));  // (pageable: staged at once)
            d.n_refit_global = refit_total;
            d.phases = DSQ_PH_FINISH;
            if ((rc = pipeline_run(&d, &od, st))) return rc;
        }
    }

    const double t_enq = now_ms();
    // ---- per-gene columns down: three packed blocks, scattered into the caller's columns at this range's rows
    static thread_local std::vector<double> hv;
    static thread_local std::vector<int32_t> hi;
    hv.resize((V_COUNT + 4 * pcol) * cnt + DSQ_SC_COUNT + DSQ_ST_COUNT);
    hi.resize(I_COUNT * cnt);
    double *hvec = hv.data(), *hmat = hvec + V_COUNT * cnt, *hsc = hmat + 4 * pcol * cnt;
    int32_t *hst = (int32_t *)(hsc + DSQ_SC_COUNT);
    HD_HIP(hipMemcpyAsync(hsc, scalars, DSQ_SC_COUNT * 8 + (DSQ_ST_COUNT + 4) * 4, hipMemcpyDeviceToHost, st));
    // (the MLE coefficients as the chain leaves them: the refit has rewritten the rows it refitted, R/core.R:2533-2534)
    if (a->betaPrior && o->mle_beta)
        HD_HIP(hipMemcpy2DAsync(o->mle_beta + lo, n * 8, mle, cnt * 8, cnt * 8, p, hipMemcpyDeviceToHost, st));
    if ((rc = stage_d2h(hvec, vec, 1, V_COUNT * cnt * 8, 0, V_COUNT * cnt * 8, 1, st))) return rc;
    if ((rc = stage_d2h(hmat, mat, 1, 4 * pcol * cnt * 8, 0, 4 * pcol * cnt * 8, 1, st))) return rc;
    if ((rc = stage_d2h(hi.data(), ivec, 1, I_COUNT * cnt * 4, 0, I_COUNT * cnt * 4, 1, st))) return rc;
    HD_HIP(hipStreamSynchronize(st));
    if (hst[DSQ_ST_COUNT] != 0) return capi_fail(DSQ_ERR_VALUE, "count matrix holds negative, non-finite or non-integer values");
    if (hst[DSQ_ST_COUNT + 1] != 0) return capi_fail(DSQ_ERR_VALUE, "all(weights >= 0) is not TRUE");
    memcpy(X.status.data() + (size_t)shard * DSQ_ST_COUNT, hst, DSQ_ST_COUNT * 4);
    memcpy(X.scalars.data() + (size_t)shard * DSQ_SC_COUNT, hsc, DSQ_SC_COUNT * 8);
    double *const dcol[V_COUNT] = {o->baseMean, o->baseVar, o->dispGeneEst, o->dispFit, o->dispMAP, o->dispersion, o->betaIter,
                                   o->logLike, o->logLikeReduced, o->maxCooks};
    for (int k = 0; k < V_COUNT; k++)
        if (dcol[k]) memcpy(dcol[k] + lo, hvec + (size_t)k * cnt, cnt * 8);
    double *const mcol[4] = {o->beta, o->betaSE, o->stat, o->pvalue};
    for (int k = 0; k < 4; k++)
        if (mcol[k] && (k < 2 || a->test == 0))
            for (size_t c = 0; c < pcol; c++) memcpy(mcol[k] + c * n + lo, hmat + ((size_t)k * pcol + c) * cnt, cnt * 8);

    int32_t *const icol[6] = {o->allZero, o->dispGeneIter, o->dispIter, o->dispOutlier, o->betaConv, o->replace};
    for (int k = 0; k < 6; k++) memcpy(icol[k] + lo, hi.data() + (size_t)k * cnt, cnt * 4);
    if (o->weightsFail) {
        if (a->weights) memcpy(o->weightsFail + lo, hi.data() + (size_t)I_FORCEZERO * cnt, cnt * 4);
        else memset(o->weightsFail + lo, 0, cnt * 4);
    }
    // `replace` is NA on the rows that were all zero from the start (and everywhere without a replaceable sample)
    for (size_t i = 0; i < cnt; i++)
        if (!F.do_replace || (o->allZero[lo + i] && o->replace[lo + i] == 0)) o->replace[lo + i] = -1;

    const double t_cols = now_ms();
    double t_assay[3] = {0, 0, 0};
    // ---- assays on request: gene-major -> R layout on the device -> the caller's n x m matrix
    double *const want[3] = {o->mu, o->H, o->cooks};
    for (int k = 0; k < 3; k++) {
        if (!want[k] || a->geneEstOnly) continue;        // (geneEstOnly: no test has run, nothing to bring down)
        if ((rc = capi_ws_get(HD_OUTR, cnt * m * 8, &v))) return rc;
        HD_HIP(launch_transpose_gm_to_r_f64(mats[k + 1], (double *)v, (int)cnt, (int)m, ld, st));
        if ((rc = stage_d2h(want[k], v, 8, n, lo, cnt, m, st))) return rc;
        t_assay[k] = now_ms();
    }
    if (timing_on())
        fprintf(stderr, "[dsq_deseq range %d] upload+enqueue %.2f ms, chain done + columns down %.2f ms, assays %.2f / %.2f / %.2f ms\n", shard,
                t_enq - t_begin, t_cols - t_enq, t_assay[0] ? t_assay[0] - t_cols : 0.0, t_assay[1] ? t_assay[1] - t_assay[0] : 0.0,
                t_assay[2] ? t_assay[2] - t_assay[1] : 0.0);
    if (o->replaceCounts) {
        if ((rc = capi_ws_get(HD_OUTR, cnt * m * 8, &v))) return rc;
        HD_HIP(launch_transpose_gm_to_r_i32(repc, (int32_t *)v, (int)cnt, (int)m, ld, st));
        if ((rc = stage_d2h(o->replaceCounts, v, 4, n, lo, cnt, m, st))) return rc;
    }
    return DSQ_OK;
}

}  // namespace
}  // namespace dsq

using namespace dsq;

extern "C" int dsq_deseq(const DsqDeseqHostArgs *a, DsqDeseqHostOut *o) {
    std::lock_guard<std::mutex> lk(capi_mutex());
    capi_latch_stream(nullptr);
    int rc = check_args(a, o);
    if (rc) return rc;
    if ((rc = capi_check_device())) return rc;
    Facts F;
    design_facts(a, &F);
    PrefaultScope pf;               // the n x m assays land in fresh pages of the caller: fault them in while the chain runs
    for (double *w : {o->mu, o->H, o->cooks}) stage_prefault(w, (size_t)a->n * a->m * 8);
    stage_prefault(o->replaceCounts, (size_t)a->n * a->m * 4);
    Exchange X;
    if (a->betaPrior) { X.mle.resize((size_t)a->n * a->p); X.dfit.resize(a->n); X.allzero.resize(a->n); X.bm.resize(a->n); }
    // a normalization-factor matrix: momentsDispEstimate averages it over ALL non-zero rows of the object, a sum that
    // gene ranges could only reproduce in another order -- one range
    const int S = a->normalizationFactors ? 1 : capi_host_shards((size_t)a->n);
    X.target = S;
    if (S > 1) { X.bm.resize(a->n); X.dge.resize(a->n); }
    X.status.assign((size_t)S * DSQ_ST_COUNT, 0);
    X.scalars.assign((size_t)S * DSQ_SC_COUNT, 0.0);
    rc = capi_host_sharded((size_t)a->n, [&](size_t lo, size_t cnt, hipStream_t st, int shard, int nshards) {
        const int r = deseq_range(a, o, F, X, lo, cnt, st, shard, nshards);
        if (r) X.fail();             // (ranges waiting at the exchange give up instead of waiting for this one)
        return r;
    }, S);
    if (rc) return rc;
    if (a->betaPrior && X.prior_state < 0 && X.trend_ok) return -X.prior_state;      // (beta_prior_var's own failure; its message is set)
    if (a->betaPrior) for (int c = 0; c < F.pcol && c < DSQ_MAX_P; c++) o->betaPriorVar[c] = X.prior_state == 1 ? X.bpv[c] : NAN;
    // counters: per-range counts add up; the trend's (fitted by every range over the same gathered vectors) are range 0's
    memset(o->status, 0, sizeof o->status);
    for (int s = 0; s < S; s++)
        for (int k = 0; k < DSQ_ST_COUNT; k++)
            if (k != DSQ_ST_N_TREND && k != DSQ_ST_TREND_STATUS && k != DSQ_ST_N_ABOVE_MIN) o->status[k] += X.status[(size_t)s * DSQ_ST_COUNT + k];
    o->status[DSQ_ST_N_TREND] = X.status[DSQ_ST_N_TREND];
    o->status[DSQ_ST_TREND_STATUS] = X.status[DSQ_ST_TREND_STATUS];
    o->status[DSQ_ST_N_ABOVE_MIN] = X.status[DSQ_ST_N_ABOVE_MIN];
    for (int k = 0; k < DSQ_SC_COUNT; k++) o->dispersionFunction[k] = X.scalars[k];
    if (o->status[DSQ_ST_N_NONZERO] == 0) return capi_fail(DSQ_ERR_FIT, "all genes have zero counts in every sample");
    if (a->geneEstOnly) return DSQ_OK;
    if (o->status[DSQ_ST_N_TREND] == 0)
        return capi_fail(DSQ_ERR_FIT, "all gene-wise dispersion estimates are within 2 orders of magnitude from the minimum value");
    if (o->status[DSQ_ST_TREND_STATUS] != 0 || o->status[DSQ_ST_N_ABOVE_MIN] == 0)
        return capi_fail(DSQ_ERR_FIT, "the parametric dispersion trend did not fit (status %d): fitType = DSQ_FIT_PARAMETRIC_OR_MEAN / DSQ_FIT_MEAN, or 'local' through the call-by-call routines (R/core.R:885-893)", o->status[DSQ_ST_TREND_STATUS]);
    return DSQ_OK;
}
