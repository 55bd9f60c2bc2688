// stage.hip -- host <-> device copies of the host-pointer entry points through PINNED staging buffers.
//
// R's matrices (REAL() / INTEGER() of the .Call arguments, freshly allocated result vectors) are pageable memory:
// a plain hipMemcpy of them goes through the runtime's own single-threaded staging at a fraction of the link rate,
// and a gene range [lo, lo + cnt) of a column-major n x m matrix is a strided 2-D copy on top.  Here a copy is cut into
// chunks of a few MiB; a small pool of host threads packs a chunk into one of three pinned buffers (column segments
// of the range, contiguous in the buffer) while the DMA engine moves the previous chunk, so PCIe runs at link speed
// and the packed device image is the contiguous cnt x cols matrix the layout-conversion kernels expect.  Downloads
// run the same pipeline backwards (DMA of chunk k+1 and k+2 in flight while chunk k is scattered into R's matrix).
// No arithmetic here; DSQ_STAGE=0 switches back to plain hipMemcpy (tuning runs), DSQ_COPY_THREADS / DSQ_STAGE_MB
// size the pool and the chunks.
#include "../../include/deseq2_mi355x.h"
#include "dsq_internal.hpp"

#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23          /* Linux >= 5.14 */
#endif

namespace dsq {

namespace {

struct Piece { char *dst; const char *src; size_t len; };

static int env_int_(const char *name, int dflt) {
    const char *v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}

// ---- a few host threads that memcpy pieces; the calling thread works too ------------------------------------------
class CopyPool {
  public:
    static CopyPool &get() {
        static CopyPool *p = new CopyPool();      // never destroyed: the threads live as long as the process
        return *p;
    }
    void run(const std::vector<Piece> &pieces) {
        if (pieces.empty()) return;
        std::lock_guard<std::mutex> user(user_mu_);    // one copy at a time: the threads share the memory bandwidth anyway
        if (helpers_.empty() || pieces.size() == 1) {
            for (const Piece &q : pieces) memcpy(q.dst, q.src, q.len);
            return;
        }
        {
            std::lock_guard<std::mutex> lk(m_);
            p_ = pieces.data(); n_ = pieces.size(); next_.store(0);
            pending_ = (int)helpers_.size();
            gen_++;
        }
        cv_.notify_all();
        work();
        std::unique_lock<std::mutex> lk(m_);
        done_.wait(lk, [&] { return pending_ == 0; });
        p_ = nullptr; n_ = 0;
    }

  private:
    CopyPool() {
        unsigned hw = std::thread::hardware_concurrency();
        if (hw == 0) hw = 1;
        int t = env_int_("DSQ_COPY_THREADS", (int)(hw < 8 ? hw : 8));
        if (t < 1) t = 1;
        for (int k = 1; k < t; k++) {
            helpers_.emplace_back([this] { loop(); });
            helpers_.back().detach();
        }
    }
    void work() {
        for (;;) {
            const size_t i = next_.fetch_add(1);
            if (i >= n_) break;
            memcpy(p_[i].dst, p_[i].src, p_[i].len);
        }
    }
    void loop() {
        unsigned long long seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return gen_ != seen; });
                seen = gen_;
            }
            work();
            std::lock_guard<std::mutex> lk(m_);
            if (--pending_ == 0) done_.notify_all();
        }
    }
    std::vector<std::thread> helpers_;
    std::mutex user_mu_, m_;
    std::condition_variable cv_, done_;
    const Piece *p_ = nullptr;
    size_t n_ = 0;
    std::atomic<size_t> next_{0};
    int pending_ = 0;
    unsigned long long gen_ = 0;
};

// ---- first touch of the caller's result matrices, ahead of the copies ---------------------------------------------------
// The n x m results go into memory R has just allocated (Rf_allocMatrix -> mmap): pages nobody has touched.  A copy into
// them is bound by the page faults (zeroing + mapping, 4 KiB at a time), not by PCIe or memcpy: tools/d2h_probe.hip on an
// MI355X box measured 15-23 ms per 200 MiB into fresh pages against 4.4-5 ms (45 GB/s) into touched ones -- round 3's
// "mu / H / cooks come down at 10 GB/s".  MADV_POPULATE_WRITE faults a range in without changing its contents, a few
// threads reach 25-35 GB/s, and it needs neither the device nor the data: an entry point announces its large outputs
// when it starts (stage_prefault) and the faults are taken while the uploads and the kernels run.  stage_d2h announces
// its own destination if nobody has (the copy then runs behind the populate threads).  stage_prefault_finish() waits for
// the outstanding ranges: called before an entry point returns, so nothing touches the caller's memory afterwards.
class Prefaulter {
  public:
    static Prefaulter &get() {
        static Prefaulter *p = new Prefaulter();
        return *p;
    }
    bool on() const { return nthreads_ > 0; }
    void add(void *host, size_t bytes) {
        if (!on() || bytes < ((size_t)4 << 20)) return;
        const uintptr_t pg = (uintptr_t)page_;
        uintptr_t a = (uintptr_t)host & ~(pg - 1), b = ((uintptr_t)host + bytes + pg - 1) & ~(pg - 1);
        std::lock_guard<std::mutex> lk(m_);
        for (const auto &r : ranges_) if (a >= r.first && b <= r.second) return;       // already announced
        ranges_.push_back({a, b});
        const uintptr_t piece = (uintptr_t)4 << 20;
        for (uintptr_t q = a; q < b; q += piece) { tasks_.push_back({q, (b - q < piece) ? b - q : piece}); pending_++; }
        cv_.notify_all();
    }
    void finish() {
        if (!on()) return;
        std::unique_lock<std::mutex> lk(m_);
        done_.wait(lk, [&] { return pending_ == 0; });
        ranges_.clear();
    }

  private:
    Prefaulter() {
        page_ = sysconf(_SC_PAGESIZE);
        if (page_ <= 0) page_ = 4096;
        unsigned hw = std::thread::hardware_concurrency();
        int t = env_int_("DSQ_PREFAULT_THREADS", hw >= 16 ? 8 : (hw >= 4 ? (int)hw / 2 : 0));
        if (env_int_("DSQ_PREFAULT", 1) == 0 || t < 0) t = 0;
        // probe once: kernels before 5.14 answer EINVAL
        if (t > 0) {
            void *probe = mmap(nullptr, (size_t)page_, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
            if (probe == MAP_FAILED || madvise(probe, (size_t)page_, MADV_POPULATE_WRITE) != 0) t = 0;
            if (probe != MAP_FAILED) munmap(probe, (size_t)page_);
        }
        nthreads_ = t;
        for (int k = 0; k < t; k++) std::thread([this] { loop(); }).detach();
    }
    void loop() {
        for (;;) {
            std::pair<uintptr_t, uintptr_t> t;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return !tasks_.empty(); });
                t = tasks_.front();
                tasks_.pop_front();
            }
            (void)madvise((void *)t.first, (size_t)t.second, MADV_POPULATE_WRITE);     // (a failure leaves the faults to the copy)
            std::lock_guard<std::mutex> lk(m_);
            if (--pending_ == 0) done_.notify_all();
        }
    }
    long page_ = 4096;
    int nthreads_ = 0;
    std::mutex m_;
    std::condition_variable cv_, done_;
    std::deque<std::pair<uintptr_t, uintptr_t>> tasks_;
    std::vector<std::pair<uintptr_t, uintptr_t>> ranges_;
    size_t pending_ = 0;
};

// ---- three pinned chunks per (host thread, device) ------------------------------------------------------------------
struct Stager {
    static constexpr int NB = 3;
    char *buf[NB] = {nullptr, nullptr, nullptr};
    hipEvent_t ev[NB] = {nullptr, nullptr, nullptr};
    bool busy[NB] = {false, false, false};
    size_t cap = 0;
};
static thread_local Stager g_stagers[64];

static size_t chunk_bytes() {
    // (16 MiB: tools/d2h_probe.hip, three chunks in flight, 8 copy threads -> 47 GB/s into touched pages; 8 MiB: 31)
    static size_t c = (size_t)(env_int_("DSQ_STAGE_MB", 16) < 1 ? 1 : env_int_("DSQ_STAGE_MB", 16)) << 20;
    return c;
}
static bool staging_on() {
    static int on = env_int_("DSQ_STAGE", 1);
    return on != 0;
}

static int stager_get(Stager **out) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return capi_fail(DSQ_ERR_DEVICE, "hipGetDevice failed");
    Stager &s = g_stagers[dev];
    if (s.cap == 0) {
        const size_t c = chunk_bytes();
        for (int b = 0; b < Stager::NB; b++) {
            void *p = nullptr;
            if (hipHostMalloc(&p, c, hipHostMallocPortable) != hipSuccess || !p)
                return capi_fail(DSQ_ERR_NOMEM, "hipHostMalloc(%zu) for a staging chunk failed", c);
            s.buf[b] = (char *)p;
            if (hipEventCreateWithFlags(&s.ev[b], hipEventDisableTiming) != hipSuccess)
                return capi_fail(DSQ_ERR_DEVICE, "hipEventCreate failed");
        }
        s.cap = c;
    }
    *out = &s;
    return DSQ_OK;
}

// pieces of the packed byte range [a, b) of a cnt x cols block: column c of the block is the `seg` bytes at
// host + c * stride (+ the row offset already folded into host)
static void build_pieces(std::vector<Piece> *out, char *buf, const char *host, size_t seg, size_t stride, size_t a, size_t b,
                         bool to_buf) {
    const size_t kMax = 512 << 10;            // pieces small enough to balance the threads
    out->clear();
    size_t pos = a;
    while (pos < b) {
        const size_t col = pos / seg, off = pos % seg;
        size_t len = seg - off;
        if (len > b - pos) len = b - pos;
        size_t done = 0;
        while (done < len) {
            const size_t l = (len - done > kMax) ? kMax : len - done;
            char *pb = buf + (pos - a) + done;
            const char *ph = host + col * stride + off + done;
            if (to_buf) out->push_back({pb, ph, l});
            else out->push_back({const_cast<char *>(ph), pb, l});
            done += l;
        }
        pos += len;
    }
}

#define ST_HIP(expr)                                                                                     \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess)                                                                            \
            return capi_fail(e_ == hipErrorOutOfMemory ? DSQ_ERR_NOMEM : DSQ_ERR_DEVICE, "%s: %s", #expr, \
                             hipGetErrorString(e_));                                                     \
    } while (0)

}  // namespace

static std::atomic<bool> g_prefault_live{false};      // (entry points that never announce anything never start the threads)
void stage_prefault(void *host, size_t bytes) {
    if (!host || bytes < ((size_t)4 << 20)) return;
    g_prefault_live.store(true);
    Prefaulter::get().add(host, bytes);
}
void stage_prefault_finish() {
    if (g_prefault_live.load()) Prefaulter::get().finish();
}

// rows [lo, lo + cnt) of a column-major n_total x cols host matrix (elements of e bytes) -> the contiguous column-major
// cnt x cols device matrix `dev`.  Asynchronous on `st` for the device side; the host source has been read when
// the call returns.
int stage_h2d(void *dev, const void *host, size_t e, size_t n_total, size_t lo, size_t cnt, size_t cols, hipStream_t st) {
    if (!cnt || !cols) return DSQ_OK;
    const size_t seg = cnt * e, stride = n_total * e, total = seg * cols;
    const char *h = (const char *)host + lo * e;
    if (!staging_on() || total < (1u << 20)) {
        if (cnt == n_total) ST_HIP(hipMemcpyAsync(dev, host, total, hipMemcpyHostToDevice, st));
        else ST_HIP(hipMemcpy2DAsync(dev, seg, h, stride, seg, cols, hipMemcpyHostToDevice, st));
        return DSQ_OK;
    }
    Stager *s;
    if (int rc = stager_get(&s)) return rc;
    // a contiguous source is one long segment
    const size_t seg_eff = (cnt == n_total) ? total : seg;
    static thread_local std::vector<Piece> pieces;
    int k = 0;
    for (size_t off = 0; off < total; off += s->cap, k++) {
        const int b = k % Stager::NB;
        const size_t len = (total - off < s->cap) ? total - off : s->cap;
        if (s->busy[b]) { ST_HIP(hipEventSynchronize(s->ev[b])); s->busy[b] = false; }
        build_pieces(&pieces, s->buf[b], h, seg_eff, stride, off, off + len, true);
        CopyPool::get().run(pieces);
        ST_HIP(hipMemcpyAsync((char *)dev + off, s->buf[b], len, hipMemcpyHostToDevice, st));
        ST_HIP(hipEventRecord(s->ev[b], st));
        s->busy[b] = true;
    }
    return DSQ_OK;
}

// the reverse; SYNCHRONOUS: the host rows are complete when the call returns (everything enqueued on `st` before the
// call has finished by then as well)
int stage_d2h(void *host, const void *dev, size_t e, size_t n_total, size_t lo, size_t cnt, size_t cols, hipStream_t st) {
    if (!cnt || !cols) return DSQ_OK;
    const size_t seg = cnt * e, stride = n_total * e, total = seg * cols;
    char *h = (char *)host + lo * e;
    if (!staging_on() || total < (1u << 20)) {
        if (cnt == n_total) ST_HIP(hipMemcpyAsync(host, dev, total, hipMemcpyDeviceToHost, st));
        else ST_HIP(hipMemcpy2DAsync(h, stride, dev, seg, seg, cols, hipMemcpyDeviceToHost, st));
        ST_HIP(hipStreamSynchronize(st));
        return DSQ_OK;
    }
    Stager *s;
    if (int rc = stager_get(&s)) return rc;
    // (the whole matrix the rows belong to: a no-op when the entry point has announced it)
    stage_prefault((char *)host, (cnt == n_total) ? total : stride * cols);
    for (int b = 0; b < Stager::NB; b++)
        if (s->busy[b]) { ST_HIP(hipEventSynchronize(s->ev[b])); s->busy[b] = false; }
    const size_t seg_eff = (cnt == n_total) ? total : seg;
    const size_t nchunk = (total + s->cap - 1) / s->cap;
    auto issue = [&](size_t k) -> hipError_t {
        const int b = (int)(k % Stager::NB);
        const size_t off = k * s->cap, len = (total - off < s->cap) ? total - off : s->cap;
        hipError_t e1 = hipMemcpyAsync(s->buf[b], (const char *)dev + off, len, hipMemcpyDeviceToHost, st);
        if (e1 != hipSuccess) return e1;
        return hipEventRecord(s->ev[b], st);
    };
    static thread_local std::vector<Piece> pieces;
    const size_t ahead = Stager::NB - 1;
    for (size_t k = 0; k < ahead && k < nchunk; k++) ST_HIP(issue(k));
    for (size_t k = 0; k < nchunk; k++) {
        if (k + ahead < nchunk) ST_HIP(issue(k + ahead));     // its buffer was scattered in the previous round
        const int b = (int)(k % Stager::NB);
        const size_t off = k * s->cap, len = (total - off < s->cap) ? total - off : s->cap;
        ST_HIP(hipEventSynchronize(s->ev[b]));
        build_pieces(&pieces, s->buf[b], h, seg_eff, stride, off, off + len, false);
        CopyPool::get().run(pieces);
    }
    return DSQ_OK;
}

}  // namespace dsq
