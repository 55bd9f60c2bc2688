// tools/d2h_probe.hip -- what does it cost to bring an n x m f64 assay (mu / H / cooks: 200 MB each at C3) down into
// FRESH pageable host memory, as R hands it over (Rf_allocMatrix -> mmap, untouched pages)?  Measures the candidates for
// csrc/stage.hip on the box it runs on:
//   a  first-touch of the destination (page faults + zeroing) by T threads: plain stores / MADV_POPULATE_WRITE
//   b  hipHostRegister / hipHostUnregister of the destination (fresh and already touched)
//   c  DMA device -> registered destination
//   d  DMA device -> pinned staging chunk, memcpy by T threads into the (fresh / touched) destination, chunk sizes 8..64 MiB
//   hipcc -O2 -std=c++17 --offload-arch=gfx950 tools/d2h_probe.hip -o tools/d2h_probe -lpthread
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static char *fresh(size_t bytes) {
    void *p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    return p == MAP_FAILED ? nullptr : (char *)p;
}

template <class F>
static void par(int T, size_t bytes, F &&f) {
    std::vector<std::thread> th;
    const size_t per = ((bytes / T) + 4095) & ~(size_t)4095;
    for (int t = 0; t < T; t++) {
        const size_t a = (size_t)t * per, b = a + per > bytes ? bytes : a + per;
        if (a >= bytes) break;
        th.emplace_back([=] { f(a, b); });
    }
    for (auto &x : th) x.join();
}

int main() {
    const size_t B = (size_t)200 << 20;
    char *dev;
    CK(hipMalloc(&dev, B));
    CK(hipMemset(dev, 1, B));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    printf("hardware threads %u\n", std::thread::hardware_concurrency());
    for (int T : {1, 4, 8, 16, 32}) {
        char *d = fresh(B);
        double t0 = now();
        par(T, B, [&](size_t a, size_t b) { for (size_t i = a; i < b; i += 4096) d[i] = 0; });
        double t1 = now();
        munmap(d, B);
        d = fresh(B);
        double t2 = now();
        int rc = 0;
        par(T, B, [&](size_t a, size_t b) { if (madvise(d + a, b - a, MADV_POPULATE_WRITE)) rc = 1; });
        double t3 = now();
        munmap(d, B);
        printf("a first touch 200 MiB, %2d threads: stores %.1f ms (%.1f GB/s)   MADV_POPULATE_WRITE %.1f ms (%.1f GB/s)%s\n", T,
               (t1 - t0) * 1e3, B / (t1 - t0) / 1e9, (t3 - t2) * 1e3, B / (t3 - t2) / 1e9, rc ? " [madvise failed]" : "");
    }
    for (int touched = 0; touched < 2; touched++) {
        char *d = fresh(B);
        if (touched) par(16, B, [&](size_t a, size_t b) { for (size_t i = a; i < b; i += 4096) d[i] = 0; });
        double t0 = now();
        hipError_t e = hipHostRegister(d, B, hipHostRegisterDefault);
        double t1 = now();
        if (e != hipSuccess) { printf("b hipHostRegister failed: %s\n", hipGetErrorString(e)); munmap(d, B); continue; }
        CK(hipMemcpyAsync(d, dev, B, hipMemcpyDeviceToHost, st));
        CK(hipStreamSynchronize(st));
        double t2 = now();
        CK(hipMemcpyAsync(d, dev, B, hipMemcpyDeviceToHost, st));
        CK(hipStreamSynchronize(st));
        double t3 = now();
        CK(hipHostUnregister(d));
        double t4 = now();
        printf("b/c destination %s: hipHostRegister %.1f ms, DMA into it %.1f ms (%.1f GB/s), again %.1f ms (%.1f GB/s), hipHostUnregister %.1f ms\n",
               touched ? "touched" : "fresh  ", (t1 - t0) * 1e3, (t2 - t1) * 1e3, B / (t2 - t1) / 1e9, (t3 - t2) * 1e3, B / (t3 - t2) / 1e9,
               (t4 - t3) * 1e3);
        munmap(d, B);
    }
    {   // register in T slices concurrently
        for (int T : {4, 8}) {
            char *d = fresh(B);
            double t0 = now();
            int bad = 0;
            par(T, B, [&](size_t a, size_t b) { if (hipHostRegister(d + a, b - a, hipHostRegisterDefault) != hipSuccess) bad = 1; });
            double t1 = now();
            par(T, B, [&](size_t a, size_t b) { if (hipHostUnregister(d + a) != hipSuccess) bad = 1; });
            double t2 = now();
            printf("b fresh destination registered in %d concurrent slices: %.1f ms, unregistered %.1f ms%s\n", T, (t1 - t0) * 1e3, (t2 - t1) * 1e3,
                   bad ? " [failed]" : "");
            munmap(d, B);
        }
    }
    for (size_t cm : {8, 16, 32, 64}) {
        const size_t C = cm << 20;
        const int NB = 3;
        char *pin[NB];
        hipEvent_t ev[NB];
        for (int b = 0; b < NB; b++) { CK(hipHostMalloc((void **)&pin[b], C, hipHostMallocPortable)); CK(hipEventCreateWithFlags(&ev[b], hipEventDisableTiming)); }
        for (int T : {8, 16, 32}) {
            for (int touched = 0; touched < 2; touched++) {
                char *d = fresh(B);
                if (touched) par(16, B, [&](size_t a, size_t b) { for (size_t i = a; i < b; i += 4096) d[i] = 0; });
                const size_t nch = (B + C - 1) / C;
                double t0 = now();
                auto issue = [&](size_t k) {
                    const size_t off = k * C, len = B - off < C ? B - off : C;
                    (void)hipMemcpyAsync(pin[k % NB], dev + off, len, hipMemcpyDeviceToHost, st);
                    (void)hipEventRecord(ev[k % NB], st);
                };
                for (size_t k = 0; k < 2 && k < nch; k++) issue(k);
                for (size_t k = 0; k < nch; k++) {
                    if (k + 2 < nch) issue(k + 2);
                    (void)hipEventSynchronize(ev[k % NB]);
                    const size_t off = k * C, len = B - off < C ? B - off : C;
                    const char *src = pin[k % NB];
                    par(T, len, [&](size_t a, size_t b) { memcpy(d + off + a, src + a, b - a); });
                }
                double t1 = now();
                printf("d staged: chunk %2zu MiB, %2d copy threads (spawned per chunk), destination %s: %.1f ms (%.1f GB/s)\n", cm, T,
                       touched ? "touched" : "fresh  ", (t1 - t0) * 1e3, B / (t1 - t0) / 1e9);
                munmap(d, B);
            }
        }
        for (int b = 0; b < NB; b++) { CK(hipHostFree(pin[b])); CK(hipEventDestroy(ev[b])); }
    }
    return 0;
}
