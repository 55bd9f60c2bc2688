// tools/lu_probe.hip -- repro harness for the LU<P>::solve pivot-swap forms (csrc/dsq_wave.hpp).
//
// Round 2 saw the select-chain form of the right-hand-side swap give wrong results on the device inside the kernels of
// that time (fitDisp at P = 5, 6 with observation weights; the second-derivative kernel at P = 10: last_d2lp off by
// orders of magnitude) and fenced it in with DSQ_LU_SELECT_MAXP = 4 without a root cause.  This probe instantiates
// LU<P, FORM> for the three semantically equal forms (conditional swap, select chain, pairwise select) at
// P = 4, 5, 6, 10 in the usage patterns of those kernels and compares every result BIT FOR BIT with the same template
// compiled for the host:
//   usage A  inverse of a wave-uniform matrix (every lane the same values; what cr_algebra does)
//   usage B  the second-derivative algebra: det, B^-1, tr(B^-1 dB), tr((B^-1 dB)^2), tr(B^-1 d2B)
//   usage C  a different matrix per lane (divergent pivot rows)
// over matrix families chosen for the pivoting they trigger: Gram matrices of a factor design (hardly any row swap),
// the same with rows / columns masked out and 1 on the diagonal (the observation-weight subsetting of
// src/DESeq2.cpp:41-43: swaps on most steps), general random matrices (a swap on nearly every step), and matrices with
// exact ties in the pivot column.
//
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -mfma tools/lu_probe.hip -o tools/lu_probe
//   tools/lu_probe            -> one line per (P, form, usage, family): mismatching results / total; exit code 1 on any
//   tools/lu_probe host       -> no device: the host build of the three forms against each other (run it from a build
//                                with -Xarch_host -fsanitize=address,undefined to look for out-of-range piv[] / b[] accesses)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#include <random>
#include "../deseq2_amd/csrc/dsq_wave.hpp"

using namespace dsq;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

template <int P> struct Out { double inv[P][P]; double det, tr1, tr2, tr3; };

template <int P, int FORM>
__host__ __device__ inline void algebra(const double *m3, Out<P> *o) {
    LU<P, FORM> lu;
    double dB[P][P], d2B[P][P];
    for (int i = 0; i < P; i++)
        for (int j = 0; j < P; j++) {
            lu.a[i][j] = m3[i * P + j];
            dB[i][j] = m3[P * P + i * P + j];
            d2B[i][j] = m3[2 * P * P + i * P + j];
        }
    lu.factor();
    o->det = lu.det();
    double Bi[P][P], M[P][P];
    lu.inverse(Bi);
    for (int i = 0; i < P; i++)
        for (int j = 0; j < P; j++) o->inv[i][j] = Bi[i][j];
    o->tr1 = trace_sym<P>(Bi, dB);
    mat_mul<P>(Bi, dB, M);
    o->tr2 = trace_prod<P>(M, M);
    o->tr3 = trace_sym<P>(Bi, d2B);
}

// uniform = 1: every lane of a wave works on the wave's matrix (usage A / B); 0: one matrix per lane (usage C)
template <int P, int FORM>
__global__ void probe_kernel(const double *mats, Out<P> *out, int nmat, int uniform) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int idx = uniform ? tid / 64 : tid;
    if (idx >= nmat) return;
    Out<P> o;
    algebra<P, FORM>(mats + (size_t)idx * 3 * P * P, &o);
    // every lane must hold the bits lane 0 holds (the kernels read such values wave-uniformly); the cross-lane reads
    // happen with the whole wave active (a read from an inactive lane returns garbage -- the first version of this probe
    // did exactly that inside the branch below and "found" mismatches in every form)
    bool differs = false;
    if (uniform) {
        const double d0 = __shfl(o.det, 0, 64), t0 = __shfl(o.tr2, 0, 64), i0 = __shfl(o.inv[P - 1][0], 0, 64);
        differs = __double_as_longlong(d0) != __double_as_longlong(o.det) || __double_as_longlong(t0) != __double_as_longlong(o.tr2) ||
                  __double_as_longlong(i0) != __double_as_longlong(o.inv[P - 1][0]);
        differs = __any(differs);
    }
    if (!uniform || (tid & 63) == 0) {
        if (differs) o.tr3 = __longlong_as_double(0x7ff8dead00000000LL);
        out[idx] = o;
    }
}

template <int P>
static void make_family(int fam, int nmat, std::vector<double> *mats, uint64_t seed) {
    std::mt19937_64 rng(seed);
    std::uniform_real_distribution<double> U(0.05, 3.0);
    std::normal_distribution<double> N(0.0, 1.0);
    mats->assign((size_t)nmat * 3 * P * P, 0.0);
    for (int t = 0; t < nmat; t++) {
        double *B = mats->data() + (size_t)t * 3 * P * P, *dB = B + P * P, *d2B = dB + P * P;
        if (fam == 0 || fam == 1) {
            // Gram matrices X' diag(w) X of a factor design with an intercept (cells = P), w, -w^2, 2 w^3
            double w[P];
            for (int c = 0; c < P; c++) w[c] = U(rng) * 40.0;
            unsigned drop = 0;
            if (fam == 1) { drop = (unsigned)(rng() % (1u << P)); drop &= ~1u; if (drop == (unsigned)((1u << P) - 2)) drop >>= 1; }
            for (int k = 0; k < 3; k++) {
                double *M = B + k * P * P;
                for (int c = 0; c < P; c++) {
                    const double v = k == 0 ? w[c] : (k == 1 ? -w[c] * w[c] : 2.0 * w[c] * w[c] * w[c]);
                    // cell c: x = e_0 + e_c (c > 0), e_0 for the reference level
                    if ((drop >> c) & 1u) continue;                    // (a cell whose samples are all weighted out)
                    M[0] += v;
                    if (c) { M[c * P + c] += v; M[c] += v; M[c * P] += v; }
                }
                for (int c = 1; c < P; c++)
                    if ((drop >> c) & 1u) {
                        for (int j = 0; j < P; j++) { M[c * P + j] = 0.0; M[j * P + c] = 0.0; }
                        if (k == 0) M[c * P + c] = 1.0;
                    }
            }
        } else if (fam == 2) {
            for (int i = 0; i < 3 * P * P; i++) B[i] = N(rng);
            for (int k = 1; k < 3; k++)                                // dB, d2B symmetric as in the kernels
                for (int i = 0; i < P; i++)
                    for (int j = 0; j < i; j++) B[k * P * P + i * P + j] = B[k * P * P + j * P + i];
        } else {
            // exact ties in the pivot column: small integers, made non-singular by a dominant last row
            for (int i = 0; i < P * P; i++) B[i] = (double)((int)(rng() % 3) - 1);
            for (int j = 0; j < P; j++) B[(P - 1) * P + j] += (j == (int)(t % P)) ? 7.0 : 0.0;
            for (int i = 0; i < P; i++) B[i * P + i] += (rng() & 1) ? 1.0 : -1.0;
            for (int i = 0; i < 2 * P * P; i++) dB[i] = (double)((int)(rng() % 5) - 2);
            for (int k = 1; k < 3; k++)
                for (int i = 0; i < P; i++)
                    for (int j = 0; j < i; j++) B[k * P * P + i * P + j] = B[k * P * P + j * P + i];
        }
    }
}

static bool same_bits(double a, double b) {
    uint64_t x, y;
    memcpy(&x, &a, 8); memcpy(&y, &b, 8);
    return x == y || (a != a && b != b);
}

static bool g_host_only = false;

template <int P, int FORM>
static int run_form(int *total_bad) {
    static const char *fname[] = {"swap", "select", "pairsel"};
    static const char *famname[] = {"gram", "gram+masked", "random", "ties"};
    const int nmat = 4096;
    for (int fam = 0; fam < 4; fam++) {
        std::vector<double> mats;
        make_family<P>(fam, nmat, &mats, 1234 + 17 * fam + P);
        std::vector<Out<P>> ref(nmat);
        int swaps = 0;
        for (int t = 0; t < nmat; t++) {
            algebra<P, FORM>(mats.data() + (size_t)t * 3 * P * P, &ref[t]);
            LU<P, FORM> lu;
            for (int i = 0; i < P; i++) for (int j = 0; j < P; j++) lu.a[i][j] = mats[(size_t)t * 3 * P * P + i * P + j];
            lu.factor();
            for (int k = 0; k < P; k++) swaps += lu.piv[k] != k;
        }
        if (g_host_only) {
            // sanitizer leg (no device): the three forms must agree with each other bit for bit on the host as well
            int bad = 0;
            for (int t = 0; t < nmat; t++) {
                Out<P> o;
                algebra<P, LU_SWAP>(mats.data() + (size_t)t * 3 * P * P, &o);
                bad += memcmp(&o, &ref[t], sizeof o) != 0;
            }
            printf("P=%-2d form=%-7s host-only family=%-11s pivots!=k %5.2f/matrix  differs from the swap form on %d of %d\n", P, fname[FORM],
                   famname[fam], (double)swaps / nmat, bad, nmat);
            *total_bad += bad;
            continue;
        }
        double *dm; Out<P> *dout;
        CK(hipMalloc(&dm, mats.size() * 8)); CK(hipMalloc(&dout, sizeof(Out<P>) * nmat));
        CK(hipMemcpy(dm, mats.data(), mats.size() * 8, hipMemcpyHostToDevice));
        for (int uniform = 1; uniform >= 0; uniform--) {
            CK(hipMemset(dout, 0, sizeof(Out<P>) * nmat));
            const int threads = uniform ? nmat * 64 : nmat;
            hipLaunchKernelGGL((probe_kernel<P, FORM>), dim3((threads + 255) / 256), dim3(256), 0, 0, dm, dout, nmat, uniform);
            CK(hipDeviceSynchronize());
            std::vector<Out<P>> got(nmat);
            CK(hipMemcpy(got.data(), dout, sizeof(Out<P>) * nmat, hipMemcpyDeviceToHost));
            int bad_inv = 0, bad_tr = 0, first = -1;
            for (int t = 0; t < nmat; t++) {
                bool bi = false, bt = false;
                for (int i = 0; i < P; i++) for (int j = 0; j < P; j++) bi |= !same_bits(got[t].inv[i][j], ref[t].inv[i][j]);
                bt = !same_bits(got[t].det, ref[t].det) || !same_bits(got[t].tr1, ref[t].tr1) ||
                     !same_bits(got[t].tr2, ref[t].tr2) || !same_bits(got[t].tr3, ref[t].tr3);
                bad_inv += bi; bad_tr += bt;
                if ((bi || bt) && first < 0) first = t;
            }
            printf("P=%-2d form=%-7s usage=%s family=%-11s pivots!=k %5.2f/matrix  inverse mismatches %4d  det/trace mismatches %4d  of %d%s\n",
                   P, fname[FORM], uniform ? "A/B wave-uniform" : "C per-lane     ", famname[fam], (double)swaps / nmat, bad_inv, bad_tr, nmat,
                   (bad_inv || bad_tr) ? "   <-- MISMATCH" : "");
            if (first >= 0) {
                printf("    first mismatch: matrix %d  device tr2 %.17g host %.17g  device inv[0][0] %.17g host %.17g\n", first,
                       got[first].tr2, ref[first].tr2, got[first].inv[0][0], ref[first].inv[0][0]);
            }
            *total_bad += bad_inv + bad_tr;
        }
        CK(hipFree(dm)); CK(hipFree(dout));
    }
    return 0;
}

template <int P>
static int run_p(int *bad) {
    int rc;
    if ((rc = run_form<P, LU_SWAP>(bad))) return rc;
    if ((rc = run_form<P, LU_SELECT>(bad))) return rc;
    if ((rc = run_form<P, LU_PAIRSEL>(bad))) return rc;
    return 0;
}

int main(int argc, char **argv) {
    g_host_only = argc > 1 && !strcmp(argv[1], "host");
    int bad = 0, rc;
    if ((rc = run_p<4>(&bad))) return rc;
    if ((rc = run_p<5>(&bad))) return rc;
    if ((rc = run_p<6>(&bad))) return rc;
    if ((rc = run_p<10>(&bad))) return rc;
    printf("lu_probe: %d mismatching results in total\n", bad);
    return bad ? 1 : 0;
}
