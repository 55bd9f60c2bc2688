// dsq_prof.hpp -- phase profile of the rolled wide kernels (profiling builds only: `make prof`, -DDSQ_WIDE_PROF): thread 0
// of every workgroup adds the shader-clock cycles it spends in each phase; the launch prints the shares.  (PC sampling is
// not available on this pool; this is what directs the work on these kernels.)
#pragma once
#ifdef DSQ_WIDE_PROF
#define DSQ_PROF_SLOTS 12
#define DSQ_PROF_DECL unsigned long long prof_t0 = clock64(), prof_acc[DSQ_PROF_SLOTS] = {}
#define DSQ_PROF(slot) do { if (tid == 0) { const unsigned long long t1_ = clock64(); prof_acc[slot] += t1_ - prof_t0; prof_t0 = t1_; } } while (0)
#define DSQ_PROF_FLUSH(sym) do { if (tid == 0) for (int q_ = 0; q_ < DSQ_PROF_SLOTS; q_++) atomicAdd(&sym[q_], prof_acc[q_]); } while (0)
#else
#define DSQ_PROF_DECL
#define DSQ_PROF(slot)
#define DSQ_PROF_FLUSH(sym)
#endif
