// Optional per-phase cycle accounting of a kernel (development builds: `make prof` defines KAMD_PHASE_PROF; see
// tools/phase_prof.py).  Every wavefront accumulates the clock ticks between PHASE_MARKs in registers and adds them to a
// per-translation-unit table when it ends.  Compiles to nothing in the product build.
#pragma once
#ifdef KAMD_PHASE_PROF
#define PHASE_TABLE(name) __device__ unsigned long long name[16];
#define PHASE_DECL unsigned long long ph_t = clock64(), ph_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define PHASE_MARK(i) { const unsigned long long ph_n = clock64(); ph_acc[i] += ph_n - ph_t; ph_t = ph_n; }
#define PHASE_FLUSH(name) { if ((threadIdx.x & 63) == 0) for (int ph_i = 0; ph_i < 10; ++ph_i) atomicAdd(&name[ph_i], ph_acc[ph_i]); }
#else
#define PHASE_TABLE(name)
#define PHASE_DECL
#define PHASE_MARK(i)
#define PHASE_FLUSH(name)
#endif
