// Optional per-phase cycle accounting of a kernel (development builds: `make prof` defines KAMD_PHASE_PROF; see
// tools/phase_prof.py).  Every wavefront accumulates the clock ticks between PHASE_MARKs in registers and adds them to a
// per-translation-unit table when it ends.  Compiles to nothing in the product build.
#pragma once
#ifdef KAMD_PHASE_PROF
// (Round 6: PHASE_ROWS rows of 16 counters, a workgroup adds to the row of its block id: with ONE row, a launch of 131k wavefronts
// queued 1.3 M atomics on ten addresses -- ~40 ns each, tens of ms -- and every memory phase of the profiled kernel, waiting behind
// them, read 50x too long next to the compute phases: round 5's "the coverage loops are 1 % of raster_tile" was that artefact.)
#define PHASE_ROWS 2048
#define PHASE_TABLE(name) __device__ unsigned long long name[PHASE_ROWS * 16];
#define PHASE_ROW(name) (name + ((blockIdx.x + blockIdx.y * gridDim.x + blockIdx.z * gridDim.x * gridDim.y) % PHASE_ROWS) * 16)
// host side: sum the rows into out16 (and clear them)
#define PHASE_READ(name, out16, reset, rc)                                                                         \
  {                                                                                                              \
    static unsigned long long ph_host[PHASE_ROWS * 16];                                                          \
    rc = (int)hipMemcpyFromSymbol(ph_host, HIP_SYMBOL(name), sizeof(ph_host));                                   \
    for (int ph_i = 0; ph_i < 16; ++ph_i) {                                                                      \
      out16[ph_i] = 0;                                                                                           \
      for (int ph_r = 0; ph_r < PHASE_ROWS; ++ph_r) out16[ph_i] += ph_host[ph_r * 16 + ph_i];                    \
    }                                                                                                            \
    if (reset) {                                                                                                 \
      for (size_t ph_i = 0; ph_i < sizeof(ph_host) / 8; ++ph_i) ph_host[ph_i] = 0;                               \
      rc |= (int)hipMemcpyToSymbol(HIP_SYMBOL(name), ph_host, sizeof(ph_host));                                  \
    }                                                                                                            \
  }
#define PHASE_DECL unsigned long long ph_t = clock64(), ph_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define PHASE_MARK(i) { const unsigned long long ph_n = clock64(); ph_acc[i] += ph_n - ph_t; ph_t = ph_n; }
#define PHASE_FLUSH(name) { if ((threadIdx.x & 63) == 0) for (int ph_i = 0; ph_i < 10; ++ph_i) atomicAdd(&PHASE_ROW(name)[ph_i], ph_acc[ph_i]); }
#else
#define PHASE_TABLE(name)
#define PHASE_DECL
#define PHASE_MARK(i)
#define PHASE_FLUSH(name)
#endif
