// Microbenchmark: VALU issue rates on gfx950 that decide the chamfer inner-loop form.
//   v_fma_f32 vs v_pk_fma_f32 (does packed fp32 double the per-instruction rate?),
//   v_sub/v_mul/v_min3, and the actual 6.5-op distance body.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o tools/ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %d at %d\n", (int)e, __LINE__); return 1; } } while (0)

constexpr int ITERS = 2048;

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, float seed) {
  float a[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = seed + threadIdx.x * 1e-3f + i;
  float b = seed * 0.5f, c = seed * 0.25f;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (MODE == 0) {  // 16 independent v_fma_f32
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
      } else if (MODE == 1) {  // 8 independent v_pk_fma_f32 (16 lanes of work)
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
          asm volatile("v_pk_fma_f32 %0, %1, %2, %0"
                       : "+v"(*reinterpret_cast<double*>(&a[i]))
                       : "v"(*reinterpret_cast<double*>(&a[(i + 2) & 15])), "v"(*reinterpret_cast<double*>(&a[(i + 4) & 15])));
        }
      } else if (MODE == 2) {  // v_min3_f32
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      } else if (MODE == 3) {  // v_sub_f32
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
      } else if (MODE == 4) {  // v_pk_add_f32
#pragma unroll
        for (int i = 0; i < 16; i += 2)
          asm volatile("v_pk_add_f32 %0, %1, %0"
                       : "+v"(*reinterpret_cast<double*>(&a[i]))
                       : "v"(*reinterpret_cast<double*>(&a[(i + 2) & 15])));
      } else if (MODE == 5) {  // v_pk_mul_f32
#pragma unroll
        for (int i = 0; i < 16; i += 2)
          asm volatile("v_pk_mul_f32 %0, %1, %0"
                       : "+v"(*reinterpret_cast<double*>(&a[i]))
                       : "v"(*reinterpret_cast<double*>(&a[(i + 2) & 15])));
      }
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += a[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
int run(const char* name, double lane_ops_per_inst, int insts_per_iter) {
  const int blocks = 256 * 8;
  float* out;
  CK(hipMalloc(&out, blocks * 256 * sizeof(float)));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 1.0f);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 1.0f);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= 5;
  double insts = (double)blocks * 4 /*waves*/ * ITERS * insts_per_iter;
  double wave_inst_per_s = insts / (ms * 1e-3);
  // per SIMD: 1024 SIMDs
  double cyc_per_inst = 2.4e9 / (wave_inst_per_s / 1024.0);
  printf("%-14s %8.3f ms  %7.2f Tlane-op/s  ~%.2f cycles/wave-inst/SIMD @2.4GHz\n", name, ms,
         wave_inst_per_s * 64 * lane_ops_per_inst / 1e12, cyc_per_inst);
  CK(hipFree(out));
  return 0;
}

int main() {
  run<0>("v_fma_f32", 1, 64);
  run<1>("v_pk_fma_f32", 2, 32);
  run<2>("v_min3_f32", 1, 64);
  run<3>("v_sub_f32", 1, 64);
  run<4>("v_pk_add_f32", 2, 32);
  run<5>("v_pk_mul_f32", 2, 32);
  return 0;
}
