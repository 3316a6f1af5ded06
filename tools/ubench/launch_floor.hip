// Development probe: what does a kernel that does (almost) nothing cost on MI355X, as a function of its grid, its register
// allocation and what ran before it?  hipcc --offload-arch=gfx950 -O3 -o launch_floor launch_floor.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_empty(int* out) { if (out == (int*)1) out[0] = 1; }
template <int NV>
__global__ __launch_bounds__(256) void k_regs(const float* in, float* out, int n) {   // NV live floats per lane, never stored
  float v[NV];
  const int i = blockIdx.x * 256 + threadIdx.x;
#pragma unroll
  for (int k = 0; k < NV; ++k) v[k] = (float)(i * (k + 1));
  float s = 0;
#pragma unroll
  for (int k = 0; k < NV; ++k) s += v[k] * v[(k + 7) % NV];
  if (s == 1234567.f) out[i] = s;
}
__global__ __launch_bounds__(256) void k_load(const float* in, float* out, int n) {    // one coalesced 4-byte load per lane
  const int i = blockIdx.x * 256 + threadIdx.x;
  const float s = i < n ? in[i] : 0.f;
  if (s == 1234567.f) out[i] = s;
}
__global__ __launch_bounds__(256) void k_store(float* out, int n) {                    // one coalesced 4-byte store per lane
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = (float)i;
}
__global__ __launch_bounds__(256) void k_div64(const long long* in, float* out, int F) {
  const long long f = (long long)blockIdx.x * 256 + threadIdx.x;
  const int b = (int)(f / F);
  if (b == 123456789) out[0] = 1.f;
}

template <typename F>
float time_us(F launch, int reps, hipStream_t st) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 5; ++i) launch();
  hipStreamSynchronize(st);
  hipEventRecord(a, st);
  for (int i = 0; i < reps; ++i) launch();
  hipEventRecord(b, st);
  hipEventSynchronize(b);
  float ms = 0; hipEventElapsedTime(&ms, a, b);
  return ms * 1000.f / reps;
}

int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  float *in, *out; const int N = 64 << 20;
  CK(hipMalloc(&in, N * 4)); CK(hipMalloc(&out, N * 4));
  CK(hipMemsetAsync(in, 0, N * 4, st));
  printf("back-to-back launches on one stream, us per launch (100 launches)\n");
  for (int wgs : {1, 256, 1563, 2048, 8192, 32768}) {
    const float e = time_us([&] { hipLaunchKernelGGL(k_empty, dim3(wgs), dim3(256), 0, st, (int*)nullptr); }, 100, st);
    const float r16 = time_us([&] { hipLaunchKernelGGL(k_regs<16>, dim3(wgs), dim3(256), 0, st, in, out, N); }, 100, st);
    const float r64 = time_us([&] { hipLaunchKernelGGL(k_regs<64>, dim3(wgs), dim3(256), 0, st, in, out, N); }, 100, st);
    const float l = time_us([&] { hipLaunchKernelGGL(k_load, dim3(wgs), dim3(256), 0, st, in, out, N); }, 100, st);
    const float s = time_us([&] { hipLaunchKernelGGL(k_store, dim3(wgs), dim3(256), 0, st, out, N); }, 100, st);
    const float d = time_us([&] { hipLaunchKernelGGL(k_div64, dim3(wgs), dim3(256), 0, st, (const long long*)in, out, 50000); }, 100, st);
    printf("  %6d workgroups x 256: empty %6.2f | 16 live regs %6.2f | 64 live regs %6.2f | one load %6.2f | one store %6.2f | 64-bit division %6.2f\n", wgs, e, r16, r64, l, s, d);
  }
  // a kernel after a big write: does it pay for the dirty lines the previous kernel left in L2?
  for (int mb : {0, 8, 32, 128}) {
    const float t = time_us([&] {
      if (mb) hipLaunchKernelGGL(k_store, dim3(mb * 1024), dim3(256), 0, st, out, mb << 18);
      hipLaunchKernelGGL(k_empty, dim3(1563), dim3(256), 0, st, (int*)nullptr);
    }, 100, st);
    const float t0 = mb ? time_us([&] { hipLaunchKernelGGL(k_store, dim3(mb * 1024), dim3(256), 0, st, out, mb << 18); }, 100, st) : 0.f;
    printf("  store %3d MB then empty(1563): pair %7.2f us, the store kernel alone %7.2f us\n", mb, t, t0);
  }
  return 0;
}
