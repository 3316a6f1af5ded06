// Cold-cache read bandwidth of a two-array dot product on one MI355X, for a few ways of walking the arrays
// (hipcc --offload-arch=gfx950 -O3 stream_read.hip -o stream_read).  A pool of buffer pairs larger than the 256 MB MALL is
// rotated so that no pass finds its data in a cache.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int U, int MODE>  // MODE 0: contiguous share per workgroup, 1: grid-stride, 2: contiguous + nontemporal
__global__ __launch_bounds__(256) void dot_kernel(const f32x4* __restrict__ x, const f32x4* __restrict__ w, long long nv,
                                                   double* __restrict__ partial) {
  double acc = 0;
  long long i, hi, step;
  if (MODE == 1) {
    i = (long long)blockIdx.x * 256 + threadIdx.x;
    hi = nv;
    step = (long long)gridDim.x * 256;
  } else {
    const long long share = (nv + gridDim.x - 1) / gridDim.x;
    i = (long long)blockIdx.x * share + threadIdx.x;
    hi = (long long)blockIdx.x * share + share < nv ? (long long)blockIdx.x * share + share : nv;
    step = 256;
  }
  for (; i + (U - 1) * step < hi; i += U * step) {
    f32x4 a[U], b[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (MODE == 2) {
        a[u] = __builtin_nontemporal_load(x + i + u * step);
        b[u] = __builtin_nontemporal_load(w + i + u * step);
      } else {
        a[u] = x[i + u * step];
        b[u] = w[i + u * step];
      }
    }
    float part = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) part += a[u].x * b[u].x + a[u].y * b[u].y + a[u].z * b[u].z + a[u].w * b[u].w;
    acc += (double)part;
  }
  for (; i < hi; i += step) {
    const f32x4 a = x[i], b = w[i];
    acc += (double)(a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w);
  }
  for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
  if ((threadIdx.x & 63) == 0) atomicAdd(partial + blockIdx.x, acc);
}

template <int U, int MODE>
void run(const char* name, int groups, std::vector<float*>& xs, std::vector<float*>& ws, long long n, double* partial) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int reps = 24;
  for (int r = 0; r < 4; ++r)
    hipLaunchKernelGGL((dot_kernel<U, MODE>), dim3(groups), dim3(256), 0, 0, (const f32x4*)xs[r % xs.size()],
                       (const f32x4*)ws[r % ws.size()], n / 4, partial);
  hipEventRecord(e0, 0);
  for (int r = 0; r < reps; ++r)
    hipLaunchKernelGGL((dot_kernel<U, MODE>), dim3(groups), dim3(256), 0, 0, (const f32x4*)xs[r % xs.size()],
                       (const f32x4*)ws[r % ws.size()], n / 4, partial);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / reps;
  printf("%-34s groups %5d: %7.1f us  %6.0f GB/s\n", name, groups, us, 2.0 * n * 4 / us / 1e3);
}

int main() {
  const long long n = 8ll * 1024 * 1024 * 4;  // 33.5 M floats per array = 134 MB; a pair = 268 MB
  const int pool = 6;
  std::vector<float*> xs(pool), ws(pool);
  for (int i = 0; i < pool; ++i) {
    hipMalloc(&xs[i], n * 4);
    hipMalloc(&ws[i], n * 4);
    hipMemset(xs[i], 0, n * 4);
    hipMemset(ws[i], 0, n * 4);
  }
  double* partial;
  hipMalloc(&partial, 65536 * 8);
  hipMemset(partial, 0, 65536 * 8);
  for (int groups : {1024, 2048, 4096, 8192, 16384}) {
    run<4, 0>("contiguous share, 4 in flight", groups, xs, ws, n, partial);
    run<8, 0>("contiguous share, 8 in flight", groups, xs, ws, n, partial);
    run<4, 1>("grid stride, 4 in flight", groups, xs, ws, n, partial);
    run<8, 1>("grid stride, 8 in flight", groups, xs, ws, n, partial);
    run<4, 2>("contiguous, nontemporal, 4", groups, xs, ws, n, partial);
    run<1, 1>("grid stride, 1 in flight", groups, xs, ws, n, partial);
  }
  std::vector<float*> one_x(1, xs[0]), one_w(1, ws[0]);
  run<4, 0>("SAME pair every pass (MALL)", 4096, one_x, one_w, n, partial);
  return 0;
}
