// Development probe: the speed of light of raster_tile's ACCESS PATTERN at C4 (8 views x 1024^2, 16 x 16 tiles).
// No arithmetic: every workgroup reads its tile's candidate count (one dependent scalar load, as the product must), a tile with
// candidates reads count x REC_BYTES of tile-contiguous records (16-byte loads) and writes 36 B/pixel, a background tile writes
// 24 B/pixel (sel_idx 8 + features 12 + soft mask 4; the fused operator's weights are not written there) -- nontemporal 16-byte
// stores, rows of 16 pixels, the product's grid order (views fastest, then tile columns, then tile rows).
//   form A  a 256-thread workgroup per tile, every wavefront its own 16 x 4 strip          (the product's shape)
//   form B  the same grid, background tiles written by wavefront 0 alone (7 store instructions), wavefronts 1-3 leave
//   form C  a 64-thread workgroup per tile (one wavefront does everything)
//   form D  a workgroup per (view, tile row): whole image rows of the background as one contiguous stream, tiles with faces skipped
//           + form A restricted to the tiles with faces  (two launches: "fill" + "tiles")
//   form E  the same bytes as one flat nontemporal float4 fill + a flat 16-byte read of the records (no tile shape at all)
// Build: hipcc --offload-arch=gfx950 -O3 -o raster_ceiling raster_ceiling.hip ; run: ./raster_ceiling [reps]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cmath>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int B = 8, H = 1024, W = 1024, TX = W / 16, TY = H / 16, NT = TX * TY, D = 3, REC_BYTES = 96;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st16(void* p, u32x4 v) { __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p)); }

struct Out { int64_t* sel; float* feat; float* wts; float* soft; };

// one wavefront writes `rows` rows of 16 pixels starting at pixel p0 (row stride W): sel 8 chunks/row, feat 12, soft 4, weights 12
__device__ __forceinline__ void write_rows(const Out& o, size_t p0, int rows, int lane, bool with_w, unsigned int val) {
  const u32x4 v = {val, val, val, val};
  for (int r = lane >> 3; r < rows; r += 8) st16(reinterpret_cast<char*>(o.sel + p0 + (size_t)r * W) + (lane & 7) * 16, v);
  if ((lane & 15) < 12) {
    for (int r = lane >> 4; r < rows; r += 4) st16(reinterpret_cast<char*>(o.feat + (p0 + (size_t)r * W) * D) + (lane & 15) * 16, v);
    if (with_w)
      for (int r = lane >> 4; r < rows; r += 4) st16(reinterpret_cast<char*>(o.wts + (p0 + (size_t)r * W) * 3) + (lane & 15) * 16, v);
  }
  for (int r = lane >> 2; r < rows; r += 16) st16(reinterpret_cast<char*>(o.soft + p0 + (size_t)r * W) + (lane & 3) * 16, v);
}
// the workgroup's threads read n records of the tile (16-byte pieces, coalesced); returns something that depends on them
__device__ __forceinline__ unsigned int read_records(const uint4* rec, unsigned int first, unsigned int n, int tid, int nthreads) {
  unsigned int acc = 0;
  const uint4* p = rec + (size_t)first * (REC_BYTES / 16);
  for (unsigned int i = tid; i < n * (REC_BYTES / 16); i += nthreads) { const uint4 q = p[i]; acc += q.x ^ q.y ^ q.z ^ q.w; }
  return acc;
}

template <int FORM>  // 0 = A, 1 = B
__global__ __launch_bounds__(256, 8) void k_tile_wg(const uint2* __restrict__ cnt, const uint4* __restrict__ rec, Out o, int only_faces) {
  const int b = blockIdx.x, tx = blockIdx.y, ty = blockIdx.z, tile = ty * TX + tx;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint2 c = cnt[(size_t)b * NT + tile];  // {first record, count}: uniform -> scalar load
  if (c.y == 0u) {
    if (only_faces) return;
    if (FORM == 1) {
      if (wave != 0) return;
      write_rows(o, ((size_t)b * H + ty * 16) * W + tx * 16, 16, lane, false, 0u);
    } else {
      write_rows(o, ((size_t)b * H + ty * 16 + wave * 4) * W + tx * 16, 4, lane, false, 0u);
    }
    return;
  }
  const unsigned int acc = read_records(rec, c.x, c.y, threadIdx.x, 256);
  write_rows(o, ((size_t)b * H + ty * 16 + wave * 4) * W + tx * 16, 4, lane, true, acc == 0x12345u ? 1u : 0u);
}
// form F: the count and out -- what 32 768 workgroups cost before they do anything
__global__ __launch_bounds__(256, 8) void k_count_only(const uint2* __restrict__ cnt, unsigned int* out) {
  const int b = blockIdx.x, tx = blockIdx.y, ty = blockIdx.z, tile = ty * TX + tx;
  const uint2 c = cnt[(size_t)b * NT + tile];
  if (c.y == 0xFFFFFFFFu) out[0] = c.x;
}
__global__ __launch_bounds__(64) void k_tile_wave(const uint2* __restrict__ cnt, const uint4* __restrict__ rec, Out o) {
  const int b = blockIdx.x, tx = blockIdx.y, ty = blockIdx.z, tile = ty * TX + tx;
  const uint2 c = cnt[(size_t)b * NT + tile];
  unsigned int acc = 0;
  if (c.y != 0u) acc = read_records(rec, c.x, c.y, threadIdx.x, 64);
  write_rows(o, ((size_t)b * H + ty * 16) * W + tx * 16, 16, threadIdx.x, c.y != 0u, acc == 0x12345u ? 1u : 0u);
}
// a workgroup per (view, tile row): the 64 counts of the row, then 16 image rows streamed, tiles with candidates skipped
__global__ __launch_bounds__(256) void k_fill_rows(const uint2* __restrict__ cnt, Out o) {
  const int b = blockIdx.x % B, ty = blockIdx.x / B;
  __shared__ unsigned long long s_faces;
  if (threadIdx.x < 64) {
    const unsigned long long m = __ballot(cnt[(size_t)b * NT + ty * TX + threadIdx.x].y != 0u);
    if (threadIdx.x == 0) s_faces = m;
  }
  __syncthreads();
  const unsigned long long faces = s_faces;
  const u32x4 z = {0u, 0u, 0u, 0u};
  const size_t p0 = ((size_t)b * H + ty * 16) * W;
  // sel: 16 rows x 8192 B = 512 chunks per row
  for (int i = threadIdx.x; i < 16 * 512; i += 256) {
    const int r = i >> 9, c = i & 511, t = c >> 3;
    if (!((faces >> t) & 1ull)) st16(reinterpret_cast<char*>(o.sel + p0 + (size_t)r * W) + c * 16, z);
  }
  for (int i = threadIdx.x; i < 16 * 768; i += 256) {   // features: 12288 B per row = 768 chunks, 12 per tile
    const int r = i / 768, c = i - r * 768, t = c / 12;
    if (!((faces >> t) & 1ull)) st16(reinterpret_cast<char*>(o.feat + (p0 + (size_t)r * W) * D) + c * 16, z);
  }
  for (int i = threadIdx.x; i < 16 * 256; i += 256) {   // soft: 4096 B per row = 256 chunks, 4 per tile
    const int r = i >> 8, c = i & 255, t = c >> 2;
    if (!((faces >> t) & 1ull)) st16(reinterpret_cast<char*>(o.soft + p0 + (size_t)r * W) + c * 16, z);
  }
}
__global__ __launch_bounds__(256) void k_flat_fill(u32x4* p, size_t n16) {
  const u32x4 z = {0u, 0u, 0u, 0u};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) st16(p + i, z);
}
__global__ __launch_bounds__(256) void k_flat_read(const uint4* p, size_t n16, unsigned int* out) {
  unsigned int acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) { const uint4 q = p[i]; acc += q.x ^ q.y ^ q.z ^ q.w; }
  if (acc == 0x12345u) out[0] = acc;
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

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 100;
  hipStream_t st; CK(hipStreamCreate(&st));
  // the C4 sphere, as the binning launch sees it: a disc of ~852 tiles per view, ~30 candidates per tile inside, up to ~200 on the ring
  std::vector<uint2> cnt((size_t)B * NT);
  size_t pairs = 0, face_tiles = 0;
  for (int b = 0; b < B; ++b)
    for (int ty = 0; ty < TY; ++ty)
      for (int tx = 0; tx < TX; ++tx) {
        const double dx = tx + 0.5 - TX / 2.0, dy = ty + 0.5 - TY / 2.0, d = std::sqrt(dx * dx + dy * dy), R = 16.45;
        unsigned int n = 0;
        if (d < R + 0.5) n = d > R - 0.5 ? 150 : (d > R - 1.5 ? 90 : (d > R - 3.5 ? 45 : 24));
        cnt[(size_t)b * NT + ty * TX + tx] = make_uint2((unsigned int)pairs, n);
        pairs += n;
        face_tiles += n != 0;
      }
  const size_t npx = (size_t)B * H * W;
  Out o;
  uint2* d_cnt; uint4* d_rec; unsigned int* d_out;
  CK(hipMalloc(&o.sel, npx * 8)); CK(hipMalloc(&o.feat, npx * 4 * D)); CK(hipMalloc(&o.wts, npx * 12)); CK(hipMalloc(&o.soft, npx * 4));
  CK(hipMalloc(&d_cnt, cnt.size() * 8)); CK(hipMalloc(&d_rec, pairs * REC_BYTES + 256)); CK(hipMalloc(&d_out, 256));
  CK(hipMemcpy(d_cnt, cnt.data(), cnt.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemset(d_rec, 1, pairs * REC_BYTES + 256));
  const double bg_tiles = (double)B * NT - face_tiles;
  const double wr = bg_tiles * 256 * 24 + (double)face_tiles * 256 * 36, rd = (double)pairs * REC_BYTES + (double)B * NT * 8;
  printf("C4 shape: %zu tiles with candidates of %d, %zu (tile, face) pairs; bytes written %.1f MB, read %.1f MB, together %.1f MB\n",
         face_tiles, B * NT, pairs, wr / 1e6, rd / 1e6, (wr + rd) / 1e6);
  const dim3 grid(B, TX, TY);
  auto report = [&](const char* name, float us) { printf("%-72s %7.1f us  %5.2f TB/s\n", name, us, (wr + rd) / us / 1e6); };
  for (int round = 0; round < 2; ++round) {
    report("A  workgroup per tile, a strip per wavefront (the product's shape)",
           time_us([&] { hipLaunchKernelGGL(k_tile_wg<0>, grid, dim3(256), 0, st, d_cnt, d_rec, o, 0); }, reps, st));
    report("B  workgroup per tile, background tiles by wavefront 0 alone",
           time_us([&] { hipLaunchKernelGGL(k_tile_wg<1>, grid, dim3(256), 0, st, d_cnt, d_rec, o, 0); }, reps, st));
    report("F  workgroup per tile: the count and out (no stores, no records)",
           time_us([&] { hipLaunchKernelGGL(k_count_only, grid, dim3(256), 0, st, d_cnt, d_out); }, reps, st));
    report("C  one wavefront per tile",
           time_us([&] { hipLaunchKernelGGL(k_tile_wave, grid, dim3(64), 0, st, d_cnt, d_rec, o); }, reps, st));
    report("D  background rows streamed per (view, tile row) + tiles with candidates (2 launches)",
           time_us([&] { hipLaunchKernelGGL(k_fill_rows, dim3(B * TY), dim3(256), 0, st, d_cnt, o);
                         hipLaunchKernelGGL(k_tile_wg<0>, grid, dim3(256), 0, st, d_cnt, d_rec, o, 1); }, reps, st));
    const float fill = time_us([&] { hipLaunchKernelGGL(k_fill_rows, dim3(B * TY), dim3(256), 0, st, d_cnt, o); }, reps, st);
    printf("   D's fill alone %.1f us (%.2f TB/s on its %.1f MB); A on an all-background image:", fill, bg_tiles * 256 * 24 / fill / 1e6, bg_tiles * 256 * 24 / 1e6);
    {
      std::vector<uint2> zero((size_t)B * NT, make_uint2(0, 0));
      uint2* d_zero; CK(hipMalloc(&d_zero, zero.size() * 8)); CK(hipMemcpy(d_zero, zero.data(), zero.size() * 8, hipMemcpyHostToDevice));
      const float a0 = time_us([&] { hipLaunchKernelGGL(k_tile_wg<0>, grid, dim3(256), 0, st, d_zero, d_rec, o, 0); }, reps, st);
      const float b0 = time_us([&] { hipLaunchKernelGGL(k_tile_wg<1>, grid, dim3(256), 0, st, d_zero, d_rec, o, 0); }, reps, st);
      printf(" %.1f us (A), %.1f us (B) = %.2f / %.2f TB/s on %.1f MB\n", a0, b0, (double)B * NT * 256 * 24 / a0 / 1e6, (double)B * NT * 256 * 24 / b0 / 1e6,
             (double)B * NT * 256 * 24 / 1e6);
      hipFree(d_zero);
    }
    {
      const size_t n16 = (size_t)(wr / 16);
      static u32x4* flat = nullptr;
      if (flat == nullptr) CK(hipMalloc(&flat, n16 * 16));
      const float f = time_us([&] { hipLaunchKernelGGL(k_flat_fill, dim3(256 * 16), dim3(256), 0, st, flat, n16);
                                    hipLaunchKernelGGL(k_flat_read, dim3(256 * 8), dim3(256), 0, st, d_rec, pairs * REC_BYTES / 16, d_out); }, reps, st);
      report("E  the same bytes as one flat nontemporal fill + a flat read (2 launches)", f);
    }
  }
  return 0;
}
