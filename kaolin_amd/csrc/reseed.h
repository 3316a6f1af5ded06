// The reference's per-tile re-seed of the running minimum, shared by the sided-distance searches.
//
// kaolin/csrc/metrics/sided_distance_cuda.cu:60-196 walks the targets in tiles of 512: inside a tile the running best is
// seeded UNCONDITIONALLY by the tile's first target (`k == 0 ||`, :88,138,187) and only improved by `d < best`; tiles are
// merged with `k2 == 0 || result > best` (:193).  With a NaN distance at a tile's first target nothing in that tile compares
// below the seed and the tile's NaN never compares below the running result: the whole tile is ignored for that query (tile 0:
// the NaN sticks as the result).  So the reference's answer is
//     (NaN, 0)                                            if d(q, target 0) is NaN,
//     lexicographic min (d, index) over the LIVE tiles     otherwise (a tile is live when d(q, its first target) is not NaN;
//                                                          NaN distances elsewhere never win).
// The searches here (all pairs and the uniform grid) compute the minimum over ALL targets with target 0's seed rule.  That is
// the reference's answer unless the winner sits in a dead tile -- which takes a non-finite coordinate at a target index 512 k.
// The fix-up below costs nothing without such targets (callers gate it on a flag found once per call) and is exact with them:
// the wavefront takes its affected queries in turn, lane = target, and redoes the search over the live tiles only.
// Cost with them: a redo is O(M / 64) steps of one wavefront per affected query, one query after the other -- a cloud or
// mesh with non-finite coordinates at MANY indices 512 k (every tile dead for most queries) degrades towards O(N * M / 64) serial
// work, a cliff that depends on the data; such inputs are outside what the reference itself handles meaningfully (its answer for
// them is "the best of whatever tiles happen to be alive"), and the tests pin the answers, not the speed.
#pragma once
#include <hip/hip_runtime.h>

namespace kamd {

constexpr int SD_REF_TILE = 512;  // `const int batch=512` (sided_distance_cuda.cu:57)

template <typename V>
__device__ __forceinline__ V reseed_shfl(V v, int lane) { return __shfl(v, lane, 64); }

// does the winner `best_i` of query (qx, qy, qz) sit in a tile the reference ignores?  D(tx, ty, tz, qx, qy, qz) is the
// caller's distance expression, L loads a coordinate
template <typename Acc, typename In, typename D, typename L>
__device__ __forceinline__ bool sd_winner_in_dead_tile(const In* __restrict__ T, int best_i, Acc qx, Acc qy, Acc qz, D dist, L load) {
  if (best_i < SD_REF_TILE) return false;
  const size_t f = (size_t)(best_i & ~(SD_REF_TILE - 1)) * 3;
  const Acc d = dist(load(T + f), load(T + f + 1), load(T + f + 2), qx, qy, qz);
  return d != d;
}

// Every lane of the wavefront must call this (it contains cross-lane operations); lanes with `need` hold a query whose
// winner sits in a dead tile and return with the reference's (best, best_i).  T = the item's M targets in their ORIGINAL order.
template <typename Acc, typename In, typename D, typename L>
__device__ __forceinline__ void sd_reseed_fix(bool need, Acc qx, Acc qy, Acc qz, const In* T_lane, int M, Acc& best,
                                              int& best_i, D dist, L load) {
  const int lane = threadIdx.x & 63;
  for (unsigned long long m = __ballot(need); m != 0ull; m &= m - 1ull) {
    const int src = __ffsll((long long)m) - 1;
    const Acc x = reseed_shfl(qx, src), y = reseed_shfl(qy, src), z = reseed_shfl(qz, src);
    const In* __restrict__ T = (const In*)reseed_shfl((unsigned long long)T_lane, src);  // (lanes may hold different batch items)
    // tile 0 is live (its seed lost against the winner, so it was not NaN): every lane starts from target 0
    Acc b = dist(load(T), load(T + 1), load(T + 2), x, y, z);
    int bi = 0;
    for (int j0 = 0; j0 < M; j0 += 64) {
      const int jf = j0 & ~(SD_REF_TILE - 1);  // wave-uniform: 64 divides 512
      if (jf != 0 && j0 == jf) {
        const Acc df = dist(load(T + (size_t)jf * 3), load(T + (size_t)jf * 3 + 1), load(T + (size_t)jf * 3 + 2), x, y, z);
        if (df != df) {  // a dead tile: on to the next one
          j0 = jf + SD_REF_TILE - 64;
          continue;
        }
      }
      const int j = j0 + lane;
      if (j < M) {
        const Acc d = dist(load(T + (size_t)j * 3), load(T + (size_t)j * 3 + 1), load(T + (size_t)j * 3 + 2), x, y, z);
        if (d < b) {  // ascending j per lane: the lowest index of a lane's ties stays
          b = d;
          bi = j;
        }
      }
    }
#pragma unroll
    for (int sh = 32; sh >= 1; sh >>= 1) {
      const Acc od = __shfl_xor(b, sh, 64);
      const int oi = __shfl_xor(bi, sh, 64);
      if (od < b || (od == b && oi < bi)) {
        b = od;
        bi = oi;
      }
    }
    if (lane == src) {
      best = b;
      best_i = bi;
    }
  }
}

}  // namespace kamd
