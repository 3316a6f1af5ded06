// Screen-tile face lists shared by the DIB-R rasterizer and the soft-mask kernels (gfx950), second generation.
//
// The reference tests every pixel against every face (rasterization_cuda.cu:88-170, dibr_soft_mask_cuda.cu:80-172).
// A face only ever acts on a pixel whose centre lies inside the face's (possibly enlarged) bounding box, so the search
// is made sub-linear by binning the boxes into screen tiles.  The first generation (tile_bins.h) kept one BIT per
// (tile, face): a workspace of ntiles x F bits that had to be cleared and scanned on every call (51 MB at 8 views x
// 1024^2 x 50k faces, O(tiles x F) in general).  Here a tile owns a compact LIST of entries
//
//        entry = { block : the index of 64 consecutive faces of the packed face list,
//                  mask  : which of those 64 faces may touch the tile,
//                  sub   : (soft pass) which of the tile's 16 sub-tiles those faces reach }   (16 bytes)
//
// produced by one ballot of the wavefront that holds those 64 faces -- consecutive faces of a mesh are usually
// neighbours on screen, so an entry carries many faces, and inside an entry the ascending face order the reference's
// loops rely on is simply the bit order.  Lists are built in ONE pass over the faces: a tile owns INLINE slots for its
// first entries (at an address known from the tile index: a consumer reads count and entries in one round trip) and, past
// those, chunks of 31 entries taken from a pool and published in the tile's chunk table.  Nothing is sized by the data
// (shape-only workspace, no host sync): a tile that outgrows its table, or finds the pool empty, is flagged and falls
// back to scanning every block of its mesh (slow, still exact).  Faces whose box spans more than 8 x 8 tiles (or is NaN)
// go to a per-mesh "big face" list that every tile tests directly.
//
// Work per call: O(faces + entries), nothing proportional to tiles x faces, and the only memory that must be cleared is
// one counter and one small chunk table per tile.
#pragma once
#include "common.h"
#include "phase_prof.h"
#include "tile_bins.h"  // Box4 / Rec4 / pixel_x / pixel_y / FaceLayout / wave helpers

namespace kamd {
namespace tl {

constexpr int R_TILE = 16;            // rasterizer tile: 16 x 16 pixels = one 256-thread workgroup (4 sub-tiles of 16 x 4)
constexpr int S_TILE = 32;            // soft-mask tile: 32 x 32 pixels = 16 sub-tiles of 16 x 4 (work items of the search)
constexpr int S_SUBS = (S_TILE / SUB_W) * (S_TILE / SUB_H);  // 16
constexpr int REC_R = 20;             // raster record scalars: box[4] (empty for a filtered face) | a.xy b.xy | c.xy z.ab | A0 A1 B0 B1 | C0 C1 K z.c (edge_coefficients)
// soft record: large box[4] | body: a.xy b.xy c.xy pad2, then the three edges' reciprocals 1 / (|edge|^2 + EPS) as doubles
// (dibr_soft_mask_cuda.cu:128-139 divides by that sum for every (pixel, face) pair; the divisor depends on the face only).
// The boxes of all faces come first, as an array of their own (16 bytes per face), then the bodies: the select kernel
// streams thousands of boxes per work item and never looks at a body, the eval kernel the other way round.
constexpr int rec_s_scalars(int elem_size) { return elem_size == 4 ? 20 : 16; }
constexpr int rec_s_body_scalars(int elem_size) { return rec_s_scalars(elem_size) - 4; }
template <typename T>
__host__ __device__ inline const T* soft_box(const T* rec, size_t face) { return rec + face * 4; }
template <typename T>
__host__ __device__ inline const T* soft_body(const T* rec, size_t total_faces, size_t face) {
  return rec + total_faces * 4 + face * rec_s_body_scalars((int)sizeof(T));
}
// Conservative edge functions of one face, formed once by the binning kernel for the fp32 rasterizer's first sweep.
// The reference evaluates, per pixel p inside the face's box and in float,
//   w0 = (b - p) x (c - p), w1 = (c - p) x (a - p), w2 = (a - p) x (b - p), norm = w0 + w1 + w2 (+- eps)
// and drops the face when some w_i / norm < 0 (rasterization_cuda.cu:139-150).  In real arithmetic w_i is affine in p
// (w0 = b x c + px (by - cy) + py (cx - bx), ...) and norm is twice the signed area.  Each computed w_i differs from the
// real one by at most 4.0001 u S_i (u = 2^-24, S_0 = |bx - px||cy - py| + |by - py||cx - px|, ...: two subtractions, a
// product and the final subtraction, each within u), the computed norm by at most 6.1 u (S_0 + S_1 + S_2).  So when the
// area exceeds that bound for every pixel of the box, the sign s of norm is the area's at every pixel, and a pixel whose
// real s w_i lies below -(4.0001 u S_i + 1e-30) has a computed w_i of the opposite sign to norm and of magnitude > 1e-30:
// the quotient is a negative number (no underflow to -0 while |norm| < 1.3e6, eps <= 1) and the reference drops the face.
// The kernel tests, for edges 0 and 1,
//   e_i = fma(A_i, px, fma(B_i, py, C_i)) < 0,   A_i = fl(s alpha_i), B_i = fl(s beta_i), C_i = fl(s gamma_i + M_i)
// with the coefficients formed in double and M_i = 32 u (S_i,max + Q_i) + 1e-30, Q_i = |alpha_i| X + |beta_i| Y + |gamma_i|
// (X, Y = the box's largest |px|, |py|): 4.0001 u S_i for the reference's own rounding plus <= 3 u (Q_i + M_i) =: d_i for
// the three coefficient roundings and the two fma roundings -- an eightfold margin.  The third edge costs no coefficients:
// w0 + w1 + w2 is the area in real arithmetic, so s w2 = |area| - s w0 - s w1 and the kernel tests e0 + e1 > K with
// K = (|area| + M0 + M1 + M2 + 8 u (Q0 + Q1 + M0 + M1)) (1 + 4u) rounded to float: e0 + e1 <= |area| - s w2 + M0 + M1 +
// d0 + d1, the float sum adds u |e0 + e1| <= u (Q0 + Q1 + M0 + M1)(1 + ...), so e0 + e1 > K implies s w2 < -M2.
// No test drops a face the reference keeps; faces kept by mistake (pixels within ~1e-4 of an edge, relative to the face)
// fall to the exact sweep that follows, which repeats the reference's arithmetic.  Faces whose area is not safely away
// from zero, whose box is not finite (a NaN limit rejects no pixel) or whose coefficients are not, get A = B = 0, C = 1,
// K = 3: never dropped here.   out: e0.A e0.B e0.C e1.A e1.B e1.C K
__device__ __forceinline__ void edge_coefficients(const float* v, float x0, float y0, float x1, float y1, float* out) {
  const double ax = v[0], ay = v[1], bx = v[2], by = v[3], cx = v[4], cy = v[5];
  const double xlo = x0, xhi = x1, ylo = y0, yhi = y1;
  const double al[3] = {by - cy, cy - ay, ay - by};
  const double be[3] = {cx - bx, ax - cx, bx - ax};
  const double ga[3] = {bx * cy - by * cx, cx * ay - cy * ax, ax * by - ay * bx};
  const double area2 = ga[0] + ga[1] + ga[2];
  const double axm = fmax(fabs(ax - xlo), fabs(ax - xhi)), aym = fmax(fabs(ay - ylo), fabs(ay - yhi));
  const double bxm = fmax(fabs(bx - xlo), fabs(bx - xhi)), bym = fmax(fabs(by - ylo), fabs(by - yhi));
  const double cxm = fmax(fabs(cx - xlo), fabs(cx - xhi)), cym = fmax(fabs(cy - ylo), fabs(cy - yhi));
  const double S[3] = {bxm * cym + bym * cxm, cxm * aym + cym * axm, axm * bym + aym * bxm};
  const double X = fmax(fabs(xlo), fabs(xhi)), Y = fmax(fabs(ylo), fabs(yhi));
  const double u = 1.0 / 16777216.0;
  const double sgn = area2 < 0 ? -1.0 : 1.0;
  bool ok = fabs(area2) > 32.0 * u * (S[0] + S[1] + S[2]) && fabs(area2) < 1e6 && X < 1e30 && Y < 1e30;  // (NaN: false)
  double Q[3], M[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    Q[i] = fabs(al[i]) * X + fabs(be[i]) * Y + fabs(ga[i]);
    M[i] = 32.0 * u * (S[i] + Q[i]) + 1e-30;
  }
  const double K = (fabs(area2) + M[0] + M[1] + M[2] + 8.0 * u * (Q[0] + Q[1] + M[0] + M[1])) * (1.0 + 4.0 * u);
  ok = ok && K < 1e30;  // (bounds every coefficient)
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    out[i * 3 + 0] = ok ? (float)(sgn * al[i]) : 0.0f;
    out[i * 3 + 1] = ok ? (float)(sgn * be[i]) : 0.0f;
    out[i * 3 + 2] = ok ? (float)(sgn * ga[i] + M[i]) : 1.0f;
  }
  out[6] = ok ? (float)K : 3.0f;
}
constexpr double SOFT_EPS = 1e-7;     // the reference's literal EPS (dibr_soft_mask_cuda.cu:23), a double
// Append counters live one per 128-byte line (COUNTER_STRIDE words apart): device-scope atomics on one line complete one at a
// time (measured on MI355X: ~33 ns each on one address, ~15 ns each on neighbouring words of a line), so counters that are hit
// thousands of times per launch must not share lines.
constexpr int COUNTER_STRIDE = 32;
#ifndef KAMD_MEDIUM_TILES
#define KAMD_MEDIUM_TILES 4
#endif
constexpr int MEDIUM_TILES = KAMD_MEDIUM_TILES;  // a face whose tile rectangle holds more tiles than this is binned on its own (wave_bin): single-face entries, no merging
#ifndef KAMD_BIG_TILES_R
#define KAMD_BIG_TILES_R 16
#endif
constexpr int BIG_TILES_R = KAMD_BIG_TILES_R;   // rasterizer pass: a face whose rectangle holds more tiles goes to its view's big list (as do rectangles beyond 8 x 8)
constexpr int SPAN_SAMPLE = 16;       // every SPAN_SAMPLE-th workgroup of the binning launch reports the tile rows its faces cover (note_row_span)
constexpr int WORK_SHARDS = 8;        // worklist shards (one append counter each; workgroup id & 7 picks the shard)

struct PassGeom {
  int tile, tiles_x, tiles_y, ntiles;
};
__host__ __device__ inline PassGeom pass_geom(int H, int W, int tile) {
  PassGeom g;
  g.tile = tile;
  g.tiles_x = (W + tile - 1) / tile;
  g.tiles_y = (H + tile - 1) / tile;
  g.ntiles = g.tiles_x * g.tiles_y;
  return g;
}

constexpr int OVC = 32;               // pool chunk: header {next free, -, -, -} is unused; slots 1..31 hold entries
constexpr int OVC_PAYLOAD = OVC - 1;
constexpr unsigned int BRUTE_BIT = 0x80000000u;  // set in a tile's counter when an entry could not be stored

// device view of one pass' lists
struct Lists {
  unsigned int* count;        // [B * ntiles]         zeroed; entries appended to the tile (| BRUTE_BIT)
  unsigned int* tab;          // [B * ntiles * maxc]  zeroed; chunk c of the tile lives at pool chunk tab[..] - 1
  uint4* inl;                 // [B * ntiles * C]     the tile's first C entries {block, sub-tile bits, mask.lo, mask.hi}
  uint4* pool;                // [cap_chunks * OVC]
  unsigned int* pool_top;     // zeroed
  unsigned int cap_chunks;
  int C, maxc;
  unsigned int* big_count;    // [B]               zeroed; elements of mesh b's big list
  unsigned int* big_list;     // [3 * total_faces]  3 words per element {64-face block, mask.lo, mask.hi}, mesh b's segment at 3 x its first packed face
  unsigned int* big_hash;     // soft pass of the fused operator (nullptr: none): work + WORK_BIGHASH_WORD, where big faces are entered
  // raster pass only (0: none): big_rows_off 32-bit words behind big_count, zeroed: per view and tile row ceil(tiles_x / 64) 64-bit
  // words, bit = some big face's tile rectangle holds the tile.  A view with ONE big face (a floor under the object) otherwise
  // costs every tile of the view its background fast path: the big list is a candidate of all of them.
  unsigned int big_rows_off;
  unsigned int* sub_touched;  // [B * ntiles]      zeroed; soft pass only: bit s = some enlarged box reaches sub-tile s
  int tiles_x, ntiles;
  // raster pass only (nullptr: none): the tile rows the mesh's boxes cover, per view -- where the tile kernels start
  unsigned int* row_span;     // [B * 2] zeroed; view b: (last covered row + 1), (tiles_y - first covered row); 0 = none
};

inline size_t a256(size_t x) { return (x + 255) & ~(size_t)255; }
__host__ __device__ inline int big_row_words(int tiles_x) { return (tiles_x + 63) >> 6; }
inline unsigned int pool_chunks(long long total_faces, long long n_tiles_total) {
  // entries the pool can take beyond the inline slots.  A face adds at most one entry per tile of its rectangle (fewer when its
  // wavefront's faces share tiles), so this is 4 tiles' worth per face: past it the appends flag their tiles BRUTE, which is
  // correct but slow -- a mesh whose face ORDER has no locality (49 210 faces shuffled: every entry holds one face) ran the
  // rasterizer's tile kernel at 6.1 ms instead of 0.15 with the pool of rounds 2-3 (faces / 2).  16 bytes each, never cleared.
  long long e = total_faces * 4 + 2 * n_tiles_total + 64;
  long long c = e / OVC_PAYLOAD + 64;
  if (c > 0x3fffffll) c = 0x3fffffll;
  return (unsigned int)c;
}

// Host-side layout of one pass inside a workspace.  All zeroed arrays of both passes are placed in ONE contiguous region
// at the start of the workspace so that a single fill kernel clears them.
struct PassLayout {
  size_t count, tab, pool_top, big_count, big_rows, sub_touched, row_span;  // inside the zero region
  size_t inl, pool, big_list, rec;                      // after it
  size_t pixcnt, prob_pm;                // soft pass: hits per (item, pixel) (u16); pixel-major probabilities (knum > 128 only)
  size_t over_count, over_list;          // soft pass: the work places soft_select hands to its rounds launch (count zeroed; soft2.inc)
  size_t ord_count, ord_list;            // soft pass: the items dealt by pair count for soft_eval (counts zeroed, a line each)
  unsigned int cap_chunks;
  int C, maxc;
  PassGeom g;
};
struct Layout {
  size_t zero_bytes;      // [0, zero_bytes) is cleared on every call
  PassLayout r, s;        // raster / soft (either may be absent: offsets 0)
  size_t total;
};
inline unsigned int work_shard_cap(int B, int H, int W);  // (the worklist's capacity per shard, below)
inline Layout make_layout(int B, int H, int W, long long total_faces, int esz, bool with_r, bool with_s, int K = 0) {
  Layout L{};
  size_t off = 0;
  L.r.g = pass_geom(H, W, R_TILE);
  L.s.g = pass_geom(H, W, S_TILE);
  L.r.C = 16; L.r.maxc = 16;    // 16 + 16 * 31 = 512 entries per 16 x 16 tile before the fallback
  L.s.C = 64; L.s.maxc = 32;    // 64 + 32 * 31 = 1056 entries per 32 x 32 tile
  const size_t ntr = (size_t)B * L.r.g.ntiles, nts = (size_t)B * L.s.g.ntiles;
  if (with_r) {
    L.r.count = off; off += a256(ntr * 4);
    L.r.tab = off; off += a256(ntr * L.r.maxc * 4);
    L.r.pool_top = off; off += 256;
    L.r.big_count = off; off += a256((size_t)B * 4);
    L.r.big_rows = off; off += a256((size_t)B * L.r.g.tiles_y * big_row_words(L.r.g.tiles_x) * 8);
    L.r.row_span = off; off += a256((size_t)B * 2 * 4);
  }
  if (with_s) {
    L.s.count = off; off += a256(nts * 4);
    L.s.tab = off; off += a256(nts * L.s.maxc * 4);
    L.s.pool_top = off; off += 256;
    L.s.big_count = off; off += a256((size_t)B * 4);
    L.s.sub_touched = off; off += a256(nts * 4);
    L.s.over_count = off; off += 256;
    L.s.ord_count = off; off += 4 * COUNTER_STRIDE * 4;  // (S2_CLASSES counters, a 128-byte line each)
  }
  L.zero_bytes = off;
  if (with_r) {
    L.r.cap_chunks = pool_chunks(total_faces, (long long)ntr);
    L.r.inl = off; off += a256(ntr * L.r.C * 16);
    L.r.pool = off; off += a256((size_t)L.r.cap_chunks * OVC * 16);
    L.r.big_list = off; off += a256((size_t)total_faces * 3 * 4);  // entries {block, mask.lo, mask.hi}: at most one per face (wave_bin)
    L.r.rec = off; off += a256((size_t)total_faces * REC_R * esz);
  }
  if (with_s) {
    L.s.cap_chunks = pool_chunks(total_faces, (long long)nts);
    L.s.inl = off; off += a256(nts * L.s.C * 16);
    L.s.pool = off; off += a256((size_t)L.s.cap_chunks * OVC * 16);
    L.s.big_list = off; off += a256((size_t)total_faces * 3 * 4);  // entries {block, mask.lo, mask.hi}: at most one per face (wave_bin)
    L.s.rec = off; off += a256((size_t)total_faces * rec_s_scalars(esz) * esz);
    const size_t item_pixels = nts * S_SUBS * 64;
    L.s.pixcnt = off; off += a256(item_pixels * 2);
    L.s.over_list = off; off += a256((size_t)WORK_SHARDS * work_shard_cap(B, H, W) * 4);
    L.s.ord_list = off; off += a256((size_t)4 * WORK_SHARDS * work_shard_cap(B, H, W) * 16);  // (S2_CLASSES lists of work records)
    L.s.prob_pm = off; off += K > 128 ? a256(item_pixels * (size_t)K * esz) : 0;
  }
  L.total = off + 256;
  return L;
}
// the worklist lives in its own buffer (the autograd path keeps it for the backward pass): WORK_HEADER words of counters,
// then WORK_SHARDS x shard_cap items of 16 bytes.  One 256-thread workgroup per 16 x 16 pixels appends at most 4 items.
constexpr int WORK_FLAT_WORD = 8 * COUNTER_STRIDE;                    // the flat hit list's shard counters (soft2.inc) follow the worklist's
constexpr int COV_SHARDS = 32;                                        // shards of the list of tiles that hold a covered pixel
constexpr int WORK_COV_WORD = WORK_FLAT_WORD + 64 * COUNTER_STRIDE;   // ... whose counters follow the flat list's
// ... then the hot faces of the soft mask's backward pass.  A face of the soft pass' big list (an enlarged box of more than 8 x 8 soft
// tiles: the knot scene's ~170 image-sized faces per view) collects a gradient term from every soft pixel it reaches -- thousands of
// atomic requests on ONE 24-byte record, and requests on one line are performed one at a time, ~40 ns each (the knot: soft backward
// 3.0x the sphere's time for 2.0x the hits).  The binning launch enters such faces into a hash table (key = packed face index,
// open addressing, at most BIG_HASH_MAX keys: beyond that the table is not used), the fused backward adds their terms to the
// copy of the workgroup's XCD (BIG_SIDE_COPIES records per face, a line apart from each other) and the rasterizer's backward
// launch, which runs next, folds the copies into the gradient and clears them.
constexpr int BIG_HASH_SLOTS = 4096;
constexpr int BIG_HASH_MAX = 2048;
constexpr int BIG_SIDE_COPIES = 8;
constexpr int BIG_HASH_BITS_OFF = 32;                                           // behind the count: one bit per slot, set when the slot is taken
constexpr int BIG_HASH_TAGS_OFF = BIG_HASH_BITS_OFF + BIG_HASH_SLOTS / 32;      // then the slots' tags (key + 1; 0: free)
constexpr int WORK_BIGHASH_WORD = WORK_COV_WORD + COV_SHARDS * COUNTER_STRIDE;  // [0] keys entered, bits, tags
constexpr int WORK_BIGSIDE_WORD = WORK_BIGHASH_WORD + BIG_HASH_TAGS_OFF + BIG_HASH_SLOTS;  // BIG_SIDE_COPIES x BIG_HASH_SLOTS x 8 floats (6 used)
constexpr int WORK_HEADER = WORK_BIGSIDE_WORD + BIG_SIDE_COPIES * BIG_HASH_SLOTS * 8;  // words; all zeroed per call
__device__ __forceinline__ unsigned int big_hash_of(unsigned int key) { return (key * 2654435761u) >> 20; }  // 12 bits
static_assert(BIG_HASH_SLOTS == 4096, "big_hash_of yields 12 bits");
// bh = work + WORK_BIGHASH_WORD
__device__ __forceinline__ void big_hash_insert(unsigned int* bh, unsigned int key) {
  if (atomicAdd(bh, 1u) >= (unsigned int)BIG_HASH_MAX) return;  // (readers find the count above the limit and leave the table alone)
  unsigned int h = big_hash_of(key);
  for (int probe = 0; probe < BIG_HASH_SLOTS; ++probe) {
    const unsigned int old = atomicCAS(bh + BIG_HASH_TAGS_OFF + h, 0u, key + 1u);
    if (old == 0u) atomicOr(bh + BIG_HASH_BITS_OFF + (h >> 5), 1u << (h & 31u));
    if (old == 0u || old == key + 1u) return;
    h = (h + 1u) & (unsigned int)(BIG_HASH_SLOTS - 1);
  }
}
__device__ __forceinline__ bool big_hash_usable(unsigned int n) { return n > 0u && n <= (unsigned int)BIG_HASH_MAX; }
__device__ __forceinline__ int big_hash_find(const unsigned int* bh, unsigned int key) {  // -> slot, or -1
  unsigned int h = big_hash_of(key);
  for (int probe = 0; probe < BIG_HASH_SLOTS; ++probe) {
    const unsigned int t = bh[BIG_HASH_TAGS_OFF + h];
    if (t == key + 1u) return (int)h;
    if (t == 0u) return -1;
    h = (h + 1u) & (unsigned int)(BIG_HASH_SLOTS - 1);
  }
  return -1;
}
// A free word of the header's first line: the fused forward (its last launch) leaves a signature of the layout here, and the
// fused backward's covered-tile walk trusts the list only when it finds it -- a work buffer that did not come from
// kamd_dibr_rasterization_forward_* of THIS build and shape (another operator's, a tool's, stale memory) makes it visit every
// tile instead of reading garbage as tile indices.
constexpr int WORK_MAGIC_WORD = 1;
__host__ __device__ inline unsigned int work_magic(int B, int H, int W) {
  return 0xD1B40004u ^ ((unsigned int)B * 2654435761u) ^ ((unsigned int)H * 40503u) ^ ((unsigned int)W << 16);
}
inline unsigned int work_shard_cap(int B, int H, int W) {
  const PassGeom g = pass_geom(H, W, R_TILE);
  const size_t n_groups = (size_t)B * g.ntiles;
  return (unsigned int)(4 * ((n_groups + WORK_SHARDS - 1) / WORK_SHARDS));
}
// ... then one byte per (mesh, 16 x 16 tile) [b * ntiles + tile]: does the tile hold a covered pixel?  (Written by the
// rasterizer's tile kernel in the fused path; the rasterizer's backward kernel leaves a tile without one at once.)
inline size_t work_cov_offset_words(int B, int H, int W) { return WORK_HEADER + (size_t)WORK_SHARDS * work_shard_cap(B, H, W) * 4; }
// ... then a copy of the forward's covered-row spans (Lists::row_span, 2 B words): the rasterizer's backward kernel starts from
// the same rows as the forward's tile kernel
inline size_t work_span_offset_words(int B, int H, int W) {
  return work_cov_offset_words(B, H, W) + ((size_t)B * pass_geom(H, W, R_TILE).ntiles + 3) / 4;
}
// ... then the list of the tiles that hold a covered pixel ([b * ntiles + tile], COV_SHARDS shards of equal capacity, appended
// by the rasterizer's tile kernel in dispatch order): the rasterizer's backward pass walks it with a persistent grid instead
// of launching one workgroup per tile of the image to find out that 85 % of them have nothing to do
// Shard of tile `tile_order` (its place in the tile kernel's order of tiles) of view b: 8 groups x 4.  With 8 or more views the
// group is the view's (b % 8): the backward's workgroups of XCD x take group x, so that a view's gradient lines are updated
// through one XCD's L2 (as in the forward, whose consecutive workgroups are the views of one tile); with fewer views the
// groups just spread the tiles.  Capacity: no shard receives more than ceil(B / 8) views' quarter of the tiles.
static_assert(COV_SHARDS == 32, "8 groups x 4");
__host__ __device__ inline unsigned int cov_shard_of(int B, int b, unsigned int tile_order) {
  const unsigned int order = tile_order * (unsigned int)B + (unsigned int)b;
  return B >= 8 ? (((unsigned int)b & 7u) << 2) | (tile_order & 3u) : ((order & 7u) << 2) | ((order >> 3) & 3u);
}
__host__ __device__ inline unsigned int cov_shard_cap(size_t B, size_t ntiles) { return (unsigned int)(((B + 7) / 8) * ((ntiles + 3) / 4)); }
__host__ __device__ inline size_t cov_list_words_after_cov(size_t B, size_t n_groups) { return (n_groups + 3) / 4 + 2 * B; }
inline size_t work_covlist_offset_words(int B, int H, int W) {
  return work_cov_offset_words(B, H, W) + cov_list_words_after_cov((size_t)B, (size_t)B * pass_geom(H, W, R_TILE).ntiles);
}
inline size_t work_words(int B, int H, int W) {
  return work_covlist_offset_words(B, H, W) + (size_t)COV_SHARDS * cov_shard_cap((size_t)B, (size_t)pass_geom(H, W, R_TILE).ntiles);
}

inline Lists lists_of(void* ws, const PassLayout& p, int B, bool soft) {
  char* c = (char*)ws;
  Lists l;
  l.count = (unsigned int*)(c + p.count);
  l.tab = (unsigned int*)(c + p.tab);
  l.inl = (uint4*)(c + p.inl);
  l.pool = (uint4*)(c + p.pool);
  l.pool_top = (unsigned int*)(c + p.pool_top);
  l.cap_chunks = p.cap_chunks;
  l.C = p.C;
  l.maxc = p.maxc;
  l.big_count = (unsigned int*)(c + p.big_count);
  l.big_list = (unsigned int*)(c + p.big_list);
  l.big_hash = nullptr;
  l.big_rows_off = (!soft && p.big_rows != 0) ? (unsigned int)((p.big_rows - p.big_count) / 4) : 0u;
  l.sub_touched = soft ? (unsigned int*)(c + p.sub_touched) : nullptr;
  l.tiles_x = p.g.tiles_x;
  l.ntiles = p.g.ntiles;
  l.row_span = (!soft && p.row_span != 0) ? (unsigned int*)(c + p.row_span) : nullptr;
  return l;
}

// ---- conservative pixel range of a half-open box -----------------------------------------------------------------------
// col(x) = (x*W/mult + W - 1)/2 increasing in x, row(y) = (H - 1 - y*H/mult)/2 decreasing in y; +-1 pixel of slack covers
// the rounding of the float pixel-centre expressions.  Returns false when the box misses the image.  NaN limits (or a
// non-positive multiplier) make the box "everywhere", as in the reference where NaN comparisons never reject.
struct PixRange {
  int c_lo, c_hi, r_lo, r_hi;
  bool everywhere;
};
template <typename T>
__device__ __forceinline__ bool pixel_range(T xmin, T ymin, T xmax, T ymax, int H, int W, float multiplier, PixRange* out) {
  out->c_lo = 0;
  out->c_hi = W - 1;
  out->r_lo = 0;
  out->r_hi = H - 1;
  // float is enough: coordinates are O(multiplier), so the column / row estimates are off by ~1e-4 pixel at most, against
  // one whole pixel of slack on either side (values beyond float range become +-inf and clamp)
  const float fxmin = (float)xmin, fxmax = (float)xmax, fymin = (float)ymin, fymax = (float)ymax;
  out->everywhere = !(multiplier > 0.f) || !(fxmin == fxmin) || !(fxmax == fxmax) || !(fymin == fymin) || !(fymax == fymax);
  if (out->everywhere) return true;
  const float sx = (float)W / multiplier, sy = (float)H / multiplier;
#ifdef KAMD_PIXEL_RANGE_SLACK_ONLY
  // (rounds 2-4, A/B builds: a whole pixel of slack on either side)
  const float cl = floorf((fxmin * sx + (float)(W - 1)) * 0.5f) - 1.0f, ch = ceilf((fxmax * sx + (float)(W - 1)) * 0.5f) + 1.0f;
  const float rl = floorf(((float)(H - 1) - fymax * sy) * 0.5f) - 1.0f, rh = ceilf(((float)(H - 1) - fymin * sy) * 0.5f) + 1.0f;
#else
  // Round 5: tight.  A pixel centre passes the kernels' test `x >= xmin && x < xmax` exactly when its column lies in
  // [u(xmin), u(xmax)) with u(x) = (x W / multiplier + W - 1) / 2 the column as a real number: the columns are ceil(u(xmin)) ...
  // ceil(u(xmax)) - 1.  The float evaluation of u and of the kernels' centres (pixel_x: fl(multiplier / W) times an exact integer)
  // is off by ~1e-4 of a pixel at 1024 columns, so the two ends move OUTWARDS by d = 1e-3 + 1e-6 W (rows likewise): at most one
  // column too many, once in a few hundred boxes.  The whole pixel of slack on either side that rounds 2-4 used here binned a face
  // six pixels across as if it were nine: 437 k (tile, face) pairs at C4 where the boxes hold pixel centres in 301 k, and every
  // sub-tile bit -- hence every work item of the soft mask -- as generous.  (Shrinking the generous range with the kernels' own
  // test, four short loops per face, is exact but cost the binning launch 4 us: profiles/r05u_*.)
  const float dx = 1e-3f + 1e-6f * (float)W, dy = 1e-3f + 1e-6f * (float)H;
  const float cl = ceilf((fxmin * sx + (float)(W - 1)) * 0.5f - dx), ch = ceilf((fxmax * sx + (float)(W - 1)) * 0.5f + dx) - 1.0f;
  const float rl = ceilf(((float)(H - 1) - fymax * sy) * 0.5f - dy) , rh = ceilf(((float)(H - 1) - fymin * sy) * 0.5f + dy) - 1.0f;
#endif
  if (ch < 0.0f || cl > (float)(W - 1) || rh < 0.0f || rl > (float)(H - 1)) return false;
  out->c_lo = (int)fmaxf(cl, 0.0f);
  out->c_hi = (int)fminf(ch, (float)(W - 1));
  out->r_lo = (int)fmaxf(rl, 0.0f);
  out->r_hi = (int)fminf(rh, (float)(H - 1));
  // an INVERTED box (min > max: the contract operators take the caller's boxes as they come, and a negative boxlen inverts the
  // enlarged ones) holds no pixel centre -- x >= xmin and x < xmax cannot both hold -- and with more than the two pixels of slack
  // its range comes out empty: it must not reach the tile arithmetic, whose rectangles assume lo <= hi (a rectangle of negative
  // width made the binning kernel's tile loop spin for ever: found by tools/round4/fuzz_rasterize_ops.py)
  if (out->c_lo > out->c_hi || out->r_lo > out->r_hi) return false;
  return true;
}

// torch.min / torch.max semantics (NaN propagates), as the reference's Python glue computes the boxes (rasterization.py:325-327)
template <typename T>
__device__ __forceinline__ T nan_min(T a, T b) { return (a != a || b != b) ? (T)NAN : (a < b ? a : b); }
template <typename T>
__device__ __forceinline__ T nan_max(T a, T b) { return (a != a || b != b) ? (T)NAN : (a > b ? a : b); }

// OR over the wavefront (result wave-uniform): row shifts inside the four rows of 16 lanes (DPP: register moves, no LDS
// crossbar), then the last lane of every row
__device__ __forceinline__ unsigned int wave_or_u32(unsigned int v) {
  int x = (int)v;
  x |= row_shr<1>(x);
  x |= row_shr<2>(x);
  x |= row_shr<4>(x);
  x |= row_shr<8>(x);
  return (unsigned int)(__builtin_amdgcn_readlane(x, 15) | __builtin_amdgcn_readlane(x, 31) | __builtin_amdgcn_readlane(x, 47) |
                        __builtin_amdgcn_readlane(x, 63));
}

// ---- appending one entry to a tile -----------------------------------------------------------------------------------------
// slot = count++; the first C slots are inline.  Past them, slot o = slot - C lives in chunk c = o / 31 at position o % 31;
// the lane that draws position 0 takes a chunk from the pool and publishes it in the tile's table, the others wait for
// the table entry.  Allocation never waits for anything (it happens before any spinning of the same wavefront), so every
// awaited publication is made by a lane that already holds its slot and is on its way: no cycle of waits.
__device__ __forceinline__ void append_entry(bool on, const Lists& L, size_t ti, uint4 entry) {
  // (called by whole wavefronts, `on` = the lane has an entry: the three steps below are separate, reconverging blocks, so
  // that every allocating lane of a wavefront has published before any of its lanes starts to wait)
  unsigned int slot = 0, c = 0, i = 0, v = 0;
  bool pooled = false;
  if (on) {
    slot = atomicAdd(L.count + ti, 1u) & ~BRUTE_BIT;
    if (slot < (unsigned int)L.C) {
      L.inl[ti * L.C + slot] = entry;
    } else {
      const unsigned int o = slot - (unsigned int)L.C;
      c = o / OVC_PAYLOAD;
      i = o - c * OVC_PAYLOAD;
      if (c >= (unsigned int)L.maxc)
        atomicOr(L.count + ti, BRUTE_BIT);
      else
        pooled = true;
    }
  }
  unsigned int* link = L.tab + ti * L.maxc + c;
  if (pooled && i == 0) {  // step 1: allocate and publish (never waits)
    const unsigned int p = atomicAdd(L.pool_top, 1u);
    v = p < L.cap_chunks ? p + 1u : 0xFFFFFFFFu;
    __hip_atomic_store(link, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (pooled && i != 0) {  // step 2: wait for the chunk's allocator (another wavefront, or an earlier append of this one)
    do {
      v = __hip_atomic_load(link, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (v == 0u) __builtin_amdgcn_s_sleep(1);
    } while (v == 0u);
  }
  if (pooled) {            // step 3
    if (v == 0xFFFFFFFFu)
      atomicOr(L.count + ti, BRUTE_BIT);
    else
      L.pool[(size_t)(v - 1u) * OVC + 1u + i] = entry;
  }
}

// The same for one entry of each of two lists (the rasterizer's and the soft mask's): each step is taken for both before
// the next one, so that the two counter atomics (and the two pool atomics) are in flight together -- the binning kernel
// is bound by these dependent round trips, not by instructions.  The no-cycle argument above holds step by step.
struct PendingEntry {
  bool on;
  size_t ti;
  uint4 entry;
};
__device__ __forceinline__ void append_entry_pair(const PendingEntry& pa, const Lists& La, const PendingEntry& pb, const Lists& Lb) {
  const PendingEntry* pe[2] = {&pa, &pb};
  const Lists* Ls[2] = {&La, &Lb};
  unsigned int slot[2] = {0, 0}, c[2] = {0, 0}, i[2] = {0, 0}, v[2] = {0, 0};
  bool pooled[2] = {false, false};
  // (the returned values are not touched before both atomics are issued: the first use of a result is what makes the wavefront
  // wait for it -- masking the flag bit in the same statement serialised the two round trips)
#pragma unroll
  for (int q = 0; q < 2; ++q)
    if (pe[q]->on) slot[q] = atomicAdd(Ls[q]->count + pe[q]->ti, 1u);
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const Lists& L = *Ls[q];
    slot[q] &= ~BRUTE_BIT;
    if (pe[q]->on) {
      if (slot[q] < (unsigned int)L.C) {
        L.inl[pe[q]->ti * L.C + slot[q]] = pe[q]->entry;
      } else {
        const unsigned int o = slot[q] - (unsigned int)L.C;
        c[q] = o / OVC_PAYLOAD;
        i[q] = o - c[q] * OVC_PAYLOAD;
        if (c[q] >= (unsigned int)L.maxc)
          atomicOr(L.count + pe[q]->ti, BRUTE_BIT);
        else
          pooled[q] = true;
      }
    }
  }
  unsigned int p[2] = {0, 0};
#pragma unroll
  for (int q = 0; q < 2; ++q)
    if (pooled[q] && i[q] == 0) p[q] = atomicAdd(Ls[q]->pool_top, 1u);
#pragma unroll
  for (int q = 0; q < 2; ++q)
    if (pooled[q] && i[q] == 0) {
      v[q] = p[q] < Ls[q]->cap_chunks ? p[q] + 1u : 0xFFFFFFFFu;
      __hip_atomic_store(Ls[q]->tab + pe[q]->ti * Ls[q]->maxc + c[q], v[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#pragma unroll
  for (int q = 0; q < 2; ++q)
    if (pooled[q] && i[q] != 0) {
      unsigned int* link = Ls[q]->tab + pe[q]->ti * Ls[q]->maxc + c[q];
      do {
        v[q] = __hip_atomic_load(link, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v[q] == 0u) __builtin_amdgcn_s_sleep(1);
      } while (v[q] == 0u);
    }
#pragma unroll
  for (int q = 0; q < 2; ++q)
    if (pooled[q]) {
      if (v[q] == 0xFFFFFFFFu)
        atomicOr(Ls[q]->count + pe[q]->ti, BRUTE_BIT);
      else
        Ls[q]->pool[(size_t)(v[q] - 1u) * OVC + 1u + i[q]] = pe[q]->entry;
    }
}

// The lane's own face (bit `own` of block `block`) appended as a single-face entry to the next NQ tiles of its rectangle, walked
// row by row from (cx, cy) -- the medium faces of wave_bin.  A returning device-scope atomic is a ~2 us round trip here (it is
// performed memory-side, beyond the XCD's L2) and nothing else in the kernel takes long, so the NQ counter atomics are issued
// back to back -- a result's FIRST USE is what makes the wavefront wait, so none is touched before all are out -- and the entries
// are written in a second walk over the same tiles (recomputing a tile index is cheaper than keeping it: NQ registers hold the
// slots and nothing else; the fused binning kernel has to stay at 7 wavefronts per SIMD, the whole launch is resident at once).
// The pool steps keep append_entry's order: every allocation of the wavefront is published before any of its lanes waits.
struct RectWalk {
  int x, y;
  __device__ __forceinline__ void step(int tx0, int tx1) {
    if (++x > tx1) {
      x = tx0;
      ++y;
    }
  }
};
template <bool SOFT>
__device__ __forceinline__ unsigned int sub_tiles_reached(int tx, int ty, int c_lo, int c_hi, int r_lo, int r_hi) {
  if (!SOFT) return 0u;
  // the 16 x 4-pixel sub-tiles of tile (tx, ty) the pixel range reaches: bit = sy * 2 + sx
  const int px0 = tx * S_TILE, py0 = ty * S_TILE;
  const int sx0 = max(c_lo - px0, 0) / SUB_W, sx1 = min(c_hi - px0, S_TILE - 1) / SUB_W;
  const int sy0 = max(r_lo - py0, 0) / SUB_H, sy1 = min(r_hi - py0, S_TILE - 1) / SUB_H;
  const unsigned int colbits = (sx0 == 0 ? 1u : 0u) | (sx1 >= 1 ? 2u : 0u);
  const unsigned int rowsel = ((1u << (2 * (sy1 + 1))) - 1u) & ~((1u << (2 * sy0)) - 1u) & 0x5555u;
  return colbits * rowsel;
}
#ifndef KAMD_MEDIUM_BATCH
#define KAMD_MEDIUM_BATCH 8  // tiles a medium face appends per step (their counter atomics in flight together): 4 / 8 / 12 / 16 = 68 / 68 / 77 / 85 VGPRs, same time
#endif
constexpr int MEDIUM_BATCH_MAX = KAMD_MEDIUM_BATCH;
__device__ __forceinline__ unsigned int* medium_slot_scratch() {
  __shared__ unsigned int s[MEDIUM_BATCH_MAX * 256];  // (the binning kernels run 256 threads per workgroup)
  return s;
}
template <bool SOFT, int NQ>
__device__ __forceinline__ void append_own(int n_on, RectWalk& at, int tx0, int tx1, unsigned int tile_base, unsigned int block,
                                           unsigned long long own, int c_lo, int c_hi, int r_lo, int r_hi, const Lists& L) {
  static_assert(NQ <= MEDIUM_BATCH_MAX, "medium_slot_scratch");
  unsigned int slot[NQ];
  const RectWalk start = at;
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    slot[q] = 0u;
    if (q < n_on) {
      const unsigned int ti = tile_base + (unsigned int)(at.y * L.tiles_x + at.x);
      const unsigned int sub = sub_tiles_reached<SOFT>(at.x, at.y, c_lo, c_hi, r_lo, r_hi);
      if (SOFT && sub != 0u) atomicOr(L.sub_touched + ti, sub);
      slot[q] = atomicAdd(L.count + ti, 1u);
    }
    at.step(tx0, tx1);
  }
  bool any_pooled = false;
  RectWalk w = start;
  // (the compiler must not see that this walk repeats the first one: it would keep all NQ tile indices and sub-tile masks alive
  // instead of recomputing them -- 137 registers)
  asm volatile("" : "+v"(w.x), "+v"(w.y) : "v"(slot[NQ - 1]));  // (... and not before the results are in: nothing of it is needed earlier)
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    slot[q] &= ~BRUTE_BIT;
    if (q < n_on) {
      const unsigned int ti = tile_base + (unsigned int)(w.y * L.tiles_x + w.x);
      if (slot[q] < (unsigned int)L.C) {
        const unsigned int sub = sub_tiles_reached<SOFT>(w.x, w.y, c_lo, c_hi, r_lo, r_hi);
        L.inl[(size_t)ti * L.C + slot[q]] = make_uint4(block, sub, (unsigned int)own, (unsigned int)(own >> 32));
      } else if ((slot[q] - (unsigned int)L.C) / OVC_PAYLOAD >= (unsigned int)L.maxc) {
        atomicOr(L.count + ti, BRUTE_BIT);
      } else {
        any_pooled = true;
      }
    }
    w.step(tx0, tx1);
  }
  if (__ballot(any_pooled) == 0ull) return;  // (uniform)
  // Past a tile's inline slots (rare).  Rolled loops over slots parked in LDS: unrolled, their NQ x (slot, chunk, position)
  // took the kernel from 68 to 137 registers.
  unsigned int* parked = medium_slot_scratch() + threadIdx.x;
#pragma unroll
  for (int q = 0; q < NQ; ++q) parked[q * 256] = q < n_on ? slot[q] : 0u;
  const int n = min(n_on, NQ);
  w = start;
#pragma unroll 1
  for (int q = 0; q < n; ++q) {  // step 1: allocate and publish (never waits)
    const unsigned int sl = parked[q * 256];
    const unsigned int o = sl - (unsigned int)L.C, c = o / OVC_PAYLOAD;
    if (sl >= (unsigned int)L.C && c < (unsigned int)L.maxc && o - c * OVC_PAYLOAD == 0u) {
      const unsigned int ti = tile_base + (unsigned int)(w.y * L.tiles_x + w.x);
      const unsigned int p = atomicAdd(L.pool_top, 1u);
      __hip_atomic_store(L.tab + (size_t)ti * L.maxc + c, p < L.cap_chunks ? p + 1u : 0xFFFFFFFFu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    w.step(tx0, tx1);
  }
  w = start;
#pragma unroll 1
  for (int q = 0; q < n; ++q) {  // steps 2 and 3: the chunk (one's own publication is read back), then the entry
    const unsigned int sl = parked[q * 256];
    const unsigned int o = sl - (unsigned int)L.C, c = o / OVC_PAYLOAD;
    if (sl >= (unsigned int)L.C && c < (unsigned int)L.maxc) {
      const unsigned int ti = tile_base + (unsigned int)(w.y * L.tiles_x + w.x);
      unsigned int* link = L.tab + (size_t)ti * L.maxc + c;
      unsigned int v;
      do {
        v = __hip_atomic_load(link, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v == 0u) __builtin_amdgcn_s_sleep(1);
      } while (v == 0u);
      if (v == 0xFFFFFFFFu) {
        atomicOr(L.count + ti, BRUTE_BIT);
      } else {
        const unsigned int sub = sub_tiles_reached<SOFT>(w.x, w.y, c_lo, c_hi, r_lo, r_hi);
        L.pool[(size_t)(v - 1u) * OVC + 1u + (o - c * OVC_PAYLOAD)] = make_uint4(block, sub, (unsigned int)own, (unsigned int)(own >> 32));
      }
    }
    w.step(tx0, tx1);
  }
}

// ---- one wavefront bins its 64 faces into one pass' lists ---------------------------------------------------------------
// `active`: the lane's face takes part; (b, first_b): its mesh and the mesh's first packed face; tile rectangle
// [tx0,tx1] x [ty0,ty1]; `big`: the rectangle exceeds 8 x 8 tiles (or the box is NaN).  `block` = packed face index >> 6
// (the same for the whole wavefront).
// Distinct tiles are enumerated without walking the union rectangle: every lane keeps the not-yet-emitted tiles of its
// own rectangle as a 64-bit mask over an 8 x 8 local grid; each step takes the first pending tile of the first pending
// lane, ballots the lanes whose rectangle holds it and clears it everywhere.  Steps = distinct tiles.  The results are
// parked one per lane and appended 64 tiles at a time (a returning atomic per step would serialise the loop on L2 latency).
#ifdef KAMD_PHASE_PROF
static __device__ unsigned long long g_phase_bin[PHASE_ROWS * 16];  // [0..7] phases, [10] longest wavefront, [11] > 1000 ticks (10 us at 100 MHz), [12] > 2500;
                                                       // wave_bin, both passes: [8] big-list appends, [9] medium faces, [15] the merging loop
#endif
template <bool SOFT>
__device__ __forceinline__ void wave_bin(bool active, bool big, int b, long long first_b, long long f, int tx0, int tx1,
                                         int ty0, int ty1, int c_lo, int c_hi, int r_lo, int r_hi, const Lists& L,
                                         PendingEntry* deferred = nullptr) {
  // `deferred`: the wavefront's LAST batch of entries is handed back instead of appended (the caller appends it together
  // with the other pass' last batch: append_entry_pair)
  if (deferred != nullptr) deferred->on = false;
  const int lane = threadIdx.x & 63;
  const unsigned int block = (unsigned int)(f >> 6);
  unsigned long long remaining = __ballot(active);
  while (remaining != 0ull) {
    const int leader = __ffsll((long long)remaining) - 1;
    const int bL = __builtin_amdgcn_readlane(b, leader);
    const bool mine = active && b == bL;
    remaining &= ~__ballot(mine);
#ifdef KAMD_PHASE_PROF
    unsigned long long wb_t = clock64();
#define WB_MARK(i) { const unsigned long long wb_n = clock64(); if (lane == 0) atomicAdd(&PHASE_ROW(g_phase_bin)[i], wb_n - wb_t); wb_t = wb_n; }
#else
#define WB_MARK(i)
#endif
    // big faces: appended to the mesh's big list
    const unsigned long long bigm = __ballot(mine && big);
    if (bigm != 0ull) {
      {
        // The big list holds ENTRIES {block, face mask} like the tiles' own: the big faces of this wavefront, merged per 64-face block
        // (at most two blocks: its faces are consecutive).  Every soft work item of the mesh orders the big list among its tile's
        // entries (soft2.inc), every rasterizer tile a big face's rectangle holds expands it behind its own (raster2.inc): the knot's
        // ~170 image-sized faces per view are consecutive in its face list -- a handful of entries instead of 170, few soft tiles
        // left over the ordered-slot limit, rasterizer tiles back under the 64 entries one wavefront holds.  (Rounds 2-6: one list
        // element per face.)
        for (unsigned long long todo = bigm; todo != 0ull;) {
          const int l0 = __ffsll((long long)todo) - 1;
          const unsigned int blk = (unsigned int)__builtin_amdgcn_readlane((int)block, l0);
          const bool same = mine && big && block == blk;
          todo &= ~__ballot(same);
          const unsigned long long bit = same ? 1ull << ((unsigned long long)f & 63ull) : 0ull;
          const unsigned int mlo = wave_or_u32((unsigned int)bit), mhi = wave_or_u32((unsigned int)(bit >> 32));
          if (lane == l0) {
            const unsigned int at = atomicAdd(L.big_count + bL, 1u);
            unsigned int* e = L.big_list + 3 * ((size_t)first_b + at);
            e[0] = blk;
            e[1] = mlo;
            e[2] = mhi;
          }
        }
        if (SOFT && L.big_hash != nullptr && mine && big) big_hash_insert(L.big_hash, (unsigned int)f);
      }
      if (!SOFT && L.big_rows_off != 0u) {
        // the tiles a big face's rectangle holds, a 64-bit word per 64 tiles of a row: one face at a time, a lane per row
        const int wpr = big_row_words(L.tiles_x), tiles_y = L.ntiles / L.tiles_x;
        unsigned long long* rows = reinterpret_cast<unsigned long long*>(L.big_count + L.big_rows_off) + (size_t)bL * tiles_y * wpr;
        for (unsigned long long bm = bigm; bm != 0ull; bm &= bm - 1ull) {
          const int l = __ffsll((long long)bm) - 1;
          const int x0 = __builtin_amdgcn_readlane(tx0, l), x1 = __builtin_amdgcn_readlane(tx1, l);
          const int y0 = __builtin_amdgcn_readlane(ty0, l), y1 = __builtin_amdgcn_readlane(ty1, l);
          for (int row = y0 + lane; row <= y1; row += 64)
            for (int w = x0 >> 6; w <= (x1 >> 6); ++w) {
              const int lo = max(x0 - w * 64, 0), hi = min(x1 - w * 64, 63);  // bits lo..hi of word w
              const unsigned long long m = (hi >= 63 ? ~0ull : ((1ull << (hi + 1)) - 1ull)) & ~((1ull << lo) - 1ull);
              atomicOr(rows + (size_t)row * wpr + w, m);
            }
        }
      }
    }
    WB_MARK(8);
    // medium faces -- a rectangle of more than MEDIUM_TILES tiles (at most 8 x 8: beyond that a face is `big`): every lane walks the
    // tiles of ITS OWN face and appends a single-face entry to each, MQ tiles per step with their counter atomics in flight
    // together (append_own); steps = the largest rectangle of the wavefront / MQ, no cross-lane work at all.  The loop below merges the faces
    // of a wavefront that share a tile into one entry, at one step of ~40 dependent instructions per DISTINCT tile of the
    // wavefront: right for faces of a few pixels (64 of them share ~6 tiles), wrong for faces 30-200 pixels across (64 of them
    // touch 500-3000 tiles, hardly any shared) -- the knot scene's bowl (170 faces per view) took the launch from 38 to 145 us,
    // the same bowl cut into 2 720 faces of ~30 pixels to 133 us, with every other wavefront long gone.
    const bool medium = mine && !big && (tx1 - tx0 + 1) * (ty1 - ty0 + 1) > MEDIUM_TILES;
    {
      constexpr int MQ = KAMD_MEDIUM_BATCH;
      const int n_own = medium ? (tx1 - tx0 + 1) * (ty1 - ty0 + 1) : 0;
      RectWalk at{tx0, ty0};
      for (int k0 = 0; __ballot(k0 < n_own) != 0ull; k0 += MQ)
        append_own<SOFT, MQ>(n_own - k0, at, tx0, tx1, (unsigned int)bL * (unsigned int)L.ntiles, block, 1ull << lane, c_lo, c_hi, r_lo, r_hi, L);
    }
    WB_MARK(9);
    const bool small = mine && !big && !medium;
    unsigned long long pending = 0ull;
    if (small) {
      const int w = tx1 - tx0 + 1, h = ty1 - ty0 + 1;  // 1..8 each
      const unsigned long long rowm = (1ull << w) - 1ull;
      const unsigned long long rows = h >= 8 ? ~0ull : ((1ull << (8 * h)) - 1ull);
      pending = (rowm * 0x0101010101010101ull) & rows;
    }
    int my_t = -1, k = 0;
    unsigned long long my_bal = 0ull;
    unsigned int my_sub = 0u;
    auto flush = [&]() {
      const size_t ti = (size_t)bL * L.ntiles + (my_t >= 0 ? my_t : 0);
      // (soft pass: the entry also carries the sub-tiles its faces reach, so that a work item can skip whole entries)
      append_entry(my_t >= 0, L, ti, make_uint4(block, SOFT ? my_sub : 0u, (unsigned int)my_bal, (unsigned int)(my_bal >> 32)));
      if (SOFT && my_t >= 0 && my_sub != 0u) atomicOr(L.sub_touched + ti, my_sub);
      my_t = -1;
      k = 0;
    };
    for (;;) {
      const unsigned long long todo = __ballot(pending != 0ull);
      if (todo == 0ull) break;
      const int l2 = __ffsll((long long)todo) - 1;
      const int i = pending != 0ull ? __ffsll((long long)pending) - 1 : 0;
      const int tx = __builtin_amdgcn_readlane(tx0 + (i & 7), l2);
      const int ty = __builtin_amdgcn_readlane(ty0 + (i >> 3), l2);
      const bool inr = small && tx >= tx0 && tx <= tx1 && ty >= ty0 && ty <= ty1;
      const unsigned long long bal = __ballot(inr);
      if (inr) pending &= ~(1ull << ((ty - ty0) * 8 + (tx - tx0)));
      unsigned int sub = 0u;
      if (SOFT) {
        // the 16 x 4-pixel sub-tiles of tile (tx, ty) the lane's pixel range reaches: bit = sy * 2 + sx
        unsigned int m16 = 0u;
        if (inr) {
          const int px0 = tx * S_TILE, py0 = ty * S_TILE;
          const int sx0 = max(c_lo - px0, 0) / SUB_W, sx1 = min(c_hi - px0, S_TILE - 1) / SUB_W;
          const int sy0 = max(r_lo - py0, 0) / SUB_H, sy1 = min(r_hi - py0, S_TILE - 1) / SUB_H;
          const unsigned int colbits = (sx0 == 0 ? 1u : 0u) | (sx1 >= 1 ? 2u : 0u);
          const unsigned int rowsel = ((1u << (2 * (sy1 + 1))) - 1u) & ~((1u << (2 * sy0)) - 1u) & 0x5555u;
          m16 = colbits * rowsel;
        }
        sub = wave_or_u32(m16);
      }
      if (lane == k) {
        my_t = ty * L.tiles_x + tx;
        my_bal = bal;
        my_sub = sub;
      }
      if (++k == 64) flush();
    }
    WB_MARK(15);
    if (deferred != nullptr && remaining == 0ull) {
      deferred->on = my_t >= 0;
      deferred->ti = (size_t)bL * L.ntiles + (my_t >= 0 ? my_t : 0);
      deferred->entry = make_uint4(block, SOFT ? my_sub : 0u, (unsigned int)my_bal, (unsigned int)(my_bal >> 32));
      if (SOFT && my_t >= 0 && my_sub != 0u) atomicOr(L.sub_touched + deferred->ti, my_sub);
    } else {
      flush();
    }
  }
}

// ---- the bin kernel -------------------------------------------------------------------------------------------------------
// One thread per face of the packed face list, ONE pass.  Inputs come in two flavours:
//   raw     the Python layer's (B, F, ...) tensors: vertices scaled by `mult` here, boxes = min / max over the three
//           vertices (-+ margin for the soft pass), faces with valid[f] == 0 or front[f] < 0 skipped by the rasterizer
//           pass -- the torch glue of rasterization.py:292-327 / dibr.py:31-39 (incl. its torch.where host sync);
//   packed  the reference operators' own inputs: scaled vertices, boxes given, `first` (B+1) face ranges (rasterizer) or
//           a dense batch (soft mask).
template <typename T>
struct BinIn {
  int B, F;                  // dense batch: mesh b owns faces [b*F, (b+1)*F); with `first`: F unused
  long long total_faces;
  const int64_t* first;      // (B+1) packed face ranges, or nullptr
  const T* img;              // (total, 3, 2)
  const T* z;                // raster: (total, 3) through `lay`; may be nullptr
  FaceLayout lay;
  const uint8_t* valid;      // raster, raw: optional
  const T* front;            // raster, raw: optional (kept when >= 0)
  const T* bbox_r;           // raster, packed: given boxes (total, 4); nullptr -> from the vertices
  const T* bbox_s;           // soft, packed: given large boxes (total, 4); nullptr -> from the vertices -+ margin
  T mult;                    // raw: scale applied to img (1 when already scaled)
  T margin;                  // soft, raw: boxlen * multiplier
  float multiplier;          // the operator's float multiplier (pixel-centre arithmetic)
  int H, W;
  T* rec_r;
  T* rec_s;
};

// ---- where the tile kernels start ------------------------------------------------------------------------------------------
// The workgroup of a tile with faces lives ~100x longer than a background tile's, so the tile kernels visit a view's tile
// rows outwards from the middle of the rows the mesh covers: the long workgroups start first and the background rows stream
// out beside their tail (round 2 started from the middle of the IMAGE -- right for a centred object only).  The span of
// covered rows is an ESTIMATE (any order is correct) and has to cost nothing, because everything that was tried to get it
// exactly did: every wavefront of the binning launch is resident and they all finish together, so ANY per-workgroup
// publication -- two atomicMax per workgroup on a view's line, even behind the last barrier and skipped when a read shows the
// span already covered -- queues ~3 000 same-line atomics (~25 ns each) at the end of the launch: +9 to +17 us; ranking the
// rows by the faces their tiles list (a ticket per workgroup, a sort in the last one) +8 us, and with a device-scope fence in
// the ticket (an L2 write-back per workgroup) the launch tripled.  So only every SPAN_SAMPLE-th workgroup reports -- ~100
// groups of 256 consecutive faces, spread evenly over the face list -- and the tile kernels shift their fixed visiting order
// (rows from the middle of the image outwards) cyclically so that it starts in the middle of the reported span: one scalar
// load and three scalar instructions per workgroup (a closed-form "outwards from c" map and a ranked row table were measured
// too: they pushed raster_tile, which sits at its 64-VGPR limit, into spilling -- 98 -> 121-150 us).
// row k of the visiting order: rows from the middle of the image outwards, shifted cyclically to start at row `centre`
__host__ __device__ inline int row_of_order(int k, int centre, int tiles_y) {
  const int mid = tiles_y >> 1;
  int ty = ((k & 1) ? mid - ((k + 1) >> 1) : mid + (k >> 1)) + (centre - mid);
  return ty < 0 ? ty + tiles_y : (ty >= tiles_y ? ty - tiles_y : ty);
}
// the middle of view b's reported span (the middle of the image when nothing was reported); uniform loads
__device__ __forceinline__ int row_centre(const unsigned int* __restrict__ row_span, int b, int tiles_y) {
  if (row_span == nullptr) return tiles_y >> 1;
  const unsigned int hi1 = row_span[2 * b], lo_inv = row_span[2 * b + 1];
  if (hi1 == 0u || lo_inv == 0u) return tiles_y >> 1;
  const int c = (tiles_y - (int)lo_inv + (int)hi1 - 1) >> 1;
  return c < 0 ? 0 : (c > tiles_y - 1 ? tiles_y - 1 : c);
}
// called by every thread of a REPORTING workgroup of the binning kernel (uniform per workgroup): `act` = the lane's face is
// kept and reaches the image, tile rows [r0, r1]
__device__ __forceinline__ void note_row_span(const Lists& L, bool act, int b, int r0, int r1) {
  const int tiles_y = L.ntiles / L.tiles_x;
  const unsigned long long am = __ballot(act);
  if (am == 0ull) return;
  // the wavefront's first view is reduced (a wavefront of 64 consecutive faces rarely straddles two)
  const int b0 = __builtin_amdgcn_readlane(b, __ffsll((long long)am) - 1);
  const bool mine = act && b == b0;
  int lo = mine ? r0 : 0x7FFFFFFF, hi = mine ? r1 : -1;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    lo = min(lo, __shfl_xor(lo, d, 64));
    hi = max(hi, __shfl_xor(hi, d, 64));
  }
  if ((threadIdx.x & 63) == 0) {
    atomicMax(L.row_span + 2 * b0, (unsigned int)(hi + 1));
    atomicMax(L.row_span + 2 * b0 + 1, (unsigned int)(tiles_y - lo));
  }
}

#ifndef KAMD_BIN_WAVES
#define KAMD_BIN_WAVES 0  // waves per SIMD the binning kernel is compiled for (0: the compiler's choice, 70 VGPRs = 7; 8 spills: 48.6 vs 44.2 us)
#endif
template <typename T, bool DO_R, bool DO_S>
#if KAMD_BIN_WAVES > 0
__global__ __launch_bounds__(256, KAMD_BIN_WAVES) void bin_faces_kernel2(
#else
__global__ __launch_bounds__(256) void bin_faces_kernel2(
#endif
    BinIn<T> in, Lists LR, Lists LS) {
  PHASE_DECL;
#ifdef KAMD_PHASE_PROF
  const unsigned long long wall0 = wall_clock64();  // 100 MHz
#endif
  const long long f = (long long)blockIdx.x * 256 + threadIdx.x;
  bool live = f < in.total_faces;
  int b = 0;
  long long first_b = 0;
  if (live) {
    if (in.first == nullptr) {
      b = (int)(f / in.F);
      first_b = (long long)b * in.F;
    } else {
      while (b + 1 < in.B && in.first[b + 1] <= f) ++b;
      first_b = in.first[b];
      if (f >= in.first[in.B]) live = false;
    }
  }
  PHASE_MARK(0);
  bool act_r = false, act_s = false, big_r = false, big_s = false;
  int rx0 = 0, rx1 = 0, ry0 = 0, ry1 = 0, sx0 = 0, sx1 = 0, sy0 = 0, sy1 = 0;
  PixRange pr_s{0, 0, 0, 0, false};
  if (live) {
    T v[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) v[i] = in.img[f * 6 + i] * in.mult;
    // everything the record needs is requested up front, whether or not the face turns out to be kept: the kernel is bound
    // by its dependent round trips, and `valid` / `front` -> (kept?) -> z would be two more of them
    uint8_t valid_f = 1;
    T front_f = 0, z0 = 0, z1 = 0, z2 = 0;
    if (DO_R) {
      if (in.valid != nullptr) valid_f = in.valid[f];
      if (in.front != nullptr) front_f = in.front[f * in.lay.front_stride];
      if (in.z != nullptr) {
        z0 = in.z[f * in.lay.z_face + 0 * in.lay.z_vertex];
        z1 = in.z[f * in.lay.z_face + 1 * in.lay.z_vertex];
        z2 = in.z[f * in.lay.z_face + 2 * in.lay.z_vertex];
      }
    }
#ifdef KAMD_PHASE_PROF
    asm volatile("s_waitcnt vmcnt(0)");
#endif
    PHASE_MARK(1);
    T xmin = nan_min<T>(nan_min<T>(v[0], v[2]), v[4]), xmax = nan_max<T>(nan_max<T>(v[0], v[2]), v[4]);
    T ymin = nan_min<T>(nan_min<T>(v[1], v[3]), v[5]), ymax = nan_max<T>(nan_max<T>(v[1], v[3]), v[5]);
    if (DO_R) {
      bool keep = valid_f != 0;
      if (keep && in.front != nullptr && !(front_f >= (T)0)) keep = false;
      T bx0 = xmin, by0 = ymin, bx1 = xmax, by1 = ymax;
      if (in.bbox_r != nullptr) {
        bx0 = in.bbox_r[f * 4 + 0];
        by0 = in.bbox_r[f * 4 + 1];
        bx1 = in.bbox_r[f * 4 + 2];
        by1 = in.bbox_r[f * 4 + 3];
      }
      {
        Rec4<T>* r = reinterpret_cast<Rec4<T>*>(in.rec_r + (size_t)f * REC_R);
        // (a face the rasterizer filters -- invalid / back facing, half of a closed mesh -- is on no list; the overflow
        // fallback walks every record of the mesh, so its box is stored empty: every strip rejects it, and the rest of
        // its record is never looked at nor written)
        r[0] = keep ? Rec4<T>{bx0, by0, bx1, by1} : Rec4<T>{(T)INFINITY, (T)INFINITY, -(T)INFINITY, -(T)INFINITY};
        if (keep) {
          r[1] = Rec4<T>{v[0], v[1], v[2], v[3]};
          r[2] = Rec4<T>{v[4], v[5], z0, z1};
          T e7[7] = {0, 0, 1, 0, 0, 1, 3};
          if constexpr (sizeof(T) == 4) {
            edge_coefficients(v, bx0, by0, bx1, by1, e7);
          }
          r[3] = Rec4<T>{e7[0], e7[3], e7[1], e7[4]};  // the two edges' A, then B: operand pairs of one packed fma
          r[4] = Rec4<T>{e7[2], e7[5], e7[6], z2};
        }
      }
      PHASE_MARK(2);
      PixRange pr;
      if (keep && pixel_range<T>(bx0, by0, bx1, by1, in.H, in.W, in.multiplier, &pr)) {
        act_r = true;
        rx0 = pr.c_lo / R_TILE;
        rx1 = pr.c_hi / R_TILE;
        ry0 = pr.r_lo / R_TILE;
        ry1 = pr.r_hi / R_TILE;
        // (the rasterizer's tile kernel tests every big face of the view against its tile -- a box compare per face -- so a face
        // is cheaper there than as BIG_TILES_R single-face entries, each a memory-side atomic here: the knot scene's bowl, 170
        // faces of 50-130 tiles per view, cost this launch 25 us as entries)
        big_r = pr.everywhere || rx1 - rx0 >= 8 || ry1 - ry0 >= 8 || (rx1 - rx0 + 1) * (ry1 - ry0 + 1) > BIG_TILES_R;
      }
    }
    PHASE_MARK(3);
    if (DO_S) {
      T bx0, by0, bx1, by1;
      if (in.bbox_s != nullptr) {
        bx0 = in.bbox_s[f * 4 + 0];
        by0 = in.bbox_s[f * 4 + 1];
        bx1 = in.bbox_s[f * 4 + 2];
        by1 = in.bbox_s[f * 4 + 3];
      } else {
        bx0 = xmin - in.margin;
        by0 = ymin - in.margin;
        bx1 = xmax + in.margin;
        by1 = ymax + in.margin;
      }
      {
        const size_t fs = (size_t)f;
        *reinterpret_cast<Rec4<T>*>(const_cast<T*>(soft_box<T>(in.rec_s, fs))) = Rec4<T>{bx0, by0, bx1, by1};
        T* body = const_cast<T*>(soft_body<T>(in.rec_s, (size_t)in.total_faces, fs));
        Rec4<T>* r = reinterpret_cast<Rec4<T>*>(body);
        r[0] = Rec4<T>{v[0], v[1], v[2], v[3]};
        r[1] = Rec4<T>{v[4], v[5], 0, 0};
        double* rc = reinterpret_cast<double*>(body + 8);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const T x1 = v[k * 2], y1 = v[k * 2 + 1], x2 = v[((k + 1) % 3) * 2], y2 = v[((k + 1) % 3) * 2 + 1];
          const T A = y2 - y1, Bc = x1 - x2;
          const T down = A * A + Bc * Bc;
          rc[k] = 1.0 / ((double)down + SOFT_EPS);
        }
      }
      PHASE_MARK(4);
      if (pixel_range<T>(bx0, by0, bx1, by1, in.H, in.W, in.multiplier, &pr_s)) {
        act_s = true;
        sx0 = pr_s.c_lo / S_TILE;
        sx1 = pr_s.c_hi / S_TILE;
        sy0 = pr_s.r_lo / S_TILE;
        sy1 = pr_s.r_hi / S_TILE;
        big_s = pr_s.everywhere || sx1 - sx0 >= 8 || sy1 - sy0 >= 8;
      }
    }
  }
  PHASE_MARK(5);
  if (DO_R && LR.row_span != nullptr && blockIdx.x % SPAN_SAMPLE == 0) note_row_span(LR, act_r, b, ry0, ry1);
  if (DO_R && DO_S) {
    PendingEntry er, es;
    wave_bin<false>(act_r, big_r, b, first_b, f, rx0, rx1, ry0, ry1, 0, 0, 0, 0, LR, &er);
    PHASE_MARK(6);
    wave_bin<true>(act_s, big_s, b, first_b, f, sx0, sx1, sy0, sy1, pr_s.c_lo, pr_s.c_hi, pr_s.r_lo, pr_s.r_hi, LS, &es);
    append_entry_pair(er, LR, es, LS);
  } else {
    if (DO_R) wave_bin<false>(act_r, big_r, b, first_b, f, rx0, rx1, ry0, ry1, 0, 0, 0, 0, LR);
    PHASE_MARK(6);
    if (DO_S) wave_bin<true>(act_s, big_s, b, first_b, f, sx0, sx1, sy0, sy1, pr_s.c_lo, pr_s.c_hi, pr_s.r_lo, pr_s.r_hi, LS);
  }
  PHASE_MARK(7);
  PHASE_FLUSH(g_phase_bin);
#ifdef KAMD_PHASE_PROF
  if ((threadIdx.x & 63) == 0) {
    const unsigned long long tot = wall_clock64() - wall0;
    atomicAdd(&PHASE_ROW(g_phase_bin)[14], tot);
    atomicMax(&PHASE_ROW(g_phase_bin)[10], tot);
    if (tot > 1000ull) atomicAdd(&PHASE_ROW(g_phase_bin)[11], 1ull);
    if (tot > 2500ull) atomicAdd(&PHASE_ROW(g_phase_bin)[12], 1ull);
    atomicAdd(&PHASE_ROW(g_phase_bin)[13], 1ull);
  }
#endif
}

// ---- consumers: the candidate faces of a tile ------------------------------------------------------------------------------
// A tile's candidates are (a) its entries, (b) the faces of its mesh's big list; a tile flagged BRUTE (an entry could not
// be stored) takes every block of its mesh instead.
struct TileSrc {
  size_t ti;
  unsigned int n;   // entries
  bool brute;
};
__device__ __forceinline__ TileSrc tile_src(const Lists& L, int b, int tile) {
  TileSrc s;
  s.ti = (size_t)b * L.ntiles + tile;
  const unsigned int raw = L.count[s.ti];
  s.brute = (raw & BRUTE_BIT) != 0u;
  s.n = raw & ~BRUTE_BIT;
  return s;
}
// e-th entry of a tile (e < n; the tile is not BRUTE, so every chunk it needs was published)
__device__ __forceinline__ uint4 tile_entry(const Lists& L, const TileSrc& s, unsigned int e) {
  if (e < (unsigned int)L.C) return L.inl[s.ti * L.C + e];
  const unsigned int o = e - (unsigned int)L.C, c = o / OVC_PAYLOAD, i = o - c * OVC_PAYLOAD;
  const unsigned int chunk = L.tab[s.ti * L.maxc + c] - 1u;
  return L.pool[(size_t)chunk * OVC + 1u + i];
}

// what the rasterizer's tile kernel needs to settle the soft mask's trivial pixels and queue the search's work items
struct ClassifyOut {
  void* soft_mask;                  // T* (B, H, W)
  const unsigned int* sub_touched;  // soft pass: [B * ntiles_s]
  const unsigned int* big_count_s;  // soft pass: [B]
  int tiles_x_s, ntiles_s;
  uint4* work_items;
  unsigned int* work_counts;
  unsigned int shard_cap;
  unsigned char* tile_cov;          // [B * ntiles_r] (work_cov_offset_words), or nullptr
};

// exclusive prefix over the 256 threads of a workgroup; *total = sum.  `scratch`: 4 ints of LDS.
__device__ __forceinline__ int block_scan_256(int v, int* scratch, int* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int inc = wave_inclusive_scan(v);
  __syncthreads();  // scratch may still be read by the previous round
  if (lane == 63) scratch[wave] = inc;
  __syncthreads();
  const int s0 = scratch[0], s1 = scratch[1], s2 = scratch[2], s3 = scratch[3];
  *total = s0 + s1 + s2 + s3;
  const int before = wave == 0 ? 0 : (wave == 1 ? s0 : (wave == 2 ? s0 + s1 : s0 + s1 + s2));
  return before + inc - v;
}

// ---- the soft-mask search's worklist ----------------------------------------------------------------------------------------
// A 256-thread workgroup covers 16 x 16 pixels = four 16 x 4 sub-tiles (one per wavefront).  Wavefront `wave` hands in
// the bit mask `unc` of its uncovered pixels; the sub-tile becomes a work item {item id, unc} when some enlarged face box
// reaches it (the soft pass' sub_touched bits; a mesh with big faces reaches everything).  One append per workgroup;
// the counter is sharded WORK_SHARDS ways by workgroup id.  item id = (soft tile * B + b) * 16 + sub-tile of the tile.
__device__ __forceinline__ void queue_items(unsigned long long unc, bool has_faces, int B, int b, int tile_x, int tile_y,
                                            int wave, int lane, const unsigned int* __restrict__ sub_touched,
                                            const unsigned int* __restrict__ big_count_s, int tiles_x_s, int ntiles_s,
                                            uint4* __restrict__ work_items, unsigned int* __restrict__ work_counts,
                                            unsigned int shard_cap, unsigned long long* s_item_unc, bool covered = false,
                                            unsigned char* __restrict__ tile_cov = nullptr, size_t cov_index = 0) {
  bool item = false;
  if (unc != 0ull && has_faces) {
    const int sy = tile_y + wave * SUB_H;
    const int st = (sy / S_TILE) * tiles_x_s + tile_x / S_TILE;
    const int ss = ((sy % S_TILE) / SUB_H) * (S_TILE / SUB_W) + (tile_x % S_TILE) / SUB_W;
    item = ((sub_touched[(size_t)b * ntiles_s + st] >> ss) & 1u) != 0u || big_count_s[b] != 0u;
  }
  if (lane == 0) s_item_unc[wave] = item ? unc : 0ull;
  const int any_covered = __syncthreads_or(covered ? 1 : 0);
  if (threadIdx.x == 0) {
    if (tile_cov != nullptr) tile_cov[cov_index] = any_covered ? 1 : 0;
    int n = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) n += s_item_unc[w] != 0ull ? 1 : 0;
    if (n > 0) {
      const unsigned int shard = blockIdx.x & (WORK_SHARDS - 1);
      unsigned int pos = atomicAdd(work_counts + shard * COUNTER_STRIDE, (unsigned int)n);
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const unsigned long long u = s_item_unc[w];
        if (u == 0ull) continue;
        const int sy = tile_y + w * SUB_H;
        const int st = (sy / S_TILE) * tiles_x_s + tile_x / S_TILE;
        const int ss = ((sy % S_TILE) / SUB_H) * (S_TILE / SUB_W) + (tile_x % S_TILE) / SUB_W;
        const unsigned int item_id = (unsigned int)((st * B + b) * S_SUBS + ss);
        if (pos < shard_cap)
          work_items[(size_t)shard * shard_cap + pos] = make_uint4(item_id, (unsigned int)u, (unsigned int)(u >> 32), 0u);
        ++pos;
      }
    }
  }
}

// The rasterizer's tile kernel reads what queue_items needs at its start, next to the tile's list head (one round trip
// instead of a second one at the tail).  Result: bit 2 w = some enlarged box reaches the 16 x 4 sub-tile of wavefront w of
// rasterizer tile (tx, ty) (the soft tile's word, shifted: its sub-tiles are numbered row by row, two per row).  Every
// operand is uniform over the workgroup: scalar loads.
static_assert(R_TILE == SUB_W && R_TILE == 4 * SUB_H && S_TILE == 2 * R_TILE, "reach_of_tile: four sub-tiles per rasterizer tile, 2 x 2 rasterizer tiles per soft tile");
constexpr unsigned int REACH_ALL = 0x55u;
__device__ __forceinline__ unsigned int reach_of_tile(const unsigned int* __restrict__ sub_touched, const unsigned int* __restrict__ big_count_s,
                                                      int tiles_x_s, int ntiles_s, int b, int tx, int ty) {
  const unsigned int big = big_count_s[b];
  const unsigned int w = sub_touched[(size_t)b * ntiles_s + (ty >> 1) * tiles_x_s + (tx >> 1)];
  return big != 0u ? REACH_ALL : (w >> ((ty & 1) * 8 + (tx & 1))) & REACH_ALL;
}
__device__ __forceinline__ bool reached(unsigned int reach, int w) { return ((reach >> (2 * w)) & 1u) != 0u; }
__device__ __forceinline__ unsigned int item_id_of(int B, int b, int tx, int ty, int tiles_x_s, int w) {
  const int st = (ty >> 1) * tiles_x_s + (tx >> 1), ss = (ty & 1) * 8 + 2 * w + (tx & 1);
  return (unsigned int)((st * B + b) * S_SUBS + ss);
}
// queue_items with the reach bits in hand (`shard`: the workgroup's worklist shard)
// (`tile_order`: the tile's place in the kernel's order of tiles; the workgroup's place in dispatch order is tile_order * B + b
// -- its shard of the worklist is that % WORK_SHARDS: no shard receives more than its share of the tiles, which is what the
// capacities assume; cov_shard_of picks the covered-tile list's)
__device__ __forceinline__ void queue_items_reached(unsigned long long unc, unsigned int reach, int B, int b, int tx, int ty, int tiles_x_s,
                                                    int wave, int lane, unsigned int tile_order, uint4* __restrict__ work_items,
                                                    unsigned int* __restrict__ work_counts, unsigned int shard_cap,
                                                    unsigned long long* s_item_unc, bool covered, unsigned char* __restrict__ tile_cov,
                                                    size_t cov_index, size_t ntiles_r) {
  const unsigned int shard = (tile_order * (unsigned int)B + (unsigned int)b) & (WORK_SHARDS - 1);
  const bool item = unc != 0ull && reached(reach, wave);
  if (lane == 0) s_item_unc[wave] = item ? unc : 0ull;
  const int any_covered = __syncthreads_or(covered ? 1 : 0);
  if (threadIdx.x == 0) {
    int n = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) n += s_item_unc[w] != 0ull ? 1 : 0;
    // the two appends' counter atomics are issued together (a returning atomic is a ~2 us round trip performed memory-side, and the
    // workgroup cannot retire before it: the first USE of a result is what makes the wavefront wait -- the lesson of the binning kernel)
    const bool list_cov = tile_cov != nullptr && any_covered;
    const unsigned int cs = cov_shard_of(B, b, tile_order);
    unsigned int pos_c = 0u, pos = 0u;
    if (list_cov) pos_c = atomicAdd(work_counts + WORK_COV_WORD + cs * COUNTER_STRIDE, 1u);
    if (n > 0) pos = atomicAdd(work_counts + shard * COUNTER_STRIDE, (unsigned int)n);
    if (tile_cov != nullptr) tile_cov[cov_index] = any_covered ? 1 : 0;
    if (list_cov) {  // the covered-tile list lives behind the coverage bytes and the span copy (work_covlist_offset_words)
      unsigned int* list = reinterpret_cast<unsigned int*>(tile_cov) + cov_list_words_after_cov((size_t)B, (size_t)B * ntiles_r);
      const unsigned int cap = cov_shard_cap((size_t)B, ntiles_r);
      if (pos_c < cap) list[(size_t)cs * cap + pos_c] = (unsigned int)cov_index;  // (always: cov_shard_of sends no shard more; the consumer clamps too)
    }
    if (n > 0) {
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const unsigned long long u = s_item_unc[w];
        if (u == 0ull) continue;
        if (pos < shard_cap)
          work_items[(size_t)shard * shard_cap + pos] = make_uint4(item_id_of(B, b, tx, ty, tiles_x_s, w), (unsigned int)u, (unsigned int)(u >> 32), 0u);
        ++pos;
      }
    }
  }
}
// The same in two steps, for a caller that has its own barrier (the rasterizer's tile kernel joins the barrier that frees its staging
// arrays with this one, and lets the counters' round trips run beside its output stores): every wavefront's lane 0 has written
// s_item_unc[wave] (queue_item_word) BEFORE the caller's __syncthreads_or(covered); then ONE thread calls queue_begin (issues the
// returning atomics, uses nothing) and, later, queue_finish (the stores that need the positions).
__device__ __forceinline__ unsigned long long queue_item_word(unsigned long long unc, unsigned int reach, int wave) {
  return (unc != 0ull && reached(reach, wave)) ? unc : 0ull;
}
struct QueueTicket {
  unsigned int pos_c, pos, cs, shard;
  int n;
  bool list_cov;
};
__device__ __forceinline__ QueueTicket queue_begin(bool any_covered, int B, int b, unsigned int tile_order, unsigned int* __restrict__ work_counts,
                                                   const unsigned long long* s_item_unc, unsigned char* __restrict__ tile_cov) {
  QueueTicket t;
  t.shard = (tile_order * (unsigned int)B + (unsigned int)b) & (WORK_SHARDS - 1);
  t.n = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) t.n += s_item_unc[w] != 0ull ? 1 : 0;
  t.list_cov = tile_cov != nullptr && any_covered;
  t.cs = cov_shard_of(B, b, tile_order);
  t.pos_c = 0u;
  t.pos = 0u;
  if (t.list_cov) t.pos_c = atomicAdd(work_counts + WORK_COV_WORD + t.cs * COUNTER_STRIDE, 1u);
  if (t.n > 0) t.pos = atomicAdd(work_counts + t.shard * COUNTER_STRIDE, (unsigned int)t.n);
  return t;
}
__device__ __forceinline__ void queue_finish(const QueueTicket& t, bool any_covered, int B, int b, int tx, int ty, int tiles_x_s,
                                             uint4* __restrict__ work_items, unsigned int shard_cap, const unsigned long long* s_item_unc,
                                             unsigned char* __restrict__ tile_cov, size_t cov_index, size_t ntiles_r) {
  if (tile_cov != nullptr) tile_cov[cov_index] = any_covered ? 1 : 0;
  if (t.list_cov) {
    unsigned int* list = reinterpret_cast<unsigned int*>(tile_cov) + cov_list_words_after_cov((size_t)B, (size_t)B * ntiles_r);
    const unsigned int cap = cov_shard_cap((size_t)B, ntiles_r);
    if (t.pos_c < cap) list[(size_t)t.cs * cap + t.pos_c] = (unsigned int)cov_index;
  }
  if (t.n > 0) {
    unsigned int pos = t.pos;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const unsigned long long u = s_item_unc[w];
      if (u == 0ull) continue;
      if (pos < shard_cap)
        work_items[(size_t)t.shard * shard_cap + pos] = make_uint4(item_id_of(B, b, tx, ty, tiles_x_s, w), (unsigned int)u, (unsigned int)(u >> 32), 0u);
      ++pos;
    }
  }
}
// A background tile that lies fully inside the image: every pixel of every sub-tile is uncovered, so ONE lane can queue
// the tile's items without hearing from the other wavefronts (no LDS, no barrier).  Called by one lane.
__device__ __forceinline__ void queue_items_background(unsigned int reach, int B, int b, int tx, int ty, int tiles_x_s, unsigned int tile_order,
                                                       uint4* __restrict__ work_items, unsigned int* __restrict__ work_counts,
                                                       unsigned int shard_cap, unsigned char* __restrict__ tile_cov, size_t cov_index) {
  const unsigned int shard = (tile_order * (unsigned int)B + (unsigned int)b) & (WORK_SHARDS - 1);
  if (tile_cov != nullptr) tile_cov[cov_index] = 0;
  if (reach == 0u) return;
  unsigned int pos = atomicAdd(work_counts + shard * COUNTER_STRIDE, (unsigned int)__popc(reach));
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    if (!reached(reach, w)) continue;
    if (pos < shard_cap) work_items[(size_t)shard * shard_cap + pos] = make_uint4(item_id_of(B, b, tx, ty, tiles_x_s, w), ~0u, ~0u, 0u);
    ++pos;
  }
}

}  // namespace tl
}  // namespace kamd
