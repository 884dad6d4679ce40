// fdf.hip — foreground filtering (reference fdf.hpp:40-89, fde.hpp:83-146, cte.hpp:64-166) for
// a batch of frames of one fragment, resident in HBM.
//
// Per frame the reference flood-fills, seed by seed in scan order, the 4-connected component of
// equal MEDIAN colour around every interior pixel whose FRAME pixel differs from the fragment's
// background (its blended canvas), keeps the components of at most W*H/5 pixels, and masks
// their pixels plus their bounding boxes out of a second blit of the frame.  What it computes
// is order-free except for one quirk (below), so here it is a connected-component labelling:
//   fdf_init     label = own index for interior pixels (rows 1..H-3, columns 1..W-2: the
//                reference's "horizon" ring includes row H-2, cte.hpp:149-166), statistics reset
//   fdf_merge    union-find over left/up neighbours of equal median colour (atomicMin on roots)
//   fdf_stats    labels flattened to roots; area / bounding box / seeded per root, atomics
//                aggregated over runs of equal roots inside a wave
//   fdf_keep     roots that are seeded and small enough -> per-frame list; mask = own pixels
//   fdf_boxes    one wave per kept component: the box's left side as cdt::limits::update
//                leaves it (cdt.hpp:183-190: `else if` — the smallest left/right-edge x that is
//                <= the largest edge x of the rows above; no such pixel -> empty box), then the
//                box, without its last row and column (ctr.hpp:118-131, fde.hpp:134-143), is
//                painted into the mask
// The masked blit itself is fgm_blit with a mask plane (fgm.hip).
#include "common.h"

namespace rmx {

struct FdfParams {
  const uint8_t* bg;      // background image bw x bh (blend of the fragment's canvas)
  const uint8_t* frames;  // n x H x W
  const uint8_t* median;  // n x H x W
  const int32_t* pos;     // n x 2: frame position inside the background
  uint8_t* mask;          // n x H x W out: 1 = foreground
  uint32_t* ncontours;    // n out (may be null)
  // workspace, per frame P = W*H entries
  uint32_t* label;
  uint32_t* area;
  uint32_t* minx;
  uint32_t* maxx;
  uint32_t* miny;
  uint32_t* maxy;
  uint8_t* seeded;
  uint32_t* kept;  // per frame: [0] = count, [1..cap] root indices
  uint32_t* status;
  uint32_t n, W, H, bw, bh, kept_cap;
};

constexpr uint32_t FDF_NONE = 0xffffffffu;
constexpr uint32_t FDF_KEPT_CAP = 4096;  // kept components per frame the list holds (more -> RMX_ERR_CAPACITY)

__device__ __forceinline__ bool fdf_interior(const FdfParams& p, uint32_t x, uint32_t y) {
  return x >= 1 && x + 1 < p.W && y >= 1 && y + 2 < p.H;
}

// Labels start as "leftmost pixel of my run of equal colour inside my wave" (a ballot, no
// memory traffic), so the union-find only has to stitch runs: across wave boundaries and
// between rows, once per pair of overlapping runs (fdf_merge).  A per-pixel union with the
// left neighbour makes chains as long as the flat areas are wide: 7.5 ms vs 0.86 ms per 800 frames.
__global__ __launch_bounds__(256) void fdf_init(FdfParams p) {
  const uint32_t P = p.W * p.H;
  const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
  const bool live = g < (size_t)p.n * P;
  const uint32_t i = live ? (uint32_t)(g % P) : 0, x = i % p.W, y = i / p.W, lane = threadIdx.x & 63u;
  const bool in = live && fdf_interior(p, x, y);
  const bool left_same = in && x >= 2 && p.median[g - 1] == p.median[g];
  const uint64_t heads = __builtin_amdgcn_ballot_w64(!left_same || lane == 0);
  if (!live) return;
  const uint32_t start = 63u - (uint32_t)__builtin_clzll(heads & (~0ull >> (63u - lane)));  // last head at or before me
  p.label[g] = in ? i - (lane - start) : FDF_NONE;
  p.area[g] = 0;
  p.minx[g] = FDF_NONE;
  p.maxx[g] = 0;
  p.miny[g] = FDF_NONE;
  p.maxy[g] = 0;
  p.seeded[g] = 0;
  if (i == 0) p.kept[(size_t)(g / P) * (p.kept_cap + 1)] = 0;
}

__device__ __forceinline__ uint32_t fdf_find(const uint32_t* L, uint32_t a) {
  for (;;) {
    const uint32_t q = L[a];
    if (q == a) return a;
    a = q;
  }
}
__device__ __forceinline__ void fdf_union(uint32_t* L, uint32_t a, uint32_t b) {
  for (;;) {
    a = fdf_find(L, a);
    b = fdf_find(L, b);
    if (a == b) return;
    if (a < b) {
      const uint32_t t = a;
      a = b;
      b = t;
    }
    const uint32_t old = atomicMin(&L[a], b);  // hang the larger root under the smaller
    if (old == a) return;
    a = old;  // someone re-rooted a meanwhile: go on from there
  }
}

__global__ __launch_bounds__(256) void fdf_merge(FdfParams p) {
  const uint32_t P = p.W * p.H;
  const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (g >= (size_t)p.n * P) return;
  const uint32_t i = (uint32_t)(g % P), x = i % p.W, y = i / p.W;
  if (!fdf_interior(p, x, y)) return;
  const size_t f0 = g - i;
  const uint8_t* med = p.median + f0;
  uint32_t* L = p.label + f0;
  const uint8_t c = med[i];
  const bool left_same = x >= 2 && med[i - 1] == c;
  // a run that continues across a wave boundary
  if (left_same && (threadIdx.x & 63u) == 0) fdf_union(L, i, i - 1);
  // the run above: once per pair of overlapping runs — if my left neighbour continues my run
  // and its upper neighbour continues the upper run, that pixel (or one further left) does it
  if (y >= 2 && med[i - p.W] == c && !(left_same && med[i - p.W - 1] == c)) fdf_union(L, i, i - p.W);
}

// Lanes hold consecutive pixels.  True on the first lane of every run of lanes with the same
// root in the same row; len = lanes in that run (inside this wave).  All lanes must call.
__device__ __forceinline__ bool fdf_run_head(uint32_t root, uint32_t x, uint32_t& len) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t prev = (uint32_t)__shfl_up((int)root, 1);
  const uint32_t px = (uint32_t)__shfl_up((int)x, 1);
  const bool head = root != FDF_NONE && (lane == 0 || prev != root || px + 1 != x);
  const uint64_t breaks = __builtin_amdgcn_ballot_w64(root == FDF_NONE || head);
  const uint64_t later = lane == 63 ? 0ull : (breaks >> (lane + 1));
  len = later ? (uint32_t)__builtin_ctzll(later) + 1 : 64 - lane;
  return head;
}

__global__ __launch_bounds__(256) void fdf_stats(FdfParams p) {
  const uint32_t P = p.W * p.H;
  const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
  const bool live = g < (size_t)p.n * P;
  const uint32_t i = live ? (uint32_t)(g % P) : 0, x = i % p.W, y = i / p.W;
  const size_t f0 = g - i;
  uint32_t root = FDF_NONE;
  if (live && fdf_interior(p, x, y)) {
    root = fdf_find(p.label + f0, i);
    p.label[g] = root;  // (a pixel only ever shortens its own chain)
    const int32_t bx = p.pos[2 * (g / P)], by = p.pos[2 * (g / P) + 1];
    if (p.frames[g] != p.bg[(size_t)(by + (int32_t)y) * p.bw + (size_t)(bx + (int32_t)x)]) p.seeded[f0 + root] = 1;
  }
  // runs of equal roots along a row inside the wave: the run's first lane adds for all.  Only
  // the area here — a frame's big background component takes thousands of these per frame on
  // one address; its bounding box is never needed (fdf_keep adds boxes of kept components only)
  uint32_t len;
  if (fdf_run_head(root, x, len)) atomicAdd(&p.area[f0 + root], len);
}

__global__ __launch_bounds__(256) void fdf_keep(FdfParams p) {
  const uint32_t P = p.W * p.H;
  const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
  const bool live = g < (size_t)p.n * P;
  const uint32_t i = live ? (uint32_t)(g % P) : 0, x = i % p.W, y = i / p.W;
  const size_t f0 = g - i;
  const uint32_t root = live ? p.label[g] : FDF_NONE;
  const uint32_t limit = P / 5;  // fde.hpp:109-116
  const bool keep = root != FDF_NONE && p.seeded[f0 + root] && p.area[f0 + root] <= limit;
  if (keep && root == i) {
    uint32_t* list = p.kept + (size_t)(g / P) * (p.kept_cap + 1);
    const uint32_t at = atomicAdd(&list[0], 1u);
    if (at < p.kept_cap)
      list[1 + at] = i;
    else
      atomicOr(&p.status[3], 1u);
  }
  uint32_t len;
  if (fdf_run_head(keep ? root : FDF_NONE, x, len)) {  // bounding boxes of the kept components only
    atomicMin(&p.minx[f0 + root], x);
    atomicMax(&p.maxx[f0 + root], x + len - 1);
    atomicMin(&p.miny[f0 + root], y);
    atomicMax(&p.maxy[f0 + root], y);
  }
  if (live) p.mask[g] = keep ? 1 : 0;
}

// one workgroup per frame, its four waves take the kept components in turn
__global__ __launch_bounds__(256) void fdf_boxes(FdfParams p) {
  const uint32_t f = blockIdx.x, lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t* list = p.kept + (size_t)f * (p.kept_cap + 1);
  const uint32_t cnt = min(list[0], p.kept_cap);
  if (threadIdx.x == 0 && p.ncontours) p.ncontours[f] = list[0];
  const uint32_t P = p.W * p.H;
  const size_t f0 = (size_t)f * P;
  const uint32_t* L = p.label + f0;
  uint8_t* m = p.mask + f0;
  for (uint32_t slot = wave; slot < cnt; slot += 4) {
    const uint32_t root = list[1 + slot];
    const uint32_t x0 = p.minx[f0 + root], x1 = p.maxx[f0 + root], y0 = p.miny[f0 + root], y1 = p.maxy[f0 + root];
    // cdt::limits fed with the x of the left/right edge pixels in row-major order
    uint32_t upper_prev = 0, lower = FDF_NONE;  // upper_prev: largest edge x of the rows above
    for (uint32_t y = y0; y <= y1; ++y) {
      uint32_t row_max = 0, row_low = FDF_NONE;
      for (uint32_t xb = x0; xb <= x1; xb += 64) {
        const uint32_t x = xb + lane;
        if (x <= x1) {
          const uint32_t i = y * p.W + x;
          if (L[i] == root && (L[i - 1] != root || L[i + 1] != root)) {  // interior x: both neighbours exist
            row_max = max(row_max, x);
            if (x <= upper_prev) row_low = min(row_low, x);
          }
        }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        row_max = max(row_max, (uint32_t)__shfl_xor((int)row_max, o));
        row_low = min(row_low, (uint32_t)__shfl_xor((int)row_low, o));
      }
      lower = min(lower, row_low);
      upper_prev = max(upper_prev, row_max);
    }
    if (lower == FDF_NONE) continue;  // left_ stays SIZE_MAX in the reference: nothing is painted
    for (uint32_t y = y0; y < y1; ++y)
      for (uint32_t x = lower + lane; x < x1; x += 64) m[y * p.W + x] = 1;
  }
}

}  // namespace rmx
