// fdf.hip — foreground filtering (reference fdf.hpp:40-89, fde.hpp:83-146, cte.hpp:64-166) for
// a batch of frames of one fragment, resident in HBM.
//
// Per frame the reference flood-fills, seed by seed in scan order, the 4-connected component of
// equal MEDIAN colour around every interior pixel whose FRAME pixel differs from the fragment's
// background (its blended canvas), keeps the components of at most W*H/5 pixels, and masks
// their pixels plus their bounding boxes out of a second blit of the frame.  What it computes
// is order-free except for one quirk (below), so here it is a connected-component labelling:
//   fdf_local    a workgroup labels one strip of rows of one frame in LDS: run labels from a
//                ballot, a union-find (ds atomics) over the runs of the strip, labels flattened
//                to the strip's roots (rows 1..H-3, columns 1..W-2 only: the reference's
//                "horizon" ring includes row H-2, cte.hpp:149-166); statistics reset
//   fdf_stitch   unions across the strip boundaries (global atomicMin on the strips' roots)
//   fdf_stats    labels flattened to roots; area / seeded per root, atomics aggregated over
//                runs of equal roots inside a wave
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
  uint32_t* area;  // per root: pixels | FDF_SEEDED
  uint32_t* minx;
  uint32_t* maxx;
  uint32_t* miny;
  uint32_t* maxy;
  uint32_t* kept;  // per frame: [0] = count, [1..cap] root indices
  uint32_t* status;
  uint32_t n, W, H, bw, bh, kept_cap;
};

constexpr uint32_t FDF_NONE = 0xffffffffu;
constexpr uint32_t FDF_SEEDED = 0x80000000u;  // bit 31 of a root's area word: one of its pixels differs from the background
constexpr uint32_t FDF_KEPT_CAP = 4096;  // kept components per frame the list holds (more -> RMX_ERR_CAPACITY)

__device__ __forceinline__ bool fdf_interior(const FdfParams& p, uint32_t x, uint32_t y) {
  return x >= 1 && x + 1 < p.W && y >= 1 && y + 2 < p.H;
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

// Strip-local labelling.  Labels start as "leftmost pixel of my run of equal colour inside my
// wave" (a ballot, no memory traffic), so the union-find only has to stitch runs: across wave
// boundaries and between rows, once per pair of overlapping runs.  (A per-pixel union with the
// left neighbour makes chains as long as the flat areas are wide.)  All of it runs on LDS; HBM
// sees one read of the medians and one write of the labels.  History per 833 frames 320x240:
// whole-frame union-find in HBM with one thread per pixel: init 268 + merge 856 us, and 700 us
// for fdf_stats to walk the chains it left.
#ifndef FDF_TILE_PX
#define FDF_TILE_PX 3840
#endif
// pixels of a strip (5 B of LDS each: 19 KB, eight workgroups per CU).  Per 833 frames 320x240,
// local + stitch + stats: 7680 -> 319 + 28 + 232 us, 5120 -> 263 + 44 + 245, 3840 -> 212 + 60 + 251,
// 2560 -> 206 + 92 + 260: smaller strips label faster and leave more to stitch and longer chains
constexpr uint32_t FDF_TILE = FDF_TILE_PX;
constexpr uint32_t FDF_TILE_WIDE = 2 * FDF_TILE;  // the instance for frames wider than FDF_TILE

template <uint32_t TILE>
__global__ __launch_bounds__(256) void fdf_local(FdfParams p, uint32_t R, uint32_t S) {
  __shared__ uint32_t lab[TILE];
  __shared__ uint8_t col[TILE];
  const uint32_t f = blockIdx.x / S, s = blockIdx.x - f * S, tid = threadIdx.x, lane = tid & 63u;
  const uint32_t W = p.W, P = p.W * p.H;
  const uint32_t ys = 1 + s * R, ye = min(ys + R, p.H >= 2 ? p.H - 2 : 0u);  // interior rows [ys, ye)
  const uint32_t npx = ye > ys ? (ye - ys) * W : 0u, base = ys * W;
  const size_t f0 = (size_t)f * P;
  if (s == 0 && tid == 0) p.kept[(size_t)f * (p.kept_cap + 1)] = 0;
  const uint8_t* med = p.median + f0 + base;
  for (uint32_t t = tid; t < npx; t += 256) col[t] = med[t];
  __syncthreads();
  // (x of a thread's pixels without a division per pixel: t advances by 256)
  const uint32_t xstep = 256u % W;
  auto next_x = [&](uint32_t x) { return x + xstep >= W ? x + xstep - W : x + xstep; };
  // 1. run labels; a label that points at itself may become a root: reset its statistics
  for (uint32_t t0 = 0, x = tid % W; t0 < npx; t0 += 256, x = next_x(x)) {
    const uint32_t t = t0 + tid;
    const bool live = t < npx;
    const bool in = live && x >= 1 && x + 1 < W;
    const bool left_same = in && x >= 2 && col[t - 1] == col[t];
    const uint64_t heads = __builtin_amdgcn_ballot_w64(!left_same || lane == 0);
    const uint32_t start = 63u - (uint32_t)__builtin_clzll(heads & (~0ull >> (63u - lane)));  // last head at or before me
    if (live) {
      const uint32_t l = in ? t - (lane - start) : FDF_NONE;
      lab[t] = l;
      if (l == t) {
        const size_t g = f0 + base + t;
        p.area[g] = 0;  // (one scattered store per run; the bounding boxes are reset by fdf_stats, for roots only)
      }
    }
  }
  __syncthreads();
  // 2. unions inside the strip
  for (uint32_t t = tid, x = tid % W; t < npx; t += 256, x = next_x(x)) {
    if (x < 1 || x + 1 >= W) continue;
    const uint8_t c = col[t];
    const bool left_same = x >= 2 && col[t - 1] == c;
    // a run that continues across a wave boundary
    if (left_same && lane == 0) fdf_union(lab, t, t - 1);
    // the run above: once per pair of overlapping runs — if my left neighbour continues my run
    // and its upper neighbour continues the upper run, that pixel (or one further left) does it
    if (t >= W && col[t - W] == c && !(left_same && col[t - W - 1] == c)) fdf_union(lab, t, t - W);
  }
  __syncthreads();
  // 3. labels to the strip's roots (frame-relative pixel index)
  for (uint32_t t = tid, x = tid % W; t < npx; t += 256, x = next_x(x))
    p.label[f0 + base + t] = (x >= 1 && x + 1 < W) ? base + fdf_find(lab, t) : FDF_NONE;
}

// First row of every strip but the first against the row above it.
__global__ __launch_bounds__(256) void fdf_stitch(FdfParams p, uint32_t R, uint32_t S) {
  const uint32_t W = p.W, P = p.W * p.H;
  const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t per_frame = (size_t)(S - 1) * W;
  if (g >= (size_t)p.n * per_frame) return;
  const uint32_t f = (uint32_t)(g / per_frame), rest = (uint32_t)(g % per_frame), s = 1 + rest / W, x = rest % W;
  const uint32_t y = 1 + s * R;
  if (!fdf_interior(p, x, y)) return;
  const size_t f0 = (size_t)f * P;
  const uint8_t* med = p.median + f0;
  const uint32_t i = y * W + x;
  const uint8_t c = med[i];
  if (med[i - W] == c && !(x >= 2 && med[i - 1] == c && med[i - W - 1] == c)) fdf_union(p.label + f0, i, i - W);
}

// fdf_stats and fdf_keep walk label -> root -> statistics: chains of dependent loads.  A thread
// takes FDF_PPT pixels (256 apart, so a wave still holds 64 consecutive pixels of each) and
// issues the loads of all of them side by side (per 833 frames, stats / keep: one pixel per
// thread 645 / 247 us, four 251 / 94, eight 226 / 78).
constexpr uint32_t FDF_PPT = 8;

__global__ __launch_bounds__(256) void fdf_stats(FdfParams p) {
  // grid: x = frame, y = chunk of 256 * FDF_PPT pixels of it (no 64-bit divisions per pixel)
  const uint32_t P = p.W * p.H, f = blockIdx.x;
  const size_t fbase = (size_t)f * P;
  size_t g[FDF_PPT], f0[FDF_PPT];
  uint32_t x[FDF_PPT], y[FDF_PPT], a[FDF_PPT];
  bool in[FDF_PPT];
#pragma unroll
  for (uint32_t k = 0; k < FDF_PPT; ++k) {
    const uint32_t i = (blockIdx.y * FDF_PPT + k) * 256 + threadIdx.x;
    const bool live = i < P;
    g[k] = fbase + i;
    y[k] = i / p.W;
    x[k] = i - y[k] * p.W;
    f0[k] = fbase;
    in[k] = live && fdf_interior(p, x[k], y[k]);
    a[k] = in[k] ? i : FDF_NONE;
  }
  // all chains one step at a time
  for (;;) {
    uint32_t q[FDF_PPT];
    bool moved = false;
#pragma unroll
    for (uint32_t k = 0; k < FDF_PPT; ++k) q[k] = in[k] ? p.label[f0[k] + a[k]] : a[k];
#pragma unroll
    for (uint32_t k = 0; k < FDF_PPT; ++k) {
      moved = moved || q[k] != a[k];
      a[k] = q[k];
    }
    if (!moved) break;
  }
  // the root of the workgroup's first pixel is usually the frame's background, which collects
  // over a thousand run lengths per frame on one address: those of this workgroup meet in LDS
  __shared__ size_t s_hot;
  __shared__ uint32_t s_sum;
  if (threadIdx.x < 64) {  // the first interior pixel of the first wave
    const uint64_t any = __builtin_amdgcn_ballot_w64(in[0]);
    const int src = any ? __builtin_ctzll(any) : 0;
    const size_t key = f0[0] + a[0];
    const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)key, src), hi = (uint32_t)__shfl((int)(uint32_t)(key >> 32), src);
    if (threadIdx.x == 0) {
      s_hot = any ? ((size_t)hi << 32 | lo) : ~size_t(0);
      s_sum = 0;
    }
  }
  __syncthreads();
  const int32_t bx = p.pos[2 * f], by = p.pos[2 * f + 1];
  bool differs[FDF_PPT];
#pragma unroll
  for (uint32_t k = 0; k < FDF_PPT; ++k) {
    differs[k] = false;
    if (in[k]) {
      differs[k] = p.frames[g[k]] != p.bg[(size_t)(by + (int32_t)y[k]) * p.bw + (size_t)(bx + (int32_t)x[k])];
    }
  }
#pragma unroll
  for (uint32_t k = 0; k < FDF_PPT; ++k) {
    if (in[k]) {
      p.label[g[k]] = a[k];  // (a pixel only ever shortens its own chain)
      if (differs[k]) atomicOr(&p.area[f0[k] + a[k]], FDF_SEEDED);
      if (a[k] == (uint32_t)(g[k] - f0[k])) {  // a root: its bounding box starts empty
        p.minx[g[k]] = FDF_NONE;
        p.maxx[g[k]] = 0;
        p.miny[g[k]] = FDF_NONE;
        p.maxy[g[k]] = 0;
      }
    }
    // runs of equal roots along a row inside the wave: the run's first lane adds for all.  Only
    // the area here — a frame's big background component takes thousands of these per frame on
    // one address; its bounding box is never needed (fdf_keep adds boxes of kept components only)
    uint32_t len;
    if (fdf_run_head(a[k], x[k], len)) {
      if (f0[k] + a[k] == s_hot)
        atomicAdd(&s_sum, len);  // LDS
      else
        atomicAdd(&p.area[f0[k] + a[k]], len);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0 && s_sum) atomicAdd(&p.area[s_hot], s_sum);
}

__global__ __launch_bounds__(256) void fdf_keep(FdfParams p) {
  const uint32_t P = p.W * p.H, f = blockIdx.x;  // grid as fdf_stats
  const size_t fbase = (size_t)f * P;
  const uint32_t limit = P / 5;  // fde.hpp:109-116
  size_t g[FDF_PPT], f0[FDF_PPT];
  uint32_t x[FDF_PPT], y[FDF_PPT], i[FDF_PPT], root[FDF_PPT];
  bool live[FDF_PPT], keep[FDF_PPT];
#pragma unroll
  for (uint32_t k = 0; k < FDF_PPT; ++k) {
    i[k] = (blockIdx.y * FDF_PPT + k) * 256 + threadIdx.x;
    live[k] = i[k] < P;
    g[k] = fbase + i[k];
    y[k] = i[k] / p.W;
    x[k] = i[k] - y[k] * p.W;
    f0[k] = fbase;
    root[k] = live[k] && fdf_interior(p, x[k], y[k]) ? p.label[g[k]] : FDF_NONE;  // (no labels outside the strips)
  }
  uint32_t area[FDF_PPT];
#pragma unroll
  for (uint32_t k = 0; k < FDF_PPT; ++k) area[k] = root[k] != FDF_NONE ? p.area[f0[k] + root[k]] : 0u;
#pragma unroll
  for (uint32_t k = 0; k < FDF_PPT; ++k) {
    keep[k] = root[k] != FDF_NONE && (area[k] & FDF_SEEDED) && (area[k] & ~FDF_SEEDED) <= limit;
    if (keep[k] && root[k] == i[k]) {
      uint32_t* list = p.kept + (size_t)f * (p.kept_cap + 1);
      const uint32_t at = atomicAdd(&list[0], 1u);
      if (at < p.kept_cap)
        list[1 + at] = i[k];
      else
        atomicOr(&p.status[3], 1u);
    }
    uint32_t len;
    if (fdf_run_head(keep[k] ? root[k] : FDF_NONE, x[k], len)) {  // bounding boxes of the kept components only
      atomicMin(&p.minx[f0[k] + root[k]], x[k]);
      atomicMax(&p.maxx[f0[k] + root[k]], x[k] + len - 1);
      atomicMin(&p.miny[f0[k] + root[k]], y[k]);
      atomicMax(&p.maxy[f0[k] + root[k]], y[k]);
    }
    if (live[k]) p.mask[g[k]] = keep[k] ? 1 : 0;
  }
}

// one workgroup per frame, its sixteen waves take the kept components in turn (four waves: 160 us
// per 833 frames with ~230 kept components each — three waves per SIMD, every one a serial loop)
__global__ __launch_bounds__(1024) void fdf_boxes(FdfParams p) {
  const uint32_t f = blockIdx.x, lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t* list = p.kept + (size_t)f * (p.kept_cap + 1);
  const uint32_t cnt = min(list[0], p.kept_cap);
  if (threadIdx.x == 0 && p.ncontours) p.ncontours[f] = list[0];
  const uint32_t P = p.W * p.H;
  const size_t f0 = (size_t)f * P;
  const uint32_t* L = p.label + f0;
  uint8_t* m = p.mask + f0;
  for (uint32_t slot = wave; slot < cnt; slot += blockDim.x >> 6) {
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
