// fgm.hip — fragment canvas kernels (reference fgm.hpp) and the frc position scan.
#include "common.h"

namespace rmx {

// ---- frc::collector::process_frame state update (frc.hpp:109-115,124-127) ---------------------
// pos[f] = matched[f] ? pos[f-1] + dxdy[f] : (0,0);  frag[f] = frag[f-1] + !matched[f]
// (frame 0 of a stream starts fragment 0 at (0,0), frc.hpp:83-95).  Segmented scan with the
// associative operator (reset, sum, breaks); one 1024-thread block, contiguous chunks.
struct ScanParams {
  const int32_t* dxdy;
  const uint8_t* matched;
  int32_t* pos;
  uint32_t* frag;
  int32_t* carry;  // device: [0..1] position, [2] fragment id (in/out)
  uint32_t n;
  uint32_t first_is_start;
};

__global__ __launch_bounds__(1024) void frc_scan(ScanParams p) {
  __shared__ int32_t s_reset[1024], s_dx[1024], s_dy[1024], s_cnt[1024];
  const uint32_t tid = threadIdx.x;
  const uint32_t chunk = (p.n + 1023) / 1024;
  const uint32_t i0 = min(p.n, tid * chunk), i1 = min(p.n, i0 + chunk);
  int32_t reset = 0, dx = 0, dy = 0, cnt = 0;
  for (uint32_t i = i0; i < i1; ++i) {
    bool brk = (i == 0 && p.first_is_start) || !p.matched[i];
    if (brk) {
      reset = 1;
      dx = dy = 0;
      cnt += (i == 0 && p.first_is_start) ? 0 : 1;
    } else {
      dx += p.dxdy[2 * i];
      dy += p.dxdy[2 * i + 1];
    }
  }
  s_reset[tid] = reset;
  s_dx[tid] = dx;
  s_dy[tid] = dy;
  s_cnt[tid] = cnt;
  __syncthreads();
  if (tid == 0) {  // exclusive scan of 1024 chunk summaries, seeded with the carry
    int32_t ax = p.carry[0], ay = p.carry[1], ac = p.carry[2];
    for (uint32_t t = 0; t < 1024; ++t) {
      int32_t r = s_reset[t], x = s_dx[t], y = s_dy[t], c = s_cnt[t];
      s_dx[t] = ax;
      s_dy[t] = ay;
      s_cnt[t] = ac;
      ax = r ? x : ax + x;
      ay = r ? y : ay + y;
      ac += c;
    }
    p.carry[0] = ax;
    p.carry[1] = ay;
    p.carry[2] = ac;
  }
  __syncthreads();
  int32_t x = s_dx[tid], y = s_dy[tid], c = s_cnt[tid];
  for (uint32_t i = i0; i < i1; ++i) {
    bool start = (i == 0 && p.first_is_start);
    if (start || !p.matched[i]) {
      x = y = 0;
      c += start ? 0 : 1;
    } else {
      x += p.dxdy[2 * i];
      y += p.dxdy[2 * i + 1];
    }
    p.pos[2 * i] = x;
    p.pos[2 * i + 1] = y;
    p.frag[i] = (uint32_t)c;
  }
}

// ---- fgm::fragment::blit(pos, image, …) (fgm.hpp:87-97,176-188) -------------------------------
// ++canvas[y+py-zy][x+px-zx][pixel] with uint16 wraparound.  Bins are packed two per dword;
// the add is a 32-bit atomic on the containing dword and a carry out of the low half is
// taken back, which keeps every bin exact modulo 2^16 (fgm.hpp:14).
struct BlitParams {
  const uint8_t* frames;
  const int32_t* pos;
  const uint32_t* frag_id;
  uint32_t* canvas;  // canvas_h x canvas_w x 8 dwords
  uint32_t n, W, H, Wd;
  uint32_t fragment, cw, ch;
  int32_t zx, zy;
};

__global__ __launch_bounds__(256) void fgm_blit(BlitParams p) {
  const uint32_t f = blockIdx.y;
  if (p.frag_id && p.frag_id[f] != p.fragment) return;
  const int64_t ax = (int64_t)p.pos[2 * f] - p.zx, ay = (int64_t)p.pos[2 * f + 1] - p.zy;
  const uint8_t* src = p.frames + (size_t)f * p.W * p.H;
  const uint32_t total = p.Wd * p.H;
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    uint32_t y = i / p.Wd, xd = i - y * p.Wd;
    uint32_t w = *reinterpret_cast<const uint32_t*>(src + (size_t)y * p.W + xd * 4);
    uint32_t* row = p.canvas + ((size_t)(ay + y) * p.cw + (size_t)(ax + xd * 4)) * 8;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      uint32_t v = (w >> (8 * j)) & 0xfu;
      uint32_t* word = row + j * 8 + (v >> 1);
      if (v & 1u) {
        atomicAdd(word, 0x10000u);  // high half: the carry leaves the dword
      } else {
        uint32_t old = atomicAdd(word, 1u);
        if ((old & 0xffffu) == 0xffffu) atomicSub(word, 0x10000u);  // undo the carry into the high bin
      }
    }
  }
}

// ---- fgm::fragment::blit(pos, fragment&&) canvas part (fgm.hpp:99-113) ------------------------
struct MergeParams {
  uint32_t* dst;
  const uint32_t* src;
  uint32_t dw, dh, sw, sh;
  int32_t ox, oy;
};

__device__ __forceinline__ uint32_t add_u16x2(uint32_t a, uint32_t b) {
  return ((a + b) & 0xffffu) | (((a >> 16) + (b >> 16)) << 16);
}

__global__ __launch_bounds__(256) void fgm_merge(MergeParams p) {
  // one thread per (source pixel, half dot): 4 dwords = 8 bins
  const uint64_t total = (uint64_t)p.sw * p.sh * 2;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (uint64_t)gridDim.x * 256) {
    uint64_t px = i >> 1;
    uint32_t half = (uint32_t)(i & 1);
    uint32_t y = (uint32_t)(px / p.sw), x = (uint32_t)(px - (uint64_t)y * p.sw);
    const uint4 s = reinterpret_cast<const uint4*>(p.src)[px * 2 + half];
    uint4* dp = reinterpret_cast<uint4*>(p.dst) + (((uint64_t)(y + p.oy)) * p.dw + (uint64_t)(x + p.ox)) * 2 + half;
    uint4 d = *dp;
    d.x = add_u16x2(d.x, s.x);
    d.y = add_u16x2(d.y, s.y);
    d.z = add_u16x2(d.z, s.z);
    d.w = add_u16x2(d.w, s.w);
    *dp = d;
  }
}

// ---- fgm::fragment::blend (fgm.hpp:115-135) -----------------------------------------------------
struct BlendParams {
  const uint32_t* canvas;
  uint8_t* image;
  uint8_t* mask;
  uint64_t npix;
};

__global__ __launch_bounds__(256) void fgm_blend(BlendParams p) {
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < p.npix; i += (uint64_t)gridDim.x * 256) {
    const uint4* d = reinterpret_cast<const uint4*>(p.canvas) + i * 2;
    uint4 a = d[0], b = d[1];
    uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    uint32_t best = 0, arg = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      uint32_t v = (w[k >> 1] >> (16 * (k & 1))) & 0xffffu;
      if (v > best) {  // strict: first maximum wins (std::max_element)
        best = v;
        arg = (uint32_t)k;
      }
    }
    p.image[i] = best ? (uint8_t)arg : 0;
    p.mask[i] = best ? 1 : 0;
  }
}

}  // namespace rmx
