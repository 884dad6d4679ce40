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
  int32_t* carry;  // device: [0..1] position, [2] fragment id (out)
  uint32_t n;
  uint32_t first_is_start;
  int32_t carry_in[3];  // the carry on entry (by value: no upload)
  int32_t* summary;     // optional, device: the block's scan summary [has a start, x, y, starts] (out)
  // with a summary: the matcher's status word — bit 128 (tickets left to rmx_kpm_resolve) travels
  // as bit 1 of summary[0], so that every rank of a sharded stream learns that SOME rank has to
  // resolve and all of them derive the positions again together
  const uint32_t* kpm_status;
};

// Summary of a run of frames: did it contain a fragment start, the position offset since the
// last start (or since the run's begin), and the number of starts.  Associative.
struct ScanSeg {
  int32_t r, x, y, c;
};
__device__ __forceinline__ ScanSeg seg_join(const ScanSeg& a, const ScanSeg& b) {  // a then b
  return ScanSeg{a.r | b.r, b.r ? b.x : a.x + b.x, b.r ? b.y : a.y + b.y, a.c + b.c};
}
__device__ __forceinline__ ScanSeg seg_shfl_up(const ScanSeg& v, int o) {
  return ScanSeg{__shfl_up(v.r, o), __shfl_up(v.x, o), __shfl_up(v.y, o), __shfl_up(v.c, o)};
}

// One 1024-thread workgroup; the stream is taken in spans of 16 384 frames (16 per thread, held
// in registers: every frame is loaded once, all loads of a thread in flight together) with the
// carry handed from span to span.
constexpr uint32_t SCAN_PER = 16;

__global__ __launch_bounds__(1024) void frc_scan(ScanParams p) {
  __shared__ ScanSeg s_wave[16];
  __shared__ ScanSeg s_carry;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  if (tid == 0) s_carry = ScanSeg{0, p.carry_in[0], p.carry_in[1], p.carry_in[2]};
  for (uint32_t base = 0; base < p.n; base += 1024 * SCAN_PER) {
    const uint32_t span = min(p.n - base, 1024u * SCAN_PER);
    const uint32_t chunk = (span + 1023) / 1024;  // <= SCAN_PER
    const uint32_t i0 = base + min(span, tid * chunk), i1 = base + min(span, tid * chunk + chunk);
    int32_t ddx[SCAN_PER], ddy[SCAN_PER];
    uint32_t brk = 0;  // bit k: frame i0 + k starts a fragment
#pragma unroll
    for (uint32_t k = 0; k < SCAN_PER; ++k) {
      const uint32_t i = i0 + k;
      const bool in = i < i1;
      const uint32_t ic = in ? i : (p.n - 1);
      const uint8_t m = p.matched[ic];
      ddx[k] = p.dxdy[2 * ic];
      ddy[k] = p.dxdy[2 * ic + 1];
      if (in && ((i == 0 && p.first_is_start) || !m)) brk |= 1u << k;
    }
    const uint32_t cntk = i1 - i0;
    int32_t reset = 0, dx = 0, dy = 0, cnt = 0;
#pragma unroll
    for (uint32_t k = 0; k < SCAN_PER; ++k)
      if (k < cntk) {
        if ((brk >> k) & 1u) {
          reset = 1;
          dx = dy = 0;
          cnt += (i0 + k == 0 && p.first_is_start) ? 0 : 1;
        } else {
          dx += ddx[k];
          dy += ddy[k];
        }
      }
    // segmented scan of the 1024 chunk summaries: wave-inclusive by shuffles, the 16 wave
    // totals by one thread, seeded with the carry
    ScanSeg inc{reset, dx, dy, cnt};
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const ScanSeg t = seg_shfl_up(inc, o);
      if (lane >= (uint32_t)o) inc = seg_join(t, inc);
    }
    __syncthreads();  // s_wave / s_carry of the previous span have been read
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    if (tid == 0) {
      ScanSeg acc = s_carry;
      for (uint32_t w = 0; w < 16; ++w) {
        const ScanSeg t = s_wave[w];
        s_wave[w] = acc;
        acc = seg_join(acc, t);
      }
      s_carry = ScanSeg{0, acc.x, acc.y, acc.c};
    }
    __syncthreads();
    const ScanSeg before = seg_shfl_up(inc, 1);
    const ScanSeg st = lane == 0 ? s_wave[wave] : seg_join(s_wave[wave], before);
    int32_t x = st.x, y = st.y, c = st.c;
#pragma unroll
    for (uint32_t k = 0; k < SCAN_PER; ++k)
      if (k < cntk) {
        const uint32_t i = i0 + k;
        if ((brk >> k) & 1u) {
          x = y = 0;
          c += (i == 0 && p.first_is_start) ? 0 : 1;
        } else {
          x += ddx[k];
          y += ddy[k];
        }
        p.pos[2 * i] = x;
        p.pos[2 * i + 1] = y;
        p.frag[i] = (uint32_t)c;
      }
  }
  __syncthreads();
  if (tid == 0) {
    p.carry[0] = s_carry.x;
    p.carry[1] = s_carry.y;
    p.carry[2] = s_carry.c;
    if (p.summary) {
      p.summary[0] = ((p.first_is_start || s_carry.c > 0) ? 1 : 0) | ((p.kpm_status && (*p.kpm_status & 128u)) ? 2 : 0);
      p.summary[1] = s_carry.x;
      p.summary[2] = s_carry.y;
      p.summary[3] = s_carry.c;
    }
  }
}

// Frame-sharded streams: every rank scans its block as if the stream began there and publishes
// the block's summary; this kernel folds the summaries of the blocks before `rank` (the same
// associative rule as the scan) and moves the block into place — positions of the frames that
// continue the previous block's fragment (local fragment 0) by the carried position, all
// fragment numbers by the carried count.  No host round trip between scan, exchange and fix-up.
struct ApplyParams {
  const int32_t* summaries;  // world x 4
  int32_t* pos;
  uint32_t* frag;
  uint32_t* nfrag_out;  // fragments of the whole stream; [1]: some block has tickets left to rmx_kpm_resolve
  uint32_t n, rank, world;
};
__global__ __launch_bounds__(256) void frc_apply(ApplyParams p) {
  int32_t ax = 0, ay = 0;
  uint32_t ac = 0, all = 0, pending = 0;
  for (uint32_t q = 0; q < p.world; ++q) {  // (world is small; every thread folds for itself)
    const int32_t r = p.summaries[4 * q] & 1, x = p.summaries[4 * q + 1], y = p.summaries[4 * q + 2];
    const uint32_t c = (uint32_t)p.summaries[4 * q + 3];
    pending |= (uint32_t)p.summaries[4 * q] & 2u;
    if (q < p.rank) {
      ax = r ? x : ax + x;
      ay = r ? y : ay + y;
      ac += c;
    }
    all += c;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    p.nfrag_out[0] = all + 1;
    p.nfrag_out[1] = pending;
  }
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < p.n; i += gridDim.x * 256) {
    const uint32_t fl = p.frag[i];
    if (fl == 0) {
      p.pos[2 * i] += ax;
      p.pos[2 * i + 1] += ay;
    }
    p.frag[i] = fl + ac;
  }
}

// ---- fgm::fragment::blit(pos, image, …) (fgm.hpp:87-97,176-188) -------------------------------
// ++canvas[y+py-zy][x+px-zx][pixel] with uint16 wraparound, for a whole batch of frames.
// Output-centric: a workgroup OWNS a 128x8 tile of the canvas in LDS (32 B per pixel), walks
// the frames binned to its tile column, and for every frame that covers the tile adds that frame's pixels
// to its own bins — no atomics, every frame byte is read once (by the tile it lands in),
// every canvas dot is read and written once.  Bins wrap modulo 2^16 exactly (fgm.hpp:14).
struct BlitParams {
  const uint8_t* frames;
  const uint8_t* mask;   // optional, same layout as frames: pixels with a non-zero byte are not counted (fgm.hpp:71-86)
  const int32_t* pos;
  const uint32_t* frag_id;
  uint32_t* canvas;  // canvas_h x canvas_w x 8 dwords
  uint32_t n, W, H;
  uint32_t fragment, cw, ch;
  int32_t zx, zy;
  uint32_t tiles_x;
  // frames binned by the tile COLUMNS they overlap (built by blit_bin_* below)
  uint32_t* col_cnt;   // tiles_x + 1: counts, then exclusive offsets (in place)
  uint32_t* col_cur;   // tiles_x fill cursors
  uint32_t* col_list;  // frame ids, <= n * (W / 128 + 2)
};

constexpr int BL_TW = 128, BL_TH = 8;

// A frame overlaps at most W/128 + 2 tile columns; three tiny passes bin the batch's frames
// by tile column so that a tile walks ~(W + 128) / (mean step) frames instead of all n.
// counter[t] += 1 for every active lane, one atomic per distinct t in the wave (neighbouring
// frames fall into the same few columns); returns the lane's own slot (old value + rank).
__device__ __forceinline__ uint32_t wave_claim(uint32_t* counter, int t, bool active) {
  const uint32_t lane = threadIdx.x & 63u;
  uint32_t mine = 0;
  uint64_t todo = __builtin_amdgcn_ballot_w64(active);
  while (todo) {
    const int leader = (int)__builtin_ctzll(todo);
    const int tt = __shfl(t, leader);
    const uint64_t same = __builtin_amdgcn_ballot_w64(active && t == tt);
    uint32_t old = 0;
    if ((int)lane == leader) old = atomicAdd(&counter[tt], (uint32_t)__builtin_popcountll(same));
    old = (uint32_t)__shfl((int)old, leader);
    if (active && t == tt) mine = old + (uint32_t)__builtin_popcountll(same & ((1ull << lane) - 1ull));
    todo &= ~same;
  }
  return mine;
}

__global__ __launch_bounds__(256) void blit_bin_count(BlitParams p) {
  const uint32_t f = blockIdx.x * 256 + threadIdx.x;
  const bool live = f < p.n && !(p.frag_id && p.frag_id[f] != p.fragment);
  int t0 = 0, t1 = -1;
  if (live) {
    const int ax = p.pos[2 * f] - p.zx;
    t0 = max(ax, 0) / BL_TW;
    t1 = min((ax + (int)p.W - 1) / BL_TW, (int)p.tiles_x - 1);
  }
  for (int k = 0; __builtin_amdgcn_ballot_w64(t0 + k <= t1) != 0; ++k) wave_claim(p.col_cnt, t0 + k, t0 + k <= t1);
}
__global__ __launch_bounds__(1024) void blit_bin_scan(BlitParams p) {
  __shared__ uint32_t part[1024];
  const uint32_t tid = threadIdx.x, n = p.tiles_x;
  const uint32_t chunk = (n + 1023) / 1024, i0 = min(n, tid * chunk), i1 = min(n, i0 + chunk);
  uint32_t s = 0;
  for (uint32_t i = i0; i < i1; ++i) s += p.col_cnt[i];
  part[tid] = s;
  __syncthreads();
  if (tid == 0) {
    uint32_t acc = 0;
    for (uint32_t t = 0; t < 1024; ++t) {
      uint32_t v = part[t];
      part[t] = acc;
      acc += v;
    }
    p.col_cnt[n] = acc;
  }
  __syncthreads();
  uint32_t acc = part[tid];
  for (uint32_t i = i0; i < i1; ++i) {
    uint32_t v = p.col_cnt[i];
    p.col_cnt[i] = acc;
    p.col_cur[i] = acc;
    acc += v;
  }
}
__global__ __launch_bounds__(256) void blit_bin_fill(BlitParams p) {
  const uint32_t f = blockIdx.x * 256 + threadIdx.x;
  const bool live = f < p.n && !(p.frag_id && p.frag_id[f] != p.fragment);
  int t0 = 0, t1 = -1;
  if (live) {
    const int ax = p.pos[2 * f] - p.zx;
    t0 = max(ax, 0) / BL_TW;
    t1 = min((ax + (int)p.W - 1) / BL_TW, (int)p.tiles_x - 1);
  }
  for (int k = 0; __builtin_amdgcn_ballot_w64(t0 + k <= t1) != 0; ++k) {
    const bool act = t0 + k <= t1;
    const uint32_t at = wave_claim(p.col_cur, t0 + k, act);
    if (act) p.col_list[at] = f;
  }
}

__device__ __forceinline__ uint32_t add_u16x2(uint32_t a, uint32_t b) {
  return ((a + b) & 0xffffu) | (((a >> 16) + (b >> 16)) << 16);
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t blit_rsrc(const uint8_t* base, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(base), (short)0, (int)bytes, 0x00020000);
}

__global__ __launch_bounds__(256) void fgm_blit(BlitParams p) {
  // bins[k][slot]: dword k (bins 2k, 2k+1) of the pixel in `slot`; a thread's pixel j sits in
  // slot row*128 + j*32 + (tid & 31), so for any k the 32 lanes of a half-wave hit 32
  // different banks whatever colours they carry (a [pixel][8] layout is a 32-way conflict)
  __shared__ uint32_t bins[9][BL_TW * BL_TH];  // 32 KB + a ninth row nobody reads: pixels outside a frame or masked out land there
  __shared__ uint32_t list[256];
  __shared__ int list_ax[256], list_ay[256];  // canvas origin of the listed frames
  __shared__ uint32_t nlist;
  const uint32_t tid = threadIdx.x;
  const int tx0 = (int)(blockIdx.x % p.tiles_x) * BL_TW, ty0 = (int)(blockIdx.x / p.tiles_x) * BL_TH;
  const int row = (int)(tid >> 5), xq = (int)(tid & 31u) * 4;  // this thread: 4 pixels of one row
  const int X = tx0 + xq, Y = ty0 + row;                      // canvas coordinates
  const uint32_t slot0 = (uint32_t)row * BL_TW + (tid & 31u);
  // The bins hold only THIS launch's increments (at most 65 535 frames per launch, so no half
  // of a word can overflow into the other): a pixel is then one fire-and-forget ds_add_u32,
  // no read-modify-write chain, and the canvas content is added once at the end with its
  // uint16 wraparound (fgm.hpp:14).
  const bool in_canvas_row = Y < (int)p.ch;
#pragma unroll
  for (int k = 0; k < 8; ++k)
#pragma unroll
    for (int j = 0; j < 4; ++j) bins[k][slot0 + (uint32_t)j * 32] = 0;
  const int W = (int)p.W, H = (int)p.H;
  const int YW = Y * W;
  const uint32_t tcol = blockIdx.x % p.tiles_x;
  const uint32_t lbeg = p.col_cnt[tcol], lend = p.col_cnt[tcol + 1];  // this tile column's frames
  for (uint32_t base = lbeg; base < lend; base += 256) {
    __syncthreads();
    if (tid == 0) nlist = 0;
    __syncthreads();
    if (base + tid < lend) {
      const uint32_t f = p.col_list[base + tid];
      const int ax = p.pos[2 * f] - p.zx, ay = p.pos[2 * f + 1] - p.zy;  // x overlap is guaranteed by the column list
      if (ay < ty0 + BL_TH && ay + H > ty0) {
        const uint32_t at = atomicAdd(&nlist, 1u);
        list[at] = f;
        list_ax[at] = ax;
        list_ay[at] = ay;
      }
    }
    __syncthreads();
    const uint32_t m = nlist;
    // fetch() only ISSUES the loads of one frame's four pixels (and mask bytes): every use of
    // the loaded dwords is in apply(), so the loads of a whole trip are in flight together.
    // A dword that would straddle the frame's left/right edge is loaded from the clamped
    // position and shifted into place afterwards (sh = -3..3); sh = 8: nothing to add.
    auto fetch = [&](uint32_t i, uint32_t& raw, uint32_t& mraw, int& sh) {
      raw = 0;
      mraw = 0;
      sh = 8;
      if (i >= m) return;  // (uniform)
      // the frame and its canvas origin are the same for the whole workgroup: scalar registers, a
      // buffer descriptor per frame.  Rows above or below the frame give offsets outside the
      // descriptor (the hardware returns 0, no branch); they are dropped through sh all the same
      const uint32_t g = (uint32_t)__builtin_amdgcn_readfirstlane((int)list[i]);
      const int ax = __builtin_amdgcn_readfirstlane(list_ax[i]), ay = __builtin_amdgcn_readfirstlane(list_ay[i]);
      const size_t fbase = (size_t)g * (size_t)(H * W);
      const int fx = X - ax, fy = Y - ay;
      const int fxc = min(max(fx, 0), W - 4);
      const uint32_t voff = (uint32_t)(YW - ay * W + fxc);
      raw = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(blit_rsrc(p.frames + fbase, (uint32_t)(H * W)), voff, 0, 0);
      if (p.mask) mraw = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(blit_rsrc(p.mask + fbase, (uint32_t)(H * W)), voff, 0, 0);
      const bool inside = (uint32_t)fy < (uint32_t)H && (uint32_t)(fx + 3) < (uint32_t)(W + 3);
      sh = inside ? fx - fxc : 8;
    };
    auto place = [](uint32_t w, int sh, uint32_t fill) {  // bytes of the clamped dword -> pixel order, `fill` elsewhere
      if (sh < 0) return (w << (8 * -sh)) | (fill >> (32 + 8 * sh));
      if (sh > 0) return (w >> (8 * sh)) | (fill << (32 - 8 * sh));
      return w;
    };
    auto apply = [&](uint32_t raw, uint32_t mraw, int sh) {
      if (sh == 8) return;
      uint32_t w = place(raw & 0x0f0f0f0fu, sh, 0x10101010u);  // 0x10 in a byte = "outside the frame"
      if (p.mask) {
        const uint32_t mw = place(mraw, sh, 0u);
        const uint32_t nz = (((mw & 0x7f7f7f7fu) + 0x7f7f7f7fu) | mw) & 0x80808080u;  // bit 7 of every non-zero byte
        const uint32_t out = nz >> 3;                                                  // 0x10 per masked pixel
        w = (w & ~(out - (out >> 4))) | out;                                           // masked pixel -> "outside"
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        // branch-free, five VALU ops per pixel: colour c adds 1 to half c & 1 of dword c >> 1;
        // "outside" (16) adds to the ninth row (a compare-and-select to add 0 instead: ten)
        const uint32_t v = (w >> (8 * j)) & 0x1fu;
        const uint32_t inc = 1u + (v & 1u) * 0xffffu;
        __hip_atomic_fetch_add(&bins[v >> 1][slot0 + (uint32_t)j * 32], inc, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    };
    // eight frames per trip, the next trip's loads issued before this trip's updates
    uint32_t cur[8], curm[8], nxt[8], nxtm[8];
    int csh[8], nsh[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) fetch((uint32_t)k, cur[k], curm[k], csh[k]);
    for (uint32_t i = 0; i < m; i += 8) {
#pragma unroll
      for (int k = 0; k < 8; ++k) fetch(i + 8 + (uint32_t)k, nxt[k], nxtm[k], nsh[k]);
#pragma unroll
      for (int k = 0; k < 8; ++k) apply(cur[k], curm[k], csh[k]);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        cur[k] = nxt[k];
        curm[k] = nxtm[k];
        csh[k] = nsh[k];
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (in_canvas_row && X + j < (int)p.cw) {
      const uint32_t sl = slot0 + (uint32_t)j * 32;
      uint4* dst = reinterpret_cast<uint4*>(p.canvas + ((size_t)Y * p.cw + (size_t)(X + j)) * 8);
      const uint4 a = dst[0], b = dst[1];
      dst[0] = make_uint4(add_u16x2(a.x, bins[0][sl]), add_u16x2(a.y, bins[1][sl]), add_u16x2(a.z, bins[2][sl]),
                          add_u16x2(a.w, bins[3][sl]));
      dst[1] = make_uint4(add_u16x2(b.x, bins[4][sl]), add_u16x2(b.y, bins[5][sl]), add_u16x2(b.z, bins[6][sl]),
                          add_u16x2(b.w, bins[7][sl]));
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
