// fgm_bs.hip — fgm::fragment::blit (reference fgm.hpp:87-97: bins[y][x][colour]++ for every pixel of every frame) on
// bit planes.
//
// fgm_blit (fgm.hip) counts a pixel with four lane operations (a 64-bit shift of a one by 4 x colour, two adds, the
// LDS read) plus its share of the folds and of the per-frame scalar work: ~14 wave instructions per 64 pixels, and
// it is bound by the instructions it issues.  Here a lane operation works on SIXTEEN pixels of one
// frame row and TWO colours: a 32-bit word holds, for the lane's 16 canvas columns, one bit per pixel of "colour c"
// in its low half and of "colour c + 8" in its high half — eight such words are the one-hot image of a frame's 16
// pixels, and the per-dot, per-colour counters are eight words deep (bit k of the count of 32 (pixel, colour)
// slots per word: "vertical" counters, 64 registers for 16 pixels x 16 colours x 8 bits).  Per frame and lane:
//
//   one unaligned 16-byte load (the frame row's 16 bytes under the lane's canvas columns, clamped into the row)
//   bytes -> planes, two frames per transposition (bitslice.h planes_from_bytes: 36 operations per pair)
//   edge frames only: the planes shifted by the clamp (v_pk_lsh*_b16, per frame), the validity mask
//   planes duplicated into both halves (4 v_perm), one-hot words: (q3 ^ 0x0000ffff) & mask, x q2, x (q0, q1): 11
//   counters: a carry-save tree — full adder (ones, h1, h2) per pair of frames, (twos, a, b) per two pairs,
//   (fours, ..) per four, then a ripple into bits 3..7 (RMX_BB_PEND8=1: one more level of waiting carries, a ripple
//   into bits 4..7 every 16 frames): ~2.5-3 operations per word and frame
//
// ~75 lane operations for 16 pixels against ~220 in fgm_blit.  A wave owns 64 canvas columns x 16 rows (lane:
// row lane / 4, columns 16 (lane % 4)), a workgroup four such tiles side by side on the same binned frame list.
// The list is taken 256 positions at a time: a first pass tests the records against the tile (a lane per record) and
// queues the matching frames in the wave's LDS; the second pass runs GROUPS OF EIGHT PAIRS in straight-line code —
// ring slot and tree position of a pair are compile-time, so no counter update sits in a conditional block (with
// run-time positions the register allocator shuffled the 88 counter registers at every join) — with D pairs in
// flight through an LDS-DMA ring; the queue is padded with frames that count nothing.  The workgroup's waves meet
// at a barrier once per 16 list positions, each where its own queue crosses into the window, as in fgm_blit
// (neighbours ask for a line of a frame row within a few frames of each other).  After 240 frames at most, and at
// the end, the 8 x 8 planes are transposed in place (three delta-swap stages: byte b of word i = the count of slot
// 8 b + i), widened to the canvas's uint16 bins with one v_perm per dword and added to / written over the canvas in
// 64-byte pieces (two dots of a lane per phase, turned round through LDS — the ring's memory, idle then — so that
// four lanes write one piece).  The flush is a CALL with the planes passed by value: its register allocation is its
// own, and what it spills stays out of the frame loop, where a spill store would void the hand-counted vmcnt.
//
// Status: bit-exact, 0.6 of fgm_blit's instructions, and slower — 150-250 registers leave two waves
// per SIMD, too few to hide the dependent-issue latency of a pair's ~180 instructions.  Behind RMX_BLIT_BS=1.
//
// Frames with a mask and frames narrower than 64 pixels stay with fgm.hip's kernels.
#pragma once

namespace rmx {

constexpr int BB_ROWS = 16;                   // rows of a wave's tile
constexpr uint32_t BB_XSTRIDE = 80;           // bytes between two lanes' 64-byte chunks in the flush buffer (bank spread)
constexpr uint32_t BB_FLUSH_PAIRS = 120;      // 240 frames: the 8-bit counters hold 255
#ifndef RMX_BB_DEPTH
#define RMX_BB_DEPTH 4                        // pairs of frames in a wave's pipe (LDS-DMA ring: 2 KB per pair and wave); divides 8
#endif
constexpr int BB_DEPTH = RMX_BB_DEPTH;
#ifndef RMX_BB_PEND8
#define RMX_BB_PEND8 0                        // a third level of waiting carries (weight 8)
#endif
#ifndef RMX_BB_WGS
#define RMX_BB_WGS 2                          // workgroups per CU the register budget is set for
#endif

typedef unsigned short bb_u16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int bb_u32x4 __attribute__((__vector_size__(4 * sizeof(unsigned int))));

__device__ __forceinline__ uint32_t bb_pk_shl(uint32_t v, uint32_t s) {
  return __builtin_bit_cast(uint32_t, (bb_u16x2)(__builtin_bit_cast(bb_u16x2, v) << __builtin_bit_cast(bb_u16x2, s)));
}
__device__ __forceinline__ uint32_t bb_pk_shr(uint32_t v, uint32_t s) {
  return __builtin_bit_cast(uint32_t, (bb_u16x2)(__builtin_bit_cast(bb_u16x2, v) >> __builtin_bit_cast(bb_u16x2, s)));
}

// one-hot truth tables: (q0 == B0) & (q1 == B1) & c
template <int B0, int B1>
struct f_bb_hot {
  constexpr uint32_t operator()(uint32_t a, uint32_t b, uint32_t c) const { return (B0 ? a : ~a) & (B1 ? b : ~b) & c; }
};
struct f_bb_sel3 {  // (q3 ^ K) & m
  constexpr uint32_t operator()(uint32_t q3, uint32_t k, uint32_t m) const { return (q3 ^ k) & m; }
};
struct f_bb_andn {  // ~a & b
  constexpr uint32_t operator()(uint32_t a, uint32_t b, uint32_t) const { return ~a & b; }
};

// the eight one-hot words of a frame's 16 pixels: q[k] = plane of colour bit k in BOTH halves, m = valid pixels in both halves
__device__ __forceinline__ void bb_onehot(const uint32_t (&q)[4], uint32_t m, uint32_t (&h)[8]) {
  const uint32_t t3 = bs::bop3<f_bb_sel3>(q[3], 0x0000ffffu, m);  // low half: colours 0..7 (bit 3 clear), high half: 8..15
  const uint32_t z0 = bs::bop3<f_bb_andn>(q[2], t3, t3), z1 = q[2] & t3;
  h[0] = bs::bop3<f_bb_hot<0, 0>>(q[0], q[1], z0);
  h[1] = bs::bop3<f_bb_hot<1, 0>>(q[0], q[1], z0);
  h[2] = bs::bop3<f_bb_hot<0, 1>>(q[0], q[1], z0);
  h[3] = bs::bop3<f_bb_hot<1, 1>>(q[0], q[1], z0);
  h[4] = bs::bop3<f_bb_hot<0, 0>>(q[0], q[1], z1);
  h[5] = bs::bop3<f_bb_hot<1, 0>>(q[0], q[1], z1);
  h[6] = bs::bop3<f_bb_hot<0, 1>>(q[0], q[1], z1);
  h[7] = bs::bop3<f_bb_hot<1, 1>>(q[0], q[1], z1);
}

struct BbCounters {
  uint32_t pl[8][8];                // pl[k][w]: bit k of the counts of word w's 32 (pixel, colour) slots
  uint32_t pa[8], pA[8];            // carries waiting for their partner: weight 2, 4
#if RMX_BB_PEND8
  uint32_t pE[8];                   // weight 8
#endif
  uint32_t np;                      // pairs counted since the last flush (uniform; a multiple of eight between groups)
  bool wrote;                       // this tile's dots have been stored once (uniform)

  __device__ __forceinline__ void clear() {
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
      for (int w = 0; w < 8; ++w) pl[k][w] = 0;
    np = 0;
    wrote = false;
  }
  // A carry of weight 2 per word joins the tree.  K = number of the pair within its group of eight (compile time:
  // which levels meet their partner is straight-line code — with run-time tests the register allocator shuffled
  // the 88 counter registers at every join, a hundred moves per pair).
  template <int K>
  __device__ __forceinline__ void carry_in(const uint32_t (&c2)[8]) {
    if constexpr ((K & 1) == 0) {
#pragma unroll
      for (int w = 0; w < 8; ++w) pa[w] = c2[w];
    } else {
      uint32_t c4[8];
#pragma unroll
      for (int w = 0; w < 8; ++w) {
        const bs::Sum2 s = bs::full_add(pl[1][w], pa[w], c2[w]);
        pl[1][w] = s.s, c4[w] = s.c;
      }
      if constexpr ((K & 2) == 0) {
#pragma unroll
        for (int w = 0; w < 8; ++w) pA[w] = c4[w];
      } else {
        uint32_t c8[8];
#pragma unroll
        for (int w = 0; w < 8; ++w) {
          const bs::Sum2 s = bs::full_add(pl[2][w], pA[w], c4[w]);
          pl[2][w] = s.s, c8[w] = s.c;
        }
#if RMX_BB_PEND8
        if constexpr ((K & 4) == 0) {
#pragma unroll
          for (int w = 0; w < 8; ++w) pE[w] = c8[w];
        } else {
#pragma unroll
          for (int w = 0; w < 8; ++w) {
            const bs::Sum2 s = bs::full_add(pl[3][w], pE[w], c8[w]);
            pl[3][w] = s.s;
            uint32_t cy = s.c;  // weight 16: ripples through bits 4..7
#pragma unroll
            for (int k = 4; k < 8; ++k) {
              const uint32_t nx = pl[k][w] & cy;
              pl[k][w] ^= cy;
              cy = nx;
            }
          }
        }
#else
        // (no third level of waiting carries: eight registers fewer, nine operations more per pair — what lets three
        // waves share a SIMD)
#pragma unroll
        for (int w = 0; w < 8; ++w) {
          uint32_t cy = c8[w];  // weight 8: ripples through bits 3..7
#pragma unroll
          for (int k = 3; k < 8; ++k) {
            const uint32_t nx = pl[k][w] & cy;
            pl[k][w] ^= cy;
            cy = nx;
          }
        }
#endif
      }
    }
  }
  // two frames' one-hot words
  template <int K>
  __device__ __forceinline__ void add_pair(const uint32_t (&h1)[8], const uint32_t (&h2)[8]) {
    uint32_t c2[8];
#pragma unroll
    for (int w = 0; w < 8; ++w) {
      const bs::Sum2 s = bs::full_add(pl[0][w], h1[w], h2[w]);
      pl[0][w] = s.s, c2[w] = s.c;
    }
    carry_in<K>(c2);
  }
};

// counters -> canvas.  xb: BB_XSTRIDE * 64 bytes of the wave's LDS.
// (NOT inlined, the counters passed by value: the flush gets its own register allocation, and what it spills stays
// out of the frame loop — spill stores there would void the loop's hand-counted vmcnt, which takes the loads to
// complete in order)
struct BbPlanes {
  uint32_t pl[8][8];
};
struct BbCanvas {
  uint32_t* canvas;
  uint32_t cw, ch;
};
__device__ __attribute__((noinline)) void bb_flush_planes(BbCanvas p, BbPlanes C, bool rd, int X0, int Y0, uint32_t lane, uint8_t* xb) {
  // 8 x 8 bit transposition per byte column, in place: afterwards byte b of pl[i][w] = count of slot 8 b + i of word w
#pragma unroll
  for (int w = 0; w < 8; ++w) {
#pragma unroll
    for (int k = 0; k < 4; ++k) bs::delta_swap(C.pl[k][w], C.pl[k + 4][w], 4, 0x0f0f0f0fu);
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if ((k & 2) == 0) bs::delta_swap(C.pl[k][w], C.pl[k + 2][w], 2, 0x33333333u);
#pragma unroll
    for (int k = 0; k < 8; k += 2) bs::delta_swap(C.pl[k][w], C.pl[k + 1][w], 1, 0x55555555u);
  }
  // The tile's 16 rows through a buffer descriptor at its first dot.  A phase moves two dots (64 bytes) of every
  // lane; through LDS four lanes share one such chunk, 16 bytes each, so that whole 64-byte pieces of a row cross
  // the memory interface instead of 16 bytes per lane at a stride of 512: instruction j serves chunks 16 j .. 16 j
  // + 15 (chunk c = lane c's dots).
  const uint32_t row_bytes = p.cw * 32u;
  const __amdgpu_buffer_rsrc_t crs = __builtin_amdgcn_make_buffer_rsrc(
      p.canvas + ((size_t)Y0 * p.cw + (size_t)X0) * 8, (short)0, (int)min((uint64_t)BB_ROWS * row_bytes, (uint64_t)0x7fffffffu), 0x00020000);
  auto wsync = [&]() __attribute__((always_inline)) {  // (LDS operations of a wave execute in order; this keeps the compiler from moving them)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };
  bb_u32x4* mine = reinterpret_cast<bb_u32x4*>(xb + lane * BB_XSTRIDE);  // my chunk (as its owner)
  const uint32_t piece = lane & 3u;
  // memory side, instruction j of a phase, this lane: chunk c = 16 j + lane / 4 (row c / 4, dots 16 (c % 4) + 2 t ..),
  // piece lane % 4 (dot piece / 2 of the two); phase t adds 64 bytes
  uint32_t vo[4], lo[4];
  bool rowin[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint32_t c = 16u * j + (lane >> 2), crow = c >> 2;
    rowin[j] = Y0 + (int)crow < (int)p.ch;
    vo[j] = crow * row_bytes + ((c & 3u) * 16u + (piece >> 1)) * 32u + (piece & 1u) * 16u;
    lo[j] = c * BB_XSTRIDE + piece * 16u;
  }
  const int xdot = X0 + (int)((lane >> 2) & 3u) * 16 + (int)(piece >> 1);  // (the chunk's column does not depend on j)
#pragma unroll
  for (int t = 0; t < 8; ++t) {  // dots 2 t, 2 t + 1 of every lane
    uint32_t va[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) va[j] = rowin[j] && xdot + 2 * t < (int)p.cw ? vo[j] + 64u * t : 0x80000000u;
    bb_u32x4 old[4];
    if (rd) {
#pragma unroll
      for (int j = 0; j < 4; ++j) *reinterpret_cast<bb_u32x4*>(xb + lo[j]) = __builtin_amdgcn_raw_buffer_load_b128(crs, va[j], 0, 0);
      wsync();
#pragma unroll
      for (int j = 0; j < 4; ++j) old[j] = mine[j];
      wsync();
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {  // dot x = 2 t + u of my chunk: slot index i = x & 7, byte x >> 3 (+ 2 for colours 8..15)
      const int i = (2 * t + u) & 7, b0 = (2 * t + u) >> 3;
      uint32_t o[8];
#pragma unroll
      for (int m = 0; m < 8; ++m) {  // bins 2 m, 2 m + 1
        const int w = (2 * m) & 7, b = b0 + 2 * (m >> 2);
        o[m] = __builtin_amdgcn_perm(C.pl[i][w + 1], C.pl[i][w], 0x0c000c00u | (uint32_t)(4 + b) << 16 | (uint32_t)b);
      }
      if (rd) {
#pragma unroll
        for (int m = 0; m < 8; ++m) o[m] = add_u16x2(o[m], old[2 * u + (m >> 2)][m & 3]);
      }
      mine[2 * u] = bb_u32x4{o[0], o[1], o[2], o[3]};
      mine[2 * u + 1] = bb_u32x4{o[4], o[5], o[6], o[7]};
    }
    wsync();
    bb_u32x4 out[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) out[j] = *reinterpret_cast<const bb_u32x4*>(xb + lo[j]);
    wsync();
#pragma unroll
    for (int j = 0; j < 4; ++j) __builtin_amdgcn_raw_buffer_store_b128(out[j], crs, va[j], 0, 0);
  }
  // (the frame pipe counts its LDS-DMA loads by hand: nothing of this flush may still be in flight behind them)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
__device__ __forceinline__ void bb_flush(const BlitParams& p, BbCounters& C, int X0, int Y0, uint32_t lane, uint8_t* xb) {
  BbPlanes P;
#pragma unroll
  for (int k = 0; k < 8; ++k)
#pragma unroll
    for (int w = 0; w < 8; ++w) P.pl[k][w] = C.pl[k][w];
  bb_flush_planes(BbCanvas{p.canvas, p.cw, p.ch}, P, !p.fresh || C.wrote, X0, Y0, lane, xb);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  C.clear();
  C.wrote = true;
}

constexpr uint32_t BB_CHUNK = 256;              // list positions per pass: 16 windows of 16
constexpr uint32_t BB_QCAP = BB_CHUNK + 16;     // a chunk's frames, padded to a multiple of 16
constexpr uint32_t BB_NULL_LO = 0u, BB_NULL_HI = 0x80000000u;  // the frame that counts nothing: fy = -32768 (its loads read frame 0's first row)

__global__ __launch_bounds__(256, RMX_BB_WGS) void fgm_blit_bs(BlitParams p) {
  constexpr int D = BB_DEPTH;
  constexpr uint32_t RING_BYTES = D * 2 * 64 * 16, XBUF_BYTES = 64 * BB_XSTRIDE;
  // [wave][pair in flight][frame of the pair][lane's 16 bytes]; the flush turns the canvas rows round in the same
  // memory (no pair is in flight then: the pipe holds its breath in the group before a flush)
  __shared__ __attribute__((aligned(16))) uint8_t ring_mem[4][RING_BYTES > XBUF_BYTES ? RING_BYTES : XBUF_BYTES];
  __shared__ __attribute__((aligned(16))) uint2 queue[4][BB_QCAP];  // the chunk's frames that touch this wave's tile
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint32_t b = blit_tile_of_block();
  const uint32_t group = b % p.groups_x, tcol = group * 4 + wv;
  // Waves without a tile leave BEFORE the first barrier (s_barrier counts only the waves that have not ended); every
  // wave that stays walks the SAME list and executes exactly sixteen barriers per chunk of 256 records, whatever its
  // tile matches (the invariant of fgm_blit: no per-wave exit or per-wave list below).
  if (tcol >= p.tiles_x) return;
  const int Y0 = (int)(b / p.groups_x) * BB_ROWS, X0 = (int)(tcol * BL_TW);
  const int W = (int)p.W, H = (int)p.H;
  const uint64_t fbytes = (uint64_t)H * (uint64_t)W;
  const int lrow = (int)(lane >> 2), lcol = (int)(lane & 3u) * 16;
  const uint32_t lane_off = (uint32_t)(lrow * W + lcol);
  uint2* q = queue[wv];
  uint8_t* ring = ring_mem[wv];

  BbCounters C;
  C.clear();

  // A queue record (frame | window << 28, sx & 0xffff | fy << 16: the tile's origin in frame coordinates), uniform.
  // The lane's 16 bytes are loaded from inside the frame's row (window clamped), `l` / `r` say how far the planes
  // have to move to undo the clamp, `m` marks the lane's pixels that lie in the frame (both halves).
  auto covers = [&](uint32_t d1) __attribute__((always_inline)) -> bool {  // (uniform) the frame covers the whole tile
    const int sx = (int)(short)(d1 & 0xffffu), fy = (int)d1 >> 16;
    return sx >= 0 && sx + BL_TW <= W && fy >= 0 && fy + BB_ROWS <= H;
  };
  auto load_one = [&](uint32_t d0, uint32_t d1, uint8_t* dst) __attribute__((always_inline)) {
    const int sx = (int)(short)(d1 & 0xffffu), fy = (int)d1 >> 16;
    const uint8_t* fbase = p.frames + (uint64_t)(d0 & 0x0fffffffu) * fbytes;
    uint32_t off;
    if (covers(d1)) {
      off = (uint32_t)(fy * W + sx) + lane_off;
    } else {
      const int rowc = min(max(fy + lrow, 0), H - 1), xsc = min(max(sx + lcol, 0), W - 16);
      off = (uint32_t)(rowc * W + xsc);
    }
#if defined(RMX_BB_NOLOAD)  /* ablation: everything but the loads */
    asm volatile("" ::"v"(off), "s"(fbase));
#else
    __builtin_amdgcn_global_load_lds((blit_glb_ptr)(fbase + off), (blit_lds_ptr)dst, 16, 0, 0);
#endif
  };
  auto mask_one = [&](uint32_t d1, uint32_t& m, uint32_t& l, uint32_t& r) __attribute__((always_inline)) {
    const int sx = (int)(short)(d1 & 0xffffu), fy = (int)d1 >> 16;
    const int fyl = fy + lrow, xs = sx + lcol;
    const int d = xs - min(max(xs, 0), W - 16);
    l = (uint32_t)max(-d, 0), r = (uint32_t)max(d, 0);
    const int wd = (uint32_t)fyl < (uint32_t)H ? max(16 - (int)l - (int)r, 0) : 0;
    const uint32_t m16 = ((1u << wd) - 1u) << (l & 15u);
    m = m16 | m16 << 16;
    l &= 15u, r &= 15u;
  };
  auto record = [&](uint32_t pi, uint32_t& d0, uint32_t& d1, uint32_t& e0, uint32_t& e1) __attribute__((always_inline)) {
    const uint4 d = *reinterpret_cast<const uint4*>(&q[2 * pi]);  // (the same address in every lane)
    d0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)d.x), d1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)d.y);
    e0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)d.z), e1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)d.w);
  };
  uint32_t win = 0;  // barriers of this chunk executed so far (uniform)
  // pair `pi` of the chunk's queue goes out into ring slot S; the workgroup's waves meet once per window of 16 list
  // positions, each where its own queue crosses into the window (a wave without a frame there catches up later)
  auto issue = [&](auto stag, uint32_t pi) __attribute__((always_inline)) {
    constexpr int S = decltype(stag)::value;
    uint32_t d0, d1, e0, e1;
    record(pi, d0, d1, e0, e1);
#if !defined(RMX_BB_NOSYNC)
    while (win < (d0 >> 28)) {
      __builtin_amdgcn_s_barrier();
      ++win;
    }
#endif
    load_one(d0, d1, ring + (S * 2 + 0) * 1024);
    load_one(e0, e1, ring + (S * 2 + 1) * 1024);
  };
  auto consume = [&](auto ktag, uint32_t pi) __attribute__((always_inline)) {
    constexpr int K = decltype(ktag)::value, s = K % D;
    const bb_u32x4 ra = *reinterpret_cast<const bb_u32x4*>(ring + (s * 2 + 0) * 1024 + lane * 16);
    const bb_u32x4 rb = *reinterpret_cast<const bb_u32x4*>(ring + (s * 2 + 1) * 1024 + lane * 16);
    uint32_t d0, d1, e0, e1;
    record(pi, d0, d1, e0, e1);
    const uint32_t a[4] = {ra[0], ra[1], ra[2], ra[3]}, bq[4] = {rb[0], rb[1], rb[2], rb[3]};
    uint32_t n[4];
    bs::planes_from_bytes(a, bq, n);
    uint32_t m1 = 0xffffffffu, m2 = 0xffffffffu;
    if (!covers(d1) || !covers(e1)) {  // (uniform) a frame's edge crosses the tile
      uint32_t l1, r1, l2, r2;
      mask_one(d1, m1, l1, r1);
      mask_one(e1, m2, l2, r2);
      const uint32_t shl = l1 | l2 << 16, shr = r1 | r2 << 16;
#pragma unroll
      for (int k = 0; k < 4; ++k) n[k] = bb_pk_shr(bb_pk_shl(n[k], shl), shr);
    }
    uint32_t qq[4], h1[8], h2[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) qq[k] = __builtin_amdgcn_perm(n[k], n[k], 0x01000100u);
    bb_onehot(qq, m1, h1);
#pragma unroll
    for (int k = 0; k < 4; ++k) qq[k] = __builtin_amdgcn_perm(n[k], n[k], 0x03020302u);
    bb_onehot(qq, m2, h2);
    C.template add_pair<K>(h1, h2);
  };
  static_assert(8 % D == 0, "a group of eight pairs walks the ring a whole number of times");
  auto wsync = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };

  const uint32_t lend = p.col_list ? p.col_cnt[group + 1] : p.n;
  for (uint32_t cbase = p.col_list ? p.col_cnt[group] : 0u; cbase < lend; cbase += BB_CHUNK) {
    // ---- pass 1: the chunk's records against the tile (lane i tests record i of a batch of 64), the matching ones
    // in list order into the wave's queue, each with the window (16 list positions) it came from
    uint32_t nq = 0;
    uint4 rec = blit_record(p, cbase + lane, lend);
#pragma unroll 1
    for (uint32_t bb = 0; bb < BB_CHUNK / 64; ++bb) {
      const uint4 rn = blit_record(p, cbase + 64 * (bb + 1) + lane, bb + 1 < BB_CHUNK / 64 ? lend : 0u);  // (one batch ahead)
      const int ax = (int)rec.y, ay = (int)rec.z;
      const bool hit = rec.w != 0 && ay < Y0 + BB_ROWS && ay + H > Y0 && ax < X0 + BL_TW && ax + W > X0;
      const uint64_t m = __builtin_amdgcn_ballot_w64(hit);
      if (hit) q[nq + mbcnt(m)] = make_uint2(rec.x | (4u * bb + (lane >> 4)) << 28, (uint32_t)((X0 - ax) & 0xffff) | (uint32_t)(Y0 - ay) << 16);
      nq += (uint32_t)__builtin_popcountll(m);
      rec = rn;
    }
    if (lane < 16) q[nq + lane] = make_uint2(BB_NULL_LO | 15u << 28, BB_NULL_HI);  // complete the last group of eight pairs
    wsync();
    // ---- pass 2: groups of eight pairs in straight-line code (pair k of a group: ring slot k % D, tree position k);
    // D pairs in the pipe.  LDS-DMA loads land in order, two per pair: all but the 2 (D - 1) youngest have landed
    // after vmcnt(that many); where fewer pairs follow (the chunk's end, the group before a flush) wait for everything.
    const uint32_t npairs = (nq + 15u) / 16u * 8u;
    win = 0;
    auto prime = [&](uint32_t base) __attribute__((always_inline)) {
      if (base + 0 < npairs) issue(std::integral_constant<int, 0>{}, base + 0);
      if (D > 1 && base + 1 < npairs) issue(std::integral_constant<int, 1 % D>{}, base + 1);
      if (D > 2 && base + 2 < npairs) issue(std::integral_constant<int, 2 % D>{}, base + 2);
      if (D > 2 && base + 3 < npairs) issue(std::integral_constant<int, 3 % D>{}, base + 3);
      if (D > 4 && base + 4 < npairs) issue(std::integral_constant<int, 4 % D>{}, base + 4);
      if (D > 4 && base + 5 < npairs) issue(std::integral_constant<int, 5 % D>{}, base + 5);
      if (D > 4 && base + 6 < npairs) issue(std::integral_constant<int, 6 % D>{}, base + 6);
      if (D > 4 && base + 7 < npairs) issue(std::integral_constant<int, 7 % D>{}, base + 7);
    };
    prime(0);
    for (uint32_t g = 0; g < npairs;) {
      // the groups up to the next flush (or the chunk's end): no pair beyond `stop` goes out before it.  (The flush is
      // a call: kept out of the innermost loop, so is what the compiler saves around it.)
      const uint32_t stop = min(npairs, g + (BB_FLUSH_PAIRS - C.np));
#pragma unroll 1
      for (; g < stop; g += 8) {
        auto step = [&](auto ktag, uint32_t pi) __attribute__((always_inline)) {
          constexpr int K = decltype(ktag)::value;
          if (pi + D <= stop) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (D - 1)) : "memory");
          else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#if !defined(RMX_BB_NOCOMPUTE)  /* ablation: the loads and the walk only */
          consume(ktag, pi);
#endif
          if (pi + D < stop) issue(std::integral_constant<int, K % D>{}, pi + D);
        };
        step(std::integral_constant<int, 0>{}, g);
        step(std::integral_constant<int, 1>{}, g + 1);
        step(std::integral_constant<int, 2>{}, g + 2);
        step(std::integral_constant<int, 3>{}, g + 3);
        step(std::integral_constant<int, 4>{}, g + 4);
        step(std::integral_constant<int, 5>{}, g + 5);
        step(std::integral_constant<int, 6>{}, g + 6);
        step(std::integral_constant<int, 7>{}, g + 7);
        C.np += 8;
      }
      if (C.np == BB_FLUSH_PAIRS) {
        bb_flush(p, C, X0, Y0, lane, ring);
        prime(g);
      }
    }
#if !defined(RMX_BB_NOSYNC)
    while (win < BB_CHUNK / 16) {
      __builtin_amdgcn_s_barrier();
      ++win;
    }
#endif
    wsync();  // (the queue is rewritten by the next chunk)
  }
  if (C.np != 0 || (p.fresh && !C.wrote)) bb_flush(p, C, X0, Y0, lane, ring);
}

}  // namespace rmx
