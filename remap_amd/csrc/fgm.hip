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
  int32_t* carry;  // device: [0..1] position, [2] fragment id (out); with a summary also [3] fragments, [4] pending (= frc_apply's output)
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

// One workgroup; the stream is taken in spans of SCAN_NT x 16 frames (16 per thread, held in registers: every
// frame is loaded once, all loads of a thread in flight together) with the carry handed from span to span.
// 256 threads, not 1 024 (round 3): the kernel runs beside the other lane's extraction kernel, which leaves a CU a
// wave slot and a few dozen registers per SIMD — a 16-wave workgroup waited ~250 us for a CU to drain, and the
// whole chain of its lane with it (the kernel trace of the two-lane bench); alone it takes 16 us either way.
#ifndef RMX_SCAN_NT
#define RMX_SCAN_NT 256
#endif
constexpr uint32_t SCAN_PER = 16, SCAN_NT = RMX_SCAN_NT;

__global__ __launch_bounds__(SCAN_NT) void frc_scan(ScanParams p) {
  __shared__ ScanSeg s_wave[SCAN_NT / 64];
  __shared__ ScanSeg s_carry;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  if (tid == 0) s_carry = ScanSeg{0, p.carry_in[0], p.carry_in[1], p.carry_in[2]};
  for (uint32_t base = 0; base < p.n; base += SCAN_NT * SCAN_PER) {
    const uint32_t span = min(p.n - base, SCAN_NT * SCAN_PER);
    const uint32_t chunk = (span + SCAN_NT - 1) / SCAN_NT;  // <= SCAN_PER
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
    // segmented scan of the chunk summaries: wave-inclusive by shuffles, the wave
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
      for (uint32_t w = 0; w < SCAN_NT / 64; ++w) {
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
      // bit 1: tickets left to rmx_kpm_resolve; bit 2: this block failed for good (a table overflowed in
      // rmx_kpe_batch, status word 0, or the matcher hit a limit other than "tickets left") — so that every
      // rank of a sharded stream takes the same branch after the exchange
      int32_t pending = 0;
      if (p.kpm_status) {
        const uint32_t km = *p.kpm_status, ke = *(p.kpm_status - 1);
        pending = ((km & 128u) ? 2 : 0) | ((ke || (km & ~128u)) ? 4 : 0);
      }
      p.summary[0] = ((p.first_is_start || s_carry.c > 0) ? 1 : 0) | pending;
      p.summary[1] = s_carry.x;
      p.summary[2] = s_carry.y;
      p.summary[3] = s_carry.c;
      // what frc_apply leaves for rmx_frc_finish when this block is the whole stream (one rank:
      // no apply launch needed; with several ranks frc_apply overwrites both words)
      p.carry[3] = s_carry.c + 1;
      p.carry[4] = pending;
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
  uint32_t* nfrag_out;  // fragments of the whole stream; [1]: bit 1 some block has tickets left to rmx_kpm_resolve, bit 2 some block failed
  uint32_t n, rank, world;
};
__global__ __launch_bounds__(256) void frc_apply(ApplyParams p) {
  int32_t ax = 0, ay = 0;
  uint32_t ac = 0, all = 0, pending = 0;
  for (uint32_t q = 0; q < p.world; ++q) {  // (world is small; every thread folds for itself)
    const int32_t r = p.summaries[4 * q] & 1, x = p.summaries[4 * q + 1], y = p.summaries[4 * q + 2];
    const uint32_t c = (uint32_t)p.summaries[4 * q + 3];
    pending |= (uint32_t)p.summaries[4 * q] & 6u;
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
// Output-centric, counters in REGISTERS: a wave owns 64 canvas columns x R rows, a lane one
// column of them.  It walks the frames binned to its group of tile columns; of every frame that
// covers the tile a lane takes its R pixels (one byte each; how they get there: at the kernels)
// and counts them in sixteen 4-bit counters per pixel (two dwords: `v_lshlrev_b64` of a one by
// 4 x colour, two adds).  Every 15 frames the nibbles are folded into 8-bit counters (four dwords
// per pixel), every 255 frames — and at the end — those into the canvas with its uint16
// wraparound (fgm.hpp:14).  No atomics, no barriers; every canvas dot is read and written once
// per 255 covering frames.  Lanes left or right of a frame shift a zero instead of a one: the
// frame's edges cost nothing per pixel.
struct BlitParams {
  const uint8_t* frames;
  const uint8_t* mask;   // optional, same layout as frames: pixels with a non-zero byte are not counted (fgm.hpp:71-86)
  const int32_t* pos;
  const uint32_t* frag_id;
  uint32_t* canvas;  // canvas_h x canvas_w x 8 dwords
  uint32_t n, W, H;
  uint32_t fragment, cw, ch;
  int32_t zx, zy;
  uint32_t tiles_x;   // 64-pixel tile columns
  uint32_t groups_x;  // 256-pixel groups of four tile columns (one workgroup's waves)
  // frames binned by the GROUPS they overlap (built by blit_bin_* below): the four waves of a
  // workgroup walk the same list in the same order, each skipping what misses its own tile
  uint32_t* col_cnt;  // groups_x + 1: counts, then exclusive offsets (in place)
  uint32_t* col_cur;  // groups_x fill cursors
  uint4* col_list;    // (frame, canvas x, canvas y, -) records, <= n * (W / 256 + 2); nullptr: a short batch, not binned
  // Order of a group's list (blit_bin_sort): by frame, or class by class of (frame mod `stride`) — then EVERY
  // group's list runs through the batch in the same order of residues, and two workgroups that need the same frame
  // (x-neighbouring groups share the lines at their common border) reach it at about the same list position.  The
  // order of a list has no influence on the result.
  uint32_t stride;    // 0: the lists stay as the atomics of blit_bin_fill left them (chunks of 64 frames in arbitrary order); 1: frame order
  uint32_t fresh;     // the canvas holds no counts yet and need not be initialised: every dot is WRITTEN (first fold of a
                      // tile without reading, tiles no frame touches write zeros) — no memset, no read of the canvas
};

constexpr int BL_TW = 64, BL_GW = 4 * BL_TW;
// rows of a wave's tile and frames in its pipe (measured: 8 x 2..8 and 4 x 4..8 within 3 %, 16 rows slower);
// the byte-per-lane kernel for narrow frames holds two frames of 4 rows in registers
#ifndef RMX_BL_ROWS
#define RMX_BL_ROWS 8
#endif
#ifndef RMX_BL_DEPTH
#define RMX_BL_DEPTH 6  // frames in a wave's pipe (4: 0.375 / 4.45 ms per call on the scroller / rooms stream, 6: 0.357 / 4.05, 8: the same, 12: twice as slow)
#endif
constexpr int BL_ROWS = RMX_BL_ROWS, BL_DEPTH = RMX_BL_DEPTH, BL_ROWS_NARROW = 4;
#ifndef RMX_BLIT_SYNC
#define RMX_BLIT_SYNC 16  // list positions between two barriers of a workgroup's waves (a divisor of 64; 0: none): 2.17 -> 1.43 GB fetched per 10 000 frames, same time (4: also 1.43 GB, +10 % time)
#endif
#ifndef RMX_BLIT_ST_AUX
#define RMX_BLIT_ST_AUX 0  // cache policy of the canvas stores (2: non-temporal)
#endif
#ifndef RMX_BLIT_XFOLD
#define RMX_BLIT_XFOLD 1  // canvas rows cross the memory interface in memory order (BlitCounters::fold_canvas_x)
#endif
constexpr uint32_t BL_DIRECT_MAX = 1024;  // batches up to this many frames skip the binning passes

// A frame overlaps at most W/256 + 2 groups; three tiny passes bin the batch's frames by
// group so that a tile walks ~(W + 256) / (mean step) frames instead of all n.
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
    t0 = max(ax, 0) / BL_GW;
    t1 = min((ax + (int)p.W - 1) / BL_GW, (int)p.groups_x - 1);
  }
  for (int k = 0; __builtin_amdgcn_ballot_w64(t0 + k <= t1) != 0; ++k) wave_claim(p.col_cnt, t0 + k, t0 + k <= t1);
}
__global__ __launch_bounds__(1024) void blit_bin_scan(BlitParams p) {
  __shared__ uint32_t part[1024];
  const uint32_t tid = threadIdx.x, n = p.groups_x;
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
  int t0 = 0, t1 = -1, ax = 0, ay = 0;
  if (live) {
    ax = p.pos[2 * f] - p.zx;
    ay = p.pos[2 * f + 1] - p.zy;
    t0 = max(ax, 0) / BL_GW;
    t1 = min((ax + (int)p.W - 1) / BL_GW, (int)p.groups_x - 1);
  }
  for (int k = 0; __builtin_amdgcn_ballot_w64(t0 + k <= t1) != 0; ++k) {
    const bool act = t0 + k <= t1;
    const uint32_t at = wave_claim(p.col_cur, t0 + k, act);
    if (act) p.col_list[at] = make_uint4(f, (uint32_t)ax, (uint32_t)ay, 0u);
  }
}

// One workgroup per group: its list in the order of (frame mod stride, frame).  Lists longer than BIN_SORT_MAX stay
// as they are.
constexpr uint32_t BIN_SORT_MAX = 4096;
__global__ __launch_bounds__(256) void blit_bin_sort(BlitParams p) {
  __shared__ uint64_t key[BIN_SORT_MAX];
  __shared__ uint4 rec[BIN_SORT_MAX];
  const uint32_t g = blockIdx.x, beg = p.col_cnt[g], L = p.col_cnt[g + 1] - beg;
  if (L < 2 || L > BIN_SORT_MAX) return;  // (uniform)
  uint32_t n2 = 2;
  while (n2 < L) n2 <<= 1;
  const uint32_t per = (p.n + p.stride - 1) / p.stride;
  for (uint32_t i = threadIdx.x; i < n2; i += 256) {
    uint64_t k = ~0ull;
    if (i < L) {
      const uint4 r = p.col_list[beg + i];
      rec[i] = r;
      k = ((uint64_t)(r.x % p.stride) * per + r.x / p.stride) << 16 | i;
    }
    key[i] = k;
  }
  for (uint32_t k = 2; k <= n2; k <<= 1)
    for (uint32_t j = k >> 1; j > 0; j >>= 1) {
      __syncthreads();
      for (uint32_t i = threadIdx.x; i < n2; i += 256) {
        const uint32_t ixj = i ^ j;
        if (ixj > i) {
          const uint64_t x = key[i], y = key[ixj];
          if ((x > y) == ((i & k) == 0)) key[i] = y, key[ixj] = x;
        }
      }
    }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < L; i += 256) p.col_list[beg + i] = rec[(uint32_t)key[i] & 0xffffu];
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t blit_rsrc(const uint8_t* base, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(base), (short)0, (int)bytes, 0x00020000);
}

// nibble (c & 15) of the 64-bit counter word (lo, hi) += one.  v_lshlrev_b64 takes the low six
// bits of its shift operand, so c << 2 needs no mask; no nibble overflows before the fold (15
// increments at most), so the two halves add independently.
__device__ __forceinline__ void blit_count(uint32_t& lo, uint32_t& hi, uint32_t c, uint32_t one) {
  uint64_t inc;
  const uint64_t o64 = one;
  asm("v_lshlrev_b64 %0, %1, %2" : "=v"(inc) : "v"(c << 2), "v"(o64));
  lo += (uint32_t)inc;
  hi += (uint32_t)(inc >> 32);
}

typedef unsigned short blit_us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t add_u16x2(uint32_t a, uint32_t b) {
  return __builtin_bit_cast(uint32_t, __builtin_bit_cast(blit_us2, a) + __builtin_bit_cast(blit_us2, b));
}

// R rows x (two nibble words + four byte words) of counters of one lane's canvas column
template <int R, bool XF = false>
struct BlitCounters {
  uint32_t nib[R][2], byt[R][4];
  uint32_t in_nib, in_byt;  // frames counted since the last fold (uniform)
  bool wrote;               // this tile's dots have been stored once (uniform)
  uint4* xb;                // 128 x 16 bytes of the wave's LDS: canvas rows in memory order (nullptr: every lane stores its own dot)

  __device__ __forceinline__ void clear(uint4* xb_ = nullptr) {
    xb = xb_;
    wrote = false;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      nib[r][0] = nib[r][1] = 0;
      byt[r][0] = byt[r][1] = byt[r][2] = byt[r][3] = 0;
    }
    in_nib = in_byt = 0;
  }
  __device__ __forceinline__ void add(int r, uint32_t c, uint32_t one) { blit_count(nib[r][0], nib[r][1], c, one); }
  // The canvas side in MEMORY ORDER (fgm_blit: xb = 2 KB of the wave's LDS).  A lane owns a column, i.e. one dot of
  // 32 bytes per row: stored by its owner, a row of the tile is two instructions of 16 bytes per lane at a stride
  // of 32 — half of every 32-byte sector per instruction.  Here the wave's LDS turns the row round: lane i moves
  // 16-byte piece i of the row's first 32 dots and piece i of the next 32 (two instructions of 1 KB, contiguous),
  // the owners read / write their dots in LDS.
  __device__ __forceinline__ void fold_canvas_x(const BlitParams& p, int X, int Y0, uint4* xb) {
    typedef unsigned int u32x4 __attribute__((__vector_size__(4 * sizeof(unsigned int))));
    const uint32_t lane = threadIdx.x & 63u;
    const int X0 = X - (int)lane;
    // the tile's R rows through a buffer descriptor at its first dot: row offsets are scalar, a lane's two
    // offsets are all the addressing that lives in vector registers; pieces right of the canvas get an offset the
    // range check drops
    const uint32_t row_bytes = p.cw * 32u;
    const __amdgpu_buffer_rsrc_t crs = __builtin_amdgcn_make_buffer_rsrc(
        p.canvas + ((size_t)Y0 * p.cw + (size_t)X0) * 8, (short)0, (int)min((uint64_t)R * row_bytes, (uint64_t)0x7fffffffu), 0x00020000);
    const uint32_t vo0 = X0 + (int)(lane >> 1) < (int)p.cw ? lane * 16u : 0x80000000u;
    const uint32_t vo1 = X0 + 32 + (int)(lane >> 1) < (int)p.cw ? 1024u + lane * 16u : 0x80000000u;
    const bool rd = !p.fresh || wrote;
    auto wsync = [&]() {  // (LDS operations of a wave execute in order; this keeps the compiler from moving them)
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
    };
    u32x4* xv = reinterpret_cast<u32x4*>(xb);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int Y = Y0 + r;
      if (Y < (int)p.ch) {  // (uniform)
        const uint32_t so = (uint32_t)r * row_bytes;
        uint4 a = make_uint4(0, 0, 0, 0), c = a;
        if (rd) {
          xv[lane] = __builtin_amdgcn_raw_buffer_load_b128(crs, vo0, so, 0);
          xv[64 + lane] = __builtin_amdgcn_raw_buffer_load_b128(crs, vo1, so, 0);
          wsync();
          a = xb[2 * lane], c = xb[2 * lane + 1];
          wsync();
        }
        const uint32_t e0 = byt[r][0], o0 = byt[r][1], e1 = byt[r][2], o1 = byt[r][3];
        a.x = add_u16x2(a.x, __builtin_amdgcn_perm(o0, e0, 0x0c040c00u));
        a.y = add_u16x2(a.y, __builtin_amdgcn_perm(o0, e0, 0x0c050c01u));
        a.z = add_u16x2(a.z, __builtin_amdgcn_perm(o0, e0, 0x0c060c02u));
        a.w = add_u16x2(a.w, __builtin_amdgcn_perm(o0, e0, 0x0c070c03u));
        c.x = add_u16x2(c.x, __builtin_amdgcn_perm(o1, e1, 0x0c040c00u));
        c.y = add_u16x2(c.y, __builtin_amdgcn_perm(o1, e1, 0x0c050c01u));
        c.z = add_u16x2(c.z, __builtin_amdgcn_perm(o1, e1, 0x0c060c02u));
        c.w = add_u16x2(c.w, __builtin_amdgcn_perm(o1, e1, 0x0c070c03u));
        xb[2 * lane] = a, xb[2 * lane + 1] = c;
        wsync();
        const u32x4 m0 = xv[lane], m1 = xv[64 + lane];
        wsync();
        // (as in kpe_bs / kpe_pc: nothing may be scheduled between a 16-byte store with a scalar offset and the
        // wait states behind it — tools/check_store_hazard.py)
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_raw_buffer_store_b128(m0, crs, vo0, so, RMX_BLIT_ST_AUX);
        __builtin_amdgcn_raw_buffer_store_b128(m1, crs, vo1, so, RMX_BLIT_ST_AUX);
        asm volatile("s_nop 1");
        __builtin_amdgcn_sched_barrier(0);
      }
      byt[r][0] = byt[r][1] = byt[r][2] = byt[r][3] = 0;
    }
    in_byt = 0;
    wrote = true;
  }
  __device__ __forceinline__ void fold_canvas(const BlitParams& p, int X, int Y0) {
    if constexpr (XF) return fold_canvas_x(p, X, Y0, xb);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int Y = Y0 + r;
      if (Y < (int)p.ch && X < (int)p.cw) {
        uint4* dst = reinterpret_cast<uint4*>(p.canvas + ((size_t)Y * p.cw + (size_t)X) * 8);
        uint4 a = make_uint4(0, 0, 0, 0), c = a;
        if (!p.fresh || wrote) a = dst[0], c = dst[1];
        // dword k of a dot = bins 2k (low half) and 2k+1 (high half); byt[r][0/1] hold the even/odd
        // colours 0..7, byt[r][2/3] those of 8..15, one byte per colour pair k
        const uint32_t e0 = byt[r][0], o0 = byt[r][1], e1 = byt[r][2], o1 = byt[r][3];
        a.x = add_u16x2(a.x, __builtin_amdgcn_perm(o0, e0, 0x0c040c00u));
        a.y = add_u16x2(a.y, __builtin_amdgcn_perm(o0, e0, 0x0c050c01u));
        a.z = add_u16x2(a.z, __builtin_amdgcn_perm(o0, e0, 0x0c060c02u));
        a.w = add_u16x2(a.w, __builtin_amdgcn_perm(o0, e0, 0x0c070c03u));
        c.x = add_u16x2(c.x, __builtin_amdgcn_perm(o1, e1, 0x0c040c00u));
        c.y = add_u16x2(c.y, __builtin_amdgcn_perm(o1, e1, 0x0c050c01u));
        c.z = add_u16x2(c.z, __builtin_amdgcn_perm(o1, e1, 0x0c060c02u));
        c.w = add_u16x2(c.w, __builtin_amdgcn_perm(o1, e1, 0x0c070c03u));
        dst[0] = a;
        dst[1] = c;
      }
      byt[r][0] = byt[r][1] = byt[r][2] = byt[r][3] = 0;
    }
    in_byt = 0;
    wrote = true;
  }
  __device__ __forceinline__ void fold_nibbles(const BlitParams& p, int X, int Y0) {
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const uint32_t a = nib[r][h], e = a & 0x0f0f0f0fu;
        byt[r][2 * h] += e;
        byt[r][2 * h + 1] += (a - e) >> 4;
        nib[r][h] = 0;
      }
    in_nib = 0;
    if (++in_byt == 17) fold_canvas(p, X, Y0);  // 17 x 15 = 255: a byte is full
  }
  __device__ __forceinline__ void frame_done(const BlitParams& p, int X, int Y0) {
    if (++in_nib == 15) fold_nibbles(p, X, Y0);
  }
  __device__ __forceinline__ void finish(const BlitParams& p, int X, int Y0) {
    if (in_nib) fold_nibbles(p, X, Y0);
    if (in_byt || (p.fresh && !wrote)) fold_canvas(p, X, Y0);
  }
};

// Record i of a tile's frame list: (frame, canvas x, canvas y, live).  With binned lists it is
// read from the group's list; a short batch (col_list == nullptr) is not binned at all — every
// tile tests every frame of the batch, the record comes straight from the position arrays.
__device__ __forceinline__ uint4 blit_record(const BlitParams& p, uint32_t i, uint32_t lend) {
  if (i >= lend) return make_uint4(0, 0, 0, 0);
  if (p.col_list) {
    uint4 r = p.col_list[i];
    r.w = 1u;
    return r;
  }
  const bool live = !(p.frag_id && p.frag_id[i] != p.fragment);
  return make_uint4(i, (uint32_t)(p.pos[2 * i] - p.zx), (uint32_t)(p.pos[2 * i + 1] - p.zy), live ? 1u : 0u);
}

// The frames of one tile, one at a time: a batch of 64 records is tested against the tile's
// rows by the wave's lanes, the survivors are handed out through readlane (all scalar).
struct BlitWalk {
  const BlitParams* pp;
  uint32_t base, lend, lane;
  int X0, Y0, R, W, H;
  uint64_t todo;
  uint4 rec, rec_next;
  __device__ __forceinline__ void start(const BlitParams& p, uint32_t group, uint32_t lane_, int X0_, int Y0_, int R_) {
    pp = &p;
    base = p.col_list ? p.col_cnt[group] : 0u;
    lend = p.col_list ? p.col_cnt[group + 1] : p.n;
    lane = lane_;
    X0 = X0_;
    Y0 = Y0_;
    R = R_;
    W = (int)p.W;
    H = (int)p.H;
    todo = 0;
    rec = make_uint4(0, 0, 0, 0);
    rec_next = blit_record(p, base + lane, lend);
  }
  __device__ __forceinline__ bool next(uint32_t& f, int& ax, int& ay) {
    while (todo == 0) {
      if (base >= lend) return false;
      rec = rec_next;
      const bool live = rec.w != 0;
      base += 64;
      rec_next = blit_record(*pp, base + lane, lend);
      const int rx = (int)rec.y, ry = (int)rec.z;
      todo = __builtin_amdgcn_ballot_w64(live && ry < Y0 + R && ry + H > Y0 && rx < X0 + BL_TW && rx + W > X0);
    }
    const int j = (int)__builtin_ctzll(todo);
    todo &= todo - 1;
    f = (uint32_t)__builtin_amdgcn_readlane((int)rec.x, j);
    ax = __builtin_amdgcn_readlane((int)rec.y, j);
    ay = __builtin_amdgcn_readlane((int)rec.z, j);
    return true;
  }
};

// tile index of this workgroup: workgroups of one XCD (blockIdx % 8) take a contiguous range
// of tiles, x fastest — neighbours in time share the frames' lines in that XCD's L2
__device__ __forceinline__ uint32_t blit_tile_of_block() {
  const uint32_t b = blockIdx.x, nb = gridDim.x, bq = nb >> 3, br = nb & 7u, x = b & 7u;
  return (x < br ? x * (bq + 1) : br * (bq + 1) + (x - br) * bq) + (b >> 3);
}

typedef __attribute__((address_space(3))) void* blit_lds_ptr;
typedef const __attribute__((address_space(1))) void* blit_glb_ptr;

// Frames at least 64 pixels wide.  The memory side is what bounds this kernel (the 128-byte lines
// its 64-byte windows touch), so the load path is kept as light as it gets: ONE
// 16-byte-per-lane load per frame brings the tile's R rows x 64 bytes (lane j: row j / 4,
// piece j % 4) straight into LDS (LDS-DMA: no VGPRs, D frames in flight per wave), and a lane
// picks its column's R bytes out of LDS.  The 64-byte window is clamped into the frame's row,
// the lanes read it at the matching shift; lanes beside the frame add zero.
template <int R, int D, bool MASK>
__global__ __launch_bounds__(256) void fgm_blit(BlitParams p) {
  __shared__ __attribute__((aligned(16))) uint8_t ring[4][D][MASK ? 2 : 1][R * 64];
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // (uniform: the frame walk runs on the scalar unit)
  const uint32_t b = blit_tile_of_block();
  const uint32_t group = b % p.groups_x, tcol = group * 4 + wv;
  // Waves without a tile leave here, BEFORE the first barrier: refill() below synchronises the workgroup
  // (RMX_BLIT_SYNC), and s_barrier counts only the waves that have not ended.  The invariant the barriers rest
  // on: every wave that stays walks the SAME list (its workgroup's column bin, both passes) and executes exactly
  // 64 / RMX_BLIT_SYNC barriers per batch of 64 records, whatever its tile matches — no per-wave exit or
  // per-wave list may be added below without taking the barriers out.
  if (tcol >= p.tiles_x) return;
  const int Y0 = (int)(b / p.groups_x) * R;
  const int X0 = (int)(tcol * BL_TW), X = X0 + (int)lane;
  const int W = (int)p.W, H = (int)p.H;
  const uint64_t fbytes = (uint64_t)H * (uint64_t)W;
  const int lrow = (int)(lane >> 2), lpiece = (int)(lane & 3u) * 16;
  const uint32_t lane_off = (uint32_t)lrow * (uint32_t)W + (uint32_t)lpiece;  // this lane's 16 bytes within the R x 64 window

#if RMX_BLIT_XFOLD
  __shared__ uint4 xfold[4][128];
  BlitCounters<R, true> cnt;
  cnt.clear(xfold[wv]);
#else
  BlitCounters<R> cnt;
  cnt.clear();
#endif

  // ---- the frame walk.  64 records at a time: lane i tests record i against the tile and works
  // out everything a frame needs as a vector operation (once per 64 frames, not per frame):
  //   w_base  byte offset of the window's first row in the frame plane (all R rows inside the
  //           frame: "full"), else of the window's column in row 0
  //   w_info  (shift of the lanes against the window + 64) | rows inside the frame << 16
  //   w_fy0   first tile row in frame coordinates (not-full frames only)
  // a frame is then popped with three readlanes.
  // The list is walked TWICE: first the frames that cover all R rows of the tile (nearly all),
  // counted by a straight sequence of 4 R instructions, then the others with a test per row.
  // (One loop with both sequences made the compiler copy the 16 counters back where they met —
  // a quarter of the kernel's VALU instructions; one loop with a test per row for every frame
  // spends as many scalar instructions and branches.  The kernel is bound by the instructions
  // it issues, of all kinds: DESIGN.md 4.4.)
  const uint32_t lds_lane = (uint32_t)(uintptr_t)&ring[wv][0][0][0] + lane;  // (LDS byte address: low 32 bits of the pointer)
  auto pass = [&](auto full_tag) {
    constexpr bool FULL = decltype(full_tag)::value;
    const uint32_t lend = p.col_list ? p.col_cnt[group + 1] : p.n;
    uint32_t base = p.col_list ? p.col_cnt[group] : 0u;
    uint64_t todo = 0;
    uint64_t w_base = 0;
    uint32_t w_info = 0;
    int w_fy0 = 0;
    uint4 rec_next = blit_record(p, base + lane, lend);
#if RMX_BLIT_SYNC
    // The four waves of a workgroup hold x-neighbouring tiles and walk the same list: a 128-byte line of a
    // frame's row is needed by two (three) of them.  Left to themselves they drift apart by tens of frames and
    // each fetches its own copy from beyond the L2 (2.9 fetches per line); here a batch of 64 records is
    // served in windows of RMX_BLIT_SYNC list positions with a workgroup barrier between the windows — the same
    // number of barriers for every wave, whatever its tile matches — so neighbours ask for a line within a few
    // frames of each other.  (Waves without a tile have left: the barrier counts the remaining ones.)
    uint64_t todo_all = 0;
    uint32_t sub = 64 / RMX_BLIT_SYNC;
#endif
    auto refill = [&]() -> bool {  // false: the list is exhausted
      while (todo == 0) {
#if RMX_BLIT_SYNC
        if (sub < 64 / RMX_BLIT_SYNC) {
          todo = todo_all & (((RMX_BLIT_SYNC == 64 ? 0ull : (1ull << (RMX_BLIT_SYNC & 63))) - 1ull) << (sub * RMX_BLIT_SYNC));
          ++sub;
          __builtin_amdgcn_s_barrier();
          continue;
        }
#endif
        if (base >= lend) return false;
        const uint4 rec = rec_next;
        const bool live = rec.w != 0;
        base += 64;
        rec_next = blit_record(p, base + lane, lend);
        const int ax = (int)rec.y, ay = (int)rec.z;
        const int fy0 = Y0 - ay, sx = X0 - ax;
        const int sxc = min(max(sx, 0), W - BL_TW);  // the 64-byte window, clamped into the row
        const bool full = fy0 >= 0 && fy0 + R <= H;
        uint32_t rows = 0;  // the tile's rows that lie inside the frame
        if (!FULL) {
#pragma unroll
          for (int r = 0; r < R; ++r)
            if ((uint32_t)(fy0 + r) < (uint32_t)H) rows |= 1u << r;
        }
        w_base = (uint64_t)rec.x * fbytes + (uint64_t)((FULL ? (uint32_t)fy0 * (uint32_t)W : 0u) + (uint32_t)sxc);  // (W H < 2^32)
        w_info = (uint32_t)(sx - sxc + 64) | rows << 16;
        w_fy0 = fy0;
#if RMX_BLIT_SYNC
        todo_all = __builtin_amdgcn_ballot_w64(live && full == FULL && ay < Y0 + R && ay + H > Y0 && ax < X0 + BL_TW && ax + W > X0);
        sub = 0;
#else
        todo = __builtin_amdgcn_ballot_w64(live && full == FULL && ay < Y0 + R && ay + H > Y0 && ax < X0 + BL_TW && ax + W > X0);
#endif
      }
      return true;
    };
    uint32_t q_info[D];
    auto issue = [&](int s, int j) {
      q_info[s] = (uint32_t)__builtin_amdgcn_readlane((int)w_info, j);
      const uint64_t wb = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)w_base, j) |
                          (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(w_base >> 32), j) << 32;
      uint64_t off = wb + lane_off;  // (full: a scalar base and the lane's constant offset)
      if (!FULL) {
        const int fy0 = __builtin_amdgcn_readlane(w_fy0, j);
        const int row = min(max(fy0 + lrow, 0), H - 1);  // rows outside the frame: a valid address, not counted
        off = wb + (uint64_t)((uint32_t)row * (uint32_t)W + (uint32_t)lpiece);
      }
      if (R == 16 || lane < 4u * R) {
        __builtin_amdgcn_global_load_lds((blit_glb_ptr)(p.frames + off), (blit_lds_ptr)&ring[wv][s][0][0], 16, 0, 0);
        if (MASK) __builtin_amdgcn_global_load_lds((blit_glb_ptr)(p.mask + off), (blit_lds_ptr)&ring[wv][s][MASK ? 1 : 0][0], 16, 0, 0);
      }
    };
    auto count = [&](int s) {
      const uint32_t info = q_info[s];
      const int d = (int)(info & 255u) - 64;
      // the window lies inside the frame's row, so a lane is inside the frame iff it is inside the
      // window; the others shift a zero (and read some other byte of LDS, or none)
      const uint32_t t = lane + (uint32_t)d;
      const uint32_t one = t < (uint32_t)BL_TW ? 1u : 0u;
      typedef __attribute__((address_space(3))) const uint8_t* lds_bytes;
      lds_bytes px = (lds_bytes)(uintptr_t)(lds_lane + (uint32_t)d + (uint32_t)(s * (MASK ? 2 : 1) * R * 64));
      uint32_t v[R], m[R];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        v[r] = px[r * 64];
        if (MASK) m[r] = px[R * 64 + r * 64];
      }
#pragma unroll
      for (int r = 0; r < R; ++r)
        if (FULL || (info >> (16 + r) & 1u)) cnt.add(r, v[r], MASK ? (m[r] == 0 ? one : 0u) : one);
      cnt.frame_done(p, X, Y0);
    };
    // D frames in the pipe: a slot's frame is counted D - 1 issues after its load went out.  The
    // compiler does not count LDS-DMA: loads land in order, so all but the D - 1 youngest (each
    // frame is one load, two with a mask) have landed after vmcnt(that many); draining: vmcnt(0)
    bool have[D];
#pragma unroll
    for (int s = 0; s < D; ++s) have[s] = false;
    bool more = true;
    while (more) {
#pragma unroll
      for (int s = 0; s < D; ++s) {
        if (!more) break;
        if (have[s]) {
          if (MASK) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (D - 1)) : "memory");
          else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D - 1) : "memory");
          count(s);
        }
        more = have[s] = refill();
        if (more) {
          const int j = (int)__builtin_ctzll(todo);
          todo &= todo - 1;
          issue(s, j);
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int s = 0; s < D; ++s)
      if (have[s]) count(s);
  };
  pass(std::true_type{});
  pass(std::false_type{});
  cnt.finish(p, X, Y0);
}

// Frames narrower than 64 pixels: a byte per lane and row, loaded straight into registers.
template <int R, bool MASK>
__global__ __launch_bounds__(256) void fgm_blit_narrow(BlitParams p) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint32_t b = blit_tile_of_block();
  const uint32_t group = b % p.groups_x, tcol = group * 4 + wv;
  if (tcol >= p.tiles_x) return;
  const int Y0 = (int)(b / p.groups_x) * R;
  const int X = (int)(tcol * BL_TW + lane);
  const int W = (int)p.W, H = (int)p.H;
  const uint32_t fbytes = (uint32_t)(H * W);
  BlitCounters<R> cnt;
  cnt.clear();
  BlitWalk walk;
  walk.start(p, group, lane, (int)(tcol * BL_TW), Y0, R);
  struct Pend {
    uint32_t v[R], m[R];
    uint32_t one;
    int fy0;
  };
  auto issue = [&](Pend& q, uint32_t f, int ax, int ay) {
    const int xoff = X - ax;
    const bool in = (uint32_t)xoff < (uint32_t)W;
    q.one = in ? 1u : 0u;
    q.fy0 = Y0 - ay;
    const uint32_t voff = in ? (uint32_t)xoff : 0u;  // lanes beside the frame read its column 0 and add zero
    const size_t fbase = (size_t)f * fbytes;
    const __amdgpu_buffer_rsrc_t rs = blit_rsrc(p.frames + fbase, fbytes);
    const __amdgpu_buffer_rsrc_t rm = blit_rsrc((MASK ? p.mask : p.frames) + fbase, fbytes);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int so = min(max(q.fy0 + r, 0), H - 1) * W;  // rows outside the frame: a valid address, not counted
      q.v[r] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b8(rs, (int)voff, so, 0);
      if (MASK) q.m[r] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b8(rm, (int)voff, so, 0);
    }
  };
  auto count = [&](const Pend& q) {
#pragma unroll
    for (int r = 0; r < R; ++r)
      if ((uint32_t)(q.fy0 + r) < (uint32_t)H) cnt.add(r, q.v[r], MASK ? (q.m[r] == 0 ? q.one : 0u) : q.one);
    cnt.frame_done(p, X, Y0);
  };
  Pend q[2];
  bool have[2] = {false, false};
  bool more = true;
  while (more) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      if (!more) break;
      if (have[s]) count(q[s]);
      uint32_t f;
      int ax, ay;
      more = have[s] = walk.next(f, ax, ay);
      if (more) issue(q[s], f, ax, ay);
    }
  }
#pragma unroll
  for (int s = 0; s < 2; ++s)
    if (have[s]) count(q[s]);
  cnt.finish(p, X, Y0);
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
