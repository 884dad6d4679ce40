// kpe_bs.hip — keypoint extraction with the 3x3 pass on bit planes (gfx950), frames whose width is a
// multiple of 16.
//
// Replaces the same reference code as kpe.hip (kpe::extractor::extract, kpe.hpp:92-108, with sum_row /
// col_* / compute_pixel / median_pixel / encode_keypoint, kpe.hpp:111-379) and produces the same
// bytes; what differs is the arithmetic of compute_pixel's first half (kpe.hpp:308-315).  kpe_mf keeps
// one pixel per byte and selects the 4th largest of nine with a min3/med3/max3 network: 12.5 VALU
// lane-operations per pixel, nearly all in opcodes that issue every 4.3 cycles (profiles/
// r02_valu_ops.txt) — the kernel is VALU-issue-bound at 0.38 of the HBM peak.  Here a lane turns 32
// pixels (16 of a row of strip A, 16 of a row of strip B) into four bit planes and runs the radix
// selection of bitslice.h on them: ~8.5 lane-operations per pixel, a third of them full-rate
// and/xor, the rest v_bitop3_b32 / v_perm_b32 / packed 16-bit shifts.
//
// Work decomposition: a lane owns one 16-byte chunk of a row (one dwordx4 load, one dwordx4 store per
// strip and row: fully coalesced 1-KB wave accesses) and walks down two strips of `ra` rows at once,
// keeping three rows of planes (own, left-neighbour and right-neighbour pixels) in registers;
// neighbouring chunks are neighbouring lanes (DPP wave shifts, lanes 0 and 63 are halo lanes).  As in
// kpe.hip the lanes are laid over the batch's frames back to back (widths whose chunks divide 64 with few
// lanes idle: whole image rows of consecutive frames per wave, no halo lanes).  The candidate flags (own
// rank != 4th largest) ARE a bit plane, one word per lane and row step.  Two kernels share this streaming
// pass (kpe_bs_body) and differ in how the candidates are finished (5x5 rank 12 of 25, weight, code bytes,
// append to the frame's raw table):
//   kpe_bs       the words go to LDS; after its walk the workgroup hands its candidates out densely and
//                gathers their 5x5 windows from global memory (dense streams).
//   kpe_bs_ring  every wave keeps the last seven rows it streamed in an LDS ring and finishes the candidates of
//                three rows every three row steps by itself, windows from the ring; no workgroup barrier
//                (sparse streams: 8 % faster, 0.66 x the HBM traffic).  api.hip picks by the density of the
//                batch before.
#include "bitslice.h"
#include "common.h"

namespace rmx {

constexpr uint32_t KPE_BS_RA_MAX = 30;  // rows per strip (a multiple of 6; row index: 5 bits of a list entry)
#ifndef RMX_BS_ST_AUX
#define RMX_BS_ST_AUX 2  // cache policy of the median stores (gfx950: 1 sc0, 2 nt, 16 sc1): nt, they are not read again and
                         // leave the L2 to the frames' lines the finish gathers from (0.405 -> 0.39 ms)
#endif
#ifndef RMX_BS_LD_AUX
#define RMX_BS_LD_AUX 0
#endif
#ifndef RMX_BS_S1
#define RMX_BS_S1 2  // candidates per thread and pass of the finish's first stage
#endif
#ifndef RMX_BS_LIST
#define RMX_BS_LIST 2048
#endif
constexpr uint32_t KPE_BS_LIST = RMX_BS_LIST;  // candidates handed out per round

static_assert(bs::pack_tbl(kN2O, false) == bs::pack_tbl(bs::kRankOfNative, false) && bs::pack_tbl(kO2N, false) == bs::pack_tbl(bs::kNativeOfRank, false),
              "bitslice.h carries the same tables as common.h");

__device__ __forceinline__ uint32_t from_prev_lane(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, true);  // wave_shr:1
}
__device__ __forceinline__ uint32_t from_next_lane(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, true);  // wave_shl:1
}

typedef unsigned int u32x4 __attribute__((__vector_size__(4 * sizeof(unsigned int))));

struct BsRaw {
  u32x4 a, b;
};

// 16 + 16 native pixels -> rank planes of the pixels and of their horizontal neighbours
__device__ __forceinline__ void bs_make_row(const BsRaw& w, bs::Row& R) {
#if defined(RMX_BS_COPY)
  for (int q = 0; q < 4; ++q) R.c[q] = w.a[q], R.l[q] = w.b[q];
  return;
#endif
  const uint32_t a[4] = {w.a[0], w.a[1], w.a[2], w.a[3]}, b[4] = {w.b[0], w.b[1], w.b[2], w.b[3]};
  uint32_t n[4];
  bs::planes_from_bytes(a, b, n);
  bs::rank_planes(n, R.c);
#if defined(RMX_BS_NODPPASM)  /* experiment: the compiler's v_mov_b32_dpp + v_or_b32 (it does not fold wave shifts) */
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    R.l[q] = bs::left_of(R.c[q], from_prev_lane(bs::send_right(R.c[q])));
    R.r[q] = bs::right_of(R.c[q], from_next_lane(bs::send_left(R.c[q])));
  }
#else
  // neighbour planes: what the neighbouring lane sends is OR-ed in by the DPP operand of the v_or itself
  // (eight v_mov_b32_dpp fewer per row).  A DPP source written by the instruction before needs two wait
  // states, and the compiler does not look into an asm block: all sent words are computed first, one
  // s_nop 1 in front covers whichever of them came last.
  uint32_t sr[4], sl[4], ul[4], ur[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    sr[q] = bs::send_right(R.c[q]);
    sl[q] = bs::send_left(R.c[q]);
    ul[q] = bs::half_shl(R.c[q], 1);
    ur[q] = bs::half_shr(R.c[q], 1);
  }
  asm volatile(
      "s_nop 1\n\t"
      "v_or_b32_dpp %0, %8, %16 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_or_b32_dpp %1, %9, %17 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_or_b32_dpp %2, %10, %18 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_or_b32_dpp %3, %11, %19 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_or_b32_dpp %4, %12, %20 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_or_b32_dpp %5, %13, %21 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_or_b32_dpp %6, %14, %22 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_or_b32_dpp %7, %15, %23 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
      : "=&v"(R.l[0]), "=&v"(R.l[1]), "=&v"(R.l[2]), "=&v"(R.l[3]), "=&v"(R.r[0]), "=&v"(R.r[1]), "=&v"(R.r[2]), "=&v"(R.r[3])
      : "v"(sr[0]), "v"(sr[1]), "v"(sr[2]), "v"(sr[3]), "v"(sl[0]), "v"(sl[1]), "v"(sl[2]), "v"(sl[3]), "v"(ul[0]), "v"(ul[1]),
        "v"(ul[2]), "v"(ul[3]), "v"(ur[0]), "v"(ur[1]), "v"(ur[2]), "v"(ur[3]));
#endif
  bs::row_sums(R);
}

// The streaming pass of one workgroup: medians to HBM, candidate words to s_c[row step][thread].
// s_f / s_col [thread]: the lane's frame and 16-pixel chunk (0xffffffff: halo / idle lane or past the batch).
//
// Lanes.  p.HB = rows per wave (whole-row layout) or 0.  Whole rows: a wave holds HB complete image rows
// (same y, consecutive frames: 3 x 20 chunks at 320 pixels, four lanes idle); what a row's first and last
// chunk receive from the neighbouring lanes belongs to pixels the reference never processes (x < 2,
// x > W - 3), so no halo lanes are needed and a wave access touches whole 320-byte pieces
// (tools/micro/walk16_bw.hip: 0.32 ms per 10 000 frames against 0.36 for the layout below).  Otherwise
// (rows of more than 64 chunks, or ones that would leave many lanes idle) the lanes are laid over the
// frames back to back, 62 chunks per wave plus two halo lanes, as in kpe.hip.
//
// Memory operations are unconditional: a row that is not wanted (past the strip, past the frame) is given
// an out-of-range lane offset and dropped by the buffer range check.  With conditional loads or stores
// the compiler's s_waitcnt pass must assume the fewest younger operations at every join and ends up
// waiting for the stores of the previous step (vmcnt(0)) before every row; for the same reason the
// prologue issues its loads and (dropped) stores in the order of a steady-state step.
#ifndef RMX_BS_DEPTH
#define RMX_BS_DEPTH 2  // row pairs in flight
#endif
// XCD-aware block remap (bijective): blocks that share an XCD get neighbouring groups.
__device__ __forceinline__ uint32_t bs_logical_block() {
  const uint32_t nb = gridDim.x, bid = blockIdx.x;
  const uint32_t bq = nb >> 3, br = nb & 7u, xcd = bid & 7u;
  return (xcd < br ? xcd * (bq + 1) : br * (bq + 1) + (xcd - br) * bq) + (bid >> 3);
}
// first frame of wave group g (the frames of a workgroup's four waves follow each other)
__device__ __forceinline__ uint32_t bs_first_frame(const KpeParams& p, uint32_t g) {
  if (p.HB) return g * p.HB;
  const int64_t L0 = (int64_t)g * 62 - 1;
  return (uint32_t)((L0 < 0 ? 0 : (uint64_t)L0) / p.Wd);
}
// ---- the rolling finish (kpe_bs_ring) -------------------------------------------------------------------------
// The workgroup-wide finish gathers every candidate's 5x5 window from global memory 12 row steps after the
// stream read those rows: by then they have left the L2 (0.04 ms of the kernel) — and gathering them INSIDE the
// row loop is no better, the oldest row of a window is seven steps old.  Here a wave keeps the
// last SEVEN rows it streamed in LDS, two pixels per byte exactly as planes_from_bytes packs them first (16 B
// per lane and row for both strips: 7 KB per wave), and every three row steps it finishes the candidates of
// three rows by itself: hand-out among its own lanes through a list in LDS, windows read from the ring (only the
// two rows just outside a strip, which the stream never loads, come from global memory: one window row in
// thirty), stage-1 test, and for the keypoints weight and code bytes at once; finished records wait in a small
// per-wave list and are appended to their frames' tables in batches (table rows reserved with one global atomic
// per frame and batch, all frames in one instruction).  No workgroup barrier is left in the kernel; 40 960 B of
// LDS per workgroup and 128 VGPRs = four workgroups per CU (three cost 5 %: tools/ab_ring.sh, round 4).
// (kpe_pc.hip is this kernel's successor for widths whose rows fill a wave: same ring and roll, the candidate
// test by pairwise comparison, medians patched in the ring.)
#ifndef RMX_BSR_ROWS
#define RMX_BSR_ROWS 7
#endif
constexpr uint32_t BSR_ROWS = RMX_BSR_ROWS;   // ring rows (>= 7: a roll reads rows k0 - 2 .. k0 + 4)
constexpr uint32_t BSR_WL = 128;   // candidates a wave hands out per round
constexpr uint32_t BSR_KP = 64;    // finished records a wave holds before it appends them
__device__ __forceinline__ uint32_t bsr_slot(int q) {  // ring slot of strip row q (-1 ..)
  return (uint32_t)(q + 1) % BSR_ROWS;
}
struct alignas(16) BsWaveLds {
  uint32_t ring[BSR_ROWS][64][4];  // [slot][lane]: c0, c1 (strip A), c2, c3 (strip B) of planes_from_bytes
  uint32_t wl[BSR_WL];             // lane | row (0..2) << 6 | bit << 8
  uint32_t kp[BSR_KP][6];          // code dwords 0..3, x | y << 16, frame - first frame of the wave
  uint32_t fcnt[64], fbase[64];
};
struct BsWin {
  uint32_t w[5], e[5];  // natives: pixels x-2 .. x+1 of each row, and pixel x+2
};
__device__ __forceinline__ void bs_window(const CandData& d, uint32_t x, BsWin& W);
__device__ __forceinline__ void bs_ranks(const BsWin& W, uint32_t (&v)[7]);
__device__ __forceinline__ uint32_t bs_p3(const uint32_t (&v)[7]);
__device__ __forceinline__ bool bs_rank12_is(const uint32_t (&v)[7], uint32_t t);
__device__ __forceinline__ void bs_record(const BsWin& W, const uint32_t (&v)[7], uint4& c4);

struct BsRoll {
  BsWaveLds* L;
  const uint32_t* s_f;   // the wave's 64 entries
  const uint32_t* s_col;
  uint32_t lane, ya, RA, f0;  // f0: first frame of the wave
  uint32_t nkp;               // records waiting in L->kp (uniform)
};
// append the waiting records to their frames' raw tables
__device__ __forceinline__ void bs_flush(const KpeParams& p, BsRoll& R) {
  BsWaveLds& L = *R.L;
  const uint32_t lane = R.lane, n = R.nkp;
  if (n == 0) return;  // uniform
  L.fcnt[lane] = 0;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  uint32_t rank[BSR_KP / 64];
#pragma unroll
  for (uint32_t q = 0; q < BSR_KP / 64; ++q) {
    const uint32_t i = q * 64 + lane;
    rank[q] = i < n ? atomicAdd(&L.fcnt[L.kp[i][5]], 1u) : 0u;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  {
    const uint32_t c = L.fcnt[lane];
    L.fbase[lane] = c ? atomicAdd(&p.kcount[R.f0 + lane], c) : 0u;  // (lanes past the batch's last frame have no records)
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (uint32_t q = 0; q < BSR_KP / 64; ++q) {
    const uint32_t i = q * 64 + lane;
    if (i < n) {
      const uint32_t fr = L.kp[i][5], slot = L.fbase[fr] + rank[q];
      if (slot < p.kmax) {
        RawDesc* dd = p.raw + (size_t)(R.f0 + fr) * p.kmax + slot;
        reinterpret_cast<uint4*>(dd)[0] = make_uint4(L.kp[i][0], L.kp[i][1], L.kp[i][2], L.kp[i][3]);
        reinterpret_cast<uint4*>(dd)[1] = make_uint4(L.kp[i][4], 0u, 0u, 0u);
      } else {
        atomicOr(&p.status[0], 1u);
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  R.nkp = 0;
}
// the candidates of strip rows k0, k0 + 1, k0 + 2 (words w0..w2 of every lane); ring rows k0 - 1 .. k0 + 3 are in LDS
__device__ __forceinline__ void bs_roll(const KpeParams& p, BsRoll& R, uint32_t w0, uint32_t w1, uint32_t w2, int k0) {
  BsWaveLds& L = *R.L;
  const uint32_t lane = R.lane;
  const uint32_t c = (uint32_t)(__builtin_popcount(w0) + __builtin_popcount(w1) + __builtin_popcount(w2));
  uint32_t incl = c;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    uint32_t t = (uint32_t)__shfl_up((int)incl, o);
    if (lane >= (uint32_t)o) incl += t;
  }
  const uint32_t T = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
  for (uint32_t r0 = 0; r0 < T; r0 += BSR_WL) {  // uniform
    uint32_t idx = incl - c;
    const uint32_t ws[3] = {w0, w1, w2};
#pragma unroll
    for (uint32_t j = 0; j < 3; ++j) {
      uint32_t w = ws[j];
      while (w) {
        const uint32_t b = (uint32_t)__builtin_ctz(w);
        w &= w - 1;
        if (idx >= r0 && idx < r0 + BSR_WL) L.wl[idx - r0] = lane | j << 6 | b << 8;
        ++idx;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const uint32_t m = min(BSR_WL, T - r0);
    for (uint32_t q0 = 0; q0 < m; q0 += 64) {  // uniform
      const bool act = q0 + lane < m;
      const uint32_t e = act ? L.wl[q0 + lane] : (2u << 8);
      const uint32_t owner = e & 63u, j = (e >> 6) & 3u, b = e >> 8;
      const uint32_t s = b >> 4, xin = b & 15u;
      const int rr = k0 + (int)j;  // strip row of the candidate
      const uint32_t col = act ? R.s_col[owner] : 0u, f = act ? R.s_f[owner] : R.f0;
      const uint32_t x = col * 16 + xin, y = R.ya + s * R.RA + (uint32_t)rr;
      // window dwords (four pixels as bytes each): lo holds pixel (x - 2) & ~3, hi the next four
      const int A = (int)(owner * 4) + (((int)xin - 2) >> 2);  // byte-dword index along the wave's row (>= 0 for a processed pixel)
      const uint32_t Alo = (uint32_t)max(A, 0), Ahi = min(Alo + 1u, 255u);
      const uint32_t off_lo = (Alo >> 2) * 4 + 2 * s + ((Alo >> 1) & 1u), sh_lo = 4 * (Alo & 1u);
      const uint32_t off_hi = (Ahi >> 2) * 4 + 2 * s + ((Ahi >> 1) & 1u), sh_hi = 4 * (Ahi & 1u);
      const uint32_t* ring = &L.ring[0][0][0];
      CandData d;
      bool outside = false;
      const uint32_t slot0 = (uint32_t)(rr + (int)BSR_ROWS - 1) % BSR_ROWS;  // slot of window row 0 (strip row rr - 2)
#pragma unroll
      for (int r = 0; r < 5; ++r) {
        const int wr = rr - 2 + r;  // strip row of this window row: -2 .. RA + 1
        const uint32_t slot = slot0 + (uint32_t)r - (slot0 + (uint32_t)r >= BSR_ROWS ? BSR_ROWS : 0u);
        d.lo[r] = ring[slot * 256 + off_lo] >> sh_lo;
        d.hi[r] = ring[slot * 256 + off_hi] >> sh_hi;
        outside = outside || (act && (wr < -1 || wr > (int)R.RA));
      }
      if (__builtin_amdgcn_ballot_w64(outside)) {  // the rows just outside the strip: not streamed, from global memory
        const uint8_t* base = p.frames + (size_t)f * p.W * p.H + ((x - 2) & ~3u);
        const int r_out = rr < 2 ? 0 : 4;  // (row -2 is window row 0 of strip row 0, row RA + 1 window row 4 of strip row RA - 1)
        if (outside) {
          const uint2 v = *reinterpret_cast<const uint2*>(base + (size_t)(y - 2 + r_out) * p.W);
          if (r_out == 0) d.lo[0] = v.x, d.hi[0] = v.y;
          else d.lo[4] = v.x, d.hi[4] = v.y;
        }
      }
      // (window and ranks are made once, for the test and for the record)
      BsWin Wn;
      uint32_t vr[7];
      bs_window(d, x, Wn);
      bs_ranks(Wn, vr);
      const bool kp = act && !bs_rank12_is(vr, bs_p3(vr));
      const uint64_t km = __builtin_amdgcn_ballot_w64(kp);
      if (km) {
        uint4 c4;
        bs_record(Wn, vr, c4);
        if (R.nkp + (uint32_t)__builtin_popcountll(km) > BSR_KP) bs_flush(p, R);  // (uniform; the new records are still in registers)
        if (kp) {
          uint32_t* rec = L.kp[R.nkp + mbcnt(km)];
          rec[0] = c4.x, rec[1] = c4.y, rec[2] = c4.z, rec[3] = c4.w;
          rec[4] = x | y << 16;
          rec[5] = f - R.f0;
        }
        R.nkp += (uint32_t)__builtin_popcountll(km);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

// ROLL: the rolling finish (candidate words stay in registers, s_wave holds the waves' rings and lists); else the
// candidate words go to s_c for the workgroup's finish
template <bool ROLL>
__device__ __forceinline__ void kpe_bs_body(const KpeParams& p, uint32_t (*s_c)[256], uint32_t* s_f, uint32_t* s_col, BsWaveLds* s_wave) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  s_f[threadIdx.x] = 0;  // (written before the walk: nothing but the thread index lives across it)
  s_col[threadIdx.x] = 0xffffffffu;
  const uint32_t g = bs_logical_block() * 4 + wv;
  if (g >= p.groups) return;  // whole wave

  const uint32_t Wc = p.Wd;  // 16-byte chunks per row
  const size_t frame_bytes = (size_t)p.W * p.H;
  bool lane_ok;
  uint32_t f, c, f0;
  if (p.HB) {  // whole rows: lane = (row of frame f0 + lane / Wc, chunk lane % Wc)
    f0 = g * p.HB;
    const uint32_t r = lane / Wc;
    c = lane - r * Wc;
    f = f0 + r;
    lane_ok = r < p.HB && f < p.n;
    if (!lane_ok) f = f0, c = 0;
  } else {
    const int64_t L0 = (int64_t)g * 62 - 1;  // wave-uniform: virtual chunk column of lane 0
    const int64_t L = L0 + (int64_t)lane;
    lane_ok = lane >= 1 && lane <= 62 && L >= 0 && L < (int64_t)p.V;
    const uint64_t Lc = L < 0 ? 0 : (L >= (int64_t)p.V ? p.V - 1 : (uint64_t)L);
    f = (uint32_t)(Lc / Wc), c = (uint32_t)(Lc % Wc);
    const uint64_t Lb = L0 < 0 ? 0 : (uint64_t)L0;
    f0 = (uint32_t)(Lb / Wc);
  }
  const uint32_t voff = (uint32_t)((size_t)(f - f0) * frame_bytes) + c * 16;  // < 5 frames
  const size_t left = (size_t)(p.n - f0) * frame_bytes;
  const uint32_t recs = (uint32_t)(left < 0x7fffffffull ? left : 0x7fffffffull);
  const rsrc_t rin = make_rsrc(p.frames + (size_t)f0 * frame_bytes, recs);
  const rsrc_t rout = make_rsrc(p.median + (size_t)f0 * frame_bytes, recs);
  const uint32_t voff_st = lane_ok ? voff : 0xffffffffu;  // dropped by the range check
  if (lane_ok) {
    s_f[threadIdx.x] = f;
    s_col[threadIdx.x] = c;
  }
  // processed x: [2, W-3] (kpe.hpp:183-184): the first two pixels of a row's first chunk and the last
  // two of its last chunk are not
  uint32_t maskx = 0xffffu;
  if (c == 0) maskx &= 0xfffcu;
  if (c == Wc - 1) maskx &= 0x3fffu;
  maskx |= maskx << 16;
  if (!lane_ok) maskx = 0;

  const int H = (int)p.H, W = (int)p.W, RA = (int)p.rows_per_strip;  // RA: a multiple of 6
  const int ya = (int)blockIdx.y * 2 * RA, yb = ya + RA;  // first rows of the two strips
  // rows ya + j and yb + j (clamped to the frame: the window rows of unprocessed pixels); j > RA is not
  // wanted (the strip's last window row is RA): out of range, no memory access
  auto load_rows = [&](int j) -> BsRaw {
    BsRaw w;
    const int ra = min(max(ya + j, 0), H - 1), rb = min(max(yb + j, 0), H - 1);
    const uint32_t vo = j <= RA ? voff : 0xffffffffu;
    w.a = __builtin_amdgcn_raw_buffer_load_b128(rin, vo, (uint32_t)(ra * W), RMX_BS_LD_AUX);
    w.b = __builtin_amdgcn_raw_buffer_load_b128(rin, vo, (uint32_t)(rb * W), RMX_BS_LD_AUX);
    return w;
  };
  // A 16-byte store reads its data registers a moment after it issues: an instruction that overwrites them
  // right behind it (the register allocator reuses them at once; the compiler sees no hazard because the
  // store has a scalar offset) corrupted the first dword now and then (tools/cmp_kpe_paths.py: the lane
  // offset of the NEXT store appeared in the medians).  Both lane offsets are computed first, nothing may
  // be scheduled between the stores and the wait states behind them.
  auto store_rows = [&](int oa, int ob, const uint32_t (&a4)[4], const uint32_t (&b4)[4]) {
    const u32x4 va = {a4[0], a4[1], a4[2], a4[3]}, vb = {b4[0], b4[1], b4[2], b4[3]};
    const uint32_t voa = oa < H ? voff_st : 0xffffffffu, vob = ob < H ? voff_st : 0xffffffffu;
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_raw_buffer_store_b128(va, rout, voa, (uint32_t)(oa * W), RMX_BS_ST_AUX);
    __builtin_amdgcn_raw_buffer_store_b128(vb, rout, vob, (uint32_t)(ob * W), RMX_BS_ST_AUX);
    asm volatile("s_nop 1");
    __builtin_amdgcn_sched_barrier(0);
  };
  const uint32_t zero4[4] = {0, 0, 0, 0};
  bs::Row T0, T1, T2;
  BsRaw w0 = load_rows(-1), w1 = load_rows(0);
  BsRaw pf0 = load_rows(1);
  store_rows(H, H, zero4, zero4);  // dropped: the loads in flight see the same younger operations as in the loop
  BsRaw pf1 = load_rows(2);
  store_rows(H, H, zero4, zero4);
#if RMX_BS_DEPTH == 3
  BsRaw pf2 = load_rows(3);
  store_rows(H, H, zero4, zero4);
#endif
  bs_make_row(w0, T0);
  bs_make_row(w1, T1);
  // one row step: rows j of both strips from the window (A = j-1, B = j, Cn = j+1, made from PF);
  // the prefetch slot is refilled with rows j + 1 + RMX_BS_DEPTH
#if defined(RMX_BS_COPY)  /* ablation: loads and stores only (tools/ab_kpe_bs.sh) */
#define RMX_BS_SELECT(A, B, Cn)                                                                            \
  oa4[0] = Cn.c[0] & valid, oa4[1] = Cn.c[1], oa4[2] = Cn.c[2], oa4[3] = Cn.c[3];                          \
  ob4[0] = Cn.l[0] & valid, ob4[1] = Cn.l[1], ob4[2] = Cn.l[2], ob4[3] = Cn.l[3];
#else
#define RMX_BS_SELECT(A, B, Cn)                                                                            \
  bs::select4of9(A, B, Cn, nr);                                                                            \
  RMX_BS_CAND(jj, bs::differs(nr, B.c, valid));                                                            \
  bs::native_planes_of_inverted_rank(nr, nat);                                                             \
  nat[0] &= valid, nat[1] &= valid, nat[2] &= valid, nat[3] &= valid;                                      \
  bs::bytes_from_planes(nat, oa4, ob4);
#endif
  uint32_t cdw[3] = {0, 0, 0};
  BsRoll roll{ROLL ? s_wave + wv : nullptr, s_f + wv * 64u, s_col + wv * 64u, lane, (uint32_t)ya, (uint32_t)RA, f0, 0u};
  auto ring_write = [&](int q, const BsRaw& w) {  // strip row q (-1 .. RA) of both strips
    const u32x4 c = {w.a[0] | w.a[1] << 4, w.a[2] | w.a[3] << 4, w.b[0] | w.b[1] << 4, w.b[2] | w.b[3] << 4};
    *reinterpret_cast<u32x4*>(&roll.L->ring[bsr_slot(q)][lane][0]) = c;
  };
  if constexpr (ROLL) {
    ring_write(-1, w0);
    ring_write(0, w1);
  }
#define RMX_BS_CAND(jj, word)                     \
  if constexpr (ROLL) cdw[rr % 3] = (word);       \
  else s_c[jj][threadIdx.x] = (word);
#define RMX_BS_RING(PF) \
  if constexpr (ROLL) ring_write(jj + 1, PF);
#if defined(RMX_BSR_NOROLL)  /* ablation: the streaming pass and the ring writes alone (tools/ab_kpe_bs.sh) */
#define RMX_BS_ROLLNOW(R)
#else
#define RMX_BS_ROLLNOW(R) \
  if constexpr (ROLL && (R) % 3 == 0) bs_roll(p, roll, cdw[0], cdw[1], cdw[2], jj - 3);
#endif
#define RMX_BS_STEP(A, B, Cn, PF, R)                                                                       \
  {                                                                                                        \
    const int jj = j + R;                                                                                  \
    constexpr int rr = R;                                                                                  \
    (void)rr;                                                                                              \
    bs_make_row(PF, Cn);                                                                                   \
    RMX_BS_RING(PF)                                                                                        \
    PF = load_rows(jj + 1 + RMX_BS_DEPTH);                                                                 \
    RMX_BS_ROLLNOW(R) /* the three rows before this one: their windows' last row is the one just written */ \
    const int oa = ya + jj, ob = yb + jj;                                                                  \
    uint32_t valid = maskx;                                                                                \
    if (!(oa >= 2 && oa + 5 <= H)) valid &= 0xffff0000u; /* processed rows: [2, H-5] (kpe.hpp:268) */      \
    if (!(ob >= 2 && ob + 5 <= H)) valid &= 0x0000ffffu;                                                   \
    uint32_t nr[4], nat[4], oa4[4], ob4[4];                                                                \
    RMX_BS_SELECT(A, B, Cn)                                                                                \
    store_rows(oa, ob, oa4, ob4);                                                                          \
  }
  for (int j = 0; j < RA; j += 6) {
#if RMX_BS_DEPTH == 3
    RMX_BS_STEP(T0, T1, T2, pf0, 0)
    RMX_BS_STEP(T1, T2, T0, pf1, 1)
    RMX_BS_STEP(T2, T0, T1, pf2, 2)
    RMX_BS_STEP(T0, T1, T2, pf0, 3)
    RMX_BS_STEP(T1, T2, T0, pf1, 4)
    RMX_BS_STEP(T2, T0, T1, pf2, 5)
#else
    RMX_BS_STEP(T0, T1, T2, pf0, 0)
    RMX_BS_STEP(T1, T2, T0, pf1, 1)
    RMX_BS_STEP(T2, T0, T1, pf0, 2)
    RMX_BS_STEP(T0, T1, T2, pf1, 3)
    RMX_BS_STEP(T1, T2, T0, pf0, 4)
    RMX_BS_STEP(T2, T0, T1, pf1, 5)
#endif
  }
#undef RMX_BS_STEP
#undef RMX_BS_SELECT
#undef RMX_BS_CAND
#undef RMX_BS_ROLLNOW
#undef RMX_BS_RING
#if !defined(RMX_BSR_NOROLL)
  if constexpr (ROLL) {
    bs_roll(p, roll, cdw[0], cdw[1], cdw[2], RA - 3);  // the strip's last three rows (RA is a multiple of six)
    bs_flush(p, roll);
  }
#endif
}

// ---- finishing the candidates -----------------------------------------------------------------------------
// compute_pixel's second half (kpe.hpp:316-321) asks two questions about p5 = median_pixel(H5, 12), the 12th
// largest rank of the 5x5 window: is it p3 (then the pixel is no keypoint) and, for keypoints, is it p1 (the
// weight).  Neither needs p5 itself: with C(t) = #(rank >= t), p5 == v  <=>  C(v) >= 12 > C(v + 1)
// (kpe.hpp:326-340 returns the largest t whose count from the top reaches 12).  So every candidate gets two
// counts (p3, p3 + 1) instead of kpe.hip's four-step search, and only the survivors (7-18 %) — queued per
// wave and handled 64 at a time — get two more (p1, p1 + 1), their code bytes and the append.
__device__ __forceinline__ void bs_window(const CandData& d, uint32_t x, BsWin& W) {
  const uint32_t s = (x - 2) & 3u;
#pragma unroll
  for (int r = 0; r < 5; ++r) {
    const uint32_t lo = d.lo[r] & 0x0f0f0f0fu, hi = d.hi[r] & 0x0f0f0f0fu;
    W.w[r] = __builtin_amdgcn_alignbyte(hi, lo, s);
    W.e[r] = (hi >> (8 * s)) & 0xffu;
  }
}
__device__ __forceinline__ uint32_t map_ord80(uint32_t w) {  // rank | 0x80 per byte
  return map16<ord4(0) | 0x80808080u, ord4(4) | 0x80808080u, ord4(8) | 0x80808080u, ord4(12) | 0x80808080u>(w);
}
// ranks of the 25 pixels (bit 7 of every byte set; the three unused bytes of v[6] hold rank 0, which only a
// threshold of 0 counts — and then the count is past 12 anyway)
__device__ __forceinline__ void bs_ranks(const BsWin& W, uint32_t (&v)[7]) {
#pragma unroll
  for (int r = 0; r < 5; ++r) v[r] = map_ord80(W.w[r]);
  v[5] = map_ord80(W.e[0] | W.e[1] << 8 | W.e[2] << 16 | W.e[3] << 24);
  v[6] = map_ord80(W.e[4]);
}
// 128 * #(rank >= t), t in 0..16
__device__ __forceinline__ uint32_t bs_count_ge(const uint32_t (&v)[7], uint32_t t) {
  const uint32_t T = t * 0x01010101u;
  uint32_t acc = 0;
#pragma unroll
  for (int i = 0; i < 7; ++i) acc = __builtin_amdgcn_sad_u8((v[i] - T) & 0x80808080u, 0u, acc);
  return acc;
}
__device__ __forceinline__ bool bs_rank12_is(const uint32_t (&v)[7], uint32_t t) {
  return bs_count_ge(v, t) >= 12u * 128u && bs_count_ge(v, t + 1) < 12u * 128u;
}
// 4th largest of the inner 3x3 (kpe.hpp:313-314) from the ranks of rows 1-3, with kpe.hip's network
__device__ __forceinline__ uint32_t bs_p3(const uint32_t (&v)[7]) {
  uint32_t lo[3], md[3], hi[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const uint32_t q = v[r + 1];
    sort3((q >> 8) & 0xffu, (q >> 16) & 0xffu, q >> 24, lo[r], md[r], hi[r]);
  }
  const uint32_t a = umax3(lo[0], lo[1], md[2]);
  const uint32_t b = umed3(md[0], md[1], hi[2]);
  const uint32_t c = umed3(hi[0], hi[1], lo[2]);
  return umed3(a, b, c) & 0xfu;
}
// stage 1: is the candidate a keypoint?
__device__ __forceinline__ bool bs_is_keypoint(uint32_t x, const CandData& d) {
  BsWin W;
  uint32_t v[7];
  bs_window(d, x, W);
  bs_ranks(W, v);
  return !bs_rank12_is(v, bs_p3(v));
}
// stage 2: weight, code bytes (kpe.hpp:342-379), the record goes to row `slot` of the frame's raw table
__device__ __forceinline__ void bs_emit_keypoint(const KpeParams& p, uint32_t x, uint32_t y, uint32_t f, bool kp, uint32_t slot,
                                                 const CandData& d) {
  BsWin W;
  uint32_t v[7];
  bs_window(d, x, W);
  bs_ranks(W, v);
  const uint32_t p1 = (v[2] >> 16) & 0xfu;
  const uint32_t weight = bs_rank12_is(v, p1) ? 1u : 2u;
  const uint32_t* w = W.w;
  const uint32_t* e = W.e;
  const uint32_t a0 = nib16(w[0]);
  const uint32_t b2 = (w[1] & 0xfu) | (e[0] << 4);
  const uint32_t a1 = nib16((w[1] >> 8) | (e[1] << 24));
  const uint32_t a2 = nib16(w[2]);
  const uint32_t b7 = (w[3] & 0xfu) | (e[2] << 4);
  const uint32_t a3 = nib16((w[3] >> 8) | (e[3] << 24));
  const uint32_t a4 = nib16(w[4]);
  const uint32_t b12 = weight | (e[4] << 4);
  if (kp) {
    if (slot < p.kmax) {
      RawDesc* dd = p.raw + (size_t)f * p.kmax + slot;
      const uint4 c4 = make_uint4(a0 | b2 << 16 | (a1 & 0xffu) << 24, (a1 >> 8) | a2 << 8 | b7 << 24, a3 | a4 << 16, b12);
      const uint4 m4 = make_uint4(x | y << 16, 0u, 0u, 0u);
      reinterpret_cast<uint4*>(dd)[0] = c4;
      reinterpret_cast<uint4*>(dd)[1] = m4;
    } else {
      atomicOr(&p.status[0], 1u);
    }
  }
}

// the same, for the rolling finish: the keypoint's 16 code bytes (weight included); the caller appends the record
__device__ __forceinline__ void bs_record(const BsWin& W, const uint32_t (&v)[7], uint4& c4) {
  const uint32_t p1 = (v[2] >> 16) & 0xfu;
  const uint32_t weight = bs_rank12_is(v, p1) ? 1u : 2u;
  const uint32_t* w = W.w;
  const uint32_t* e = W.e;
  const uint32_t a0 = nib16(w[0]);
  const uint32_t b2 = (w[1] & 0xfu) | (e[0] << 4);
  const uint32_t a1 = nib16((w[1] >> 8) | (e[1] << 24));
  const uint32_t a2 = nib16(w[2]);
  const uint32_t b7 = (w[3] & 0xfu) | (e[2] << 4);
  const uint32_t a3 = nib16((w[3] >> 8) | (e[3] << 24));
  const uint32_t a4 = nib16(w[4]);
  const uint32_t b12 = weight | (e[4] << 4);
  c4 = make_uint4(a0 | b2 << 16 | (a1 & 0xffu) << 24, (a1 >> 8) | a2 << 8 | b7 << 24, a3 | a4 << 16, b12);
}

// LDS: s_c[ra][256] candidate words (dynamic), then the hand-out tables.
#ifndef RMX_BS_WAVES
#define RMX_BS_WAVES 4
#endif
#ifndef RMX_BSR_WGS
#define RMX_BSR_WGS 4
#endif
// the whole extraction with the rolling finish: four independent waves, no workgroup barrier
__global__ __launch_bounds__(256, RMX_BSR_WGS) void kpe_bs_ring(KpeParams p) {
  __shared__ uint32_t s_f[256], s_col[256];
  __shared__ BsWaveLds s_wave[4];
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) p.status[15] = 0;  // (keypoints of sampled frames: kpe_table)
  kpe_bs_body<true>(p, nullptr, s_f, s_col, s_wave);
}
__global__ __launch_bounds__(256, RMX_BS_WAVES) void kpe_bs(KpeParams p) {
  extern __shared__ __attribute__((aligned(16))) uint32_t s_dyn[];
  uint32_t(*s_c)[256] = reinterpret_cast<uint32_t(*)[256]>(s_dyn);
  __shared__ uint32_t s_f[256], s_col[256];
  __shared__ uint32_t s_scan[8];
  __shared__ uint32_t s_list[KPE_BS_LIST];  // candidates of a round: thread | row << 8 | bit << 13
  __shared__ uint32_t s_surv[KPE_BS_LIST];  // its keypoints: the same | place among its frame's keypoints of this round << 18
  __shared__ uint32_t s_fcnt[256], s_fbase[256];  // per frame of the workgroup: keypoints of this round, their first table row
  __shared__ uint32_t s_ns;
  const uint32_t tid = threadIdx.x;
  const uint32_t RA = p.rows_per_strip;
#if defined(RMX_BS_STAMP)  /* diagnostic build (tools/bs_stamps.py): phase times of every workgroup, in p.cand */
  uint64_t stamp[8];
  uint32_t nstamp = 0;
#define RMX_STAMP() stamp[nstamp++] = __builtin_amdgcn_s_memtime()
#else
#define RMX_STAMP()
#endif
  RMX_STAMP();
  for (uint32_t k = 0; k < RA; ++k) s_c[k][tid] = 0;
  if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) p.status[15] = 0;
  kpe_bs_body<false>(p, s_c, s_f, s_col, nullptr);
  RMX_STAMP();  // 1: streamed
#if defined(RMX_BS_FINPRIO)  /* experiment: the finishing phase issues ahead of other workgroups' streaming waves */
  __builtin_amdgcn_s_setprio(RMX_BS_FINPRIO);
#endif
  const uint32_t lane = tid & 63u, wv = tid >> 6;
  uint32_t cnt = 0;  // this thread's candidates (RA is a multiple of six: six words in flight)
  for (uint32_t k0 = 0; k0 < RA; k0 += 6) {
    uint32_t w[6];
#pragma unroll
    for (uint32_t j = 0; j < 6; ++j) w[j] = s_c[k0 + j][tid];
#pragma unroll
    for (uint32_t j = 0; j < 6; ++j) cnt += (uint32_t)__builtin_popcount(w[j]);
  }
  uint32_t incl = cnt;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    uint32_t t = (uint32_t)__shfl_up((int)incl, o);
    if (lane >= (uint32_t)o) incl += t;
  }
  if (lane == 63) s_scan[wv] = incl;
  __syncthreads();  // also: s_f, s_col and the candidate words of all threads
  uint32_t wbase = 0, total = 0;
  for (uint32_t k = 0; k < 4; ++k) {
    uint32_t sc = s_scan[k];
    if (k < wv) wbase += sc;
    total += sc;
  }
#if defined(RMX_BS_NOFIN)  /* ablation: the streaming pass alone */
  total = 0;
#endif
  const uint32_t my_base = wbase + incl - cnt;
  const uint32_t ya = blockIdx.y * 2 * RA;
  const uint32_t first_frame = bs_first_frame(p, bs_logical_block() * 4);
  // Hand-out: every thread writes its own candidates (thread, row, bit) to a list in LDS at the places its
  // prefix count assigns them; the list is then read densely.  (kpe.hip's hand-out searches the prefix
  // counts for the owner of candidate k, then its words, then the bit: ~100 instructions per candidate
  // against ~8 per candidate written and a handful to read it back.)  Workgroups with more candidates than
  // the list holds go around again.
  auto decode = [&](uint32_t e, uint32_t& x, uint32_t& y, uint32_t& f) {
    const uint32_t t = e & 255u, kb = (e >> 8) & 31u, b = e >> 13;
    x = s_col[t] * 16 + (b & 15u);
    y = ya + (b >> 4) * RA + kb;
    f = s_f[t];
  };
  // Table rows are reserved once per round and frame: every keypoint takes a place among its frame's keypoints
  // with an LDS atomic, one lane per frame then adds the round's count to the frame's global counter (all
  // frames of the workgroup in one instruction), and the records are written without further atomics.  (A
  // returning global atomic per wave and frame in the resolve loop cost ~2 us of latency each, one after the
  // other: the workgroup's finish, not its instruction count, was what the kernel waited for.)
  const uint32_t fwg = first_frame;  // the workgroup's frames are [fwg, fwg + 256)
  s_fcnt[tid] = 0;
  if (tid == 0) s_ns = 0;
  RMX_STAMP();  // 2: counted
  for (uint32_t r0 = 0; r0 < total; r0 += KPE_BS_LIST) {  // uniform: barriers inside
    __syncthreads();  // the previous round is done (first round: s_fcnt, s_ns)
    if (my_base < r0 + KPE_BS_LIST && my_base + cnt > r0) {
      uint32_t idx = my_base;
      for (uint32_t k0 = 0; k0 < RA && idx < r0 + KPE_BS_LIST; k0 += 6) {
        uint32_t w6[6];
#pragma unroll
        for (uint32_t j = 0; j < 6; ++j) w6[j] = s_c[k0 + j][tid];
#pragma unroll
        for (uint32_t j = 0; j < 6; ++j) {
          uint32_t w = w6[j];
          const uint32_t c = (uint32_t)__builtin_popcount(w);
          if (idx + c <= r0) w = 0;  // (all before this round)
          else
            while (w) {
              const uint32_t b = (uint32_t)__builtin_ctz(w);
              w &= w - 1;
              const uint32_t at = idx + c - 1u - (uint32_t)__builtin_popcount(w);  // rank of bit b among the word's bits
              if (at >= r0 && at < r0 + KPE_BS_LIST) s_list[at - r0] = tid | (k0 + j) << 8 | b << 13;
            }
          idx += c;
        }
      }
    }
    __syncthreads();
    if (r0 == 0) RMX_STAMP();  // 3: listed
    const uint32_t m = min(KPE_BS_LIST, total - r0);
    // stage 1: RMX_BS_S1 candidates per thread and pass, all gathers in flight before the first test
    for (uint32_t k0 = 0; k0 < m; k0 += 256 * RMX_BS_S1) {
      if (k0 + (tid & ~63u) >= m) break;  // this wave's lanes are past the round's last candidate
      uint32_t e[RMX_BS_S1], x[RMX_BS_S1], y[RMX_BS_S1], f[RMX_BS_S1];
      CandData d[RMX_BS_S1];
#pragma unroll
      for (int c = 0; c < RMX_BS_S1; ++c) {
        const uint32_t k = k0 + 256 * c + tid;
        e[c] = k < m ? s_list[k] : 0xffffffffu;
        x[c] = 2, y[c] = 2, f[c] = fwg;
        if (e[c] != 0xffffffffu) decode(e[c], x[c], y[c], f[c]);
#if defined(RMX_BS_NOGATHER)  /* ablation: the same loads, all from one window (served by the vector L1) */
        gather_candidate(p, 2, 2, fwg, d[c]);
#elif defined(RMX_BS_L2GATHER)  /* ablation: the candidates' own windows, but of the batch's first four frames (L2-resident) */
        gather_candidate(p, x[c], y[c], f[c] & 3u, d[c]);
#else
        gather_candidate(p, x[c], y[c], f[c], d[c]);
#endif
      }
#pragma unroll
      for (int c = 0; c < RMX_BS_S1; ++c) {
        if (k0 + 256 * c + (tid & ~63u) >= m) break;  // uniform
        const bool kp = e[c] != 0xffffffffu && bs_is_keypoint(x[c], d[c]);
        const uint64_t km = __builtin_amdgcn_ballot_w64(kp);
        if (km) {
          uint32_t base = 0;
          const uint32_t leader = (uint32_t)__builtin_amdgcn_readfirstlane((int)__builtin_ctzll(km));
          if (lane == leader) base = atomicAdd(&s_ns, (uint32_t)__builtin_popcountll(km));
          base = (uint32_t)__builtin_amdgcn_readlane((int)base, (int)leader);
          if (kp) s_surv[base + mbcnt(km)] = e[c] | atomicAdd(&s_fcnt[f[c] - fwg], 1u) << 18;
        }
      }
    }
    __syncthreads();
    if (r0 == 0) RMX_STAMP();  // 4: tested
    {  // one lane per frame: this round's rows
      const uint32_t c = s_fcnt[tid];
      if (c) {
        s_fbase[tid] = atomicAdd(&p.kcount[fwg + tid], c);
        s_fcnt[tid] = 0;
      }
    }
    __syncthreads();
    if (r0 == 0) RMX_STAMP();  // 5: rows reserved
    const uint32_t ns = s_ns;
    for (uint32_t k0 = 0; k0 < ns; k0 += 512) {
      if (k0 + (tid & ~63u) >= ns) break;
      uint32_t e[2], x[2], y[2], f[2];
      CandData d[2];
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const uint32_t k = k0 + 256 * c + tid;
        e[c] = k < ns ? s_surv[k] : 0xffffffffu;
        x[c] = 2, y[c] = 2, f[c] = fwg;
        if (e[c] != 0xffffffffu) decode(e[c] & 0x3ffffu, x[c], y[c], f[c]);
        gather_candidate(p, x[c], y[c], f[c], d[c]);
      }
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        if (k0 + 256 * c + (tid & ~63u) >= ns) break;
        const bool kp = e[c] != 0xffffffffu;
        bs_emit_keypoint(p, x[c], y[c], f[c], kp, kp ? s_fbase[f[c] - fwg] + (e[c] >> 18) : 0u, d[c]);
      }
    }
    __syncthreads();  // s_ns has been read by everyone
    if (r0 == 0) RMX_STAMP();  // 6: emitted
    if (tid == 0) s_ns = 0;
  }
#if defined(RMX_BS_STAMP)
  if (tid == 0) {
    uint64_t* out = reinterpret_cast<uint64_t*>(p.cand) + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8;
    for (uint32_t k = 0; k < 7; ++k) out[k] = k < nstamp ? stamp[k] : 0;
    out[7] = total | (uint64_t)s_ns << 32;
  }
#endif
#undef RMX_STAMP
}

}  // namespace rmx
