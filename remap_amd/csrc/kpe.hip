// kpe.hip — batched keypoint extraction for gfx950 (MI355X).
//
// Replaces kpe::extractor<Grid,Overlap>::extract (reference kpe.hpp:92-108) with its
// callees sum_row / col_in / compute_pixel / median_pixel / encode_keypoint
// (kpe.hpp:111-379).  The reference builds two sliding 16-bin histograms per pixel
// (3x3 and 5x5) and scans them from the top; the same results are obtained here with a
// rank-selection network:
//
//   * every pixel is mapped once to a KEY = rank << 4 | native (one v_perm-based LUT per
//     4 packed pixels), so ordering keys == ordering luminance ranks and the selected
//     key carries the native value of the selected pixel (no inverse LUT);
//   * 4th-largest-of-9 (median_pixel(H3, 4), kpe.hpp:313,326-340): each row's horizontal
//     triple (x-1,x,x+1) is sorted once (min3/med3/max3) and reused by the three output
//     rows that see it; three sorted triples A,B,C are merged with
//         med3( max3(A.lo,B.lo,C.md), med3(A.md,B.md,C.hi), med3(A.hi,B.hi,C.lo) )
//     which equals the 4th largest of the nine (exhaustively verified over the 0/1
//     principle, tests/test_host_logic.py) — 7 VALU ops per pixel;
//   * pixels whose centre differs from that value (1-8 % on the synthetic streams) are
//     candidates.  The shipped kernel, kpe_mf, keeps a workgroup's candidate flags in LDS (one word per
//     lane and 8 rows) and, after its walk, hands the set bits out densely: the 5x5 window is gathered
//     from L2 / Infinity Cache (the workgroup has just streamed those lines), the 12th-largest-of-25
//     (median_pixel(H5, 12)) is found by a 4-step binary search with SWAR byte compares + v_sad_u8
//     counting, and keypoints are encoded and appended.  kpe_main + kpe_finish are the same two halves as
//     separate launches with a 1-bit-per-pixel candidate plane in HBM between them (RMX_KPE_PAIR: the form
//     of rounds 1-2a, kept for A/B timings).
//
// Since round 3 this file serves frame widths that are NOT multiples of 16; the others go through
// kpe_bs.hip (3x3 pass on bit planes), which reuses gather_candidate from here.
//
// Work decomposition: a lane owns one dword (4 pixels) of a row and walks down a strip of
// rows keeping three rows of sorted triples in registers; horizontally adjacent pixels
// come from the neighbouring lanes by DPP wave shifts (lanes 0 and 63 of a wave are halo
// lanes).  Lanes are laid over the batch's frames back to back ("virtual dword columns"),
// so waves stay full for any frame width; pixels whose window would cross a frame edge
// are exactly the border pixels the reference never processes.
//
// HBM traffic of the streaming half: every frame byte is read once (+3 % halo re-reads served by L2),
// every median byte written once; kpe_main adds 1 bit per pixel of candidate plane (+6 %).
#include "common.h"

namespace rmx {

struct KpeParams {
  const uint8_t* frames;
  uint8_t* median;
  RawDesc* raw;
  uint32_t* kcount;
  uint32_t* status;  // [0] overflow flag
  uint32_t n, W, H, Wd;
  uint64_t V;        // n * Wd virtual dword columns
  uint32_t groups;   // ceil(V / 62)
  uint32_t rows_per_strip;  // multiple of 8
  uint32_t kmax;
  uint32_t* cand;    // candidate bit plane: n * HB * Wd dwords
  uint32_t HB;       // ceil(H / 8)
  uint32_t strips_long, rows_short;  // kpe_pc: the first strips_long strips have rows_per_strip rows, the others rows_short
};

#ifndef RMX_XLANE_SHFL
// DPP wave shifts (GFX9 family): wave_shr:1 -> lane i reads lane i-1; wave_shl:1 -> lane i+1.
// Lanes without a source (lane 0 / lane 63) read 0: they are halo lanes whose own results
// are never stored.
__device__ __forceinline__ uint32_t xlane_prev(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, true);
}
__device__ __forceinline__ uint32_t xlane_next(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, true);
}
#else
__device__ __forceinline__ uint32_t xlane_prev(uint32_t v) { return (uint32_t)__shfl_up((int)v, 1); }
__device__ __forceinline__ uint32_t xlane_next(uint32_t v) { return (uint32_t)__shfl_down((int)v, 1); }
#endif

// row (0..7) of an 8-row block flagged by bit `bb` of a pixel's byte in a candidate word
__device__ __forceinline__ uint32_t cand_row(uint32_t bb) { return bb < 4u ? 3u - bb : 11u - bb; }

struct RowT {
  uint32_t lo[4], md[4], hi[4];
  uint32_t key;  // the row's 4 keys, packed
};

__device__ __forceinline__ void sort3(uint32_t a, uint32_t b, uint32_t c, uint32_t& lo, uint32_t& md, uint32_t& hi) {
  lo = umin3(a, b, c);
  md = umed3(a, b, c);
  hi = umax3(a, b, c);
}

// Row of native pixels (4 per lane) -> keys -> horizontally sorted triples.
__device__ __forceinline__ void make_row(uint32_t w, RowT& R) {
  uint32_t k = map_key(w);
  uint32_t k0 = k & 0xff, k1 = (k >> 8) & 0xff, k2 = (k >> 16) & 0xff, k3 = k >> 24;
  uint32_t kl = xlane_prev(k3);
  uint32_t kr = xlane_next(k0);
  sort3(kl, k0, k1, R.lo[0], R.md[0], R.hi[0]);
  sort3(k0, k1, k2, R.lo[1], R.md[1], R.hi[1]);
  sort3(k1, k2, k3, R.lo[2], R.md[2], R.hi[2]);
  sort3(k2, k3, kr, R.lo[3], R.md[3], R.hi[3]);
  R.key = k;
}

// 4th largest of the 3x3 window for the lane's 4 pixels, packed (keys).
__device__ __forceinline__ uint32_t merge_rows(const RowT& A, const RowT& B, const RowT& C) {
  uint32_t r[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    uint32_t a = umax3(A.lo[j], B.lo[j], C.md[j]);
    uint32_t b = umed3(A.md[j], B.md[j], C.hi[j]);
    uint32_t c = umed3(A.hi[j], B.hi[j], C.lo[j]);
    r[j] = umed3(a, b, c);
  }
  // three v_lshl_or_b32 (written out: the compiler turns the expression into two shifts, one
  // shift-or and a 3-input or)
  uint32_t lo, hi, pk;
  asm("v_lshl_or_b32 %0, %1, 8, %2" : "=v"(lo) : "v"(r[1]), "v"(r[0]));
  asm("v_lshl_or_b32 %0, %1, 8, %2" : "=v"(hi) : "v"(r[3]), "v"(r[2]));
  asm("v_lshl_or_b32 %0, %1, 16, %2" : "=v"(pk) : "v"(hi), "v"(lo));
  return pk;
}

// [b0|b1<<4, b2|b3<<4] of a dword of four nibble-valued bytes, as a 16-bit value.
__device__ __forceinline__ uint32_t nib16(uint32_t w) {
  uint32_t t = w | (w >> 4);
  return perm(0u, t, 0x0c0c0200u);
}

// Wave-uniform raw buffer descriptor (T8/T20 of the CDNA guide): 32-bit per-lane offsets,
// hardware range check (an out-of-range lane's store is dropped, its load returns 0), so
// the row loop needs neither 64-bit address arithmetic nor exec-masked branches.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* base, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), (short)0, (int)bytes, 0x00020000);
}

// ---- kernel 1: streaming pass -------------------------------------------------------------------
// frames -> median + candidate bit plane (1 bit per pixel: centre != 4th largest of 3x3).
// Candidate word: one dword per (frame, 8-row block yb, dword column c); byte j = pixel
// x = 4c + j, its bit cand_bit(r) = row y = 8*yb + r (rows 0-3 at bits 3..0, rows 4-7 at bits 7..4:
// the order in which four rows double their way up a nibble, see the row step).
// ALIGNED = frame width is a multiple of 4.  Otherwise rows start at unaligned bytes (the
// loads simply are unaligned dwords), the last dword column of a row is partial, and its
// median bytes are stored one by one so the next row's first pixels are not overwritten.
constexpr uint32_t KPE_WPS_MAX = 6;  // 8-row blocks per strip a workgroup of kpe_mf keeps in LDS (strips of <= 48 rows)

// The streaming pass.  LDS_CAND = false: kpe_main (candidate words to the plane in HBM);
// true: kpe_mf (candidate words to s_c[8-row block of the strip][thread]).  f_out / c_out /
// ok_out: the lane's frame, dword column and whether it owns one (not a halo lane).
template <bool ALIGNED, bool LDS_CAND>
__device__ __forceinline__ void kpe_main_body(const KpeParams& p, uint32_t (*s_c)[256], uint32_t& f_out, uint32_t& c_out, bool& ok_out) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  ok_out = false;
  f_out = 0;
  c_out = 0;

  // XCD-aware block remap (bijective): blocks that share an XCD get neighbouring groups.
  const uint32_t nb = gridDim.x, bid = blockIdx.x;
  const uint32_t bq = nb >> 3, br = nb & 7u, xcd = bid & 7u;
  const uint32_t lb = (xcd < br ? xcd * (bq + 1) : br * (bq + 1) + (xcd - br) * bq) + (bid >> 3);
  const uint32_t g = lb * 4 + wv;
  if (g >= p.groups) return;  // whole wave; no block-level barriers are used below

  // virtual dword column of this lane; lanes 0 and 63 are halo lanes
  const int64_t L0 = (int64_t)g * 62 - 1;  // wave-uniform
  const int64_t L = L0 + (int64_t)lane;
  const bool lane_ok = lane >= 1 && lane <= 62 && L >= 0 && L < (int64_t)p.V;
  const uint64_t Lc = L < 0 ? 0 : (L >= (int64_t)p.V ? p.V - 1 : (uint64_t)L);
  const uint32_t f = (uint32_t)(Lc / p.Wd), c = (uint32_t)(Lc % p.Wd);
  // wave-uniform base = first byte of the frame holding the wave's first (clamped) column
  const uint64_t Lb = L0 < 0 ? 0 : (uint64_t)L0;
  const uint32_t f0 = (uint32_t)(Lb / p.Wd);
  const size_t frame_bytes = (size_t)p.W * p.H;
  const uint32_t voff = (uint32_t)((size_t)(f - f0) * frame_bytes) + c * 4;  // < 3 frames
  // the descriptors end where the batch ends: a partial last dword of the batch reads 0
  const size_t left = (size_t)(p.n - f0) * frame_bytes;
  const uint32_t recs = (uint32_t)(left < 0x7fffffffull ? left : 0x7fffffffull);
  const rsrc_t rin = make_rsrc(p.frames + (size_t)f0 * frame_bytes, recs);
  const rsrc_t rout = make_rsrc(p.median + (size_t)f0 * frame_bytes, recs);
  const uint32_t part = ALIGNED ? 0u : (c == p.Wd - 1 ? (p.W & 3u) : 0u);  // valid bytes of a partial dword
  const rsrc_t rcand = make_rsrc(p.cand + (size_t)f0 * p.HB * p.Wd, 0x7fffffffu);
  const uint32_t voff_st = lane_ok ? voff : 0xffffffffu;  // dropped by the range check
  const uint32_t voff_c = lane_ok ? ((f - f0) * p.HB * p.Wd + c) * 4 : 0xffffffffu;
  ok_out = lane_ok;
  f_out = f;
  c_out = c;

  // per-pixel x validity: processed x in [2, W-3] (kpe.hpp:183-184)
  uint32_t maskx = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    uint32_t x = c * 4 + j;
    if (x >= 2 && x + 3 <= p.W) maskx |= 0x0fu << (8 * j);
  }
  if (!lane_ok) maskx = 0;

  const int H = (int)p.H, W = (int)p.W;
  const int y0 = (int)(blockIdx.y * p.rows_per_strip);  // multiple of 8
  const int y1 = min(H, y0 + (int)p.rows_per_strip);
  const int vy0 = max(y0, 2), vy1 = min(y1, H - 4);  // processed rows: [2, H-5] (kpe.hpp:268)
  auto load_row = [&](int y) -> uint32_t {
    return (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rin, voff, (uint32_t)(y * W), 0);
  };
  auto store_row = [&](int y, uint32_t v) {
    if (!ALIGNED && part) {
      for (uint32_t j = 0; j < part; ++j)
        __builtin_amdgcn_raw_buffer_store_b8((uint8_t)(v >> (8 * j)), rout, voff_st + j, (uint32_t)(y * W), 0);
    } else {
      __builtin_amdgcn_raw_buffer_store_b32(v, rout, voff_st, (uint32_t)(y * W), 0);
    }
  };
  auto store_cand = [&](int yb, uint32_t v) {
    if (LDS_CAND)
      s_c[yb - (y0 >> 3)][threadIdx.x] = lane_ok ? v : 0u;
    else
      __builtin_amdgcn_raw_buffer_store_b32(v, rcand, voff_c, (uint32_t)(yb * (int)p.Wd * 4), 0);
  };
  // unprocessed border rows of this strip are written 0
  for (int y = y0; y < min(vy0, y1); ++y) store_row(y, 0u);
  for (int y = max(vy1, y0); y < y1; ++y) store_row(y, 0u);
  if (vy0 >= vy1) {
    for (int yb = y0 >> 3; yb * 8 < y1; ++yb) store_cand(yb, 0u);
    return;
  }

  RowT T0, T1, T2;
  make_row(load_row(vy0 - 1), T0);
  make_row(load_row(vy0), T1);
  // rows up to y+7 are prefetched, clamped to the last row this strip needs (vy1): running
  // past it would pull up to six rows per strip from HBM for nothing
  uint32_t pf0 = load_row(vy0 + 1), pf1 = load_row(min(vy0 + 2, vy1)), pf2 = load_row(min(vy0 + 3, vy1)),
           pf3 = load_row(min(vy0 + 4, vy1)), pf4 = load_row(min(vy0 + 5, vy1)), pf5 = load_row(min(vy0 + 6, vy1));

  uint32_t C = 0, hold = 0;  // flags of the current group of four rows / of rows 0-3 of the current 8-row block
  // one output row: consume the prefetched row y+1, refill the prefetch slot with row y+7
#define RMX_STEP(A, B, Cn, PF, R)                                                        \
  {                                                                                      \
    const int yy = y + R;                                                                \
    make_row(PF, Cn);                                                                    \
    PF = load_row(min(yy + 7, vy1));                                                   \
    uint32_t pk = merge_rows(A, B, Cn);                                                  \
    store_row(yy, pk & maskx);                                                           \
    uint32_t t = ((((pk ^ B.key) & maskx) + 0x0f0f0f0fu) & 0x10101010u);                 \
    C = (C + C) | t; /* full-rate ops only: a row enters at bit 4 of its pixel's byte, */ \
    if (((uint32_t)yy & 3u) == 3u) { /* four rows fill the high nibble (row k at bit 7-k) */ \
      if (((uint32_t)yy & 4u) == 0u) {                                                    \
        hold = C >> 4;                                                                   \
      } else {                                                                           \
        store_cand(yy >> 3, hold | C);                                                   \
        hold = 0;                                                                        \
      }                                                                                  \
      C = 0;                                                                             \
    }                                                                                    \
  }
  int y = vy0;
  for (; y + 6 <= vy1; y += 6) {  // steady state: six rows, no per-row conditions on validity
    RMX_STEP(T0, T1, T2, pf0, 0)
    RMX_STEP(T1, T2, T0, pf1, 1)
    RMX_STEP(T2, T0, T1, pf2, 2)
    RMX_STEP(T0, T1, T2, pf3, 3)
    RMX_STEP(T1, T2, T0, pf4, 4)
    RMX_STEP(T2, T0, T1, pf5, 5)
  }
  if (y < vy1) RMX_STEP(T0, T1, T2, pf0, 0)
  if (y + 1 < vy1) RMX_STEP(T1, T2, T0, pf1, 1)
  if (y + 2 < vy1) RMX_STEP(T2, T0, T1, pf2, 2)
  if (y + 3 < vy1) RMX_STEP(T0, T1, T2, pf3, 3)
  if (y + 4 < vy1) RMX_STEP(T1, T2, T0, pf4, 4)
#undef RMX_STEP
  // last (partial) block of the strip and blocks that hold only border rows
  if ((vy1 & 7) != 0) {  // the block's last rows are missing: finish the shifts of the open group, write what there is
    if ((vy1 & 3) != 0) C <<= 4 - (vy1 & 3);
    store_cand(vy1 >> 3, (vy1 & 4) ? (hold | C) : ((vy1 & 3) ? C >> 4 : hold));
  }
  for (int yb = (vy1 + 7) >> 3; yb * 8 < y1; ++yb) store_cand(yb, 0u);
}

template <bool ALIGNED>
__global__ __launch_bounds__(256) void kpe_main(KpeParams p) {
  uint32_t f, c;
  bool ok;
  kpe_main_body<ALIGNED, false>(p, nullptr, f, c, ok);
}

// ---- kernel 2: finish the candidates ---------------------------------------------------------------
// compute_pixel's second half (kpe.hpp:316-321) + encode_keypoint (kpe.hpp:342-379).
// A candidate's data moves in two phases so that several candidates per lane can be in
// flight: gather (10 loads from L2 / Infinity Cache: kpe_main just streamed these lines)
// and resolve (rank 12 of 25 by a 4-step binary search with SWAR byte compares counted by
// v_sad_u8, weight, code bytes, append).
struct CandData {
  uint32_t lo[5], hi[5];
};

__device__ __forceinline__ void gather_candidate(const KpeParams& p, uint32_t x, uint32_t y, uint32_t f, CandData& d) {
  const size_t fo = (size_t)f * p.W * p.H;
  const uint8_t* base = p.frames + fo + (size_t)(y - 2) * p.W + ((x - 2) & ~3u);
#pragma unroll
  for (int r = 0; r < 5; ++r) {
    const uint2 v = *reinterpret_cast<const uint2*>(base + (size_t)r * p.W);  // 4-byte aligned dwordx2
    d.lo[r] = v.x;
    d.hi[r] = v.y;
  }
}

// MULTI: the lanes of a wave may belong to different frames (kpe_mf); otherwise f is uniform.
template <bool MULTI = false>
__device__ __forceinline__ void resolve_candidate(const KpeParams& p, uint32_t x, uint32_t y, uint32_t f, bool active,
                                                  const CandData& d) {
  const uint32_t s = (x - 2) & 3u;
  uint32_t w[5], e[5];
#pragma unroll
  for (int r = 0; r < 5; ++r) {
    uint32_t lo = d.lo[r] & 0x0f0f0f0fu, hi = d.hi[r] & 0x0f0f0f0fu;
    w[r] = __builtin_amdgcn_alignbyte(hi, lo, s);  // pixels x-2 .. x+1
    e[r] = (hi >> (8 * s)) & 0xffu;                // pixel  x+2
  }
  uint32_t v[7];
#pragma unroll
  for (int r = 0; r < 5; ++r) v[r] = map_ord(w[r]) | 0x80808080u;
  v[5] = map_ord(e[0] | e[1] << 8 | e[2] << 16 | e[3] << 24) | 0x80808080u;
  v[6] = map_ord(e[4]) | 0x80808080u;  // unused bytes hold rank 0: never >= a threshold >= 1
  const uint32_t p1 = (v[2] >> 16) & 0xfu;
  // p3 = 4th largest of the inner 3x3 (kpe.hpp:313-314), recomputed from the window already
  // gathered (cheaper than touching the median plane's line for one byte) with kpe_main's
  // network: three row triples, one merge — 22 ops against 58 for a search on the rank
  uint32_t p3;
  {
    uint32_t lo[3], md[3], hi[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const uint32_t q = v[r + 1];  // ranks of x-2 .. x+1 (bit 7 of every byte set: same order)
      sort3((q >> 8) & 0xffu, (q >> 16) & 0xffu, q >> 24, lo[r], md[r], hi[r]);
    }
    const uint32_t a = umax3(lo[0], lo[1], md[2]);
    const uint32_t b = umed3(md[0], md[1], hi[2]);
    const uint32_t c = umed3(hi[0], hi[1], lo[2]);
    p3 = umed3(a, b, c) & 0xfu;
  }
  // largest t with #(rank >= t) >= 12  ==  first bin from the top whose running total
  // reaches 12 (kpe.hpp:326-340)
  uint32_t t = 0;
#pragma unroll
  for (uint32_t bit = 8; bit; bit >>= 1) {
    uint32_t c = t | bit;
    uint32_t T = c * 0x01010101u;
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < 7; ++i) acc = __builtin_amdgcn_sad_u8((v[i] - T) & 0x80808080u, 0u, acc);
    if ((acc >> 7) >= 12u) t = c;
  }
  const uint32_t p5 = t;
  const bool kp = active && (p3 != p5);
  const uint64_t km = __builtin_amdgcn_ballot_w64(kp);
  if (km == 0) return;
  const uint32_t weight = (p1 != p5) ? 2u : 1u;
  // code bytes (kpe.hpp:342-379): see oracle/remap_oracle.cpp encode_ref for the derivation
  uint32_t a0 = nib16(w[0]);
  uint32_t b2 = (w[1] & 0xfu) | (e[0] << 4);
  uint32_t a1 = nib16((w[1] >> 8) | (e[1] << 24));
  uint32_t a2 = nib16(w[2]);
  uint32_t b7 = (w[3] & 0xfu) | (e[2] << 4);
  uint32_t a3 = nib16((w[3] >> 8) | (e[3] << 24));
  uint32_t a4 = nib16(w[4]);
  uint32_t b12 = weight | (e[4] << 4);
  // one atomic per wave for the keypoints of its lanes (f is the workgroup's frame): dense
  // stream 0.81 -> 0.78 ms per 10 000 frames, sparse unchanged
  uint32_t slot = 0;
  if (!MULTI) {
    uint32_t base = 0;
    const uint32_t leader = (uint32_t)__builtin_amdgcn_readfirstlane((int)__builtin_ctzll(km));
    if (lane_id() == leader) base = atomicAdd(&p.kcount[f], (uint32_t)__builtin_popcountll(km));
    base = (uint32_t)__builtin_amdgcn_readlane((int)base, (int)leader);
    slot = base + mbcnt(km);
  } else {  // one atomic per wave and distinct frame (a wave spans at most a few)
    uint64_t todo = km;
    while (todo) {
      const uint32_t leader = (uint32_t)__builtin_amdgcn_readfirstlane((int)__builtin_ctzll(todo));
      const uint32_t fl = (uint32_t)__builtin_amdgcn_readlane((int)f, (int)leader);
      const uint64_t same = __builtin_amdgcn_ballot_w64(kp && f == fl);
      uint32_t base = 0;
      if (lane_id() == leader) base = atomicAdd(&p.kcount[fl], (uint32_t)__builtin_popcountll(same));
      base = (uint32_t)__builtin_amdgcn_readlane((int)base, (int)leader);
      if (kp && f == fl) slot = base + mbcnt(same);
      todo &= ~same;
    }
  }
  if (kp) {
    if (slot < p.kmax) {
      RawDesc* dd = p.raw + (size_t)f * p.kmax + slot;
      uint4 c4 = make_uint4(a0 | b2 << 16 | (a1 & 0xffu) << 24, (a1 >> 8) | a2 << 8 | b7 << 24, a3 | a4 << 16, b12);
      uint4 m4 = make_uint4(x | y << 16, 0u, 0u, 0u);
      reinterpret_cast<uint4*>(dd)[0] = c4;
      reinterpret_cast<uint4*>(dd)[1] = m4;
    } else {
      atomicOr(&p.status[0], 1u);
    }
  }
}

constexpr uint32_t FIN_WORDS = 4096;  // candidate words per workgroup (16 per thread)

// One workgroup per (frame, chunk of FIN_WORDS candidate words): every thread owns 16
// consecutive words, the set bits are numbered by a block scan and handed out densely, two
// candidates per thread per pass with both gathers issued before either is resolved.
__global__ __launch_bounds__(256) void kpe_finish(KpeParams p) {
  __shared__ uint32_t s_word[FIN_WORDS];
  __shared__ uint32_t s_tbase[257];
  __shared__ uint32_t s_scan[8];
  const uint32_t tid = threadIdx.x;
  const uint32_t wpf = p.HB * p.Wd;                       // words per frame
  const uint32_t cpf = (wpf + FIN_WORDS - 1) / FIN_WORDS;  // chunks per frame
  const uint32_t f = blockIdx.x / cpf, chunk = blockIdx.x - f * cpf;
  const uint32_t w0 = chunk * FIN_WORDS;
  const uint32_t nw = min(FIN_WORDS, wpf - w0);
  const uint32_t* src = p.cand + (size_t)f * wpf + w0;
  // coalesced load into LDS, then per-thread counts over 16 consecutive words
  for (uint32_t i = tid; i < FIN_WORDS; i += 256) s_word[i] = i < nw ? src[i] : 0u;
  __syncthreads();
  uint32_t cnt = 0;
#pragma unroll
  for (uint32_t i = 0; i < 16; ++i) cnt += (uint32_t)__builtin_popcount(s_word[tid * 16 + i]);
  const uint32_t lane = tid & 63u, wv = tid >> 6;
  uint32_t incl = cnt;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    uint32_t t = (uint32_t)__shfl_up((int)incl, o);
    if (lane >= (uint32_t)o) incl += t;
  }
  if (lane == 63) s_scan[wv] = incl;
  __syncthreads();
  uint32_t wbase = 0, total = 0;
  for (uint32_t k = 0; k < 4; ++k) {
    uint32_t sc = s_scan[k];
    if (k < wv) wbase += sc;
    total += sc;
  }
  s_tbase[tid] = wbase + incl - cnt;
  if (tid == 0) s_tbase[256] = total;
  __syncthreads();

  auto locate = [&](uint32_t k, uint32_t& x, uint32_t& y) {
    uint32_t lo = 0, hi = 256;  // owner thread: largest t with s_tbase[t] <= k
    while (hi - lo > 1) {
      uint32_t mid = (lo + hi) >> 1;
      if (s_tbase[mid] <= k)
        lo = mid;
      else
        hi = mid;
    }
    uint32_t nth = k - s_tbase[lo], wi = lo * 16, w = s_word[wi];
    for (;;) {  // walk the owner's words
      uint32_t c = (uint32_t)__builtin_popcount(w);
      if (nth < c) break;
      nth -= c;
      w = s_word[++wi];
    }
    uint32_t b = 0;
#pragma unroll
    for (uint32_t sft = 16; sft >= 1; sft >>= 1) {
      uint32_t lowc = (uint32_t)__builtin_popcount(w & ((1u << sft) - 1u));
      if (nth >= lowc) {
        nth -= lowc;
        w >>= sft;
        b += sft;
      }
    }
    const uint32_t gw = w0 + wi, yb = gw / p.Wd, c = gw - yb * p.Wd;
    x = c * 4 + (b >> 3);
    y = yb * 8 + cand_row(b & 7u);
  };

  for (uint32_t k0 = 0; k0 < total; k0 += 512) {
    const uint32_t ka = k0 + tid, kb = k0 + 256 + tid;
    const bool act_a = ka < total, act_b = kb < total;
    uint32_t xa = 2, ya = 2, xb = 2, yb = 2;
    if (act_a) locate(ka, xa, ya);
    if (act_b) locate(kb, xb, yb);
    CandData da, db;
    gather_candidate(p, xa, ya, act_a ? f : 0u, da);
    gather_candidate(p, xb, yb, act_b ? f : 0u, db);
    resolve_candidate(p, xa, ya, f, act_a, da);
    if (k0 + 256 < total) resolve_candidate(p, xb, yb, f, act_b, db);
  }
}

// ---- one kernel: the streaming pass, then the workgroup's own candidates -------------------------------
// kpe_finish is bound by its gathers: it re-reads nearly every line of the frames from HBM
// (817 MB per 10 000 frames), long after kpe_main streamed them.  Here the workgroup that
// streamed a tile (4 waves x 62 dword columns x one strip of rows) keeps the tile's candidate
// words in LDS and, after its walk, finishes them itself: the 5x5 windows are gathered while
// the tile's lines are still in the XCD's L2 / the Infinity Cache (rows of the neighbouring
// strips included), there is no candidate plane in HBM and no second launch.  Same arithmetic
// as the pair (kpe_main_body + the hand-out / gather / resolve of kpe_finish).
template <bool ALIGNED>
__global__ __launch_bounds__(256) void kpe_mf(KpeParams p) {
  __shared__ uint32_t s_c[KPE_WPS_MAX][256];
  __shared__ uint32_t s_f[256], s_col[256];
  __shared__ uint32_t s_tbase[257];
  __shared__ uint32_t s_scan[8];
  const uint32_t tid = threadIdx.x;
#pragma unroll
  for (uint32_t k = 0; k < KPE_WPS_MAX; ++k) s_c[k][tid] = 0;
  {
    uint32_t f, c;
    bool ok;
    kpe_main_body<ALIGNED, true>(p, s_c, f, c, ok);
    s_f[tid] = ok ? f : 0xffffffffu;
    s_col[tid] = c;
  }
  __syncthreads();
  const uint32_t wps = p.rows_per_strip >> 3, yb0 = blockIdx.y * wps;
  uint32_t cnt = 0;
#pragma unroll
  for (uint32_t k = 0; k < KPE_WPS_MAX; ++k) cnt += (uint32_t)__builtin_popcount(s_c[k][tid]);
  const uint32_t lane = tid & 63u, wv = tid >> 6;
  uint32_t incl = cnt;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    uint32_t t = (uint32_t)__shfl_up((int)incl, o);
    if (lane >= (uint32_t)o) incl += t;
  }
  if (lane == 63) s_scan[wv] = incl;
  __syncthreads();
  uint32_t wbase = 0, total = 0;
  for (uint32_t k = 0; k < 4; ++k) {
    uint32_t sc = s_scan[k];
    if (k < wv) wbase += sc;
    total += sc;
  }
  s_tbase[tid] = wbase + incl - cnt;
  if (tid == 0) s_tbase[256] = total;
  __syncthreads();
  auto locate = [&](uint32_t k, uint32_t& x, uint32_t& y, uint32_t& f) {
    uint32_t lo = 0, hi = 256;  // owner thread: largest t with s_tbase[t] <= k
    while (hi - lo > 1) {
      uint32_t mid = (lo + hi) >> 1;
      if (s_tbase[mid] <= k)
        lo = mid;
      else
        hi = mid;
    }
    uint32_t nth = k - s_tbase[lo], kb = 0, w = s_c[0][lo];
    for (;;) {  // walk the owner's words
      uint32_t c = (uint32_t)__builtin_popcount(w);
      if (nth < c) break;
      nth -= c;
      w = s_c[++kb][lo];
    }
    uint32_t b = 0;
#pragma unroll
    for (uint32_t sft = 16; sft >= 1; sft >>= 1) {
      uint32_t lowc = (uint32_t)__builtin_popcount(w & ((1u << sft) - 1u));
      if (nth >= lowc) {
        nth -= lowc;
        w >>= sft;
        b += sft;
      }
    }
    x = s_col[lo] * 4 + (b >> 3);
    y = (yb0 + kb) * 8 + cand_row(b & 7u);
    f = s_f[lo];
  };
  for (uint32_t k0 = 0; k0 < total; k0 += 512) {
    const uint32_t ka = k0 + tid, kb = k0 + 256 + tid;
    const bool act_a = ka < total, act_b = kb < total;
    if (__builtin_amdgcn_ballot_w64(act_a) == 0) break;  // this wave's lanes are past the last candidate (no barriers below)
    uint32_t xa = 2, ya = 2, fa = 0, xb = 2, yb = 2, fb = 0;
    if (act_a) locate(ka, xa, ya, fa);
    if (act_b) locate(kb, xb, yb, fb);
    CandData da, db;
    gather_candidate(p, xa, ya, fa, da);
    gather_candidate(p, xb, yb, fb, db);
    resolve_candidate<true>(p, xa, ya, fa, act_a, da);
    if (k0 + 256 < total) resolve_candidate<true>(p, xb, yb, fb, act_b, db);
  }
}

}  // namespace rmx
