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
// kpe.hip the lanes are laid over the batch's frames back to back.  The candidate flags (own rank !=
// 4th largest) ARE a bit plane: one word per lane and row step goes to LDS, and after its walk the
// workgroup hands its candidates out densely and finishes them with kpe.hip's gather / resolve
// (5x5 rank 12 of 25, weight, code bytes, append).
#include "bitslice.h"
#include "common.h"

namespace rmx {

constexpr uint32_t KPE_BS_RA_MAX = 32;  // rows per strip (a row-presence mask fits one word)

constexpr uint64_t kBsRankTbl = bs::pack_tbl(kN2O, false);
constexpr uint64_t kBsNativeOfInvRank = bs::pack_tbl(kO2N, true);

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
  bs::map_planes<kBsRankTbl>(n, R.c);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    R.l[q] = bs::left_of(R.c[q], from_prev_lane(bs::send_right(R.c[q])));
    R.r[q] = bs::right_of(R.c[q], from_next_lane(bs::send_left(R.c[q])));
  }
  bs::row_sums(R);
}

// The streaming pass of one workgroup: medians to HBM, candidate words to s_c[row step][thread].
// f_out / c_out: the lane's frame and 16-pixel chunk (c_out = 0xffffffff: halo / idle lane or past the batch).
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
__device__ __forceinline__ void kpe_bs_body(const KpeParams& p, uint32_t (*s_c)[256], uint32_t& f_out, uint32_t& c_out) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  f_out = 0;
  c_out = 0xffffffffu;
  // XCD-aware block remap (bijective): blocks that share an XCD get neighbouring groups.
  const uint32_t nb = gridDim.x, bid = blockIdx.x;
  const uint32_t bq = nb >> 3, br = nb & 7u, xcd = bid & 7u;
  const uint32_t lb = (xcd < br ? xcd * (bq + 1) : br * (bq + 1) + (xcd - br) * bq) + (bid >> 3);
  const uint32_t g = lb * 4 + wv;
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
    f_out = f;
    c_out = c;
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
    w.a = __builtin_amdgcn_raw_buffer_load_b128(rin, vo, (uint32_t)(ra * W), 0);
    w.b = __builtin_amdgcn_raw_buffer_load_b128(rin, vo, (uint32_t)(rb * W), 0);
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
    __builtin_amdgcn_raw_buffer_store_b128(va, rout, voa, (uint32_t)(oa * W), 0);
    __builtin_amdgcn_raw_buffer_store_b128(vb, rout, vob, (uint32_t)(ob * W), 0);
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
  s_c[jj][threadIdx.x] = bs::differs(nr, B.c, valid);                                                      \
  bs::map_planes<kBsNativeOfInvRank>(nr, nat);                                                             \
  nat[0] &= valid, nat[1] &= valid, nat[2] &= valid, nat[3] &= valid;                                      \
  bs::bytes_from_planes(nat, oa4, ob4);
#endif
#define RMX_BS_STEP(A, B, Cn, PF, R)                                                                       \
  {                                                                                                        \
    const int jj = j + R;                                                                                  \
    bs_make_row(PF, Cn);                                                                                   \
    PF = load_rows(jj + 1 + RMX_BS_DEPTH);                                                                 \
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
}

// LDS: s_c[ra][256] candidate words (dynamic), then the hand-out tables.
#ifndef RMX_BS_WAVES
#define RMX_BS_WAVES 4
#endif
__global__ __launch_bounds__(256, RMX_BS_WAVES) void kpe_bs(KpeParams p) {
  extern __shared__ __attribute__((aligned(16))) uint32_t s_dyn[];
  uint32_t(*s_c)[256] = reinterpret_cast<uint32_t(*)[256]>(s_dyn);
  __shared__ uint32_t s_f[256], s_col[256], s_cum[256];
  __shared__ uint32_t s_tbase[257];
  __shared__ uint32_t s_scan[8];
  const uint32_t tid = threadIdx.x;
  const uint32_t RA = p.rows_per_strip;
  for (uint32_t k = 0; k < RA; ++k) s_c[k][tid] = 0;
  {
    uint32_t f, c;
#if defined(RMX_BS_PRIO)  /* experiment: the streaming phase ahead of other workgroups' finishing phase */
    __builtin_amdgcn_s_setprio(1);
#endif
    kpe_bs_body(p, s_c, f, c);
#if defined(RMX_BS_PRIO)
    __builtin_amdgcn_s_setprio(0);
#endif
    s_f[tid] = f;
    s_col[tid] = c;
  }
  // this thread's candidates, with the running counts after 8, 16 and 24 rows packed beside them
  uint32_t cnt = 0, cum = 0;
  for (uint32_t k = 0; k < RA; ++k) {
    if (k == 8 || k == 16 || k == 24) cum |= cnt << (10 * (k / 8 - 1));
    cnt += (uint32_t)__builtin_popcount(s_c[k][tid]);
  }
  if (RA <= 8) cum |= cnt;
  if (RA <= 16) cum |= cnt << 10;
  if (RA <= 24) cum |= cnt << 20;
  s_cum[tid] = cum;
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
#if defined(RMX_BS_NOFIN)  /* ablation: the streaming pass alone */
  total = 0;
#endif
  s_tbase[tid] = wbase + incl - cnt;
  if (tid == 0) s_tbase[256] = total;
  __syncthreads();
  const uint32_t ya = blockIdx.y * 2 * RA;
  auto locate = [&](uint32_t k, uint32_t& x, uint32_t& y, uint32_t& f) {
    uint32_t lo = 0, hi = 256;  // owner thread: largest t with s_tbase[t] <= k
    while (hi - lo > 1) {
      uint32_t mid = (lo + hi) >> 1;
      if (s_tbase[mid] <= k)
        lo = mid;
      else
        hi = mid;
    }
    uint32_t nth = k - s_tbase[lo];
    const uint32_t cm = s_cum[lo];
    uint32_t kb = 0;  // the block of eight rows, then the row
    if (nth >= ((cm >> 20) & 1023u))
      kb = 24, nth -= (cm >> 20) & 1023u;
    else if (nth >= ((cm >> 10) & 1023u))
      kb = 16, nth -= (cm >> 10) & 1023u;
    else if (nth >= (cm & 1023u))
      kb = 8, nth -= cm & 1023u;
    uint32_t w = s_c[kb][lo];
    for (;;) {
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
    x = s_col[lo] * 16 + (b & 15u);
    y = ya + (b >> 4) * RA + kb;
    f = s_f[lo];
  };
  for (uint32_t k0 = 0; k0 < total; k0 += 512) {
    const uint32_t ka = k0 + tid, kb = k0 + 256 + tid;
    const bool act_a = ka < total, act_b = kb < total;
    if (__builtin_amdgcn_ballot_w64(act_a) == 0) break;  // this wave's lanes are past the last candidate (no barriers below)
    uint32_t xa = 2, ya2 = 2, fa = 0, xb = 2, yb2 = 2, fb = 0;
    if (act_a) locate(ka, xa, ya2, fa);
    if (act_b) locate(kb, xb, yb2, fb);
    CandData da, db;
    gather_candidate(p, xa, ya2, fa, da);
    gather_candidate(p, xb, yb2, fb, db);
    resolve_candidate<true>(p, xa, ya2, fa, act_a, da);
    if (k0 + 256 < total) resolve_candidate<true>(p, xb, yb2, fb, act_b, db);
  }
}

}  // namespace rmx
