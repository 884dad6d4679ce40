// kpe_pc.hip — keypoint extraction with the candidate test by pairwise comparison (gfx950), frames whose width
// is a multiple of 32 and whose rows fill a wave (chunks per row divide 64 with at most eight lanes idle).
//
// Replaces the same reference code as kpe.hip / kpe_bs.hip (kpe::extractor::extract, kpe.hpp:92-108, with
// sum_row / col_* / compute_pixel / median_pixel / encode_keypoint, kpe.hpp:111-379) and produces the same bytes.
// What differs from kpe_bs_ring (whose structure it keeps: a wave walks down a strip, the last seven rows in an
// LDS ring, candidates finished by the wave itself every three rows):
//
//   * The streaming pass no longer SELECTS the 4th largest of every 3x3 window.  It only decides which pixels are
//     candidates (own rank != 4th largest), by comparing adjacent pixels' ranks on bit planes (paircmp.h: ~70
//     operations per 32 pixels instead of ~150), and writes no median plane of its own.
//   * The median plane is the INPUT wherever a pixel is no candidate (ordered_to_native(native_to_ordered(v)) ==
//     v); a candidate's median — the 4th largest of its 3x3, which the finish computes for its test anyway — is
//     XOR-ed into the LDS ring as a nibble patch once no later window needs the raw value, and the rows leave
//     for HBM from the ring (one 16-byte LDS read, twelve unpack operations, two 16-byte stores per lane and
//     row) three to five rows behind the stream: no rank -> native tables, no planes -> bytes transposition.
//   * A lane owns 32 CONSECUTIVE pixels of ONE row, so horizontal neighbours are a v_alignbit_b32 with the
//     neighbouring lane's word (two operations per plane instead of three).  The ring doubles as the transposer
//     between that ownership and the memory accesses: a 16-byte load or store instruction covers the wave's row
//     contiguously (lane i: sixteen pixels of chunk i / 2, then of chunk 32 + i / 2), rows enter the ring as
//     8-byte pieces and the owner reads its 16 packed bytes back (one ds_read_b128 per row step).
//   * The hand-out of a roll's candidates needs no per-bit loops: owners mark where their candidates start, a
//     max-scan spreads the owner over the places, every lane picks "the k-th set bit of its owner's words".
//   * Keypoints wait as raw windows in an LDS stash and get weight, code bytes and table row 64 at a time
//     (the encode ran for ~5 of 64 lanes per roll).
//
// Patches and rows: the roll of strip rows k0 .. k0+2 runs when row k0+4 is in the ring, which then holds rows
// k0-2 .. k0+4.  Windows of the NEXT roll still read rows k0+1 and k0+2 raw, so a roll applies its patches of row
// k0 and the previous roll's patches of rows k0-2 and k0-1 (held in registers in between: up to 256
// candidates per roll; beyond that — noise — they go through a small global buffer), then stores rows k0-2 .. k0.
#include "paircmp.h"
#include "common.h"

namespace rmx {

constexpr uint32_t PC_RING = 7;    // ring rows: a roll reads rows k0 - 2 .. k0 + 4
constexpr uint32_t PC_STASH = 64;  // keypoints a wave holds before it encodes and appends them (>= 64: a pass may find that many)
// Passes (of 64 candidates) of a roll whose patches wait in registers: 4 for sparse streams (an average roll has 60
// candidates; two registers per pass), 8 for dense ones (460 per roll at 1 100 keypoints per frame; what does not
// fit goes through global atomics: 1.73 ms per 10 000 dense frames with 4, 0.85 with 8 — and 2 % more on the sparse
// stream, whose rounds pay a select per pass).  api.hip picks by the density of the batch before.
constexpr uint32_t PC_OVF_DWORDS = 2 * 3 * 64 * 4;  // per wave: [generation][row of the roll][lane][dword of the ring row]
#ifndef RMX_PC_DEPTH
#define RMX_PC_DEPTH 2  // rows in flight
#endif
#ifndef RMX_PC_ST_AUX
#define RMX_PC_ST_AUX 2  // median stores non-temporal, as in kpe_bs
#endif
#ifndef RMX_PC_WGS
#define RMX_PC_WGS 4
#endif

constexpr uint32_t pc_o2n_pack(int base) {  // ordered_to_native[base .. base + 7], a nibble each
  uint32_t r = 0;
  for (int i = 0; i < 8; ++i) r |= (uint32_t)kO2N[base + i] << (4 * i);
  return r;
}
static_assert(pc_o2n_pack(0) == 0xa48b6290u && pc_o2n_pack(8) == 0x1d73fe5cu, "ordered_to_native");

struct alignas(16) PcWaveLds {
  uint32_t ring[PC_RING][64][4];  // [slot][lane]: 32 pixels, two per byte (dword k: pixels 8k + j in the low, 8k + 4 + j in the high nibble of byte j)
  uint32_t stash[PC_STASH][8];    // w[0..4] (pixels x-2 .. x+1 of the window's rows as bytes), e nibbles | frame << 20, x | y << 16, -
  uint32_t own[64];
  uint32_t fcnt[64], fbase[64];
};
static_assert(sizeof(PcWaveLds) * 4 <= 160 * 1024 / 4, "four workgroups per CU");

struct PcRow {
  uint32_t c[4], r[4];  // rank planes of the row's pixels and of their right neighbours
};
struct PcRaw {
  u32x4 a, b;  // what the lane loads of a row: half i & 1 of chunk i / 2 and of chunk 32 + i / 2 (sixteen pixels each)
};

__device__ __forceinline__ uint32_t pc_from_prev_lane(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, true);  // wave_shr:1
}
__device__ __forceinline__ uint32_t pc_from_next_lane(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, true);  // wave_shl:1
}
__device__ __forceinline__ uint32_t pc_shift(uint32_t p) { return pc::from_left(p, pc_from_prev_lane(p)); }
__device__ __forceinline__ pc::Cmp pc_shift(const pc::Cmp& c) { return pc::Cmp{pc_shift(c.gt), pc_shift(c.ge)}; }

// the lane's chunk of a row as the ring holds it (32 pixels, two per byte) -> rank planes
__device__ __forceinline__ void pc_make_row(const u32x4& w, PcRow& R) {
#if defined(RMX_PC_COPY)
  for (int q = 0; q < 4; ++q) R.c[q] = w[q], R.r[q] = w[q] >> 1;
  return;
#endif
  const uint32_t c[4] = {w[0], w[1], w[2], w[3]};
  uint32_t n[4];
  bs::planes_from_packed(c, n);
  bs::rank_planes(n, R.c);
#pragma unroll
  for (int q = 0; q < 4; ++q) R.r[q] = pc::from_right(R.c[q], pc_from_next_lane(R.c[q]));
}

// inclusive scans over the wave (DPP: four steps inside a row of sixteen, two row broadcasts)
template <bool MAX>
__device__ __forceinline__ uint32_t pc_scan(uint32_t v) {
#define RMX_PC_SCAN_STEP(CTRL, ROWS)                                                                       \
  {                                                                                                        \
    const uint32_t t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROWS, 0xf, true);            \
    v = MAX ? max(v, t) : v + t;                                                                           \
  }
  RMX_PC_SCAN_STEP(0x111, 0xf)  // row_shr:1
  RMX_PC_SCAN_STEP(0x112, 0xf)  // row_shr:2
  RMX_PC_SCAN_STEP(0x114, 0xf)  // row_shr:4
  RMX_PC_SCAN_STEP(0x118, 0xf)  // row_shr:8
  RMX_PC_SCAN_STEP(0x142, 0xa)  // row_bcast:15 into rows 1 and 3
  RMX_PC_SCAN_STEP(0x143, 0xc)  // row_bcast:31 into rows 2 and 3
#undef RMX_PC_SCAN_STEP
  return v;
}

template <uint32_t NP>
struct PcRoll {
  static constexpr uint32_t PC_REG_PASSES = NP;
  PcWaveLds* L;
  uint32_t* ovf;  // this wave's PC_OVF_DWORDS of global memory (all zero between launches)
  uint32_t lane, ya, RA, f0, Wc, magic;  // magic: lane / Wc == lane * magic >> 16
  uint32_t st_a, st_b;                   // lane offsets of the two median stores of a row (out of range for idle chunks)
  uint32_t ma0, ma3, mb0, mb3;           // nibble masks of their first / last four pixels (x < 2, x > W - 3 are not processed)
  uint32_t nst;                          // keypoints in the stash (uniform)
  uint32_t cur[PC_REG_PASSES], prev[PC_REG_PASSES];  // patches: owner | row << 6 | bit << 8 | delta << 13 | 1 << 31
  uint32_t nprev;                        // (uniform) how many of prev[] the roll before filled
  uint32_t ovf_cur, ovf_prev, gen;       // (uniform) patches of that roll in the global buffer; its half
};

// weight, code bytes and table rows for the stashed keypoints
template <uint32_t NP>
__device__ __forceinline__ void pc_flush(const KpeParams& p, PcRoll<NP>& R) {
  PcWaveLds& L = *R.L;
  const uint32_t lane = R.lane, n = R.nst;
  if (n == 0) return;  // uniform
  L.fcnt[lane] = 0;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const bool act = lane < n;
  const uint32_t* st = L.stash[act ? lane : 0];
  BsWin W;
#pragma unroll
  for (int r = 0; r < 5; ++r) W.w[r] = st[r], W.e[r] = (st[5] >> (4 * r)) & 15u;
  const uint32_t fr = st[5] >> 20, xy = st[6];
  uint32_t v[7];
  bs_ranks(W, v);
  uint4 c4;
  bs_record(W, v, c4);
  const uint32_t rank = act ? atomicAdd(&L.fcnt[fr], 1u) : 0u;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  {
    const uint32_t c = L.fcnt[lane];
    L.fbase[lane] = c ? atomicAdd(&p.kcount[R.f0 + lane], c) : 0u;  // (lanes past the wave's frames have no records)
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  if (act) {
    const uint32_t slot = L.fbase[fr] + rank;
    if (slot < p.kmax) {
      RawDesc* dd = p.raw + (size_t)(R.f0 + fr) * p.kmax + slot;
      reinterpret_cast<uint4*>(dd)[0] = c4;
      reinterpret_cast<uint4*>(dd)[1] = make_uint4(xy, 0u, 0u, 0u);
    } else {
      atomicOr(&p.status[0], 1u);
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  R.nst = 0;
}

__device__ __forceinline__ uint32_t pc_slot(int q) { return (uint32_t)(q + 1 + (int)PC_RING) % PC_RING; }  // ring slot of strip row q (>= -2)

// XOR a patch into the ring row in `slot`
__device__ __forceinline__ void pc_patch(PcWaveLds& L, uint32_t e, uint32_t slot, bool on) {
  const uint32_t owner = e & 63u, b = (e >> 8) & 31u, delta = (e >> 13) & 15u;
  const uint32_t sh = 8u * (b & 3u) + (b & 4u);
  if (on) __hip_atomic_fetch_xor(&L.ring[slot][owner][b >> 3], delta << sh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}

// one ring row (final: every patch applied) -> the median plane
template <uint32_t NP>
__device__ __forceinline__ void pc_store_row(const KpeParams& p, const PcRoll<NP>& R, const rsrc_t& rout, uint32_t slot, int y) {
  const uint32_t* row = &R.L->ring[slot][0][0];
  u32x4 va = {0, 0, 0, 0}, vb = {0, 0, 0, 0};
  if (y >= 2 && y + 5 <= (int)p.H) {  // (uniform) processed rows: [2, H-5] (kpe.hpp:268); the others are zero
    const uint2 ca = *reinterpret_cast<const uint2*>(row + 2 * R.lane), cb = *reinterpret_cast<const uint2*>(row + 128 + 2 * R.lane);
    const uint32_t m = 0x0f0f0f0fu;
    va = u32x4{ca.x & R.ma0, (ca.x >> 4) & m, ca.y & m, (ca.y >> 4) & R.ma3};
    vb = u32x4{cb.x & R.mb0, (cb.x >> 4) & m, cb.y & m, (cb.y >> 4) & R.mb3};
  }
  const uint32_t voa = y < (int)p.H ? R.st_a : 0x80000000u, vob = y < (int)p.H ? R.st_b : 0x80000000u;
  // (as in kpe_bs: nothing may be scheduled between a 16-byte store and the wait states behind it)
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_raw_buffer_store_b128(va, rout, voa, (uint32_t)(y * (int)p.W), RMX_PC_ST_AUX);
  __builtin_amdgcn_raw_buffer_store_b128(vb, rout, vob, (uint32_t)(y * (int)p.W), RMX_PC_ST_AUX);
  asm volatile("s_nop 1");
  __builtin_amdgcn_sched_barrier(0);
}

// The candidates of strip rows k0, k0 + 1, k0 + 2 (words w0..w2 of every lane: bit x = pixel x of the lane's chunk);
// ring rows k0 - 2 .. k0 + 4 are in LDS.  Afterwards rows k0 - 2 .. k0 are final and stored.
template <uint32_t NP>
__device__ __forceinline__ void pc_roll(const KpeParams& p, PcRoll<NP>& R, const rsrc_t& rout, uint32_t w0, uint32_t w1, uint32_t w2, int k0) {
  constexpr uint32_t PC_REG_PASSES = NP;
  PcWaveLds& L = *R.L;
  const uint32_t lane = R.lane;
  const uint32_t c = (uint32_t)(__builtin_popcount(w0) + __builtin_popcount(w1) + __builtin_popcount(w2));
  const uint32_t incl = pc_scan<false>(c), excl = incl - c;
  const uint32_t T = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
#if defined(RMX_PC_STATS)  /* diagnostic build (tools/ab_pc.py): rolls, candidates, rounds, rolls past the registers */
  if (lane == 0) {
    atomicAdd(&p.status[40], 1u);
    atomicAdd(&p.status[41], T);
    atomicAdd(&p.status[42], (T + 63) / 64);
    if (T > 64 * PC_REG_PASSES) atomicAdd(&p.status[43], 1u);
  }
#endif
  // ring rows k0 - 2 + m, m = 0 .. 6, as dword offsets into the ring (uniform): a candidate of row k0 + j reads rows j .. j + 4
  uint32_t S[7];
#pragma unroll
  for (int m = 0; m < 7; ++m) S[m] = pc_slot(k0 - 2 + m) * 256u;
  const bool edge = k0 == 0 || k0 + 3 >= (int)R.RA;  // (uniform) rolls whose windows reach past the rows the strip streams
  R.ovf_cur = 0;
  for (uint32_t r0 = 0; r0 < T; r0 += 64) {  // uniform
    // ---- hand-out: place r0 + lane takes the k-th candidate of its owner
    L.own[lane] = 0;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (c && excl < r0 + 64 && incl > r0) L.own[max(excl, r0) - r0] = lane;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const uint32_t owner = pc_scan<true>(L.own[lane]);
    const int oaddr = (int)(owner << 2);
    const uint32_t eo = (uint32_t)__builtin_amdgcn_ds_bpermute(oaddr, (int)excl);
    const uint32_t a0 = (uint32_t)__builtin_amdgcn_ds_bpermute(oaddr, (int)w0);
    const uint32_t a1 = (uint32_t)__builtin_amdgcn_ds_bpermute(oaddr, (int)w1);
    const uint32_t a2 = (uint32_t)__builtin_amdgcn_ds_bpermute(oaddr, (int)w2);
    const bool act = r0 + lane < T;
    uint32_t k = r0 + lane - eo;
    const uint32_t c0 = (uint32_t)__builtin_popcount(a0), c01 = c0 + (uint32_t)__builtin_popcount(a1);
    const bool j1 = k >= c0, j2 = k >= c01;  // the candidate's row of the roll: j = j1 + j2
    uint32_t w = j2 ? a2 : (j1 ? a1 : a0);
    k -= j2 ? c01 : (j1 ? c0 : 0u);
    if (!act) w = 4u, k = 0;  // (a harmless place: pixel 2 of the owner's chunk)
    for (uint32_t t = 0; __builtin_amdgcn_ballot_w64(k > t); ++t)  // uniform; k is a handful at most
      if (k > t) w &= w - 1u;
    const uint32_t b = (uint32_t)__builtin_ctz(w);
    // ---- the 5x5 window from the ring: per window row the two dwords (sixteen pixels, two per byte) around pixel x - 2
    const int A = (int)(owner * 8) + (((int)b - 2) >> 2);  // byte-dword (four pixels) along the wave's row that holds pixel x - 2
    const uint32_t Au = (uint32_t)max(A, 0), D = Au >> 1;
    const bool odd = (Au & 1u) != 0;
    const uint32_t sh_lo = odd ? 4u : 0u, sh_hi = 4u - sh_lo;
    const uint32_t* ring = &L.ring[0][0][0];
    uint32_t lo[5], hi[5];
#pragma unroll
    for (int r = 0; r < 5; ++r) {
      const uint32_t so = j2 ? S[r + 2] : (j1 ? S[r + 1] : S[r]);
      const uint32_t d0 = ring[so + D], d1 = ring[so + D + 1];  // (D + 1 == 256 only where its half is not used)
      lo[r] = d0 >> sh_lo;
      hi[r] = (odd ? d1 : d0) >> sh_hi;
    }
    const uint32_t j = (j1 ? 1u : 0u) + (j2 ? 1u : 0u);
    if (edge) {  // the rows just outside the strip (-2, RA + 1) are not streamed: from global memory
      const int rr = k0 + (int)j;  // strip row of the candidate
      const bool outside = act && (rr < 1 || rr + 2 > (int)R.RA);
      if (__builtin_amdgcn_ballot_w64(outside)) {
        const uint32_t fr = (owner * R.magic) >> 16, col = owner - fr * R.Wc;
        const uint8_t* base = p.frames + (size_t)(R.f0 + fr) * p.W * p.H + ((col * 32 + b - 2) & ~3u);
        const int r_out = rr < 2 ? 0 : 4;  // (row -2 is window row 0 of strip row 0, row RA + 1 window row 4 of strip row RA - 1)
        if (outside) {
          const uint2 v = *reinterpret_cast<const uint2*>(base + (size_t)((int)R.ya + rr - 2 + r_out) * p.W);
          if (r_out == 0) lo[0] = v.x, hi[0] = v.y;
          else lo[4] = v.x, hi[4] = v.y;
        }
      }
    }
    BsWin Wn;  // natives: pixels x-2 .. x+1 of each row as bytes, and pixel x+2
    {
      const uint32_t sa = (b - 2u) & 3u;
#pragma unroll
      for (int r = 0; r < 5; ++r) {
        Wn.w[r] = __builtin_amdgcn_alignbyte(hi[r], lo[r], sa) & 0x0f0f0f0fu;
        Wn.e[r] = (hi[r] >> (8u * sa)) & 15u;
      }
    }
    uint32_t vr[7];
    bs_ranks(Wn, vr);
    const uint32_t p3 = bs_p3(vr);
    // the pixel's median: the native colour of the 4th largest of its 3x3 (kpe.hpp:313-315), as a patch of its nibble
    const uint32_t nat3 = ((p3 & 8u ? pc_o2n_pack(8) : pc_o2n_pack(0)) >> (4u * (p3 & 7u))) & 15u;  // ordered_to_native, cpl.hpp:163-217
    const uint32_t delta = nat3 ^ ((Wn.w[2] >> 16) & 15u);
    const uint32_t res = act ? (owner | j << 6 | b << 8 | delta << 13 | 0x80000000u) : 0u;
    if (r0 < 64 * PC_REG_PASSES) {  // uniform
#pragma unroll
      for (uint32_t q = 0; q < PC_REG_PASSES; ++q) R.cur[q] = r0 == 64 * q ? res : R.cur[q];  // (selects, not a tree of scalar branches)
    } else {
      if (act) atomicXor(&R.ovf[((R.gen * 3 + j) * 64 + owner) * 4 + (b >> 3)], delta << (8u * (b & 3u) + (b & 4u)));
      R.ovf_cur = 1;
    }
    // ---- keypoints: the window waits in the stash
    const bool kp = act && !bs_rank12_is(vr, p3);
    const uint64_t km = __builtin_amdgcn_ballot_w64(kp);
    if (km) {
      if (R.nst + (uint32_t)__builtin_popcountll(km) > PC_STASH) pc_flush(p, R);  // (uniform; the new ones are still in registers)
      if (kp) {
        const uint32_t fr = (owner * R.magic) >> 16, col = owner - fr * R.Wc;
        const uint32_t x = col * 32 + b, y = R.ya + (uint32_t)k0 + j;
        uint32_t* st = L.stash[R.nst + mbcnt(km)];
        *reinterpret_cast<uint4*>(st) = make_uint4(Wn.w[0], Wn.w[1], Wn.w[2], Wn.w[3]);
        *reinterpret_cast<uint4*>(st + 4) =
            make_uint4(Wn.w[4], Wn.e[0] | Wn.e[1] << 4 | Wn.e[2] << 8 | Wn.e[3] << 12 | Wn.e[4] << 16 | fr << 20, x | y << 16, 0u);
      }
      R.nst += (uint32_t)__builtin_popcountll(km);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  // ---- rows k0 - 2 .. k0 are final once these patches are in: row k0's from this roll, the two above from the one before
  const uint32_t s0 = pc_slot(k0 - 2), s1 = pc_slot(k0 - 1), s2 = pc_slot(k0);
  const uint32_t ncur = min((T + 63u) >> 6, PC_REG_PASSES);  // (uniform) register passes of this roll
#pragma unroll
  for (uint32_t q = 0; q < PC_REG_PASSES; ++q) {
    if (q < R.nprev) {
      const uint32_t ep = R.prev[q], jp = (ep >> 6) & 3u;
      pc_patch(L, ep, jp == 1 ? s0 : s1, (ep >> 31) && jp != 0);
    }
    if (q < ncur) {
      const uint32_t ec = R.cur[q];
      pc_patch(L, ec, s2, (ec >> 31) && ((ec >> 6) & 3u) == 0);
    }
  }
  if (R.ovf_prev | R.ovf_cur) {  // uniform: the patches that did not fit the registers, per owner
    // (this wave's own atomics, performed at its XCD's L2: they only have to have left the wave — an agent-scope
    // fence would write the whole L2 back)
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    const uint32_t slots[3] = {s2, s0, s1};
#pragma unroll
    for (uint32_t jj = 0; jj < 3; ++jj) {
      const bool on = jj == 0 ? R.ovf_cur != 0 : R.ovf_prev != 0;
      const uint32_t g = jj == 0 ? R.gen : R.gen ^ 1u;
      if (on) {
        uint32_t* src = &R.ovf[((g * 3 + jj) * 64 + lane) * 4];
        uint32_t* dst = &L.ring[slots[jj]][lane][0];
#pragma unroll
        for (int k = 0; k < 4; ++k) dst[k] ^= atomicExch(&src[k], 0u);
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const int row = k0 - 2 + q;
    if (row >= 0 && row < (int)R.RA) pc_store_row(p, R, rout, q == 0 ? s0 : (q == 1 ? s1 : s2), (int)R.ya + row);  // uniform
  }
#pragma unroll
  for (uint32_t q = 0; q < PC_REG_PASSES; ++q) R.prev[q] = R.cur[q];
  R.nprev = ncur;
  R.ovf_prev = R.ovf_cur;
  R.gen ^= 1u;
}

template <uint32_t NP>
__global__ __launch_bounds__(256, RMX_PC_WGS) void kpe_pc(KpeParams p) {
  constexpr uint32_t PC_REG_PASSES = NP;
  __shared__ PcWaveLds s_wave[4];
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) p.status[15] = 0;  // (keypoints of sampled frames: kpe_table)
#if defined(RMX_PC_PRIO)  /* experiment: issue ahead of the other streams' small kernels */
  __builtin_amdgcn_s_setprio(RMX_PC_PRIO);
#endif
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint32_t g = bs_logical_block() * 4 + wv;
  if (g >= p.groups) return;  // whole wave

  // Compute lanes: HB complete image rows per wave (same y, consecutive frames), Wc chunks of 32 pixels each, lane =
  // chunk.  What a row's first and last chunk receive from the neighbouring lanes belongs to pixels the reference
  // never processes (x < 2, x > W - 3): no halo lanes.
  const uint32_t Wc = p.Wd;
  const size_t frame_bytes = (size_t)p.W * p.H;
  const uint32_t f0 = g * p.HB;
  const uint32_t r = lane / Wc, c = lane - r * Wc;
  const bool lane_ok = r < p.HB && f0 + r < p.n;
  const size_t left = (size_t)(p.n - f0) * frame_bytes;
  const uint32_t recs = (uint32_t)(left < 0x7fffffffull ? left : 0x7fffffffull);
  const rsrc_t rin = make_rsrc(p.frames + (size_t)f0 * frame_bytes, recs);
  const rsrc_t rout = make_rsrc(p.median + (size_t)f0 * frame_bytes, recs);
  uint32_t maskx = 0xffffffffu;  // processed x: [2, W-3] (kpe.hpp:183-184)
  if (c == 0) maskx &= 0xfffffffcu;
  if (c == Wc - 1) maskx &= 0x3fffffffu;
  if (!lane_ok) maskx = 0;
  // Memory lanes: the first 16-byte access of a row covers chunks 0 .. 31 (lane i: half i & 1 of chunk i / 2), the
  // second chunks 32 .. 63 — every instruction touches whole 32-byte sectors.
  PcRoll<NP> roll;
  uint32_t ld_a, ld_b;
  {
    const uint32_t half = lane & 1u;
    const uint32_t ch_a = lane >> 1, ch_b = 32u + (lane >> 1);
    const uint32_t ra = ch_a / Wc, ca = ch_a - ra * Wc, rb = ch_b / Wc, cb = ch_b - rb * Wc;
    const bool ok_a = ra < p.HB && f0 + ra < p.n, ok_b = rb < p.HB && f0 + rb < p.n;
    ld_a = ok_a ? (uint32_t)((size_t)ra * frame_bytes) + ca * 32 + half * 16 : 0u;  // < 64 frames; idle chunks read frame f0
    ld_b = ok_b ? (uint32_t)((size_t)rb * frame_bytes) + cb * 32 + half * 16 : 0u;
    roll.st_a = ok_a ? ld_a : 0x80000000u;  // dropped by the range check
    roll.st_b = ok_b ? ld_b : 0x80000000u;
    roll.ma0 = ca == 0 && half == 0 ? 0x0f0f0000u : 0x0f0f0f0fu;
    roll.ma3 = ca == Wc - 1 && half == 1 ? 0x00000f0fu : 0x0f0f0f0fu;
    roll.mb0 = cb == 0 && half == 0 ? 0x0f0f0000u : 0x0f0f0f0fu;
    roll.mb3 = cb == Wc - 1 && half == 1 ? 0x00000f0fu : 0x0f0f0f0fu;
  }

  // Strips: long ones first, short ones last (workgroups are dispatched in index order: the short strips fill the
  // chip while the last long ones finish — a wave alone on a SIMD streams no faster than four do).
  const bool long_strip = blockIdx.y < p.strips_long;
  const int H = (int)p.H, W = (int)p.W, RA = (int)(long_strip ? p.rows_per_strip : p.rows_short);  // RA: a multiple of 6
  const int ya = long_strip ? (int)blockIdx.y * RA : (int)(p.strips_long * p.rows_per_strip + (blockIdx.y - p.strips_long) * p.rows_short);
  // row ya + j of the strip (clamped to the frame: window rows of unprocessed pixels); j > RA is not wanted
  auto load_rows = [&](int j) -> PcRaw {
    PcRaw w;
    const int ry = min(max(ya + j, 0), H - 1);
    w.a = __builtin_amdgcn_raw_buffer_load_b128(rin, j <= RA ? ld_a : 0x80000000u, (uint32_t)(ry * W), 0);  // (or dropped by the range check)
    w.b = __builtin_amdgcn_raw_buffer_load_b128(rin, j <= RA ? ld_b : 0x80000000u, (uint32_t)(ry * W), 0);
    return w;
  };
  roll.L = s_wave + wv;
  roll.ovf = p.cand + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + wv) * PC_OVF_DWORDS;
  roll.lane = lane, roll.ya = (uint32_t)ya, roll.RA = (uint32_t)RA, roll.f0 = f0, roll.Wc = Wc;
  roll.magic = (65536u + Wc - 1u) / Wc;
  roll.nst = 0;
#pragma unroll
  for (uint32_t q = 0; q < PC_REG_PASSES; ++q) roll.cur[q] = roll.prev[q] = 0;
  roll.ovf_cur = roll.ovf_prev = roll.gen = roll.nprev = 0;
  // a loaded row (strip row q, -1 .. ) into the ring, two pixels per byte: the row's 2 KB in natural order
  auto ring_put = [&](int q, const PcRaw& w) {
    uint32_t* row = &roll.L->ring[pc_slot(q)][0][0];
    *reinterpret_cast<uint2*>(row + 2 * lane) = make_uint2(w.a[0] | w.a[1] << 4, w.a[2] | w.a[3] << 4);
    *reinterpret_cast<uint2*>(row + 128 + 2 * lane) = make_uint2(w.b[0] | w.b[1] << 4, w.b[2] | w.b[3] << 4);
  };
  auto ring_get = [&](int q) -> u32x4 {  // the lane's own chunk of strip row q
    return *reinterpret_cast<const u32x4*>(&roll.L->ring[pc_slot(q)][lane][0]);
  };
  auto ring_sync = [&]() {  // (LDS operations of a wave execute in order; this only keeps the compiler from moving them)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };

  PcRow T0, T1;
  pc::Upper U;
  u32x4 nb;  // the lane's chunk of the row that becomes B in the next step
  PcRaw pf0, pf1;
#if RMX_PC_DEPTH == 3
  PcRaw pf2;
#endif
  {  // rows -1, 0 and 1 into the ring; what row 0's own test needs from above
    const PcRaw wm = load_rows(-1), w0 = load_rows(0), w1 = load_rows(1);
    pf0 = load_rows(2);
    pf1 = load_rows(3);
#if RMX_PC_DEPTH == 3
    pf2 = load_rows(4);
#endif
    ring_put(-1, wm);
    ring_put(0, w0);
    ring_put(1, w1);
    ring_sync();
    PcRow TA;
    pc_make_row(ring_get(-1), TA);
    pc_make_row(ring_get(0), T0);
    nb = ring_get(1);
    const pc::Cmp east = pc::compare(T0.r, T0.c);
    U = pc::Upper{pc::compare(T0.c, TA.c), pc::compare(TA.r, T0.c), pc_shift(pc::compare(T0.r, TA.c)), east, pc_shift(east)};
  }
  uint32_t cw[3] = {0, 0, 0};
#if defined(RMX_PC_NOROLL)
  uint32_t sink = 0;
#endif
  // One row step: the flags of row jj (M) once row jj + 1 (B, from the chunk read back in the step before) is in
  // planes.  Before that: every third step the roll of the three rows before jj (ring rows up to jj + 1 are in);
  // then row jj + 2, loaded two steps ago, replaces row jj - 5 in the ring and its prefetch slot is refilled.
#if defined(RMX_PC_NOROLL)  /* ablation (tools/ab_pc.sh): the streaming pass, the ring and the row stores, no candidates */
#define RMX_PC_ROLLNOW(R)                                                                                 \
  if constexpr ((R) % 3 == 0) {                                                                           \
    _Pragma("unroll") for (int q = 0; q < 3; ++q) {                                                       \
      const int row = jj - 5 + q;                                                                         \
      if (row >= 0) pc_store_row(p, roll, rout, pc_slot(row), ya + row);                                  \
    }                                                                                                     \
    sink ^= cw[0] ^ cw[1] ^ cw[2];                                                                        \
  }
#else
#define RMX_PC_ROLLNOW(R) \
  if constexpr ((R) % 3 == 0) pc_roll(p, roll, rout, cw[0], cw[1], cw[2], jj - 3); /* the three rows before this one */
#endif
#if defined(RMX_PC_COPY)  /* ablation: loads, ring and stores only */
#define RMX_PC_FLAGS(M, B, R) cw[(R) % 3] = B.c[0] ^ B.r[3];
#else
#define RMX_PC_FLAGS(M, B, R)                                                                              \
  const pc::Cmp south = pc::compare(B.c, M.c), se = pc::compare(B.r, M.c), sw = pc::compare(M.r, B.c);     \
  const pc::Cmp east = pc::compare(B.r, B.c);                                                              \
  const int yy = ya + jj;                                                                                  \
  const uint32_t valid = (yy >= 2 && yy + 5 <= H) ? maskx : 0u; /* processed rows: [2, H-5] (kpe.hpp:268) */ \
  cw[(R) % 3] = pc::not_fourth(U, south, se, pc_shift(sw), valid);                                         \
  U = pc::Upper{south, sw, pc_shift(se), east, pc_shift(east)};
#endif
#define RMX_PC_STEP(M, B, PF, R)                                                                           \
  {                                                                                                        \
    const int jj = j + R;                                                                                  \
    if (jj >= 3) { RMX_PC_ROLLNOW(R) }                                                                     \
    ring_sync();                                                                                           \
    ring_put(jj + 2, PF);                                                                                  \
    PF = load_rows(jj + 2 + RMX_PC_DEPTH);                                                                 \
    ring_sync();                                                                                           \
    const u32x4 nbn = ring_get(jj + 2);                                                                    \
    pc_make_row(nb, B);                                                                                    \
    RMX_PC_FLAGS(M, B, R)                                                                                  \
    nb = nbn;                                                                                              \
  }
  static_assert(RMX_PC_DEPTH == 2 || RMX_PC_DEPTH == 3, "two or three rows in flight");
  for (int j = 0; j < RA; j += 6) {
#if RMX_PC_DEPTH == 3
    RMX_PC_STEP(T0, T1, pf0, 0)
    RMX_PC_STEP(T1, T0, pf1, 1)
    RMX_PC_STEP(T0, T1, pf2, 2)
    RMX_PC_STEP(T1, T0, pf0, 3)
    RMX_PC_STEP(T0, T1, pf1, 4)
    RMX_PC_STEP(T1, T0, pf2, 5)
#else
    RMX_PC_STEP(T0, T1, pf0, 0)
    RMX_PC_STEP(T1, T0, pf1, 1)
    RMX_PC_STEP(T0, T1, pf0, 2)
    RMX_PC_STEP(T1, T0, pf1, 3)
    RMX_PC_STEP(T0, T1, pf0, 4)
    RMX_PC_STEP(T1, T0, pf1, 5)
#endif
  }
#undef RMX_PC_STEP
#undef RMX_PC_FLAGS
#undef RMX_PC_ROLLNOW
#if defined(RMX_PC_NOROLL)
  if (sink == 0x9e3779b9u) p.status[47] = sink;  // (keeps the flags alive)
#else
  pc_roll(p, roll, rout, cw[0], cw[1], cw[2], RA - 3);  // the strip's last three rows (RA is a multiple of six)
  pc_roll(p, roll, rout, 0u, 0u, 0u, RA);               // ... whose last two wait for their patches: rows RA - 2, RA - 1
  pc_flush(p, roll);
#endif
}

}  // namespace rmx
