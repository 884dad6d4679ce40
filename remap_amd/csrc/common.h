// common.h — shared device/host definitions for the gfx950 kernels of remap_amd.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/remap_amd.h"

namespace rmx {

// ---- luminance-rank LUTs (reference: cpl.hpp:163-217; values verified in SURVEY.md §8a-1) ----
// native_to_ordered = {0,15,2,12,6,9,3,13,5,1,7,4,8,14,10,11}
// ordered_to_native = {0,9,2,6,11,8,4,10,12,5,14,15,3,7,13,1}
// Derived on the host at library load from the palette (api.hip: init_luts) and checked
// against these constants; the kernels use byte tables packed for v_perm_b32:
//   kKey[v] = native_to_ordered[v] << 4 | v   ("key": orders like the rank, carries the
//             native value in its low nibble, so rank selection needs no inverse LUT)
//   kOrd[v] = native_to_ordered[v]
#define RMX_N2O                                                  \
  { 0, 15, 2, 12, 6, 9, 3, 13, 5, 1, 7, 4, 8, 14, 10, 11 }
#define RMX_O2N                                                  \
  { 0, 9, 2, 6, 11, 8, 4, 10, 12, 5, 14, 15, 3, 7, 13, 1 }

constexpr uint8_t kN2O[16] = RMX_N2O;
constexpr uint8_t kO2N[16] = RMX_O2N;

constexpr uint32_t key_of(int v) { return (uint32_t)(kN2O[v] << 4 | v); }
constexpr uint32_t ord_of(int v) { return (uint32_t)kN2O[v]; }
constexpr uint32_t key4(int b) { return key_of(b) | key_of(b + 1) << 8 | key_of(b + 2) << 16 | key_of(b + 3) << 24; }
constexpr uint32_t ord4(int b) { return ord_of(b) | ord_of(b + 1) << 8 | ord_of(b + 2) << 16 | ord_of(b + 3) << 24; }

// v_perm_b32: D.byte[i] = {S0:S1}.byte[sel.byte[i]], selector 0-3 -> S1, 4-7 -> S0.
__device__ __forceinline__ uint32_t perm(uint32_t s0, uint32_t s1, uint32_t sel) {
  return __builtin_amdgcn_perm(s0, s1, sel);
}

// Map four packed pixels (one 0..15 value per byte) through a 16-entry byte table given as
// four dwords (entries 0-3, 4-7, 8-11, 12-15): 7 full-rate VALU ops per 4 pixels.
// Two v_perm lookups (entries 0-7 / 8-15) are merged under a byte mask that replicates
// bit 3 of every pixel; the mask itself is one more v_perm using the sign-replicate
// selectors (8..11) on the pixels shifted so that bit 3 sits in a byte's sign position.
template <uint32_t T0, uint32_t T1, uint32_t T2, uint32_t T3>
__device__ __forceinline__ uint32_t map16(uint32_t w) {
  uint32_t sel = w & 0x07070707u;
  uint32_t a = perm(T1, T0, sel);
  uint32_t b = perm(T3, T2, sel);
  // byte i of m = 0xff iff bit 3 of byte i of w: bytes 1,3 from S1 = w << 4 (selectors 8, 9),
  // bytes 0,2 from S0 = w << 12 (selectors 10, 11)
  uint32_t m = perm(w << 12, w << 4, 0x090b080au);
  return (b & m) | (a & ~m);
}
__device__ __forceinline__ uint32_t map_key(uint32_t w) { return map16<key4(0), key4(4), key4(8), key4(12)>(w); }
__device__ __forceinline__ uint32_t map_ord(uint32_t w) { return map16<ord4(0), ord4(4), ord4(8), ord4(12)>(w); }

// 3-input integer min/max/median, one VALU op each.  Written as asm because hipcc CSEs the
// min/max pairs of a sort3 into six 2-input ops instead of v_min3/v_med3/v_max3.
__device__ __forceinline__ uint32_t umin3(uint32_t a, uint32_t b, uint32_t c) {
  uint32_t d;
  asm("v_min3_u32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ uint32_t umax3(uint32_t a, uint32_t b, uint32_t c) {
  uint32_t d;
  asm("v_max3_u32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ uint32_t umed3(uint32_t a, uint32_t b, uint32_t c) {
  uint32_t d;
  asm("v_med3_u32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}

__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
__device__ __forceinline__ uint32_t mbcnt(uint64_t mask) {
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// Raw keypoint record emitted by the kpe kernel (unordered within a frame).
struct __attribute__((aligned(16))) RawDesc {
  uint32_t code[4];  // 13 code bytes, little-endian, byte 12 = weight | p24 << 4; bytes 13-15 zero
  uint32_t xy;       // x | y << 16
  uint32_t pad[3];
};
static_assert(sizeof(RawDesc) == 32, "RawDesc");

// Region band limits (kpe.hpp:84-90,157-192,235-277 in closed form).
struct Bands {
  uint32_t gw, gh;
  uint16_t x0[4], x1[4], y0[4], y1[4];
};

__host__ __device__ inline uint32_t region_mask(const Bands& b, uint32_t x, uint32_t y) {
  uint32_t m = 0;
  for (uint32_t c = 0; c < b.gw; ++c)
    if (x >= b.x0[c] && x < b.x1[c])
      for (uint32_t r = 0; r < b.gh; ++r)
        if (y >= b.y0[r] && y < b.y1[r]) m |= 1u << (b.gh * c + r);
  return m;
}

// Ascending bitonic sort of up to 64 * KPL keys held by ONE wave in registers: element i lives
// in key[i >> 6] of lane i & 63.  Exchanges at distance < 64 are lane shuffles, larger ones
// stay inside the lane.  n2 = power of two >= number of keys (unused elements = 0xffffffff).
template <uint32_t KPL, uint32_t JS>
__device__ __forceinline__ void wave_sort_inlane(uint32_t (&key)[KPL], uint32_t n2, uint32_t k) {
#pragma unroll
  for (uint32_t s = 0; s < KPL; ++s)
    if ((s & JS) == 0 && (s | JS) < KPL && (s | JS) * 64 < n2) {
      const bool up = ((s * 64) & k) == 0;
      const uint32_t a = key[s], b = key[s | JS];
      const uint32_t lo = min(a, b), hi = max(a, b);
      key[s] = up ? lo : hi;
      key[s | JS] = up ? hi : lo;
    }
}
// Value of lane (lane ^ J): DPP where the pattern exists (quad_perm for 1 and 2, row_ror:8 = xor 8
// inside a row of 16), ds_swizzle for 4 and 16, ds_bpermute for 32.  All lanes must be active.
template <uint32_t J>
__device__ __forceinline__ uint32_t lane_xor(uint32_t v, uint32_t lane) {
  if constexpr (J == 1)
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false);  // quad_perm:[1,0,3,2]
  else if constexpr (J == 2)
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, false);  // quad_perm:[2,3,0,1]
  else if constexpr (J == 4)
    return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x101f);  // bit mode: and 0x1f, xor 4
  else if constexpr (J == 8)
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, false);  // row_ror:8
  else if constexpr (J == 16)
    return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x401f);  // bit mode: and 0x1f, xor 16
  else
    return (uint32_t)__builtin_amdgcn_ds_bpermute((int)((lane ^ 32u) << 2), (int)v);
}

// The network is unrolled at compile time (stage K, distance J): the exchange pattern and the
// min/max choice are constants of the step — 3-4 instructions per key and step against ~9 with
// run-time distances through ds_bpermute.  Stages beyond n2 are skipped (uniform).
template <uint32_t KPL, uint32_t K, uint32_t J>
__device__ __forceinline__ void wave_sort_step(uint32_t (&key)[KPL], uint32_t n2, uint32_t lane) {
  if constexpr (J >= 64) {
    wave_sort_inlane<KPL, J / 64>(key, n2, K);
  } else {
    const bool lower = (lane & J) == 0;
#pragma unroll
    for (uint32_t s = 0; s < KPL; ++s)
      if (s * 64 < n2) {
        const uint32_t other = lane_xor<J>(key[s], lane);
        const bool up = K < 64 ? (lane & K) == 0 : ((s * 64) & K) == 0;
        key[s] = (lower == up) ? min(key[s], other) : max(key[s], other);
      }
  }
  if constexpr (J > 1) wave_sort_step<KPL, K, J / 2>(key, n2, lane);
}
template <uint32_t KPL, uint32_t K>
__device__ __forceinline__ void wave_sort_stage(uint32_t (&key)[KPL], uint32_t n2, uint32_t lane) {
  if (K <= n2) wave_sort_step<KPL, K, K / 2>(key, n2, lane);
  if constexpr (K < 64 * KPL) wave_sort_stage<KPL, 2 * K>(key, n2, lane);
}
template <uint32_t KPL>
__device__ __forceinline__ void wave_sort(uint32_t (&key)[KPL], uint32_t n2, uint32_t lane) {
  static_assert(KPL == 1 || KPL == 2 || KPL == 4 || KPL == 8, "wave_sort: 1, 2, 4 or 8 keys per lane");
  wave_sort_stage<KPL, 2>(key, n2, lane);
}
// one key per lane
__device__ __forceinline__ uint32_t wave_sort1(uint32_t key, uint32_t n2, uint32_t lane) {
  uint32_t k1[1] = {key};
  wave_sort<1>(k1, n2, lane);
  return k1[0];
}

}  // namespace rmx
