// paircmp.h — bit-plane arithmetic of the keypoint extraction's candidate test, by pairwise comparison.
//
// compute_pixel's first half (reference kpe.hpp:308-315) marks a pixel whose own luminance rank p1 differs
// from p3 = median_pixel(H3, 4), the 4th largest rank of its 3x3 window (kpe.hpp:326-340: the first bin, from
// the top, whose running count reaches 4).  bitslice.h SELECTS p3 for every pixel (radix selection: ~120
// operations per 32 pixels) and needs the inverse tables and a planes -> bytes transposition to write the median
// plane.  But median[x, y] = ordered_to_native(p3) equals the input wherever p1 == p3, and
//
//     p1 == p3   <=>   #(neighbours > centre) <= 3   and   #(neighbours >= centre) >= 3
//
// (over the eight neighbours; the centre itself is the fourth pixel that is >= centre).  A comparison of two
// 4-bit ranks on bit planes is a chain of four three-input operations ("the bits differ ? a's bit : the result
// of the lower bits"), one chain for >, one for >=, and every unordered pair of adjacent pixels serves BOTH of
// its windows: [q > p] and [q >= p] are what p's window counts, their complements [p >= q]' = [q > p] ... are
// what q's window counts.  So a row costs four comparisons (east, south, south-east, south-west pairs: 32
// operations), six one-pixel shifts of result planes (12), and two eight-input threshold counters (24): ~70
// operations per 32 pixels instead of ~150 — and no median plane at all: the kernel stores the input rows and
// patches the candidates' bytes, whose 3x3 selection its finish computes anyway (kpe_pc.hip).
//
// Plain integer code shared by the HIP kernel and a host test (tests/cpp/paircmp_check.cpp against the oracle).
//
// Layout of a plane word: bit x = pixel x of the lane's chunk of 32 CONSECUTIVE pixels of one image row
// (bitslice.h's planes_from_bytes with a = the first sixteen bytes, b = the next sixteen).
#pragma once
#include "bitslice.h"

namespace rmx {
namespace pc {

using bs::bop3;

// ---- neighbours -----------------------------------------------------------------------------------------
// bit x of the result = bit x - 1 of p (bit 0 = bit 31 of the previous lane's word) / bit x + 1 (bit 31 = bit 0
// of the next lane's word): v_alignbit_b32 on the device
RMX_HD uint32_t from_left(uint32_t p, uint32_t prev) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_alignbit(p, prev, 31);
#else
  return (p << 1) | (prev >> 31);
#endif
}
RMX_HD uint32_t from_right(uint32_t p, uint32_t next) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_alignbit(next, p, 1);
#else
  return (p >> 1) | (next << 31);
#endif
}

// ---- comparison of two 4-bit values on planes -------------------------------------------------------------
struct f_gt0 { constexpr uint32_t operator()(uint32_t a, uint32_t c, uint32_t) const { return a & ~c; } };
struct f_ge0 { constexpr uint32_t operator()(uint32_t a, uint32_t c, uint32_t) const { return a | ~c; } };
// the bits differ ? a's : what the lower bits said
struct f_cmp { constexpr uint32_t operator()(uint32_t a, uint32_t c, uint32_t g) const { return (a & ~c) | (~(a ^ c) & g); } };

struct Cmp {
  uint32_t gt, ge;  // a > c, a >= c
};
RMX_HD Cmp compare(const uint32_t (&a)[4], const uint32_t (&c)[4]) {
  uint32_t g = bop3<f_gt0>(a[0], c[0], c[0]), e = bop3<f_ge0>(a[0], c[0], c[0]);
  g = bop3<f_cmp>(a[1], c[1], g), e = bop3<f_cmp>(a[1], c[1], e);
  g = bop3<f_cmp>(a[2], c[2], g), e = bop3<f_cmp>(a[2], c[2], e);
  g = bop3<f_cmp>(a[3], c[3], g), e = bop3<f_cmp>(a[3], c[3], e);
  return Cmp{g, e};
}

// ---- threshold counters ------------------------------------------------------------------------------------
// Inputs arrive in two polarities (a pair's bit is direct for one of its windows and complemented for the
// other); the complement is folded into the truth tables: INV bit i set = input i is complemented.
template <int INV>
struct f_xor3i {
  constexpr uint32_t operator()(uint32_t a, uint32_t b, uint32_t c) const {
    return ((INV & 1) ? ~a : a) ^ ((INV & 2) ? ~b : b) ^ ((INV & 4) ? ~c : c);
  }
};
template <int INV>
struct f_maji {
  constexpr uint32_t operator()(uint32_t a, uint32_t b, uint32_t c) const {
    const uint32_t x = (INV & 1) ? ~a : a, y = (INV & 2) ? ~b : b, z = (INV & 4) ? ~c : c;
    return (x & y) | (z & (x ^ y));
  }
};
template <int INV>
RMX_HD bs::Sum2 full_add_i(uint32_t a, uint32_t b, uint32_t c) {
  return bs::Sum2{bop3<f_xor3i<INV>>(a, b, c), bop3<f_maji<INV>>(a, b, c)};
}
struct f_andn { constexpr uint32_t operator()(uint32_t a, uint32_t b, uint32_t) const { return a & ~b; } };
struct f_or_and { constexpr uint32_t operator()(uint32_t a, uint32_t b, uint32_t c) const { return a | (b & c); } };
struct f_flag {  // valid & (more than three are greater | fewer than three are greater or equal)
  constexpr uint32_t operator()(uint32_t valid, uint32_t g4, uint32_t e3) const { return valid & (g4 | ~e3); }
};

// Carries of an eight-input count: d[0..3] direct, i[0..3] complemented.  count = s + 2 (k0 + k1 + k2 + k3).
struct Pop8 {
  uint32_t s, k0, k1, k2, k3;
};
RMX_HD Pop8 pop8(uint32_t d0, uint32_t d1, uint32_t d2, uint32_t d3, uint32_t i0, uint32_t i1, uint32_t i2, uint32_t i3) {
  const bs::Sum2 a = bs::full_add(d0, d1, d2);
  const bs::Sum2 b = full_add_i<7>(i0, i1, i2);
  const uint32_t s3 = ~(d3 ^ i3), k2 = bop3<f_andn>(d3, i3, i3);  // d3 + ~i3
  const bs::Sum2 c = bs::full_add(a.s, b.s, s3);
  return Pop8{c.s, a.c, b.c, k2, c.c};
}
// count >= 4  <=>  at least two of the four carries
RMX_HD uint32_t at_least_4(const Pop8& p) {
  const bs::Sum2 t = bs::full_add(p.k0, p.k1, p.k2);
  return bop3<f_or_and>(t.c, t.s, p.k3);
}
// count >= 3  <=>  two carries, or one and the units bit
RMX_HD uint32_t at_least_3(const Pop8& p) {
  const bs::Sum2 t = bs::full_add(p.k0, p.k1, p.k2);
  return t.c | bop3<bs::f_maj>(t.s, p.k3, p.s);
}

// ---- the window ---------------------------------------------------------------------------------------------
// What a row M contributes to the test of its own pixels once the row below (B) is known.  Rows: A above, M, B
// below; c = rank planes of a row's pixels, r = of their right neighbours (pixel x + 1).
//   east   = compare(rM, cM)        [M(x+1) > M(x)]          window x: E direct, window x + 1: W complemented
//   south  = compare(cB, cM)        [B(x)   > M(x)]          S of M's window direct, N of B's window complemented
//   sedge  = compare(rB, cM)        [B(x+1) > M(x)]          SE of M's window direct, NW of B's window (x + 1) complemented
//   swedge = compare(rM, cB)        [M(x+1) > B(x)]          NE of B's window direct, SW of M's window (x + 1) complemented
// "shifted" = from_left of the plane: bit x holds the pair that starts at pixel x - 1.
struct Upper {  // what the step of row M (as the row below A) leaves for M's own test
  Cmp north;    // compare(cM, cA): complemented for M
  Cmp ne;       // compare(rA, cM): direct
  Cmp nw_sh;    // shifted compare(rM, cA): complemented
  Cmp east;     // compare(rM, cM): direct
  Cmp west_sh;  // shifted east: complemented
};
// candidate flags of row M: pixels inside `valid` whose own rank is not the 4th largest of their 3x3 window
RMX_HD uint32_t not_fourth(const Upper& u, const Cmp& south, const Cmp& se, const Cmp& sw_sh, uint32_t valid) {
  // greater than the centre: S, SE, NE, E direct; N, SW, NW, W are complements of "the centre is >= them"
  const Pop8 g = pop8(south.gt, se.gt, u.ne.gt, u.east.gt, u.north.ge, sw_sh.ge, u.nw_sh.ge, u.west_sh.ge);
  // greater or equal: complements of "the centre is > them"
  const Pop8 e = pop8(south.ge, se.ge, u.ne.ge, u.east.ge, u.north.gt, sw_sh.gt, u.nw_sh.gt, u.west_sh.gt);
  return bop3<f_flag>(valid, at_least_4(g), at_least_3(e));
}

}  // namespace pc
}  // namespace rmx
