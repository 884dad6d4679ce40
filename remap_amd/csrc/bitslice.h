// bitslice.h — bit-plane ("bit-sliced") arithmetic of the keypoint extraction's 3x3 pass.
//
// compute_pixel's first half (reference kpe.hpp:308-315) needs, for every pixel, the 4th largest
// luminance rank of its 3x3 window (median_pixel(H3, 4), kpe.hpp:326-340).  Ranks are 4 bits, so the
// selection is done on bit planes: a 32-bit word holds ONE bit of 32 pixels and every and/or/xor (or
// gfx950's three-input v_bitop3_b32) works on 32 pixels at once — a radix selection from the most
// significant rank bit down, with carry-save adders counting the window:
//
//   level b:  z_i = alive_i & x_i[b]                       (i = 0..8: the window's pixels still in the race)
//             total = above + sum z_i                      (above = pixels known to be larger than the answer, < 4)
//             r_b = total >= 4                             (bit b of the 4th largest)
//             r_b ? alive_i = z_i : (alive_i &= ~x_i[b], above = total)
//
// Everything here is plain integer code shared by the HIP kernel (kpe_bs.hip) and a host test
// (tests/test_host_logic.py builds tests/cpp/bitslice_check.cpp against the oracle): the kernel adds
// only loads, stores and the lane-to-lane moves.
//
// Layout of a plane word: bit x (0..15) = pixel x of the lane's 16-pixel chunk in strip A, bit 16 + x =
// the same chunk in strip B (two rows of the image are processed per step, one in each half).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define RMX_HD __host__ __device__ __forceinline__
#else
#define RMX_HD inline
#endif

namespace rmx {
namespace bs {

// ---- three-input boolean operations ----------------------------------------------------------------
// v_bitop3_b32 takes an 8-bit truth table indexed by (a << 2 | b << 1 | c); the table of a functor is
// obtained by evaluating it on 0xF0 / 0xCC / 0xAA.  On the host the functor is evaluated directly.
template <class F>
RMX_HD uint32_t bop3(uint32_t a, uint32_t b, uint32_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr uint32_t imm = F{}(0xF0u, 0xCCu, 0xAAu) & 0xFFu;
  return __builtin_amdgcn_bitop3_b32(a, b, c, imm);
#else
  return F{}(a, b, c);
#endif
}
struct f_xor3 { constexpr uint32_t operator()(uint32_t a, uint32_t b, uint32_t c) const { return a ^ b ^ c; } };
struct f_maj { constexpr uint32_t operator()(uint32_t a, uint32_t b, uint32_t c) const { return (a & b) | (c & (a ^ b)); } };
struct f_nor3 { constexpr uint32_t operator()(uint32_t a, uint32_t b, uint32_t c) const { return ~(a | b | c); } };
struct f_mux { constexpr uint32_t operator()(uint32_t a, uint32_t b, uint32_t c) const { return (a & b) | (~a & c); } };  // a ? b : c
struct f_keep { constexpr uint32_t operator()(uint32_t z, uint32_t al, uint32_t nr) const { return z ^ (al & nr); } };
struct f_xorand { constexpr uint32_t operator()(uint32_t a, uint32_t b, uint32_t c) const { return (a ^ b) & c; } };

struct Sum2 { uint32_t s, c; };  // s + 2 c
RMX_HD Sum2 full_add(uint32_t a, uint32_t b, uint32_t c) { return Sum2{bop3<f_xor3>(a, b, c), bop3<f_maj>(a, b, c)}; }

// ---- bytes <-> planes -----------------------------------------------------------------------------------
// v_perm_b32 on the device; the same byte selection spelled out on the host.
RMX_HD uint32_t byte_perm(uint32_t hi, uint32_t lo, uint32_t sel) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_perm(hi, lo, sel);
#else
  const uint64_t v = (uint64_t)hi << 32 | lo;
  uint32_t r = 0;
  for (int i = 0; i < 4; ++i) r |= (uint32_t)((v >> (8 * ((sel >> (8 * i)) & 7u))) & 0xffu) << (8 * i);
  return r;
#endif
}

// exchange the bits of x at (mask << s) with the bits of y at mask
RMX_HD void delta_swap(uint32_t& x, uint32_t& y, uint32_t s, uint32_t mask) {
  const uint32_t t = ((x >> s) ^ y) & mask;
  y ^= t;
  x ^= t << s;
}

// a[0..3]: the 16 pixels of strip A (one 0..15 value per byte, pixel 4k + j in byte j of a[k]), b[0..3]:
// strip B.  n[q]: plane of bit q of the values, bit 16 s + x = pixel x of strip s.
// 4 shift-or + 8 byte permutes + 4 delta swaps (6 ops each) = 36 operations for 32 pixels (planes_from_packed:
// without the first four, for rows that are already two pixels per byte).
RMX_HD void planes_from_packed(const uint32_t (&c)[4], uint32_t (&n)[4]) {
  // c[k1] (k1 = 0, 1: strip A, 2, 3: strip B): pixels 8 k1 + 4 k0 + j of the 32 at bit 8 j + 4 k0 + q
  // 4 x 4 byte transpose: e_j byte i = byte j of c_i
  const uint32_t u0 = byte_perm(c[1], c[0], 0x05010400u), u1 = byte_perm(c[1], c[0], 0x07030602u);
  const uint32_t u2 = byte_perm(c[3], c[2], 0x05010400u), u3 = byte_perm(c[3], c[2], 0x07030602u);
  uint32_t e0 = byte_perm(u2, u0, 0x05040100u), e1 = byte_perm(u2, u0, 0x07060302u);
  uint32_t e2 = byte_perm(u3, u1, 0x05040100u), e3 = byte_perm(u3, u1, 0x07060302u);
  // e_j: bit 8 i + 4 k0 + q, i = (s, k1).  Exchange the dword index j with the bit index q.
  delta_swap(e0, e1, 1, 0x55555555u);
  delta_swap(e2, e3, 1, 0x55555555u);
  delta_swap(e0, e2, 2, 0x33333333u);
  delta_swap(e1, e3, 2, 0x33333333u);
  n[0] = e0, n[1] = e1, n[2] = e2, n[3] = e3;
}
RMX_HD void planes_from_bytes(const uint32_t (&a)[4], const uint32_t (&b)[4], uint32_t (&n)[4]) {
  // two pixels per byte: dword (s, k1) holds pixels 8 k1 + 4 k0 + j at bit 8 j + 4 k0 + q
  const uint32_t c[4] = {a[0] | a[1] << 4, a[2] | a[3] << 4, b[0] | b[1] << 4, b[2] | b[3] << 4};
  planes_from_packed(c, n);
}

// the inverse: 44 operations for 32 pixels
RMX_HD void bytes_from_planes(const uint32_t (&n)[4], uint32_t (&a)[4], uint32_t (&b)[4]) {
  uint32_t e0 = n[0], e1 = n[1], e2 = n[2], e3 = n[3];
  delta_swap(e0, e2, 2, 0x33333333u);
  delta_swap(e1, e3, 2, 0x33333333u);
  delta_swap(e0, e1, 1, 0x55555555u);
  delta_swap(e2, e3, 1, 0x55555555u);
  const uint32_t u0 = byte_perm(e1, e0, 0x05010400u), u1 = byte_perm(e1, e0, 0x07030602u);
  const uint32_t u2 = byte_perm(e3, e2, 0x05010400u), u3 = byte_perm(e3, e2, 0x07030602u);
  const uint32_t c0 = byte_perm(u2, u0, 0x05040100u), c1 = byte_perm(u2, u0, 0x07060302u);
  const uint32_t c2 = byte_perm(u3, u1, 0x05040100u), c3 = byte_perm(u3, u1, 0x07060302u);
  const uint32_t m = 0x0f0f0f0fu;
  a[0] = c0 & m, a[1] = (c0 >> 4) & m, a[2] = c1 & m, a[3] = (c1 >> 4) & m;
  b[0] = c2 & m, b[1] = (c2 >> 4) & m, b[2] = c3 & m, b[3] = (c3 >> 4) & m;
}

// ---- neighbours -----------------------------------------------------------------------------------------
// Both 16-bit halves of a plane shift on their own (v_pk_lshlrev_b16 / v_pk_lshrrev_b16); the bit that
// leaves a half enters the same half of the neighbouring lane's word.
RMX_HD uint32_t half_shl(uint32_t v, uint32_t s) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
  u16x2 a = __builtin_bit_cast(u16x2, v);
  a = a << (unsigned short)s;
  return __builtin_bit_cast(uint32_t, a);
#else
  return ((v & 0xffffu) << s & 0xffffu) | ((v >> 16) << s & 0xffffu) << 16;
#endif
}
RMX_HD uint32_t half_shr(uint32_t v, uint32_t s) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
  u16x2 a = __builtin_bit_cast(u16x2, v);
  a = a >> (unsigned short)s;
  return __builtin_bit_cast(uint32_t, a);
#else
  return ((v & 0xffffu) >> s) | ((v >> 16) >> s) << 16;
#endif
}
// plane of the left neighbours (pixel x - 1) given what the previous lane sends (send_right of ITS plane)
RMX_HD uint32_t send_right(uint32_t p) { return half_shr(p, 15); }  // pixel 15 -> bit 0 of the next lane
RMX_HD uint32_t send_left(uint32_t p) { return half_shl(p, 15); }   // pixel 0 -> bit 15 of the previous lane
RMX_HD uint32_t left_of(uint32_t p, uint32_t from_prev) { return half_shl(p, 1) | from_prev; }
RMX_HD uint32_t right_of(uint32_t p, uint32_t from_next) { return half_shr(p, 1) | from_next; }

// ---- 16-entry tables on planes --------------------------------------------------------------------------
// out bit q of T[v]: a function of the four input planes, as mux(v3; g1(v2,v1,v0), g0(v2,v1,v0)).
template <uint64_t TBL, int Q>  // TBL: sixteen nibbles, entry v at bits 4 v
struct f_tbl_half {
  static constexpr uint32_t tt(int hi) {
    uint32_t t = 0;
    for (int i = 0; i < 8; ++i) t |= (uint32_t)((TBL >> (4 * (8 * hi + i) + Q)) & 1u) << i;
    return t;
  }
};
template <uint32_t TT>
struct f_tt {  // the function of (a, b, c) with truth table TT (index a << 2 | b << 1 | c)
  constexpr uint32_t operator()(uint32_t a, uint32_t b, uint32_t c) const {
    uint32_t r = 0;
    for (int i = 0; i < 8; ++i)
      if ((TT >> i) & 1u) r |= ((i & 4) ? a : ~a) & ((i & 2) ? b : ~b) & ((i & 1) ? c : ~c);
    return r;
  }
};
constexpr uint64_t pack_tbl(const uint8_t (&t)[16], bool invert_index) {
  uint64_t r = 0;
  for (int i = 0; i < 16; ++i) r |= (uint64_t)(t[invert_index ? 15 - i : i] & 15u) << (4 * i);
  return r;
}
template <uint64_t TBL, int Q>
RMX_HD uint32_t map_plane(const uint32_t (&v)[4]) {
  const uint32_t g0 = bop3<f_tt<f_tbl_half<TBL, Q>::tt(0)>>(v[2], v[1], v[0]);
  const uint32_t g1 = bop3<f_tt<f_tbl_half<TBL, Q>::tt(1)>>(v[2], v[1], v[0]);
  return bop3<f_mux>(v[3], g1, g0);
}
template <uint64_t TBL>
RMX_HD void map_planes(const uint32_t (&v)[4], uint32_t (&o)[4]) {
  o[0] = map_plane<TBL, 0>(v);
  o[1] = map_plane<TBL, 1>(v);
  o[2] = map_plane<TBL, 2>(v);
  o[3] = map_plane<TBL, 3>(v);
}

// The two tables of the extraction (reference cpl.hpp:163-217): native colour -> luminance rank, and
// (complemented rank) -> native colour for the way back.  Every output bit is a function of four planes:
// generically three operations (two halves and a mux); an exhaustive search over two-operation forms
// (a three-input function of three planes, then one of its result and two planes) found shorter ones for
// four of the eight bits — 19 operations for both tables instead of 24.  tests/test_host_logic.py checks
// these against the tables on all sixteen values and, through the medians, against the oracle.
constexpr uint8_t kRankOfNative[16] = {0, 15, 2, 12, 6, 9, 3, 13, 5, 1, 7, 4, 8, 14, 10, 11};
constexpr uint8_t kNativeOfRank[16] = {0, 9, 2, 6, 11, 8, 4, 10, 12, 5, 14, 15, 3, 7, 13, 1};
RMX_HD void rank_planes(const uint32_t (&n)[4], uint32_t (&x)[4]) {
  constexpr uint64_t T = pack_tbl(kRankOfNative, false);
  x[0] = map_plane<T, 0>(n);
  x[1] = map_plane<T, 1>(n);
  x[2] = map_plane<T, 2>(n);
  x[3] = bop3<f_tt<216>>(n[0], n[2], n[3]);
}
RMX_HD void native_planes_of_inverted_rank(const uint32_t (&nr)[4], uint32_t (&o)[4]) {
  constexpr uint64_t T = pack_tbl(kNativeOfRank, true);
  o[0] = bop3<f_tt<208>>(bop3<f_tt<61>>(nr[0], nr[2], nr[3]), nr[1], nr[3]);
  o[1] = bop3<f_tt<54>>(bop3<f_tt<10>>(nr[0], nr[1], nr[3]), nr[1], nr[2]);
  o[2] = map_plane<T, 2>(nr);
  o[3] = bop3<f_tt<150>>(bop3<f_tt<24>>(nr[0], nr[1], nr[2]), nr[2], nr[3]);
}

// ---- the window -----------------------------------------------------------------------------------------
// One image row as the selection sees it: rank planes of the pixel itself (c), of its left (l) and right
// (r) neighbour, and the horizontal count of the most significant bit (hs + 2 hc = l[3] + c[3] + r[3]).
struct Row {
  uint32_t c[4], l[4], r[4];
  uint32_t hs, hc;
};
RMX_HD void row_sums(Row& R) {
  const Sum2 h = full_add(R.l[3], R.c[3], R.r[3]);
  R.hs = h.s;
  R.hc = h.c;
}

// Planes of the complemented bits of the 4th largest rank of the 3x3 window made of rows T (above),
// M (the pixel's own row) and B (below): nr[q] = ~bit q.
RMX_HD void select4of9(const Row& T, const Row& M, const Row& B, uint32_t (&nr)[4]) {
  uint32_t al[9], a0, a1;
  {  // bit 3: the count is separable (row sums are shared by the three windows a row belongs to)
    const Sum2 lo = full_add(T.hs, M.hs, B.hs);  // weights 1, 2
    const Sum2 hi = full_add(T.hc, M.hc, B.hc);  // weights 2, 4
    const uint32_t t1 = hi.s ^ lo.c, w = hi.s & lo.c;  // weights 2, 4
    const uint32_t n = bop3<f_nor3>(hi.c, w, w);       // total < 4
    nr[3] = n;
    a0 = lo.s & n;  // bit 3 of the answer is 0: the pixels with the bit set lie above it
    a1 = t1 & n;
    al[0] = T.l[3] ^ n, al[1] = T.c[3] ^ n, al[2] = T.r[3] ^ n;  // x ^ ~r: the pixels whose bit equals r
    al[3] = M.l[3] ^ n, al[4] = M.c[3] ^ n, al[5] = M.r[3] ^ n;
    al[6] = B.l[3] ^ n, al[7] = B.c[3] ^ n, al[8] = B.r[3] ^ n;
  }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int q = 2; q >= 0; --q) {
    uint32_t z[9];
    z[0] = al[0] & T.l[q], z[1] = al[1] & T.c[q], z[2] = al[2] & T.r[q];
    z[3] = al[3] & M.l[q], z[4] = al[4] & M.c[q], z[5] = al[5] & M.r[q];
    z[6] = al[6] & B.l[q], z[7] = al[7] & B.c[q], z[8] = al[8] & B.r[q];
    const Sum2 r0 = full_add(z[0], z[1], z[2]), r1 = full_add(z[3], z[4], z[5]), r2 = full_add(z[6], z[7], z[8]);
    const Sum2 lo = full_add(r0.s, r1.s, r2.s);        // weights 1, 2
    const uint32_t t0 = lo.s ^ a0, d = lo.s & a0;      // weights 1, 2
    const Sum2 m0 = full_add(r0.c, r1.c, r2.c);        // weights 2, 4
    const Sum2 m1 = full_add(lo.c, d, a1);             // weights 2, 4
    const uint32_t t1 = m0.s ^ m1.s, i = m0.s & m1.s;  // weights 2, 4
    const uint32_t n = bop3<f_nor3>(m0.c, m1.c, i);    // above + count < 4: bit q of the answer is 0
    nr[q] = n;
    if (q) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
      for (int k = 0; k < 9; ++k) al[k] = bop3<f_keep>(z[k], al[k], n);  // r ? z : alive & ~x
      a0 = bop3<f_mux>(n, t0, a0);
      a1 = bop3<f_mux>(n, t1, a1);
    }
  }
}

// candidate flags: pixels (inside `valid`) whose own rank differs from the selected one
RMX_HD uint32_t differs(const uint32_t (&nr)[4], const uint32_t (&c)[4], uint32_t valid) {
  uint32_t eq = nr[0] ^ c[0];  // ~r ^ c = (r == c)
  eq = bop3<f_xorand>(nr[1], c[1], eq);
  eq = bop3<f_xorand>(nr[2], c[2], eq);
  eq = bop3<f_xorand>(nr[3], c[3], eq);
  return valid & ~eq;
}

}  // namespace bs
}  // namespace rmx
