// bitslice_check.cpp — host-side driver of remap_amd/csrc/bitslice.h for the CPU test suite.
// Runs the bit-plane 3x3 pass exactly as kpe_bs.hip arranges it (16-pixel chunks, two strips per word,
// neighbours handed between "lanes") on one frame and returns the median plane and the candidate flags;
// tests/test_host_logic.py compares them with the oracle's kpe (reference kpe.hpp:308-340).
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../remap_amd/csrc/bitslice.h"

using namespace rmx::bs;

namespace {
constexpr uint8_t kN2O[16] = {0, 15, 2, 12, 6, 9, 3, 13, 5, 1, 7, 4, 8, 14, 10, 11};
constexpr uint8_t kO2N[16] = {0, 9, 2, 6, 11, 8, 4, 10, 12, 5, 14, 15, 3, 7, 13, 1};
constexpr uint64_t kRankTbl = pack_tbl(kN2O, false);
constexpr uint64_t kNativeOfInvRank = pack_tbl(kO2N, true);
}  // namespace

// the plane forms of both tables on all sixteen values (bit v of plane q = bit q of value v)
extern "C" int bs_tables_ok() {
  uint32_t v[4] = {0xaaaau, 0xccccu, 0xf0f0u, 0xff00u}, x[4], inv[4], o[4];
  rank_planes(v, x);
  for (int q = 0; q < 4; ++q) inv[q] = ~x[q];
  native_planes_of_inverted_rank(inv, o);
  for (int i = 0; i < 16; ++i) {
    int r = 0, back = 0;
    for (int q = 0; q < 4; ++q) r |= ((x[q] >> i) & 1) << q, back |= ((o[q] >> i) & 1) << q;
    if (r != kN2O[i] || back != i) return 0;
  }
  uint32_t g[4];
  map_planes<kRankTbl>(v, g);  // the generic three-operation form gives the same planes
  for (int q = 0; q < 4; ++q)
    if ((g[q] & 0xffffu) != (x[q] & 0xffffu)) return 0;
  map_planes<kNativeOfInvRank>(inv, g);
  for (int q = 0; q < 4; ++q)
    if ((g[q] & 0xffffu) != (o[q] & 0xffffu)) return 0;
  return 1;
}

extern "C" int bs_planes_roundtrip(const uint8_t* px32, uint8_t* out32, uint32_t* planes4) {
  uint32_t a[4], b[4], n[4], a2[4], b2[4];
  std::memcpy(a, px32, 16);
  std::memcpy(b, px32 + 16, 16);
  planes_from_bytes(a, b, n);
  std::memcpy(planes4, n, 16);
  bytes_from_planes(n, a2, b2);
  std::memcpy(out32, a2, 16);
  std::memcpy(out32 + 16, b2, 16);
  return 0;
}

// frame: H rows of W bytes (W % 16 == 0).  Strip A of a step is row y, strip B row y + H / 2 (wrapping):
// both halves of every word carry real pixels.  median / cand: H x W bytes.
extern "C" int bs_kpe_3x3(const uint8_t* frame, uint32_t W, uint32_t H, uint8_t* median, uint8_t* cand) {
  if (W % 16 || H < 8) return -1;
  const uint32_t nc = W / 16;
  std::memset(median, 0, (size_t)W * H);
  std::memset(cand, 0, (size_t)W * H);
  auto rowB = [&](int y) { return (uint32_t)((y + (int)H / 2) % (int)H); };
  // rank planes per (row pair, chunk)
  std::vector<Row> rows((size_t)H * nc);
  for (uint32_t y = 0; y < H; ++y) {
    std::vector<uint32_t> pl((size_t)nc * 4);
    for (uint32_t c = 0; c < nc; ++c) {
      uint32_t a[4], b[4], n[4], x[4];
      std::memcpy(a, frame + (size_t)y * W + 16 * c, 16);
      std::memcpy(b, frame + (size_t)rowB((int)y) * W + 16 * c, 16);
      planes_from_bytes(a, b, n);
      rank_planes(n, x);
      std::memcpy(&pl[4 * c], x, 16);
    }
    for (uint32_t c = 0; c < nc; ++c) {
      Row& R = rows[(size_t)y * nc + c];
      for (int q = 0; q < 4; ++q) {
        const uint32_t p = pl[4 * c + q];
        const uint32_t fp = c ? send_right(pl[4 * (c - 1) + q]) : 0u, fn = c + 1 < nc ? send_left(pl[4 * (c + 1) + q]) : 0u;
        R.c[q] = p;
        R.l[q] = left_of(p, fp);
        R.r[q] = right_of(p, fn);
      }
      row_sums(R);
    }
  }
  for (uint32_t y = 0; y < H; ++y) {
    const uint32_t yb = rowB((int)y);
    const bool va = y >= 2 && y + 5 <= H, vb = yb >= 2 && yb + 5 <= H;
    // window rows of strip B are the strip-B halves of steps y - 1, y, y + 1 only where rowB is contiguous
    const uint32_t ym = (y + H - 1) % H, yp = (y + 1) % H;
    const bool contig = rowB((int)ym) + 1 == yb && yb + 1 == rowB((int)yp);
    for (uint32_t c = 0; c < nc; ++c) {
      uint32_t valid = 0;
      for (uint32_t x = 0; x < 16; ++x) {
        const uint32_t gx = 16 * c + x;
        if (gx >= 2 && gx + 3 <= W) valid |= 0x00010001u << x;
      }
      if (!(va && ym + 1 == y && y + 1 == yp)) valid &= 0xffff0000u;
      if (!(vb && contig)) valid &= 0x0000ffffu;
      uint32_t nr[4], nat[4], a[4], b[4];
      select4of9(rows[(size_t)ym * nc + c], rows[(size_t)y * nc + c], rows[(size_t)yp * nc + c], nr);
      const uint32_t cd = differs(nr, rows[(size_t)y * nc + c].c, valid);
      native_planes_of_inverted_rank(nr, nat);
      for (int q = 0; q < 4; ++q) nat[q] &= valid;
      bytes_from_planes(nat, a, b);
      if (valid & 0xffffu) {
        std::memcpy(median + (size_t)y * W + 16 * c, a, 16);
        for (uint32_t x = 0; x < 16; ++x) cand[(size_t)y * W + 16 * c + x] = (cd >> x) & 1u;
      }
      if (valid >> 16) {
        std::memcpy(median + (size_t)yb * W + 16 * c, b, 16);
        for (uint32_t x = 0; x < 16; ++x) cand[(size_t)yb * W + 16 * c + x] = (cd >> (16 + x)) & 1u;
      }
    }
  }
  return 0;
}
