// paircmp_check.cpp — host-side driver of remap_amd/csrc/paircmp.h for the CPU test suite.
// Runs the pairwise-comparison candidate test exactly as kpe_pc.hip arranges it (chunks of 32 consecutive
// pixels per "lane", a row's comparisons handed from the step of the row above to the step that finishes it,
// neighbours handed between lanes — rubbish at the ends of a row, as the kernel's lanes receive it from the
// neighbouring frame) and returns the candidate flags; tests/test_host_logic.py compares them with the pixels
// whose median differs in the oracle's kpe (reference kpe.hpp:308-340).
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../remap_amd/csrc/paircmp.h"

using namespace rmx;

namespace {
struct RowPlanes {
  uint32_t c[4], r[4];
};
uint32_t shifted(const std::vector<uint32_t>& row, uint32_t c, uint32_t junk) {  // from_left over the lanes of a row
  return pc::from_left(row[c], c ? row[c - 1] : junk);
}
}  // namespace

// frame: H rows of W bytes (W % 32 == 0), cand: H x W bytes (1 = candidate).  junk: what the first / last lane
// of a row receives from outside the row.
extern "C" int pc_kpe_cand(const uint8_t* frame, uint32_t W, uint32_t H, uint8_t* cand, uint32_t junk) {
  if (W % 32 || H < 8) return -1;
  const uint32_t nc = W / 32;
  std::memset(cand, 0, (size_t)W * H);
  std::vector<RowPlanes> rows((size_t)H * nc);
  for (uint32_t y = 0; y < H; ++y) {
    for (uint32_t c = 0; c < nc; ++c) {
      uint32_t a[4], b[4], n[4];
      std::memcpy(a, frame + (size_t)y * W + 32 * c, 16);
      std::memcpy(b, frame + (size_t)y * W + 32 * c + 16, 16);
      bs::planes_from_bytes(a, b, n);
      bs::rank_planes(n, rows[(size_t)y * nc + c].c);
    }
    for (uint32_t c = 0; c < nc; ++c)
      for (int q = 0; q < 4; ++q)
        rows[(size_t)y * nc + c].r[q] = pc::from_right(rows[(size_t)y * nc + c].c[q], c + 1 < nc ? rows[(size_t)y * nc + c + 1].c[q] : junk);
  }
  // the steps: row B arrives, row M = B - 1 is finished with what the step before left (Upper)
  std::vector<pc::Upper> up(nc);
  std::vector<uint32_t> g(nc), e(nc);
  auto shift_pair = [&](const std::vector<pc::Cmp>& v, std::vector<pc::Cmp>& out) {
    for (uint32_t c = 0; c < nc; ++c) g[c] = v[c].gt, e[c] = v[c].ge;
    for (uint32_t c = 0; c < nc; ++c) out[c] = pc::Cmp{shifted(g, c, junk), shifted(e, c, junk)};
  };
  std::vector<pc::Cmp> south(nc), se(nc), sw(nc), east(nc), se_sh(nc), sw_sh(nc), east_sh(nc);
  for (uint32_t yb = 0; yb < H; ++yb) {  // B = yb, M = yb - 1, A = yb - 2
    for (uint32_t c = 0; c < nc; ++c) {
      const RowPlanes& B = rows[(size_t)yb * nc + c];
      const RowPlanes& M = rows[(size_t)(yb ? yb - 1 : 0) * nc + c];
      south[c] = pc::compare(B.c, M.c);
      se[c] = pc::compare(B.r, M.c);
      sw[c] = pc::compare(M.r, B.c);
      east[c] = pc::compare(B.r, B.c);
    }
    shift_pair(se, se_sh);
    shift_pair(sw, sw_sh);
    shift_pair(east, east_sh);
    const int ym = (int)yb - 1;
    for (uint32_t c = 0; c < nc; ++c) {
      if (ym >= 2 && ym + 5 <= (int)H) {
        uint32_t valid = 0;
        for (uint32_t x = 0; x < 32; ++x) {
          const uint32_t gx = 32 * c + x;
          if (gx >= 2 && gx + 3 <= W) valid |= 1u << x;
        }
        const uint32_t cd = pc::not_fourth(up[c], south[c], se[c], sw_sh[c], valid);
        for (uint32_t x = 0; x < 32; ++x) cand[(size_t)ym * W + 32 * c + x] = (cd >> x) & 1u;
      }
      up[c] = pc::Upper{south[c], sw[c], se_sh[c], east[c], east_sh[c]};
    }
  }
  return 0;
}
