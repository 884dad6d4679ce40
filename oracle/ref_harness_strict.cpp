// ref_harness_strict.cpp — thin extern "C" shell around the REFERENCE's own headers,
// compiled UNMODIFIED from where they lie (/root/reference/src) by oracle/build_ref.sh
// into oracle/_ref/libref_strict.so.  Test infrastructure only; never shipped, never
// linked by the product.  Contains no reference code: it only calls it.
//
// Covers every hot-path header that g++ 11 accepts as-is (flags only: -fpermissive -w):
//   cpl.hpp (LUTs), kpr.hpp (region/grid), kpm.hpp (both match variants), fgm.hpp (canvas),
//   ful.hpp (fragment files).
// kpe.hpp / frc.hpp / fgs.hpp are MSVC-only as written; see ref_harness_full.cpp.

// The reference headers rely on MSVC's transitive includes and on include order.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <list>
#include <numeric>
#include <optional>

#include "fgm.hpp"  // pulls icd.hpp -> sid.hpp -> cpl.hpp, mrl.hpp (kpm.hpp needs sid::)
#include "kpm.hpp"
#include "nic.hpp"  // the codec main.cpp plugs into collect as Comp (main.cpp:113-125)
#include "ful.hpp"  // fragment file format; includes fgm.hpp again (pragma once)

#include "../include/remap_amd.h"

namespace {

using alloc_t = std::allocator<char>;

// Same constants as frc::collector::match_config (frc.hpp:30-44), which is private there.
struct seq_config {
  using allocator_type = alloc_t;
  static constexpr std::size_t weight_switch{10};
  static constexpr std::size_t region_votes{3};
  allocator_type get_allocator() const noexcept { return {}; }
};

template <std::size_t... I>
void add_masked(kpr::grid<4, 2, alloc_t>& g, kpr::code const& c, mrl::point_t p, uint8_t mask,
                std::index_sequence<I...>) {
  // kpe.hpp adds a keypoint to each region of its band set in ascending region index
  // (kpr.hpp:189-219); per-region maps are independent, so one add per set bit is equal.
  ((mask >> I & 1 ? g.add(c, p, std::index_sequence<I>{}) : void()), ...);
}

kpr::grid<4, 2, alloc_t> make_grid8(const rmx_desc* d, uint32_t n) {
  kpr::grid<4, 2, alloc_t> g{alloc_t{}};
  for (uint32_t i = 0; i < n; ++i) {
    kpr::code c;
    std::memcpy(c.data(), d[i].code, 13);
    add_masked(g, c, mrl::point_t{d[i].x, d[i].y}, d[i].region_mask, std::make_index_sequence<8>{});
  }
  return g;
}

kpr::grid<1, 1, alloc_t> make_grid1(const rmx_desc* d, uint32_t n) {
  kpr::grid<1, 1, alloc_t> g{alloc_t{}};
  for (uint32_t i = 0; i < n; ++i) {
    kpr::code c;
    std::memcpy(c.data(), d[i].code, 13);
    g.add(c, mrl::point_t{d[i].x, d[i].y}, std::index_sequence<0>{});
  }
  return g;
}

}  // namespace

extern "C" {

void ref_luts(uint8_t n2o[16], uint8_t o2n[16]) {
  for (uint8_t i = 0; i < 16; ++i) {
    n2o[i] = cpl::native_to_ordered({i}).value;
    o2n[i] = cpl::ordered_to_native({i}).value;
  }
}

uint64_t ref_hash_code(const uint8_t code[13]) {
  kpr::code c;
  std::memcpy(c.data(), code, 13);
  return kpr::code_hash{}(c);
}
uint64_t ref_hash_offset(int32_t x, int32_t y) { return cdt::offset_hash{}(cdt::offset_t{x, y}); }

// kpm::match(cfg, previous, current), kpm.hpp:395-415
int ref_kpm_grid(const rmx_desc* prev, uint32_t nprev, const rmx_desc* cur, uint32_t ncur, int32_t dxdy[2],
                 rmx_vote* tickets /* 8*3 or NULL */) {
  auto P = make_grid8(prev, nprev);
  auto C = make_grid8(cur, ncur);
  dxdy[0] = dxdy[1] = 0;
  if (tickets) {
    std::memset(tickets, 0, sizeof(rmx_vote) * 24);
    for (std::size_t r = 0; r < 8; ++r) {
      auto t = kpm::details::cast_vote(seq_config{}, P.regions()[r], C.regions()[r]);
      for (std::size_t j = 0; j < t.size(); ++j)
        tickets[r * 3 + j] = rmx_vote{t[j].offset_.x_, t[j].offset_.y_, uint32_t(t[j].count_), 1};
    }
  }
  auto r = kpm::match(seq_config{}, P, C);
  if (!r) return 0;
  dxdy[0] = r->x_;
  dxdy[1] = r->y_;
  return 1;
}

// kpm::match(preg, pmask, creg, cmask, cell), kpm.hpp:371-393
int ref_kpm_cell(const rmx_desc* prev, uint32_t nprev, const uint8_t* pmask, uint32_t pw, uint32_t ph,
                 const rmx_desc* cur, uint32_t ncur, const uint8_t* cmask, uint32_t cw, uint32_t ch,
                 uint32_t cell_w, uint32_t cell_h, rmx_vote* vote) {
  auto P = make_grid1(prev, nprev);
  auto C = make_grid1(cur, ncur);
  sid::mon::dimg_t pm{mrl::dimensions_t{pw, ph}}, cm{mrl::dimensions_t{cw, ch}};
  for (std::size_t i = 0; i < pm.size(); ++i) pm.data()[i] = cpl::mon_bv{pmask[i]};
  for (std::size_t i = 0; i < cm.size(); ++i) cm.data()[i] = cpl::mon_bv{cmask ? cmask[i] : uint8_t(1)};
  *vote = rmx_vote{0, 0, 0, 0};
  auto r = kpm::match(P[0], pm, C[0], cm, kpm::cell_size_t{uint8_t(cell_w), uint8_t(cell_h)});
  if (!r) return 0;
  *vote = rmx_vote{r->offset_.x_, r->offset_.y_, uint32_t(r->count_), 1};
  return 1;
}

// ---- fgm::fragment (fgm.hpp:39-304) through an opaque handle ------------------------------
void* ref_frag_new(uint32_t step_w, uint32_t step_h) {
  return new fgm::fragment(mrl::dimensions_t{step_w, step_h});
}
void ref_frag_free(void* f) { delete static_cast<fgm::fragment*>(f); }
void ref_frag_blit(void* f, int32_t px, int32_t py, const uint8_t* img, uint32_t W, uint32_t H, uint64_t no) {
  sid::nat::dimg_t image{mrl::dimensions_t{W, H}};
  for (std::size_t i = 0; i < image.size(); ++i) image.data()[i] = cpl::nat_cc{img[i]};
  static_cast<fgm::fragment*>(f)->blit(fgm::point_t{px, py}, image, fgm::packed_data{}, no);
}
// dst.blit(pos, std::move(src)); src is consumed (and freed).
void ref_frag_merge(void* dst, int32_t px, int32_t py, void* src) {
  static_cast<fgm::fragment*>(dst)->blit(fgm::point_t{px, py}, std::move(*static_cast<fgm::fragment*>(src)));
  delete static_cast<fgm::fragment*>(src);
}
void ref_frag_normalize(void* f) { static_cast<fgm::fragment*>(f)->normalize(); }
void ref_frag_info(void* f, uint32_t* w, uint32_t* h, int32_t* zx, int32_t* zy, uint64_t* nframes) {
  auto* fr = static_cast<fgm::fragment*>(f);
  *w = uint32_t(fr->dimensions().width_);
  *h = uint32_t(fr->dimensions().height_);
  *zx = fr->zero().x_;
  *zy = fr->zero().y_;
  *nframes = fr->frames().size();
}
void ref_frag_dots(void* f, uint16_t* out) {
  auto* fr = static_cast<fgm::fragment*>(f);
  std::memcpy(out, fr->dots().data(), fr->dots().size() * sizeof(fgm::dot_type));
}
void ref_frag_frames(void* f, uint64_t* numbers, int32_t* pos) {
  auto* fr = static_cast<fgm::fragment*>(f);
  std::size_t i = 0;
  for (auto& fm : fr->frames()) {
    numbers[i] = fm.number_;
    pos[2 * i] = fm.position_.x_;
    pos[2 * i + 1] = fm.position_.y_;
    ++i;
  }
}
// blit with the frame's compressed blobs attached (what frc::collector stores, frc.hpp:118)
void ref_frag_blit_packed(void* f, int32_t px, int32_t py, const uint8_t* img, uint32_t W, uint32_t H, uint64_t no,
                          const uint8_t* bi, uint64_t ni, const uint8_t* bm, uint64_t nm) {
  sid::nat::dimg_t image{mrl::dimensions_t{W, H}};
  for (std::size_t i = 0; i < image.size(); ++i) image.data()[i] = cpl::nat_cc{img[i]};
  fgm::packed_data pk{icd::compressed_t(bi, bi + ni), icd::compressed_t(bm, bm + nm)};
  static_cast<fgm::fragment*>(f)->blit(fgm::point_t{px, py}, image, std::move(pk), no);
}
void ref_frag_blob_sizes(void* f, uint64_t* ni, uint64_t* nm) {
  std::size_t i = 0;
  for (auto& fm : static_cast<fgm::fragment*>(f)->frames()) {
    ni[i] = fm.data_.image_.size();
    nm[i] = fm.data_.median_.size();
    ++i;
  }
}
void ref_frag_blobs(void* f, uint8_t* bi, uint8_t* bm) {
  for (auto& fm : static_cast<fgm::fragment*>(f)->frames()) {
    std::memcpy(bi, fm.data_.image_.data(), fm.data_.image_.size());
    std::memcpy(bm, fm.data_.median_.data(), fm.data_.median_.size());
    bi += fm.data_.image_.size();
    bm += fm.data_.median_.size();
  }
}

// ---- nic::compress / nic::decompress (nic.hpp:8-166) -----------------------------------------
uint64_t ref_nic_compress(const uint8_t* px, uint32_t W, uint32_t H, uint8_t* out, uint64_t cap) {
  sid::nat::dimg_t image{mrl::dimensions_t{W, H}};
  for (std::size_t i = 0; i < image.size(); ++i) image.data()[i] = cpl::nat_cc{px[i]};
  auto r = nic::compress(image);
  std::memcpy(out, r.data(), std::min<std::size_t>(cap, r.size()));
  return r.size();
}
void ref_nic_decompress(const uint8_t* pack, uint64_t size, uint32_t W, uint32_t H, uint8_t* out) {
  auto img = nic::decompress(icd::compressed_t(pack, pack + size), mrl::dimensions_t{W, H});
  for (std::size_t i = 0; i < img.size(); ++i) out[i] = img.data()[i].value;
}

// ---- ful::write / ful::read (ful.hpp:10-102) ------------------------------------------------
namespace {
struct frag_ptr_iter {  // iterator over fragment handles with the `first->` shape ful::write uses
  void** p;
  fgm::fragment* operator->() const { return static_cast<fgm::fragment*>(*p); }
  frag_ptr_iter& operator++() {
    ++p;
    return *this;
  }
  bool operator!=(frag_ptr_iter const& o) const { return p != o.p; }
};
}  // namespace
void ref_ful_write(const char* dir, void** frags, uint32_t n) {
  ful::write(std::filesystem::path{dir}, frag_ptr_iter{frags}, frag_ptr_iter{frags + n});
}
void* ref_ful_read(const char* dir, uint32_t* n) {
  auto* v = new std::vector<fgm::fragment>(ful::read(std::filesystem::path{dir}));
  *n = uint32_t(v->size());
  return v;
}
void* ref_ful_get(void* vec, uint32_t i) { return &(*static_cast<std::vector<fgm::fragment>*>(vec))[i]; }
void ref_ful_free(void* vec) { delete static_cast<std::vector<fgm::fragment>*>(vec); }

// fragment::margins (fgm.hpp:145-153,235-295): {left, top, right, bottom}
void ref_frag_margins(void* f, uint64_t out[4]) {
  auto r = static_cast<fgm::fragment*>(f)->margins();
  out[0] = r.left_;
  out[1] = r.top_;
  out[2] = r.right_;
  out[3] = r.bottom_;
}

void ref_frag_blend(void* f, uint8_t* image, uint8_t* mask) {
  auto b = static_cast<fgm::fragment*>(f)->blend();
  for (std::size_t i = 0; i < b.image_.size(); ++i) {
    image[i] = b.image_.data()[i].value;
    mask[i] = b.mask_.data()[i].value;
  }
}

}  // extern "C"
