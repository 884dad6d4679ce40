// ref_harness_fdf.cpp — extern "C" shell around the reference's fdf.hpp / fde.hpp / cte.hpp
// (foreground filtering).  Test infrastructure only; contains no
// reference code, it only calls it.
//
// TIER 2, its own library (oracle/_ref/libref_fdf.so): fde.hpp includes kpe.hpp (the four
// spelling patches of build_ref.sh) and <intrin.h> itself (a fifth, fde.hpp:10: -> <immintrin.h>), uses
// an AVX-512VL/BW intrinsic (_mm256_loadu_epi8: -mavx512vl -mavx512bw, so this library is
// only ever loaded on the machine that built it, never by bench.py) and dereferences
// __m256i pointers into std::vector storage, which MSVC compiles to unaligned moves and g++
// to aligned ones (SIGSEGV as is).  The harness therefore gives the library
// its own 32-byte-aligned global operator new/delete (linked -Bsymbolic): memory comes back
// aligned the way the reference assumes, no reference line is touched for it.

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <list>
#include <new>
#include <numeric>
#include <optional>

void* operator new(std::size_t n) {
  void* p = std::aligned_alloc(32, (n + 31) & ~std::size_t(31));
  if (!p) throw std::bad_alloc();
  return p;
}
void* operator new[](std::size_t n) { return operator new(n); }
void operator delete(void* p) noexcept { std::free(p); }
void operator delete[](void* p) noexcept { std::free(p); }
void operator delete(void* p, std::size_t) noexcept { std::free(p); }
void operator delete[](void* p, std::size_t) noexcept { std::free(p); }

#include "frc.hpp"
#include "fdf.hpp"

namespace {

struct mem_feed {
  const uint8_t* frames;
  uint32_t n, W, H;
  uint32_t next{0};
  bool has_more() const noexcept { return next < n; }
  template <typename Alloc>
  auto produce(Alloc const& alloc) {
    using image_type = sid::nat::aimg_t<Alloc>;
    image_type img{mrl::dimensions_t{W, H}, alloc};
    std::memcpy(img.data(), frames + std::size_t(next) * W * H, std::size_t(W) * H);
    return ifd::frame<image_type>{std::size_t(next++), std::move(img)};
  }
};

// a codec that stores the pixels as they are (alignment results and the filter do not depend
// on the codec; nic::compress cannot encode a blank frame, see oracle/remap_oracle.cpp)
struct raw_codec {
  template <typename Alloc>
  icd::compressed_t operator()(sid::nat::aimg_t<Alloc> const& image) const {
    auto p = reinterpret_cast<std::uint8_t const*>(image.data());
    return icd::compressed_t(p, p + image.size());
  }
  sid::nat::dimg_t operator()(icd::compressed_t const& c, mrl::dimensions_t const& dim) const {
    sid::nat::dimg_t r{dim};
    std::memcpy(r.data(), c.data(), dim.area());
    return r;
  }
};

}  // namespace

extern "C" {

// frc::collector::collect + complete, then fdf::filter (fdf.hpp:40-89) over the fragments.
// Per frame: the foreground mask (fde::mask, non-zero = foreground), number of contours kept
// by fde::extractor::extract, index of its fragment.  Per filtered fragment: dims and a
// checksum of its canvas; the dots of fragment `want` are copied to dots_out (if it fits).
uint32_t ref_fdf(const uint8_t* frames, uint32_t n, uint32_t W, uint32_t H, uint8_t* masks, uint32_t* ncontours,
                 uint32_t* frag_of, uint32_t* frag_dims, uint64_t* canvas_sum, uint32_t want, uint16_t* dots_out,
                 uint64_t dots_cap) {
  frc::collector col{mrl::dimensions_t{W, H}};
  mem_feed feed{frames, n, W, H};
  auto cb0 = [](fgm::fragment const&, frc::frame_type const&, frc::image_type const&, frc::grid_type const&) {};
  col.collect(feed, raw_codec{}, cb0);
  auto list = col.complete();
  std::vector<fgm::fragment> frags;
  for (auto& f : list) frags.push_back(std::move(f));
  auto cb = [&](fgm::fragment const&, std::size_t idx, sid::nat::dimg_t const&, std::size_t no, sid::nat::dimg_t const&,
                fgm::point_t, fdf::contours_t const& foreground, auto const& mask) {
    frag_of[no] = std::uint32_t(idx);
    ncontours[no] = std::uint32_t(foreground.size());
    for (std::size_t i = 0; i < std::size_t(W) * H; ++i) masks[no * std::size_t(W) * H + i] = mask.data()[i].value;
  };
  auto out = fdf::filter(frags, mrl::dimensions_t{W, H}, raw_codec{}, cb);
  std::uint32_t fi = 0;
  for (auto& fr : out) {
    frag_dims[2 * fi] = std::uint32_t(fr.dimensions().width_);
    frag_dims[2 * fi + 1] = std::uint32_t(fr.dimensions().height_);
    std::uint64_t s = 0;
    for (auto const* d = fr.dots().data(); d < fr.dots().end(); ++d)
      for (int b = 0; b < 16; ++b) s = s * 1000003u + (*d)[b];
    canvas_sum[fi] = s;
    if (fi == want && fr.dots().size() * 16 <= dots_cap)
      std::memcpy(dots_out, fr.dots().data(), fr.dots().size() * sizeof(fgm::dot_type));
    ++fi;
  }
  return fi;
}

}  // extern "C"
