// remap/fragment.hpp — namespace fgm of the reference (fgm.hpp): the fragment canvas, a
// 16-bin uint16 colour histogram per pixel, here resident in HBM.
//
// Every heavy member forwards to the C ABI: blit -> rmx_fgm_blit_batch, blit(fragment&&)
// and canvas growth -> rmx_fgm_merge, blend -> rmx_fgm_blend.  The host keeps only the
// metadata the reference keeps (step, zero, frame list); dots() downloads on demand.
#pragma once

#include <algorithm>
#include <cstring>
#include <vector>

#include "core.hpp"
#include "device.hpp"

namespace frc { class collector; }

namespace fgm {

inline constexpr std::uint8_t depth{16};
using dot_type = std::array<std::uint16_t, depth>;
inline bool is_empty(dot_type const& d) noexcept {
  return std::all_of(d.begin(), d.end(), [](auto v) { return v == 0; });
}
using point_t = cdt::point<std::int32_t>;

struct fragment_blend {
  sid::nat::dimg_t image_;
  sid::mon::dimg_t mask_;
};
struct packed_data {
  icd::compressed_t image_;
  icd::compressed_t median_;
};
struct frame {
  std::size_t number_;
  point_t position_;
  packed_data data_;
};

class fragment {
public:
  using matrix_type = mrl::matrix<dot_type>;

  fragment() noexcept : step_{1, 1}, dim_{1, 1} {}
  explicit fragment(mrl::dimensions_t const& step) noexcept : step_{step}, dim_{step} {}
  fragment(matrix_type dots, mrl::dimensions_t const& step, point_t const& zero, std::vector<fgm::frame>&& frames)
      : step_{step}, dim_{dots.dimensions()}, zero_{zero}, frames_{std::move(frames)} {
    upload(dots);
  }
  fragment(mrl::dimensions_t const& dimensions, point_t const& zero) noexcept : step_{1, 1}, dim_{dimensions}, zero_{zero} {}
  fragment(fragment&&) noexcept = default;
  fragment& operator=(fragment&&) noexcept = default;

  // fgm.hpp:87-97 — accumulate one frame at `pos` (fragment coordinates)
  template <typename Alloc>
  void blit(point_t pos, sid::nat::aimg_t<Alloc> const& image, packed_data&& packed, std::size_t frame_no) {
    ensure(pos, image.dimensions());
    rmx_ctx* c = frame_ctx(image.dimensions());
    stage_.reserve(c, image.size() + 8);
    remap::check(rmx_upload(c, stage_.get(), image.data(), image.size()), c, "rmx_upload");
    std::int32_t p[2] = {pos.x_, pos.y_};
    remap::check(rmx_upload(c, stage_.get<std::uint8_t>() + image.size(), p, 8), c, "rmx_upload");
    remap::check(rmx_fgm_blit_batch(c, stage_.get<std::uint8_t>(),
                                    reinterpret_cast<std::int32_t const*>(stage_.get<std::uint8_t>() + image.size()),
                                    nullptr, 0, 1, canvas_.get<std::uint16_t>(), std::uint32_t(dim_.width_),
                                    std::uint32_t(dim_.height_), zero_.x_, zero_.y_),
                 c, "rmx_fgm_blit_batch");
    remap::check(rmx_sync(c), c, "rmx_sync");  // the staging buffer is reused by the next call
    frames_.emplace_back(fgm::frame{frame_no, pos, std::move(packed)});
  }

  // fgm.hpp:99-113 — add another fragment's canvas at `pos`, adopt its frames
  void blit(point_t pos, fragment&& other) {
    ensure(pos, other.dim_);
    if (other.canvas_) {
      rmx_ctx* c = any_ctx();
      remap::check(rmx_fgm_merge(c, canvas_.get<std::uint16_t>(), std::uint32_t(dim_.width_),
                                 std::uint32_t(dim_.height_), other.canvas_.get<std::uint16_t>(),
                                 std::uint32_t(other.dim_.width_), std::uint32_t(other.dim_.height_),
                                 pos.x_ - zero_.x_, pos.y_ - zero_.y_),
                   c, "rmx_fgm_merge");
      remap::check(rmx_sync(c), c, "rmx_sync");
    }
    frames_.reserve(frames_.size() + other.frames_.size());
    for (auto& f : other.frames_)
      frames_.emplace_back(fgm::frame{f.number_, f.position_ - other.zero_ + pos, std::move(f.data_)});
  }

  // fgm.hpp:115-135 — per pixel: first arg-max bin, mask = any bin non-zero
  fragment_blend blend() const {
    fragment_blend out{sid::nat::dimg_t{dim_}, sid::mon::dimg_t{dim_}};
    if (!canvas_) return out;
    rmx_ctx* c = any_ctx();
    remap::device_buffer img{c, dim_.area()}, msk{c, dim_.area()};
    remap::check(rmx_fgm_blend(c, canvas_.get<std::uint16_t>(), std::uint32_t(dim_.width_), std::uint32_t(dim_.height_),
                               img.get<std::uint8_t>(), msk.get<std::uint8_t>()),
                 c, "rmx_fgm_blend");
    remap::check(rmx_download(c, out.image_.data(), img.get(), dim_.area()), c, "rmx_download");
    remap::check(rmx_download(c, out.mask_.data(), msk.get(), dim_.area()), c, "rmx_download");
    return out;
  }

  void normalize() noexcept {
    for (auto& f : frames_) f.position_ -= zero_;
    zero_ = {0, 0};
  }

  // fgm.hpp:145-153,235-295 — the empty border of the canvas: left_ / right_ = the fewest empty dots before
  // the first / after the last non-empty dot of any row, top_ / bottom_ the same per column; a side keeps
  // the canvas width (height) if no row (column) has a dot.  "Non-empty" is the mask rmx_fgm_blend
  // computes (any bin != 0), so one byte per dot comes to the host instead of thirty-two.
  mrl::region_t margins() const {
    const std::size_t w = dim_.width_, h = dim_.height_;
    mrl::region_t region{w, h, w, h};
    if (!canvas_) return region;
    rmx_ctx* c = any_ctx();
    remap::device_buffer img{c, dim_.area()}, msk{c, dim_.area()};
    remap::check(rmx_fgm_blend(c, canvas_.get<std::uint16_t>(), std::uint32_t(w), std::uint32_t(h), img.get<std::uint8_t>(),
                               msk.get<std::uint8_t>()),
                 c, "rmx_fgm_blend");
    std::vector<std::uint8_t> mask(dim_.area());
    remap::check(rmx_download(c, mask.data(), msk.get(), mask.size()), c, "rmx_download");
    std::vector<std::size_t> first_y(w, h), last_y(w, h);  // per column: first / last non-empty row (h: none)
    for (std::size_t y = 0; y < h; ++y) {
      std::uint8_t const* row = mask.data() + y * w;
      std::size_t lo = w, hi = w;
      for (std::size_t x = 0; x < w; ++x)
        if (row[x]) {
          if (lo == w) lo = x;
          hi = x;
          if (first_y[x] == h) first_y[x] = y;
          last_y[x] = y;
        }
      if (lo != w) {
        region.left_ = std::min(region.left_, lo);
        region.right_ = std::min(region.right_, w - 1 - hi);
      }
    }
    for (std::size_t x = 0; x < w; ++x)
      if (first_y[x] != h) {
        region.top_ = std::min(region.top_, first_y[x]);
        region.bottom_ = std::min(region.bottom_, h - 1 - last_y[x]);
      }
    return region;
  }

  // canvas as a host matrix (downloads; an untouched fragment is all zero)
  matrix_type dots() const {
    matrix_type m{dim_};
    if (canvas_) {
      rmx_ctx* c = any_ctx();
      remap::check(rmx_download(c, m.data(), canvas_.get(), dim_.area() * sizeof(dot_type)), c, "rmx_download");
    }
    return m;
  }
  mrl::dimensions_t dimensions() const noexcept { return dim_; }
  mrl::dimensions_t step() const noexcept { return step_; }
  point_t zero() const noexcept { return zero_; }
  std::vector<frame> const& frames() const noexcept { return frames_; }

  // device view for callers that stay on the GPU (frc::collector, fgs)
  std::uint16_t* device_canvas() const noexcept { return canvas_.get<std::uint16_t>(); }
  // for device-side batch blits (fdf::filter): the canvas, allocated and zeroed if it is not
  // there yet, and the bookkeeping half of fragment::blit(pos, image, mask, no) (fgm.hpp:85)
  std::uint16_t* device_canvas_ensure() {
    if (!canvas_) {
      rmx_ctx* c = any_ctx();
      canvas_ = remap::device_buffer{c, dim_.area() * sizeof(dot_type)};
      remap::check(rmx_memset(c, canvas_.get(), 0, dim_.area() * sizeof(dot_type)), c, "rmx_memset");
      remap::check(rmx_sync(c), c, "rmx_sync");
    }
    return canvas_.get<std::uint16_t>();
  }
  void add_frame(std::size_t frame_no, point_t pos) { frames_.emplace_back(fgm::frame{frame_no, pos, {}}); }

private:
  friend class frc::collector;

  static rmx_ctx* any_ctx() { return remap::context_cache::get(16, 8, 1, 1, 0, 64); }
  static rmx_ctx* frame_ctx(mrl::dimensions_t const& d) {
    return remap::context_cache::get(std::uint32_t(d.width_), std::uint32_t(d.height_), 1, 1, 0, 64);
  }
  void upload(matrix_type const& m) {
    rmx_ctx* c = any_ctx();
    canvas_ = remap::device_buffer{c, m.size() * sizeof(dot_type)};
    remap::check(rmx_upload(c, canvas_.get(), m.data(), m.size() * sizeof(dot_type)), c, "rmx_upload");
    remap::check(rmx_sync(c), c, "rmx_sync");
  }
  static std::size_t round_up(std::size_t change, std::size_t step) noexcept { return (change + step - 1) / step * step; }

  // fgm.hpp:190-233 — grow the canvas in whole steps so that [pos, pos + dim) fits
  void ensure(point_t pos, mrl::dimensions_t const& dim) {
    const std::int32_t xy[2] = {pos.x_, pos.y_};
    ensure_run(xy, 1, dim);
  }
  // ... for a run of blits at once (xy: count pairs of x, y): every side grows by whole steps from where it is, so
  // the box after the run does not depend on the order of its blits and ONE growth to the run's extremes is what
  // `count` growths one after the other end with (a scroller's canvas grew 25 times per 4 096-frame block: an
  // allocation, a memset, a merge kernel and a wait each)
  void ensure_run(std::int32_t const* xy, std::size_t count, mrl::dimensions_t const& dim) {
    std::int64_t l = 0, t = 0, r = 0, b = 0;
    const std::int64_t zx = zero_.x_, zy = zero_.y_, w = std::int64_t(dim_.width_), h = std::int64_t(dim_.height_);
    for (std::size_t i = 0; i < count; ++i) {
      const std::int64_t px = xy[2 * i], py = xy[2 * i + 1];
      if (px < zx) l = std::max(l, std::int64_t(round_up(std::size_t(zx - px), step_.width_)));
      if (py < zy) t = std::max(t, std::int64_t(round_up(std::size_t(zy - py), step_.height_)));
      const std::int64_t rx = px + std::int64_t(dim.width_), ry = py + std::int64_t(dim.height_);
      if (rx > 0 && rx > zx + w) r = std::max(r, std::int64_t(round_up(std::size_t(rx - (zx + w)), step_.width_)));
      if (ry > 0 && ry > zy + h) b = std::max(b, std::int64_t(round_up(std::size_t(ry - (zy + h)), step_.height_)));
    }
    const bool grow = l || t || r || b;
    if (!grow && canvas_) return;
    rmx_ctx* c = any_ctx();
    mrl::dimensions_t nd{std::size_t(w + l + r), std::size_t(h + t + b)};
    remap::device_buffer next{c, nd.area() * sizeof(dot_type)};
    remap::check(rmx_memset(c, next.get(), 0, nd.area() * sizeof(dot_type)), c, "rmx_memset");
    if (canvas_)
      remap::check(rmx_fgm_merge(c, next.get<std::uint16_t>(), std::uint32_t(nd.width_), std::uint32_t(nd.height_),
                                 canvas_.get<std::uint16_t>(), std::uint32_t(dim_.width_), std::uint32_t(dim_.height_),
                                 std::int32_t(l), std::int32_t(t)),
                   c, "rmx_fgm_merge");
    remap::check(rmx_sync(c), c, "rmx_sync");
    canvas_ = std::move(next);
    dim_ = nd;
    zero_.x_ -= std::int32_t(l);
    zero_.y_ -= std::int32_t(t);
  }

  mrl::dimensions_t step_;
  mrl::dimensions_t dim_;
  point_t zero_{};
  std::vector<frame> frames_;
  remap::device_buffer canvas_;  // dim_.area() dots; allocated on first blit
  remap::device_buffer stage_;   // one frame + its position, for single-frame blits
};

}  // namespace fgm
