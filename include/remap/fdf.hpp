// remap/fdf.hpp — namespace fdf of the reference (fdf.hpp:12-89): foreground filtering.
//
//   fdf::filter(fragments, frame_dim, comp, cb)  -> std::vector<fgm::fragment>
//
// For every fragment the background is the blend of its canvas (fdf.hpp:22-36); every frame of
// the fragment is decoded with `comp`, its foreground contours are extracted from the median
// (fde.hpp:95-120 over cte.hpp:64-166), masked (fde.hpp:125-146) and the frame is blitted
// again without the masked pixels (fgm.hpp:71-86).  Here one fragment is one
// rmx_fdf_filter_batch call: background, frames, medians, masks and the filtered canvas stay in
// HBM; a nic::compression codec decodes all blobs of the fragment in one device batch, any
// other decompressor is called per frame on the host and its output uploaded.
// The callback receives what the reference's does.  `foreground` has one entry per kept
// contour; an entry carries its index only — areas, edges and enclosures stay on the device
// (main.cpp's fdf_callback, main.cpp:150-163, counts calls and reads none of them).
#pragma once

#include <type_traits>
#include <vector>

#include "fragment.hpp"
#include "nic.hpp"

namespace fdf {

struct background {
  fgm::point_t zero_;
  sid::nat::dimg_t image_;
};

struct contour {
  std::uint32_t id_;
  [[nodiscard]] std::uint32_t id() const noexcept { return id_; }
};
using contours_t = std::vector<contour>;

template <typename Comp, typename Callback>
[[nodiscard]] std::vector<fgm::fragment> filter(std::vector<fgm::fragment> const& fragments,
                                                mrl::dimensions_t const& frame_dim, Comp&& comp, Callback&& cb)
  requires(icd::decompressor<std::decay_t<Comp>, std::allocator<cpl::nat_cc>>)
{
  std::vector<fgm::fragment> results{};
  const std::size_t area = frame_dim.area();
  rmx_ctx* c = remap::context_cache::get(std::uint32_t(frame_dim.width_), std::uint32_t(frame_dim.height_), 4, 2, 16, 64);
  std::size_t index = 0;
  for (auto const& fragment : fragments) {
    const auto bdim = fragment.dimensions();
    auto& result = results.emplace_back(bdim, fragment.zero());
    const std::uint32_t n = std::uint32_t(fragment.frames().size());
    if (n == 0 || !fragment.device_canvas()) {
      ++index;
      continue;
    }
    // background = blend of the canvas, on the device (details::get_background)
    remap::device_buffer bg{c, bdim.area()}, bgmask{c, bdim.area()};
    remap::check(rmx_fgm_blend(c, fragment.device_canvas(), std::uint32_t(bdim.width_), std::uint32_t(bdim.height_),
                               bg.get<std::uint8_t>(), bgmask.get<std::uint8_t>()),
                 c, "rmx_fgm_blend");
    // frames and medians of the fragment, decoded into HBM
    remap::device_buffer d_img{c, n * area}, d_med{c, n * area}, d_mask{c, n * area}, d_pos{c, n * 8 + 8}, d_cnt{c, n * 4 + 8};
    std::vector<std::uint8_t> h_img(n * area), h_med(n * area);
    if constexpr (std::is_same_v<std::decay_t<Comp>, nic::compression>) {
      for (int plane = 0; plane < 2; ++plane) {
        std::vector<std::uint8_t> packed;
        std::vector<std::uint64_t> offs(n);
        std::vector<std::uint32_t> sizes(n);
        for (std::uint32_t i = 0; i < n; ++i) {
          auto const& blob = plane ? fragment.frames()[i].data_.median_ : fragment.frames()[i].data_.image_;
          offs[i] = packed.size();
          sizes[i] = std::uint32_t(blob.size());
          packed.insert(packed.end(), blob.begin(), blob.end());
        }
        remap::device_buffer d_pack{c, packed.size() + 8}, d_off{c, n * 8}, d_sz{c, n * 4};
        remap::check(rmx_upload(c, d_pack.get(), packed.data(), packed.size()), c, "rmx_upload");
        remap::check(rmx_upload(c, d_off.get(), offs.data(), n * 8), c, "rmx_upload");
        remap::check(rmx_upload(c, d_sz.get(), sizes.data(), n * 4), c, "rmx_upload");
        auto& dst = plane ? d_med : d_img;
        remap::check(rmx_nic_decompress_batch(c, d_pack.get<std::uint8_t>(), d_off.get<std::uint64_t>(),
                                              d_sz.get<std::uint32_t>(), n, std::uint32_t(area), dst.get<std::uint8_t>()),
                     c, "rmx_nic_decompress_batch");
        remap::check(rmx_nic_status(c), c, "rmx_nic_status");
        remap::check(rmx_download(c, (plane ? h_med : h_img).data(), dst.get(), n * area), c, "rmx_download");
      }
    } else {
      for (std::uint32_t i = 0; i < n; ++i) {
        auto const& data = fragment.frames()[i].data_;
        auto image = comp(data.image_, frame_dim);
        auto median = comp(data.median_, frame_dim);
        std::memcpy(h_img.data() + i * area, image.data(), area);
        std::memcpy(h_med.data() + i * area, median.data(), area);
      }
      remap::check(rmx_upload(c, d_img.get(), h_img.data(), n * area), c, "rmx_upload");
      remap::check(rmx_upload(c, d_med.get(), h_med.data(), n * area), c, "rmx_upload");
    }
    std::vector<std::int32_t> pos(2 * n);
    for (std::uint32_t i = 0; i < n; ++i) {
      auto p = fragment.frames()[i].position_ - result.zero();
      pos[2 * i] = p.x_;
      pos[2 * i + 1] = p.y_;
    }
    remap::check(rmx_upload(c, d_pos.get(), pos.data(), n * 8), c, "rmx_upload");
    remap::check(rmx_fdf_filter_batch(c, bg.get<std::uint8_t>(), std::uint32_t(bdim.width_), std::uint32_t(bdim.height_),
                                      d_img.get<std::uint8_t>(), d_med.get<std::uint8_t>(), d_pos.get<std::int32_t>(), n,
                                      d_mask.get<std::uint8_t>(), d_cnt.get<std::uint32_t>(), result.device_canvas_ensure()),
                 c, "rmx_fdf_filter_batch");
    remap::check(rmx_fdf_status(c), c, "rmx_fdf_status");
    std::vector<std::uint8_t> h_mask(n * area);
    std::vector<std::uint32_t> h_cnt(n);
    remap::check(rmx_download(c, h_mask.data(), d_mask.get(), n * area), c, "rmx_download");
    remap::check(rmx_download(c, h_cnt.data(), d_cnt.get(), n * 4), c, "rmx_download");
    // bookkeeping + callback per frame, in the fragment's frame order (fdf.hpp:60-72)
    for (std::uint32_t i = 0; i < n; ++i) {
      auto const& fr = fragment.frames()[i];
      result.add_frame(fr.number_, fr.position_);
      sid::nat::dimg_t image{frame_dim}, median{frame_dim};
      sid::mon::dimg_t mask{frame_dim};
      std::memcpy(image.data(), h_img.data() + i * area, area);
      std::memcpy(median.data(), h_med.data() + i * area, area);
      std::memcpy(mask.data(), h_mask.data() + i * area, area);
      contours_t foreground(h_cnt[i]);
      for (std::uint32_t k = 0; k < h_cnt[i]; ++k) foreground[k].id_ = k + 1;
      cb(result, index, image, fr.number_, median, fr.position_, foreground, mask);
    }
    ++index;
  }
  return results;
}

}  // namespace fdf
