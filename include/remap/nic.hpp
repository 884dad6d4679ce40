// remap/nic.hpp — namespace nic of the reference (nic.hpp:8-166): the frame codec.
//
//   nic::compress(image)            -> icd::compressed_t      (nic.hpp:8-101)
//   nic::decompress(pack, dim)      -> sid::nat::dimg_t       (nic.hpp:103-164)
//   nic::compression                   the functor main.cpp's adapter hands to collect
//                                      (main.cpp:113-125: operator()(image), operator()(pack, dim))
//
// Both directions run on the device (rmx_nic_compress_batch / rmx_nic_decompress_batch); the
// single-image forms are batches of one.  frc::collector recognises nic::compression and
// compresses the images and medians of a whole block of frames in two launches instead of
// calling the functor twice per frame (collector.hpp), which is what keeps the collect loop
// from being paced by the codec.  Streams are byte-identical to the reference's inside its
// domain; a run > 65535 pixels or a literal > 16383 (where the reference corrupts its heap)
// is split into well-formed tokens instead.
#pragma once

#include <cstring>
#include <vector>

#include "core.hpp"
#include "device.hpp"

namespace nic {

namespace detail {

inline rmx_ctx* codec_ctx() { return remap::context_cache::get(16, 8, 1, 1, 0, 64); }

// n images of `pixels` bytes at `d_images` (device) -> n blobs on the host
inline std::vector<icd::compressed_t> compress_device(rmx_ctx* c, std::uint8_t const* d_images, std::uint32_t n,
                                                      std::uint32_t pixels) {
  std::vector<icd::compressed_t> result(n);
  if (n == 0) return result;
  const std::uint32_t stride = rmx_nic_bound(pixels);
  remap::device_buffer blobs{c, std::size_t(n) * stride}, sizes{c, std::size_t(n) * 4}, offs{c, std::size_t(n) * 8};
  remap::check(rmx_nic_compress_batch(c, d_images, n, pixels, blobs.get<std::uint8_t>(), stride,
                                      sizes.get<std::uint32_t>()),
               c, "rmx_nic_compress_batch");
  remap::check(rmx_nic_status(c), c, "rmx_nic_status");
  std::vector<std::uint32_t> h_sizes(n);
  remap::check(rmx_download(c, h_sizes.data(), sizes.get(), std::size_t(n) * 4), c, "rmx_download");
  std::vector<std::uint64_t> h_offs(n);
  std::uint64_t total = 0;
  for (std::uint32_t i = 0; i < n; ++i) {
    h_offs[i] = total;
    total += h_sizes[i];
  }
  remap::device_buffer packed{c, std::size_t(total) + 8};
  remap::check(rmx_upload(c, offs.get(), h_offs.data(), std::size_t(n) * 8), c, "rmx_upload");
  remap::check(rmx_nic_pack(c, blobs.get<std::uint8_t>(), stride, sizes.get<std::uint32_t>(), offs.get<std::uint64_t>(), n,
                            packed.get<std::uint8_t>()),
               c, "rmx_nic_pack");
  std::vector<std::uint8_t> host(total);
  if (total) remap::check(rmx_download(c, host.data(), packed.get(), std::size_t(total)), c, "rmx_download");
  for (std::uint32_t i = 0; i < n; ++i)
    result[i].assign(host.begin() + std::ptrdiff_t(h_offs[i]), host.begin() + std::ptrdiff_t(h_offs[i] + h_sizes[i]));
  return result;
}

}  // namespace detail

template <typename Alloc>
[[nodiscard]] icd::compressed_t compress(sid::nat::aimg_t<Alloc> const& image) {
  rmx_ctx* c = detail::codec_ctx();
  remap::device_buffer d{c, image.size() + 8};
  remap::check(rmx_upload(c, d.get(), image.data(), image.size()), c, "rmx_upload");
  return std::move(detail::compress_device(c, d.get<std::uint8_t>(), 1, std::uint32_t(image.size()))[0]);
}

[[nodiscard]] inline sid::nat::dimg_t decompress(icd::compressed_t const& pack, mrl::dimensions_t const& dim) {
  rmx_ctx* c = detail::codec_ctx();
  sid::nat::dimg_t result{dim};
  const std::uint32_t pixels = std::uint32_t(dim.area()), size = std::uint32_t(pack.size());
  const std::uint64_t off = 0;
  remap::device_buffer d_pack{c, pack.size() + 8}, d_meta{c, 16}, d_img{c, dim.area() + 8};
  remap::check(rmx_upload(c, d_pack.get(), pack.data(), pack.size()), c, "rmx_upload");
  remap::check(rmx_upload(c, d_meta.get(), &off, 8), c, "rmx_upload");
  remap::check(rmx_upload(c, d_meta.get<std::uint8_t>() + 8, &size, 4), c, "rmx_upload");
  remap::check(rmx_nic_decompress_batch(c, d_pack.get<std::uint8_t>(), d_meta.get<std::uint64_t>(),
                                        reinterpret_cast<std::uint32_t const*>(d_meta.get<std::uint8_t>() + 8), 1, pixels,
                                        d_img.get<std::uint8_t>()),
               c, "rmx_nic_decompress_batch");
  remap::check(rmx_nic_status(c), c, "rmx_nic_status");
  remap::check(rmx_download(c, result.data(), d_img.get(), dim.area()), c, "rmx_download");
  return result;
}

// The reference's native_compression (main.cpp:113-125) under the name the collector knows.
struct compression {
  template <typename Alloc>
  [[nodiscard]] icd::compressed_t operator()(sid::nat::aimg_t<Alloc> const& image) const {
    return compress(image);
  }
  [[nodiscard]] sid::nat::dimg_t operator()(icd::compressed_t const& compressed, mrl::dimensions_t const& dim) const {
    return decompress(compressed, dim);
  }
};

}  // namespace nic
