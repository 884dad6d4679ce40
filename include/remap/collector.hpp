// remap/collector.hpp — namespaces kpe, kpm and frc of the reference on top of the C ABI.
//
//   kpe::extractor<Grid, Overlap>::extract      -> rmx_kpe_batch            (kpe.hpp:69-108)
//   kpm::match(cfg, previous, current)          -> rmx_kpm_seq              (kpm.hpp:395-415)
//   frc::collector::collect / complete / current                             (frc.hpp:28-144)
//
// frc::collector is the hot path.  The reference's loop handles one frame at a time
// (extract -> match -> position -> blit -> callback).  Here the feed is drained in blocks of
// `block_frames` frames: a block is staged into HBM, extracted, matched (the previous
// block's last table is the halo), scanned and blitted on the device; the callbacks are
// then replayed on the calling thread, once per frame, in frame order, as frc.hpp:119
// promises.  Nothing is computed on the host.
#pragma once

#include <exception>
#include <list>
#include <optional>
#include <future>
#include <thread>
#include <type_traits>
#include <utility>
#include <vector>

#include "core.hpp"
#include "device.hpp"
#include "fragment.hpp"
#include "grid.hpp"
#include "nic.hpp"

namespace remap::detail {

// rmx_desc table (scan order) -> kpr::grid: one add per set region bit, ascending region
// index — the insertion sequence kpe::extractor produces (kpe.hpp:201-205,289; kpr.hpp:189-219).
template <typename Grid, std::size_t... I>
inline void add_row(Grid& g, rmx_desc const& d, std::index_sequence<I...>) {
  kpr::code c;
  std::memcpy(c.data(), d.code, kpr::code_length);
  mrl::point_t p{d.x, d.y};
  ((d.region_mask >> I & 1 ? g.add(c, p, std::index_sequence<I>{}) : void()), ...);
}
template <typename Grid>
inline void fill_grid(Grid& g, void const* rows, std::uint32_t n) {
  auto const* r = static_cast<rmx_desc const*>(rows);
  for (std::uint32_t i = 0; i < n; ++i) add_row(g, r[i], std::make_index_sequence<Grid::region_count>{});
}
template <typename Grid>
inline Grid table_to_grid(rmx_desc const* rows, std::uint32_t n, typename Grid::allocator_type const& a) {
  Grid g{a};
  fill_grid(g, rows, n);
  return g;
}
// ... the same, built when a region is first looked at (rows must stay valid until then)
template <typename Grid>
inline Grid table_to_grid_deferred(rmx_desc const* rows, std::uint32_t n, typename Grid::allocator_type const& a) {
  return Grid{a, &fill_grid<Grid>, rows, n};
}

// kpr::grid -> rmx_desc table in scan order (x, then y) with region masks.
template <typename Grid>
inline std::vector<rmx_desc> grid_to_table(Grid const& g) {
  std::vector<rmx_desc> rows;
  for (std::size_t r = 0; r < Grid::region_count; ++r)
    for (auto const& [code, pts] : g[r].points())
      for (auto const& p : pts) {
        rmx_desc d{};
        std::memcpy(d.code, code.data(), kpr::code_length);
        d.region_mask = std::uint8_t(1u << r);
        d.x = std::uint16_t(p.x_);
        d.y = std::uint16_t(p.y_);
        rows.push_back(d);
      }
  std::sort(rows.begin(), rows.end(), [](rmx_desc const& a, rmx_desc const& b) {
    return a.x != b.x ? a.x < b.x : a.y < b.y;
  });
  std::size_t w = 0;
  for (std::size_t i = 0; i < rows.size(); ++i) {
    if (w && rows[w - 1].x == rows[i].x && rows[w - 1].y == rows[i].y)
      rows[w - 1].region_mask |= rows[i].region_mask;
    else
      rows[w++] = rows[i];
  }
  rows.resize(w);
  return rows;
}

}  // namespace remap::detail

namespace kpe {

using ksize_t = std::uint8_t;
inline constexpr ksize_t kernel_size{5};
inline constexpr ksize_t kernel_half{kernel_size / 2};

template <kpr::gridlike Grid, std::size_t Overlap>
class extractor {
public:
  using grid_type = Grid;
  using matrix_type = sid::nat::aimg_t<all::rebind_alloc_t<typename grid_type::allocator_type, cpl::nat_cc>>;

  explicit extractor(mrl::dimensions_t const& d, std::uint32_t kmax = 16384)
      : dim_{d}, kmax_{kmax},
        ctx_{remap::context_cache::get(std::uint32_t(d.width_), std::uint32_t(d.height_), std::uint32_t(Grid::width),
                                       std::uint32_t(Grid::height), std::uint32_t(Overlap), kmax)} {}

  // kpe.hpp:92-108: keypoint grid of `image`; `median` receives the 3x3 rank image.  An image
  // with more keypoints than the table holds is extracted again with a larger one (the
  // reference has no limit).
  grid_type extract(matrix_type const& image, matrix_type& median, typename grid_type::allocator_type const& alloc) {
    const std::size_t area = dim_.area();
    for (;;) {
      buf_.reserve(ctx_, 2 * area + std::size_t(kmax_) * sizeof(rmx_desc) + 16);
      auto* d_img = buf_.template get<std::uint8_t>();
      auto* d_med = d_img + area;
      auto* d_cnt = reinterpret_cast<std::uint32_t*>(d_med + area);
      auto* d_tab = reinterpret_cast<rmx_desc*>(d_med + area + 16);
      remap::check(rmx_upload(ctx_, d_img, image.data(), area), ctx_, "rmx_upload");
      remap::check(rmx_kpe_batch(ctx_, d_img, 1, d_med, d_tab, d_cnt), ctx_, "rmx_kpe_batch");
      std::uint32_t maxk = 0;
      const int st = rmx_kpe_status(ctx_, &maxk);
      if (st == RMX_ERR_KMAX && maxk <= (1u << 22)) {
        while (kmax_ < maxk) kmax_ <<= 1;
        ctx_ = remap::context_cache::get(std::uint32_t(dim_.width_), std::uint32_t(dim_.height_), std::uint32_t(Grid::width),
                                         std::uint32_t(Grid::height), std::uint32_t(Overlap), kmax_);
        continue;
      }
      remap::check(st, ctx_, "rmx_kpe_status");
      std::uint32_t k = 0;
      remap::check(rmx_download(ctx_, &k, d_cnt, 4), ctx_, "rmx_download");
      rows_.resize(k);
      if (k) remap::check(rmx_download(ctx_, rows_.data(), d_tab, k * sizeof(rmx_desc)), ctx_, "rmx_download");
      remap::check(rmx_download(ctx_, median.data(), d_med, area), ctx_, "rmx_download");
      return remap::detail::table_to_grid<grid_type>(rows_.data(), k, alloc);
    }
  }

private:
  mrl::dimensions_t dim_;
  std::uint32_t kmax_;
  rmx_ctx* ctx_;
  remap::device_buffer buf_;
  std::vector<rmx_desc> rows_;
};

}  // namespace kpe

namespace kpm {

template <typename T>
concept match_config = requires(T cfg) {
  requires std::unsigned_integral<decltype(T::weight_switch)>;
  requires std::unsigned_integral<decltype(T::region_votes)>;
  requires T::region_votes >= 1;
  typename T::allocator_type;
  { cfg.get_allocator() } -> std::convertible_to<typename T::allocator_type>;
};

struct vote {
  using pair_t = std::pair<cdt::offset_t const, std::size_t>;
  vote() noexcept = default;
  vote(cdt::offset_t const& o, std::size_t c) noexcept : offset_{o}, count_{c} {}
  explicit vote(pair_t const& p) noexcept : offset_{p.first}, count_{p.second} {}
  vote reverse() const noexcept { return {-offset_, count_}; }
  cdt::offset_t offset_{};
  std::size_t count_{};
};

using cell_size_t = cdt::dimensions<std::uint8_t>;

// Grid/Borda variant (kpm.hpp:395-415) for two host grids: both tables are uploaded as a
// two-frame batch and matched on the device.  (frc::collector does not go through this
// per-pair entry point; it matches whole blocks.)
template <match_config Cfg, typename Grid>
std::optional<cdt::offset_t> match(Cfg const&, Grid const& previous, Grid const& current) {
  static_assert(Grid::region_count == 8 && Cfg::region_votes == 3 && Cfg::weight_switch == 10,
                "the device matcher implements frc::collector's configuration (grid 4x2, votes 3, switch 10)");
  auto tp = remap::detail::grid_to_table(previous), tc = remap::detail::grid_to_table(current);
  std::uint32_t kmax = 2048;  // table stride: the next power of two that holds both, up to the matcher's 16 384
  while (kmax < std::max(tp.size(), tc.size()) && kmax < 16384) kmax <<= 1;
  if (tp.size() > kmax || tc.size() > kmax) throw remap::device_error(RMX_ERR_KMAX, "kpm::match: more than 16384 keypoints");
  rmx_ctx* c = remap::context_cache::get(4096, 4096, 4, 2, 16, kmax);
  remap::device_buffer buf{c, 2 * kmax * sizeof(rmx_desc) + 64};
  auto* d_tab = buf.get<rmx_desc>();
  auto* d_cnt = reinterpret_cast<std::uint32_t*>(buf.get<std::uint8_t>() + 2 * kmax * sizeof(rmx_desc));
  auto* d_dxdy = reinterpret_cast<std::int32_t*>(d_cnt + 2);
  auto* d_m = reinterpret_cast<std::uint8_t*>(d_dxdy + 4);
  std::uint32_t cnt[2] = {std::uint32_t(tp.size()), std::uint32_t(tc.size())};
  if (!tp.empty()) remap::check(rmx_upload(c, d_tab, tp.data(), tp.size() * sizeof(rmx_desc)), c, "rmx_upload");
  if (!tc.empty()) remap::check(rmx_upload(c, d_tab + kmax, tc.data(), tc.size() * sizeof(rmx_desc)), c, "rmx_upload");
  remap::check(rmx_upload(c, d_cnt, cnt, 8), c, "rmx_upload");
  remap::check(rmx_kpm_seq(c, d_tab, d_cnt, 2, nullptr, nullptr, d_dxdy, d_m), c, "rmx_kpm_seq");
  std::uint32_t flags = 0;
  int st = rmx_kpm_status(c, &flags);
  if (st == RMX_ERR_CAPACITY && flags == 128u)  // a region beyond the LDS matcher: its global-memory instance
    st = rmx_kpm_resolve(c, d_tab, d_cnt, 2, nullptr, nullptr, d_dxdy, d_m, nullptr);
  remap::check(st, c, "rmx_kpm_status");
  std::int32_t dxdy[4];
  std::uint8_t m[2];
  remap::check(rmx_download(c, dxdy, d_dxdy, 16), c, "rmx_download");
  remap::check(rmx_download(c, m, d_m, 2), c, "rmx_download");
  if (!m[1]) return {};
  return cdt::offset_t{dxdy[2], dxdy[3]};
}

}  // namespace kpm

namespace frc {

template <typename T>
using allocator_t = all::frame_allocator<T>;
using image_type = sid::nat::aimg_t<allocator_t<cpl::nat_cc>>;
using frame_type = ifd::frame<image_type>;

inline constexpr std::size_t grid_horizontal{4};
inline constexpr std::size_t grid_vertical{2};
inline constexpr std::size_t grid_overlap{16};
using grid_type = kpr::grid<grid_horizontal, grid_vertical, allocator_t<char>>;

class collector {
  using pixel_alloc_t = allocator_t<cpl::nat_cc>;

public:
  // frames staged per device block (a block of 320x240 frames is 75 KB/frame of HBM x3)
  static inline constexpr std::uint32_t block_frames = 4096;
  // rows per descriptor table to start with; a frame with more keypoints makes the collector
  // re-run its block with larger tables (the reference has no limit, kpr.hpp:121-124) — a frame
  // cannot have more keypoints than pixels
  static inline constexpr std::uint32_t kmax = 2048;
  static inline constexpr std::uint32_t max_kmax = 1u << 22;

  collector(mrl::dimensions_t d)
      : dim_{d}, ctx_{remap::context_cache::get(std::uint32_t(d.width_), std::uint32_t(d.height_), 4, 2, 16, kmax)} {}

  // The callback of frc.hpp:119 borrows the current fragment.  By default a run of frames of one
  // fragment is blitted in one launch BEFORE its callbacks are replayed, so a callback that
  // reads the fragment's canvas (blend(), dots(), margins()) sees the frames of the whole run.
  // exact_callbacks(true) blits and waits frame by frame: the callback of frame t then sees the
  // canvas exactly after frame t, as the reference's does (frc.hpp:117-119) — at per-frame
  // launch cost.  main.cpp's callback ignores the fragment (main.cpp:139-149).
  collector& exact_callbacks(bool on) noexcept {
    exact_callbacks_ = on;
    return *this;
  }
  // The next block of the feed is produced by a helper thread while the current one is on the device and in its
  // callbacks.  Off by default: the feeder then runs concurrently with the callbacks (never with itself), which
  // the reference's single loop never does — for feeders that share no unsynchronised state with them
  // (main.cpp's file_feed and counting callback qualify).
  // frames per block (default block_frames; smaller blocks: less host and device memory, more launches)
  collector& block_size(std::uint32_t frames) noexcept {
    block_size_ = std::max<std::uint32_t>(1, frames);
    return *this;
  }
  collector& prefetch_feed(bool on) noexcept {
    prefetch_feed_ = on;
    return *this;
  }

  template <typename Feeder, typename Comp, typename Callback>
  void collect(Feeder&& feed, Comp&& comp, Callback&& cb)
    requires(ifd::feeder<std::decay_t<Feeder>, pixel_alloc_t> && icd::compressor<std::decay_t<Comp>, pixel_alloc_t>)
  {
    const std::size_t area = dim_.area();
    bool first = true;
    bool fresh = true;  // process_init always opens a fragment (frc.hpp:83-95), also on a second collect()
    std::int32_t carry_pos[2] = {0, 0};
    std::uint32_t carry_frag = 0;
    remap::detail::phase_clock clk;
    // ---- the feed is drained in blocks into one of two host arenas (reused block after block: their pages are
    // touched once).  With prefetch_feed(true) the NEXT block is produced by a helper thread while this one is
    // on the device and in its callbacks; otherwise after them, on this thread (the reference's order of calls
    // per block) ----
    std::vector<frame_type> block, next_block;
    unsigned cur = 0;
    auto fill = [&](std::vector<frame_type>& b, all::memory_pool& pool) {
      b.clear();
      pool.reset();
      while (feed.has_more() && b.size() < block_size_) b.push_back(feed.produce(pixel_alloc_t{pool}));
    };
    std::future<void> pending;  // (declared after the blocks: joined before they go)
    clk.mark(nullptr);
    fill(block, feed_pool_[0]);
    clk.mark("feed.produce");
    while (!block.empty()) {
      clk.mark(nullptr);
      if (prefetch_feed_) pending = std::async(std::launch::async, [&, nxt = cur ^ 1u] { fill(next_block, feed_pool_[nxt]); });
      const std::uint32_t n = std::uint32_t(block.size());
      std::uint32_t maxk = 0;
      const std::int32_t carry_pos0[2] = {carry_pos[0], carry_pos[1]};
      const std::uint32_t carry_frag0 = carry_frag;
      for (;;) {  // one pass, unless a frame has more keypoints than the tables hold
      reserve(n);
      // frames are gathered in page-locked memory and cross PCIe in one copy
      stage_.reserve(ctx_, std::size_t(n) * area);
      {
        std::uint8_t* stage = stage_.template get<std::uint8_t>();  // (thread_local: this thread's buffer)
        const unsigned gatherers = std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 8u);
        auto gather = [&, stage](unsigned w) {
          for (std::uint32_t i = w; i < n; i += gatherers) std::memcpy(stage + i * area, std::as_const(block[i].image_).data(), area);  // (const: a view stays a view)
        };
        std::vector<std::thread> helpers;
        for (unsigned w = 1; w < gatherers; ++w) helpers.emplace_back(gather, w);
        gather(0);
        for (auto& th : helpers) th.join();
      }
      remap::check(rmx_upload(ctx_, frames_.get<std::uint8_t>(), stage_.get(), std::size_t(n) * area), ctx_, "rmx_upload");
      clk.mark("stage + upload");
      // ---- the hot path, all on the device ----
      remap::check(rmx_kpe_batch(ctx_, frames_.get<std::uint8_t>(), n, median_.get<std::uint8_t>(),
                                 desc_.get<rmx_desc>(), kcount_.get<std::uint32_t>()),
                   ctx_, "rmx_kpe_batch");
      remap::check(rmx_kpm_seq(ctx_, desc_.get<rmx_desc>(), kcount_.get<std::uint32_t>(), n,
                               first ? nullptr : halo_.get<rmx_desc>(),
                               first ? nullptr : reinterpret_cast<std::uint32_t const*>(
                                                     halo_.get<std::uint8_t>() + kmax_ * sizeof(rmx_desc)),
                               dxdy_.get<std::int32_t>(), matched_.get<std::uint8_t>()),
                   ctx_, "rmx_kpm_seq");
      remap::check(rmx_frc_scan(ctx_, dxdy_.get<std::int32_t>(), matched_.get<std::uint8_t>(), n, first ? 1 : 0,
                                carry_pos, &carry_frag, pos_.get<std::int32_t>(), frag_.get<std::uint32_t>()),
                   ctx_, "rmx_frc_scan");
      const int kst = rmx_kpe_status(ctx_, &maxk);
      if (kst == RMX_ERR_KMAX && maxk <= max_kmax) {  // larger tables, same block again
        (void)rmx_kpm_status(ctx_, nullptr);          // (its flags belong to the truncated tables)
        grow_tables(maxk, !first);
        carry_pos[0] = carry_pos0[0], carry_pos[1] = carry_pos0[1], carry_frag = carry_frag0;
        continue;
      }
      remap::check(kst, ctx_, "rmx_kpe_status");
      std::uint32_t kflags = 0;
      int mst = rmx_kpm_status(ctx_, &kflags);
      if (mst == RMX_ERR_CAPACITY && kflags == 128u) {
        // tickets the LDS matcher left to its global-memory instance (thousands of keypoints in
        // one region): computed there, offsets rewritten, positions derived again
        auto const* hd = first ? nullptr : halo_.get<rmx_desc>();
        auto const* hk = first ? nullptr : reinterpret_cast<std::uint32_t const*>(halo_.get<std::uint8_t>() + kmax_ * sizeof(rmx_desc));
        remap::check(rmx_kpm_resolve(ctx_, desc_.get<rmx_desc>(), kcount_.get<std::uint32_t>(), n, hd, hk, dxdy_.get<std::int32_t>(),
                                     matched_.get<std::uint8_t>(), nullptr),
                     ctx_, "rmx_kpm_resolve");
        carry_pos[0] = carry_pos0[0], carry_pos[1] = carry_pos0[1], carry_frag = carry_frag0;
        remap::check(rmx_frc_scan(ctx_, dxdy_.get<std::int32_t>(), matched_.get<std::uint8_t>(), n, first ? 1 : 0, carry_pos,
                                  &carry_frag, pos_.get<std::int32_t>(), frag_.get<std::uint32_t>()),
                     ctx_, "rmx_frc_scan");
        mst = RMX_OK;
      }
      remap::check(mst, ctx_, "rmx_kpm_status");
      break;
      }
      // the last table is the next block's halo
      remap::check(rmx_copy(ctx_, halo_.get(), desc_.get<rmx_desc>() + std::size_t(n - 1) * kmax_, kmax_ * sizeof(rmx_desc)),
                   ctx_, "rmx_copy");
      remap::check(rmx_copy(ctx_, halo_.get<std::uint8_t>() + kmax_ * sizeof(rmx_desc), kcount_.get<std::uint32_t>() + (n - 1), 4),
                   ctx_, "rmx_copy");
      clk.mark("device: kpe + kpm + scan");
      // ---- results to the host ----
      h_pos_.resize(2 * n);
      h_frag_.resize(n);
      h_k_.resize(n);
      h_med_.reserve(ctx_, std::size_t(n) * area);
      const std::uint32_t tstride = std::max<std::uint32_t>(1, std::min(maxk, kmax_));  // rows actually used
      h_tab_.resize(std::size_t(n) * tstride);
      remap::check(rmx_download(ctx_, h_pos_.data(), pos_.get(), 8 * n), ctx_, "rmx_download");
      remap::check(rmx_download(ctx_, h_frag_.data(), frag_.get(), 4 * n), ctx_, "rmx_download");
      remap::check(rmx_download(ctx_, h_k_.data(), kcount_.get(), 4 * n), ctx_, "rmx_download");
      remap::check(rmx_download_2d(ctx_, h_tab_.data(), tstride * sizeof(rmx_desc), desc_.get(), kmax_ * sizeof(rmx_desc),
                                   tstride * sizeof(rmx_desc), n),
                   ctx_, "rmx_download_2d");
      clk.mark("downloads");
      // ---- frame blobs: nic::compression is run on the device for the whole block (two
      // launches: images, medians); any other compressor is called per frame below ----
      constexpr bool device_codec = std::is_same_v<std::decay_t<Comp>, nic::compression>;
      std::vector<icd::compressed_t> pk_image, pk_median;
      if constexpr (device_codec) {
        pk_image = nic::detail::compress_device(ctx_, frames_.get<std::uint8_t>(), n, std::uint32_t(area));
        pk_median = nic::detail::compress_device(ctx_, median_.get<std::uint8_t>(), n, std::uint32_t(area));
      }
      clk.mark("codec");
      // ---- the per-frame host objects the callbacks borrow (median image, keypoint grid).  Both are made on
      // demand: the median is a view on the page-locked block that came back from the device, the grid
      // holds its descriptor rows and builds its maps when a region is first opened — a callback that
      // only counts frames (main.cpp:139-149) costs nothing per frame; one that reads them pays what the
      // eager construction cost (22 us per frame on one thread), in the callback ----
      all::memory_pool arena;
      std::vector<std::optional<image_type>> medians(n);
      std::vector<std::optional<grid_type>> grids(n);
      // (the block's medians cross PCIe in one copy, when the first of them is read: a host-side compressor does
      // that with frame 0, main.cpp's callback with nic::compression never does)
      mrl::cell_loader med_loader{[&] {
        remap::check(rmx_download(ctx_, h_med_.get(), median_.get(), std::size_t(n) * area), ctx_, "rmx_download");
      }};
      {
        const auto* med_src = reinterpret_cast<cpl::nat_cc const*>(h_med_.template get<std::uint8_t>());
        const auto* tab_src = h_tab_.data();
        for (std::uint32_t t = 0; t < n; ++t) {
          medians[t].emplace(image_type::view(dim_, med_src + std::size_t(t) * area, pixel_alloc_t{arena}, &med_loader));
          if (first && t == 0) continue;  // process_init has no callback (frc.hpp:83-95)
          grids[t].emplace(remap::detail::table_to_grid_deferred<grid_type>(tab_src + std::size_t(t) * tstride, h_k_[t],
                                                                            allocator_t<char>{arena}));
        }
      }
      clk.mark("host objects (median views, deferred grids)");
      // ---- device blit of the block, fragment by fragment (frc.hpp:129-135) ----
      for (std::uint32_t i = 0; i < n;) {
        std::uint32_t j = i;
        while (j < n && h_frag_[j] == h_frag_[i]) ++j;
        if (fresh || fragments_.empty() || h_frag_[i] != current_id_) add_fragment(h_frag_[i]);
        fresh = false;
        auto blit_run = [&](std::uint32_t a, std::uint32_t b) {
          // grow the canvas once for the whole run, then one batched blit
          for (std::uint32_t t = a; t < b; ++t) current_->ensure({h_pos_[2 * t], h_pos_[2 * t + 1]}, dim_);
          remap::check(rmx_fgm_blit_batch(ctx_, frames_.get<std::uint8_t>() + a * area, pos_.get<std::int32_t>() + 2 * a,
                                          nullptr, 0, b - a, current_->device_canvas(),
                                          std::uint32_t(current_->dim_.width_), std::uint32_t(current_->dim_.height_),
                                          current_->zero_.x_, current_->zero_.y_),
                       ctx_, "rmx_fgm_blit_batch");
        };
        if (!exact_callbacks_) blit_run(i, j);
        // ---- replay: bookkeeping + callback per frame, in order (frc.hpp:117-119) ----
        for (std::uint32_t t = i; t < j; ++t) {
          if (exact_callbacks_) {  // the callback sees the canvas exactly after frame t
            blit_run(t, t + 1);
            remap::check(rmx_sync(ctx_), ctx_, "rmx_sync");
          }
          image_type const& median = *medians[t];
          fgm::point_t p{h_pos_[2 * t], h_pos_[2 * t + 1]};
          if constexpr (device_codec)
            current_->frames_.emplace_back(
                fgm::frame{block[t].number_, p, fgm::packed_data{std::move(pk_image[t]), std::move(pk_median[t])}});
          else
            current_->frames_.emplace_back(
                fgm::frame{block[t].number_, p, fgm::packed_data{comp(block[t].image_), comp(median)}});
          if (first && t == 0) continue;  // process_init has no callback (frc.hpp:83-95)
          cb(*current_, block[t], median, *grids[t]);
        }
        i = j;
      }
      remap::check(rmx_sync(ctx_), ctx_, "rmx_sync");
      clk.mark("blit + replay (callbacks)");
      first = false;
      if (prefetch_feed_) pending.get();
      else fill(next_block, feed_pool_[cur ^ 1u]);
      clk.mark(prefetch_feed_ ? "feed.produce (what the helper thread had left)" : "feed.produce");
      std::swap(block, next_block);
      cur ^= 1u;
    }
  }

  fgm::fragment const& current() const noexcept { return *current_; }

  std::list<fgm::fragment> complete() noexcept {
    for (auto& f : fragments_) f.normalize();
    return std::move(fragments_);
  }

private:
  void add_fragment(std::uint32_t id) {  // frc.hpp:124-127
    current_ = &fragments_.emplace_back(dim_);
    current_id_ = id;
  }
  // A frame had `need` keypoints: switch to a context whose tables hold them (power of two,
  // at most max_kmax); the halo table moves to the new stride.  (Above 16 384 rows the tables of
  // a 4 096-frame block take gigabytes — 288 GB of HBM are there for that.)
  void grow_tables(std::uint32_t need, bool have_halo) {
    std::uint32_t k = kmax_;
    while (k < need) k <<= 1;
    k = std::min(k, max_kmax);
    rmx_ctx* c = remap::context_cache::get(std::uint32_t(dim_.width_), std::uint32_t(dim_.height_), 4, 2, 16, k);
    remap::device_buffer halo{c, k * sizeof(rmx_desc) + 16};
    if (have_halo) {
      remap::check(rmx_copy(c, halo.get(), halo_.get(), kmax_ * sizeof(rmx_desc)), c, "rmx_copy");
      remap::check(rmx_copy(c, halo.get<std::uint8_t>() + k * sizeof(rmx_desc),
                            halo_.get<std::uint8_t>() + kmax_ * sizeof(rmx_desc), 4),
                   c, "rmx_copy");
      remap::check(rmx_sync(c), c, "rmx_sync");
    }
    halo_ = std::move(halo);
    ctx_ = c;
    kmax_ = k;
  }
  void reserve(std::uint32_t n) {
    const std::size_t area = dim_.area();
    const std::uint32_t kmax = kmax_;
    frames_.reserve(ctx_, n * area);
    median_.reserve(ctx_, n * area);
    desc_.reserve(ctx_, std::size_t(n) * kmax * sizeof(rmx_desc));
    kcount_.reserve(ctx_, 4 * std::size_t(n));
    dxdy_.reserve(ctx_, 8 * std::size_t(n));
    matched_.reserve(ctx_, n);
    pos_.reserve(ctx_, 8 * std::size_t(n));
    frag_.reserve(ctx_, 4 * std::size_t(n));
    halo_.reserve(ctx_, kmax * sizeof(rmx_desc) + 16);
  }

  mrl::dimensions_t dim_;
  rmx_ctx* ctx_;
  std::uint32_t kmax_{kmax};
  bool exact_callbacks_{false};
  bool prefetch_feed_{false};
  std::uint32_t block_size_{block_frames};
  all::memory_pool feed_pool_[2];  // the frames of the block in flight and of the next one (feed.produce allocates from them)
  remap::device_buffer frames_, median_, desc_, kcount_, dxdy_, matched_, pos_, frag_, halo_;
  std::vector<std::int32_t> h_pos_;
  std::vector<std::uint32_t> h_frag_, h_k_;
  // page-locked staging (medians coming back, frames going in); shared by the collectors of
  // a thread: allocating 300 MB of pinned memory costs tens of milliseconds
  static inline thread_local remap::pinned_buffer h_med_, stage_;
  std::vector<rmx_desc> h_tab_;
  std::list<fgm::fragment> fragments_;
  fgm::fragment* current_{nullptr};
  std::uint32_t current_id_{0};
};

}  // namespace frc
