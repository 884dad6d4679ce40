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

#include <condition_variable>
#include <exception>
#include <mutex>
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
      : dim_{d}, ctx_{remap::context_cache::get(std::uint32_t(d.width_), std::uint32_t(d.height_), 4, 2, 16, kmax)} {
    if (char const* e = std::getenv("REMAP_BLOCK_FRAMES")) block_size_ = std::max<std::uint32_t>(1, std::uint32_t(std::atoi(e)));  // (tuning: tools/time_dropin.py)
  }

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
    // ---- Blocks in flight.  The feed is drained in blocks into `lanes` slots, each with its own context (a HIP
    // stream of its own), device buffers and page-locked staging.  A helper thread STAGES the blocks in feed
    // order — frames gathered into page-locked memory, one copy across PCIe, the extraction kernel, all enqueued
    // on the slot's stream — while the calling thread FINISHES the block before: matcher (the previous block's last
    // table is its halo), position scan, results to the host, canvas blit, and the callbacks, once per frame in
    // frame order (frc.hpp:119).  With three slots the gathering of block t + 2 runs beside the PCIe copy of block
    // t + 1 and the callbacks of block t.  Nothing else is concurrent: the feeder is called by ONE thread at a time
    // (the calling thread, blocks ahead of their callbacks; with prefetch_feed(true) the helper), the callbacks
    // by the calling thread only. ----
    std::mutex mu;
    std::condition_variable cv;
    bool stop = false;
    enum : int { slot_free = 0, slot_filled = 1, slot_ready = 2, slot_eof = 3 };
    for (auto& s : slots_) s.state = slot_free, s.error = nullptr, s.block.clear();
    auto fill = [&](slot_t& s) {
      s.block.clear();
      s.pool.reset();
      while (feed.has_more() && s.block.size() < block_size_) s.block.push_back(feed.produce(pixel_alloc_t{s.pool}));
    };
    const bool profile = std::getenv("REMAP_PROFILE") != nullptr;
    double t_fill = 0, t_reserve = 0, t_gather = 0, t_enqueue = 0, t_idle = 0;  // (the helper's own clock)
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto since = [](std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
    auto stage = [&](slot_t& s, std::uint32_t lane) {  // (helper thread) frames -> HBM, extraction enqueued
      const std::uint32_t n = std::uint32_t(s.block.size());
      auto t0 = now();
      std::uint32_t km;
      {
        std::lock_guard<std::mutex> lock(mu);
        km = kmax_;
      }
      s.n = n;
      reserve_slot(s, lane, n, km);
      t_reserve += since(t0), t0 = now();
      std::uint8_t* dst = s.stage.template get<std::uint8_t>();
      const unsigned gatherers = std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 8u);
      auto gather = [&, dst](unsigned w) {
        for (std::uint32_t i = w; i < n; i += gatherers) std::memcpy(dst + i * area, std::as_const(s.block[i].image_).data(), area);  // (const: a view stays a view)
      };
      std::vector<std::thread> helpers;
      for (unsigned w = 1; w < gatherers; ++w) helpers.emplace_back(gather, w);
      gather(0);
      for (auto& th : helpers) th.join();
      t_gather += since(t0), t0 = now();
      // (a kernel copy, not a DMA: the few hundred kilobytes the block before hands back would queue behind these
      // 300 MB on the same engine — measured: its downloads took 37 ms per block instead of 0.6)
      remap::check(rmx_upload_pinned(s.ctx, s.frames.template get<std::uint8_t>(), dst, std::size_t(n) * area), s.ctx, "rmx_upload_pinned");
      remap::check(rmx_kpe_batch(s.ctx, s.frames.template get<std::uint8_t>(), n, s.median.template get<std::uint8_t>(),
                                 s.desc.template get<rmx_desc>(), s.kcount.template get<std::uint32_t>()),
                   s.ctx, "rmx_kpe_batch");
      t_enqueue += since(t0);
    };
    const bool prefetch = prefetch_feed_;
    std::thread helper([&] {
      for (std::uint32_t b = 0;; ++b) {
        slot_t& s = slots_[b % lanes];
        {
          auto t0 = now();
          std::unique_lock<std::mutex> lock(mu);
          cv.wait(lock, [&] { return stop || s.state == (prefetch ? slot_free : slot_filled) || (!prefetch && s.state == slot_eof); });
          t_idle += since(t0);
          if (stop || s.state == slot_eof) break;
        }
        int next = slot_ready;
        try {
          auto t0 = now();
          if (prefetch) fill(s);
          t_fill += since(t0);
          if (s.block.empty()) next = slot_eof;
          else stage(s, b % lanes);
        } catch (...) {
          s.error = std::current_exception();
        }
        {
          std::lock_guard<std::mutex> lock(mu);
          s.state = next;
        }
        cv.notify_all();
        if (next == slot_eof || s.error) break;
      }
      if (profile)
        std::fprintf(stderr, "[remap] staging thread: feed.produce %.1f ms, buffers %.1f ms, gather %.1f ms, enqueue (copy + extraction) %.1f ms, waiting for a free slot %.1f ms\n",
                     t_fill * 1e3, t_reserve * 1e3, t_gather * 1e3, t_enqueue * 1e3, t_idle * 1e3);
    });
    struct joiner {  // the helper is stopped and joined on every way out of collect()
      std::thread& t;
      std::mutex& mu;
      std::condition_variable& cv;
      bool& stop;
      ~joiner() {
        {
          std::lock_guard<std::mutex> lock(mu);
          stop = true;
        }
        cv.notify_all();
        if (t.joinable()) t.join();
      }
    } join_helper{helper, mu, cv, stop};
    std::uint32_t produced = 0;  // blocks the calling thread has filled (without prefetch_feed)
    bool feed_done = false;
    clk.mark(nullptr);
    for (std::uint32_t b = 0;; ++b) {
      // the feed, on this thread: blocks ahead of their callbacks, never beside them
      while (!prefetch && !feed_done && produced < b + lanes) {
        slot_t& f = slots_[produced % lanes];  // (free: the block that held it was finished `lanes` blocks ago)
        fill(f);
        feed_done = f.block.empty();
        {
          std::lock_guard<std::mutex> lock(mu);
          f.state = feed_done ? slot_eof : slot_filled;
        }
        cv.notify_all();
        ++produced;
      }
      clk.mark("feed.produce");
      slot_t& s = slots_[b % lanes];
      {
        std::unique_lock<std::mutex> lock(mu);
        cv.wait(lock, [&] { return s.state == slot_ready || s.state == slot_eof; });
      }
      if (s.error) std::rethrow_exception(s.error);
      if (s.state == slot_eof) break;
      clk.mark("waiting for the staged block (gather, PCIe copy, extraction)");
      const std::uint32_t n = s.n;
      std::vector<frame_type>& block = s.block;
      std::uint32_t maxk = 0;
      const std::int32_t carry_pos0[2] = {carry_pos[0], carry_pos[1]};
      const std::uint32_t carry_frag0 = carry_frag;
      reserve(n);
      for (;;) {  // one pass, unless a frame has more keypoints than the tables hold
        if (s.kmax != kmax_) {  // the tables have grown since the block was staged: extracted again with the new stride
          reserve_slot(s, b % lanes, n, kmax_);
          remap::check(rmx_kpe_batch(s.ctx, s.frames.template get<std::uint8_t>(), n, s.median.template get<std::uint8_t>(),
                                     s.desc.template get<rmx_desc>(), s.kcount.template get<std::uint32_t>()),
                       s.ctx, "rmx_kpe_batch");
        }
        rmx_ctx* c = s.ctx;
        auto const* hd = first ? nullptr : halo_.get<rmx_desc>();
        auto const* hk = first ? nullptr : reinterpret_cast<std::uint32_t const*>(halo_.get<std::uint8_t>() + kmax_ * sizeof(rmx_desc));
        remap::check(rmx_kpm_seq(c, s.desc.template get<rmx_desc>(), s.kcount.template get<std::uint32_t>(), n, hd, hk,
                                 dxdy_.get<std::int32_t>(), matched_.get<std::uint8_t>()),
                     c, "rmx_kpm_seq");
        remap::check(rmx_frc_scan(c, dxdy_.get<std::int32_t>(), matched_.get<std::uint8_t>(), n, first ? 1 : 0,
                                  carry_pos, &carry_frag, pos_.get<std::int32_t>(), frag_.get<std::uint32_t>()),
                     c, "rmx_frc_scan");
        const int kst = rmx_kpe_status(c, &maxk);
        if (kst == RMX_ERR_KMAX && maxk <= max_kmax) {  // larger tables, same block again
          (void)rmx_kpm_status(c, nullptr);             // (its flags belong to the truncated tables)
          {
            std::lock_guard<std::mutex> lock(mu);  // (the helper reads the stride when it stages a block)
            grow_tables(maxk, !first);
          }
          carry_pos[0] = carry_pos0[0], carry_pos[1] = carry_pos0[1], carry_frag = carry_frag0;
          continue;
        }
        remap::check(kst, c, "rmx_kpe_status");
        std::uint32_t kflags = 0;
        int mst = rmx_kpm_status(c, &kflags);
        if (mst == RMX_ERR_CAPACITY && kflags == 128u) {
          // tickets the LDS matcher left to its global-memory instance (thousands of keypoints in
          // one region): computed there, offsets rewritten, positions derived again
          remap::check(rmx_kpm_resolve(c, s.desc.template get<rmx_desc>(), s.kcount.template get<std::uint32_t>(), n, hd, hk,
                                       dxdy_.get<std::int32_t>(), matched_.get<std::uint8_t>(), nullptr),
                       c, "rmx_kpm_resolve");
          carry_pos[0] = carry_pos0[0], carry_pos[1] = carry_pos0[1], carry_frag = carry_frag0;
          remap::check(rmx_frc_scan(c, dxdy_.get<std::int32_t>(), matched_.get<std::uint8_t>(), n, first ? 1 : 0, carry_pos,
                                    &carry_frag, pos_.get<std::int32_t>(), frag_.get<std::uint32_t>()),
                       c, "rmx_frc_scan");
          mst = RMX_OK;
        }
        remap::check(mst, c, "rmx_kpm_status");
        break;
      }
      rmx_ctx* const ctx = s.ctx;
      // the last table is the next block's halo
      remap::check(rmx_copy(ctx, halo_.get(), s.desc.template get<rmx_desc>() + std::size_t(n - 1) * kmax_, kmax_ * sizeof(rmx_desc)),
                   ctx, "rmx_copy");
      remap::check(rmx_copy(ctx, halo_.get<std::uint8_t>() + kmax_ * sizeof(rmx_desc), s.kcount.template get<std::uint32_t>() + (n - 1), 4),
                   ctx, "rmx_copy");
      clk.mark("device: kpm + scan");
      // ---- results to the host ----
      h_med_.reserve(ctx_, std::size_t(n) * area);
      const std::uint32_t tstride = std::max<std::uint32_t>(1, std::min(maxk, kmax_));  // rows actually used
      // (one page-locked block: positions | fragment ids | counts | tables cut to the rows in use; written by the
      // device itself, one wait)
      const std::size_t o_frag = 8 * std::size_t(n), o_k = o_frag + 4 * std::size_t(n), o_tab = (o_k + 4 * std::size_t(n) + 15) / 16 * 16;
      h_res_.reserve(ctx_, o_tab + std::size_t(n) * tstride * sizeof(rmx_desc));
      std::uint8_t* const res = h_res_.template get<std::uint8_t>();
      remap::check(rmx_download_pinned(ctx, res, pos_.get(), 8 * std::size_t(n)), ctx, "rmx_download_pinned");
      remap::check(rmx_download_pinned(ctx, res + o_frag, frag_.get(), 4 * std::size_t(n)), ctx, "rmx_download_pinned");
      remap::check(rmx_download_pinned(ctx, res + o_k, s.kcount.get(), 4 * std::size_t(n)), ctx, "rmx_download_pinned");
      remap::check(rmx_download_2d_pinned(ctx, res + o_tab, s.desc.get(), kmax_ * sizeof(rmx_desc), tstride * sizeof(rmx_desc), n),
                   ctx, "rmx_download_2d_pinned");
      remap::check(rmx_sync(ctx), ctx, "rmx_sync");
      const std::int32_t* const h_pos_ = reinterpret_cast<const std::int32_t*>(res);
      const std::uint32_t* const h_frag_ = reinterpret_cast<const std::uint32_t*>(res + o_frag);
      const std::uint32_t* const h_k_ = reinterpret_cast<const std::uint32_t*>(res + o_k);
      const rmx_desc* const h_tab = reinterpret_cast<const rmx_desc*>(res + o_tab);
      clk.mark("downloads");
      // ---- frame blobs: nic::compression is run on the device for the whole block (two
      // launches: images, medians); any other compressor is called per frame below ----
      constexpr bool device_codec = std::is_same_v<std::decay_t<Comp>, nic::compression>;
      std::vector<icd::compressed_t> pk_image, pk_median;
      if constexpr (device_codec) {
        pk_image = nic::detail::compress_device(ctx, s.frames.template get<std::uint8_t>(), n, std::uint32_t(area));
        pk_median = nic::detail::compress_device(ctx, s.median.template get<std::uint8_t>(), n, std::uint32_t(area));
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
        remap::check(rmx_download(ctx, h_med_.get(), s.median.get(), std::size_t(n) * area), ctx, "rmx_download");
      }};
      {
        const auto* med_src = reinterpret_cast<cpl::nat_cc const*>(h_med_.template get<std::uint8_t>());
        const auto* tab_src = h_tab;
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
        auto blit_run = [&](std::uint32_t a, std::uint32_t e) {
          // grow the canvas once for the whole run, then one batched blit
          current_->ensure_run(h_pos_ + 2 * a, e - a, dim_);
          remap::check(rmx_fgm_blit_batch(ctx, s.frames.template get<std::uint8_t>() + a * area, pos_.get<std::int32_t>() + 2 * a,
                                          nullptr, 0, e - a, current_->device_canvas(),
                                          std::uint32_t(current_->dim_.width_), std::uint32_t(current_->dim_.height_),
                                          current_->zero_.x_, current_->zero_.y_),
                       ctx, "rmx_fgm_blit_batch");
        };
        if (!exact_callbacks_) blit_run(i, j);
        // ---- replay: bookkeeping + callback per frame, in order (frc.hpp:117-119) ----
        for (std::uint32_t t = i; t < j; ++t) {
          if (exact_callbacks_) {  // the callback sees the canvas exactly after frame t
            blit_run(t, t + 1);
            remap::check(rmx_sync(ctx), ctx, "rmx_sync");
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
      remap::check(rmx_sync(ctx), ctx, "rmx_sync");
      clk.mark("blit + replay (callbacks)");
      first = false;
      {
        std::lock_guard<std::mutex> lock(mu);
        s.state = slot_free;  // its buffers and host frames may be reused
      }
      cv.notify_all();
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
  // one block in flight: its own context (stream), device buffers, page-locked staging and host frames
  static inline constexpr std::uint32_t lanes = 3;
  struct slot_t {
    rmx_ctx* ctx{nullptr};   // the lane's context for the current stride (a HIP stream of its own)
    rmx_ctx* home{nullptr};  // the context the slot's buffers were allocated with
    std::uint32_t kmax{0};  // stride the block's tables were extracted with
    std::uint32_t n{0};
    remap::device_buffer frames, median, desc, kcount;
    remap::pinned_buffer stage;
    std::vector<frame_type> block;
    all::memory_pool pool;  // feed.produce allocates the block's frames from it
    int state{0};
    std::exception_ptr error;
  };
  void reserve_slot(slot_t& s, std::uint32_t lane, std::uint32_t n, std::uint32_t kmax) {
    const std::size_t area = dim_.area();
    s.ctx = remap::context_cache::get_lane(std::uint32_t(dim_.width_), std::uint32_t(dim_.height_), 4, 2, 16, kmax, lane);
    s.kmax = kmax;
    // (allocations belong to the lane's first context whatever the stride grows to: a buffer follows the context it
    // is reserved with, and the frames already uploaded must survive a change of stride)
    s.home = s.home ? s.home : s.ctx;
    s.frames.reserve(s.home, n * area);
    s.median.reserve(s.home, n * area);
    s.desc.reserve(s.home, std::size_t(n) * kmax * sizeof(rmx_desc));
    s.kcount.reserve(s.home, 4 * std::size_t(n));
    s.stage.reserve(s.home, n * area);
  }
  void reserve(std::uint32_t n) {
    const std::uint32_t kmax = kmax_;
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
  slot_t slots_[lanes];
  remap::device_buffer dxdy_, matched_, pos_, frag_, halo_;  // (the calling thread's: matcher and scan outputs, the halo table)
  // page-locked block the medians come back into; shared by the collectors of a thread: allocating 300 MB of
  // pinned memory costs tens of milliseconds
  static inline thread_local remap::pinned_buffer h_med_, h_res_;  // (h_res_: a block's positions, fragment ids, counts and tables)
  std::list<fgm::fragment> fragments_;
  fgm::fragment* current_{nullptr};
  std::uint32_t current_id_{0};
};

}  // namespace frc
