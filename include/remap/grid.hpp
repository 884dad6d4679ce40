// remap/grid.hpp — keypoint representation (namespace kpr of the reference, kpr.hpp): the
// 13-byte code, per-region code -> points maps with weight counters, the WxH region grid.
// Host-side containers with the reference's names; the device analogue is built by
// kpm_build from rmx_desc tables (remap_amd/csrc/kpm.hip).
#pragma once

#include <array>
#include <atomic>
#include <cstddef>
#include <mutex>
#include <numeric>
#include <span>
#include <unordered_map>
#include <utility>
#include <vector>

#include "core.hpp"

namespace kpr {

inline constexpr std::uint8_t code_length = 13;
inline constexpr std::uint8_t code_max_index = code_length - 1;
using code = std::array<std::byte, code_length>;

// weight lives in the low nibble of the last byte (kpr.hpp:25-27)
inline std::byte weight(code const& c) noexcept { return c[code_max_index] & std::byte{0xf}; }

// byte-wise FNV-1a-style mix in 64-bit arithmetic (kpr.hpp:29-38)
struct code_hash {
  std::size_t operator()(code const& c) const noexcept {
    std::size_t h = 2166136261U;
    for (std::byte b : c) h = (h ^ static_cast<std::size_t>(b)) * 16777619U;
    return h;
  }
};

template <typename Alloc>
class region {
public:
  static inline constexpr std::size_t max_weight{3};
  using allocator_type = Alloc;
  using points_t = std::vector<mrl::point_t>;
  using points_alloc_t = all::rebind_alloc_t<allocator_type, std::pair<code const, points_t>>;
  using points_store = std::unordered_map<code, points_t, code_hash, std::equal_to<code>, points_alloc_t>;
  using count_store = std::array<std::size_t, max_weight>;

  region(allocator_type const& a) : by_code_{points_alloc_t{a}} {}
  region() : region(allocator_type{}) {}

  void add(code const& key, mrl::point_t const& p) {
    by_code_[key].push_back(p);
    ++per_weight_[static_cast<std::size_t>(weight(key))];
  }
  void clear() noexcept { by_code_.clear(); }
  points_store const& points() const noexcept { return by_code_; }
  count_store const& counts() const noexcept { return per_weight_; }
  std::size_t total_count() const noexcept { return std::accumulate(per_weight_.begin(), per_weight_.end(), std::size_t{}); }
  bool is_active() const noexcept {
    for (auto c : per_weight_)
      if (c) return true;
    return false;
  }
  allocator_type get_allocator() const noexcept { return by_code_.get_allocator(); }

private:
  points_store by_code_;
  count_store per_weight_{};
};

template <std::size_t Width, std::size_t Height, typename Alloc>
class grid {
  static_assert(Width > 0 && Height > 0);

public:
  static inline constexpr std::size_t width{Width};
  static inline constexpr std::size_t height{Height};
  static inline constexpr std::size_t region_count{Width * Height};
  using allocator_type = Alloc;
  using region_type = region<allocator_type>;
  using regions_t = std::span<region_type const, region_count>;
  // A grid may be handed a SOURCE instead of keypoints: `fill(grid, source, count)` adds them the first
  // time a region is looked at.  frc::collector gives every callback a grid (frc.hpp:119); most
  // callbacks — main.cpp's counts frames — never open it, and building eight unordered_maps per frame
  // was the largest share of a host-bound collect.  The source must outlive the grid or its first
  // use, whichever comes first; copies are made of the materialised grid.
  using fill_fn = void (*)(grid&, void const*, std::uint32_t);

  explicit grid(allocator_type const& a) : grid(a, std::make_index_sequence<region_count>{}) {}
  grid() : grid(allocator_type{}) {}
  grid(allocator_type const& a, fill_fn fill, void const* source, std::uint32_t count)
      : grid(a, std::make_index_sequence<region_count>{}) {
    fill_ = fill, source_ = source, count_ = count;
  }
  grid(grid const& o) : grid(o.materialised(), 0) {}  // (a copy may outlive the source: it is made of the keypoints)
  grid(grid&& o) : grid(std::move(o), 0) {}            // (a move keeps the deferred state: same lifetime contract)
  grid& operator=(grid o) {
    for (std::size_t i = 0; i < region_count; ++i) cells_[i] = std::move(o.cells_[i]);
    fill_ = o.fill_, source_ = o.source_, count_ = o.count_;
    return *this;
  }

  // add a keypoint to every region named in the index list (region index = height*col + row)
  template <std::size_t... I>
  void add(code const& key, mrl::point_t p, std::index_sequence<I...>) {
    (cells_[I].add(key, p), ...);
  }
  // (not noexcept: the first look at a deferred grid builds its maps)
  regions_t regions() const { return regions_t{materialised().cells_}; }
  region_type& operator[](std::size_t i) { return materialised_mut().cells_[i]; }
  region_type const& operator[](std::size_t i) const { return materialised().cells_[i]; }
  bool deferred() const noexcept { return fill_ != nullptr; }

private:
  template <std::size_t... I>
  grid(allocator_type const& a, std::index_sequence<I...>) : cells_{((void)I, region_type{a})...} {}
  grid(grid const& src, int) : grid(src.cells_[0].get_allocator(), std::make_index_sequence<region_count>{}) {
    for (std::size_t i = 0; i < region_count; ++i) cells_[i] = src.cells_[i];
  }
  grid(grid&& src, int) : grid(src.cells_[0].get_allocator(), std::make_index_sequence<region_count>{}) {
    for (std::size_t i = 0; i < region_count; ++i) cells_[i] = std::move(src.cells_[i]);
    fill_ = src.fill_, source_ = src.source_, count_ = src.count_;
    src.fill_ = nullptr;
  }
  // The deferred state is a cache of what the grid already "holds", filled through const access — possibly by
  // several threads of one callback at once (the reference's callers match regions under std::execution::par):
  // one of them fills under a lock, the others wait; a grid that is materialised costs one relaxed-ish load.
  static std::mutex& fill_mutex() {
    static std::mutex m;
    return m;
  }
  grid& materialised_mut() const {
    grid& self = const_cast<grid&>(*this);
    if (std::atomic_ref<fill_fn>(self.fill_).load(std::memory_order_acquire)) {
      std::lock_guard<std::mutex> lock(fill_mutex());
      if (fill_fn f = self.fill_) {
        f(self, source_, count_);
        std::atomic_ref<fill_fn>(self.fill_).store(nullptr, std::memory_order_release);
      }
    }
    return self;
  }
  grid const& materialised() const { return materialised_mut(); }

  region_type cells_[region_count];
  fill_fn fill_{nullptr};
  void const* source_{nullptr};
  std::uint32_t count_{0};
};

template <typename G>
concept gridlike = requires(G g) {
  typename G::allocator_type;
  requires G::width > 0;
  requires G::height > 0;
  { g.add(std::declval<code&&>(), std::declval<mrl::point_t&&>(), std::index_sequence<0>{}) };
};

}  // namespace kpr
